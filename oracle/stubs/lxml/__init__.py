# Test-infrastructure stub (NOT product code): lets the read-only reference at
# /root/reference import in this container, where the real lxml is absent.
from xml.etree import ElementTree as etree  # noqa: F401
