"""Test suite of theseus_amd (a regular package so that it wins over other top-level ``tests`` packages on sys.path)."""
