# Test-infrastructure stub (NOT product code): attribute-dict stand-in for omegaconf.
import re

import yaml


class _Loader(yaml.SafeLoader):
    """YAML 1.2 floats: OmegaConf reads ``1e-3`` as a float, PyYAML's YAML 1.1 resolver as a string."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?|\.[0-9_]+(?:[eE][-+]?[0-9]+)?|[0-9][0-9_]*[eE][-+]?[0-9]+"
               r"|\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"),
    list("-+0123456789."))


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return DictConfig({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _unwrap(x):
    if isinstance(x, dict):
        return {k: _unwrap(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_unwrap(v) for v in x]
    return x


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.load(f, Loader=_Loader))

    @staticmethod
    def create(d):
        return _wrap(d)

    @staticmethod
    def to_container(cfg, **kw):
        return _unwrap(cfg)
