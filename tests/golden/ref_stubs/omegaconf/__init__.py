"""Stub of omegaconf for importing the reference offline (make_golden.py only)."""
from contextlib import contextmanager


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def _plain_yaml(v):
    if isinstance(v, dict):
        return {str(k): _plain_yaml(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain_yaml(x) for x in v]
    return v if isinstance(v, (int, float, str, bool, type(None))) else str(v)


class OmegaConf:
    @staticmethod
    def to_container(cfg, resolve=True, **kw):
        return _plain(cfg)

    @staticmethod
    def create(d):
        return d

    @staticmethod
    def to_yaml(cfg):
        import yaml
        return yaml.safe_dump(_plain_yaml(cfg), sort_keys=False)


@contextmanager
def open_dict(cfg):
    yield cfg
