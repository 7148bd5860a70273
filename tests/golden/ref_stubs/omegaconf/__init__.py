"""Stub of omegaconf for importing the reference offline (make_golden.py only)."""
from contextlib import contextmanager


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


class OmegaConf:
    @staticmethod
    def to_container(cfg, resolve=True):
        return _plain(cfg)

    @staticmethod
    def create(d):
        return d


@contextmanager
def open_dict(cfg):
    yield cfg
