"""Minimal attribute/dict config node standing in for OmegaConf's DictConfig (not installable
here).  The reference reads configs both as attributes (`cfg.model.vision.name`) and through
`.get(key, default)`, and converts sub-trees to kwargs with `cfg2dict` (common/type_utils.py:6-7).
Real OmegaConf nodes are accepted everywhere too: `cfg2dict` dispatches on type."""
from __future__ import annotations

from typing import Any, Mapping


class ConfigNode(dict):
    """dict with attribute access; nested mappings are wrapped recursively."""

    def __init__(self, data: Mapping[str, Any] | None = None, **kw: Any) -> None:
        super().__init__()
        for k, v in {**(dict(data) if data else {}), **kw}.items():
            self[k] = v

    @staticmethod
    def _wrap(v: Any) -> Any:
        if isinstance(v, ConfigNode):
            return v
        if isinstance(v, Mapping):
            return ConfigNode(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigNode._wrap(x) for x in v)
        return v

    def __setitem__(self, k: str, v: Any) -> None:
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v

    def __delattr__(self, k: str) -> None:
        del self[k]

    def to_dict(self) -> dict:
        def un(v: Any) -> Any:
            if isinstance(v, Mapping):
                return {kk: un(vv) for kk, vv in v.items()}
            if isinstance(v, (list, tuple)):
                return [un(x) for x in v]
            return v
        return un(self)


def cfg2dict(cfg: Any) -> dict:
    """Sub-tree -> plain kwargs dict (reference: OmegaConf.to_container(cfg, resolve=True))."""
    if cfg is None:
        return {}
    if isinstance(cfg, ConfigNode):
        return cfg.to_dict()
    try:  # a real OmegaConf node, when the caller's environment has it
        from omegaconf import OmegaConf  # type: ignore
        if OmegaConf.is_config(cfg):
            return OmegaConf.to_container(cfg, resolve=True)
    except ImportError:
        pass
    return dict(cfg)
