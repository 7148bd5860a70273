"""The four module registries and builders -- the plugin API of the reference's
modules/build.py:6-31, kept name for name so that `model/*.py` and the trainer resolve
`cfg.name` exactly as before: `REG.get(cfg.name)(cfg, **cfg2dict(cfg.args))`."""
from ..common.config import cfg2dict
from ..common.registry import Registry

VISION_REGISTRY = Registry("vision")
LANGUAGE_REGISTRY = Registry("language")
GROUNDING_REGISTRY = Registry("grounding")
HEADS_REGISTRY = Registry("heads")

_BY_TYPE = {
    "vision": VISION_REGISTRY,
    "language": LANGUAGE_REGISTRY,
    "grounding": GROUNDING_REGISTRY,
    "heads": HEADS_REGISTRY,
}


def build_module(module_type, cfg):
    registry = _BY_TYPE.get(module_type)
    if registry is None:
        raise NotImplementedError(f"module type {module_type} not implemented")
    return registry.get(cfg.name)(cfg, **cfg2dict(cfg.args))


def build_module_by_name(cfg):
    for registry in _BY_TYPE.values():
        if cfg.name in registry:
            print(f"Using {cfg.name} module from Registry {registry._name}")
            kwargs = cfg2dict(cfg.args) if hasattr(cfg, "args") else {}
            return registry.get(cfg.name)(cfg, **kwargs)
    raise NotImplementedError(f"Unknown module: {cfg.name}")
