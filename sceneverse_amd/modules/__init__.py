from .build import (GROUNDING_REGISTRY, HEADS_REGISTRY, LANGUAGE_REGISTRY, VISION_REGISTRY,
                    build_module, build_module_by_name)
from . import grounding, heads, language, vision  # noqa: F401  (populate the registries)
