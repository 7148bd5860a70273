"""Model registry and builder -- the upper drop-in boundary (reference model/build.py:5-18):
`build_model(cfg)` = `MODEL_REGISTRY.get(cfg.model.name)(cfg)`; models are nn.Modules with
`forward(data_dict) -> data_dict` and `get_opt_params()`."""
import torch.nn as nn

from ..common.registry import Registry

MODEL_REGISTRY = Registry("model")


class BaseModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()

    def get_opt_params(self):
        raise NotImplementedError("Function to obtain all default parameters for optimization")


def build_model(cfg):
    return MODEL_REGISTRY.get(cfg.model.name)(cfg)
