"""`build_optim(cfg, params, total_steps) -> (loss, optimizer, scheduler)` (reference
optim/build.py:6-10, optimizer lookup optim/optimizer/optim.py:9-14)."""
import torch.optim as optim

from ..common.config import cfg2dict
from .loss import Loss
from .scheduler import get_scheduler


def get_optimizer(cfg, params):
    cls = getattr(optim, cfg.solver.optim.name, None)
    if cls is None:
        raise KeyError(f"unknown optimizer {cfg.solver.optim.name}")
    return cls(params, **cfg2dict(cfg.solver.optim.args))


def build_optim(cfg, params, total_steps):
    loss = Loss(cfg)
    optimizer = get_optimizer(cfg, params)
    scheduler = get_scheduler(cfg, optimizer, total_steps)
    return loss, optimizer, scheduler
