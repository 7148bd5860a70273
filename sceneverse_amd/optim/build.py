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
    args = dict(cfg2dict(cfg.solver.optim.args))
    # same update rule, one multi-tensor launch per group instead of ~10 foreach launches: only
    # when every parameter already lives on the GPU (the fused implementation requires it)
    flat = [p for g in params for p in (g["params"] if isinstance(g, dict) else [g])]
    if "fused" not in args and "foreach" not in args and cls in (optim.AdamW, optim.Adam) \
            and flat and all(p.is_cuda for p in flat):
        args["fused"] = True
    return cls(params, **args)


def build_optim(cfg, params, total_steps):
    loss = Loss(cfg)
    optimizer = get_optimizer(cfg, params)
    scheduler = get_scheduler(cfg, optimizer, total_steps)
    return loss, optimizer, scheduler
