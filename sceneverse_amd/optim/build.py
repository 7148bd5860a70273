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
    flat = [p for g in params for p in (g["params"] if isinstance(g, dict) else [g])]
    on_gpu = bool(flat) and all(p.is_cuda for p in flat)
    # AdamW on the GPU: libgps_hip.so's clip + AdamW pass (optim/fused_adamw.py), same update rule and state keys;
    # `native_optimizer: False` in the optimizer arguments keeps torch's implementation (A/B runs)
    native = args.pop("native_optimizer", True)
    if cls is optim.AdamW and on_gpu and native and not args.get("amsgrad", False) \
            and all(p.dtype.is_floating_point and p.element_size() == 4 for p in flat):
        from .fused_adamw import GpsAdamW
        return GpsAdamW(params, **args)
    # torch's multi-tensor implementation elsewhere: one launch per group instead of ~10 foreach launches, only
    # when every parameter already lives on the GPU (the fused implementation requires it)
    if "fused" not in args and "foreach" not in args and cls in (optim.AdamW, optim.Adam) and on_gpu:
        args["fused"] = True
    return cls(params, **args)


def build_optim(cfg, params, total_steps):
    loss = Loss(cfg)
    optimizer = get_optimizer(cfg, params)
    scheduler = get_scheduler(cfg, optimizer, total_steps)
    return loss, optimizer, scheduler
