"""One GPS training step on MI355X: the body of the reference's `DefaultTrainer.train_step` /
`backward` (trainer/default_trainer.py:18-24,30-48) without its logging, as a reusable object.

    forward(data_dict) -> Loss -> backward -> clip_grad_norm_(grad_norm) -> AdamW.step -> LambdaLR.step

Data parallelism is the reference's only strategy (Accelerate -> torch DDP over NCCL,
trainer/build.py:66-75,121).  Here: one process per GPU, `torch.distributed` backend "nccl"
(= RCCL over xGMI), torch DDP with gradient buckets viewed in place and all-reduced while backward
is still running; `find_unused_parameters=True` for the same reason as the reference (13 trainable
tensors never receive a gradient).

bf16: the transformer stack, BERT, heads and losses run under `torch.autocast(bfloat16)`;
the point ops always compute in fp32 (indices must be bit-exact).
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch
import torch.nn as nn

from .common import dist_utils
from .model.build import build_model
from .optim.build import build_optim


class GPSTrainStep:
    def __init__(self, cfg, device: torch.device | str = "cuda", total_steps: int = 100000,
                 amp_dtype: Optional[torch.dtype] = torch.bfloat16, ddp: Optional[bool] = None,
                 bucket_cap_mb: int = 64, seed: int = 42):
        self.cfg = cfg
        self.device = torch.device(device)
        torch.manual_seed(seed)
        self.model = build_model(cfg).to(self.device)
        self.loss, self.optimizer, self.scheduler = build_optim(cfg, self.model.get_opt_params(),
                                                                total_steps)
        self.loss = self.loss.to(self.device)
        self.grad_norm = cfg.solver.get("grad_norm", None)
        self.amp_dtype = amp_dtype if self.device.type == "cuda" else None
        world = dist_utils.get_world_size()
        use_ddp = (world > 1) if ddp is None else ddp
        self.net: nn.Module = self.model
        if use_ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            kw = dict(find_unused_parameters=True, gradient_as_bucket_view=True,
                      bucket_cap_mb=bucket_cap_mb, broadcast_buffers=False)
            if self.device.type == "cuda":
                self.net = DDP(self.model, device_ids=[self.device.index], **kw)
            else:
                self.net = DDP(self.model, **kw)
        self.global_step = 0

    def _autocast(self):
        if self.amp_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast(device_type="cuda", dtype=self.amp_dtype)

    def forward_loss(self, data_dict):
        with self._autocast():
            out = self.net(data_dict)
            total, losses = self.loss(out)
        return out, total, losses

    def step(self, data_dict):
        """One optimisation step; returns (total_loss tensor, dict of loss tensors).  No host sync."""
        self.net.train()
        data_dict['cur_step'] = self.global_step
        data_dict['total_steps'] = 1 << 30
        out, total, losses = self.forward_loss(data_dict)
        self.optimizer.zero_grad(set_to_none=True)
        total.backward()
        if self.grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_norm)
        self.optimizer.step()
        self.scheduler.step()
        self.global_step += 1
        return total.detach(), {k: v.detach() for k, v in losses.items()}

    @torch.no_grad()
    def evaluate(self, data_dict):
        self.net.eval()
        out, total, losses = self.forward_loss(data_dict)
        return out, total, losses


def scanrefer_accuracy(og3d_logits: torch.Tensor, iou25_onehot: torch.Tensor,
                       iou50_onehot: torch.Tensor) -> dict:
    """acc@0.25 / acc@0.5 exactly as the reference's ScanReferEval.batch_metrics computes them
    (evaluator/scanrefer_eval.py:14-87): the arg-max object counts as correct when its one-hot
    entry in `tgt_object_id_iou25/50` is set."""
    pred = torch.argmax(og3d_logits, dim=-1)
    pick = pred[:, None]
    hit25 = torch.gather(iou25_onehot, 1, pick).squeeze(1).bool()
    hit50 = torch.gather(iou50_onehot, 1, pick).squeeze(1).bool()
    n = float(max(1, pred.shape[0]))
    return {"og_acc_iou25": hit25.sum().item() / n, "og_acc_iou50": hit50.sum().item() / n}
