"""AdamW + gradient clipping on libgps_hip.so (include/gps_hip.h gps_adamw_step): the optimizer half of the
reference's training step

    accelerator.clip_grad_norm_(model.parameters(), grad_norm); optimizer.step()     trainer/default_trainer.py:18-24
    torch.optim.AdamW built by name with the config's arguments                       optim/optimizer/optim.py:9-14

as three launches over all parameter tensors instead of ~50 foreach launches -- and the same pass writes the bf16
shadows (and packed fp32 bias copies) the MFMA GEMMs read (modules/layers/gemm.py), so no weight is ever re-cast
in the forward pass.  Same hyper-parameters, parameter-group keys and state keys ('step', 'exp_avg', 'exp_avg_sq')
as torch.optim.AdamW: state dicts are interchangeable.  GPU parameters only (fp32); learning rates live in device
tensors that torch's LR schedulers fill in place, which also makes the step capturable in a HIP graph.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from .. import _native


class _TensorRec(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("shadow_bf16", ctypes.c_void_p), ("mirror_f32", ctypes.c_void_p),
                ("numel", ctypes.c_longlong), ("group", ctypes.c_int), ("step_slot", ctypes.c_int)]


class _GroupRec(ctypes.Structure):
    _fields_ = [("lr_dev", ctypes.c_void_p), ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float), ("reserved", ctypes.c_int)]


class GpsAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False,
                 maximize=False, capturable=True, **unused):
        if amsgrad or maximize:
            raise NotImplementedError("GpsAdamW implements the reference's configuration: amsgrad=False, maximize=False")
        for k in unused:
            if k not in ("fused", "foreach", "differentiable"):
                raise TypeError(f"unexpected optimizer argument {k!r}")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        capturable=True)
        super().__init__(params, defaults)
        dev = None
        for g in self.param_groups:
            for p in g["params"]:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise ValueError("GpsAdamW needs fp32 parameters on the GPU")
                dev = dev or p.device
                if p.device != dev:
                    raise ValueError("all parameters must live on one GPU")
        self._device = dev
        for g in self.param_groups:
            if not torch.is_tensor(g["lr"]):
                # float `initial_lr` keeps LambdaLR's arithmetic on the host; the live value is a device word
                g.setdefault("initial_lr", float(g["lr"]))
                g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
        self._scalars = torch.zeros(2, dtype=torch.float32, device=dev)   # clip coefficient, grad norm
        # per-parameter step counts (torch.optim.AdamW semantics), one persistent slot per parameter
        self._slot = {}
        for g in self.param_groups:
            for p in g["params"]:
                self._slot.setdefault(id(p), len(self._slot))
        self._steps = torch.zeros(max(len(self._slot), 1), dtype=torch.float32, device=dev)
        self._sig = None
        self._tables = None
        self._keep = []          # device tables referenced by a captured graph must outlive it
        # pinned staging area for tables built DURING a HIP-graph capture (no host allocation is allowed there, and
        # the captured copy re-reads its source at every replay: regions are handed out once and never reused)
        self._pinned = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
        self._pinned_off = 0

    # torch.optim.AdamW-compatible state ('step' is a 0-dim view of this parameter's slot in one device array)
    def _init_state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = self._steps[self._slot[id(p)]]
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @property
    def last_grad_norm(self) -> torch.Tensor:
        """Total L2 norm of the gradients of the last step (what clip_grad_norm_ returns), a device scalar."""
        return self._scalars[1]

    def load_state_dict(self, state_dict):
        # checked BEFORE anything is touched: a failed load must not leave a half-loaded optimizer behind
        if self._keep:
            raise RuntimeError("GpsAdamW.load_state_dict after a HIP-graph capture: the captured tables point at the old "
                               "learning-rate / state words -- load the checkpoint before the first captured step")
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            if "step" in st:
                self._steps[self._slot[id(p)]] = float(st["step"])
                st["step"] = self._steps[self._slot[id(p)]]
        for g in self.param_groups:
            # a checkpoint loaded with map_location='cpu' carries lr as a CPU tensor: the kernel dereferences the
            # word on the device, so it must live there (fp32, 0-dim)
            g["lr"] = self._device_lr(g["lr"])
        self._sig = None

    def _device_lr(self, lr) -> torch.Tensor:
        if torch.is_tensor(lr) and lr.device == self._device and lr.dtype == torch.float32:
            return lr
        return torch.full((), float(lr), dtype=torch.float32, device=self._device)

    def _build_tables(self, entries, targets):
        lib = _native.load()
        chunk = int(lib.gps_adamw_chunk_elems())
        trec = (_TensorRec * len(entries))()
        chunks = []
        for i, (p, gi) in enumerate(entries):
            st = self._init_state(p)
            sh, mir = targets.get(id(p), (None, None))
            if sh is not None and sh.numel() != p.numel():      # never write past a buffer that is not this tensor's
                sh = None
            if mir is not None and mir.numel() != p.numel():
                mir = None
            r = trec[i]
            r.param, r.grad = p.data_ptr(), p.grad.data_ptr()
            r.exp_avg, r.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            r.shadow_bf16 = sh.data_ptr() if sh is not None else None
            r.mirror_f32 = mir.data_ptr() if mir is not None else None
            r.numel, r.group, r.step_slot = p.numel(), gi, self._slot[id(p)]
            chunks.extend((i, c) for c in range((p.numel() + chunk - 1) // chunk))
        grec = (_GroupRec * len(self.param_groups))()
        for gi, g in enumerate(self.param_groups):
            r = grec[gi]
            r.lr_dev, r.lr = g["lr"].data_ptr(), float(g.get("initial_lr", 0.0))
            r.beta1, r.beta2 = float(g["betas"][0]), float(g["betas"][1])
            r.eps, r.weight_decay = float(g["eps"]), float(g["weight_decay"])
        host = [np.frombuffer(bytes(trec), dtype=np.uint8).copy(), np.frombuffer(bytes(grec), dtype=np.uint8).copy(),
                np.asarray(chunks, dtype=np.int32).reshape(-1)]
        capturing = torch.cuda.is_current_stream_capturing()
        dev_tabs = []
        for a in host:
            t = torch.from_numpy(a)
            if capturing:
                nbytes = (t.numel() * t.element_size() + 63) // 64 * 64
                if self._pinned_off + nbytes > self._pinned.numel():
                    raise RuntimeError("GpsAdamW: pinned staging area exhausted (too many table rebuilds inside captures)")
                stage = self._pinned[self._pinned_off:self._pinned_off + t.numel() * t.element_size()].view(t.dtype)
                self._pinned_off += nbytes
                stage.copy_(t)
                d = torch.empty_like(t, device=self._device)
                d.copy_(stage, non_blocking=True)
            else:
                d = t.to(self._device)
            dev_tabs.append(d)
        partial = torch.empty(max(len(chunks), 1), dtype=torch.float32, device=self._device)
        tabs = (dev_tabs[0], dev_tabs[1], dev_tabs[2], partial, len(chunks), len(entries))
        if capturing:
            self._keep.append(tabs)
        return tabs

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        """One AdamW update of every parameter that has a gradient; with max_grad_norm the gradients are first
        scaled by min(1, max_grad_norm / (total L2 norm + 1e-6)) inside the same pass (the reference's
        clip_grad_norm_ + step; the .grad tensors themselves are left untouched)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from ..modules.layers import gemm
        entries = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    p.grad = p.grad.float().contiguous()
                entries.append((p, gi))
        if not entries:
            return loss
        for g in self.param_groups:                 # a plain float (g["lr"] = 1e-4) or a CPU tensor assigned from outside
            g["lr"] = self._device_lr(g["lr"])      # moves onto the device: the kernel reads learning rates from device words
        sig = (gemm.registry_version(), tuple(g["lr"].data_ptr() for g in self.param_groups),
               tuple((p.data_ptr(), p.grad.data_ptr()) for p, _ in entries))
        if sig != self._sig:
            self._tables = self._build_tables(entries, gemm.shadow_targets())
            self._sig = sig
        tens, groups, chunks, partial, n_chunks, n_tensors = self._tables
        with torch.cuda.device(self._device):
            from ..pointnet2._ext import _timed
            n = sum(p.numel() for p, _ in entries)
            with _timed(f"adamw_step(params={n})", 28 * n + (4 * n if max_grad_norm else 0)):
                st = _native.load().gps_adamw_step(n_tensors, n_chunks, tens.data_ptr(), groups.data_ptr(), chunks.data_ptr(),
                                                   float(max_grad_norm) if max_grad_norm else 0.0, partial.data_ptr(),
                                                   self._scalars.data_ptr(), self._steps.data_ptr(),
                                                   torch.cuda.current_stream(self._device).cuda_stream)
        _native.check(st, "adamw_step")
        return loss
