"""bf16 shadow weights for every nn.Linear of a mixed-precision step (engine-level optimisation).

Under `torch.autocast(bfloat16)` each F.linear casts its fp32 weight (and bias) to bf16 on the way in
and casts the bf16 weight gradient back to fp32 on the way out: ~280 tiny launches and ~1.5 ms per GPS
step (profiles/r1/bench_s_kernel_stats.csv: bfloat16_copy / bfloat16tofloat32_copy).  Here the bf16
copies of all Linear parameters live in one cache that is refreshed by ONE multi-tensor copy per step
(`refresh_all`, captured at the head of the HIP graph), F.linear is routed through an autograd
function that reads the shadows, and the weight gradient GEMM writes fp32 directly
(bf16 x bf16 -> fp32 accumulate/out), so neither direction needs a per-parameter cast.
Same arithmetic as autocast: bf16 operands, fp32 accumulation, fp32 master weights and gradients.

Scope: only active inside `with shadow_linear():` (GPSTrainStep.forward_loss); outside, F.linear is
untouched.  A shadow whose parameter changed since the last refresh (optimizer step, load_state_dict)
is re-cast on use, so the context is safe without an explicit refresh too.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

_ORIG_LINEAR = F.linear
_SHADOWS: dict = {}            # id(param) -> [param, bf16 copy, version]
_ACTIVE = False
_MM_OUT_DTYPE = None           # does torch.mm(..., out_dtype=float32) work here? decided on first use


def _shadow(p: torch.Tensor) -> torch.Tensor:
    ent = _SHADOWS.get(id(p))
    if ent is None or ent[0] is not p:
        ent = [p, p.detach().to(torch.bfloat16), p._version]
        _SHADOWS[id(p)] = ent
    elif ent[2] != p._version:
        ent[1].copy_(p.detach())
        ent[2] = p._version
    return ent[1]


def refresh_all() -> None:
    """One multi-tensor cast of every registered parameter into its shadow (graph-capturable)."""
    ents = [e for e in _SHADOWS.values() if e[0].is_cuda]
    if not ents:
        return
    with torch.no_grad():
        torch._foreach_copy_([e[1] for e in ents], [e[0].detach() for e in ents])
    for e in ents:
        e[2] = e[0]._version


def clear() -> None:
    _SHADOWS.clear()


def _wgrad(dy2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """dy2^T @ x2 with fp32 output from bf16 operands (one GEMM, no cast kernel)."""
    global _MM_OUT_DTYPE
    if _MM_OUT_DTYPE is None:
        try:
            torch.mm(dy2[:8].t(), x2[:8], out_dtype=torch.float32)
            _MM_OUT_DTYPE = True
        except Exception:  # noqa: BLE001 -- older torch / backend without the kwarg
            _MM_OUT_DTYPE = False
    if _MM_OUT_DTYPE:
        return torch.mm(dy2.t(), x2, out_dtype=torch.float32)
    return torch.mm(dy2.t(), x2).float()


class _ShadowLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, w16, b16):
        x16 = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        y = _ORIG_LINEAR(x16, w16, b16)
        ctx.save_for_backward(x16, w16)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, w16 = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != torch.bfloat16:
            dy2 = dy2.to(torch.bfloat16)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, w16).view(*dy.shape[:-1], w16.shape[1])
        if ctx.needs_input_grad[1]:
            dw = _wgrad(dy2, x16.reshape(-1, x16.shape[-1]))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0, dtype=torch.float32)
        return dx, dw, db, None, None


def _linear(x, w, b=None):
    if (_ACTIVE and isinstance(w, torch.nn.Parameter) and w.is_cuda and w.dtype == torch.float32
            and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() >= 2
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and (b is None or (isinstance(b, torch.nn.Parameter) and b.dtype == torch.float32))):
        return _ShadowLinear.apply(x, w, b, _shadow(w), _shadow(b) if b is not None else None)
    return _ORIG_LINEAR(x, w, b)


@contextlib.contextmanager
def shadow_linear(enabled: bool = True):
    """Route F.linear through the bf16 shadow cache for the duration of the block."""
    global _ACTIVE
    if not enabled:
        yield
        return
    prev, prev_fn = _ACTIVE, F.linear
    _ACTIVE = True
    F.linear = _linear
    torch.nn.functional.linear = _linear
    try:
        yield
    finally:
        _ACTIVE = prev
        F.linear = prev_fn
        torch.nn.functional.linear = prev_fn
