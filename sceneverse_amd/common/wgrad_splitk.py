"""Split-K weight gradients for the big-token Linears of a mixed-precision step (engine-level).

dW = dY^T X reduces over the TOKENS (19 200 for the scene-caption BERT, 8 320 joint, 5 120 spatial)
into a small (out x in) matrix: 36-144 output tiles for 256 CUs with both operands reduction-major.
hipBLASLt's best single-GEMM solutions reach 0.28-0.65 PFLOP/s there against 0.9-1.57 for the forward
GEMMs of the same layers (profiles/tunableop_gfx9500.csv).  Here the token axis is split over a batch
dimension -- one library bmm over S chunks with fp32 partials, then an fp32 sum -- which fills the chip
and hands autograd an fp32 gradient directly (no bf16 -> fp32 cast launch); measured per shape in
profiles/r1/splitk_probe.txt (10-35 % off the tuned single GEMM).  Same arithmetic class as autocast's
own path: bf16 operands, fp32 accumulation -- the partial sums are simply kept in fp32 longer.

The bias gradient of the same calls (db = column sums of dY, up to 19 200 x 3 072 bf16) goes to
libgps_hip.so's gps_colsum_bf16 (two-stage, deterministic, HBM-bound) instead of torch's generic reduce.

Mechanism: inside `with splitk_wgrad():` F.linear is routed, for eligible calls, through
    y = F.linear(x16, W.detach(), b.detach()) # forward and dX exactly as before
    y = _AttachWGrad.apply(y, x16, W, b)      # identity in forward; backward adds dW = splitk(dY, x16)
                                              # and db = colsum(dY)
so nothing about the forward pass or the input gradient changes.  Outside the context (and for small
or unsupported shapes) F.linear is untouched.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

_ORIG_LINEAR = F.linear
_ACTIVE = False
MIN_TOKENS = 5000            # split the token reduction of dW from here on
MIN_TOKENS_BIAS = 2048       # route the bias gradient through gps_colsum_bf16 from here on (dW unsplit)


def pick_splits(tokens: int, n_out: int, n_in: int) -> int:
    """Token chunks for dW (n_out x n_in) reduced over `tokens`: about 144 workgroups of one 256 x 256
    output tile each (what profiles/r1/splitk_probe.txt favours on 256 CUs), a power of two <= 16 that
    divides the token count and leaves >= 512 tokens per chunk.  1 = leave the call alone."""
    if tokens < MIN_TOKENS or n_out < 256 or n_in < 256:
        return 1
    tiles = max(1, (n_out * n_in) // (256 * 256))
    want = min(16, 144 // tiles)
    s = 1
    while s * 2 <= want and tokens % (s * 2) == 0 and tokens // (s * 2) >= 512:
        s *= 2
    return s


def splitk_wgrad_mm(dy2: torch.Tensor, x2: torch.Tensor, splits: int) -> torch.Tensor:
    """dy2 (T, N), x2 (T, K) bf16 -> dy2^T @ x2 as fp32 (N, K), the reduction split into `splits` chunks."""
    T, N = dy2.shape
    K = x2.shape[1]
    if splits <= 1:                      # the library's (tuned) single GEMM, as autograd would run it
        return torch.mm(dy2.t(), x2).float()
    a = dy2.view(splits, T // splits, N).transpose(1, 2)
    b = x2.view(splits, T // splits, K)
    return torch.bmm(a, b, out_dtype=torch.float32).sum(0)


def colsum_bf16(dy2: torch.Tensor) -> torch.Tensor:
    """(T, N) bf16 -> (N,) fp32 column sums on libgps_hip.so (gps_colsum_bf16)."""
    from .. import _native
    lib = _native.load()
    T, N = dy2.shape
    out = torch.empty(N, dtype=torch.float32, device=dy2.device)
    parts = lib.gps_colsum_parts(T, N)
    scratch = torch.empty((max(parts, 1), N), dtype=torch.float32, device=dy2.device)
    from ..pointnet2._ext import _timed
    with torch.cuda.device(dy2.device), _timed(f"colsum_bf16(rows={T},cols={N})", 2 * T * N + 4 * N):
        st = lib.gps_colsum_bf16(T, N, dy2.data_ptr(), dy2.stride(0), scratch.data_ptr(), out.data_ptr(),
                                 torch.cuda.current_stream(dy2.device).cuda_stream)
    if st == _native.GPS_ERR_UNSUPPORTED:            # columns / pitch not a multiple of 8: not this kernel's shape
        return dy2.sum(0, dtype=torch.float32)
    _native.check(st, "colsum_bf16")
    return out


class _AttachWGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, x16, w, b, splits):
        ctx.save_for_backward(x16)
        ctx.splits = splits
        ctx.has_bias = b is not None
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        (x16,) = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != torch.bfloat16:
            dy2 = dy2.to(torch.bfloat16)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dw = splitk_wgrad_mm(dy2, x16.reshape(-1, x16.shape[-1]), ctx.splits) if ctx.needs_input_grad[2] else None
        db = colsum_bf16(dy2) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return dy, None, dw, db, None


def _linear(x, w, b=None):
    if (_ACTIVE and torch.is_tensor(w) and w.requires_grad and w.is_cuda and w.dtype == torch.float32
            and w.dim() == 2 and (b is None or (b.dtype == torch.float32 and b.dim() == 1)) and x.is_cuda and x.dim() >= 2 and x.dtype in (torch.bfloat16, torch.float32)
            and torch.is_grad_enabled() and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        tokens = x.numel() // x.shape[-1]
        splits = pick_splits(tokens, w.shape[0], w.shape[1])
        if splits > 1 or (b is not None and tokens >= MIN_TOKENS_BIAS and w.shape[0] % 8 == 0):
            x16 = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            if not x16.is_contiguous():
                x16 = x16.contiguous()
            y = _ORIG_LINEAR(x16, w.detach(), b.detach() if b is not None else None)
            return _AttachWGrad.apply(y, x16, w, b, splits)
    return _ORIG_LINEAR(x, w, b)


@contextlib.contextmanager
def splitk_wgrad(enabled: bool = True):
    """Route eligible F.linear calls through the split-K weight-gradient path for the block's duration
    (the backward pass may run after the block ends: the routing decision is taken in forward)."""
    global _ACTIVE
    if not enabled:
        yield
        return
    prev, prev_fn = _ACTIVE, F.linear
    _ACTIVE = True
    F.linear = _linear
    try:
        yield
    finally:
        _ACTIVE = prev
        F.linear = prev_fn
