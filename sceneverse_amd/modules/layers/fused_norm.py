"""y = LayerNorm(x + dropout(h)) on libgps_hip.so (gps_add_dropout_layernorm_forward/backward):
the post-norm residual step of the reference's encoder layers
(modules/layers/transformers.py:143-153, :311-315) as one launch per direction.
GPU tensors with a 256-multiple width <= 2048; everything else takes the torch formulation."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from ... import _native

_ENABLED = True


def set_fused_norm(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


_REDUCE_SCRATCH = {}


def _reduce_scratch(device: torch.device, d: int) -> torch.Tensor:
    """Zero-initialised scratch of gps_ln_reduce_partials (second-level partial rows + arrival counters), one per
    (device, width): every call leaves it zero again, and the launches of one device share one stream."""
    key = (device.index, d)
    buf = _REDUCE_SCRATCH.get(key)
    if buf is None:
        nbytes = int(_native.load().gps_ln_reduce_scratch_bytes(d))
        buf = _REDUCE_SCRATCH[key] = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=device)
    return buf


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


def _row_fraction(rows_dev: Optional[torch.Tensor], n: int):
    """bench accounting: share of the static row count a launch with a device-side row count really touches."""
    from ...pointnet2._ext import profiling
    if rows_dev is None or not profiling() or n == 0:
        return None
    snap = rows_dev.detach().clone()
    return lambda: min(1.0, max(0.0, float(snap.item()) / n))


class SharedPostGrad:
    """Gradient of an addend that enters SEVERAL layers' LayerNorm launches as `post` (the reference re-adds the location /
    type embeddings in front of every layer): the layers' backward launches -- which run last layer first -- build it in ONE
    buffer (the first to run stores, the others add, gps_add_dropout_layernorm_backward_post_acc) and only the launch
    marked `final` (the FIRST layer in forward order: its backward runs after all the others') hands the buffer to autograd;
    the others report no gradient for the addend.  Replaces one buffer + zero-fill per layer and autograd's adds.
    One object per forward call of the encoder (a fresh one every call: nothing survives a failed or partial pass).  A backward
    pass must run ALL the sharing layers' nodes, the final one last -- true whenever the gradient is taken with respect to
    anything at or below the first layer (every training step); a pass cut ABOVE the first layer (torch.autograd.grad
    towards an intermediate activation only) would leave the sum unreported, so such callers pass no `post_share`."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _AddDropoutLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, gamma, beta, eps: float, p_drop: float, seed_dev, want_bf16: bool, rows_dev=None, post=None,
                post_share=None):
        ctx.post_share = post_share                         # (SharedPostGrad, final: bool) or None
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        h2 = h.reshape(-1, d).contiguous()
        n = x2.shape[0]
        post2 = post.reshape(-1, d).float().contiguous() if post is not None else None
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        y = torch.empty_like(x2)
        y16 = torch.empty((n, d), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
        mean = torch.empty(n, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n, dtype=torch.float32, device=x.device)
        from ...pointnet2._ext import _timed
        nbytes = n * d * (2 * x2.element_size() + h2.element_size())
        with torch.cuda.device(x.device), _timed(f"add_dropout_layernorm_forward(rows={n},d={d})", nbytes,
                                                 work_fraction=_row_fraction(rows_dev, n)):
            st = _native.load().gps_add_dropout_layernorm_forward_post(
                n, d, int(x2.dtype == torch.bfloat16), int(h2.dtype == torch.bfloat16), x2.data_ptr(),
                h2.data_ptr(), g32.data_ptr(), b32.data_ptr(), float(eps), float(p_drop), 0, _ptr(seed_dev),
                y.data_ptr(), _ptr(y16), mean.data_ptr(), rstd.data_ptr(), _ptr(rows_dev), _ptr(post2),
                torch.cuda.current_stream().cuda_stream)
        _native.check(st, "add_dropout_layernorm_forward")
        ctx.save_for_backward(x2, h2, g32, mean, rstd, seed_dev, rows_dev)
        ctx.meta = (float(p_drop), x.shape, h.shape, gamma.dtype, beta.dtype)
        ctx.ln_params = (gamma, beta)                       # the parameter OBJECTS (deferred dgamma / dbeta, see backward)
        ctx.post = (post.shape, post.dtype) if post is not None else None
        if want_bf16:
            return y.view(x.shape), y16.view(x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dy16=None):
        x2, h2, g32, mean, rstd, seed_dev, rows_dev = ctx.saved_tensors
        p_drop, x_shape, h_shape, g_dtype, b_dtype = ctx.meta
        n, d = x2.shape
        if dy is None:
            dy = torch.zeros(x_shape, dtype=x2.dtype, device=x2.device)
        dy2 = dy.reshape(n, d).to(x2.dtype).contiguous()
        dy16_2 = dy16.reshape(n, d).to(torch.bfloat16).contiguous() if dy16 is not None else None
        dx = torch.empty_like(x2)
        dh = torch.empty_like(h2)
        want_dpost = ctx.post is not None and ctx.needs_input_grad[9]
        share, final = ctx.post_share if (ctx.post_share is not None and want_dpost) else (None, True)
        acc = 0
        if share is not None and share.buf is not None:
            dpost, acc = share.buf, 1                          # a later layer's launch started it
        else:
            dpost = torch.empty((n, d), dtype=torch.float32, device=x2.device) if want_dpost else None
            if dpost is not None and rows_dev is not None:
                dpost.zero_()                                  # rows past the device count are not written
            if share is not None:
                share.buf = dpost
        lib = _native.load()
        parts = int(lib.gps_ln_partial_rows(n))
        part = torch.empty((2, parts, d), dtype=torch.float32, device=x2.device)
        from ...pointnet2._ext import _timed
        nbytes = n * d * (3 * x2.element_size() + 2 * h2.element_size())
        with torch.cuda.device(x2.device), _timed(f"add_dropout_layernorm_backward(rows={n},d={d})", nbytes,
                                                  work_fraction=_row_fraction(rows_dev, n)):
            st = lib.gps_add_dropout_layernorm_backward_post_acc(
                n, d, int(x2.dtype == torch.bfloat16), int(h2.dtype == torch.bfloat16), dy2.data_ptr(),
                _ptr(dy16_2), x2.data_ptr(), h2.data_ptr(), g32.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_drop, 0,
                _ptr(seed_dev), dx.data_ptr(), dh.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), _ptr(rows_dev),
                _ptr(dpost), acc, torch.cuda.current_stream().cuda_stream)
        _native.check(st, "add_dropout_layernorm_backward")
        if share is not None:
            if final:
                share.buf = None                               # handed to autograd below; the next pass starts afresh
            else:
                dpost = None                                   # the final launch reports the sum
        if dpost is not None:
            dpost = dpost.view(ctx.post[0]).to(ctx.post[1])
        # dgamma / dbeta = column sums of the per-workgroup partial rows.  Inside gemm.grouped_wgrads() they are not
        # reduced here: every LayerNorm of the pass hands its partial rows to ONE reduce launch when the block is left
        # (gps_ln_reduce_partials_grouped writes gamma.grad / beta.grad); otherwise one reduce launch per LayerNorm.
        from . import gemm
        if gemm.defer_ln_param_grads(part, parts, d, *ctx.ln_params, needs=(ctx.needs_input_grad[2], ctx.needs_input_grad[3])):
            return (dx.view(x_shape), dh.view(h_shape), None, None, None, None, None, None, None, dpost, None)
        sums = torch.empty((2, d), dtype=torch.float32, device=x2.device)
        with torch.cuda.device(x2.device):
            st = lib.gps_ln_reduce_partials(parts, d, part.data_ptr(), sums.data_ptr(),
                                            _reduce_scratch(x2.device, d).data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
        _native.check(st, "ln_reduce_partials")
        return (dx.view(x_shape), dh.view(h_shape), sums[0].to(g_dtype), sums[1].to(b_dtype),
                None, None, None, None, None, dpost, None)


def supported(x: torch.Tensor, h: torch.Tensor, norm: nn.LayerNorm) -> bool:
    d = x.shape[-1]
    return (_ENABLED and x.is_cuda and h.is_cuda and x.shape == h.shape and d in (256, 512, 768, 1024, 2048)
            and x.dtype in (torch.float32, torch.bfloat16) and h.dtype in (torch.float32, torch.bfloat16)
            and isinstance(norm, nn.LayerNorm) and tuple(norm.normalized_shape) == (d,)
            and norm.elementwise_affine and norm.bias is not None)


def add_dropout_layer_norm(x: torch.Tensor, h: torch.Tensor, norm: nn.LayerNorm, p_drop: float = 0.0,
                           training: bool = False, want_bf16: bool = False, rows_dev: Optional[torch.Tensor] = None,
                           post: Optional[torch.Tensor] = None, post_share=None):
    """norm(x + dropout(h, p_drop, training)); y has x's dtype on the fused path.
    want_bf16: also return a bf16 copy of y written by the same launch (what the next GEMM reads
    under autocast; its gradient is added inside the fused backward) -> (y, y_bf16).
    post: optional addend of x's shape applied BEHIND the normalisation, y = norm(...) + post (and y_bf16 = bf16 of that
    sum): the per-layer `x + loc_embeds` / `joint + extra` of the NEXT encoder layer folded into this launch.
    post_share: (SharedPostGrad, final) when the SAME `post` tensor enters several layers: see SharedPostGrad (final = this is
    the first of those layers in forward order).
    rows_dev: int32 device word = number of leading rows (of the flattened (rows, d) view) that carry work; the other
    rows are neither read nor written (their content is undefined) and do not enter the dgamma / dbeta sums."""
    p = float(p_drop) if training else 0.0
    if post is not None and not (post.shape == x.shape and x.dtype == torch.float32 and post.is_cuda == x.is_cuda):
        raise ValueError("add_dropout_layer_norm: `post` must have the shape of x (fp32 rows)")
    if not supported(x, h, norm):
        if rows_dev is not None:
            raise RuntimeError("add_dropout_layer_norm: a device-side row count needs the fused kernel")
        y = norm(x + F.dropout(h, p, training=p > 0.0))
        if post is not None:
            y = y + post
        return (y, y) if want_bf16 else y
    seed_dev = None
    if p > 0.0:
        from .fused_attention import _next_device_seed
        seed_dev = _next_device_seed(x.device)
    return _AddDropoutLN.apply(x, h, norm.weight, norm.bias, norm.eps, p, seed_dev, bool(want_bf16), rows_dev, post,
                               post_share if post is not None else None)


class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps: float):
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        n = x2.shape[0]
        y = torch.empty_like(x2)
        inv = torch.empty(n, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = _native.load().gps_l2_normalize_forward(n, d, x2.data_ptr(), float(eps), y.data_ptr(), inv.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream)
        _native.check(st, "l2_normalize_forward")
        ctx.save_for_backward(y, inv)
        ctx.eps = float(eps)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        n, d = y.shape
        dy2 = dy.reshape(n, d).float().contiguous()
        dx = torch.empty_like(y)
        with torch.cuda.device(y.device):
            st = _native.load().gps_l2_normalize_backward(n, d, dy2.data_ptr(), y.data_ptr(), inv.data_ptr(), ctx.eps,
                                                          dx.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "l2_normalize_backward")
        return dx.view(dy.shape), None


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, p=2, dim=-1, eps=eps); one launch per direction for fp32 GPU rows of up to 2048 elements."""
    d = x.shape[-1]
    if not (_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1 and d % 4 == 0 and d <= 2048 and x.numel() > 0):
        return F.normalize(x, dim=-1, p=2, eps=eps)
    return _L2Normalize.apply(x, eps)


class _AddRow(torch.autograd.Function):
    """x (..., d) + row (d): the row's gradient is the column sum of dy over all leading dimensions, taken by
    gps_ln_reduce_partials in a fixed order (torch: a generic reduce_kernel, 45 - 60 us for 3200 - 5120 rows)."""

    @staticmethod
    def forward(ctx, x, row):
        ctx.row_dtype = row.dtype
        return x + row

    @staticmethod
    def backward(ctx, dy):
        return dy, column_sum(dy).to(ctx.row_dtype)


class _BroadcastRow(torch.autograd.Function):
    """row (d,) -> (*lead, d) contiguous: one copy launch instead of a zero fill + an add; gradient = column sum."""

    @staticmethod
    def forward(ctx, row, lead):
        return row.expand(*lead, row.shape[0]).contiguous()

    @staticmethod
    def backward(ctx, dy):
        return column_sum(dy).to(dy.dtype), None


def broadcast_row(row: torch.Tensor, lead) -> torch.Tensor:
    """`add_row(zeros(*lead, d), row)` without the zeros: the same values, the same (deterministic) row gradient."""
    lead = tuple(int(v) for v in lead)
    n = 1
    for v in lead:
        n *= v
    if not (_ENABLED and row.is_cuda and row.dtype == torch.float32 and row.dim() == 1 and row.shape[0] % 4 == 0 and n >= 2):
        return add_row(row.new_zeros((*lead, row.shape[0])), row)
    return _BroadcastRow.apply(row, lead)


def column_sum(x: torch.Tensor) -> torch.Tensor:
    """Sum over every dimension but the last -> (d,), fp32, deterministic."""
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    n = x2.shape[0]
    if not (_ENABLED and x2.is_cuda and x2.dtype == torch.float32 and n >= 2 and d % 4 == 0):
        return x2.float().sum(0)
    x2 = x2.contiguous()
    half = n // 2                                               # the reducer sums two stacked sets of `half` rows
    sums = torch.empty((2, d), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = _native.load().gps_ln_reduce_partials(half, d, x2.data_ptr(), sums.data_ptr(), _reduce_scratch(x.device, d).data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
    _native.check(st, "ln_reduce_partials")
    out = sums[0] + sums[1]
    return out + x2[2 * half] if n & 1 else out


def add_row(x: torch.Tensor, row: torch.Tensor) -> torch.Tensor:
    """x + row with `row` (d,) broadcast over the leading dimensions (autograd: column-sum kernel for the row)."""
    if not (_ENABLED and x.is_cuda and x.dtype == torch.float32 and row.dim() == 1 and row.shape[0] == x.shape[-1]
            and row.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.numel() // x.shape[-1] >= 2):
        return x + row
    return _AddRow.apply(x, row)
