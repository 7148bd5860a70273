"""y = LayerNorm(Linear(x)) for the box-location embeddings (k_in = 6 -> 768) on libgps_hip.so
(gps_loc_embed_forward / backward): the `loc_layers[0]` Sequential of the object encoder and the unified encoder
(reference modules/vision/pcd_openvocab_encoder.py:64-66, modules/grounding/unified_encoder.py:28-30) as one launch per
direction instead of a 6-deep library GEMM, torch's LayerNorm pair and the bias / weight-gradient reductions.
GPU fp32 parameters, hidden size 768, inputs without gradient; everything else runs the Sequential itself."""
from __future__ import annotations

import torch
from torch import nn

from ... import _native

_ENABLED = True


def set_fused_loc(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


class _LocEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, eps: float):
        k_in, d = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, k_in).float().contiguous()
        n = x2.shape[0]
        w32, g32, bt32 = w.contiguous(), gamma.contiguous(), beta.contiguous()
        b32 = b.contiguous() if b is not None else None
        y = torch.empty((n, d), dtype=torch.float32, device=x.device)
        mean = torch.empty(n, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n, dtype=torch.float32, device=x.device)
        from ...pointnet2._ext import _timed
        with torch.cuda.device(x.device), _timed(f"loc_embed_forward(rows={n},k={k_in},d={d})", n * (d + k_in) * 4):
            st = _native.load().gps_loc_embed_forward(
                n, k_in, d, x2.data_ptr(), w32.data_ptr(), b32.data_ptr() if b32 is not None else None, g32.data_ptr(),
                bt32.data_ptr(), float(eps), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                torch.cuda.current_stream(x.device).cuda_stream)
        _native.check(st, "loc_embed_forward")
        ctx.save_for_backward(x2, w32, b32, g32, mean, rstd)
        ctx.out_shape = x.shape[:-1] + (d,)
        return y.view(ctx.out_shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, b32, g32, mean, rstd = ctx.saved_tensors
        n, k_in = x2.shape
        d = w32.shape[0]
        dy2 = dy.reshape(n, d).float().contiguous()
        lib = _native.load()
        parts = int(lib.gps_loc_embed_partial_rows(n))
        scratch = torch.empty((parts, k_in + 3, d), dtype=torch.float32, device=x2.device)
        sums = torch.empty((k_in + 3, d), dtype=torch.float32, device=x2.device)
        from ...pointnet2._ext import _timed
        with torch.cuda.device(x2.device), _timed(f"loc_embed_backward(rows={n},k={k_in},d={d})", n * (d + k_in) * 4):
            st = lib.gps_loc_embed_backward(
                n, k_in, d, dy2.data_ptr(), x2.data_ptr(), w32.data_ptr(), b32.data_ptr() if b32 is not None else None,
                g32.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scratch.data_ptr(), sums.data_ptr(),
                torch.cuda.current_stream(x2.device).cuda_stream)
        _native.check(st, "loc_embed_backward")
        dw = sums[:k_in].t().contiguous()                        # (d, k_in)
        return None, dw, (sums[k_in] if b32 is not None else None), sums[k_in + 1], sums[k_in + 2], None


def supported(seq: nn.Module, x: torch.Tensor) -> bool:
    if not (_ENABLED and isinstance(seq, nn.Sequential) and len(seq) == 2 and isinstance(seq[0], nn.Linear)
            and isinstance(seq[1], nn.LayerNorm)):
        return False
    lin, ln = seq[0], seq[1]
    return (x.is_cuda and not x.requires_grad and x.shape[-1] == lin.in_features and lin.in_features in (3, 6, 8)
            and lin.out_features == 768 and lin.weight.dtype == torch.float32 and lin.weight.is_cuda
            and tuple(ln.normalized_shape) == (768,) and ln.elementwise_affine and ln.bias is not None
            and ln.weight.dtype == torch.float32)


def loc_embed(seq: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """seq(x) for seq = Sequential(Linear(k_in, 768), LayerNorm(768)); fp32 result on the fused path (what LayerNorm
    returns under autocast as well)."""
    if not supported(seq, x):
        return seq(x)
    lin, ln = seq[0], seq[1]
    return _LocEmbed.apply(x, lin.weight, lin.bias, ln.weight, ln.bias, ln.eps)
