"""Autograd front-end of libgps_hip.so's fused self-attention core (include/gps_hip.h
gps_attn_forward / gps_attn_backward): everything the reference computes between the QKV
projections and the output projection of

    MultiHeadAttentionSpatial.forward, fusion 'cond'   modules/layers/transformers.py:193-239
    nn.MultiheadAttention (key_padding_mask, dropout)  modules/layers/transformers.py:141

in one launch per direction.  The input is the PACKED projection output

    packed (B, L, 3*D [+ H*6])  =  [ q | k | v [| per-head (bias, w_1..w_5)] ]      bf16

so the projections are one GEMM, and the gradient comes back as one tensor of the same shape (the
kernel writes dq/dk/dv straight into their column blocks: no chunk/cat copies in either direction).
GPU + bf16 only; there is no CPU path (the torch formulation in transformers.attention_core is used
for CPU tensors and fp32 inputs).
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import _native

HEAD_DIM = 64
SPATIAL_VEC = 6
MAX_LEN = 256


def supported(d_model: int, n_head: int, length: int) -> bool:
    return d_model == n_head * HEAD_DIM and 0 < length <= MAX_LEN


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


class _FusedSelfAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed: torch.Tensor, pl: Optional[torch.Tensor], mask: Optional[torch.Tensor],
                n_head: int, p_drop: float, seed: int, seed_dev: Optional[torch.Tensor]) -> torch.Tensor:
        B, L, W = packed.shape
        D = n_head * HEAD_DIM
        spatial = pl is not None
        assert W == 3 * D + (n_head * SPATIAL_VEC if spatial else 0), (W, D, spatial)
        assert packed.is_cuda and packed.dtype == torch.bfloat16 and packed.is_contiguous()
        sw = packed[..., 3 * D:].float().contiguous() if spatial else None
        if spatial:
            pl = pl.float().contiguous()
            assert pl.shape == (B, L, L, 5), pl.shape
        m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
        out = torch.empty((B, L, D), dtype=torch.bfloat16, device=packed.device)
        lse = torch.empty((B, n_head, L), dtype=torch.float32, device=packed.device)
        base, esz = packed.data_ptr(), packed.element_size()
        from ...pointnet2._ext import _timed
        # algorithmic work: q,k,v,out once (+ pairwise/cond vector), 2 x L x L x 64 MACs per head
        nbytes = 2 * B * L * 4 * D + (B * L * L * 5 * 4 + B * L * n_head * 6 * 4 if spatial else 0)
        with torch.cuda.device(packed.device), _timed(f"attn_forward(L={L},spatial={int(spatial)})", nbytes,
                                                      4 * B * n_head * L * L * HEAD_DIM, "bf16"):
            st = _native.load().gps_attn_forward(
                B, n_head, L, HEAD_DIM, base, base + D * esz, base + 2 * D * esz, W,
                _ptr(sw), _ptr(pl), _ptr(m8), float(p_drop), int(seed), _ptr(seed_dev), out.data_ptr(), D,
                lse.data_ptr(), _stream())
        _native.check(st, "attn_forward")
        ctx.save_for_backward(packed, sw, pl, m8, lse, seed_dev)
        ctx.meta = (n_head, float(p_drop), int(seed), spatial)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        packed, sw, pl, m8, lse, seed_dev = ctx.saved_tensors
        n_head, p_drop, seed, spatial = ctx.meta
        B, L, W = packed.shape
        D = n_head * HEAD_DIM
        dout = dout.to(torch.bfloat16).contiguous()
        dpacked = torch.empty_like(packed)
        dsw = torch.empty_like(sw) if spatial else None
        base, esz = packed.data_ptr(), packed.element_size()
        gbase = dpacked.data_ptr()
        from ...pointnet2._ext import _timed
        nbytes = 2 * B * L * 8 * D + (B * L * L * 5 * 4 + 2 * B * L * n_head * 6 * 4 if spatial else 0)
        with torch.cuda.device(packed.device), _timed(f"attn_backward(L={L},spatial={int(spatial)})", nbytes,
                                                      10 * B * n_head * L * L * HEAD_DIM, "bf16"):
            st = _native.load().gps_attn_backward(
                B, n_head, L, HEAD_DIM, base, base + D * esz, base + 2 * D * esz, W,
                _ptr(sw), _ptr(pl), _ptr(m8), p_drop, seed, _ptr(seed_dev), dout.data_ptr(), D, lse.data_ptr(),
                gbase, gbase + D * esz, gbase + 2 * D * esz, _ptr(dsw), _stream())
        _native.check(st, "attn_backward")
        if spatial:
            dpacked[..., 3 * D:] = dsw
        return dpacked, None, None, None, None, None, None


def fused_self_attention(packed: torch.Tensor, n_head: int, pairwise_locs: Optional[torch.Tensor] = None,
                         key_padding_mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0,
                         training: bool = False) -> torch.Tensor:
    """packed (B, L, 3*D [+ H*6]) bf16 -> (B, L, D) bf16 attention output (heads merged)."""
    p = float(dropout_p) if training else 0.0
    seed, seed_dev = 0, None
    if p > 0.0:
        # the dropout stream lives ON THE DEVICE: a per-device uint64 advanced by one tiny kernel per
        # call and snapshotted for this call's forward+backward.  No host value is baked into the
        # launch, so a captured HIP graph draws a fresh mask on every replay.
        seed_dev = _next_device_seed(packed.device)
    return _FusedSelfAttention.apply(packed.contiguous(), pairwise_locs, key_padding_mask, n_head, p, seed,
                                     seed_dev)


_SEED_STATE = {}


def _next_device_seed(device: torch.device) -> torch.Tensor:
    state = _SEED_STATE.get(device)
    if state is None:
        init = int(torch.randint(0, 2 ** 62, (1,)).item())          # honours torch.manual_seed
        state = torch.tensor([init], dtype=torch.int64, device=device)
        _SEED_STATE[device] = state
    state.add_(0x632BE59BD9B4E019)       # odd 63-bit increment; wraps modulo 2^64
    return state.clone()
