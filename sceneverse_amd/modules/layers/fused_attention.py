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
MAX_LEN = 512


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
        # a bool tensor is one byte per element holding 0 / 1: the kernels read it in place (no conversion launch)
        m8 = None
        if mask is not None:
            m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
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
        ctx.save_for_backward(packed, sw, pl, m8, lse, seed_dev, out)
        ctx.meta = (n_head, float(p_drop), int(seed), spatial)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        packed, sw, pl, m8, lse, seed_dev, out = ctx.saved_tensors
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
                out.data_ptr(), gbase, gbase + D * esz, gbase + 2 * D * esz, _ptr(dsw), _stream())
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
        # the dropout stream lives ON THE DEVICE (_SeedStream): this call's forward+backward read one
        # word of a per-device block.  No host value is baked into the launch, so a captured HIP graph
        # draws a fresh mask on every replay.
        seed_dev = _next_device_seed(packed.device)
    return _FusedSelfAttention.apply(packed.contiguous(), pairwise_locs, key_padding_mask, n_head, p, seed,
                                     seed_dev)


_SEED_INC = 0x632BE59BD9B4E019        # odd 63-bit increment; the stream wraps modulo 2^64
_SEED_BLOCK = 256
_SEED_STATE = {}


def _wrap64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


class _SeedStream:
    """Per-device stream of dropout seed words kept ON THE DEVICE.  A block of _SEED_BLOCK consecutive
    stream values lives in one int64 tensor; each call hands out a one-element VIEW of it (no launch).
    `begin_step` advances the whole block in place with one tiny kernel -- inside a captured HIP graph
    that kernel is replayed, so every replay draws fresh masks.  Exhausting a block mid-step builds a
    NEW tensor from the old one's last word (never in place: views of the old block may be saved for a
    pending backward)."""

    def __init__(self, device: torch.device):
        init = int(torch.randint(0, 2 ** 62, (1,)).item())          # honours torch.manual_seed
        steps = [_wrap64(init + _SEED_INC * (i + 1)) for i in range(_SEED_BLOCK)]
        self.offsets = torch.tensor([_wrap64(_SEED_INC * (i + 1)) for i in range(_SEED_BLOCK)],
                                    dtype=torch.int64, device=device)
        self.block = torch.tensor(steps, dtype=torch.int64, device=device)
        self.pos = 0

    def begin_step(self) -> None:
        if self.pos:
            self.block.add_(_wrap64(_SEED_INC * _SEED_BLOCK))
            self.pos = 0

    def next(self) -> torch.Tensor:
        if self.pos == _SEED_BLOCK:
            self.block = self.block[-1] + self.offsets
            self.pos = 0
        word = self.block[self.pos:self.pos + 1]
        self.pos += 1
        return word


def _stream_for(device: torch.device) -> _SeedStream:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _SEED_STATE.get(device)
    if st is None:
        st = _SEED_STATE[device] = _SeedStream(device)
    return st


def begin_step(device) -> None:
    """Advance the device's dropout seed block.  Call once at the start of every training step, INSIDE the
    region a HIP graph captures (sceneverse_amd/engine.py does); plain eager use without it stays correct
    (seeds are never reused), it only costs a block rebuild every _SEED_BLOCK dropout calls."""
    _stream_for(device).begin_step()


def _next_device_seed(device: torch.device) -> torch.Tensor:
    return _stream_for(device).next()
