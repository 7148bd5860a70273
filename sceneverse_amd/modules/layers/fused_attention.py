"""Autograd front-end of libgps_hip.so's fused self-attention core (include/gps_hip.h
gps_attn_forward / gps_attn_backward): everything the reference computes between the QKV
projections and the output projection of

    MultiHeadAttentionSpatial.forward, fusion 'cond'   modules/layers/transformers.py:193-239
    nn.MultiheadAttention (key_padding_mask, dropout)  modules/layers/transformers.py:141

in one launch per direction.  The input is the PACKED projection output

    packed (B, L, 3*D [+ H*6])  =  [ q | k | v [| per-head (bias, w_1..w_5)] ]      bf16

so the projections are one GEMM, and the gradient comes back as one tensor of the same shape (the
kernel writes dq/dk/dv straight into their column blocks: no chunk/cat copies in either direction).
Also here: the cross-attention core (q from `tgt`, [k | v] from `memory`: decoder / cross layers), fp32 operands on
the fp32 MFMA (parity runs at fp32 tolerances, up to 256 tokens) and the fp8-product forward (set_fp8_products).
GPU only; there is no CPU path (the torch formulation in transformers.attention_core is used for CPU tensors).
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import _native

HEAD_DIM = 64
SPATIAL_VEC = 6
MAX_LEN = 512


MAX_LEN_F32 = 256


def supported(d_model: int, n_head: int, length: int, dtype: torch.dtype = torch.bfloat16, kv_length: Optional[int] = None) -> bool:
    """Does libgps_hip.so serve this attention call?  bf16: up to 512 tokens; fp32 operands (fp32 MFMA): up to 256."""
    cap = MAX_LEN_F32 if dtype == torch.float32 else MAX_LEN
    kv = length if kv_length is None else kv_length
    return d_model == n_head * HEAD_DIM and 0 < length <= cap and 0 < kv <= cap


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


MAX_LEN_PLANES = 144          # spatial self-attention rows served by gps_attention_sp.hip (plane form of the pairwise term)
_PLANES = True                # False: the general kernels with the interleaved fp32 pairwise tensor (A/B runs, tests)


def set_spatial_planes(flag: bool) -> None:
    global _PLANES
    _PLANES = bool(flag)


_FP8 = False        # attention-core products (Q K^T, P V) of the bf16 forward on the OCP e4m3 MFMA (bench.py --fp8)


def set_fp8_products(flag: bool) -> None:
    """BASELINE configs[4] ("fp8 MFMA attention path"): the forward products of every bf16 attention call run on
    v_mfma_f32_16x16x32_fp8_fp8 (per-row / per-tile scales, fp32 softmax); backward products stay bf16."""
    global _FP8
    _FP8 = bool(flag)


def fp8_products() -> bool:
    return _FP8


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _native.ATTN_BF16
    if t.dtype == torch.float32:
        return _native.ATTN_F32
    raise RuntimeError(f"fused attention serves bf16 and fp32 operands, not {t.dtype}")


def _mask8(mask: Optional[torch.Tensor]):
    # a bool tensor is one byte per element holding 0 / 1: the kernels read it in place (no conversion launch)
    if mask is None:
        return None
    return mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()


def _call(backward: bool, name: str, nbytes: int, flops: int, work_fraction=None, **f) -> None:
    """One gps_attn_forward_ex / gps_attn_backward_ex launch on the current stream (tensors -> pointers and pitches)."""
    import ctypes

    from ...pointnet2._ext import _timed
    a = _native.AttnArgs()
    for k, v in f.items():
        setattr(a, k, v.data_ptr() if torch.is_tensor(v) else v)
    lib = _native.load()
    fn = lib.gps_attn_backward_ex if backward else lib.gps_attn_forward_ex
    mfma = "fp32" if a.dtype == _native.ATTN_F32 else ("fp8" if (a.compute == _native.ATTN_COMPUTE_FP8 and not backward) else "bf16")
    with _timed(name, nbytes, flops, mfma, work_fraction):
        st = fn(ctypes.byref(a), _stream())
    _native.check(st, name)


class _FusedSelfAttention(torch.autograd.Function):
    """packed (B, L, 3 D [+ H 6]) = [q | k | v [| cond]] in bf16 or fp32 -> (B, L, D), same dtype."""

    @staticmethod
    def forward(ctx, packed: torch.Tensor, pl: Optional[torch.Tensor], mask: Optional[torch.Tensor],
                n_head: int, p_drop: float, seed: int, seed_dev: Optional[torch.Tensor]) -> torch.Tensor:
        B, L, W = packed.shape
        D = n_head * HEAD_DIM
        spatial = pl is not None
        assert W == 3 * D + (n_head * SPATIAL_VEC if spatial else 0), (W, D, spatial)
        assert packed.is_cuda and packed.is_contiguous()
        dt = _dtype_code(packed)
        m8 = _mask8(mask)
        out = torch.empty((B, L, D), dtype=packed.dtype, device=packed.device)
        lse = torch.empty((B, n_head, L), dtype=torch.float32, device=packed.device)
        base, esz = packed.data_ptr(), packed.element_size()
        fp8 = _FP8 and dt == _native.ATTN_BF16
        if spatial:
            assert pl.shape == (B, L, L, 5), pl.shape
        if spatial and _PLANES and dt == _native.ATTN_BF16 and p_drop == 0.0 and L <= MAX_LEN_PLANES and not fp8:
            # plane form: fp16 planes of the pairwise tensor, the conditioning vector read in place from `packed`
            from ..utils import pairwise_planes
            planes = pairwise_planes(pl)
            nbytes = esz * B * L * 4 * D + planes.numel() * 2 + B * L * n_head * 6 * 2
            with torch.cuda.device(packed.device):
                _call(False, f"attn_forward(L={L},spatial=1)", nbytes, 4 * B * n_head * L * L * HEAD_DIM,
                      B=B, H=n_head, Lq=L, Lk=L, head_dim=HEAD_DIM, dtype=dt, compute=_native.ATTN_COMPUTE_NATIVE,
                      q=base, ld_q=W, k=base + D * esz, v=base + 2 * D * esz, ld_kv=W, mask=_ptr(m8), p_drop=0.0, seed=0,
                      out=out, ld_o=D, lse=lse, pl_planes=planes, ld_pl=planes.shape[-1], sw16=base + 3 * D * esz, ld_sw=W)
            ctx.save_for_backward(packed, planes, m8, lse, out)
            ctx.meta = (n_head, 0.0, 0, "planes")
            return out
        sw = packed[..., 3 * D:].float().contiguous() if spatial else None
        if spatial:
            pl = pl.float().contiguous()
        # algorithmic work: q,k,v,out once (+ pairwise/cond vector), 2 x L x L x 64 MACs per head
        nbytes = esz * B * L * 4 * D + (B * L * L * 5 * 4 + B * L * n_head * 6 * 4 if spatial else 0)
        with torch.cuda.device(packed.device):
            _call(False, f"attn_forward(L={L},spatial={int(spatial)})" + ("[fp32]" if dt else "") + ("[fp8]" if fp8 else ""),
                  nbytes, 4 * B * n_head * L * L * HEAD_DIM,
                  B=B, H=n_head, Lq=L, Lk=L, head_dim=HEAD_DIM, dtype=dt,
                  compute=_native.ATTN_COMPUTE_FP8 if fp8 else _native.ATTN_COMPUTE_NATIVE,
                  q=base, ld_q=W, k=base + D * esz, v=base + 2 * D * esz, ld_kv=W, sw=_ptr(sw), pl=_ptr(pl), mask=_ptr(m8),
                  p_drop=float(p_drop), seed=int(seed), seed_dev=_ptr(seed_dev), out=out, ld_o=D, lse=lse)
        ctx.save_for_backward(packed, sw, pl, m8, lse, seed_dev, out)
        ctx.meta = (n_head, float(p_drop), int(seed), spatial)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        n_head, p_drop, seed, spatial = ctx.meta
        if spatial == "planes":
            packed, planes, m8, lse, out = ctx.saved_tensors
            B, L, W = packed.shape
            D = n_head * HEAD_DIM
            dout = dout.to(packed.dtype).contiguous()
            dpacked = torch.empty_like(packed)
            base, esz, gbase = packed.data_ptr(), packed.element_size(), dpacked.data_ptr()
            nbytes = esz * B * L * 9 * D + planes.numel() * 2 + 2 * B * L * n_head * 6 * 2
            with torch.cuda.device(packed.device):
                _call(True, f"attn_backward(L={L},spatial=1)", nbytes, 10 * B * n_head * L * L * HEAD_DIM,
                      B=B, H=n_head, Lq=L, Lk=L, head_dim=HEAD_DIM, dtype=_native.ATTN_BF16, compute=_native.ATTN_COMPUTE_NATIVE,
                      q=base, ld_q=W, k=base + D * esz, v=base + 2 * D * esz, ld_kv=W, mask=_ptr(m8), p_drop=0.0, seed=0,
                      out=out, ld_o=D, lse=lse, dout=dout, dq=gbase, ld_dq=W, dk=gbase + D * esz, dv=gbase + 2 * D * esz,
                      ld_dkv=W, pl_planes=planes, ld_pl=planes.shape[-1], sw16=base + 3 * D * esz, ld_sw=W,
                      dsw16=gbase + 3 * D * esz, ld_dsw=W)
            return dpacked, None, None, None, None, None, None
        packed, sw, pl, m8, lse, seed_dev, out = ctx.saved_tensors
        B, L, W = packed.shape
        D = n_head * HEAD_DIM
        dt = _dtype_code(packed)
        dout = dout.to(packed.dtype).contiguous()
        dpacked = torch.empty_like(packed)
        dsw = torch.empty_like(sw) if spatial else None
        base, esz = packed.data_ptr(), packed.element_size()
        gbase = dpacked.data_ptr()
        nbytes = esz * B * L * 8 * D + (B * L * L * 5 * 4 + 2 * B * L * n_head * 6 * 4 if spatial else 0)
        delta = _delta_ws(B, n_head, L, packed.device) if (not spatial and dt == _native.ATTN_BF16) else None
        with torch.cuda.device(packed.device):
            _call(True, f"attn_backward(L={L},spatial={int(spatial)})" + ("[fp32]" if dt else ""),
                  nbytes, 10 * B * n_head * L * L * HEAD_DIM,
                  B=B, H=n_head, Lq=L, Lk=L, head_dim=HEAD_DIM, dtype=dt, compute=_native.ATTN_COMPUTE_NATIVE,
                  q=base, ld_q=W, k=base + D * esz, v=base + 2 * D * esz, ld_kv=W, sw=_ptr(sw), pl=_ptr(pl), mask=_ptr(m8),
                  p_drop=p_drop, seed=seed, seed_dev=_ptr(seed_dev), out=out, ld_o=D, lse=lse, dout=dout,
                  dq=gbase, ld_dq=W, dk=gbase + D * esz, dv=gbase + 2 * D * esz, ld_dkv=W, dsw=_ptr(dsw), delta_ws=_ptr(delta))
        if spatial:
            dpacked[..., 3 * D:] = dsw
        return dpacked, None, None, None, None, None, None


class _FusedCrossAttention(torch.autograd.Function):
    """q (B, Lq, D) from `tgt`, kv (B, Lk, 2 D) = [k | v] from `memory`, bf16 or fp32 -> (B, Lq, D): the core of
    nn.MultiheadAttention(tgt, memory, memory, key_padding_mask=...) in the reference's CrossAttentionLayer /
    TransformerDecoderLayer / TransformerSpatialDecoderLayer (modules/layers/transformers.py:12-112, 242-282)."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, kv: torch.Tensor, mask: Optional[torch.Tensor], n_head: int, p_drop: float,
                seed_dev: Optional[torch.Tensor]) -> torch.Tensor:
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        assert D == n_head * HEAD_DIM and kv.shape == (B, Lk, 2 * D), (q.shape, kv.shape)
        assert q.is_cuda and q.is_contiguous() and kv.is_contiguous() and q.dtype == kv.dtype
        dt = _dtype_code(q)
        m8 = _mask8(mask)
        out = torch.empty_like(q)
        lse = torch.empty((B, n_head, Lq), dtype=torch.float32, device=q.device)
        esz = q.element_size()
        fp8 = _FP8 and dt == _native.ATTN_BF16
        nbytes = esz * B * (2 * Lq + 2 * Lk) * D
        with torch.cuda.device(q.device):
            _call(False, f"xattn_forward(Lq={Lq},Lk={Lk})" + ("[fp32]" if dt else "") + ("[fp8]" if fp8 else ""),
                  nbytes, 4 * B * n_head * Lq * Lk * HEAD_DIM,
                  B=B, H=n_head, Lq=Lq, Lk=Lk, head_dim=HEAD_DIM, dtype=dt,
                  compute=_native.ATTN_COMPUTE_FP8 if fp8 else _native.ATTN_COMPUTE_NATIVE,
                  q=q, ld_q=D, k=kv.data_ptr(), v=kv.data_ptr() + D * esz, ld_kv=2 * D, mask=_ptr(m8),
                  p_drop=float(p_drop), seed=0, seed_dev=_ptr(seed_dev), out=out, ld_o=D, lse=lse)
        ctx.save_for_backward(q, kv, m8, lse, seed_dev, out)
        ctx.meta = (n_head, float(p_drop))
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        q, kv, m8, lse, seed_dev, out = ctx.saved_tensors
        n_head, p_drop = ctx.meta
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        dt = _dtype_code(q)
        dout = dout.to(q.dtype).contiguous()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        esz = q.element_size()
        nbytes = esz * B * (4 * Lq + 4 * Lk) * D
        with torch.cuda.device(q.device):
            _call(True, f"xattn_backward(Lq={Lq},Lk={Lk})" + ("[fp32]" if dt else ""),
                  nbytes, 10 * B * n_head * Lq * Lk * HEAD_DIM,
                  B=B, H=n_head, Lq=Lq, Lk=Lk, head_dim=HEAD_DIM, dtype=dt, compute=_native.ATTN_COMPUTE_NATIVE,
                  q=q, ld_q=D, k=kv.data_ptr(), v=kv.data_ptr() + D * esz, ld_kv=2 * D, mask=_ptr(m8),
                  p_drop=p_drop, seed=0, seed_dev=_ptr(seed_dev), out=out, ld_o=D, lse=lse, dout=dout,
                  dq=dq, ld_dq=D, dk=dkv.data_ptr(), dv=dkv.data_ptr() + D * esz, ld_dkv=2 * D,
                  delta_ws=_ptr(_delta_ws(B, n_head, Lq, q.device) if dt == _native.ATTN_BF16 else None))
        return dq, dkv, None, None, None, None


class _FusedVarlenSelfAttention(torch.autograd.Function):
    """Plain self-attention over B VARIABLE-LENGTH sequences stored back to back: packed (T, 3 D) = [q | k | v] rows,
    cu_rows (B + 1) int32 row offsets on the device (sequence b = rows [cu[b], cu[b + 1])), cap = an upper bound of
    every length.  -> (T, D).  Rows outside every sequence (>= cu[B]) are neither read nor written.  No padding mask:
    there are no padded keys inside a sequence.  Work scales with sum L_b^2 (include/gps_hip.h gps_attn_args.cu_rows)."""

    @staticmethod
    def forward(ctx, packed: torch.Tensor, cu_rows: torch.Tensor, n_seq: int, cap: int, n_head: int, p_drop: float,
                seed_dev: Optional[torch.Tensor], order: Optional[torch.Tensor] = None,
                q_limit: Optional[torch.Tensor] = None) -> torch.Tensor:
        T, W = packed.shape
        D = n_head * HEAD_DIM
        assert W == 3 * D and packed.is_cuda and packed.dtype == torch.bfloat16 and packed.is_contiguous()
        assert cu_rows.dtype == torch.int32 and cu_rows.numel() == n_seq + 1 and cu_rows.is_cuda
        out = torch.empty((T, D), dtype=torch.bfloat16, device=packed.device)
        lse = torch.empty((n_seq, n_head, cap), dtype=torch.float32, device=packed.device)
        base, esz = packed.data_ptr(), packed.element_size()
        frac = _varlen_fraction(cu_rows, n_seq, cap)
        with torch.cuda.device(packed.device):
            _call(False, f"attn_forward(L<={cap},varlen,seqs={n_seq})", esz * n_seq * cap * 4 * D,
                  4 * n_seq * n_head * cap * cap * HEAD_DIM, work_fraction=frac,
                  B=n_seq, H=n_head, Lq=cap, Lk=cap, head_dim=HEAD_DIM, dtype=_native.ATTN_BF16,
                  compute=_native.ATTN_COMPUTE_NATIVE, q=base, ld_q=W, k=base + D * esz, v=base + 2 * D * esz, ld_kv=W,
                  p_drop=float(p_drop), seed=0, seed_dev=_ptr(seed_dev), out=out, ld_o=D, lse=lse, cu_rows=cu_rows,
                  seq_order=_ptr(order), q_limit=_ptr(q_limit))
        ctx.save_for_backward(packed, cu_rows, lse, seed_dev, out, order, q_limit)
        ctx.meta = (n_seq, cap, n_head, float(p_drop))
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        packed, cu_rows, lse, seed_dev, out, order, q_limit = ctx.saved_tensors
        n_seq, cap, n_head, p_drop = ctx.meta
        T, W = packed.shape
        D = n_head * HEAD_DIM
        dout = dout.to(torch.bfloat16).contiguous()
        dpacked = torch.empty_like(packed)
        base, esz, gbase = packed.data_ptr(), packed.element_size(), dpacked.data_ptr()
        frac = _varlen_fraction(cu_rows, n_seq, cap)
        with torch.cuda.device(packed.device):
            _call(True, f"attn_backward(L<={cap},varlen,seqs={n_seq})", esz * n_seq * cap * 8 * D,
                  10 * n_seq * n_head * cap * cap * HEAD_DIM, work_fraction=frac,
                  B=n_seq, H=n_head, Lq=cap, Lk=cap, head_dim=HEAD_DIM, dtype=_native.ATTN_BF16,
                  compute=_native.ATTN_COMPUTE_NATIVE, q=base, ld_q=W, k=base + D * esz, v=base + 2 * D * esz, ld_kv=W,
                  p_drop=p_drop, seed=0, seed_dev=_ptr(seed_dev), out=out, ld_o=D, lse=lse, dout=dout,
                  dq=gbase, ld_dq=W, dk=gbase + D * esz, dv=gbase + 2 * D * esz, ld_dkv=W, cu_rows=cu_rows,
                  seq_order=_ptr(order), q_limit=_ptr(q_limit), delta_ws=_delta_ws(n_seq, n_head, cap, packed.device))
        return dpacked, None, None, None, None, None, None, None, None


def _delta_ws(n_seq: int, n_head: int, cap: int, device) -> torch.Tensor:
    """(n_seq, H, cap) fp32 scratch of the block-streaming backward (gps_attn_args.delta_ws): the dQ launch writes
    rowsum(dout * out) per query, the dK / dV launch reads it."""
    return torch.empty((n_seq, n_head, cap), dtype=torch.float32, device=device)


PLAIN_DEFAULT = 1 | 4


def set_plain_mode(mode: int = PLAIN_DEFAULT) -> int:
    """Kernel families of the plain form (include/gps_hip.h gps_attn_set_plain_blocks): bit 1 = block-streaming forward,
    2 = block-streaming backward, 4 = K / V-resident kernels for fixed-length rows up to 144 tokens; 0 = the
    whole-sequence kernels for everything.  Returns the previous mode."""
    return int(_native.load().gps_attn_set_plain_blocks(int(mode)))


def set_plain_blocks(flag: bool) -> int:
    """True: block-streaming kernels for forward AND backward, nothing resident (tests of that family); False: the
    whole-sequence kernels.  `set_plain_mode()` restores the product default."""
    return set_plain_mode(3 if flag else 0)


def _varlen_fraction(cu_rows: torch.Tensor, n_seq: int, cap: int):
    """bench accounting: sum L_b^2 / (B cap^2), read back at profile_stop() from a snapshot of the offsets."""
    from ...pointnet2._ext import profiling
    if not profiling():
        return None
    snap = cu_rows.detach().clone()

    def frac():
        lens = (snap[1:] - snap[:-1]).double()
        return float((lens * lens).sum().item()) / float(n_seq * cap * cap)
    return frac


def fused_varlen_self_attention(packed: torch.Tensor, cu_rows: torch.Tensor, n_seq: int, cap: int, n_head: int,
                                dropout_p: float = 0.0, training: bool = False,
                                order: Optional[torch.Tensor] = None,
                                q_limit: Optional[torch.Tensor] = None) -> torch.Tensor:
    """order: optional (n_seq) int32 permutation, longest sequences first (launch balance only; results do not
    depend on it).  q_limit: optional (n_seq) int32: only the first q_limit[b] query rows of sequence b are computed (the
    other output rows are left unwritten and take no gradient)."""
    p = float(dropout_p) if training else 0.0
    seed_dev = _next_device_seed(packed.device) if p > 0.0 else None
    return _FusedVarlenSelfAttention.apply(packed.contiguous(), cu_rows, int(n_seq), int(cap), n_head, p, seed_dev, order,
                                           q_limit)


def fused_self_attention(packed: torch.Tensor, n_head: int, pairwise_locs: Optional[torch.Tensor] = None,
                         key_padding_mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0,
                         training: bool = False) -> torch.Tensor:
    """packed (B, L, 3*D [+ H*6]) bf16 / fp32 -> (B, L, D) attention output (heads merged), same dtype."""
    p = float(dropout_p) if training else 0.0
    seed, seed_dev = 0, None
    if p > 0.0:
        # the dropout stream lives ON THE DEVICE (_SeedStream): this call's forward+backward read one
        # word of a per-device block.  No host value is baked into the launch, so a captured HIP graph
        # draws a fresh mask on every replay.
        seed_dev = _next_device_seed(packed.device)
    return _FusedSelfAttention.apply(packed.contiguous(), pairwise_locs, key_padding_mask, n_head, p, seed,
                                     seed_dev)


def fused_cross_attention(q: torch.Tensor, kv: torch.Tensor, n_head: int, key_padding_mask: Optional[torch.Tensor] = None,
                          dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    """q (B, Lq, D), kv (B, Lk, 2 D) = [k | v] (bf16 / fp32, projections already applied) -> (B, Lq, D)."""
    p = float(dropout_p) if training else 0.0
    seed_dev = _next_device_seed(q.device) if p > 0.0 else None
    return _FusedCrossAttention.apply(q.contiguous(), kv.contiguous(), key_padding_mask, n_head, p, seed_dev)


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


def snapshot_seed_state(device):
    """(block values, position) of the device's dropout seed stream, or None before its first use."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _SEED_STATE.get(device)
    return None if st is None else (st.block.clone(), st.pos)


def restore_seed_state(device, state) -> None:
    """Put the stream back where `snapshot_seed_state` saw it (a probe step must leave no trace in the training
    streams).  A stream that did not exist at the snapshot is dropped, so that its first real use re-creates it."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if state is None:
        _SEED_STATE.pop(device, None)
        return
    st = _stream_for(device)
    st.block = state[0].clone()
    st.pos = state[1]


def _next_device_seed(device: torch.device) -> torch.Tensor:
    return _stream_for(device).next()
