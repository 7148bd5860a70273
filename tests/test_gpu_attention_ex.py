"""The general form of the attention core (include/gps_hip.h gps_attn_forward_ex / gps_attn_backward_ex):

  * fp32 operands on the fp32 MFMA -- the "fp32 master path": compared with the fp32 torch formulation of the
    reference's mathematics (modules/layers/transformers.py:193-239, :141; the formulation tests/test_oracle_vs_golden.py
    pins to the reference) at |diff| <= 2e-3 * max|ref| (measured ~1e-6: both sides are fp32, only the summation order
    differs), self-attention plain / spatial and cross-attention, forward and backward;
  * cross-attention (q from `tgt`, k / v from `memory`, Lq != Lk) in bf16 at the bf16 tolerances of
    tests/test_gpu_attention.py, with padded memory keys, dropout adjoint identities;
  * module level: the decoder / cross layers of the reference (a16) run with the torch fallback DISABLED
    (`set_attention_backend("hip")` raises on any call the fused core does not serve) and agree with the torch
    formulation of the same modules, outputs and input gradients (the same layers against the outputs of the
    reference's own classes: tests/test_a16_vs_golden.py::test_a16_layers_gpu_fp32_on_the_fused_core_only);
  * the fp8-product forward (OCP e4m3 MFMA, BASELINE configs[4]) against the fp32 formulation: relative L2 <= 6e-2 per
    output tensor at 256 objects / 512 joint tokens (e4m3 carries 3 mantissa bits: 2^-4 per element, averaged over the
    keys of a row), and the same lse to 2e-2.
"""
import math

import pytest
import torch

from sceneverse_amd.modules.layers import fused_attention as FA
from sceneverse_amd.modules.layers import transformers as T

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = 12
D = H * 64


def ref_core(q, k, v, mask, sw=None, pl=None):
    """fp32 formulation: q (B,Lq,D), k / v (B,Lk,D), mask (B,Lk) True = padded, sw (B,Lq,H*6), pl (B,Lq,Lk,5)."""
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    qh = q.view(B, Lq, H, 64).transpose(1, 2)
    kh = k.view(B, Lk, H, 64).transpose(1, 2)
    vh = v.view(B, Lk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(64)
    if sw is not None:
        w = sw.view(B, Lq, H, 6).permute(0, 2, 1, 3)
        loc = torch.sigmoid(torch.einsum('bhld,bltd->bhlt', w[..., 1:], pl) + w[..., :1])
        if mask is not None:
            loc = loc.masked_fill(mask[:, None, None, :], 0)
        s = s + torch.log(torch.clamp(loc, min=1e-6))
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float('-inf'))
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, D), torch.logsumexp(s, dim=-1)


def _mask(B, L, g):
    n_real = torch.randint(max(1, L // 3), L + 1, (B,), generator=g)
    return torch.arange(L)[None, :] >= n_real[:, None]


def _close(a, b, tol, what):
    a, b = a.float().cpu(), b.float().cpu()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= tol * ref + 1e-7, (what, err, ref)


def _rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


# ---------------------------------------------------------------------------------------------------------
# fp32 operands, self-attention
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,L,spatial", [(3, 80, True), (8, 80, True), (2, 130, False), (2, 37, True), (2, 16, False),
                                         (1, 200, True), (2, 256, False), (1, 256, True), (8, 50, False)])
def test_fp32_self_attention_matches_the_fp32_formulation(B, L, spatial):
    g = torch.Generator().manual_seed(B * 1000 + L)
    W = 3 * D + (H * 6 if spatial else 0)
    packed = torch.randn(B, L, W, generator=g)
    pl = (torch.rand(B, L, L, 5, generator=g) * 2 - 1) if spatial else None
    mask = _mask(B, L, g)
    go = torch.randn(B, L, D, generator=g)
    ref_in = packed.clone().requires_grad_(True)
    ref, _ = ref_core(ref_in[..., :D], ref_in[..., D:2 * D], ref_in[..., 2 * D:3 * D], mask,
                      ref_in[..., 3 * D:] if spatial else None, pl)
    ref.backward(go)
    x = packed.to(DEV).requires_grad_(True)
    out = FA._FusedSelfAttention.apply(x, pl.to(DEV) if spatial else None, mask.to(DEV), H, 0.0, 0, None)
    assert out.dtype == torch.float32
    out.backward(go.to(DEV))
    _close(out, ref, 2e-3, "out")
    gx, gr = x.grad.cpu(), ref_in.grad
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        _close(gx[..., sl], gr[..., sl], 2e-3, name)
        assert _rel_l2(gx[..., sl], gr[..., sl]) <= 1e-4, (name, _rel_l2(gx[..., sl], gr[..., sl]))
    if spatial:
        _close(gx[..., 3 * D:], gr[..., 3 * D:], 2e-3, "dsw")
    assert gx[..., D:3 * D][mask].abs().max().item() == 0.0        # padded keys: no gradient through k, v


# ---------------------------------------------------------------------------------------------------------
# cross-attention, bf16 and fp32
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,B,Lq,Lk", [(torch.float32, 2, 80, 50), (torch.float32, 3, 37, 130), (torch.float32, 8, 16, 256),
                                          (torch.bfloat16, 2, 80, 50), (torch.bfloat16, 8, 80, 300), (torch.bfloat16, 3, 300, 37),
                                          (torch.bfloat16, 2, 512, 130), (torch.bfloat16, 1, 130, 512), (torch.bfloat16, 2, 1, 50)])
def test_cross_attention_matches_the_fp32_formulation(dtype, B, Lq, Lk):
    g = torch.Generator().manual_seed(B * 100000 + Lq * 1000 + Lk)
    q = torch.randn(B, Lq, D, generator=g).to(dtype)
    kv = torch.randn(B, Lk, 2 * D, generator=g).to(dtype)
    mask = _mask(B, Lk, g)
    go = torch.randn(B, Lq, D, generator=g).to(dtype)
    rq, rkv = q.detach().clone().float().requires_grad_(True), kv.detach().clone().float().requires_grad_(True)
    ref, _ = ref_core(rq, rkv[..., :D], rkv[..., D:], mask)
    ref.backward(go.float())
    xq, xkv = q.detach().clone().to(DEV).requires_grad_(True), kv.detach().clone().to(DEV).requires_grad_(True)
    out = FA._FusedCrossAttention.apply(xq, xkv, mask.to(DEV), H, 0.0, None)
    out.backward(go.to(DEV))
    f32 = dtype == torch.float32
    tol_o, tol_g, l2 = (2e-3, 2e-3, 1e-4) if f32 else (2e-2, 4e-2, 1e-2)
    _close(out, ref, tol_o, "out")
    assert _rel_l2(out, ref.detach()) <= l2
    _close(xq.grad, rq.grad, tol_g, "dq")
    _close(xkv.grad[..., :D], rkv.grad[..., :D], tol_g, "dk")
    _close(xkv.grad[..., D:], rkv.grad[..., D:], tol_g, "dv")
    for a, b in ((xq.grad, rq.grad), (xkv.grad[..., :D], rkv.grad[..., :D]), (xkv.grad[..., D:], rkv.grad[..., D:])):
        assert _rel_l2(a, b) <= l2
    assert xkv.grad[mask].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cross_attention_dropout_is_reproducible_and_adjoint(dtype):
    B, Lq, Lk = 2, 80, 130
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, Lq, D, generator=g).to(dtype).to(DEV)
    kv = torch.randn(B, Lk, 2 * D, generator=g).to(dtype).to(DEV)
    mask = _mask(B, Lk, g).to(DEV)
    seed = torch.tensor([123456789], dtype=torch.int64, device=DEV)
    o1 = FA._FusedCrossAttention.apply(q, kv, mask, H, 0.3, seed)
    o2 = FA._FusedCrossAttention.apply(q, kv, mask, H, 0.3, seed)
    assert torch.equal(o1, o2)
    o0 = FA._FusedCrossAttention.apply(q, kv, mask, H, 0.0, None)
    assert not torch.equal(o0, o1)
    # the output is linear in v for a fixed mask: <out(v), w> == <v, d out / d v [w]>
    kv2 = kv.clone().requires_grad_(True)
    out = FA._FusedCrossAttention.apply(q, kv2, mask, H, 0.3, seed)
    # a cotangent correlated with the output, so that neither inner product is a cancelling sum
    gen = torch.Generator(device=DEV).manual_seed(3)
    w = (out.detach().float() * (1.0 + 0.5 * torch.randn(out.shape, device=DEV, generator=gen))).to(dtype)
    out.backward(w)
    lhs = (out.float() * w.float()).sum().item()
    rhs = (kv2.grad[..., D:].float() * kv[..., D:].float()).sum().item()
    assert abs(lhs - rhs) <= (2e-2 if dtype == torch.bfloat16 else 1e-4) * max(abs(lhs), abs(rhs)), (lhs, rhs)


# ---------------------------------------------------------------------------------------------------------
# the layers of a16 with the torch fallback disabled
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_and_cross_layers_run_on_the_fused_core_only(dtype):
    """TransformerDecoderLayer / TransformerSpatialDecoderLayer / CrossAttentionLayer against the torch formulation
    of the same modules; backend 'hip' raises if any attention call of these layers is not served natively."""
    torch.manual_seed(0)
    B, Lq, Lk = 4, 80, 50
    layers = [T.TransformerDecoderLayer(D, H, dim_feedforward=256, dropout=0.1, activation="gelu"),
              T.TransformerSpatialDecoderLayer(D, H, dim_feedforward=256, dropout=0.1, activation="gelu",
                                               spatial_attn_fusion="cond"),
              T.CrossAttentionLayer(D, H, dim_feedforward=256, dropout=0.1, activation="gelu", prenorm=True),
              T.CrossAttentionLayer(D, H, dim_feedforward=256, dropout=0.1, activation="relu", k_dim=384, v_dim=384,
                                    prenorm=False)]
    g = torch.Generator().manual_seed(1)
    tgt = torch.randn(B, Lq, D, generator=g).to(DEV)
    pl = (torch.rand(B, Lq, Lq, 5, generator=g) * 2 - 1).to(DEV)
    tmask, mmask = _mask(B, Lq, g).to(DEV), _mask(B, Lk, g).to(DEV)
    for layer in layers:
        layer = layer.to(DEV).eval()
        wide = isinstance(layer, T.CrossAttentionLayer) and layer.multihead_attn.kdim != D
        memory = torch.randn(B, Lk, 384 if wide else D, generator=g).to(DEV)
        kw = dict(tgt_key_padding_mask=tmask, memory_key_padding_mask=mmask)
        if isinstance(layer, T.TransformerSpatialDecoderLayer):
            kw["tgt_pairwise_locs"] = pl

        def run(backend):
            T.set_attention_backend(backend)
            try:
                x = tgt.clone().requires_grad_(True)
                mem = memory.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == torch.bfloat16)):
                    y = layer(x, mem, **kw)[0]
                y.float().square().mean().backward()
                return y.detach().float(), x.grad.float(), mem.grad.float()
            finally:
                T.set_attention_backend("auto")
        want = run("torch")
        got = run("hip")
        tol = 2e-3 if dtype == torch.float32 else 3e-2
        for a, b, what in zip(got, want, ("out", "d tgt", "d memory")):
            _close(a, b, tol, (type(layer).__name__, what))


# ---------------------------------------------------------------------------------------------------------
# fp8 products (forward)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,L,spatial", [(2, 512, False), (2, 256, True), (8, 130, False), (3, 80, True), (2, 300, False)])
def test_fp8_forward_against_the_fp32_formulation(B, L, spatial):
    g = torch.Generator().manual_seed(B * 1000 + L + 7)
    W = 3 * D + (H * 6 if spatial else 0)
    packed = torch.randn(B, L, W, generator=g)
    if spatial:
        packed[..., 3 * D:] *= 2.0
    packed = packed.to(torch.bfloat16)
    pl = (torch.rand(B, L, L, 5, generator=g) * 2 - 1) if spatial else None
    mask = _mask(B, L, g)
    pf = packed.float()
    ref, ref_lse = ref_core(pf[..., :D], pf[..., D:2 * D], pf[..., 2 * D:3 * D], mask, pf[..., 3 * D:] if spatial else None, pl)
    FA.set_fp8_products(True)
    try:
        x = packed.to(DEV).requires_grad_(True)
        out = FA._FusedSelfAttention.apply(x, pl.to(DEV) if spatial else None, mask.to(DEV), H, 0.0, 0, None)
        # backward runs the bf16 products from the lse the fp8 forward saved: finite and close to the fp32 gradients
        go = torch.randn(B, L, D, generator=g).to(torch.bfloat16)
        out.backward(go.to(DEV))
    finally:
        FA.set_fp8_products(False)
    bf = FA._FusedSelfAttention.apply(packed.to(DEV), pl.to(DEV) if spatial else None, mask.to(DEV), H, 0.0, 0, None)
    e8, e16 = _rel_l2(out, ref), _rel_l2(bf, ref)
    assert e8 <= 6e-2, ("fp8 out rel-L2", e8, "bf16", e16)
    assert e8 > e16                                          # it really is the fp8 path that ran
    assert torch.isfinite(x.grad.float()).all()
    rin = pf.clone().requires_grad_(True)
    r2, _ = ref_core(rin[..., :D], rin[..., D:2 * D], rin[..., 2 * D:3 * D], mask, rin[..., 3 * D:] if spatial else None, pl)
    r2.backward(go.float())
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel_l2(x.grad[..., sl], rin.grad[..., sl]) <= 1e-1, (name, _rel_l2(x.grad[..., sl], rin.grad[..., sl]))
