"""Block-streaming plain attention (gps_attention_fa.hip: 64 queries / keys per workgroup, the other side streamed in
64-row blocks, online softmax; two backward launches) against the whole-sequence kernels of gps_attention.hip on the SAME
inputs -- same dropout stream (one hash per (query, key pair)), same lse -- so the two must agree to bf16 rounding with
dropout ON, for every call form of the step: fixed-length self-attention with a key-padding mask (unified encoder),
packed variable-length sequences with dispatch order and query limit (BERT), cross-attention (decoder layers).
The fp32 formulation itself is the oracle of tests/test_gpu_attention.py / test_gpu_attention_ex.py, which run on the
block-streaming kernels by default.  The K / V-resident plain kernels of gps_attention_sp.hip (fixed-length rows up to 144
tokens: the joint sequences) take the same comparison (mode 4)."""
import pytest
import torch

from sceneverse_amd.modules.layers import fused_attention as FA

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = 12
D = H * 64


def _close(a, b, tol, what):
    a, b = a.float(), b.float()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= tol * ref + 1e-6, (what, err, ref)


def _both(run, new_mode=3):
    """run() on the whole-sequence kernels (mode 0) and on `new_mode` (3 = block-streaming forward + backward,
    4 = K / V-resident)."""
    res = {}
    for name, mode in (("whole", 0), ("new", new_mode)):
        FA.set_plain_mode(mode)
        try:
            res[name] = run()
        finally:
            FA.set_plain_mode()
    return res["whole"], res["new"]


@pytest.mark.parametrize("B,L", [(8, 130), (3, 130), (2, 64), (2, 65), (1, 1), (2, 300), (2, 512), (5, 37), (64, 130), (2, 144), (3, 80), (2, 81)])
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("mode", [3, 4])
def test_self_attention_with_mask_and_dropout(B, L, p, mode):
    if mode == 4 and L > 144:
        pytest.skip("the K / V-resident kernels serve rows up to 144 tokens")
    g = torch.Generator().manual_seed(B * 100 + L)
    packed = torch.randn(B, L, 3 * D, generator=g).to(torch.bfloat16).to(DEV)
    n_real = torch.randint(1, L + 1, (B,), generator=g)
    n_real[0] = L
    mask = (torch.arange(L)[None, :] >= n_real[:, None]).to(DEV)
    go = torch.randn(B, L, D, generator=g).to(torch.bfloat16).to(DEV)
    seed_dev = torch.tensor([12345], dtype=torch.int64, device=DEV)

    def run():
        x = packed.clone().requires_grad_(True)
        out = FA._FusedSelfAttention.apply(x, None, mask, H, p, 77, seed_dev if p else None)
        out.backward(go)
        return out.detach(), x.grad.detach()
    (o_w, g_w), (o_b, g_b) = _both(run, mode)
    valid = ~mask
    assert torch.isfinite(o_b[valid]).all() and torch.isfinite(g_b).all()
    _close(o_b[valid], o_w[valid], 1e-2, "out")
    # scale = the whole gradient tensor's: with one live key dq and dk are exactly 0 in exact arithmetic and whatever the
    # family's delta (rowsum(dO * O) from the bf16 output, or rowsum(P dP) in fp32) leaves of the cancellation otherwise
    scale = g_w[valid].abs().max().item()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        err = (g_b[valid][..., sl].float() - g_w[valid][..., sl].float()).abs().max().item()
        assert err <= 1.5e-2 * scale + 1e-6, (name, err, scale)
    assert g_b[..., D:][mask].abs().max().item() == 0.0 if mask.any() else True      # padded keys: no dk / dv


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("use_limit", [False, True])
def test_packed_variable_length_sequences(p, use_limit):
    g = torch.Generator().manual_seed(3)
    lens = torch.cat([torch.randint(6, 51, (10,), generator=g), torch.randint(30, 301, (10,), generator=g),
                      torch.tensor([300, 1, 64, 65, 0])])
    n_seq = lens.numel()
    cu = torch.zeros(n_seq + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    T = int(cu[-1]) + 7                                       # rows behind the last sequence: never touched
    packed = torch.randn(T, 3 * D, generator=g).to(torch.bfloat16).to(DEV)
    go = torch.randn(T, D, generator=g).to(torch.bfloat16).to(DEV)
    order = torch.argsort(lens, descending=True).to(torch.int32).to(DEV)
    q_limit = None
    if use_limit:
        q_limit = lens.clone().to(torch.int32)
        q_limit[10:] = 1                                      # the "captions": read at their first row only
        q_limit = q_limit.to(DEV)
    cu_d = cu.to(DEV)
    seed_dev = torch.tensor([999], dtype=torch.int64, device=DEV)

    def run():
        x = packed.clone().requires_grad_(True)
        out = FA._FusedVarlenSelfAttention.apply(x, cu_d, n_seq, 300, H, p, seed_dev if p else None, order, q_limit)
        rows = torch.zeros(T, dtype=torch.bool, device=DEV)
        for s in range(n_seq):
            n = int(lens[s]) if q_limit is None else min(int(lens[s]), int(q_limit[s]))
            rows[int(cu[s]):int(cu[s]) + n] = True
        out = torch.where(rows[:, None], out, torch.zeros_like(out))      # rows that are not computed hold garbage
        (out.float() * go.float()).sum().backward()
        return out.detach(), x.grad.detach()[:int(cu[-1])]
    (o_w, g_w), (o_b, g_b) = _both(run)
    assert torch.isfinite(o_b).all() and torch.isfinite(g_b).all()
    _close(o_b, o_w, 1e-2, "out")
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        _close(g_b[..., sl], g_w[..., sl], 1.5e-2, name)


@pytest.mark.parametrize("Lq,Lk", [(80, 50), (50, 80), (130, 300), (1, 70), (64, 64)])
def test_cross_attention(Lq, Lk):
    B = 3
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = torch.randn(B, Lq, D, generator=g).to(torch.bfloat16).to(DEV)
    kv = torch.randn(B, Lk, 2 * D, generator=g).to(torch.bfloat16).to(DEV)
    n_real = torch.randint(1, Lk + 1, (B,), generator=g)
    mask = (torch.arange(Lk)[None, :] >= n_real[:, None]).to(DEV)
    go = torch.randn(B, Lq, D, generator=g).to(torch.bfloat16).to(DEV)
    seed_dev = torch.tensor([5], dtype=torch.int64, device=DEV)

    def run():
        qr, kr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        out = FA._FusedCrossAttention.apply(qr, kr, mask, H, 0.1, seed_dev)
        out.backward(go)
        return out.detach(), qr.grad.detach(), kr.grad.detach()
    (o_w, dq_w, dkv_w), (o_b, dq_b, dkv_b) = _both(run)
    _close(o_b, o_w, 1e-2, "out")
    _close(dq_b, dq_w, 1.5e-2, "dq")
    _close(dkv_b[~mask], dkv_w[~mask], 1.5e-2, "dkv")
    assert dkv_b[mask].abs().max().item() == 0.0 if mask.any() else True
