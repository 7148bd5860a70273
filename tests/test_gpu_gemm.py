"""bf16 MFMA GEMMs of libgps_hip.so (gps_gemm_bf16) against fp32 torch.mm on the SAME bf16-rounded operands:
forward (NT), input gradient (NN), weight gradient (TN, split-K, bias gradient), every tile variant, ragged
edges and K tails, the fused epilogues, and the autograd functions the layers call (modules/layers/gemm.py).
Tolerance: the product of bf16 operands is exact in fp32, so the only errors are fp32 accumulation order
(~1e-6 relative) and the final bf16 rounding of the output (2^-9 relative): the bar is 2^-7 * max|ref| per
element for bf16 outputs and 1e-4 relative for fp32 outputs."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneverse_amd import _native  # noqa: E402
from sceneverse_amd.modules.layers import gemm as G  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL16 = 2.0 ** -7


def _rand16(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).to(torch.bfloat16).to(DEV)


def _close16(got, ref, what):
    ref = ref.float()
    err = (got.float() - ref).abs().max().item()
    bound = TOL16 * ref.abs().max().item() + 1e-6
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"


# (M, N, K): transformer shapes of the GPS step + ragged / tiny / tail cases
NT_SHAPES = [(5120, 768, 768), (8320, 2304, 768), (5120, 2376, 768), (5120, 2048, 768), (8320, 768, 2048),
             (3200, 3072, 768), (300, 72, 40), (129, 8, 8), (1, 768, 768), (257, 132, 200), (640, 640, 2376),
             (19200, 2304, 768), (8320, 1024, 64), (8200, 1032, 72)]      # persistent variants: 2 - 6 tiles per workgroup, 1- and 2-stage K


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_forward_nt(M, N, K, variant):
    x, w = _rand16(M, K, seed=1), _rand16(N, K, scale=0.05, seed=2)
    b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    G.gemm(_native.GEMM_NT, _native.EPI_BIAS, M, N, K, x, K, w, K, y, N, bias=b, variant=variant)
    _close16(y, x.float() @ w.float().t() + b, f"NT {M}x{N}x{K} v{variant}")


def test_forward_nt_strided_operands_and_output():
    """Row pitches larger than the logical width (views into packed buffers)."""
    M, N, K = 520, 264, 136
    xb, wb = _rand16(M, K + 24, seed=3), _rand16(N, K + 8, scale=0.1, seed=4)
    x, w = xb[:, :K], wb[:, :K]
    yb = torch.full((M, N + 12), 7.0, dtype=torch.bfloat16, device=DEV)
    G.gemm(_native.GEMM_NT, _native.EPI_BIAS, M, N, K, x, xb.stride(0), w, wb.stride(0), yb, yb.stride(0))
    _close16(yb[:, :N], x.float() @ w.float().t(), "NT strided")
    assert torch.all(yb[:, N:] == 7.0)                       # nothing written past N


NN_SHAPES = [(5120, 768, 768), (8320, 768, 2304), (5120, 768, 2376), (8320, 2048, 768), (5120, 768, 2048),
             (300, 40, 72), (129, 8, 8), (257, 200, 136), (640, 2376, 640)]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("M,N,K", NN_SHAPES)
def test_dgrad_nn(M, N, K, variant):
    dy, w = _rand16(M, K, seed=5), _rand16(K, N, scale=0.05, seed=6)      # w is (out = K, in = N)
    dx = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    G.gemm(_native.GEMM_NN, _native.EPI_BIAS, M, N, K, dy, K, w, N, dx, N, variant=variant)
    _close16(dx, dy.float() @ w.float(), f"NN {M}x{N}x{K} v{variant}")


# (tokens, out, in)
TN_SHAPES = [(5120, 768, 768), (8320, 2304, 768), (5120, 2376, 768), (8320, 2048, 768), (8320, 768, 2048),
             (19200, 768, 768), (3200, 3072, 768), (300, 72, 40), (129, 8, 8), (1000, 136, 264), (77, 768, 768)]


@pytest.mark.parametrize("splits", [1, -1, 7])
@pytest.mark.parametrize("T,N,K", TN_SHAPES)
def test_wgrad_tn(T, N, K, splits):
    dy, x = _rand16(T, N, seed=7), _rand16(T, K, seed=8)
    lib = _native.load()
    s = int(lib.gps_gemm_pick_splits(_native.GEMM_TN, N, K, T)) if splits < 0 else splits
    dw = torch.empty(N, K, device=DEV)
    db = torch.empty(N, device=DEV)
    ws = torch.empty(max(1, int(lib.gps_gemm_workspace_floats(_native.GEMM_TN, N, K, s))), device=DEV)
    for variant in (0, 1, 2, 5, 7, 12):
        dw.fill_(float("nan"))
        db.fill_(float("nan"))
        G.gemm(_native.GEMM_TN, _native.EPI_F32, N, K, T, dy, N, x, K, dw, K, workspace=ws, colsum=db, splits=s,
               variant=variant)
        ref = dy.float().t() @ x.float()
        scale = ref.abs().max().item()
        assert (dw - ref).abs().max().item() <= 1e-4 * scale + 1e-5, f"TN {T}x{N}x{K} s{s} v{variant}"
        refb = dy.float().sum(0)
        assert (db - refb).abs().max().item() <= 1e-4 * refb.abs().max().item() + 1e-4


def test_wgrad_is_deterministic():
    dy, x = _rand16(8320, 768, seed=9), _rand16(8320, 2048, seed=10)
    a, _ = G.linear_wgrad(dy, x)
    b, _ = G.linear_wgrad(dy, x)
    assert torch.equal(a, b)


def _gelu_ref(pre16):
    return F.gelu(pre16.float())


@pytest.fixture(params=[-1, 8, 9, 10, 11, 12])
def forced_variant(request):
    """-1 = the shape-based default; 8 / 9 / 10 = the persistent tile walks (several tiles per workgroup at T = 9000);
    11 = the four-wave 256 x 256 tile with register-staged operands, 12 = the eight-wave two-group 256 x 256 tile
    (both: whole 64-wide K stages only, hence their own shapes)"""
    for form in (_native.GEMM_NT, _native.GEMM_NN):
        G.set_gemm_variant(form, request.param)
    yield request.param
    for form in (_native.GEMM_NT, _native.GEMM_NN):
        G.set_gemm_variant(form, -1)


@pytest.mark.parametrize("act", ["gelu", "relu"])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_activation_epilogues_forward_and_backward(act, p, forced_variant):
    T, Kin, Hid = (1300, 136, 264) if forced_variant < 0 else ((9000, 128, 1024) if forced_variant in (11, 12) else (9000, 136, 1032))
    x, w1 = _rand16(T, Kin, seed=11), _rand16(Hid, Kin, scale=0.2, seed=12)
    b1 = torch.randn(Hid, device=DEV)
    seed_dev = torch.tensor([123456789], dtype=torch.int64, device=DEV)
    h, pre = G.linear_forward(x, w1, b1, act=act, p_drop=p, seed_dev=seed_dev, want_pre=True)
    pre_ref = (x.float() @ w1.float().t() + b1)
    if act == "gelu":
        _close16(pre, pre_ref, "pre-activation")
        full = _gelu_ref(pre)                       # the kernel applies GELU to the bf16-rounded pre-activation
    else:
        full = torch.relu(pre_ref)
    keep = torch.ones_like(full, dtype=torch.bool)
    if p > 0:
        keep = (h != 0) | (full.abs() < 1e-3)       # the mask is whatever the kernel drew
        frac = 1.0 - ((h == 0) & (full.abs() > 1e-3)).float().mean().item() / max((full.abs() > 1e-3).float().mean().item(), 1e-9)
        assert abs(frac - (1 - p)) < 0.02, frac
    ref_h = torch.where(keep, full / (1 - p), torch.zeros_like(full))
    _close16(h, ref_h, f"{act} hidden p={p}")
    # backward epilogue: dpre = (dy W2) * act'(pre) * mask / (1 - p), the mask recomputed from (seed, index)
    Out = 128 if forced_variant in (11, 12) else 72      # variants 11 / 12 need whole 64-wide K stages
    dy, w2 = _rand16(T, Out, seed=13), _rand16(Out, Hid, scale=0.2, seed=14)
    dpre = G.linear_dgrad(dy, w2, act=act, aux=pre if act == "gelu" else h, p_drop=p, seed_dev=seed_dev)
    dh = dy.float() @ w2.float()
    if act == "gelu":
        z = pre.float().requires_grad_(True)
        F.gelu(z).sum().backward()
        dact = z.grad
    else:
        dact = (h != 0).float()
    mask = (h != 0) if p > 0 else torch.ones_like(keep)
    ref = dh * dact * (mask.float() / (1 - p))
    if p > 0 and act == "gelu":                     # elements with gelu(pre) == 0 exactly cannot reveal their mask
        sel = full.abs() > 1e-3
        _close16(torch.where(sel, dpre.float(), torch.zeros_like(ref)), torch.where(sel, ref, torch.zeros_like(ref)),
                 "dgelu x mask")
    else:
        _close16(dpre, ref, f"d{act} p={p}")


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gelu_factor_epilogues_equal_the_recomputing_pair(p, forced_variant):
    """EPI_BIAS_GELU_FACTOR / EPI_MUL_AUX (the forward epilogue saves gelu'(pre) x dropout-mask / (1 - p), the backward
    multiplies by it) against EPI_BIAS_GELU / EPI_DGELU (pre-activation saved, derivative and mask recomputed): the same
    hidden activations bit for bit, the same input gradient to the bf16 rounding of the saved factor."""
    T, Kin, Hid = (1300, 136, 264) if forced_variant < 0 else ((9000, 128, 1024) if forced_variant in (11, 12) else (9000, 136, 1032))
    x, w1 = _rand16(T, Kin, seed=11), _rand16(Hid, Kin, scale=0.2, seed=12)
    b1 = torch.randn(Hid, device=DEV)
    seed_dev = torch.tensor([123456789], dtype=torch.int64, device=DEV)
    h_a, pre = G.linear_forward(x, w1, b1, act="gelu", p_drop=p, seed_dev=seed_dev, want_pre=True)
    h_b, fac = G.linear_forward(x, w1, b1, act="gelu", p_drop=p, seed_dev=seed_dev, want_pre="factor")
    assert torch.equal(h_a, h_b)
    z = pre.float().requires_grad_(True)
    F.gelu(z).sum().backward()
    mask = (h_a != 0) | (F.gelu(pre.float()).abs() < 1e-3) if p > 0 else torch.ones_like(h_a, dtype=torch.bool)
    sel = F.gelu(pre.float()).abs() >= 1e-3 if p > 0 else mask
    want = z.grad * mask.float() / (1 - p)
    _close16(torch.where(sel, fac.float(), torch.zeros_like(want)), torch.where(sel, want, torch.zeros_like(want)), "saved factor")
    Out = 128 if forced_variant in (11, 12) else 72
    dy, w2 = _rand16(T, Out, seed=13), _rand16(Out, Hid, scale=0.2, seed=14)
    d_a = G.linear_dgrad(dy, w2, act="gelu", aux=pre, p_drop=p, seed_dev=seed_dev)
    d_b = G.linear_dgrad(dy, w2, act="factor", aux=fac)
    _close16(d_b, d_a, f"input gradient p={p}")


def _rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def test_linear_and_packed_linear_autograd():
    torch.manual_seed(0)
    lins = [torch.nn.Linear(768, n).to(DEV) for n in (768, 768, 768, 72)]
    x = torch.randn(64, 80, 768, device=DEV, requires_grad=True)
    g = torch.randn(64, 80, 768 * 3 + 72, device=DEV)
    y = G.packed_linear(x, lins)
    (y.float() * g).sum().backward()
    got = [x.grad.clone()] + [m.weight.grad.clone() for m in lins] + [m.bias.grad.clone() for m in lins]
    x.grad = None
    for m in lins:
        m.zero_grad(set_to_none=True)
    x16 = x.detach().to(torch.bfloat16).float().requires_grad_(True)
    W = torch.cat([m.weight for m in lins], 0).to(torch.bfloat16).float()
    b = torch.cat([m.bias for m in lins], 0)
    Wl = W.detach().requires_grad_(True)
    bl = b.detach().requires_grad_(True)
    yr = x16 @ Wl.t() + bl
    _close16(y, yr.detach(), "packed forward")
    (yr * g.to(torch.bfloat16).float()).sum().backward()
    assert _rel_l2(got[0], x16.grad) < 1e-2
    r = 0
    for i, m in enumerate(lins):
        n = m.weight.shape[0]
        assert _rel_l2(got[1 + i], Wl.grad[r:r + n]) < 1e-2
        assert _rel_l2(got[5 + i], bl.grad[r:r + n]) < 1e-2
        r += n


@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_ffn_autograd_matches_torch(act):
    torch.manual_seed(1)
    l1, l2 = torch.nn.Linear(768, 2048).to(DEV), torch.nn.Linear(2048, 768).to(DEV)
    x = torch.randn(40, 130, 768, device=DEV, requires_grad=True)
    g = torch.randn(40, 130, 768, device=DEV)
    y = G.ffn(x, l1, l2, act, 0.1, training=False)
    (y.float() * g).sum().backward()
    got = [x.grad.clone(), l1.weight.grad.clone(), l1.bias.grad.clone(), l2.weight.grad.clone(), l2.bias.grad.clone()]
    x.grad = None
    l1.zero_grad(set_to_none=True)
    l2.zero_grad(set_to_none=True)
    fn = F.gelu if act == "gelu" else F.relu
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yr = l2(fn(l1(x)))
    (yr.float() * g).sum().backward()
    ref = [x.grad, l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad]
    _close16(y, yr.detach(), "ffn forward")
    for a, b, name in zip(got, ref, ["dx", "dW1", "db1", "dW2", "db2"]):
        assert _rel_l2(a, b) < 2e-2, (name, _rel_l2(a, b))


def test_shadow_refresh_follows_the_master():
    lin = torch.nn.Linear(64, 64).to(DEV)
    x = torch.randn(8, 64, device=DEV)
    y0 = G.linear(x, lin.weight, lin.bias).float()
    with torch.no_grad():
        lin.weight.mul_(2.0)
    y1 = G.linear(x, lin.weight, lin.bias).float()
    ref = x.to(torch.bfloat16).float() @ lin.weight.to(torch.bfloat16).float().t() + lin.bias
    _close16(y1, ref, "after in-place update")
    assert not torch.allclose(y0, y1)


def test_unsupported_shapes_are_refused():
    x, w = _rand16(16, 12), _rand16(8, 12)
    y = torch.empty(16, 8, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_native.GpsNativeError):
        G.gemm(_native.GEMM_NT, _native.EPI_BIAS, 16, 8, 12, x, 12, w, 12, y, 8)      # K not a multiple of 8


@pytest.mark.parametrize("K,variant", [(136, -1), (192, 12), (136, 6)])
def test_relu_split_epilogue_carries_fp32_values_as_bf16_pairs(K, variant):
    """GPS_GEMM_EPI_RELU_SPLIT: C = [hi | lo | hi] with hi + lo == relu(x W^T + b) to ~2^-16 relative (default tile, the
    two-group 256 x 256 kernel, the 128 x 64 tile)."""
    M, N = 1040, 264
    x, w = _rand16(M, K, seed=21), _rand16(N, K, scale=0.1, seed=22)
    b = torch.randn(N, device=DEV)
    c = torch.full((M, 3 * N + 8), 5.0, dtype=torch.bfloat16, device=DEV)
    G.gemm(_native.GEMM_NT, _native.EPI_RELU_SPLIT, M, N, K, x, K, w, K, c, c.stride(0), bias=b, variant=variant)
    ref = torch.relu(x.float() @ w.float().t() + b)
    hi, lo, hi2 = c[:, :N].float(), c[:, N:2 * N].float(), c[:, 2 * N:3 * N].float()
    assert torch.equal(hi, hi2) and torch.all(c[:, 3 * N:] == 5.0)
    assert torch.equal(hi, ref.to(torch.bfloat16).float()) or (hi - ref).abs().max() <= 2 ** -8 * ref.abs().max()
    err = (hi + lo - ref).abs().max().item()
    assert err <= 3e-5 * ref.abs().max().item() + 1e-6, err        # fp32 accumulation order + 2^-16 of the pair


@pytest.mark.parametrize("M,K", [(4112, 259), (16, 8), (81920, 259)])
def test_split3_mlp_with_max_over_16_rows_is_fp32_accurate(M, K):
    """Three-layer shared MLP + max over 16-row groups (PointNet++ group-all level) on the bf16 MFMA path with
    split operands, against the same chain in fp64."""
    g = torch.Generator().manual_seed(M + K)
    chans = [256, 512, 768] if M > 16 else [8, 16, 8]
    x = (torch.randn(M, K, generator=g) * 0.5).to(DEV)
    ws, ss, cin = [], [], K
    for co in chans:
        ws.append((torch.randn(co, cin, generator=g) * (2.0 / cin) ** 0.5).to(DEV))
        ss.append((torch.randn(co, generator=g) * 0.1).to(DEV))
        cin = co
    k_pads = [(K + 7) // 8 * 8] + chans[:-1]
    layers = [(G.split3_weight(w, kp), s) for w, s, kp in zip(ws, ss, k_pads)]
    out = G.split3_mlp_max16(x, layers)
    ref = x.double()
    for w, s in zip(ws, ss):
        ref = torch.relu(ref @ w.double().t() + s.double())
    ref = ref.view(M // 16, 16, -1).amax(1)
    assert out.shape == ref.shape and out.dtype == torch.float32
    err = (out.double() - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), (err, ref.abs().max().item())


def test_split3_points_builds_the_first_operand():
    """gps_split3_points == split3_rows(cat(xyz, feats^T)) bit for bit (same rne hi / lo, zero padding)."""
    B, n, C = 37, 16, 256
    g = torch.Generator().manual_seed(5)
    xyz = torch.randn(B, n, 3, generator=g).to(DEV)
    feats = torch.randn(B, C, n, generator=g).to(DEV)
    k_pad = (3 + C + 7) // 8 * 8
    got = G.split3_points(xyz, feats, k_pad)
    ref = G.split3_rows(torch.cat([xyz, feats.transpose(1, 2)], dim=2).reshape(B * n, 3 + C), k_pad)
    assert got.shape == ref.shape and torch.equal(got, ref)


def test_deferred_weight_gradients_equal_the_autograd_ones():
    """modules/layers/gemm.deferred_wgrads: dW / db computed on a side stream straight into param.grad must equal the
    gradients autograd accumulates on the main stream -- single Linears, a packed group, an FFN, a weight used twice
    in one backward (accumulation), and pre-existing .grad tensors (flat-buffer views of the split-graph step)."""
    torch.manual_seed(3)
    lin = torch.nn.Linear(768, 768).to(DEV)
    pack = [torch.nn.Linear(768, n).to(DEV) for n in (768, 768, 72)]
    l1, l2 = torch.nn.Linear(768, 2048).to(DEV), torch.nn.Linear(2048, 768).to(DEV)
    mods = [lin, l1, l2] + pack
    x = torch.randn(16, 80, 768, device=DEV)

    def loss_fn():
        a = G.linear(x.to(torch.bfloat16), lin.weight, lin.bias)
        a = G.linear(a, lin.weight, lin.bias)                      # second use of the same weight
        b = G.packed_linear(a, pack)
        c = G.ffn(a, l1, l2, "gelu", 0.0, False)
        return b.float().square().mean() + c.float().square().mean()

    def grads(deferred, preset):
        for m in mods:
            m.zero_grad(set_to_none=True)
            if preset:
                for p in m.parameters():
                    p.grad = torch.full_like(p, 0.25)
        if deferred:
            with G.deferred_wgrads():
                loss_fn().backward()
        else:
            loss_fn().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for m in mods for p in m.parameters()]

    for preset in (False, True):
        want, got = grads(False, preset), grads(True, preset)
        for a, b in zip(got, want):
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-7, (preset, (a - b).abs().max().item())


@pytest.mark.parametrize("variant", [7, 12])
@pytest.mark.parametrize("form", ["nt", "nn"])
def test_device_side_row_extent_forward_and_input_gradient(form, variant):
    """NT / NN with extent_dev: rows below the device-side count equal the full product, rows of LIVE tiles past it may
    hold anything, tiles that lie entirely past it are never written (a NaN prefill survives there), and operand rows
    past it may be NaN.  Variant 12 = the two-group 256 x 256 kernel (256-row tiles), 7 = the 128-row default."""
    M, N, K = 5000, 768, 1536
    tile = 256 if variant == 12 else 128
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    for ext in (0, 1, 255, 256, 3000, 4999, 5000):
        rows = torch.tensor([ext], dtype=torch.int32, device=DEV)
        xin = x.clone()
        xin[ext:] = float("nan")
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        if form == "nt":
            w = _rand16(N, K, scale=0.05, seed=42)
            G.gemm(_native.GEMM_NT, _native.EPI_BIAS, M, N, K, xin, K, w, K, y, N, variant=variant, extent_dev=rows)
            ref = x.float() @ w.float().t()
        else:
            w = _rand16(K, N, scale=0.05, seed=43)
            G.gemm(_native.GEMM_NN, _native.EPI_BIAS, M, N, K, xin, K, w, N, y, N, variant=variant, extent_dev=rows)
            ref = x.float() @ w.float()
        if ext:
            _close16(y[:ext], ref[:ext], f"{form} v{variant} extent {ext}")
        first_dead = -(-ext // tile) * tile
        assert torch.isnan(y[first_dead:].float()).all(), (form, variant, ext)


@pytest.mark.parametrize("form", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K", [(12608, 768, 3072), (5120, 2376, 768), (8320, 768, 776), (5000, 520, 1536)])
def test_stream_k_variant(form, M, N, K):
    """Variant 13 (gemm8p_sk_kernel: equal K-tile shares per CU, partial tiles through fp32 slabs + arrival flags) against
    fp32 torch.mm and against the default variant; the arrival words are back at zero after every launch, so the same
    workspace serves consecutive launches; ragged K tail (776) and a device-side row extent included.  Not selected by
    pick_variant: this test keeps the explicit variant honest."""
    lib = _native.load()
    ws = torch.zeros(int(lib.gps_gemm_sk_workspace_bytes()) // 4, device=DEV)
    x = _rand16(M, K, seed=51)
    b = torch.randn(N, device=DEV)
    if form == "nt":
        w = _rand16(N, K, scale=0.05, seed=52)
        ref = x.float() @ w.float().t() + b
        call = lambda y, v, **kw: G.gemm(_native.GEMM_NT, _native.EPI_BIAS, M, N, K, x, K, w, K, y, N, bias=b, variant=v, **kw)  # noqa: E731
    else:
        w = _rand16(K, N, scale=0.05, seed=53)
        ref = x.float() @ w.float() + b
        call = lambda y, v, **kw: G.gemm(_native.GEMM_NN, _native.EPI_BIAS, M, N, K, x, K, w, N, y, N, bias=b, variant=v, **kw)  # noqa: E731
    y7 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    call(y7, 7)
    for rep in range(2):
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        call(y, 13, workspace=ws)
        _close16(y, ref, f"stream-K {form} {M}x{N}x{K} launch {rep}")
        # different summation order of the K shares: one bf16 rounding step apart from the default variant at most
        assert (y.float() - y7.float()).abs().max().item() <= TOL16 * ref.abs().max().item()
    flags = ws[:1024].view(torch.int32)
    assert int(flags.abs().sum()) == 0, "arrival / error words not reset"
    ext = M - 300
    rows = torch.tensor([ext], dtype=torch.int32, device=DEV)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(y, 13, workspace=ws, extent_dev=rows)
    _close16(y[:ext], ref[:ext], f"stream-K {form} extent")
    assert torch.isnan(y[-(-ext // 256) * 256:].float()).all()


@pytest.mark.parametrize("ext", [0, 1, 480, 3000, 3200])
def test_split_k_input_gradient_stops_at_the_row_extent(ext):
    """NN form, fp32 output, reduction split over K (the masked-LM decoder's input gradient: 3 200 token rows of which the
    labelled ones come first, K = the vocabulary): rows below the device-side extent equal the full product, rows at or
    past it are never written -- neither by the partial-tile launch nor by the split reduction (NaN prefill survives)."""
    lib = _native.load()
    M, N, K, splits = 3200, 768, 4096, 8
    dy, w = _rand16(M, K, seed=61), _rand16(K, N, scale=0.05, seed=62)
    ws = torch.empty(int(lib.gps_gemm_workspace_floats(_native.GEMM_NN, M, N, splits)), device=DEV)
    rows = torch.tensor([ext], dtype=torch.int32, device=DEV)
    dx = torch.full((M, N), float("nan"), device=DEV)
    G.gemm(_native.GEMM_NN, _native.EPI_F32, M, N, K, dy, K, w, N, dx, N, workspace=ws, splits=splits, extent_dev=rows)
    ref = dy.float() @ w.float()
    if ext:
        assert (dx[:ext] - ref[:ext]).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert torch.isnan(dx[ext:]).all()


def _grouped(probs):
    """probs: list of dicts(dy (T, M) bf16, x (T, N) bf16, C fp32 (M, N), colsum or None, accumulate, extent or None)."""
    import ctypes
    arr = (_native.WgradProblem * len(probs))()
    for q, p in zip(arr, probs):
        q.M, q.N, q.K, q.accumulate = p["dy"].shape[1], p["x"].shape[1], p["dy"].shape[0], int(p.get("accumulate", 0))
        q.A, q.lda = p["dy"].data_ptr(), p["dy"].stride(0)
        q.B, q.ldb = p["x"].data_ptr(), p["x"].stride(0)
        q.C, q.ldc = p["C"].data_ptr(), p["C"].stride(0)
        q.colsum = p["colsum"].data_ptr() if p.get("colsum") is not None else None
        q.extent_dev = p["extent"].data_ptr() if p.get("extent") is not None else None
    st = _native.load().gps_gemm_wgrad_grouped(arr, len(probs), torch.cuda.current_stream().cuda_stream)
    _native.check(st, "gemm_wgrad_grouped")


@pytest.fixture(params=[1, 0], ids=["xcd_queues", "one_queue"])
def wgrad_queues(request):
    """Both tile schedules of gps_gemm_wgrad_grouped (XCD-local queues = the default; one global queue)."""
    lib = _native.load()
    prev = lib.gps_gemm_wgrad_grouped_set_xcd_queues(request.param)
    yield request.param
    lib.gps_gemm_wgrad_grouped_set_xcd_queues(prev)


def test_grouped_weight_gradients_against_fp32(wgrad_queues):
    """gps_gemm_wgrad_grouped: many dW = dY^T X (+ column sums of dY) in one persistent launch, no split over K: ragged
    M / N / K, a 72-row problem, strided operands (column slices of a packed dY), a device-side row extent with NaN rows
    behind it, plain stores and read-modify-write into buffers that already hold a gradient, more tiles than CUs."""
    torch.manual_seed(5)
    shapes = [(5120, 768, 768), (3000, 72, 768), (1400, 2304, 768), (777, 264, 136), (64, 8, 8), (2500, 3072, 768),
              (8320, 768, 2048), (130, 520, 264), (4096, 768, 3072)]
    probs, refs = [], []
    for i, (T, M, N) in enumerate(shapes):
        dyb = _rand16(T, M + 16, scale=0.5, seed=10 + i)           # dY as a column slice of a wider (packed) buffer
        dy = dyb[:, 8:8 + M]
        x = _rand16(T, N, seed=40 + i)
        ext, live = None, T
        if i % 3 == 1:
            live = T - 37
            ext = torch.tensor([live], dtype=torch.int32, device=DEV)
            dyb[live:] = float("nan")                              # rows past the extent must never reach the sums
            x[live:] = float("nan")
        acc = i % 2
        C = torch.full((M, N), 0.5, device=DEV) if acc else torch.full((M, N), float("nan"), device=DEV)
        cs = torch.full((M,), -2.0, device=DEV) if acc else torch.full((M,), float("nan"), device=DEV)
        ref = dy[:live].float().t() @ x[:live].float() + (0.5 if acc else 0.0)
        ref_cs = dy[:live].float().sum(0) + (-2.0 if acc else 0.0)
        probs.append(dict(dy=dy, x=x, C=C, colsum=cs if i != 4 else None, accumulate=acc, extent=ext))
        refs.append((ref, ref_cs))
    _grouped(probs)
    torch.cuda.synchronize()
    for p, (ref, ref_cs), shp in zip(probs, refs, shapes):
        err = (p["C"] - ref).abs().max().item()
        assert err <= 1e-4 * ref.abs().max().item() + 1e-5, (shp, err)
        if p["colsum"] is not None:
            err = (p["colsum"] - ref_cs).abs().max().item()
            assert err <= 1e-4 * ref_cs.abs().max().item() + 1e-4, (shp, "colsum", err)
    # deterministic: a second run into fresh buffers gives the same bits
    again = [dict(p, C=torch.full_like(p["C"], 0.5 if p["accumulate"] else 0.0),
                  colsum=None if p["colsum"] is None else torch.full_like(p["colsum"], -2.0 if p["accumulate"] else 0.0))
             for p in probs]
    _grouped(again)
    torch.cuda.synchronize()
    for a, p in zip(again, probs):
        assert torch.equal(a["C"], p["C"])


def test_grouped_schedules_give_the_same_bits_on_a_step_sized_problem_set():
    """~60 problems / ~900 tiles in three reduction-length classes (one with a device-side extent), i.e. several rounds
    of every XCD-local queue plus tiles taken from other XCDs' queues at the end: every tile is computed by exactly one
    workgroup with the same code whoever takes it, so the XCD-local schedule, the single global queue and a second run
    agree bit for bit -- and with the fp32 formulation on sampled problems."""
    lib = _native.load()
    torch.manual_seed(11)
    live = torch.tensor([6100], dtype=torch.int32, device=DEV)
    probs = []

    def add(T, parts, N, ext=None, seed=0):
        dyb = _rand16(T, sum(parts), scale=0.5, seed=seed)
        x = _rand16(T, N, seed=seed + 1)
        r = 0
        for m in parts:
            probs.append(dict(dy=dyb[:, r:r + m], x=x, C=None, colsum=None, accumulate=0, extent=ext, T=T if ext is None else 6100))
            r += m
    for l in range(3):
        add(9600, [768, 768, 768], 768, live, 100 + 10 * l), add(9600, [768], 768, live, 101 + 10 * l)
        add(9600, [3072], 768, live, 102 + 10 * l), add(9600, [768], 3072, live, 103 + 10 * l)
        add(2560, [768, 768, 768, 72], 768, None, 104 + 10 * l), add(2560, [2048], 768, None, 105 + 10 * l), add(2560, [768], 2048, None, 106 + 10 * l)
        add(4160, [2304], 768, None, 107 + 10 * l), add(4160, [768], 768, None, 108 + 10 * l), add(4160, [768], 2048, None, 109 + 10 * l)
    results = []
    for mode in (1, 0, 1):
        prev = lib.gps_gemm_wgrad_grouped_set_xcd_queues(mode)
        try:
            run = [dict(p, C=torch.full((p["dy"].shape[1], p["x"].shape[1]), float("nan"), device=DEV),
                        colsum=torch.full((p["dy"].shape[1],), float("nan"), device=DEV)) for p in probs]
            _grouped(run)
            torch.cuda.synchronize()
        finally:
            lib.gps_gemm_wgrad_grouped_set_xcd_queues(prev)
        results.append(run)
    for a, b, c in zip(*results):
        assert torch.equal(a["C"], b["C"]) and torch.equal(a["C"], c["C"])
        assert torch.equal(a["colsum"], b["colsum"]) and torch.equal(a["colsum"], c["colsum"])
    for i in range(0, len(probs), 5):
        p, T = results[0][i], probs[i]["T"]
        ref = p["dy"][:T].float().t() @ p["x"][:T].float()
        err = (p["C"] - ref).abs().max().item()
        assert err <= 1e-4 * ref.abs().max().item() + 1e-5, (i, err)
        ref_cs = p["dy"][:T].float().sum(0)
        assert (p["colsum"] - ref_cs).abs().max().item() <= 1e-4 * ref_cs.abs().max().item() + 1e-4


def test_grouped_weight_gradients_equal_the_autograd_ones():
    """modules/layers/gemm.grouped_wgrads: the deferred, grouped form must give the gradients autograd accumulates --
    single Linears, a packed group (column slices of one dY), an FFN, a weight used twice in one backward (two launches),
    with and without pre-existing .grad tensors (the flat-buffer views of the split-graph step)."""
    torch.manual_seed(3)
    lin = torch.nn.Linear(768, 768).to(DEV)
    pack = [torch.nn.Linear(768, n).to(DEV) for n in (768, 768, 72)]
    l1, l2 = torch.nn.Linear(768, 2048).to(DEV), torch.nn.Linear(2048, 768).to(DEV)
    mods = [lin, l1, l2] + pack
    x = torch.randn(16, 80, 768, device=DEV)

    def loss_fn():
        a = G.linear(x.to(torch.bfloat16), lin.weight, lin.bias)
        a = G.linear(a, lin.weight, lin.bias)                      # second use of the same weight
        b = G.packed_linear(a, pack)
        c = G.ffn(a, l1, l2, "gelu", 0.0, False)
        return b.float().square().mean() + c.float().square().mean()

    def grads(grouped, preset):
        for m in mods:
            m.zero_grad(set_to_none=True)
            if preset:
                for p in m.parameters():
                    p.grad = torch.full_like(p, 0.25)
        with G.grouped_wgrads(grouped):
            loss_fn().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for m in mods for p in m.parameters()]

    for preset in (False, True):
        want, got = grads(False, preset), grads(True, preset)
        for a, b in zip(got, want):
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-7, (preset, (a - b).abs().max().item())
