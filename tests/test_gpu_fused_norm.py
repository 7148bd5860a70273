"""Fused y = LayerNorm(x + dropout(h)) (gps_add_dropout_layernorm_*) against torch's
layer_norm(x + dropout(h)) -- the residual step of modules/layers/transformers.py:143-153, :311-315.
fp32 x/h: forward within 2e-5 absolute (O(1) outputs), gradients within 1e-4 of the largest entry.
bf16 h (autocast flow: fp32 residual stream + bf16 branch) and bf16 x: same bounds against a
reference fed the same rounded inputs; bf16 outputs/gradients add one 2^-8 rounding.
Dropout: the keep mask is a deterministic function of (device seed word, element), identical in
forward and backward, with keep rate 1 - p."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from sceneverse_amd.modules.layers.fused_norm import _AddDropoutLN, add_dropout_layer_norm

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n,d,xdt,hdt", [(8320, 768, torch.float32, torch.bfloat16),
                                         (5120, 768, torch.float32, torch.float32),
                                         (333, 1024, torch.bfloat16, torch.bfloat16),
                                         (7, 256, torch.float32, torch.bfloat16),
                                         (4100, 2048, torch.bfloat16, torch.float32)])
def test_matches_torch(n, d, xdt, hdt):
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g).to(xdt)
    h = (torch.randn(n, d, generator=g) * 0.7).to(hdt)
    norm = nn.LayerNorm(d)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(d, generator=g))
        norm.bias.copy_(0.1 * torch.randn(d, generator=g))
    go = torch.randn(n, d, generator=g)
    xr, hr = x.float().clone().requires_grad_(True), h.float().clone().requires_grad_(True)
    ref = F.layer_norm(xr + hr, (d,), norm.weight, norm.bias, norm.eps)
    ref.backward(go)
    gw_ref, gb_ref = norm.weight.grad.clone(), norm.bias.grad.clone()
    norm.zero_grad()

    norm = norm.to(DEV)
    xg, hg = x.to(DEV).requires_grad_(True), h.to(DEV).requires_grad_(True)
    y = add_dropout_layer_norm(xg, hg, norm, 0.1, training=False)
    assert y.dtype == xdt
    y.backward(go.to(DEV).to(xdt))
    out_tol = 2e-5 if xdt == torch.float32 else 2 ** -7
    assert (y.float().cpu() - ref).abs().max().item() <= out_tol * max(1.0, ref.abs().max().item())

    def close(a, b, bf16):
        tol = (2 ** -6 if bf16 else 1e-4) * b.abs().max().item() + 1e-7
        assert (a.float().cpu() - b).abs().max().item() <= tol, ((a.float().cpu() - b).abs().max().item(), tol)

    close(xg.grad, xr.grad, xdt == torch.bfloat16)
    close(hg.grad, hr.grad, xdt == torch.bfloat16 or hdt == torch.bfloat16)
    close(norm.weight.grad, gw_ref, xdt == torch.bfloat16)
    close(norm.bias.grad, gb_ref, xdt == torch.bfloat16)


def test_dropout_mask_is_consistent_between_forward_and_backward():
    n, d, p = 512, 768, 0.25
    seed_dev = torch.tensor([12345], dtype=torch.int64, device=DEV)
    x = torch.zeros(n, d, device=DEV)
    h = torch.ones(n, d, device=DEV, requires_grad=True)
    gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    y = _AddDropoutLN.apply(x, h, gamma, beta, 1e-5, p, seed_dev, False)
    y2 = _AddDropoutLN.apply(x, h, gamma, beta, 1e-5, p, seed_dev, False)
    assert torch.equal(y, y2)
    # z = keep / (1 - p) in {0, 4/3}: the normalised row takes its minimum exactly at dropped elements
    dropped = y <= y.min(dim=1, keepdim=True).values + 1e-6
    rate = dropped.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    y.backward(torch.randn_like(y))
    assert (h.grad[dropped] == 0).all()
    assert (h.grad[~dropped] != 0).float().mean().item() > 0.99
    y3 = _AddDropoutLN.apply(x, h, gamma, beta, 1e-5, p, seed_dev + 1, False)
    assert not torch.equal(y, y3)


def test_unsupported_width_takes_the_torch_path():
    norm = nn.LayerNorm(100).to(DEV)
    x, h = torch.randn(4, 100, device=DEV), torch.randn(4, 100, device=DEV)
    y = add_dropout_layer_norm(x, h, norm, 0.0, False)
    assert torch.allclose(y, norm(x + h))


def test_bf16_copy_output_and_its_gradient():
    n, d = 640, 768
    g = torch.Generator().manual_seed(3)
    x, h = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g).to(torch.bfloat16)
    norm = nn.LayerNorm(d).to(DEV)
    w = torch.randn(d, 64, generator=g).to(DEV)
    xg, hg = x.to(DEV).requires_grad_(True), h.to(DEV).requires_grad_(True)
    y, y16 = add_dropout_layer_norm(xg, hg, norm, 0.0, False, want_bf16=True)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y.to(torch.bfloat16))
    loss = (y * 0.5).sum() + (y16.float() @ w).square().mean()       # gradient through BOTH outputs
    loss.backward()
    xr, hr = x.to(DEV).requires_grad_(True), h.to(DEV).float().requires_grad_(True)
    yr = F.layer_norm(xr + hr, (d,), norm.weight, norm.bias, norm.eps)
    ((yr * 0.5).sum() + (yr.to(torch.bfloat16).float() @ w).square().mean()).backward()
    for a, b in ((xg.grad, xr.grad), (hg.grad.float(), hr.grad)):
        assert (a - b).abs().max().item() <= 2 ** -6 * b.abs().max().item(), (a - b).abs().max().item()


@pytest.mark.parametrize("parts,d", [(1024, 768), (1, 256), (63, 768), (65, 1024), (800, 2048), (130, 512)])
def test_partial_row_reduction_is_exact_order_independent_and_reusable(parts, d):
    """gps_ln_reduce_partials: two-level sum with an arrival counter.  Same bits on every call (fixed summation
    order), counters left at zero (the scratch is reused by the next call), scratch-less form agrees."""
    from sceneverse_amd import _native
    from sceneverse_amd.modules.layers.fused_norm import _reduce_scratch
    lib = _native.load()
    g = torch.Generator().manual_seed(parts * 7 + d)
    part = torch.randn(2, parts, d, generator=g).to(DEV)
    ref = part.double().sum(1)
    scratch = _reduce_scratch(torch.device(DEV, torch.cuda.current_device()), d)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(4):
        out = torch.full((2, d), float("nan"), device=DEV)
        _native.check(lib.gps_ln_reduce_partials(parts, d, part.data_ptr(), out.data_ptr(), scratch.data_ptr(), stream),
                      "reduce")
        outs.append(out)
    torch.cuda.synchronize()
    tickets = scratch[16 * 2 * d:]
    assert int(tickets.abs().sum()) == 0
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert (outs[0].double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    plain = torch.empty((2, d), device=DEV)
    _native.check(lib.gps_ln_reduce_partials(parts, d, part.data_ptr(), plain.data_ptr(), None, stream), "reduce")
    assert (plain.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(64, 80, 768), (64, 768), (3, 1, 100), (5, 2048)])
def test_l2_normalize_matches_torch(shape):
    from sceneverse_amd.modules.layers.fused_norm import l2_normalize
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to("cuda")
    x[0] = 0                                                    # zero rows: clamped norm (y = 0, dx = dy / eps)
    x.requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    y = l2_normalize(x)
    ref = F.normalize(x2, dim=-1, p=2)
    assert (y - ref).abs().max().item() <= 1e-6
    w = torch.randn(*shape, generator=g).to("cuda")
    w[0] = 0                                                    # (dy / 1e-12 would overflow the comparison's scale)
    y.backward(w)
    ref.backward(w)
    assert (x.grad - x2.grad).abs().max().item() <= 2e-6 * max(1.0, x2.grad.abs().max().item())


def test_add_row_gradient_is_the_column_sum():
    from sceneverse_amd.modules.layers.fused_norm import add_row, column_sum
    g = torch.Generator().manual_seed(3)
    for rows in ((64, 50), (5120,), (7,), (1,)):
        x = torch.randn(*rows, 768, generator=g).to("cuda").requires_grad_(True)
        r = torch.randn(768, generator=g).to("cuda").requires_grad_(True)
        w = torch.randn(*rows, 768, generator=g).to("cuda")
        y = add_row(x, r)
        assert torch.equal(y, x + r)
        y.backward(w)
        assert torch.equal(x.grad, w)
        ref = w.double().reshape(-1, 768).sum(0)
        assert (r.grad.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
        assert torch.equal(column_sum(w), column_sum(w))        # deterministic


@pytest.mark.parametrize("want_bf16,p", [(True, 0.0), (False, 0.0), (True, 0.1)])
def test_post_addend_rides_on_the_layernorm_launch(want_bf16, p):
    """y = LayerNorm(x + dropout(h)) + post in one launch: values and every gradient (x, h, gamma, beta, post) against the
    torch chain; the bf16 copy is the rounded SUM; its cotangent reaches post as well."""
    g = torch.Generator().manual_seed(31)
    n, d = 520, 768
    norm = nn.LayerNorm(d).to("cuda")
    with torch.no_grad():
        norm.weight.add_(torch.randn(d, generator=g).to("cuda") * 0.1)
        norm.bias.add_(torch.randn(d, generator=g).to("cuda") * 0.1)
    x = torch.randn(n, d, generator=g).to("cuda").requires_grad_(True)
    h = torch.randn(n, d, generator=g).to("cuda").to(torch.bfloat16).requires_grad_(True)
    post = torch.randn(n, d, generator=g).to("cuda").requires_grad_(True)
    out = add_dropout_layer_norm(x, h, norm, p, training=p > 0, want_bf16=want_bf16, post=post)
    y, y16 = out if want_bf16 else (out, None)
    base = add_dropout_layer_norm(x.detach(), h.detach(), norm, 0.0, False)
    if p == 0.0:
        assert torch.equal(y, base + post.detach()) or (y - (base + post.detach())).abs().max().item() <= 1e-6
    if y16 is not None:
        assert torch.equal(y16, y.to(torch.bfloat16))
    wy = torch.randn(n, d, generator=g).to("cuda")
    wy16 = torch.randn(n, d, generator=g).to("cuda").to(torch.bfloat16)
    if y16 is not None:
        torch.autograd.backward([y, y16], [wy, wy16])
        want_post = wy + wy16.float()
    else:
        y.backward(wy)
        want_post = wy
    assert (post.grad - want_post).abs().max().item() <= 1e-6
    if p == 0.0:
        got = (x.grad.clone(), h.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone())
        for t in (x, h, norm.weight, norm.bias):
            t.grad = None
        ref = norm(x + h.float())
        ref.backward(want_post)
        for a, b in zip(got, (x.grad, h.grad, norm.weight.grad, norm.bias.grad)):
            scale = max(1.0, b.float().abs().max().item())
            assert (a.float() - b.float()).abs().max().item() <= 2e-2 * scale      # h's gradient is bf16


@pytest.mark.parametrize("rows", [None, 700])
def test_shared_post_addend_gradient_is_built_in_one_buffer(rows):
    """Three chained LayerNorm steps that add the SAME `post` (the per-layer re-added embeddings): with SharedPostGrad the
    backward launches accumulate its gradient in one buffer and only the first layer's node reports it -- equal to
    autograd's sum of three separate gradients (same addends; fp32 sums in another order: 1e-6 relative), all other
    gradients bit-equal; a second backward pass through a fresh graph starts from an empty buffer."""
    from sceneverse_amd.modules.layers.fused_norm import SharedPostGrad
    n, d = 1000, 768
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(n, d, generator=g).to(DEV)
    hs = [(0.5 * torch.randn(n, d, generator=g)).to(torch.bfloat16).to(DEV) for _ in range(3)]
    post0 = torch.randn(n, d, generator=g).to(DEV)
    norms = [nn.LayerNorm(d).to(DEV) for _ in range(3)]
    w = torch.randn(n, d, generator=g).to(DEV)
    rows_dev = torch.tensor([rows], dtype=torch.int32, device=DEV) if rows else None
    live = rows or n

    def run(shared):
        x, post = x0.clone().requires_grad_(True), post0.clone().requires_grad_(True)
        h_in = [h.clone().requires_grad_(True) for h in hs]
        for p_ in (q for m in norms for q in m.parameters()):
            p_.grad = None
        share = SharedPostGrad() if shared else None
        y = x
        for li in range(3):
            y = add_dropout_layer_norm(y, h_in[li], norms[li], 0.0, True, rows_dev=rows_dev, post=post,
                                       post_share=(share, li == 0) if shared else None)
        (y[:live] * w[:live]).sum().backward()
        assert share is None or share.buf is None
        return [x.grad[:live], post.grad[:live]] + [h.grad[:live] for h in h_in] + [q.grad for m in norms for q in m.parameters()]

    ref, got, again = run(False), run(True), run(True)
    for i, (a, b, c) in enumerate(zip(ref, got, again)):
        assert torch.equal(b, c), i
        if i == 1:
            assert (a - b).abs().max().item() <= 1e-6 * a.abs().max().item() + 1e-7
        else:
            assert torch.equal(a.float(), b.float()), i


def test_deferred_parameter_gradients_equal_the_autograd_ones():
    """Inside gemm.grouped_wgrads() the dgamma / dbeta of every fused residual-LayerNorm of a pass are reduced by ONE
    launch into gamma.grad / beta.grad (gps_ln_reduce_partials_grouped) instead of one reduce launch each: same values
    (same summation order) as the per-LayerNorm path -- several norms of different row counts (one and many row slices),
    a norm used twice in the pass, fresh and pre-existing .grad buffers."""
    from sceneverse_amd.modules.layers import gemm as G
    torch.manual_seed(11)
    norms = [nn.LayerNorm(768).to(DEV) for _ in range(4)]
    rows = [64, 5120, 22400, 8320]
    xs = [torch.randn(n, 768, device=DEV) for n in rows]
    hs = [torch.randn(n, 768, device=DEV).to(torch.bfloat16) for n in rows]

    def loss_fn():
        tot = 0.0
        for nm, x, h in zip(norms, xs, hs):
            tot = tot + add_dropout_layer_norm(x, h, nm, 0.0, False).square().mean()
        y = add_dropout_layer_norm(xs[1], hs[1], norms[0], 0.0, False)          # norms[0] a second time
        return tot + y.abs().mean()

    def grads(grouped, preset):
        for nm in norms:
            nm.zero_grad(set_to_none=True)
            if preset:
                for p in nm.parameters():
                    p.grad = torch.full_like(p, 0.5)
        with G.grouped_wgrads(grouped):
            loss_fn().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for nm in norms for p in nm.parameters()]

    for preset in (False, True):
        want, got = grads(False, preset), grads(True, preset)
        for a, b in zip(got, want):
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item() + 1e-9, (preset, (a - b).abs().max().item())
