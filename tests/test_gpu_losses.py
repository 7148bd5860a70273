"""Row-sparse cross-entropy kernel (gps_masked_ce_*) against F.cross_entropy(ignore_index=-1), the
call the reference makes in lm_cls_loss (optim/loss/loss.py:56-61).
fp32 logits: loss within 1e-5 relative, gradient within 1e-6 absolute (exp/log intrinsics).
bf16 logits: both sides read the same bf16 values; the kernel rounds its gradient to bf16 once
(torch computes in fp32 from the up-cast): gradient within 2^-8 relative of the largest entry."""
import pytest
import torch
import torch.nn.functional as F

from sceneverse_amd.optim.loss.masked_ce import masked_cross_entropy

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n,v,dtype", [(3200, 30522, torch.bfloat16), (257, 30522, torch.float32),
                                       (64, 607, torch.float32), (5, 7, torch.bfloat16)])
def test_matches_torch_cross_entropy(n, v, dtype):
    g = torch.Generator().manual_seed(n + v)
    logits = (torch.randn(n, v, generator=g) * 3).to(dtype)
    labels = torch.randint(0, v, (n,), generator=g)
    labels[torch.rand(n, generator=g) < 0.85] = -1
    labels[0] = 3 % v                                     # at least one labelled row
    ref_in = logits.float().clone().requires_grad_(True)
    ref = F.cross_entropy(ref_in, labels, ignore_index=-1)
    ref.backward()
    x = logits.to(DEV).requires_grad_(True)
    got = masked_cross_entropy(x, labels.to(DEV), ignore_index=-1)
    got.backward()
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-6, (got.item(), ref.item())
    gd, gr = x.grad.float().cpu(), ref_in.grad
    tol = 1e-6 if dtype == torch.float32 else 2 ** -8 * gr.abs().max().item()
    assert (gd - gr).abs().max().item() <= tol, ((gd - gr).abs().max().item(), tol)
    assert gd[labels == -1].abs().max().item() == 0.0      # ignored rows: exact zeros


def test_lm_cls_loss_uses_it_and_handles_3d_labels():
    from sceneverse_amd.optim.loss.loss import lm_cls_loss
    from sceneverse_amd.pointnet2 import _ext
    logits = torch.randn(4, 50, 30522, device=DEV, dtype=torch.bfloat16)
    labels = torch.full((4, 50), -1, dtype=torch.long, device=DEV)
    labels[:, 3] = 17
    _ext.profile_start()
    val = lm_cls_loss({"txt_lm_cls_logits": logits, "masked_lm_labels": labels})
    rec = _ext.profile_stop()
    assert "masked_ce_forward" in rec
    ref = F.cross_entropy(logits.float().permute(0, 2, 1), labels, ignore_index=-1)
    assert abs(val.item() - ref.item()) < 1e-4


@pytest.mark.parametrize("n,V,frac", [(3200, 30522, 0.15), (130, 607, 0.5), (640, 30522, 0.0), (256, 1000, 1.0)])
def test_fused_lm_head_loss_matches_linear_plus_cross_entropy(n, V, frac):
    """sparse_lm_loss (labelled rows only: permutation + device-side extents in the three GEMMs) against
    F.cross_entropy(F.linear(h, W, b), labels, ignore_index=-1) on the same bf16-rounded operands: loss and the
    gradients of h, W, b (rows without a label get exactly zero)."""
    from sceneverse_amd.optim.loss.fused_lm_loss import sparse_lm_loss
    D = 768
    g = torch.Generator().manual_seed(n + V)
    h = (torch.randn(n, D, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    w = (torch.randn(V, D, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = (torch.randn(V, generator=g) * 0.1).to(DEV).requires_grad_(True)
    labels = torch.randint(0, V, (n,), generator=g)
    labels[torch.rand(n, generator=g) >= frac] = -1
    labels = labels.to(DEV)
    hf = h.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = sparse_lm_loss(hf, w, b, labels, -1)
    if frac == 0.0:
        assert torch.isnan(loss).item()                     # no labelled token: 0 / 0 like F.cross_entropy
        return
    loss.backward()
    gh, gw, gb = hf.grad.float(), w.grad.clone(), b.grad.clone()
    w.grad = b.grad = None
    hr = h.float().clone().requires_grad_(True)
    ref = F.cross_entropy(F.linear(hr, w.to(torch.bfloat16).float(), b), labels, ignore_index=-1)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-3 * abs(ref.item()) + 1e-4, (loss.item(), ref.item())

    def rel(a, r):
        return ((a - r).norm() / (r.norm() + 1e-12)).item()
    assert rel(gh, hr.grad) <= 2e-2 and rel(gw, w.grad) <= 2e-2 and rel(gb, b.grad) <= 2e-2, \
        (rel(gh, hr.grad), rel(gw, w.grad), rel(gb, b.grad))
    assert gh[labels < 0].abs().max().item() == 0.0 if (labels < 0).any() else True


@pytest.mark.parametrize("frac", [0.15, 1.0])
def test_lm_head_with_the_transform_behind_the_row_selection(frac):
    """BertLMPredictionHead in training mode inside fused_lm_loss(): the head's dense -> gelu -> LayerNorm runs on the
    labelled rows only (LazyLMLogits with `transform`), then decoder + cross-entropy.  Against the reference formulation
    F.cross_entropy(decoder(LayerNorm(gelu(dense(x)))) + bias) in fp32 on the same parameters: the loss and the
    gradients of the hidden states (exactly zero on unlabelled rows), of the transform's and of the decoder's parameters."""
    import copy
    from sceneverse_amd.modules.heads.pretrain_head import BertLMPredictionHead, fused_lm_loss
    from sceneverse_amd.optim.loss.fused_lm_loss import LazyLMLogits
    B, L, D, V = 64, 50, 768, 30522
    torch.manual_seed(3)
    head = BertLMPredictionHead(D, V).to(DEV).train()
    with torch.no_grad():
        head.transform.LayerNorm.weight.add_(0.1 * torch.randn(D, device=DEV))
        head.transform.LayerNorm.bias.add_(0.1 * torch.randn(D, device=DEV))
        head.bias.add_(0.1 * torch.randn(V, device=DEV))
    ref_head = copy.deepcopy(head).float()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, D, generator=g).to(DEV)
    labels = torch.randint(0, V, (B, L), generator=g)
    labels[torch.rand(B, L, generator=g) >= frac] = -1
    labels = labels.to(DEV)
    xf = x.clone().requires_grad_(True)
    with fused_lm_loss(True), torch.autocast("cuda", dtype=torch.bfloat16):
        lazy = head(xf)
        assert isinstance(lazy, LazyLMLogits) and lazy.transform is head.transform
        loss = lazy.loss(labels, -1)
    loss.backward()
    xr = x.clone().requires_grad_(True)
    ref = F.cross_entropy(ref_head(xr).permute(0, 2, 1), labels, ignore_index=-1)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 5e-3 * abs(ref.item()) + 1e-4, (loss.item(), ref.item())

    def rel(a, r):
        return ((a.float() - r.float()).norm() / (r.float().norm() + 1e-12)).item()
    assert rel(xf.grad, xr.grad) <= 3e-2, rel(xf.grad, xr.grad)
    assert xf.grad[labels < 0].abs().max().item() == 0.0 if (labels < 0).any() else True
    for (n, p), (_, q) in zip(head.named_parameters(), ref_head.named_parameters()):
        assert p.grad is not None and rel(p.grad, q.grad) <= 3e-2, (n, rel(p.grad, q.grad))
    # evaluation / metrics: the full logits through the same object
    with torch.no_grad():
        _close = (lazy.materialize().float() - ref_head(x)).abs().max().item()
    assert _close <= 5e-2, _close


@pytest.mark.parametrize("n,V,frac", [(3200, 30522, 0.15), (1, 7, 1.0), (63, 100, 0.5), (1024, 30522, 0.0), (1025, 607, 1.0),
                                      (5000, 30522, 0.3), (0, 10, 0.0)])
def test_row_plan_kernel_is_the_stable_partition(n, V, frac):
    """gps_lm_row_plan against the torch chain it replaces: valid mask -> count -> stable argsort -> permuted labels.
    Out-of-range labels and ignore_index rows are the second class.  Bit-exact (integers)."""
    from sceneverse_amd.optim.loss.fused_lm_loss import row_plan
    g = torch.Generator().manual_seed(17 * n + V)
    labels = torch.randint(0, V, (n,), generator=g)
    labels[torch.rand(n, generator=g) >= frac] = -1
    if n > 8:
        labels[5] = V + 3                                  # out of range: treated as ignored
        labels[7] = -5
    lab = labels.to(DEV)
    perm, lp, n_valid = row_plan(lab, V, -1)
    valid = (lab != -1) & (lab >= 0) & (lab < V)
    ref_perm = torch.argsort(valid.logical_not().to(torch.uint8), stable=True)
    ref_lp = torch.where(valid, lab, torch.full_like(lab, -1)).index_select(0, ref_perm)
    assert int(n_valid.item()) == int(valid.sum().item())
    assert torch.equal(perm, ref_perm) and torch.equal(lp, ref_lp)


@pytest.mark.parametrize("n,V,n_live", [(640, 30522, 97), (64, 607, 64), (33, 1000, 0), (200, 30522, 200)])
def test_cross_entropy_with_a_device_side_row_extent(n, V, n_live):
    """gps_masked_ce_{forward,backward}_rows (16-byte bf16 accesses, rows past *rows_dev dead) against the scalar entry
    points on the same padded bf16 rows: per-row loss / lse within 1e-5 relative, gradient within one bf16 rounding of the
    scalar form's; pad columns of live rows are zero; dead rows of dlogits are NOT touched (sentinel survives) and their
    per-row outputs are 0."""
    from sceneverse_amd import _native
    lib = _native.load()
    Vp = (V + 7) // 8 * 8
    g = torch.Generator().manual_seed(n + V)
    logits = torch.zeros(n, Vp, dtype=torch.bfloat16)
    logits[:, :V] = (torch.randn(n, V, generator=g) * 3).to(torch.bfloat16)
    logits[:, V:] = 77.0                                   # pad columns must never be read as classes
    labels = torch.randint(0, V, (n,), generator=g)
    labels[n_live:] = -1
    if n_live > 4:
        labels[2] = -1                                     # an ignored row inside the live range
    logits, labels = logits.to(DEV), labels.to(DEV)
    rows_dev = torch.tensor([n_live], dtype=torch.int32, device=DEV)
    grad_rows = (torch.rand(n, generator=g) + 0.5).to(DEV)
    st = torch.cuda.current_stream().cuda_stream

    ticket = torch.zeros(1, dtype=torch.int32, device=DEV)

    def run(fwd, bwd, extra, fused_mean=False):
        loss = torch.full((n,), -7.0, device=DEV)
        lse = torch.full((n,), -7.0, device=DEV)
        mean = torch.full((2,), -7.0, device=DEV)
        d = torch.full((n, Vp), 5.0, dtype=torch.bfloat16, device=DEV)
        assert fwd(n, V, 1, logits.data_ptr(), Vp, labels.data_ptr(), -1, *extra, loss.data_ptr(), lse.data_ptr(),
                   mean.data_ptr() if fused_mean else None, ticket.data_ptr() if fused_mean else None, st) == 0
        if fused_mean:                                     # every row's factor = *grad_out / count
            gout = torch.tensor([0.7], device=DEV)
            assert bwd(n, V, 1, logits.data_ptr(), Vp, labels.data_ptr(), -1, *extra, lse.data_ptr(), None, gout.data_ptr(),
                       mean[1:].data_ptr(), d.data_ptr(), Vp, st) == 0
        else:
            assert bwd(n, V, 1, logits.data_ptr(), Vp, labels.data_ptr(), -1, *extra, lse.data_ptr(), grad_rows.data_ptr(),
                       None, None, d.data_ptr(), Vp, st) == 0
        torch.cuda.synchronize()
        assert int(ticket.item()) == 0                     # the arrival counter is left at zero
        return (loss, lse, d, mean) if fused_mean else (loss, lse, d)

    # reference: the scalar kernels on an UNPADDED copy (pitch V is not a multiple of 8 for these V, or is forced scalar
    # by the odd pitch Vp + 1)
    wide = torch.zeros(n, Vp + 1, dtype=torch.bfloat16, device=DEV)
    wide[:, :Vp] = logits
    loss_r = torch.empty(n, device=DEV); lse_r = torch.empty(n, device=DEV)
    d_r = torch.zeros(n, Vp + 1, dtype=torch.bfloat16, device=DEV)
    assert lib.gps_masked_ce_forward(n, V, 1, wide.data_ptr(), Vp + 1, labels.data_ptr(), -1, loss_r.data_ptr(),
                                     lse_r.data_ptr(), st) == 0
    assert lib.gps_masked_ce_backward(n, V, 1, wide.data_ptr(), Vp + 1, labels.data_ptr(), -1, lse_r.data_ptr(),
                                      grad_rows.data_ptr(), d_r.data_ptr(), Vp + 1, st) == 0
    loss, lse, d = run(lib.gps_masked_ce_forward_rows, lib.gps_masked_ce_backward_rows, (rows_dev.data_ptr(),))
    live = slice(0, n_live)
    assert torch.allclose(loss[live], loss_r[live], rtol=1e-5, atol=1e-5)
    assert torch.allclose(lse[live], lse_r[live], rtol=1e-5, atol=1e-5)
    assert (loss[n_live:] == 0).all() and (lse[n_live:] == 0).all()
    dl, dr = d[live, :V].float(), d_r[live, :V].float()
    scale = max(dr.abs().max().item(), 1e-6) if n_live else 1.0
    assert n_live == 0 or (dl - dr).abs().max().item() <= 2 ** -7 * scale
    assert n_live == 0 or (d[live, V:] == 0).all()
    assert (d[n_live:] == 5.0).all()                       # dead rows: never written
    # without an extent the 16-byte forms cover every row (ignored ones zero-filled)
    loss2, lse2, d2 = run(lib.gps_masked_ce_forward_rows, lib.gps_masked_ce_backward_rows, (None,))
    assert torch.equal(loss2[live], loss[live]) and (d2[n_live:] == 0).all()
    # mean by the last workgroup to arrive + the scalar upstream gradient: equal to sum / count and to uniform row factors
    valid = (labels[:n_live] != -1)
    cnt = int(valid.sum().item())
    for rep in range(2):
        loss3, lse3, d3, mean3 = run(lib.gps_masked_ce_forward_rows, lib.gps_masked_ce_backward_rows, (rows_dev.data_ptr(),), True)
        assert int(mean3[1].item()) == cnt
        if cnt:
            assert abs(mean3[0].item() - loss[live].sum().item() / cnt) <= 1e-5 * abs(mean3[0].item()) + 1e-6
            want = (d[live, :V].float() / grad_rows[live, None]) * (0.7 / cnt)
            got3 = d3[live, :V].float()
            sc = max(want.abs().max().item(), 1e-9)
            assert (got3[valid] - want[valid]).abs().max().item() <= 2 ** -6 * sc
    # a layout the 16-byte form cannot take + an extent -> refused, not silently wrong
    assert lib.gps_masked_ce_forward_rows(n, V, 1, wide.data_ptr(), Vp + 1, labels.data_ptr(), -1, rows_dev.data_ptr(),
                                          loss_r.data_ptr(), lse_r.data_ptr(), None, None, st) == -2
