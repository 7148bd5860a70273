"""The fused contrastive losses (csrc/gps_contrastive.hip, optim/loss/fused_contra.py) against the reference's own
composition (optim/loss/contra_loss.py:11-43: F.normalize, einsum / matmul, masked_fill, F.cross_entropy) evaluated in
fp64 on the CPU: loss within 1e-5 relative, every gradient within 2e-5 of its largest entry (fp32 kernels; sums of up
to 768 x 512 terms).  Repeated calls check that the arrival tickets are left at zero."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, what, rtol=2e-5):
    ref = ref.to(torch.float64)
    scale = max(ref.abs().max().item(), 1e-12)
    err = (got.detach().cpu().to(torch.float64) - ref).abs().max().item()
    assert err <= rtol * scale, (what, err, scale)


def _ref_text_obj(obj, text, labels, masks):
    o = F.normalize(obj, dim=-1, p=2)
    t = F.normalize(text, dim=-1, p=2)
    logits = torch.einsum('bod,bd->bo', o, t).masked_fill(masks.logical_not(), -float('inf'))
    return F.cross_entropy(logits, labels)


@pytest.mark.parametrize("B,O,D", [(64, 80, 768), (3, 5, 32), (17, 130, 256), (1, 1, 4)])
def test_text_obj_within_batch(B, O, D):
    from sceneverse_amd.optim.loss.fused_contra import text_obj_ce, text_obj_ce_usable
    g = torch.Generator().manual_seed(B * 1000 + O)
    obj = torch.randn(B, O, D, generator=g, dtype=torch.float64)
    text = torch.randn(B, D, generator=g, dtype=torch.float64)
    masks = torch.rand(B, O, generator=g) < 0.7
    labels = torch.randint(0, O, (B,), generator=g)
    masks[torch.arange(B), labels] = True                      # the target is a real object
    if B > 2:
        labels[1] = -100                                       # F.cross_entropy's default ignore_index: scene not counted
        obj[2, 0] = 0.0                                        # a zero row: norm clamped at eps
    ro, rt = obj.clone().requires_grad_(True), text.clone().requires_grad_(True)
    ref = _ref_text_obj(ro, rt, labels, masks)
    (ref * 1.7).backward()
    go = obj.float().to(DEV).requires_grad_(True)
    gt = text.float().to(DEV).requires_grad_(True)
    assert text_obj_ce_usable(go, gt, labels.to(DEV), masks.to(DEV))
    for rep in range(3):                                       # the ticket word must come back to zero every time
        go.grad = gt.grad = None
        got = text_obj_ce(go, gt, labels.to(DEV), masks.to(DEV))
        (got * 1.7).backward()
        torch.cuda.synchronize()
        assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-6, (rep, got.item(), ref.item())
        _close(go.grad, ro.grad, f"dobj rep {rep}")
        _close(gt.grad, rt.grad, f"dtext rep {rep}")
    if B > 2:
        assert go.grad[1].abs().max().item() == 0.0 and gt.grad[1].abs().max().item() == 0.0


def test_text_obj_module_routes_through_the_fused_kernels_and_matches_the_torch_branch():
    """TextObjWithinBatch.forward on GPU tensors: fused (default) against its own torch composition (_FUSED = False),
    through a non-contiguous text view (txt[:, 0]) and (B, 1) labels as the model hands them over."""
    from types import SimpleNamespace
    from sceneverse_amd.optim.loss import contra_loss as CL
    g = torch.Generator().manual_seed(5)
    B, O, D = 8, 20, 768
    obj = torch.randn(B, O, D, generator=g).to(DEV)
    txt = torch.randn(B, 6, D, generator=g).to(DEV)
    masks = (torch.rand(B, O, generator=g) < 0.8).to(DEV)
    labels = torch.randint(0, O, (B, 1), generator=g).to(DEV)
    masks[torch.arange(B), labels[:, 0]] = True
    mod = CL.TextObjWithinBatch(SimpleNamespace(num_gpu=1, task="pretrain"))
    outs = []
    for fused in (True, False):
        CL._FUSED = fused
        try:
            o, t = obj.clone().requires_grad_(True), txt.clone().requires_grad_(True)
            loss = mod({"intra_obj_embeds": o, "intra_text_embed": t[:, 0], "tgt_object_id": labels, "obj_masks": masks})
            loss.backward()
            outs.append((loss.item(), o.grad.clone(), t.grad.clone()))
        finally:
            CL._FUSED = True
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[1][0])
    _close(outs[0][1], outs[1][1].cpu(), "dobj", rtol=1e-4)
    _close(outs[0][2], outs[1][2].cpu(), "dtxt", rtol=1e-4)


def _ref_clip(a, b, scale, normalize, max_scale=100.0):
    s = torch.clamp(scale, max=max_scale)
    if normalize:
        a, b = F.normalize(a, dim=-1, p=2), F.normalize(b, dim=-1, p=2)
    labels = torch.arange(a.shape[0])
    return (F.cross_entropy(s * a @ b.t(), labels) + F.cross_entropy(s * b @ a.t(), labels)) / 2


@pytest.mark.parametrize("n,D,normalize,scale,feats", [(64, 768, True, 14.2857, True), (5, 32, True, 3.0, True),
                                                       (130, 256, False, 20.0, True), (512, 768, False, 14.2857, False),
                                                       (64, 768, True, 250.0, True), (1, 4, True, 2.0, True)])
def test_symmetric_clip_loss(n, D, normalize, scale, feats):
    from sceneverse_amd.optim.loss.fused_contra import clip_loss, clip_loss_usable
    g = torch.Generator().manual_seed(n + D)
    a = torch.randn(n, D, generator=g, dtype=torch.float64)
    b = torch.randn(n, D, generator=g, dtype=torch.float64)
    if not normalize:                                          # rows as a data-parallel run hands them over
        a, b = F.normalize(a, dim=-1), F.normalize(b, dim=-1)
    elif n > 2:
        a[1] = 0.0                                             # clamped norm
    ra, rb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rs = torch.tensor(scale, dtype=torch.float64, requires_grad=True)
    ref = _ref_clip(ra, rb, rs, normalize)
    (ref * 0.6).backward()
    ga = a.float().to(DEV).requires_grad_(feats)
    gb = b.float().to(DEV).requires_grad_(feats)
    gs = torch.nn.Parameter(torch.tensor(scale, device=DEV))
    assert clip_loss_usable(ga, gb, gs)
    for rep in range(3):
        ga.grad = gb.grad = gs.grad = None
        got = clip_loss(ga, gb, gs, normalize)
        (got * 0.6).backward()
        torch.cuda.synchronize()
        assert abs(got.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-6, (rep, got.item(), ref.item())
        if feats:
            _close(ga.grad, ra.grad, f"da rep {rep}", rtol=5e-5)
            _close(gb.grad, rb.grad, f"db rep {rep}", rtol=5e-5)
        else:
            assert ga.grad is None and gb.grad is None
        assert abs(gs.grad.item() - rs.grad.item()) <= 5e-5 * max(abs(rs.grad.item()), 1e-3), (gs.grad.item(), rs.grad.item())
    if scale > 100.0:
        assert gs.grad.item() == 0.0                           # the clamp is active: no gradient, as torch.clamp


def test_between_batch_modules_match_their_torch_branch():
    from types import SimpleNamespace
    from sceneverse_amd.optim.loss import contra_loss as CL
    g = torch.Generator().manual_seed(9)
    B, O, D = 16, 12, 768
    obj = torch.randn(B, O, D, generator=g).to(DEV)
    txt = torch.randn(B, 4, D, generator=g).to(DEV)
    labels = torch.randint(0, O, (B, 1), generator=g).to(DEV)
    cfg = SimpleNamespace(num_gpu=1, task="pretrain")
    for cls, make in ((CL.TextSceneBetweenBatch, lambda o, t: {"scene_embed": o.mean(dim=1), "scene_text_embed": t[:, 0]}),
                      (CL.TextObjBetweenBatch, lambda o, t: {"inter_obj_embeds": o, "inter_text_embed": t[:, 0],
                                                             "tgt_object_id": labels})):
        mod = cls(cfg).to(DEV)
        outs = []
        for fused in (True, False):
            CL._FUSED = fused
            try:
                o, t = obj.clone().requires_grad_(True), txt.clone().requires_grad_(True)
                mod.logit_scale.grad = None
                loss = mod(make(o, t))
                loss.backward()
                outs.append((loss.item(), o.grad.clone(), t.grad.clone(), mod.logit_scale.grad.item()))
            finally:
                CL._FUSED = True
        assert abs(outs[0][0] - outs[1][0]) <= 2e-5 * abs(outs[1][0]), cls.__name__
        _close(outs[0][1], outs[1][1].cpu(), cls.__name__ + " dobj", rtol=2e-4)
        _close(outs[0][2], outs[1][2].cpu(), cls.__name__ + " dtxt", rtol=2e-4)
        assert abs(outs[0][3] - outs[1][3]) <= 2e-4 * max(abs(outs[1][3]), 1e-3)
