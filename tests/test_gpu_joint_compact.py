"""The unified encoder over the VALID joint rows only (modules/grounding/unified_encoder.py `_forward_compact`,
gps_rows_plan) against the reference's formulation that runs every padded row (ref modules/grounding/unified_encoder.py:
147-177): the same results at every valid text position and object slot, zeros at the padded ones; the same losses and
parameter gradients (no head or loss reads a padded row)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_rows_plan_matches_a_stable_sort():
    from sceneverse_amd import _native
    lib = _native.load()
    g = torch.Generator().manual_seed(3)
    for n_seq, L in ((1, 1), (7, 13), (64, 130), (33, 257)):
        valid = (torch.rand(n_seq, L, generator=g) < 0.6)
        valid[0] = False                                  # an empty sequence
        if n_seq > 2:
            valid[2] = True                               # a full one
        v = valid.to(DEV).reshape(-1)
        n = n_seq * L
        perm = torch.empty(n, dtype=torch.int64, device=DEV)
        inv = torch.empty(n, dtype=torch.int64, device=DEV)
        cu = torch.empty(n_seq + 1, dtype=torch.int32, device=DEV)
        n_live = torch.empty(1, dtype=torch.int32, device=DEV)
        _native.check(lib.gps_rows_plan(n_seq, L, v.view(torch.uint8).data_ptr(), perm.data_ptr(), inv.data_ptr(), cu.data_ptr(),
                                        n_live.data_ptr(), torch.cuda.current_stream().cuda_stream), "rows_plan")
        ref_perm = torch.argsort(v.logical_not().to(torch.uint8), stable=True)
        assert torch.equal(perm, ref_perm)
        assert torch.equal(inv[perm], torch.arange(n, device=DEV))
        lens = valid.sum(1)
        ref_cu = torch.zeros(n_seq + 1, dtype=torch.int64)
        ref_cu[1:] = torch.cumsum(lens, 0)
        assert torch.equal(cu.cpu().long(), ref_cu)
        assert int(n_live.item()) == int(valid.sum())


def _plan(valid):
    from sceneverse_amd import _native
    n_seq, L = valid.shape
    n = n_seq * L
    v = valid.to(DEV).reshape(-1)
    perm = torch.empty(n, dtype=torch.int64, device=DEV)
    inv = torch.empty(n, dtype=torch.int64, device=DEV)
    cu = torch.empty(n_seq + 1, dtype=torch.int32, device=DEV)
    n_live = torch.empty(1, dtype=torch.int32, device=DEV)
    _native.check(_native.load().gps_rows_plan(n_seq, L, v.view(torch.uint8).data_ptr(), perm.data_ptr(), inv.data_ptr(),
                                              cu.data_ptr(), n_live.data_ptr(), torch.cuda.current_stream().cuda_stream), "rows_plan")
    return v, perm, inv, n_live


@pytest.mark.parametrize("n_seq,La,Lb,D", [(7, 5, 8, 768), (64, 50, 80, 768), (3, 1, 4, 8)])
def test_pack2_unpack2_equal_cat_gather_and_split(n_seq, La, Lb, D):
    """gps_joint_embed_* through _JointEmbed and gps_rows_unpack2 / gps_rows_pack2 through _UnpackJoint against cat + add +
    index_select (zeros in the dead rows) and gather + where + split: copies and single fp32 adds in the same order, so values and
    gradients are bit-equal."""
    from sceneverse_amd.modules.grounding.unified_encoder import _JointEmbed, _UnpackJoint
    g = torch.Generator().manual_seed(31)
    T = La + Lb
    valid = torch.rand(n_seq, T, generator=g) < 0.6
    valid[0] = False
    valid[-1] = True
    v, perm, inv, n_live = _plan(valid)
    n, live = n_seq * T, int(valid.sum())
    v8 = v.view(torch.uint8)
    a = torch.randn(n_seq, La, D, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(n_seq, Lb, D, generator=g).to(DEV).requires_grad_(True)
    ea = torch.randn(n_seq, La, D, generator=g).to(DEV).requires_grad_(True)
    eb = torch.randn(n_seq, Lb, D, generator=g).to(DEV).requires_grad_(True)
    x, e, x16 = _JointEmbed.apply(a, b, ea, eb, perm, inv, v8, n_live)
    flat = torch.cat((a.detach(), b.detach()), dim=1).reshape(n, D)
    flat_e = torch.cat((ea.detach(), eb.detach()), dim=1).reshape(n, D)
    ref_x, ref_e = (flat + flat_e).index_select(0, perm), flat_e.index_select(0, perm)
    ref_x[live:] = 0
    ref_e[live:] = 0
    assert torch.equal(x.detach(), ref_x) and torch.equal(e.detach(), ref_e) and torch.equal(x16.detach(), ref_x.to(torch.bfloat16))
    w, we = torch.randn(n, D, generator=g).to(DEV), torch.randn(n, D, generator=g).to(DEV)
    w16 = torch.randn(n, D, generator=g).to(torch.bfloat16).to(DEV)
    nanify = lambda t: torch.cat((t[:live], torch.full_like(t[live:], float("nan"))))  # noqa: E731
    ga, gb, gea, geb = torch.autograd.grad((x, e, x16), (a, b, ea, eb), (nanify(w), nanify(we), nanify(w16)))

    def flat_grad(rows):
        out = torch.zeros(n, D, device=DEV)
        out[perm[:live]] = rows[:live]
        return out.view(n_seq, T, D)
    gj, gje = flat_grad(w + w16.float()), flat_grad((w + w16.float()) + we)
    assert torch.equal(ga, gj[:, :La]) and torch.equal(gb, gj[:, La:])
    assert torch.equal(gea, gje[:, :La]) and torch.equal(geb, gje[:, La:])
    ga2, gea2 = torch.autograd.grad(_JointEmbed.apply(a, b, ea, eb, perm, inv, v8, n_live)[0], (a, ea), nanify(w))   # x alone
    assert torch.equal(ga2, flat_grad(w)[:, :La]) and torch.equal(gea2, ga2)
    y = torch.randn(n, D, generator=g).to(DEV)
    y[live:] = float("nan")
    y.requires_grad_(True)
    oa, ob = _UnpackJoint.apply(y, perm, inv, v8, n_live, n_seq, La, Lb)
    assert oa.is_contiguous() and ob.is_contiguous()
    ref = torch.where(v[:, None], y.detach().nan_to_num(0.0).index_select(0, inv), torch.zeros((), device=DEV)).view(n_seq, T, D)
    assert torch.equal(oa.detach(), ref[:, :La]) and torch.equal(ob.detach(), ref[:, La:])
    wa, wb = torch.randn(n_seq, La, D, generator=g).to(DEV), torch.randn(n_seq, Lb, D, generator=g).to(DEV)
    (gy,) = torch.autograd.grad((oa, ob), y, (wa, wb))
    wf = torch.cat((wa, wb), dim=1).reshape(n, D)
    ref_g = torch.zeros(n, D, device=DEV)
    ref_g[:live] = wf.index_select(0, perm[:live])
    assert torch.equal(gy, ref_g)
    (gy1,) = torch.autograd.grad(_UnpackJoint.apply(y, perm, inv, v8, n_live, n_seq, La, Lb)[1], y, wb)     # one part unused
    wf1 = torch.cat((torch.zeros_like(wa), wb), dim=1).reshape(n, D)
    ref_g1 = torch.zeros(n, D, device=DEV)
    ref_g1[:live] = wf1.index_select(0, perm[:live])
    assert torch.equal(gy1, ref_g1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_select_rows_equals_index_select_with_a_dead_row_mask(dtype):
    """gps_rows_move through select_rows (the [CLS]-tail selection of the variable-length text path): values and gradients
    bit-equal to index_select + _ZeroDeadRows on the live rows; dead rows zero; dead entries of `sel` may alias live ones."""
    from sceneverse_amd.modules.language.bert import _ZeroDeadRows, select_rows
    g = torch.Generator().manual_seed(23)
    T, D, n_first, n_full = 900, 768, 40, 500
    x0 = torch.randn(T, D, generator=g).to(dtype).to(DEV)
    firsts = torch.randperm(T - n_full, generator=g)[:n_first] + n_full          # "first tokens" past the fully read rows
    sel = torch.cat([firsts, torch.arange(n_full + 100)]).to(DEV)                 # the last 100 entries are dead (and alias)
    live = n_first + n_full
    rows_live = torch.tensor([live], dtype=torch.int32, device=DEV)
    w = torch.randn(sel.shape[0], D, generator=g).to(dtype).to(DEV)
    w_nan = w.clone()
    w_nan[live:] = float("nan")
    xa = x0.clone().requires_grad_(True)
    ya = select_rows(xa, sel, rows_live)
    (ga,) = torch.autograd.grad(ya, xa, w_nan)
    xb = x0.clone().requires_grad_(True)
    yb = _ZeroDeadRows.apply(xb.index_select(0, sel), rows_live)
    (gb,) = torch.autograd.grad(yb, xb, w)
    assert torch.equal(ya[:live], yb[:live]) and torch.equal(ya[live:], torch.zeros_like(ya[live:]))
    assert torch.equal(ga, gb)
    # argument checks of the C entry
    from sceneverse_amd import _native
    lib, s = _native.load(), torch.cuda.current_stream().cuda_stream
    out = torch.empty(4, 8, device=DEV)
    src = torch.zeros(4, 8, device=DEV)
    assert lib.gps_rows_move(4, 4, 4, 24, src.data_ptr(), None, out.data_ptr(), None, None, 0, s) == _native.GPS_ERR_UNSUPPORTED
    assert lib.gps_rows_move(4, 4, 4, 32, None, None, out.data_ptr(), None, None, 0, s) == _native.GPS_ERR_INVALID_ARGUMENT


def test_permute_live_rows_gradient_is_one_scatter_with_zeroed_dead_rows():
    """_PermuteLiveRows (masked-LM head: labelled rows first): the gradient through gps_rows_move's scatter form with
    zero_dead equals where(live) + index_copy_ -- dead rows arrive as exact zeros even when the incoming rows are NaN."""
    from sceneverse_amd.optim.loss.fused_lm_loss import _PermuteLiveRows
    g = torch.Generator().manual_seed(5)
    for n, D, live in ((3200, 768, 471), (37, 8, 0), (64, 768, 64)):
        x = torch.randn(n, D, generator=g).to(DEV).requires_grad_(True)
        perm = torch.randperm(n, generator=g).to(DEV)
        n_valid = torch.tensor([live], dtype=torch.int32, device=DEV)
        w = torch.randn(n, D, generator=g).to(DEV)
        w[live:] = float("nan")                                   # rows the extent-aware kernels never wrote
        y = _PermuteLiveRows.apply(x, perm, n_valid)
        assert torch.equal(y, x.detach().index_select(0, perm))
        (gx,) = torch.autograd.grad(y, x, w)
        ref = torch.zeros(n, D, device=DEV)
        ref[perm[:live]] = w[:live]
        assert torch.equal(gx, ref)


def test_broadcast_row_equals_add_row_on_zeros():
    from sceneverse_amd.modules.layers.fused_norm import add_row, broadcast_row
    g = torch.Generator().manual_seed(6)
    row_a = torch.randn(768, generator=g).to(DEV).requires_grad_(True)
    row_b = row_a.detach().clone().requires_grad_(True)
    w = torch.randn(64, 50, 768, generator=g).to(DEV)
    ya = broadcast_row(row_a, (64, 50))
    yb = add_row(torch.zeros(64, 50, 768, device=DEV), row_b)
    assert ya.is_contiguous() and torch.equal(ya, yb)
    (ga,), (gb,) = torch.autograd.grad(ya, row_a, w), torch.autograd.grad(yb, row_b, w)
    assert torch.equal(ga, gb)


def test_gps_model_compact_joint_rows_equal_padded_rows(golden_cpu):
    from oracle.param_fill import fill_params
    from sceneverse_amd.model.build import build_model
    from sceneverse_amd.modules.grounding import unified_encoder as UE
    from sceneverse_amd.optim.loss import Loss
    from util import clone_batch, gps_cfg, lang_dir
    fx = golden_cpu
    model = build_model(gps_cfg(lang_dir(fx["seed"]), freeze=True))
    fill_params(model, fx["seed"])
    model = model.to(DEV).eval()                     # dropout off: the two forms index their masks differently
    loss_mod = Loss(model.cfg).to(DEV)
    batch = clone_batch(fx["batch"], DEV)
    assert bool((batch["obj_masks"] == 0).any()) or bool((batch["txt_masks"] == 0).any()), "the fixture must hold padded rows"
    res = {}
    for compact in (False, True):
        UE.set_compact_joint_rows(compact)
        try:
            for p in model.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(clone_batch(fx["batch"], DEV))
                total, losses = loss_mod(out)
            total.backward()
        finally:
            UE.set_compact_joint_rows(True)
        res[compact] = (total.detach().float(), {k: out[k].detach().float() for k in ("og3d_logits", "intra_text_embed", "intra_obj_embeds")},
                        {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None},
                        {k: float(v.detach()) for k, v in losses.items()})
    om = batch["obj_masks"].bool()
    ref, got = res[False], res[True]
    for k, v in ref[3].items():
        assert abs(got[3][k] - v) <= 3e-3 * max(1.0, abs(v)), (k, got[3][k], v)
    scale = float(ref[1]["intra_obj_embeds"].abs().max())
    torch.testing.assert_close(got[1]["intra_obj_embeds"][om], ref[1]["intra_obj_embeds"][om], rtol=2e-2, atol=2e-2 * scale)
    assert float(got[1]["intra_obj_embeds"][~om].abs().max() if (~om).any() else 0.0) == 0.0      # padded slots: zeros
    torch.testing.assert_close(got[1]["intra_text_embed"], ref[1]["intra_text_embed"], rtol=2e-2,
                               atol=2e-2 * float(ref[1]["intra_text_embed"].abs().max()))
    lscale = float(ref[1]["og3d_logits"][om].abs().max())
    torch.testing.assert_close(got[1]["og3d_logits"][om], ref[1]["og3d_logits"][om], rtol=2e-2, atol=2e-2 * lscale)
    assert got[2].keys() == ref[2].keys()
    # (a key bias shifts every logit of a query by the same amount: its exact gradient is 0 and what is stored is rounding
    # noise -- hence the absolute slack, 1e-3 of the largest per-element RMS over all parameters)
    rms = lambda t: float(t.norm()) / t.numel() ** 0.5  # noqa: E731
    slack = 1e-3 * max(rms(g) for g in ref[2].values())
    for n, g in ref[2].items():
        assert rms(got[2][n] - g) <= 4e-2 * rms(g) + slack, n
