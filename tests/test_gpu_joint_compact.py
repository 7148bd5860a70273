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


@pytest.mark.parametrize("n_seq,L,D", [(7, 13, 768), (64, 130, 768), (3, 5, 8), (1, 1, 4)])
def test_rows_gather_packs_and_unpacks_with_exact_gradients(n_seq, L, D):
    """gps_rows_gather through _GatherRows: pack (perm, n_live) == index_select with zeros in the dead rows, unpack (inv,
    valid) == the padded layout with zeros at invalid positions; both are copies, so values AND gradients are bit-equal to
    the torch formulation (index_select / where; gradient = zero-fill + index_add_ of distinct rows)."""
    from sceneverse_amd.modules.language.bert import _GatherRows
    g = torch.Generator().manual_seed(11)
    valid = torch.rand(n_seq, L, generator=g) < 0.6
    if n_seq > 2:
        valid[0] = False
        valid[2] = True
    v, perm, inv, n_live = _plan(valid)
    n, live = n_seq * L, int(valid.sum())
    v8 = v.view(torch.uint8)
    x = torch.randn(n, D, generator=g).to(DEV).requires_grad_(True)
    # pack
    packed = _GatherRows.apply(x, perm, None, n_live, inv, v8, None)
    ref = x.detach().index_select(0, perm)
    ref[live:] = 0
    assert torch.equal(packed.detach(), ref)
    w = torch.randn(n, D, generator=g).to(DEV)
    w_nan = w.clone()
    w_nan[live:] = float("nan")                       # gradients of dead packed rows are undefined memory in the step
    (gx,) = torch.autograd.grad(packed, x, w_nan)
    ref_g = torch.zeros(n, D, device=DEV)
    ref_g[perm[:live]] = w[:live]
    assert torch.equal(gx, ref_g)
    # unpack
    y = torch.randn(n, D, generator=g).to(DEV)
    y[live:] = float("nan")                           # rows no kernel wrote
    y.requires_grad_(True)
    out = _GatherRows.apply(y, inv, v8, None, perm, None, n_live)
    ref = torch.where(v[:, None], y.detach().nan_to_num(0.0).index_select(0, inv), torch.zeros((), device=DEV))
    assert torch.equal(out.detach(), ref)
    (gy,) = torch.autograd.grad(out, y, w)
    ref_g = torch.zeros(n, D, device=DEV)
    ref_g[:live] = w.index_select(0, perm[:live])
    assert torch.equal(gy, ref_g)
    # bf16 copy and argument checks of the C entry
    from sceneverse_amd import _native
    o32, o16 = _GatherRows._run(x.detach(), perm, None, n_live, want16=True)
    assert torch.equal(o16, o32.to(torch.bfloat16))
    lib = _native.load()
    s = torch.cuda.current_stream().cuda_stream
    xd = x.detach()
    assert lib.gps_rows_gather(n, n, 6, xd.data_ptr(), perm.data_ptr(), None, None, o32.data_ptr(), None, s) == _native.GPS_ERR_UNSUPPORTED
    assert lib.gps_rows_gather(n, n, D, None, perm.data_ptr(), None, None, o32.data_ptr(), None, s) == _native.GPS_ERR_INVALID_ARGUMENT
    bad = perm.clone()
    bad[0] = n + 5                                     # an index outside the source reads zeros, never memory
    o, _ = _GatherRows._run(xd, bad, None, None)
    assert torch.equal(o[0], torch.zeros(D, device=DEV)) and torch.equal(o[1:], xd.index_select(0, perm)[1:])


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
