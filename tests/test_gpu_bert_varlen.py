"""Variable-length form of the BERT fast path (modules/language/bert.py `_VARLEN`): valid tokens compacted to the
front of one row batch, GEMMs / LayerNorms bounded by a device-side row count, attention over per-sequence row offsets
(gps_attn_args.cu_rows) -- against the padded form of the same fast path and against HuggingFace's own BertModel.
Everything the model reads (valid positions; [CLS] of the caption) and every parameter gradient must agree:
  * outputs at valid positions: |diff| <= 0.06 vs HF (the bound of tests/test_gpu_model.py's BERT check), <= 2e-2 vs the
    padded fast path (same kernels, same bf16 roundings; only reduction splits differ);
  * parameter gradients: relative L2 <= 2e-2 vs the padded fast path;
  * padded positions of the returned (B, L, D) tensor are exactly zero; rows past the valid count never leak NaNs
    (the scratch rows are poisoned with NaN before the run)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _encoder(seed=0):
    from sceneverse_amd.common.config import ConfigNode
    from sceneverse_amd.modules.language.bert import BERTLanguageEncoder
    torch.manual_seed(seed)
    enc = BERTLanguageEncoder(ConfigNode({}), weights=None, hidden_size=768, num_hidden_layers=2,
                              num_attention_heads=12, type_vocab_size=2).to(DEV)
    enc.train()
    for m in enc.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return enc


def _texts(B=8, La=50, Lb=300, seed=1):
    g = torch.Generator().manual_seed(seed)
    out = []
    for L, lo in ((La, 6), (Lb, 30)):
        lens = torch.randint(lo, L + 1, (B,), generator=g)
        lens[0] = L                                            # one full row
        ids = torch.randint(1000, 30522, (B, L), generator=g)
        ids[:, 0] = 101
        mask = (torch.arange(L)[None, :] < lens[:, None]).long()
        out.append(((ids * mask).to(DEV), mask.to(DEV)))
    return out


def _run(enc, texts, varlen, probe_w):
    from sceneverse_amd.modules.language import bert as B
    B.set_varlen(varlen)
    try:
        enc.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            a, b = enc.forward_pair(texts[0][0], texts[0][1], texts[1][0], texts[1][1], cls_second=True)
        loss = (a.float() * probe_w[0]).sum() + (b[:, 0].float() * probe_w[1]).sum()
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in enc.named_parameters() if p.grad is not None}
        return a.detach().float(), b[:, 0].detach().float(), grads
    finally:
        B.set_varlen(True)


def test_varlen_bert_matches_the_padded_fast_path_and_huggingface():
    enc = _encoder()
    texts = _texts()
    g = torch.Generator().manual_seed(5)
    valid_a = texts[0][1].bool()
    probe_w = (torch.randn(8, 50, 768, generator=g).to(DEV) * valid_a[..., None], torch.randn(8, 768, generator=g).to(DEV))
    # poison the allocator's free memory: rows the variable-length kernels must never read as data
    junk = torch.full((64, 1024, 1024), float("nan"), device=DEV)
    del junk
    a_v, c_v, g_v = _run(enc, texts, True, probe_w)
    a_p, c_p, g_p = _run(enc, texts, False, probe_w)
    assert torch.isfinite(a_v).all() and torch.isfinite(c_v).all()
    assert a_v[~valid_a].abs().max().item() == 0.0                     # padded positions: exact zeros
    assert (a_v - a_p)[valid_a].abs().max().item() <= 2e-2
    assert (c_v - c_p).abs().max().item() <= 2e-2
    assert set(g_v) == set(g_p)
    for n in g_p:
        assert torch.isfinite(g_v[n]).all(), n
        if n.endswith("attention.self.key.bias"):
            # the exact gradient of a key bias is ZERO (a shift of all keys of a row leaves its softmax unchanged): what
            # both paths hold is rounding noise, to be small against the query bias of the same layer, not equal
            q = g_p[n.replace("key.bias", "query.bias")].float().norm().item()
            assert g_v[n].float().norm().item() <= 0.05 * q and g_p[n].float().norm().item() <= 0.05 * q, n
            continue
        rel = ((g_v[n].float() - g_p[n].float()).norm() / (g_p[n].float().norm() + 1e-12)).item()
        assert rel <= 2e-2, (n, rel)
    # HuggingFace's own forward (fp32) on the padded batch
    with torch.no_grad():
        hf_a = enc.model(texts[0][0], texts[0][1]).last_hidden_state
        hf_c = enc.model(texts[1][0], texts[1][1]).last_hidden_state[:, 0]
    assert (a_v - hf_a)[valid_a].abs().max().item() <= 0.06
    assert (c_v - hf_c).abs().max().item() <= 0.06


def test_cls_only_tail_of_the_last_layer_changes_nothing_observable():
    """The caption is read at [CLS] only: running the last layer's row-wise tail on [CLS rows | sentence rows] instead of
    every live row must give the same sentence states, the same [CLS] states and the same gradients (the skipped rows'
    gradients are exactly zero in the full form)."""
    from sceneverse_amd.modules.language import bert as B
    enc = _encoder(seed=4)
    texts = _texts(seed=7)
    g = torch.Generator().manual_seed(6)
    valid_a = texts[0][1].bool()
    probe_w = (torch.randn(8, 50, 768, generator=g).to(DEV) * valid_a[..., None], torch.randn(8, 768, generator=g).to(DEV))
    junk = torch.full((64, 1024, 1024), float("nan"), device=DEV)     # poison: dead rows must never be read as data
    del junk
    a_t, c_t, g_t = _run(enc, texts, True, probe_w)
    B.set_cls_tail(False)
    try:
        a_f, c_f, g_f = _run(enc, texts, True, probe_w)
    finally:
        B.set_cls_tail(True)
    assert torch.isfinite(a_t).all() and torch.isfinite(c_t).all()
    assert a_t[~valid_a].abs().max().item() == 0.0
    # identical arithmetic per row (same kernels, same operands): only the weight-gradient sums see a different row set
    assert torch.equal(a_t, a_f) and torch.equal(c_t, c_f)
    assert set(g_t) == set(g_f)
    for n in g_f:
        assert torch.isfinite(g_t[n]).all(), n
        if n.endswith("attention.self.key.bias"):
            continue                                               # exact value zero: rounding noise in both
        rel = ((g_t[n].float() - g_f[n].float()).norm() / (g_f[n].float().norm() + 1e-12)).item()
        assert rel <= 5e-3, (n, rel)


def test_varlen_bert_with_dropout_runs_and_is_finite():
    enc = _encoder(seed=2)
    for m in enc.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1
    texts = _texts(B=4, seed=3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = enc.forward_pair(texts[0][0], texts[0][1], texts[1][0], texts[1][1], cls_second=True)
    (a.float().square().mean() + b.float().square().mean()).backward()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    for n, p in enc.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("T,N,K", [(3000, 768, 768), (6000, 768, 3072), (5000, 2376, 768)],
                         ids=["128x128_tiles", "two_group_256x256", "two_group_ragged_M"])
def test_gemm_weight_gradient_ignores_rows_past_the_device_extent(T, N, K):
    """TN form with extent_dev: rows past the extent may hold NaN; dW / db must equal the sums over the live rows, for
    extents on and off a 64-row stage boundary and for every split count the heuristic picks (the small shape runs the
    128 x 128 kernel, the wide ones the two-group 256 x 256 kernel with one workgroup per CU)."""
    from sceneverse_amd import _native
    from sceneverse_amd.modules.layers import gemm as G
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(T, N, generator=g).to(torch.bfloat16).to(DEV)
    x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(DEV)
    for ext in (0, 1, 63, 64, 1000, 1984, T - 1, T):
        dy2, x2 = dy.clone(), x.clone()
        dy2[ext:] = float("nan")
        x2[ext:] = float("nan")
        rows = torch.tensor([ext], dtype=torch.int32, device=DEV)
        dw, db = G.linear_wgrad(dy2, x2, want_bias=True, rows_dev=rows)
        ref = dy[:ext].float().t() @ x[:ext].float()
        refb = dy[:ext].float().sum(0)
        assert torch.isfinite(dw).all() and torch.isfinite(db).all(), ext
        assert (dw - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), ext
        assert (db - refb).abs().max().item() <= 1e-4 * max(1.0, refb.abs().max().item()), ext


def test_varlen_attention_query_limit_equals_zeroed_cotangents():
    """q_limit: sequence b computes its first q_limit[b] query rows only.  Those rows must equal the full launch's, and
    the gradient must equal the full launch's with the cotangent of the skipped rows set to zero (dq of skipped rows 0)."""
    from sceneverse_amd.modules.layers.fused_attention import fused_varlen_self_attention
    g = torch.Generator().manual_seed(17)
    H, D = 12, 768
    lens = torch.tensor([50, 7, 300, 1, 130, 64, 17, 299], dtype=torch.int32)
    lim = torch.tensor([50, 3, 1, 1, 16, 0, 400, 33], dtype=torch.int32)
    cu = torch.zeros(9, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    T = int(cu[-1]) + 40                                        # dead tail
    packed = (torch.randn(T, 3 * D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    wy = torch.randn(T, D, generator=g).to(torch.bfloat16).to(DEV)
    keep = torch.zeros(T, dtype=torch.bool)
    for b in range(8):
        keep[int(cu[b]):int(cu[b]) + min(int(lens[b]), int(lim[b]))] = True
    keep = keep.to(DEV)
    cu_d, lim_d = cu.to(DEV), lim.to(DEV)
    p_full = packed.clone().requires_grad_(True)
    o_full = fused_varlen_self_attention(p_full, cu_d, 8, 300, H)
    o_full.backward(torch.where(keep[:, None], wy, torch.zeros_like(wy)))
    p_lim = packed.clone().requires_grad_(True)
    o_lim = fused_varlen_self_attention(p_lim, cu_d, 8, 300, H, q_limit=lim_d)
    o_lim.backward(wy * 0 + torch.where(keep[:, None], wy, torch.full_like(wy, float("nan"))).nan_to_num(0.0))
    assert torch.equal(o_lim[keep], o_full[keep])
    live = torch.arange(T, device=DEV) < int(cu[-1])
    assert torch.equal(p_lim.grad[live], p_full.grad[live])
    assert torch.all(p_lim.grad[live & ~keep][:, :D] == 0)


def _torch_plan(texts, S_full, T_full):
    """The torch formulation `_fast_forward_varlen` used before gps_varlen_plan (kept there as the fallback)."""
    dev = texts[0][0].device
    ids_all = torch.cat([ids.reshape(-1) for ids, _ in texts])
    valid = torch.cat([(m != 0).reshape(-1) for _, m in texts])
    lens = torch.cat([(m != 0).sum(dim=1) for _, m in texts]).to(torch.int32)
    pos = torch.cat([torch.arange(ids.shape[1], device=dev).repeat(ids.shape[0]) for ids, _ in texts])
    S, T = lens.numel(), ids_all.numel()
    perm = torch.argsort(valid.logical_not().to(torch.uint8), stable=True)
    cu = torch.zeros(S + 1, dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(T, device=dev)
    out = dict(lens=lens, cu=cu, n_valid=valid.sum(dtype=torch.int32).reshape(1), ids=ids_all[perm], pos=pos[perm],
               inv=inv, valid=valid)
    if 0 < S_full < S:
        out["n_live_full"] = valid[:T_full].sum(dtype=torch.int32).reshape(1)
        out["sel"] = torch.cat([cu[S_full:S].long(), torch.arange(T_full, device=dev)])
        out["rows_tail"] = out["n_live_full"] + (S - S_full)
        out["q_limit"] = torch.cat([lens[:S_full], torch.ones(S - S_full, dtype=torch.int32, device=dev)])
    return out


@pytest.mark.parametrize("mask_dtype", [torch.int64, torch.bool, torch.float32, torch.int32, torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("shapes,n_full", [(((8, 50), (8, 300)), 1), (((8, 50), (8, 300)), 0), (((3, 17),), 0),
                                           (((5, 64), (7, 65), (130, 33)), 2), (((64, 50), (64, 300)), 1)])
def test_varlen_plan_kernel_equals_the_torch_formulation(shapes, n_full, mask_dtype):
    """gps_varlen_plan (one launch) against ~30 torch launches, element for element: lengths, row offsets, the stable
    valid-first compaction (ids, positions, inverse map), the valid flags, the [CLS]-tail selection; the dispatch order
    is a permutation with non-increasing lengths (torch's unstable argsort breaks ties differently)."""
    from sceneverse_amd.modules.language import fused_embedding as FE
    g = torch.Generator().manual_seed(len(shapes) * 100 + n_full)
    texts = []
    for B, L in shapes:
        lens = torch.randint(1, L + 1, (B,), generator=g)
        lens[0] = L
        if B > 1:
            lens[1] = 1
        ids = torch.randint(1, 30522, (B, L), generator=g)
        mask = (torch.arange(L)[None, :] < lens[:, None])
        m = mask.to(mask_dtype)
        if mask_dtype == torch.float32:
            m = torch.where(mask, torch.full_like(m, 0.25), torch.full_like(m, -0.0))     # -0.0 is "not set" (== 0)
        texts.append((ids.to(DEV), m.to(DEV)))
    assert FE.varlen_plan_supported(texts)
    S_full = sum(B for B, _ in shapes[:n_full]) if 0 < n_full < len(shapes) else 0
    T_full = sum(B * L for B, L in shapes[:n_full]) if S_full else 0
    plan = FE.varlen_plan(texts, S_full)
    ref = _torch_plan(texts, S_full, T_full)
    for k in ("lens", "cu", "n_valid", "ids", "pos", "inv", "valid"):
        assert torch.equal(getattr(plan, k), ref[k]), k
    order = plan.order.long()
    assert torch.equal(torch.sort(order).values, torch.arange(order.numel(), device=DEV))
    ol = ref["lens"][order]
    assert bool((ol[1:] <= ol[:-1]).all())
    if S_full:
        for k in ("n_live_full", "sel", "rows_tail", "q_limit"):
            assert torch.equal(getattr(plan, k), ref[k]), k
    else:
        assert plan.sel is None


def test_varlen_bert_is_the_same_with_and_without_the_plan_kernel():
    """Outputs and gradients of the variable-length encoder are bit-identical whether the index plan comes from
    gps_varlen_plan or from the torch formulation (same indices -> same launches)."""
    from sceneverse_amd.modules.language import bert as B
    enc = _encoder()
    texts = _texts()
    g = torch.Generator().manual_seed(5)
    probe_w = (torch.randn(8, 50, 768, generator=g).to(DEV), torch.randn(8, 768, generator=g).to(DEV))
    assert B._PLAN_KERNEL
    a1, b1, g1 = _run(enc, texts, True, probe_w)
    B._PLAN_KERNEL = False
    try:
        a0, b0, g0 = _run(enc, texts, True, probe_w)
    finally:
        B._PLAN_KERNEL = True
    assert torch.equal(a1, a0) and torch.equal(b1, b0)
    assert g1.keys() == g0.keys()
    for k in g1:
        assert torch.equal(g1[k], g0[k]), k


@pytest.mark.parametrize("defect", ["hole", "left_padding", "empty_row", "none"])
def test_plan_kernel_flags_masks_that_are_not_prefixes(defect):
    """gps_varlen_plan's violation word (i32_out[4 S + 4]): set for a hole, left padding or an empty row anywhere in the
    batch -- EVERY batch, not only the first eager forwards -- and clear for right-padded masks."""
    from sceneverse_amd.modules.language import fused_embedding as FE
    texts = _texts(B=6, La=50, Lb=300, seed=5)
    ids, mask = texts[1]
    mask = mask.clone()
    if defect == "hole":
        mask[3, 7] = 0                      # row 3 is at least 30 tokens long
    elif defect == "left_padding":
        mask[2, 0] = 0
    elif defect == "empty_row":
        mask[4, :] = 0
    plan = FE.varlen_plan([texts[0], (ids, mask)], 0)
    assert int(plan.violation.item()) == (0 if defect == "none" else 1)


def test_a_later_batch_with_a_hole_poisons_the_output_instead_of_being_silently_wrong():
    """After the host-side warm-up checks are used up (the state of a long run, or of a captured graph), a mask with a
    hole must not produce plausible numbers: the embedding block's rows -- hence every output -- are NaN, and
    check_varlen_masks() raises."""
    enc = _encoder()
    enc._prefix_checks_left = 0
    texts = _texts(B=4, seed=9)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = enc.forward_pair(texts[0][0], texts[0][1], texts[1][0], texts[1][1], cls_second=True)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    enc.check_varlen_masks()
    bad = texts[1][1].clone()
    bad[1, 5] = 0
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = enc.forward_pair(texts[0][0], texts[0][1], texts[1][0], bad, cls_second=True)
    assert torch.isnan(a[texts[0][1].bool()]).all() and torch.isnan(b[:, 0]).all()
    with pytest.raises(RuntimeError, match="hole"):
        enc.check_varlen_masks()
