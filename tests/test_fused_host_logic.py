"""Host-side logic of the fused paths that can be checked without a GPU:
  * BN folding of a SharedMLP (what gps_sa_mlp_pack_layer is fed) reproduces conv+BN(eval)+ReLU;
  * the frozen/unfrozen decision and the cache key of the folded weights;
  * CPU tensors never reach the fused GPU-only wrappers (torch formulation is used);
  * the between-batch losses' gather protocol used by the split-graph data-parallel engine."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from sceneverse_amd.common.config import ConfigNode
from sceneverse_amd.modules.layers import transformers as T
from sceneverse_amd.modules.layers.fused_norm import add_dropout_layer_norm, supported as ln_supported
from sceneverse_amd.optim.loss.contra_loss import TextSceneBetweenBatch
from sceneverse_amd.pointnet2 import pointnet2_modules as M
from sceneverse_amd.pointnet2 import pytorch_utils as pt_utils


def _mlp(spec, seed=0):
    torch.manual_seed(seed)
    mlp = pt_utils.SharedMLP(list(spec), bn=True)
    for m in mlp.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.normal_(1, 0.1)
            m.bias.data.normal_(0, 0.1)
    return mlp


def test_fold_shared_mlp_equals_conv_bn_relu_in_eval_mode():
    mlp = _mlp([131, 128, 128, 256]).eval()
    x = torch.randn(3, 131, 16, 32)
    ws, shifts = M.fold_shared_mlp(mlp)
    assert [tuple(w.shape) for w in ws] == [(128, 131), (128, 128), (256, 128)]
    y = x
    for w, s in zip(ws, shifts):
        y = torch.relu(torch.einsum('oc,bcps->bops', w, y) + s.view(1, -1, 1, 1))
    torch.testing.assert_close(y, mlp(x), rtol=1e-5, atol=1e-5)


def test_fold_refuses_training_mode_and_frozen_detection():
    mlp = _mlp([6, 64, 64, 128]).train()
    assert M.fold_shared_mlp(mlp) is None                  # running statistics are not what BN uses
    assert not M._is_frozen(mlp)
    mlp.eval()
    assert not M._is_frozen(mlp)                           # parameters still want gradients
    with torch.no_grad():
        assert M._is_frozen(mlp)
    for p in mlp.parameters():
        p.requires_grad_(False)
    assert M._is_frozen(mlp)
    k0 = M._frozen_key(mlp)
    with torch.no_grad():
        mlp.layer0.bn.bn.running_mean.add_(1.0)
    assert M._frozen_key(mlp) != k0                        # in-place update -> repack


def test_cpu_tensors_take_the_torch_formulation():
    x, h = torch.randn(5, 768), torch.randn(5, 768)
    norm = nn.LayerNorm(768)
    assert not ln_supported(x, h, norm)
    torch.testing.assert_close(add_dropout_layer_norm(x, h, norm, 0.1, False), norm(x + h))
    y, y2 = add_dropout_layer_norm(x, h, norm, 0.0, False, want_bf16=True)
    assert y is y2
    layer = T.TransformerEncoderLayer(768, 12, dropout=0.0).eval()
    assert not T._use_hip(torch.randn(2, 10, 768), 768, 12)
    out, attn = layer(torch.randn(2, 10, 768))
    assert out.shape == (2, 10, 768) and attn is None


def test_between_batch_loss_gather_protocol():
    cfg = ConfigNode({"num_gpu": 2})
    crit = TextSceneBetweenBatch(cfg)
    d = {"scene_embed": torch.randn(4, 768), "scene_text_embed": torch.randn(4, 768)}
    sc, tx = crit.gather_inputs(d)
    torch.testing.assert_close(sc.norm(dim=-1), torch.ones(4))
    # engine-provided gathered buffers (2 ranks x 4) replace the collective
    crit._gathered = [torch.cat([sc, sc.flip(0)]).detach(), torch.cat([tx, tx.flip(0)]).detach()]
    loss = crit(d)
    assert loss.shape == () and torch.isfinite(loss)
    loss.backward()
    assert crit.logit_scale.grad is not None


def test_seed_stream_never_repeats_and_keeps_saved_views_stable():
    """The device-side dropout seed stream (fused_attention._SeedStream): words handed out are views of a
    block; begin_step advances the block in place (between steps), exhaustion builds a new block (so views
    saved for a pending backward keep their value); no word is ever handed out twice."""
    import torch
    from sceneverse_amd.modules.layers import fused_attention as FA
    torch.manual_seed(5)
    st = FA._SeedStream(torch.device("cpu"))
    seen = set()
    for step in range(3):
        st.begin_step()
        views, vals = [], []
        for _ in range(FA._SEED_BLOCK + 40):            # crosses one exhaustion inside the step
            w = st.next()
            assert w.shape == (1,) and w.dtype == torch.int64
            views.append(w)
            vals.append(int(w.item()))
        assert [int(v.item()) for v in views] == vals    # nothing handed out was modified afterwards
        assert not (seen & set(vals)) and len(set(vals)) == len(vals)
        seen |= set(vals)
    # consecutive words differ by the stream increment modulo 2^64
    a, b = st.next(), st.next()
    assert (int(b.item()) - int(a.item())) % (1 << 64) == FA._SEED_INC


def test_gemm_host_logic_on_cpu():
    """Host logic of modules/layers/gemm.py without a GPU: the native GEMMs are never chosen for CPU tensors
    (layers keep the torch formulation), the bf16 shadow of a weight group is rebuilt exactly when a master
    changed, packed groups concatenate rows and biases in order, split counts are sane."""
    import torch
    from sceneverse_amd import _native
    from sceneverse_amd.modules.layers import gemm as G
    x = torch.randn(4, 16)
    assert not G.usable(x, 16, 16) and not G.usable(x.to(torch.bfloat16), 16, 16)
    lins = [torch.nn.Linear(16, n) for n in (16, 8, 24)]
    w16, b32 = G.shadow_of([m.weight for m in lins], [m.bias for m in lins])
    assert w16.shape == (48, 16) and w16.dtype == torch.bfloat16 and b32.shape == (48,)
    assert torch.equal(w16, torch.cat([m.weight for m in lins]).to(torch.bfloat16))
    assert torch.equal(b32, torch.cat([m.bias for m in lins]))
    again, _ = G.shadow_of([m.weight for m in lins], [m.bias for m in lins])
    assert again.data_ptr() == w16.data_ptr() and again._version == w16._version       # no rebuild
    with torch.no_grad():
        lins[1].weight.add_(1.0)
    w16b, _ = G.shadow_of([m.weight for m in lins], [m.bias for m in lins])
    assert w16b.data_ptr() == w16.data_ptr()                                            # refreshed in place
    assert torch.equal(w16b[16:24], lins[1].weight.to(torch.bfloat16))
    G.invalidate_shadows()
    assert G.shadow_entry([m.weight for m in lins]).versions is None
    single_w, single_b = G.shadow_of([lins[0].weight], [lins[0].bias])
    assert single_b.data_ptr() == lins[0].bias.data_ptr()                              # one bias: no copy
    G.clear_shadows()
    # a dead module's id() may be recycled by the next one: the cache must notice (weak references to the masters)
    import gc
    for _ in range(30):
        lin = torch.nn.Linear(16, 16)
        w16, _b = G.shadow_of([lin.weight], [lin.bias])
        assert torch.equal(w16, lin.weight.to(torch.bfloat16))
        del lin, w16, _b
        gc.collect()
    G.clear_shadows()
    lib = _native.load()
    for tokens in (77, 3200, 5120, 8320, 19200):
        for n_out, n_in in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (2376, 768)):
            s = lib.gps_gemm_pick_splits(_native.GEMM_TN, n_out, n_in, tokens)
            assert 1 <= s <= 32 and (s == 1 or (tokens + 63) // 64 // s >= 8)
            assert lib.gps_gemm_workspace_floats(_native.GEMM_TN, n_out, n_in, s) == (0 if s == 1 else s * (n_out * n_in + n_out))
    assert lib.gps_gemm_pick_splits(_native.GEMM_NT, 5120, 768, 768) == 1
    import torch.nn.functional as F
    assert G.activation_name(F.gelu) == "gelu" and G.activation_name(F.relu) == "relu" and G.activation_name(F.glu) is None


def test_packed_scans_host_side_on_cpu():
    """PackedScans / batch_rows (sceneverse_amd/data/gpu_objects.py) without a GPU: instances regrouped in
    the loader's order with their own point order kept, empty instances skipped like `np.sum(mask) == 0`,
    16-byte records carry the colours, the row table pads with -1 and honours obj_select."""
    import numpy as np
    import pytest
    import torch
    from sceneverse_amd.data import gpu_objects as G
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(50, 3)).astype(np.float32)
    col = rng.integers(0, 256, size=(50, 3), dtype=np.uint8)
    lab = rng.integers(0, 4, size=50)
    for records in (True, False):
        p = G.PackedScans("cpu", records=records)
        p.add_scan("a", pts, col, lab, [2, 0, 7, 3])            # instance 7 has no points
        p.add_scan("b", pts[:10], col[:10], lab[:10] * 0 + 5, [5])
        p.finalize()
        assert p.scan_inst_ids("a") == [2, 0, 3] and list(p.scan_objects("a")) == [0, 1, 2]
        assert list(p.scan_objects("b")) == [3] and p.n_objects == 4
        off = p.obj_offsets.numpy()
        for o, inst in enumerate([2, 0, 3]):
            sel = np.flatnonzero(lab == inst)
            got = p.xyz.numpy()[off[o]:off[o + 1]]
            assert np.array_equal(got[:, :3], pts[sel])
            if records:
                assert p.rgb is None and got.shape[1] == 4
                assert np.array_equal(np.ascontiguousarray(got).view(np.uint8).reshape(-1, 16)[:, 12:15], col[sel])
            else:
                assert np.array_equal(p.rgb.numpy()[off[o]:off[o + 1]], col[sel])
        rows = G.batch_rows(p, ["b", "a"], 4)
        assert rows.dtype == torch.int32 and rows.tolist() == [[3, -1, -1, -1], [0, 1, 2, -1]]
        assert G.batch_rows(p, ["a"], 2, obj_select=[[2, 0]]).tolist() == [[2, 0]]
        with pytest.raises(ValueError):
            G.batch_rows(p, ["a"], 2)
        with pytest.raises(RuntimeError):                       # product path is GPU-only: no CPU fallback
            G.obj_processing_post(p, rows, 64)


def test_bench_rank_resolution_rules():
    """bench.py --gpus N: no launcher environment -> self-spawn for N > 1; a launcher whose WORLD_SIZE disagrees
    with --gpus is an error (never a silent one-GPU run)."""
    import sys
    import pytest
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    assert bench.resolve_world(1, {}) == ("run", 1)
    assert bench.resolve_world(8, {}) == ("spawn", None)
    assert bench.resolve_world(8, {"RANK": "3", "WORLD_SIZE": "8"}) == ("run", 8)
    for gpus, world in ((8, 2), (1, 2), (2, 1)):
        with pytest.raises(SystemExit):
            bench.resolve_world(gpus, {"RANK": "0", "WORLD_SIZE": str(world)})
    cmd = bench.spawn_command(["--gpus", "4", "--steps", "3"], 4, 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_graph_step_rejects_batches_that_do_not_match_the_capture():
    """GPSTrainStep._fill_static: same keys, shapes and dtypes as the captured batch, or a clear error."""
    import types

    from sceneverse_amd.engine import GPSTrainStep
    eng = types.SimpleNamespace(_static={"a": torch.zeros(4, 3), "b": torch.zeros(4, dtype=torch.int64)})
    fill = lambda t: GPSTrainStep._fill_static(eng, t)  # noqa: E731
    fill({"a": torch.ones(4, 3), "b": torch.arange(4)})
    assert torch.equal(eng._static["a"], torch.ones(4, 3)) and torch.equal(eng._static["b"], torch.arange(4))
    with pytest.raises(ValueError, match="keys"):
        fill({"a": torch.ones(4, 3)})
    with pytest.raises(ValueError, match="captured"):
        fill({"a": torch.ones(2, 3), "b": torch.arange(4)})                # a smaller last batch
    with pytest.raises(ValueError, match="captured"):
        fill({"a": torch.ones(4, 3), "b": torch.arange(4).float()})


def test_weight_gradient_split_rule_fills_one_resident_round():
    """gps_gemm_pick_splits.  128 x 128 tiles (two workgroups per CU): largest split count with tiles * splits <= 512
    resident workgroups and >= 8 stages per split (profiles/r2/split_sweep.json).  Wide weight gradients over a long
    token reduction (>= 18 tiles of 256 x 256, >= 24 stages per split): the two-group 256 x 256 kernel, one workgroup per
    CU, 160 .. 256 workgroups (profiles/r3/gemm_8p_shapes.log)."""
    from sceneverse_amd import _native
    lib = _native.load()
    pick = lambda M, N, K: lib.gps_gemm_pick_splits(_native.GEMM_TN, M, N, K)  # noqa: E731
    assert pick(768, 768, 19200) == 14 and pick(768, 2048, 8320) == 5 and pick(768, 768, 3200) == 6
    assert pick(30528, 768, 3200) == 1
    for M, N, K in ((768, 768, 19200), (2376, 768, 5120), (768, 2304, 8320)):          # the 128 x 128 rule
        s = pick(M, N, K)
        tiles = -(-M // 128) * -(-N // 128)
        assert tiles * s <= 512 and (K + 63) // 64 // s >= 8
    assert pick(768, 2304, 19200) == 9 and pick(768, 3072, 19200) == 7 and pick(3072, 768, 22400) == 7
    for M, N, K in ((768, 2304, 19200), (768, 3072, 22400), (2048, 768, 19200)):       # the 256 x 256 rule
        s = pick(M, N, K)
        tiles = -(-M // 256) * -(-N // 256)
        assert 160 <= tiles * s <= 256 and (K + 63) // 64 // s >= 24


def test_round3_fused_helpers_fall_back_to_torch_on_cpu():
    """l2_normalize / add_row / column_sum / loc_embed / post-addend LayerNorm: CPU tensors take the torch formulation (the
    native kernels are GPU-only and the product path never routes GPU work through these fallbacks)."""
    import torch.nn.functional as F
    from torch import nn
    from sceneverse_amd.modules.layers.fused_loc import loc_embed, supported as loc_supported
    from sceneverse_amd.modules.layers.fused_norm import add_dropout_layer_norm, add_row, column_sum, l2_normalize
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 768, generator=g)
    assert torch.equal(l2_normalize(x), F.normalize(x, dim=-1, p=2))
    row = torch.randn(768, generator=g, requires_grad=True)
    y = add_row(x, row)
    assert torch.equal(y, x + row)
    y.sum().backward()
    assert torch.allclose(row.grad, torch.full((768,), 15.0))
    assert torch.allclose(column_sum(x), x.reshape(-1, 768).sum(0))
    seq = nn.Sequential(nn.Linear(6, 768), nn.LayerNorm(768))
    locs = torch.randn(2, 4, 6, generator=g)
    assert not loc_supported(seq, locs)
    assert torch.equal(loc_embed(seq, locs), seq(locs))
    norm = nn.LayerNorm(768)
    h, post = torch.randn(3, 5, 768, generator=g), torch.randn(3, 5, 768, generator=g)
    assert torch.equal(add_dropout_layer_norm(x, h, norm, post=post), norm(x + h) + post)
    with pytest.raises(ValueError):
        add_dropout_layer_norm(x, h, norm, post=post[:, :2])


def test_encoder_layers_accept_a_post_addend():
    """`post_add` of the encoder layers = the addend the NEXT layer would apply to its input: same values as adding it
    outside (the object / unified encoders pass their re-added embeddings this way)."""
    from sceneverse_amd.modules.layers.transformers import TransformerEncoderLayer, TransformerSpatialEncoderLayer
    torch.manual_seed(0)
    x, e = torch.randn(2, 7, 64), torch.randn(2, 7, 64)
    layer = TransformerEncoderLayer(64, 4, dim_feedforward=128, dropout=0.0).eval()
    assert torch.allclose(layer(x, post_add=e)[0], layer(x)[0] + e)
    sp = TransformerSpatialEncoderLayer(64, 4, dim_feedforward=128, dropout=0.0, spatial_multihead=True, spatial_dim=5,
                                        spatial_attn_fusion='cond').eval()
    pl = torch.randn(2, 7, 7, 5)
    assert torch.allclose(sp(x, pl, post_add=e)[0], sp(x, pl)[0] + e)
