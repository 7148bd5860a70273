"""The whole GPS model on the MI355X (point ops through libgps_hip.so) against the golden outputs
of the reference's Python (tests/golden/gps_reference_cpu.pt).

Tolerances (stated per north_star "features/logits/loss within a stated fp tolerance"):
  fp32 run   : |diff| <= 2e-3 + 2e-3*|ref| on embeddings/logits, 1e-3 relative on losses --
               GPU GEMM/conv summation order vs CPU;
  bf16 run   : LayerNorm-ed embeddings (O(1)) within 0.08 absolute; og3d logits (a 768-term dot
               product of two such embeddings: |ref| up to ~450, std ~15 with the test weights)
               within 1 % of max|ref|; losses within 3 % -- bf16 autocast of every Linear (CPU
               bf16 autocast of the same model lands at 0.044 / 0.2 % of max, so these bounds
               are ~2-5x the bf16 floor); FPS/ball-query indices are fp32 and stay bit-exact."""
import os
import pytest
import torch
import torch.nn as nn

from oracle.param_fill import fill_params
from sceneverse_amd.common.config import ConfigNode
from sceneverse_amd.model.build import MODEL_REGISTRY, build_model
from sceneverse_amd.optim.loss import Loss
from sceneverse_amd.optim.loss.loss import obj_cls_loss
from util import clone_batch, gps_cfg, lang_dir

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(fx, **kw):
    model = build_model(gps_cfg(lang_dir(fx["seed"]), **kw))
    fill_params(model, fx["seed"])
    return model.to(DEV)


def test_gps_pretrain_fp32_matches_reference(golden_cpu):
    fx, g = golden_cpu, golden_cpu["gps_pretrain"]
    model = _model(fx).eval()
    loss_mod = Loss(model.cfg).to(DEV)
    out = model(clone_batch(fx["batch"], DEV))
    total, losses = loss_mod(out)
    total.backward()
    for k in ("og3d_logits", "intra_text_embed", "intra_obj_embeds", "inter_obj_embeds", "scene_embed",
              "scene_text_embed", "obj_cls_post_logits"):
        torch.testing.assert_close(out[k].detach().cpu(), g[k], rtol=2e-3, atol=2e-3,
                                   msg=lambda m, k=k: f"{k}: {m}")
    for k, v in g["losses"].items():
        assert abs(losses[k].item() - v) < 1e-3 * max(1.0, abs(v)), (k, losses[k].item(), v)
    params = dict(model.named_parameters())
    for name, ref in g["grads"].items():
        gn = params[name].grad.norm().item()
        assert abs(gn - ref["norm"]) <= 1e-2 * ref["norm"] + 1e-7, (name, gn, ref["norm"])


def test_gps_pretrain_bf16_autocast(golden_cpu):
    fx, g = golden_cpu, golden_cpu["gps_pretrain"]
    model = _model(fx).eval()
    loss_mod = Loss(model.cfg).to(DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(clone_batch(fx["batch"], DEV))
        _, losses = loss_mod(out)
    masks = fx["batch"]["obj_masks"]
    ref = g["og3d_logits"][masks]
    diff = (out["og3d_logits"].float().cpu()[masks] - ref).abs().max().item()
    assert diff < 1e-2 * ref.abs().max().item(), (diff, ref.abs().max().item())
    for k in ("intra_text_embed", "intra_obj_embeds", "inter_obj_embeds", "scene_embed"):
        got, want = out[k].float().cpu(), g[k]
        if k == "intra_obj_embeds":
            # outputs of the JOINT layers: the bf16 fast path runs them on the valid rows only and returns zeros at the
            # padded object slots, where the reference leaves the results of its masked-out rows (no head or loss reads
            # them: tests/test_gpu_joint_compact.py)
            assert float(got[~masks].abs().max() if (~masks).any() else 0.0) == 0.0
            got, want = got[masks], want[masks]
        d = (got - want).abs().max().item()
        assert d < 0.08, (k, d)
    for k, v in g["losses"].items():
        assert abs(losses[k].item() - v) < 3e-2 * max(1.0, abs(v)), (k, losses[k].item(), v)


def test_gps_pretrain_bf16_whole_step_gradients_vs_reference(golden_cpu):
    """Forward + losses + BACKWARD under bf16 autocast (fused attention, MFMA GEMMs, fused LayerNorm, masked CE)
    against the fp32 gradients of the reference's own model (golden["gps_pretrain"]["grads"]): gradient norms within
    5 % and the first 256 entries of each probed tensor within 10 % relative L2 -- bf16 rounding through 12
    transformer layers (the furthest-upstream tensor, the location embedding of the point encoder, sits at 7 - 9 %
    depending on which attention kernel family rounds where; each attention launch alone is within 2.5e-3 relative
    L2 of the fp32 formulation, tools/attn_accuracy.py); no structural error can hide inside that."""
    fx, g = golden_cpu, golden_cpu["gps_pretrain"]
    model = _model(fx).eval()                       # dropout off like the fixture; autograd on
    loss_mod = Loss(model.cfg).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(clone_batch(fx["batch"], DEV))
        total, losses = loss_mod(out)
    total.backward()
    for k, v in g["losses"].items():
        assert abs(losses[k].item() - v) < 3e-2 * max(1.0, abs(v)), (k, losses[k].item(), v)
    params = dict(model.named_parameters())
    for name, ref in g["grads"].items():
        gr = params[name].grad.float().cpu()
        assert abs(gr.norm().item() - ref["norm"]) <= 5e-2 * ref["norm"] + 1e-7, (name, gr.norm().item(), ref["norm"])
        head = gr.flatten()[:256]
        denom = max(ref["head"].norm().item(), 1e-2 * ref["norm"])
        assert (head - ref["head"]).norm().item() <= 1e-1 * denom, (name, (head - ref["head"]).norm().item(), denom)


def test_grounding_finetune_argmax_agrees(golden_cpu):
    fx, g = golden_cpu, golden_cpu["gps_ground"]
    model = _model(fx, heads="ground", use_scene_cap=False).eval()
    with torch.no_grad():
        out = model(clone_batch(fx["batch"], DEV))
    torch.testing.assert_close(out["og3d_logits"].cpu(), g["og3d_logits"], rtol=2e-3, atol=2e-3)
    assert torch.equal(out["og3d_logits"].argmax(-1).cpu(), g["pred"])      # acc@k decisions


def test_objcls_train_step_fp32(golden_cpu):
    g = golden_cpu["objcls"]
    cfg = ConfigNode({"num_gpu": 1, "solver": {"lr": 1e-3},
                      "model": {"name": "ObjCls", "model_name": "pointnet++", "language_type": "bert",
                                "open_vocab": False, "num_classes": 607, "cls_hidden": 1024}})
    model = MODEL_REGISTRY.get("ObjCls")(cfg).train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    fill_params(model, golden_cpu["seed"])
    model = model.to(DEV)
    out = model(clone_batch(g["batch"], DEV))
    loss = obj_cls_loss(out)
    loss.backward()
    torch.testing.assert_close(out["obj_logits"].detach().cpu(), g["obj_logits"], rtol=2e-3, atol=2e-3)
    assert abs(loss.item() - g["loss"]) < 1e-3
    params = dict(model.named_parameters())
    for name, ref in g["grads"].items():
        torch.testing.assert_close(params[name].grad.cpu(), ref, rtol=2e-2,
                                   atol=1e-5 + 2e-3 * ref.abs().max().item(),
                                   msg=lambda m, n=name: f"{n}: {m}")


def test_train_step_engine_runs_and_learns():
    from bench import gps_pretrain_cfg, _lang_dir
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    step = GPSTrainStep(gps_pretrain_cfg(_lang_dir()), device=DEV, ddp=False)
    for g in step.optimizer.param_groups:      # skip the 500-step warm-up for this check
        g["lr"] = g["initial_lr"] = 1e-4
    step.scheduler.base_lrs = [1e-4 for _ in step.scheduler.base_lrs]
    step.scheduler.lr_lambdas = [lambda s: 1.0 for _ in step.scheduler.lr_lambdas]
    batch = synth_batch(4, n_obj=16, seed=3, min_real=5, device=DEV)
    first = None
    for i in range(8):
        loss, _ = step.step(dict(batch))
        if i == 0:
            first = loss.item()
    last = loss.item()
    assert last == last and last < first, (first, last)


def test_hip_graph_step_matches_eager_step():
    """The captured whole-step HIP graph must be the same optimisation as the eager step: same
    losses step by step (dropout disabled so that both runs are deterministic functions of the
    weights), same learning-rate schedule, parameters still equal after several updates.
    Tolerance 2e-3 relative on the losses / 1e-3 on parameters: bf16 GEMMs may pick different
    hipBLASLt algorithms/workspaces under capture."""
    from bench import gps_pretrain_cfg, _lang_dir
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    from sceneverse_amd.modules.layers.transformers import MultiheadSelfAttention

    def make(graph):
        cfg = gps_pretrain_cfg(_lang_dir())
        cfg.solver.sched.args.warmup_steps = 4           # visible lr change within the test
        st = GPSTrainStep(cfg, device=DEV, ddp=False, graph=graph, graph_warmup=2, seed=7)
        for m in st.model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, MultiheadSelfAttention):
                m.dropout = 0.0
            if hasattr(m, "attention_probs_dropout_prob"):
                m.attention_probs_dropout_prob = 0.0
            if hasattr(m, "dropout_prob"):
                m.dropout_prob = 0.0
        return st

    batches = [synth_batch(4, n_obj=16, seed=20 + i, min_real=5, device=DEV) for i in range(5)]
    runs = {}
    for graph in (False, True, "dp"):
        st = make(graph)
        losses, lrs = [], []
        for b in batches:
            total, _ = st.step(dict(b))
            losses.append(total.item())
            lrs.append(float(st.optimizer.param_groups[0]["lr"]))
        runs[graph] = (losses, lrs, [p.detach().float().flatten()[:512].clone() for p in st.model.parameters()])
        if graph:
            assert st._graph is not None          # steps 3.. were graph replays
            assert st.graph_dp == (graph == "dp")
    le, lre, pe = runs[False]
    assert len(set(lre)) > 1                       # the schedule actually moved
    for mode in (True, "dp"):                      # one graph / split-graph data-parallel form (world 1)
        lg, lrg, pg = runs[mode]
        for a, b in zip(le, lg):
            assert abs(a - b) <= 2e-3 * abs(a), (mode, le, lg)
        assert all(abs(a - b) < 1e-12 + 1e-6 * abs(a) for a, b in zip(lre, lrg)), (mode, lre, lrg)
        worst = max((a - b).abs().max().item() for a, b in zip(pe, pg))
        assert worst < 1e-3, (mode, worst)


@pytest.mark.parametrize("post_add,obj_first", [(True, False), (False, True), (True, True)])
def test_segmented_graph_gradients_equal_single_graph_and_eager(post_add, obj_first):
    """Every parameter's gradient after the FIRST replay of the split-graph data-parallel step (forward | losses + top
    backward | bottom backward as separate HIP graphs) must equal the single graph's and the eager step's: the weights
    are identical up to that step (same seed, same batches, dropout off), so any difference is the segmentation --
    a gradient path that bypasses the stage boundary, an activation overwritten between graphs, a stale pointer.
    (VERDICT r3 / ADVICE r3: the segmented form had no gradient-parity test.)  Tolerance 5e-3 relative L2 per tensor:
    bf16 GEMMs, different accumulation orders of the weight gradients (grouped vs autograd accumulation).
    post_add: the LayerNorm post-addend fusion (the configuration that returned wrong text-encoder gradients in round 3);
    obj_first: the object encoder runs before the text encoder, so the bottom backward graph walks the text encoder
    first (round 3's engine died with a memory-aperture violation in that order, with or without the fusion)."""
    from bench import gps_pretrain_cfg, _lang_dir
    import sceneverse_amd.model.openvocab as OV
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    from sceneverse_amd.modules.layers import transformers as T
    from sceneverse_amd.modules.layers.transformers import MultiheadSelfAttention
    was = (T._FUSE_POST_ADD, OV._OBJ_FIRST)
    T.set_fuse_post_add(post_add)
    OV._OBJ_FIRST = obj_first
    try:
        _segmented_gradient_parity(gps_pretrain_cfg, _lang_dir, synth_batch, GPSTrainStep, MultiheadSelfAttention)
    finally:
        T.set_fuse_post_add(was[0])
        OV._OBJ_FIRST = was[1]


def _segmented_gradient_parity(gps_pretrain_cfg, _lang_dir, synth_batch, GPSTrainStep, MultiheadSelfAttention):

    batches = [synth_batch(4, n_obj=16, seed=20 + i, min_real=5, device=DEV) for i in range(3)]
    grads = {}
    for graph in (False, True, "dp"):
        cfg = gps_pretrain_cfg(_lang_dir())
        st = GPSTrainStep(cfg, device=DEV, ddp=False, graph=graph, graph_warmup=2, seed=7)
        for m in st.model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, MultiheadSelfAttention):
                m.dropout = 0.0
            if hasattr(m, "attention_probs_dropout_prob"):
                m.attention_probs_dropout_prob = 0.0
        for b in batches:                     # steps 0-1 eager warm-up, step 2 = capture + first replay
            st.step(dict(b))
        torch.cuda.synchronize()
        assert (st._graph is not None) == bool(graph)
        if graph == "dp":
            assert st._graph[2] is not None, "the backward pass was not segmented"
        grads[graph] = {n: p.grad.detach().float().clone() for n, p in st.model.named_parameters() if p.grad is not None}
    ref = grads[False]
    assert len(ref) > 150
    for mode in (True, "dp"):
        assert grads[mode].keys() == ref.keys(), (mode, set(grads[mode]) ^ set(ref))
        for n, g in grads[mode].items():
            assert torch.isfinite(g).all(), (mode, n)
            rel = ((g - ref[n]).norm() / (ref[n].norm() + 1e-20)).item()
            assert rel <= 5e-3, (mode, n, rel, g.norm().item(), ref[n].norm().item())


def test_graph_replays_draw_fresh_dropout_masks():
    """With dropout on and the learning rate at zero the weights never move, so any change of the loss
    between replays of the SAME batch is the dropout mask changing: every replay must differ from the
    previous one (device-side seed block advanced by a kernel inside the captured graph), for the whole
    graph and for the split-graph form."""
    from bench import gps_pretrain_cfg, _lang_dir
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    batch = synth_batch(4, n_obj=16, seed=31, min_real=5, device=DEV)
    for graph in (True, "dp"):
        cfg = gps_pretrain_cfg(_lang_dir())
        st = GPSTrainStep(cfg, device=DEV, ddp=False, graph=graph, graph_warmup=2, seed=3)
        st.scheduler.base_lrs = [0.0] * len(st.scheduler.base_lrs)      # lr = base_lr * lambda(step) = 0
        for g in st.optimizer.param_groups:
            g["lr"].zero_()
        w0 = [p.detach().clone() for p in list(st.model.parameters())[:8]]
        losses = [st.step(dict(batch))[0].item() for _ in range(8)]
        assert st._graph is not None
        replays = losses[3:]                       # steps 0-1 eager warm-up, 2 capture, 3.. replays
        assert len(set(replays)) == len(replays), (graph, losses)
        assert all(torch.equal(a, b) for a, b in zip(w0, list(st.model.parameters())[:8]))


def test_fast_bert_path_matches_huggingface_layers():
    """BERTLanguageEncoder's fused GPU path (packed QKV, fused attention / SDPA, fused residual+LN)
    against HuggingFace's own BertModel.forward on the same weights, bf16 autocast, dropout off:
    outputs within 0.06 absolute (LayerNorm-ed, O(1)); gradients of a probe loss within 5 % of the
    largest entry.  Both a fused-attention length (50) and an SDPA length (300) are covered."""
    from sceneverse_amd.modules.language import bert as B
    enc = B.BERTLanguageEncoder(None, weights=None, hidden_size=768, num_hidden_layers=2,
                                num_attention_heads=12, type_vocab_size=2).to(DEV).eval()
    torch.manual_seed(0)
    for L in (50, 300):
        ids = torch.randint(1000, 30000, (4, L), device=DEV)
        masks = (torch.arange(L, device=DEV)[None, :] < torch.tensor([L, L // 2, 7, L - 3], device=DEV)[:, None]).long()
        res = {}
        for fast in (True, False):
            B.set_fast_bert(fast)
            enc.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = enc(ids, masks)
            keep = masks.bool()
            (out.float()[keep] ** 2).mean().backward()
            res[fast] = (out.float()[keep].detach(),
                         enc.model.encoder.layer[0].attention.self.query.weight.grad.float().clone(),
                         enc.model.encoder.layer[1].output.dense.weight.grad.float().clone())
        B.set_fast_bert(True)
        assert (res[True][0] - res[False][0]).abs().max().item() < 0.06, L
        for a, b in zip(res[True][1:], res[False][1:]):
            assert (a - b).abs().max().item() <= 0.05 * b.abs().max().item() + 1e-8, L


def test_pairwise_locs_kernel_matches_the_torch_formulation(golden_cpu):
    """gps_pairwise_locs vs modules/utils.calc_pairwise_locs' torch ops on the GPU and on the CPU
    (the formulation pinned bit-exactly to the reference in tests/test_oracle_vs_golden.py): every
    feature is a ratio in [-1, 1]; |diff| <= 1e-6 (a few ulp: torch's own CPU and GPU results differ
    by as much -- different sqrt/divide rounding and 3-term sum order).
    Edge cases: coincident centres (d = sqrt(eps)), a single object, padding slots at the origin."""
    from sceneverse_amd.modules import utils as U
    g = torch.Generator().manual_seed(0)
    for B, L in ((64, 80), (3, 1), (2, 130), (1, 257)):
        c = torch.rand(B, L, 3, generator=g) * 8 - 4
        c[:, L // 2:] = 0.0                                   # padding objects: all at the origin
        cg = c.to(DEV)
        got = U.calc_pairwise_locs(cg, None)
        delta = cg.unsqueeze(2) - cg.unsqueeze(1)
        dist = torch.sqrt(torch.sum(delta ** 2, 3) + 1e-10)
        nd = dist / torch.max(dist.view(B, -1), dim=1)[0].view(-1, 1, 1)
        dxy = torch.sqrt(torch.sum(delta[..., :2] ** 2, 3) + 1e-10)
        want = torch.stack([nd, delta[..., 2] / dist, dxy / dist, delta[..., 1] / dxy, delta[..., 0] / dxy], 3)
        assert (got - want).abs().max().item() <= 1e-6, (B, L, (got - want).abs().max().item())
        cpu = U.calc_pairwise_locs(c, None)
        assert (got.cpu() - cpu).abs().max().item() <= 1e-6, (B, L)
    fx = golden_cpu
    locs = fx["batch"]["obj_locs"]
    got = U.calc_pairwise_locs(locs[:, :, :3].to(DEV), locs[:, :, 3:].to(DEV)).cpu()
    want = U.calc_pairwise_locs(locs[:, :, :3], locs[:, :, 3:])
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


def test_ddp_wrapper_runs_on_the_fused_path_with_rccl():
    """torch DDP (backend nccl = RCCL) around the GPS model on the GPU, world_size 1: exercises what
    bench.py runs at N > 1 -- the probe step that freezes the never-used tensors, bucket views, bf16
    gradient compression over RCCL, the fused autograd functions (MFMA GEMMs, fused attention / LN / CE,
    the fused AdamW pass reading bucket views), the RCCL all-gather of the between-batch loss -- minus
    the second rank.  Loss must stay finite and go down."""
    import socket
    import torch.distributed as dist
    from bench import gps_pretrain_cfg, _lang_dir
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        cfg = gps_pretrain_cfg(_lang_dir(), num_gpu=2)          # distributed loss branch on
        cfg.solver.sched.args.warmup_steps = 1
        st = GPSTrainStep(cfg, device=DEV, ddp=True, graph=False)
        from torch.nn.parallel import DistributedDataParallel as DDP
        batch = synth_batch(4, n_obj=16, seed=3, min_real=5, device=DEV)
        losses = [st.step(dict(batch))[0].item() for _ in range(6)]
        assert isinstance(st.net, DDP)                           # built by the first step, after the probe
        assert all(l == l for l in losses) and losses[-1] < losses[0], losses
        assert len(st.frozen_unused) >= 13                       # found by the probe, frozen before the wrap
        assert sum(1 for p in st.model.parameters() if p.requires_grad and p.grad is None) == 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("extra", [[], ["--graph-dp"]], ids=["ddp", "segmented_graphs"])
def test_bench_n2_code_path_on_one_gpu(extra):
    """bench.py launched exactly as the driver launches it for N = 2 (torch.distributed.run, one process
    per rank), except that both ranks share cuda:0 and talk over gloo (GPS_BENCH_SHARE_GPU=1; RCCL refuses
    two ranks on one GPU).  Checks the N > 1 flow end to end: DDP wrap, between-batch all-gather,
    max-over-ranks timing, one JSON line from rank 0 with the aggregate over both ranks.  `--graph-dp`: the same
    with the step as forward / top-backward / bottom-backward / optimizer graphs around two eager all-reduces."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GPS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8"] + extra
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 16 * 2 / (out["ms_per_step"] * 2e-3)) < 0.02 * out["value"]
    assert "cpu_baseline" not in out and out["roofline"] is not None
    assert out["config"]["final_loss"] == out["config"]["final_loss"]
    # the start-up self-check of the split-graph step against eager torch DDP ran and passed; the exposed part of the
    # gradient exchange is reported
    chk = out["config"]["dp_self_check"]
    assert chk["ok"] and chk["replayed_a_graph"] and chk["cross_rank_spread"] == 0.0, chk
    assert out["config"]["exposed_allreduce_ms"] >= 0.0


def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` with NO launcher environment: the script re-executes itself under
    torch.distributed.run with two ranks (sharing cuda:0 over gloo here) and reports n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GPS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-extras"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16


def _bench_slice(n_scenes=2):
    """The first scenes of the EXACT bench batch: synth_batch(64, seed=42) of bench.py (rank 0), 80 objects x 1024
    points, 50-token sentence + 300-token caption."""
    from sceneverse_amd.data.synthetic import synth_batch
    full = synth_batch(64, n_obj=80, n_pts=1024, txt_len=50, seed=42)
    return {k: (v[:n_scenes].clone() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 64 else v) for k, v in full.items()}


def _bf16_step_bounds():
    """Per-quantity bounds of the bf16-vs-fp32 step comparison: 1.5 x what was MEASURED on an MI355X (the committed
    tests/golden/bf16_step_measured.json; re-measure with `pytest -s -k fp32_oracle_port` and the [bf16-vs-fp32] lines),
    with a floor for quantities whose error is at the noise level, and never looser than 1.5 x the blanket bounds they
    replace (3 % losses, 5 % gradient norms, 6 % relative L2 of a gradient tensor)."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "bf16_step_measured.json")) as f:
        measured = json.load(f)["measured"]
    blanket = lambda k: 3e-2 if k.startswith("loss") else (5e-2 if k.startswith("norm") else 6e-2)  # noqa: E731
    # floors: the gradient-norm deviations are second-order in the bf16 noise and move by +-1e-2 between kernel families
    # that agree to bf16 rounding (profiles/r5/step_bounds_ab.txt: 0.0006 .. 0.0074 for the same tensor)
    floor = lambda k: 1e-3 if k.startswith("loss") else (1.5e-2 if k.startswith("norm") else 1e-2)  # noqa: E731
    return {k: min(1.5 * blanket(k), max(1.5 * v, floor(k))) for k, v in measured.items()}, blanket


@pytest.mark.timeout(900)
def test_bench_config_step_against_the_fp32_oracle_port():
    """End-to-end parity AT THE BENCH CONFIGURATION: two scenes of the bench batch through
      (a) the fp32 oracle port of the whole step on the CPU (oracle/gps_torch_reference.py + C point ops, the
          formulation tests/test_oracle_vs_golden.py pins to the reference's own Python) and
      (b) the product step on the GPU: bf16 autocast, fused kernels, eager AND replayed as one HIP graph,
    same weights, dropout off.  Compared: every loss (3 %: bf16 logits), and for a probe set of parameters from the
    bottom, middle and top of the model the gradient norm (5 %) and the relative L2 of the whole gradient tensor
    (<= 6 %: measured 1 - 4 %; per-launch bf16 rounding is 2.5e-3 and the probes sit behind up to 12 layers).
    The graph replay must reproduce the eager losses to 2e-3 relative."""
    import copy
    from bench import gps_pretrain_cfg, _lang_dir
    from oracle import gps_torch_reference as R
    from sceneverse_amd.engine import GPSTrainStep
    from sceneverse_amd.modules.layers.transformers import MultiheadSelfAttention

    def no_dropout(model):
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, MultiheadSelfAttention):
                m.dropout = 0.0
        cfgb = model.lang_encoder.model.config
        cfgb.hidden_dropout_prob = cfgb.attention_probs_dropout_prob = 0.0

    # A/B runs of the attention kernel families (tools/gpu_r5_ab.sh): GPS_TEST_SPATIAL_PLANES=0 / GPS_TEST_PLAIN_MODE=<mask>
    from sceneverse_amd.modules.layers import fused_attention as FA
    if os.environ.get("GPS_TEST_SPATIAL_PLANES"):
        FA.set_spatial_planes(os.environ["GPS_TEST_SPATIAL_PLANES"] != "0")
    if os.environ.get("GPS_TEST_PLAIN_MODE"):
        FA.set_plain_mode(int(os.environ["GPS_TEST_PLAIN_MODE"]))
    lp = _lang_dir()
    data = _bench_slice(2)
    eager = GPSTrainStep(gps_pretrain_cfg(lp), device=DEV, ddp=False, graph=False, seed=11)
    no_dropout(eager.model)
    init = copy.deepcopy(eager.model.state_dict())
    loss_sd = copy.deepcopy(eager.loss.state_dict())

    # ---- (a) the oracle port, fp32, CPU ----
    sd = {}
    for k, v in init.items():
        if k.startswith("lang_encoder."):
            continue
        t = v.detach().cpu().clone()
        t = t.float() if torch.is_floating_point(t) else t
        frozen = k.startswith("point_encoder.point_feature_extractor") or k.endswith("text_features") \
            or "running_" in k or not torch.is_floating_point(t)
        sd[k] = t if frozen else t.requires_grad_(True)
    bert = copy.deepcopy(eager.model.lang_encoder.model).cpu().float().eval()
    lang = lambda ids, masks: bert(ids, masks).last_hidden_state  # noqa: E731
    scale_key = [k for k in loss_sd if k.endswith("TextSceneBetweenBatch.logit_scale")]
    logit_scale = loss_sd[scale_key[0]].detach().cpu().float() if scale_key else torch.tensor(1 / 0.07)
    cpu = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
    ref_out = R.openvocab_forward(sd, cpu, lang)
    ref_losses = R.pretrain_losses(ref_out, cpu, logit_scale)
    ref_losses["total_loss"].backward()
    probes = ["point_encoder.loc_layers.0.0.weight", "point_encoder.spatial_encoder.0.self_attn.w_qs.weight",
              "point_encoder.spatial_encoder.0.self_attn.lang_cond_fc.weight", "point_encoder.spatial_encoder.3.linear2.weight",
              "unified_encoder.unified_encoder.0.self_attn.in_proj_weight", "unified_encoder.unified_encoder.3.linear1.weight",
              "unified_encoder.unified_encoder.3.norm2.weight"]
    assert all(p in sd and sd[p].grad is not None for p in probes), [p for p in probes if p not in sd or sd[p].grad is None]
    ref_grads = {p: sd[p].grad.clone() for p in probes}
    bert_probe = "encoder.layer.0.attention.self.query.weight"
    ref_bert_grad = dict(bert.named_parameters())[bert_probe].grad.clone()

    # ---- (b) the product step, bf16, eager: forward + losses + backward only (gradients before the update) ----
    dev_batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in data.items()}
    eager.net.train()
    out, total, losses = eager.forward_loss(dict(dev_batch, cur_step=0, total_steps=1 << 30))
    eager.optimizer.zero_grad(set_to_none=True)
    total.backward()
    measured = {}
    for k in ("lm_cls_loss", "TextObjWithinBatch", "TextSceneBetweenBatch", "total_loss"):
        a, b = float(losses[k]), float(ref_losses[k])
        measured["loss " + k] = abs(a - b) / max(1.0, abs(b))
    params = dict(eager.model.named_parameters())
    for p in probes:
        g, r = params[p].grad.float().cpu(), ref_grads[p]
        measured["norm " + p] = abs(g.norm().item() - r.norm().item()) / (r.norm().item() + 1e-20)
        measured["rel " + p] = ((g - r).norm() / (r.norm() + 1e-20)).item()
    g = params["lang_encoder.model." + bert_probe].grad.float().cpu()
    measured["rel bert " + bert_probe] = ((g - ref_bert_grad).norm() / (ref_bert_grad.norm() + 1e-20)).item()
    for k, v in measured.items():
        print(f"[bf16-vs-fp32] {k}: {v:.4f}")
    # bounds PER QUANTITY (measured x 1.5, tests/golden/bf16_step_measured.json); a quantity without a measurement is a
    # test bug, not a reason for a loose bound
    bounds, _ = _bf16_step_bounds()
    assert set(measured) <= set(bounds), sorted(set(measured) - set(bounds))
    bound = bounds.__getitem__
    bad = {k: (v, bound(k)) for k, v in measured.items() if v > bound(k)}
    assert not bad, bad

    # ---- the replayed HIP graph computes the same step as the eager one ----
    losses_e = []
    eager2 = GPSTrainStep(gps_pretrain_cfg(lp), device=DEV, ddp=False, graph=False, seed=11)
    no_dropout(eager2.model)
    eager2.model.load_state_dict(init)
    graph = GPSTrainStep(gps_pretrain_cfg(lp), device=DEV, ddp=False, graph=True, graph_warmup=2, seed=11)
    no_dropout(graph.model)
    graph.model.load_state_dict(init)
    from sceneverse_amd.modules.layers import gemm as _gemm
    _gemm.invalidate_shadows()
    for _ in range(4):
        le, _ = eager2.step(dict(dev_batch))
        lg, _ = graph.step(dict(dev_batch))
        losses_e.append((float(le), float(lg)))
    for i, (a, b) in enumerate(losses_e):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (i, a, b)
    assert abs(losses_e[0][0] - float(ref_losses["total_loss"])) <= 3e-2 * abs(float(ref_losses["total_loss"]))
