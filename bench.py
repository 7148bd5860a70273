"""bench.py -- GPS pre-train pairs/s (forward + loss + backward + AdamW) on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: the `all_pretrain.yaml` model (BERT-4L text
encoder, PointNet++ object encoder on libgps_hip.so, 4 spatial + 4 joint transformer layers,
OVPretrainHead) with losses lm_cls_loss + TextObjWithinBatch + TextSceneBetweenBatch, B = 64
scenes per GPU, 80 objects x 1024 points x 6 ch, 50-token sentence + 300-token scene caption
(BASELINE.json configs[1]; SURVEY.md section 8(d) "config 2").  Synthetic inputs, random-init
weights; inputs are resident in HBM before the timed region.  Weak scaling: per-GPU batch fixed.

`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N
ranks (reference behaviour: common/launch_utils.py:26-42); a WORLD_SIZE that disagrees with --gpus is an error.

The single JSON line (< 4 KB, last line of stdout) also carries
  roofline      the kernel family with the largest time per step among ALL kernels of the step (native launches by
                HIP events, everything else by a torch.profiler pass): ALGORITHMIC flops (or bytes) / measured
                duration vs the dense MFMA peak of its dtype (or 8 TB/s), + PMC HBM traffic of its largest shape
  headline      the north-star fractions: unfused ball_query+group vs the HBM roof (the six launches of the
                reference API timed here), all native GEMMs and the attention core vs the bf16 MFMA peak
  cpu_baseline  the oracle port (oracle/gps_torch_reference.py + C point ops + HF BERT, fp32, min(64, host cores)
                threads) timed on a bounded sample of the same workload (rank 0, N=1 only)
and gpurun_out/bench_detail.json (--detail PATH) holds what does not fit the line: `kernels` (per launch shape of
libgps_hip.so: time, algorithmic work, roof fraction), `kernel_families`, `step_kernels` (torch.profiler's top kernels).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Library-GEMM solution selection: hipBLASLt's default heuristics leave 10-20 % on the table for
# the (tokens x 768 / 2048 / 3072 / 30522) shapes of this step.  tools/tune_gemm.sh records, once, the
# fastest solution per shape with PyTorch's TunableOp; the committed CSV is used READ-ONLY here (no
# tuning inside bench.py).  Must be set before torch is imported.
_TUNED = os.path.join(ROOT, "profiles", "tunableop_gfx950.csv")   # torch appends the device ordinal: ...9500.csv
if os.path.exists(_TUNED.replace(".csv", "0.csv")) and not os.environ.get("GPS_NO_TUNABLEOP_FILE") \
        and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ:
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = _TUNED
    # every rank reads the SAME table: torch looks for <name><device ordinal>.csv, the committed file is ordinal 0's
    _lr = os.environ.get("LOCAL_RANK", "0")
    if _lr.isdigit() and int(_lr) > 0 and not os.path.exists(_TUNED.replace(".csv", f"{int(_lr)}.csv")):
        try:
            import shutil
            import tempfile
            _d = os.path.join(tempfile.gettempdir(), "gps_tunableop")
            os.makedirs(_d, exist_ok=True)
            shutil.copyfile(_TUNED.replace(".csv", "0.csv"), os.path.join(_d, f"tunableop_gfx950{int(_lr)}.csv"))
            os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(_d, "tunableop_gfx950.csv")
        except OSError:
            pass


def _pin_rank_to_its_cores() -> None:
    """One process per GPU: give local rank r the r-th slice of the host's cores (the launch thread, the autograd thread
    and RCCL's proxy threads of a rank then stay on one set of cores instead of migrating over all 256)."""
    try:
        world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world <= 1 or not hasattr(os, "sched_setaffinity"):
            return
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per >= 2:
            os.sched_setaffinity(0, cores[local * per:(local + 1) * per])
            os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(per, 16))))
    except (OSError, ValueError):
        pass


_pin_rank_to_its_cores()

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp8": 5000.0}   # dense MFMA peaks, same guide
def _latest_pmc_traffic() -> str:
    """profiles/r<N>/pmc_traffic.json of the latest round that has one (rocprofv3 --pmc passes, tools/pmc_traffic.py)."""
    import glob
    import re as _r
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")),
                   key=lambda p_: int(_r.search(r"r(\d+)", os.path.basename(os.path.dirname(p_))).group(1)))
    return cands[-1] if cands else os.path.join(ROOT, "profiles", "r2", "pmc_traffic.json")


PMC_TRAFFIC = _latest_pmc_traffic()
N_CLS = 607
MAX_LINE_BYTES = 4000   # the driver keeps the last ~8 KB of stdout: the final JSON line must fit with room to spare


def final_line(result: dict) -> str:
    """The ONE JSON line the driver parses (last line of stdout), bounded in size."""
    line = json.dumps(result)
    if len(line) >= MAX_LINE_BYTES:
        raise RuntimeError(f"bench line is {len(line)} bytes; the driver keeps only the last ~8 KB of stdout -- move "
                           f"detail into write_detail()")
    return line


def write_detail(detail: dict, path=None) -> None:
    """Per-launch-shape rows, kernel families and the profiler's view of the step: too large for the one JSON line the
    driver parses, so they go to a file (default gpurun_out/bench_detail.json, merged back by gpurun)."""
    path = path or os.environ.get("GPS_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        print(f"[bench] per-kernel detail written to {path}", file=sys.stderr)
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr)


# workload presets: BASELINE.json configs[1] (default), configs[3] and configs[4]
WORKLOADS = {
    # GPS pre-train step of all_pretrain.yaml: B = 64 (:167), 80 objects x 1024 points, 50 + 300 tokens
    "pretrain": dict(batch=64, n_obj=80, n_pts=1024, txt_len=50, heads="pretrain", scene_cap=True,
                     losses=["lm_cls_loss", "TextObjWithinBatch", "TextSceneBetweenBatch"]),
    # ScanRefer grounding fine-tune (finetune/scanrefer_finetune.yaml:164,245-251): B = 256, GroundHeadV1, og3d_loss
    "finetune": dict(batch=256, n_obj=80, n_pts=1024, txt_len=50, heads="ground", scene_cap=False, losses=["og3d_loss"]),
    # stress: 256 objects x 2048 points, 256-token text (T = 512 joint tokens), pre-train losses without the caption.
    # B = 64 since round 5 (it was 8: GEMMs of 2 048 rows on 256 CUs): 1 042 -> 2 447 pairs/s, 8 % of the 288 GB in use
    # (`--batch 8` reproduces the old figure)
    "stress": dict(batch=64, n_obj=256, n_pts=2048, txt_len=256, heads="pretrain", scene_cap=False,
                   losses=["lm_cls_loss", "TextObjWithinBatch"]),
}


def gps_pretrain_cfg(lang_path: str, num_gpu: int = 1, workload: str = "pretrain"):
    """Model/solver section of configs/final/all_pretrain.yaml:172-258 (reference); `workload` swaps the head and
    loss list for the fine-tune preset (finetune/scanrefer_finetune.yaml:245-258)."""
    from sceneverse_amd.common.config import ConfigNode
    w = WORKLOADS[workload]
    losses = list(w["losses"])
    heads = {"pretrain": {"head_list": ["pretrain_head"],
                          "pretrain_head": {"name": "OVPretrainHead",
                                            "args": {"hidden_size": 768, "vocab_size": 30522}}},
             "ground": {"head_list": ["ground_head"],
                        "ground_head": {"name": "GroundHeadV1",
                                        "args": {"hidden_size": 384, "input_size": 768, "sem_cls_size": 607,
                                                 "dropout": 0.3, "detach_all_aux_loss": True}}}}[w["heads"]]
    return ConfigNode({
        "num_gpu": num_gpu, "task": "Pretrain",
        "data": {"args": {"use_scene_cap": bool(w["scene_cap"])}},
        "solver": {"lr": 5e-4, "grad_norm": 5.0,
                   "optim": {"name": "AdamW", "args": {"betas": [0.9, 0.98]}},
                   "sched": {"name": "warmup_cosine", "args": {"warmup_steps": 500, "minimum_ratio": 0.1}}},
        "model": {
            "name": "OpenVocab",
            "language": {"name": "BERTLanguageEncoder",
                         "args": {"weights": None, "hidden_size": 768, "num_hidden_layers": 4,
                                  "num_attention_heads": 12, "type_vocab_size": 2}, "lr": 1e-5},
            "vision": {"name": "PointOpenVocabEncoder",
                       "args": {"backbone": "pointnet++", "hidden_size": 768, "freeze": True,
                                "path": None, "num_attention_heads": 12, "spatial_dim": 5,
                                "num_layers": 4, "dim_loc": 6, "dim_feedforward": 2048,
                                "attn_type": "spatial", "pairwise_rel_type": "center",
                                "use_matmul_label": False, "lang_type": "bert",
                                "lang_path": lang_path}, "lr": 1e-4},
            "grounding": {"name": "UnifiedSpatialCrossEncoderV2",
                          "args": {"hidden_size": 768, "num_attention_heads": 12, "num_layers": 4,
                                   "dim_feedforward": 2048, "dim_loc": 6}, "lr": 1e-4},
            "inter": "before",
            "heads": heads,
            "loss_list": losses, "vis_loss_list": losses,
        },
    })


def _lang_dir(seed: int = 0) -> str:
    d = tempfile.mkdtemp(prefix="gps_txt_")
    g = torch.Generator().manual_seed(1234 + seed)
    torch.save(0.02 * torch.randn(N_CLS, 768, generator=g),
               os.path.join(d, "scannet_607_bert-base-uncased_id.pth"))
    return d


def _cpu_baseline_inproc(batch_size: int, steps: int, n_obj: int, n_pts: int, threads: int = 0) -> dict:
    """The oracle port of the same training step on the host cores: functional fp32 model of
    oracle/gps_torch_reference.py over leaf parameter tensors, C point ops, HF BERT, AdamW."""
    from oracle import gps_torch_reference as R
    from sceneverse_amd.data.synthetic import synth_batch
    from transformers import BertConfig, BertModel

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if threads > 0:
        cores = max(1, min(cores, threads))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    # parameter container with the reference's names, built from shapes only (no product forward)
    from sceneverse_amd.model.build import build_model
    cfg = gps_pretrain_cfg(_lang_dir())
    shapes = build_model(cfg).state_dict()
    sd = {}
    for k, v in shapes.items():
        if k.startswith("lang_encoder."):
            continue
        t = v.detach().clone().float() if torch.is_floating_point(v) else v.clone()
        frozen = k.startswith("point_encoder.point_feature_extractor") or k.endswith("text_features") \
            or "running_" in k or not torch.is_floating_point(v)
        sd[k] = t if frozen else t.requires_grad_(True)
    del shapes
    bert = BertModel(BertConfig(hidden_size=768, num_hidden_layers=4, num_attention_heads=12,
                                type_vocab_size=2)).train()
    for m in bert.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    lang = lambda ids, masks: bert(ids, masks).last_hidden_state  # noqa: E731
    logit_scale = torch.tensor(1 / 0.07)
    params = [t for t in sd.values() if t.requires_grad] + [p for p in bert.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.98))
    data = synth_batch(batch_size, n_obj=n_obj, n_pts=n_pts, seed=123)

    def one_step():
        out = R.openvocab_forward(sd, data, lang)
        loss = R.pretrain_losses(out, data, logit_scale)["total_loss"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return loss.item()

    one_step()  # warm-up
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):     # bounded sample: at least 3 steps, then stop once ~30 s of CPU work are spent
        one_step()
        done += 1
        if done >= 3 and time.perf_counter() - t0 > 30.0:
            break
    steps = done
    dt = time.perf_counter() - t0
    return {"value": batch_size * steps / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of the same step at B={batch_size} ({n_obj} obj x {n_pts} pts, 50+300 tokens), "
                      f"fp32 torch on {cores} threads, 1 warm-up; {dt:.1f} s"}


def cpu_baseline(batch_size: int, steps: int, n_obj: int, n_pts: int) -> dict:
    """The CPU baseline, each kind in its own child process with a wall-clock bound, on min(64, host cores) torch threads
    (with all 256 threads of the GPU box's host the small operators of this model thrash and the sample does not finish
    in 120 s -- measured in round 2 -- so that attempt only runs when GPS_CPU_BASELINE_THREADS asks for it):
      kind "reference"  the reference's OWN model.openvocab.OpenVocab / modules/ / optim.loss.Loss, imported unmodified
                        (oracle/ref_cpu_baseline.py: /root/reference, or oracle/_ref/ref_python.zip on the GPU box), its
                        point ops on the CPU oracle -- what SURVEY.md 8(d) defines as the baseline;
      kind "port"       the functional fp32 restatement of the same step (oracle/gps_torch_reference.py), reported beside
                        it under "port" (and alone, with a note, where the reference's files are not available).
    The record says how many threads were used (`cores`) and how many the host has (`host_cores`)."""
    import subprocess
    total = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("GPS_CPU_BASELINE_THREADS", str(min(64, total))))
    limit = float(os.environ.get("GPS_CPU_BASELINE_LIMIT_S", "150"))

    def worker(kind):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker",
               json.dumps([batch_size, steps, n_obj, n_pts, threads, kind])]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                out = json.loads(lines[-1])
                out["host_cores"] = total
                return out, None
            return None, f"{kind}, {threads} threads: worker failed ({r.stderr[-200:]!r})"
        except subprocess.TimeoutExpired:
            return None, f"{kind}, {threads} threads: no result within {limit:.0f} s"

    ref, ref_note = worker("reference")
    port, port_note = worker("port")
    if ref is not None:
        ref["port"] = port if port is not None else {"value": None, "note": port_note}
        return ref
    if port is not None:
        port["note"] = f"kind 'reference' not measured: {ref_note}"
        return port
    return {"value": None, "unit": "pairs/s", "cores": None, "kind": "port", "sample": "not measured", "host_cores": total,
            "note": f"{ref_note}; {port_note}"}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_command(argv: list, gpus: int, port: int) -> list:
    """The torchrun command this script re-executes itself under when asked for N > 1 GPUs without a launcher
    environment (one rank per GPU; the reference's launcher does the same job: common/launch_utils.py:26-42)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def resolve_world(gpus: int, env=os.environ):
    """-> ("spawn", None) | ("run", world).  Raises SystemExit when --gpus and the launcher disagree."""
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    has_launcher = "RANK" in env and "WORLD_SIZE" in env
    if not has_launcher:
        return ("spawn", None) if gpus > 1 else ("run", 1)
    world = int(env["WORLD_SIZE"])
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
    return ("run", world)


def bq_group_unfused(batch) -> dict:
    """The north star's `ball_query + group` figure: the six launches of the reference API at this workload
    (SA1: ball_query, group xyz, group rgb; SA2: ball_query, group xyz, group feats), each timed as a HIP-graph
    replay of 10 calls, algorithmic bytes of SURVEY.md 8(d) (365 888 B per object) over the summed time."""
    from sceneverse_amd.pointnet2 import _ext as hip
    pcs = batch["obj_fts"].reshape(-1, batch["obj_fts"].shape[-2], 6)
    xyz = pcs[..., :3].contiguous()
    rgb = pcs[..., 3:].transpose(1, 2).contiguous()
    b, n = xyz.shape[0], xyz.shape[1]
    xyz_t = xyz.transpose(1, 2).contiguous()
    fps = hip.furthest_point_sampling(xyz, 32)
    new_xyz = hip.gather_points(xyz_t, fps).transpose(1, 2).contiguous()
    idx = hip.ball_query(new_xyz, xyz, 0.2, 32)
    fps2 = hip.furthest_point_sampling(new_xyz, 16)
    nx_t = new_xyz.transpose(1, 2).contiguous()
    nx2 = hip.gather_points(nx_t, fps2).transpose(1, 2).contiguous()
    idx2 = hip.ball_query(nx2, new_xyz, 0.4, 32)
    feats = torch.randn(b, 128, 32, device=xyz.device)
    ops = [("ball_query SA1", lambda: hip.ball_query(new_xyz, xyz, 0.2, 32), b * ((n + 32) * 12 + 32 * 32 * 4)),
           ("ball_query SA2", lambda: hip.ball_query(nx2, new_xyz, 0.4, 32), b * ((32 + 16) * 12 + 16 * 32 * 4)),
           ("group SA1 xyz", lambda: hip.group_points(xyz_t, idx), b * (3 * n * 4 + 1024 * 4 + 3 * 1024 * 4)),
           ("group SA1 rgb", lambda: hip.group_points(rgb, idx), b * (3 * n * 4 + 1024 * 4 + 3 * 1024 * 4)),
           ("group SA2 xyz", lambda: hip.group_points(nx_t, idx2), b * (3 * 32 * 4 + 512 * 4 + 3 * 512 * 4)),
           ("group SA2 feats", lambda: hip.group_points(feats, idx2), b * (128 * 32 * 4 + 512 * 4 + 128 * 512 * 4))]
    rows, tot_us, tot_b = [], 0.0, 0
    for name, fn, nbytes in ops:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 10
        rows.append({"op": name, "us": round(us, 2), "algorithmic_bytes": nbytes,
                     "frac_hbm": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)})
        tot_us += us
        tot_b += nbytes
    return {"launches": rows, "us": round(tot_us, 2), "algorithmic_bytes": tot_b,
            "achieved_GBps": round(tot_b / tot_us / 1e3, 1), "frac_hbm": round(tot_b / tot_us / 1e3 / HBM_PEAK_GBS, 4)}


def profile_step_kernels(step, batch, n_steps: int = 2) -> list:
    """All kernels of the step by total time (torch.profiler over eager steps): [{name, ms_per_step, launches}]."""
    try:
        from torch.profiler import ProfilerActivity, profile
        step.step(dict(batch))
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(n_steps):
                step.step(dict(batch))
            torch.cuda.synchronize()
        rows = []
        for ev in prof.key_averages():
            t = getattr(ev, "device_time_total", None)
            if t is None:
                t = getattr(ev, "cuda_time_total", 0.0)
            if t and ev.count:
                rows.append({"name": ev.key[:120], "ms_per_step": round(t / 1e3 / n_steps, 4),
                             "launches_per_step": round(ev.count / n_steps, 1)})
        rows.sort(key=lambda r: -r["ms_per_step"])
        return rows
    except Exception as e:  # noqa: BLE001 -- the profiler pass is reporting only
        print(f"[bench] torch.profiler pass failed ({type(e).__name__}: {e})", file=sys.stderr)
        return []


# in-scope transformer work per pair, fwd + bwd (probe-measured on the reference code):
# 4 spatial layers at T = 80 (3.643 GFLOP fwd) + 4 joint layers at T = 130 (5.936 GFLOP fwd), x 3
IN_SCOPE_TRANSFORMER_GFLOP_PER_PAIR = 28.73


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="pretrain",
                    help="workload preset: pretrain = BASELINE configs[1] (the headline metric), finetune = configs[3] "
                         "(B = 256, GroundHeadV1, og3d_loss + one eval pass), stress = configs[4] (256 obj x 2048 pts, "
                         "256 tokens)")
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default: the preset's)")
    ap.add_argument("--n-obj", type=int, default=None)
    ap.add_argument("--n-pts", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--fp32", action="store_true", help="disable bf16 autocast (debug)")
    ap.add_argument("--bf16-grads", action="store_true",
                    help="N > 1: gradient buckets as bf16 on the wire with fp32 accumulation (all-to-all + all-gather, "
                         "common/dist_utils.bf16_wire_fp32_acc_hook) instead of the reference's fp32 all-reduce")
    ap.add_argument("--graph-dp", action="store_true",
                    help="force the split-graph data-parallel form at world_size 1 (what N > 1 runs; for A/B)")
    ap.add_argument("--no-native-gemm", action="store_true",
                    help="projections / FFNs through F.linear -> hipBLASLt instead of libgps_hip.so's MFMA GEMMs "
                         "(A/B of sceneverse_amd/modules/layers/gemm.py)")
    ap.add_argument("--no-fused-emb", action="store_true",
                    help="BERT word-table gradient through torch's sort-based embedding backward (A/B of "
                         "modules/language/fused_embedding.py)")
    ap.add_argument("--no-varlen", action="store_true",
                    help="BERT on the padded (B, L) row batch instead of the compacted valid tokens (A/B of "
                         "modules/language/bert.py's variable-length fast path)")
    ap.add_argument("--wgrad-overlap", action="store_true",
                    help="weight-gradient GEMMs on a second stream beside the input-gradient chain (modules/layers/gemm.py "
                         "deferred_wgrads; measured slower on one GPU, off by default)")
    ap.add_argument("--no-graph", action="store_true",
                    help="single-GPU runs replay the step as one HIP graph by default; this keeps it eager")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--sa-bf16", action="store_true",
                    help="set-abstraction MLPs with ONE bf16 product per multiply-accumulate (opt-in; default: split-bf16 x3)")
    ap.add_argument("--no-self-check", action="store_true",
                    help="N > 1: skip the start-up comparison of the split-graph step with eager torch DDP")
    ap.add_argument("--detail", default=None, help="where the per-kernel detail JSON goes (default gpurun_out/bench_detail.json)")
    ap.add_argument("--fp8", action="store_true",
                    help="attention-core products (QK^T, PV) on the OCP e4m3 MFMA path (BASELINE configs[4]: --config stress --fp8)")
    ap.add_argument("--no-wgrad-group", action="store_true",
                    help="weight gradients one split-K GEMM at a time instead of one grouped launch per backward segment "
                         "(A/B of modules/layers/gemm.grouped_wgrads)")
    ap.add_argument("--wgrad-one-queue", action="store_true",
                    help="grouped weight gradients: one global longest-first tile queue instead of the XCD-local queues "
                         "(A/B of gps_gemm_wgrad_grouped_set_xcd_queues)")
    ap.add_argument("--eager-ddp", action="store_true",
                    help="N > 1: torch DDP in eager mode (bucketed all-reduce from autograd hooks) instead of the "
                         "split-graph data-parallel step")
    ap.add_argument("--all-object-slots", action="store_true",
                    help="frozen object encoder on every object slot, pad clouds included (off: distinct clouds only)")
    ap.add_argument("--text-lengths", choices=["uniform", "full"], default="uniform",
                    help="length law of the synthetic texts.  uniform (default, builder-chosen): sentence U{6..50}, scene caption "
                         "U{30..300} valid tokens, right-padded -- real captions vary in length, the reference pads them to "
                         "max_length (configs/final/all_pretrain.yaml:35-36,46) and runs every padded row; full: every text at "
                         "its maximum length (no padded rows to skip: what `value_full_length_text` reports beside `value`)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the reporting-only passes after the timed region (profiler, unfused point-op timing)")
    args = ap.parse_args()

    if args.cpu_baseline_worker:
        b, st, no, npts, th, kind = json.loads(args.cpu_baseline_worker)
        if kind == "reference":            # the reference's own modules/ + model/ on the CPU oracle ops
            from oracle import ref_cpu_baseline
            print(json.dumps(ref_cpu_baseline.run(b, st, no, npts, th)), flush=True)
        else:
            print(json.dumps(_cpu_baseline_inproc(b, st, no, npts, th)), flush=True)
        return
    preset = WORKLOADS[args.config]
    args.batch = preset["batch"] if args.batch is None else args.batch
    args.n_obj = preset["n_obj"] if args.n_obj is None else args.n_obj
    args.n_pts = preset["n_pts"] if args.n_pts is None else args.n_pts
    mode, world_expected = resolve_world(args.gpus)
    if mode == "spawn":
        import subprocess
        raise SystemExit(subprocess.call(spawn_command(sys.argv[1:], args.gpus, _free_port())))

    from sceneverse_amd.common import dist_utils
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    from sceneverse_amd.pointnet2 import _ext as hip_ext

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the GPS hot path has no CPU fallback")
    if args.no_fused_emb:
        from sceneverse_amd.modules.language import bert as _bert
        _bert.set_fused_embedding(False)
    if args.no_varlen:
        from sceneverse_amd.modules.language import bert as _bert
        _bert.set_varlen(False)
    # a shape the fused attention core stops supporting must FAIL the run, not fall back to torch ops silently
    from sceneverse_amd.modules.layers import transformers as _tf
    _tf.set_attention_backend("hip")
    if args.sa_bf16:
        from sceneverse_amd.pointnet2 import pointnet2_modules as _sam
        _sam.set_sa_precision("bf16")
    if args.all_object_slots:
        from sceneverse_amd.modules.layers import pointnet as _pn
        _pn.set_distinct_clouds(False)
    if args.fp8:
        # BASELINE configs[4]: Q K^T and P V of every bf16 attention call on the OCP e4m3 MFMA (forward)
        from sceneverse_amd.modules.layers import fused_attention as _fa
        _fa.set_fp8_products(True)
    # GPS_BENCH_SHARE_GPU=1 (tests only): every rank uses cuda:0 and the collectives go over gloo, so
    # the N > 1 code path can be exercised end to end on a one-GPU box (RCCL refuses two ranks on one GPU)
    share = os.environ.get("GPS_BENCH_SHARE_GPU") == "1"
    rank, world, local = dist_utils.init_from_env("gloo" if share else "nccl")
    if world != world_expected:
        raise SystemExit(f"--gpus {args.gpus} but {world} rank(s) were initialised")
    if share:
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    cfg = gps_pretrain_cfg(_lang_dir(), num_gpu=world, workload=args.config)
    # One GPU: the whole step as one HIP graph.  N > 1 [r4]: the split-graph data-parallel step (forward | losses + top
    # backward | bottom backward | clip + AdamW as HIP graphs around the eager RCCL all-gather and the two all-reduces);
    # --eager-ddp keeps torch DDP in eager mode (~1 200 launches per step from Python) for A/B.
    use_graph = not args.no_graph and not (world > 1 and args.eager_ddp)
    # N > 1, split-graph step: before anything is timed, the same engine class (dropout zeroed, a few scenes) runs its
    # eager warm-up steps, its captures and a first replay beside torch DDP in eager mode from the same weights; losses,
    # parameter checksums and the cross-rank identity of the parameters must agree (sceneverse_amd.engine.dp_self_check).
    # A failed check does not abort the run: it falls back to --eager-ddp and says so in the line.
    self_check = None
    if world > 1 and use_graph and not args.no_self_check:
        from sceneverse_amd.engine import dp_self_check

        def make_engine(eager_ddp):
            c = gps_pretrain_cfg(_lang_dir(), num_gpu=world, workload=args.config)
            e = GPSTrainStep(c, device=dev, amp_dtype=None if args.fp32 else torch.bfloat16, graph=not eager_ddp, seed=4321,
                             native_gemm=not args.no_native_gemm, wgrad_group=not args.no_wgrad_group)
            for m in e.model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
                if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
                    m.dropout = 0.0
            lm = getattr(getattr(e.model, "lang_encoder", None), "model", None)
            if lm is not None:
                lm.config.hidden_dropout_prob = lm.config.attention_probs_dropout_prob = 0.0
                for m in lm.modules():
                    if hasattr(m, "dropout") and hasattr(m.dropout, "p"):
                        m.dropout.p = 0.0
            return e
        nb = min(args.batch, 8)
        check_batches = []
        for i in range(4):
            cb = synth_batch(nb, n_obj=args.n_obj, n_pts=args.n_pts, txt_len=preset["txt_len"], seed=900 + 10 * i + rank, device=dev)
            if not preset["scene_cap"]:
                cb.pop("scene_txt_ids"), cb.pop("scene_txt_masks")
            check_batches.append(cb)
        self_check = dp_self_check(make_engine, check_batches)
        del check_batches
        torch.cuda.empty_cache()
        if not self_check["ok"]:
            if rank == 0:
                print(f"[bench] WARNING: split-graph data-parallel self-check FAILED ({self_check['reason']}); "
                      "falling back to torch DDP in eager mode", file=sys.stderr)
            args.eager_ddp, use_graph = True, False
    if args.wgrad_one_queue:
        from sceneverse_amd import _native as _nat
        _nat.load().gps_gemm_wgrad_grouped_set_xcd_queues(0)
    step = GPSTrainStep(cfg, device=dev, amp_dtype=None if args.fp32 else torch.bfloat16,
                        graph=("dp" if args.graph_dp else use_graph), native_gemm=not args.no_native_gemm,
                        grad_compress=("bf16_fp32acc" if (world > 1 and not share and args.bf16_grads) else None),
                        wgrad_overlap=args.wgrad_overlap, wgrad_group=not args.no_wgrad_group)
    use_graph = step.graph or step.graph_dp
    if step.graph_dp:
        step.time_exchange(True)
    batch = synth_batch(args.batch, n_obj=args.n_obj, n_pts=args.n_pts, txt_len=preset["txt_len"], seed=42 + rank,
                        device=dev)
    if not preset["scene_cap"]:
        batch.pop("scene_txt_ids"), batch.pop("scene_txt_masks")

    def full_length_texts(src, seed):
        """{ids / masks} of `src` with every padding position turned into an ordinary token ([SEP] last): all rows valid."""
        out = {}
        g = torch.Generator(device="cpu").manual_seed(seed)
        for ids_k, mask_k in (("txt_ids", "txt_masks"), ("scene_txt_ids", "scene_txt_masks")):
            if ids_k not in src:
                continue
            ids = src[ids_k].clone()
            mk = src[mask_k]
            fill = torch.randint(1000, 30522, ids.shape, generator=g).to(ids.device)
            ids = torch.where(mk != 0, ids, fill)
            ids[:, -1] = 102
            out[ids_k], out[mask_k] = ids, torch.ones_like(mk)
        return out

    if args.text_lengths == "full":
        batch.update(full_length_texts(batch, 4242 + rank))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph_note = None
    if use_graph:
        # engine preparation (untimed, before the W warm-up steps): eager steps + one capture.  A failed capture
        # is an error, not a silent eager run: `value` would be ~25 % lower without anyone noticing.
        for _ in range(step.graph_warmup + 1):
            step.step(dict(batch))
        torch.cuda.synchronize()
        graph_note = ("whole step replayed as one HIP graph" if step.graph else
                      "5 HIP graphs per step (forward | losses + top backward | text-encoder backward | object-encoder "
                      "backward | clip+AdamW) around the eager RCCL all-gather and three all-reduces (two of them beside "
                      "the next backward graph)")
        if step.wgrad_group:
            graph_note += "; weight gradients of a backward segment in one grouped launch"
    if use_graph and step.static_inputs() is not None:
        # the batch lives in the graph's own input buffers from here on (what a loader writing into them would hand
        # over): no per-step copy of the 126 MB of object points into the static buffers
        static = step.static_inputs()
        for k, v in static.items():
            v.copy_(batch[k])
        batch = {**batch, **static}
    for _ in range(args.warmup):
        step.step(dict(batch))
    lang = getattr(step.model, "lang_encoder", None)
    want_path = "padded" if args.no_varlen else "varlen"
    if lang is not None and hasattr(lang, "last_path") and not args.fp32 and lang.last_path not in (want_path, None):
        raise SystemExit(f"bench: the text encoder ran its '{lang.last_path}' formulation, expected '{want_path}' "
                         f"(libgps_hip.so fast path): refusing to report a number measured on a fallback")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step.step(dict(batch))
    barrier()
    dt = time.perf_counter() - t0
    exposed_ms = step.exposed_allreduce_ms() if step.graph_dp else None     # of the last timed step
    # The same step with EVERY sentence at 50 and every caption at 300 tokens (no padded text rows at all): what the
    # variable-length text path gains depends on the caption-length distribution of the synthetic batch (U{30..300}),
    # so the fully-populated figure is measured beside `value`, same process, same graphs (the row counts live on the
    # device), W warm-up + K timed steps.
    dt_full = None
    full = None
    if preset["scene_cap"] and not args.no_extras and args.text_lengths != "full":
        # (`batch` may alias the graph's static input buffers: keep the measured texts aside and put them back after)
        saved_text = {k: batch[k].clone() for k in ("txt_ids", "txt_masks", "scene_txt_ids", "scene_txt_masks")}
        full = {**batch, **full_length_texts(batch, 4242 + rank)}
        for _ in range(max(1, args.warmup)):
            step.step(dict(full))
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step.step(dict(full))
        barrier()
        dt_full = time.perf_counter() - t1
        for k, v in saved_text.items():     # back on the measured batch
            batch[k].copy_(v)
        step.step(dict(batch))
    # [r5] The frozen object encoder runs on the distinct clouds only (pad slots -- constant clouds, the reference's
    # pad_tensors(..., pad=1.0) -- are encoded once; modules/layers/pointnet.py).  How many slots are pads is a property of
    # the batch (builder-chosen: n_real ~ U{20..79} of 80), so the same graph is also timed on a batch WITHOUT pad clouds:
    # every pad slot gets the cloud of its scene's first object (obj_masks unchanged: only the encoder's work changes).
    dt_nopad = None
    dt_cons = None
    pad_frac = None
    if "obj_fts" in batch and "obj_masks" in batch and not args.no_extras and not args.all_object_slots:
        pads = batch["obj_masks"].logical_not()
        pad_frac = float(pads.float().mean().item())
        saved_obj = batch["obj_fts"].clone()
        first = batch["obj_fts"][:, 0:1].expand_as(batch["obj_fts"])
        batch["obj_fts"].copy_(torch.where(pads[:, :, None, None], first, batch["obj_fts"]))
        for _ in range(max(1, args.warmup)):
            step.step(dict(batch))
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step.step(dict(batch))
        barrier()
        dt_nopad = time.perf_counter() - t1
        # ... and with BOTH builder-chosen savings off at once (no pad clouds to skip, no padded text rows to skip): the
        # jointly conservative figure
        if full is not None:
            saved_text = {k: batch[k].clone() for k in ("txt_ids", "txt_masks", "scene_txt_ids", "scene_txt_masks")}
            cons = {**batch, **{k: full[k] for k in saved_text}}
            for _ in range(max(1, args.warmup)):
                step.step(dict(cons))
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step.step(dict(cons))
            barrier()
            dt_cons = time.perf_counter() - t1
            for k, v in saved_text.items():
                batch[k].copy_(v)
        batch["obj_fts"].copy_(saved_obj)
        del saved_obj
        step.step(dict(batch))
    # [r4] The loader-side object sampler in the measured step (data/datasets/base.py:697-740): raw scans
    # resident in HBM (packed once, untimed), every step draws + normalises the 80 x 1024 points of every scene straight
    # into the graph's static input buffers (gps_obj_processing_post) and then replays the step.
    dt_samp = None
    if args.config == "pretrain" and not args.no_extras and use_graph and step.static_inputs() is not None:
        import numpy as np
        from sceneverse_amd.data import gpu_objects as G
        rng = np.random.default_rng(7 + rank)
        scans = G.PackedScans(dev)
        n_real = batch["obj_masks"].sum(1).tolist()                 # as many real objects per scene as the measured batch
        for si in range(args.batch):
            n = int(n_real[si])
            ks = rng.integers(50, 4001, size=n)
            pts = (rng.normal(size=(int(ks.sum()), 3)) * rng.uniform(0.05, 1.0, size=3)).astype(np.float32)
            scans.add_scan(f"s{si}", pts, rng.integers(0, 256, size=(int(ks.sum()), 3), dtype=np.uint8),
                           np.repeat(np.arange(n), ks), list(range(n)))
        scans.finalize()
        slots = G.batch_rows(scans, [f"s{i}" for i in range(args.batch)], args.n_obj).to(dev)
        static = step.static_inputs()
        outbuf = {k: static[k] for k in ("obj_fts", "obj_locs", "obj_masks")}
        for i in range(max(1, args.warmup)):
            G.obj_processing_post(scans, slots, args.n_pts, seed=900 + i, out=outbuf)
            step.step(dict(batch))
        barrier()
        t2 = time.perf_counter()
        for i in range(args.steps):
            G.obj_processing_post(scans, slots, args.n_pts, seed=1000 + i, out=outbuf)
            step.step(dict(batch))
        barrier()
        dt_samp = time.perf_counter() - t2
    # [r6] The same step with the two set-abstraction levels of the frozen point encoder in their single-product mode
    # (`--sa-bf16`: one bf16 product per multiply-accumulate instead of the fp32-accurate split-bf16 three; opt-in, features
    # within 5.8e-3 max of the fp32 path): the product count is baked into the captured launches, so this is a second engine
    # (same configuration and seed), W warm-up + K timed steps.
    dt_sa16 = None
    if (args.config == "pretrain" and not args.no_extras and not args.sa_bf16 and not args.fp32 and world == 1 and step.graph
            and not args.all_object_slots):
        from sceneverse_amd.pointnet2 import pointnet2_modules as _sam
        _sam.set_sa_precision("bf16")
        try:
            step2 = GPSTrainStep(cfg, device=dev, amp_dtype=torch.bfloat16, graph=True, native_gemm=not args.no_native_gemm,
                                 wgrad_group=not args.no_wgrad_group)
            b2 = synth_batch(args.batch, n_obj=args.n_obj, n_pts=args.n_pts, txt_len=preset["txt_len"], seed=42 + rank, device=dev)
            if not preset["scene_cap"]:
                b2.pop("scene_txt_ids"), b2.pop("scene_txt_masks")
            if args.text_lengths == "full":
                b2.update(full_length_texts(b2, 4242 + rank))
            for _ in range(step2.graph_warmup + 1 + max(1, args.warmup)):
                step2.step(dict(b2))
            barrier()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                step2.step(dict(b2))
            barrier()
            dt_sa16 = time.perf_counter() - t3
            del step2, b2
            torch.cuda.empty_cache()
        finally:
            _sam.set_sa_precision("bf16x3")
    # Per-launch durations (HIP events around every native call) are taken from three EXTRA eager steps of the same
    # workload right after the timed region, on every rank (the steps contain the data-parallel collectives): a
    # replayed graph cannot host event pairs, and in the eager modes the ~1 200 event records per step would sit on
    # the host's launch path inside the number being reported.
    saved = (step.graph, step.graph_dp) if use_graph else None
    saved_overlap = step.wgrad_overlap
    step.wgrad_overlap = False      # per-kernel durations are taken with every launch alone on the GPU (one stream)
    if use_graph:
        step.graph = step.graph_dp = False
        step.step(dict(batch))
    hip_ext.profile_start()
    for _ in range(3):
        step.step(dict(batch))
    kern = hip_ext.profile_stop()
    for k in kern.values():
        k["launches"] = k["launches"] * args.steps / 3.0
    t = torch.tensor([dt, dt_full or 0.0, dt_samp or 0.0, dt_nopad or 0.0], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t[0].item())
    dt_full = float(t[1].item()) if dt_full is not None else None
    dt_nopad = float(t[3].item()) if dt_nopad is not None else None
    dt_samp = float(t[2].item()) if dt_samp is not None else None
    final_loss = float(loss)
    step_kernels, bqg, eval_metrics = [], None, None
    if args.config == "finetune":
        # one evaluation pass through the grounding metrics (evaluator/scanrefer_eval.py:14-87); synthetic boxes:
        # the annotated target is the only object above both IoU thresholds
        from sceneverse_amd.engine import scanrefer_accuracy
        out, _, _ = step.evaluate(dict(batch))
        onehot = torch.zeros_like(out["og3d_logits"], dtype=torch.long)
        onehot.scatter_(1, batch["tgt_object_id"], 1)
        eval_metrics = scanrefer_accuracy(out["og3d_logits"].float(), onehot, onehot)
    if rank == 0 and not args.no_extras:
        if world == 1:
            step_kernels = profile_step_kernels(step, batch)        # eager steps (graph flags are off here)
        if args.config == "pretrain":
            bqg = bq_group_unfused(batch)
    if saved is not None:
        step.graph, step.graph_dp = saved
    step.wgrad_overlap = saved_overlap

    if rank == 0:
        pairs_per_s = args.batch * world * args.steps / dt
        kernels = []
        for name, k in sorted(kern.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]):
            sec = k["avg_us"] * 1e-6
            gbs = k["bytes_per_launch"] / sec / 1e9
            row = {"kernel": name, "launches_per_step": k["launches"] / args.steps,
                   "avg_us": round(k["avg_us"], 2), "ms_per_step": round(k["avg_us"] * k["launches"] / args.steps / 1e3, 4),
                   "algorithmic_bytes": int(k["bytes_per_launch"]),
                   "achieved_GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
            # which roof bounds the kernel: the one its algorithmic work would take longer on.  The point
            # kernels compute fp32 products as 3 bf16 MFMAs: their ALGORITHMIC flops are priced against the
            # bf16 peak they run on (frac), the executed-MFMA rate is reported as mfma_utilisation.
            t_hbm = k["bytes_per_launch"] / (HBM_PEAK_GBS * 1e9)
            mfma_dtype = k.get("mfma_dtype") or ""
            split3 = "bf16x3" in name
            peak_tf = MFMA_PEAK_TFLOPS.get("bf16" if split3 else mfma_dtype, 0.0)
            algo_flops = k["flops_per_launch"] / 3.0 if split3 else k["flops_per_launch"]
            t_mfma = algo_flops / (peak_tf * 1e12) if peak_tf else 0.0
            if t_mfma > t_hbm:
                tf = algo_flops / sec / 1e12
                row.update({"bound": "mfma", "algorithmic_flops": int(algo_flops), "mfma_dtype": "bf16" if split3 else mfma_dtype,
                            "achieved_TFLOPs": round(tf, 1), "frac": round(tf / peak_tf, 4)})
                if split3:
                    row["mfma_utilisation"] = round(3.0 * tf / peak_tf, 4)
            else:
                row.update({"bound": "hbm", "frac": row["frac_hbm"]})
            row.setdefault("algorithmic_flops", int(algo_flops))
            kernels.append(row)
        # dominant kernel over ALL kernels of the step.  Native launches are grouped the way rocprofv3 groups them --
        # by kernel symbol = template instantiation: the GEMM by (form, epilogue, tile configuration), attention by direction -- so that
        # `roofline` describes the symbol with the largest share of the step and its average launch agrees with the
        # rocprofv3 --stats row of that symbol; the per-shape rows stay in `kernels`.  Compared against every other
        # kernel name the profiler saw (library GEMMs, torch elementwise / reduce kernels).
        import re as _re

        _TILE = {12: "8p256x256", 7: "128x128", 2: "128x128", 6: "128x64"}
        _FORM = {"nt": 0, "nn": 1, "tn": 2}

        def family(name: str) -> str:
            mm = _re.match(r"gemm_(\w+)\(M=(\d+),N=(\d+),K=(\d+),epi=(\d+)\)", name)
            if mm:
                # the tile configuration the library picks for the shape = the kernel symbol rocprofv3 lists the launch
                # under (gemm8p_kernel / gemm_kernel<128,128,..> / gemm_kernel<128,64,..>, each x form x epilogue)
                from sceneverse_amd import _native as _nat
                v = _nat.load().gps_gemm_pick_variant_ex(_FORM.get(mm.group(1), -1), int(mm.group(2)), int(mm.group(3)), int(mm.group(4)), 0,
                                                         int(mm.group(5)))
                return f"gemm_{mm.group(1)}(epi={mm.group(5)},tile={_TILE.get(v, v)})"
            mm = _re.match(r"(attn_\w+)\(L=\d+,spatial=(\d)\)(\[\w+\])?", name)
            if mm:
                return f"{mm.group(1)}(spatial={mm.group(2)}){mm.group(3) or ''}"
            mm = _re.match(r"(gemm_tn_grouped)\(", name)
            if mm:
                return mm.group(1)
            mm = _re.match(r"(gemm_n[nt]_grouped)\(.*epi=(\d+)\)", name)      # paired text / object products: gemm8p_grouped_kernel<form, epi>
            if mm:
                return f"{mm.group(1)}(epi={mm.group(2)},tile=8p256x256)"
            mm = _re.match(r"(add_dropout_layernorm_\w+)\(", name)
            if mm:
                return mm.group(1)
            return name

        groups = {}
        for r in kernels:
            gk = groups.setdefault(family(r["kernel"]), {"kernel": family(r["kernel"]), "ms_per_step": 0.0, "launches_per_step": 0.0,
                                                        "bytes": 0.0, "flops": 0.0, "mfma_dtype": r.get("mfma_dtype"),
                                                        "split3": "bf16x3" in r["kernel"], "shapes": 0})
            gk["ms_per_step"] += r["ms_per_step"]
            gk["launches_per_step"] += r["launches_per_step"]
            gk["bytes"] += r["algorithmic_bytes"] * r["launches_per_step"]
            gk["flops"] += r.get("algorithmic_flops", 0) * r["launches_per_step"]
            gk["mfma_dtype"] = gk["mfma_dtype"] or r.get("mfma_dtype")
            gk["shapes"] += 1
        for gk in groups.values():
            sec = gk["ms_per_step"] * 1e-3
            n_l = max(gk["launches_per_step"], 1e-9)
            gk["avg_us"] = round(gk["ms_per_step"] * 1e3 / n_l, 2)
            gbs = gk["bytes"] / sec / 1e9 if sec else 0.0
            dt_name = "bf16" if gk["split3"] else (gk["mfma_dtype"] or "")
            peak_tf = MFMA_PEAK_TFLOPS.get(dt_name, 0.0)
            t_hbm = gk["bytes"] / (HBM_PEAK_GBS * 1e9)
            t_mfma = gk["flops"] / (peak_tf * 1e12) if peak_tf else 0.0
            if t_mfma > t_hbm:
                tf = gk["flops"] / sec / 1e12
                gk.update({"bound": "mfma", "mfma_dtype": dt_name, "achieved_TFLOPs": round(tf, 1), "frac": round(tf / peak_tf, 4)})
                if gk["split3"]:
                    gk["mfma_utilisation"] = round(3.0 * tf / peak_tf, 4)
            else:
                gk.update({"bound": "hbm", "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)})
        ranked = sorted(groups.values(), key=lambda g_: -g_["ms_per_step"])
        native_names = ("gps_",)
        others = [r for r in step_kernels if not any(tag in r["name"] for tag in native_names)]
        dom = ranked[0] if ranked else None
        dom_other = others[0] if others else None
        traffic = None
        if dom is not None and os.path.exists(PMC_TRAFFIC):
            with open(PMC_TRAFFIC) as f:
                per_launch = json.load(f).get("per_launch_hbm_bytes", {})
            # PMC traffic is recorded per launch shape (tools/pmc_workload.py): report the shape of the family with the
            # largest time per step that has an entry; the counters are from the committed rocprofv3 --pmc passes
            cands = [r for r in kernels if family(r["kernel"]) == dom["kernel"] and r["kernel"] in per_launch]
            if cands:
                r = max(cands, key=lambda r_: r_["ms_per_step"])
                traffic = {"shape": r["kernel"], "hbm_bytes": per_launch[r["kernel"]], "algorithmic_bytes": r["algorithmic_bytes"],
                           "ratio": round(per_launch[r["kernel"]] / max(1, r["algorithmic_bytes"]), 3),
                           "source": os.path.relpath(PMC_TRAFFIC, ROOT)}
            else:
                # the grouped launch is ONE shape whose name carries the problem count: a counter file recorded with a
                # slightly different problem list (one Linear more or less) is still the traffic of this launch
                fam = [k for k in per_launch if k.split("(")[0] == dom["kernel"].split("(")[0]]
                rows = [r for r in kernels if family(r["kernel"]) == dom["kernel"]]
                if len(fam) == 1 and len(rows) == 1:
                    traffic = {"shape": fam[0], "hbm_bytes": per_launch[fam[0]], "algorithmic_bytes": rows[0]["algorithmic_bytes"],
                               "ratio": round(per_launch[fam[0]] / max(1, rows[0]["algorithmic_bytes"]), 3),
                               "source": os.path.relpath(PMC_TRAFFIC, ROOT),
                               "note": f"counters recorded for {fam[0]}, the step launches {rows[0]['kernel']}"}
        if dom is None:
            roofline = None
        elif dom_other is not None and dom_other["ms_per_step"] > dom["ms_per_step"]:
            roofline = {"bound": None, "kernel": dom_other["name"], "achieved": None, "peak": None, "unit": None,
                        "frac": None, "traffic": None, "ms_per_step": dom_other["ms_per_step"],
                        "note": "the kernel with the largest time per step is not a libgps_hip.so launch; its "
                                "algorithmic work is not known to this script",
                        "largest_native": {"kernel": dom["kernel"], "bound": dom["bound"], "frac": dom["frac"],
                                           "ms_per_step": round(dom["ms_per_step"], 4)}}
        elif dom["bound"] == "mfma":
            roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved_TFLOPs"],
                        "peak": MFMA_PEAK_TFLOPS[dom["mfma_dtype"]], "unit": "TFLOP/s", "frac": dom["frac"],
                        "dtype": dom["mfma_dtype"], "traffic": traffic, "ms_per_step": round(dom["ms_per_step"], 4),
                        "launches_per_step": dom["launches_per_step"], "avg_us": dom["avg_us"], "shapes": dom["shapes"],
                        **({"mfma_utilisation": dom["mfma_utilisation"]} if "mfma_utilisation" in dom else {}),
                        **({"note": "all weight+bias gradients of a backward segment in one launch (wgrad_grouped_kernel "
                                    "+ its table-write launches in rocprofv3)"} if dom["kernel"].startswith("gemm_tn_grouped") else
                           {"note": "weight+bias gradient GEMMs; avg_us per gps_gemm_bf16 call = split-K kernel + its "
                                    "splitk_reduce_kernel launch (two rocprofv3 rows)"} if dom["kernel"].startswith("gemm_tn") else {})}
        else:
            roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBps"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": traffic,
                        "ms_per_step": round(dom["ms_per_step"], 4), "launches_per_step": dom["launches_per_step"],
                        "avg_us": dom["avg_us"], "shapes": dom["shapes"]}
        kernel_families = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in g_.items() if k not in ("bytes", "flops", "split3")}
                           for g_ in ranked[:12]]
        attn = [r for r in kernels if r["kernel"].startswith("attn_")]
        attn_flops = sum(r.get("algorithmic_flops", 0) * r["launches_per_step"] for r in attn)
        attn_sec = sum(r["avg_us"] * 1e-6 * r["launches_per_step"] for r in attn)
        attn_bytes = sum(r.get("algorithmic_bytes", 0) * r["launches_per_step"] for r in attn)
        gemm_rows = [r for r in kernels if r["kernel"].startswith("gemm_")]
        gemm_flops = sum(r.get("algorithmic_flops", 0) * r["launches_per_step"] for r in gemm_rows)
        gemm_sec = sum(r["avg_us"] * 1e-6 * r["launches_per_step"] for r in gemm_rows)
        # every native launch's algorithmic (live-row) FLOPs of one step against the whole step's time and the bf16 MFMA peak
        all_flops = sum(r.get("algorithmic_flops", 0) * r["launches_per_step"] for r in kernels)
        whole_step_frac = round(all_flops / (dt / args.steps) / (MFMA_PEAK_TFLOPS["bf16"] * 1e12), 4) if all_flops else None
        in_scope_tf = IN_SCOPE_TRANSFORMER_GFLOP_PER_PAIR * pairs_per_s / 1e3
        headline_full = {
            "ball_query_group_unfused": bqg,
            "transformer_in_scope": None if args.config != "pretrain" else {
                "gflop_per_pair": IN_SCOPE_TRANSFORMER_GFLOP_PER_PAIR, "achieved_TFLOPs": round(in_scope_tf, 1),
                "frac_of_bf16_mfma_peak": round(in_scope_tf / MFMA_PEAK_TFLOPS["bf16"], 4),
                "note": "in-scope transformer FLOPs per pair x whole-step pairs/s: the whole step (BERT, point "
                        "encoder, heads, optimizer) is in the denominator's time"},
            "native_gemms": None if not gemm_sec else {
                "achieved_TFLOPs": round(gemm_flops / gemm_sec / 1e12, 1), "ms_per_step": round(gemm_sec * 1e3, 3),
                "frac_of_bf16_mfma_peak": round(gemm_flops / gemm_sec / 1e12 / MFMA_PEAK_TFLOPS["bf16"], 4)},
            "attention_core": None if not attn_sec else {
                "achieved_TFLOPs": round(attn_flops / attn_sec / 1e12, 1), "ms_per_step": round(attn_sec * 1e3, 3),
                "frac_of_bf16_mfma_peak": round(attn_flops / attn_sec / 1e12 / MFMA_PEAK_TFLOPS["bf16"], 4),
                "achieved_GBps": round(attn_bytes / attn_sec / 1e9, 1),
                "frac_hbm": round(attn_bytes / attn_sec / 1e9 / HBM_PEAK_GBS, 4)},
        }
        # the printed line carries the three north-star fractions only; everything else goes to the detail file
        fp8_rows = [r for r in attn if r["kernel"].endswith("[fp8]")]
        fp8_flops = sum(r.get("algorithmic_flops", 0) * r["launches_per_step"] for r in fp8_rows)
        fp8_sec = sum(r["avg_us"] * 1e-6 * r["launches_per_step"] for r in fp8_rows)
        headline = {
            **({"attention_fp8_forward_frac_fp8_mfma": round(fp8_flops / fp8_sec / 1e12 / MFMA_PEAK_TFLOPS["fp8"], 4),
                "attention_fp8_forward_ms_per_step": round(fp8_sec * 1e3, 3)} if fp8_sec else {}),
            "ball_query_group_unfused_frac_hbm": bqg["frac_hbm"] if bqg else None,
            "ball_query_group_unfused_us": bqg["us"] if bqg else None,
            "native_gemms_frac_bf16_mfma": headline_full["native_gemms"]["frac_of_bf16_mfma_peak"] if gemm_sec else None,
            "native_gemms_ms_per_step": headline_full["native_gemms"]["ms_per_step"] if gemm_sec else None,
            "attention_core_frac_bf16_mfma": headline_full["attention_core"]["frac_of_bf16_mfma_peak"] if attn_sec else None,
            "attention_core_frac_hbm": headline_full["attention_core"]["frac_hbm"] if attn_sec else None,
            "attention_core_ms_per_step": headline_full["attention_core"]["ms_per_step"] if attn_sec else None,
        }
        over = [r["kernel"] for r in kernels if (r.get("frac") or 0) > 1.0]
        if over:
            print(f"[bench] WARNING: launch shapes above their roof (work model wrong?): {over}", file=sys.stderr)
        result = {
            "metric": "GPS pre-train pairs/sec (fwd+bwd)" if args.config == "pretrain" else
                      f"GPS {args.config} pairs/sec (fwd+bwd)",
            "value": round(pairs_per_s, 2),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            **({"value_full_length_text": round(args.batch * world * args.steps / dt_full, 2),
                "ms_per_step_full_length_text": round(1e3 * dt_full / args.steps, 3)} if dt_full else {}),
            **({"value_with_device_sampler": round(args.batch * world * args.steps / dt_samp, 2)} if dt_samp else {}),
            **({"value_no_pad_objects": round(args.batch * world * args.steps / dt_nopad, 2),
                "ms_per_step_no_pad_objects": round(1e3 * dt_nopad / args.steps, 3)} if dt_nopad else {}),
            **({"value_conservative": round(args.batch * world * args.steps / dt_cons, 2),
                "ms_per_step_conservative": round(1e3 * dt_cons / args.steps, 3)} if dt_cons else {}),
            **({"value_sa_bf16": round(args.batch * world * args.steps / dt_sa16, 2),
                "ms_per_step_sa_bf16": round(1e3 * dt_sa16 / args.steps, 3)} if dt_sa16 else {}),
            **({"whole_step_frac_bf16_mfma": whole_step_frac} if whole_step_frac is not None else {}),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp32" if args.fp32 else ("bf16+fp8-attention" if args.fp8 else "bf16"),
            "data": "synthetic",
            "config": {"workload": (f"GPS pre-train step (all_pretrain.yaml model): {args.n_obj} obj x {args.n_pts} pts x 6 ch, "
                                    f"50-token sentence + 300-token scene caption, fwd+loss+bwd+clip+AdamW") if args.config == "pretrain" else
                                   (f"GPS {args.config} step ({'finetune/scanrefer_finetune.yaml head + og3d_loss' if args.config == 'finetune' else 'BASELINE configs[4]'}): "
                                    f"{args.n_obj} obj x {args.n_pts} pts x 6 ch, {preset['txt_len']}-token text, "
                                    f"fwd+loss+bwd+clip+AdamW"),
                       "preset": args.config,
                       **({"eval": eval_metrics} if eval_metrics is not None else {}),
                       **({"sa_mlp": "single bf16 product (opt-in)"} if args.sa_bf16 else {}),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" + (" (ranks share one GPU, gloo: test mode)" if share else ""),
                       "text_rows": "padded (B, L) batch" if args.no_varlen else "valid tokens only (variable-length BERT path)",
                       "object_slots": ("every slot encoded (--all-object-slots)" if args.all_object_slots else
                                        "distinct clouds only: the pad slots (constant clouds) are encoded once; "
                                        "compare with value_no_pad_objects"),
                       **({"pad_object_fraction": round(pad_frac, 4)} if pad_frac is not None else {}),
                       "text_lengths": ("uniform: builder-chosen law (--text-lengths); the reference pads every text to max_length "
                                        "(configs/final/all_pretrain.yaml:35-36,46): see value_full_length_text / value_conservative"
                                        if args.text_lengths == "uniform" else "full: every text at max_length (no padded rows)"),
                       **({"sentence_len": f"U{{6..{preset['txt_len']}}}", "caption_len": "U{30..300}" if preset["scene_cap"] else None,
                           "text_live_row_fraction": round(float(sum(batch[k].float().sum().item() for k in batch if k.endswith("txt_masks")))
                                                           / max(1.0, float(sum(batch[k].numel() for k in batch if k.endswith("txt_masks")))), 4)}),
                       "launch": (graph_note or "eager") + ("; weight-gradient GEMMs on a second stream" if step.wgrad_overlap else ""),
                       **({"grad_exchange": "fp32 all-reduce of the top / text / object range of one flat buffer, the first two "
                                            "beside the next backward graph" if step.graph_dp
                           else "bf16 on the wire, fp32 accumulation (all-to-all + all-gather)"
                           if (args.bf16_grads and not share) else "fp32 all-reduce (DDP buckets)"} if world > 1 else {}),
                       **({"dp_self_check": self_check} if self_check is not None else {}),
                       **({"exposed_allreduce_ms": round(exposed_ms, 3)} if exposed_ms is not None else {}),
                       "final_loss": round(final_loss, 4)},
            "roofline": roofline,
            "headline": headline,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == "pretrain":
            result["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps, args.n_obj, args.n_pts)
            # the CPU leg runs HF BERT on every row of the padded (B, L) batch: its like-for-like GPU figure is
            # value_full_length_text (no padded rows to skip), not `value`
            result["cpu_baseline"]["text_rows"] = "padded (B, L) batch; compare with value_full_length_text"
        detail = dict(result)
        detail["config"] = dict(result["config"],
                                point_ops=("single bf16 product per MAC, fp32 accumulate (opt-in --sa-bf16)" if args.sa_bf16
                                           else "fp32-accurate split-bf16 MFMA (libgps_hip.so)"),
                                gemms="hipBLASLt (A/B run)" if args.no_native_gemm else "libgps_hip.so bf16 MFMA (gps_gemm_bf16)",
                                kernel_timing="HIP events around each native launch in three eager steps right after the timed region")
        detail.update({"headline": headline_full, "kernel_families": kernel_families, "kernels": kernels,
                       "step_kernels": step_kernels[:25]})
        write_detail(detail, args.detail)
        print(final_line(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
