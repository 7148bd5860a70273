"""Per-shape timing of the hand-written bf16 GEMMs (all tile variants) against the library GEMM torch calls for
the same contraction (hipBLASLt / rocBLAS, with the tuned solutions of profiles/tunableop_gfx950*.csv when
PYTORCH_TUNABLEOP_* is set by the caller).  Interleaved rounds inside one process (guide rule 24).

    python tools/gemm_bench.py --json gpurun_out/gemm_bench.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_TUNED = os.path.join(ROOT, "profiles", "tunableop_gfx950.csv")
if os.path.exists(_TUNED.replace(".csv", "0.csv")) and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ:
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = _TUNED
import torch  # noqa: E402

from sceneverse_amd import _native  # noqa: E402
from sceneverse_amd.modules.layers import gemm as G  # noqa: E402

# (tokens, in, out) of the Linears in one GPS pre-train step at B = 64
LAYERS = [(5120, 768, 2376), (5120, 768, 768), (5120, 768, 2048), (5120, 2048, 768),
          (8320, 768, 2304), (8320, 768, 768), (8320, 768, 2048), (8320, 2048, 768),
          (19200, 768, 2304), (19200, 768, 768), (19200, 768, 3072), (19200, 3072, 768),
          (3200, 768, 2304), (3200, 768, 768), (3200, 768, 3072), (3200, 3072, 768),
          # the 22 400-row text batch with only its valid tokens (variable-length path, synthetic bench batch)
          (12608, 768, 2304), (12608, 768, 768), (12608, 768, 3072), (12608, 3072, 768)]


def timeit(fns, rounds=12, inner=10):
    """fns: {name: callable}; every candidate is captured `inner` times into ONE HIP graph and the replays are
    timed interleaved (GPU time, not the host's launch rate); -> {name: median us per call}"""
    graphs = {}
    for k, f in fns.items():
        f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner):
                f()
        g.replay()
        graphs[k] = g
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, g in graphs.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            e.synchronize()
            res[k].append(1e3 * s.elapsed_time(e) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--only", type=int, default=-1, help="index into LAYERS: time just that layer")
    ap.add_argument("--split-sweep", action="store_true", help="weight-gradient form: time every split count 1..20")
    ap.add_argument("--pmc-loop", default=None,
                    help="form:variant -- run ONE kernel configuration of layer --only 30 times (for rocprofv3 --pmc)")
    args = ap.parse_args()
    dev = "cuda"
    lib = _native.load()
    out = []
    layers = LAYERS if args.only < 0 else [LAYERS[args.only]]
    if args.split_sweep:
        res = []
        for T, K, N in layers:
            x = torch.randn(T, K, device=dev).to(torch.bfloat16)
            dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
            dw = torch.empty(N, K, device=dev)
            db = torch.empty(N, device=dev)
            nkt = (T + 63) // 64
            fns = {}
            for sp in range(1, 21):
                if sp > 1 and sp > nkt // 4:
                    break
                ws = torch.empty(max(1, int(lib.gps_gemm_workspace_floats(_native.GEMM_TN, N, K, sp))), device=dev)
                fns[f"s{sp}"] = (lambda sp=sp, ws=ws: G.gemm(_native.GEMM_TN, _native.EPI_F32, N, K, T, dy, N, x, K, dw, K,
                                                             workspace=ws, colsum=db, splits=sp, variant=2))
            t = timeit(fns, args.rounds)
            tiles = ((N + 127) // 128) * ((K + 127) // 128)
            row = {"tokens": T, "in": K, "out": N, "tiles": tiles, "default": int(lib.gps_gemm_pick_splits(_native.GEMM_TN, N, K, T)),
                   "us": {k: round(v, 1) for k, v in t.items()}}
            print(json.dumps(row), flush=True)
            res.append(row)
        if args.json:
            with open(args.json, "w") as f:
                json.dump(res, f, indent=1)
        return
    if args.pmc_loop:
        form, variant = args.pmc_loop.split(":")
        T, K, N = layers[0]
        x = torch.randn(T, K, device=dev).to(torch.bfloat16)
        w = (0.02 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
        dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
        y = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(T, K, dtype=torch.bfloat16, device=dev)
        dw = torch.empty(N, K, device=dev)
        s = int(lib.gps_gemm_pick_splits(_native.GEMM_TN, N, K, T))
        ws = torch.empty(max(1, int(lib.gps_gemm_workspace_floats(_native.GEMM_TN, N, K, s))), device=dev)
        for _ in range(30):
            if form == "nt":
                G.gemm(_native.GEMM_NT, _native.EPI_BIAS, T, N, K, x, K, w, K, y, N, variant=int(variant))
            elif form == "nn":
                G.gemm(_native.GEMM_NN, _native.EPI_BIAS, T, K, N, dy, N, w, K, dx, K, variant=int(variant))
            elif form == "lib":
                torch.nn.functional.linear(x, w)
            else:
                G.gemm(_native.GEMM_TN, _native.EPI_F32, N, K, T, dy, N, x, K, dw, K, workspace=ws, splits=s, variant=int(variant))
        torch.cuda.synchronize()
        return
    for T, K, N in layers:
        x = torch.randn(T, K, device=dev).to(torch.bfloat16)
        w = (0.02 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        b16 = b.to(torch.bfloat16)
        dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
        y = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(T, K, dtype=torch.bfloat16, device=dev)
        dw = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        flops = 2.0 * T * K * N
        row = {"tokens": T, "in": K, "out": N}
        # forward
        fns = {"lib": lambda: torch.nn.functional.linear(x, w, b16)}
        for v in range(13):
            fns[f"v{v}"] = (lambda v=v: G.gemm(_native.GEMM_NT, _native.EPI_BIAS, T, N, K, x, K, w, K, y, N, bias=b, variant=v))
        t = timeit(fns, args.rounds)
        row["fwd_us"] = {k: round(v, 2) for k, v in t.items()}
        # dgrad
        fns = {"lib": lambda: torch.mm(dy, w)}
        for v in range(13):
            fns[f"v{v}"] = (lambda v=v: G.gemm(_native.GEMM_NN, _native.EPI_BIAS, T, K, N, dy, N, w, K, dx, K, variant=v))
        t = timeit(fns, args.rounds)
        row["dgrad_us"] = {k: round(v, 2) for k, v in t.items()}
        # wgrad (+ bias gradient): library = mm + fp32 cast + column sum, as autograd runs it under autocast
        fns = {"lib": lambda: (torch.mm(dy.t(), x).float(), dy.sum(0, dtype=torch.float32))}
        # the 128 x 128 forms at their own split rule (<= 512 resident workgroups), the 256 x 256 two-group form at the
        # library's default for it (one workgroup per CU)
        t128 = ((N + 127) // 128) * ((K + 127) // 128)
        s_old = max(1, min(512 // t128, 16, ((T + 63) // 64) // 8))
        s_new = int(lib.gps_gemm_pick_splits(_native.GEMM_TN, N, K, T))
        for s, vs in ((s_old, (0, 2, 7)), (s_new, (12,))):
            ws = torch.empty(max(1, int(lib.gps_gemm_workspace_floats(_native.GEMM_TN, N, K, s))), device=dev)
            for v in vs:
                fns[f"v{v}s{s}"] = (lambda v=v, s=s, ws=ws: G.gemm(_native.GEMM_TN, _native.EPI_F32, N, K, T, dy, N, x, K, dw, K,
                                                                  workspace=ws, colsum=db, splits=s, variant=v))
        t = timeit(fns, args.rounds)
        row["wgrad_us"] = {k: round(v, 2) for k, v in t.items()}
        row["default_splits"] = int(lib.gps_gemm_pick_splits(_native.GEMM_TN, N, K, T))
        best = {k: min((v for n, v in row[k].items() if n != "lib")) for k in ("fwd_us", "dgrad_us", "wgrad_us")}
        row["best_TFLOPs"] = {k: round(flops / best[k] / 1e6, 1) for k in best}
        row["lib_TFLOPs"] = {k: round(flops / row[k]["lib"] / 1e6, 1) for k in best}
        print(json.dumps(row), flush=True)
        out.append(row)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
