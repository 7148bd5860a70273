"""Per-kernel timing of libgps_hip.so at the GPS workload shapes (B=64 scenes x 80 objects),
with the reference's own kernels (oracle/_ref, if built) timed beside them for context.
Prints one line per op: avg us, algorithmic GB/s, fraction of the 8 TB/s HBM roofline."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sceneverse_amd.data.synthetic import synth_batch  # noqa: E402
from sceneverse_amd.pointnet2 import _ext as hip  # noqa: E402


def timeit(fn, iters=20, warm=3):
    """Average GPU time of one call in us.  The `iters` calls are captured into ONE HIP graph and the
    replay is timed, so that kernels shorter than the Python/ctypes launch path (~12 us per call) are
    measured by their GPU time, not by the host's launch rate; eager back-to-back timing is the fallback
    when an op cannot be captured."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / iters
    except Exception:  # noqa: BLE001 -- not capturable: time eager launches
        torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = "cuda"
    d = synth_batch(args.batch, seed=42)
    pcs = d["obj_fts"].reshape(-1, 1024, 6).to(dev)
    xyz = pcs[..., :3].contiguous()
    rgb = pcs[..., 3:].transpose(1, 2).contiguous()
    b = xyz.shape[0]
    ref = None
    try:
        from oracle import build_ref
        if build_ref.built_path():
            ref = build_ref.load_ext()
    except Exception as ex:  # noqa: BLE001
        print("reference ext unavailable:", ex)

    xyz_t = xyz.transpose(1, 2).contiguous()
    fps = hip.furthest_point_sampling(xyz, 32)
    new_xyz = hip.gather_points(xyz_t, fps).transpose(1, 2).contiguous()
    idx = hip.ball_query(new_xyz, xyz, 0.2, 32)
    fps2 = hip.furthest_point_sampling(new_xyz, 16)
    nx2 = hip.gather_points(new_xyz.transpose(1, 2).contiguous(), fps2).transpose(1, 2).contiguous()
    idx2 = hip.ball_query(nx2, new_xyz, 0.4, 32)
    feats = torch.randn(b, 128, 32, device=dev)
    gout = torch.randn(b, 128, 16, 32, device=dev)
    nx_t = new_xyz.transpose(1, 2).contiguous()

    ops = [
        ("fps SA1 (n=1024,m=32)", lambda m: m.furthest_point_sampling(xyz, 32), b * (1024 * 12 + 32 * 4)),
        ("fps SA2 (n=32,m=16)", lambda m: m.furthest_point_sampling(new_xyz, 16), b * (32 * 12 + 16 * 4)),
        # a gather reads the m gathered columns of each channel + the m indices and writes (c, m): 896 B per
        # object -- NOT the whole (c, n) source row block (charging that gave frac > 1 in profiles/r3/kernel_bench_f.log)
        ("gather SA1 (c=3)", lambda m: m.gather_points(xyz_t, fps), b * (3 * 32 * 4 + 32 * 4 + 3 * 32 * 4)),
        ("ball_query SA1", lambda m: m.ball_query(new_xyz, xyz, 0.2, 32), b * ((1024 + 32) * 12 + 32 * 32 * 4)),
        ("ball_query SA2", lambda m: m.ball_query(nx2, new_xyz, 0.4, 32), b * ((32 + 16) * 12 + 16 * 32 * 4)),
        ("group SA1 xyz (c=3,n=1024)", lambda m: m.group_points(xyz_t, idx), b * (3 * 1024 * 4 + 1024 * 4 + 3 * 1024 * 4)),
        ("group SA1 rgb (c=3,n=1024)", lambda m: m.group_points(rgb, idx), b * (3 * 1024 * 4 + 1024 * 4 + 3 * 1024 * 4)),
        ("group SA2 xyz (c=3,n=32)", lambda m: m.group_points(nx_t, idx2), b * (3 * 32 * 4 + 512 * 4 + 3 * 512 * 4)),
        ("group SA2 feats (c=128,n=32)", lambda m: m.group_points(feats, idx2), b * (128 * 32 * 4 + 512 * 4 + 128 * 512 * 4)),
        ("group_grad SA2 feats", lambda m: m.group_points_grad(gout, idx2, 32), b * (128 * 512 * 4 + 512 * 4 + 128 * 32 * 4)),
    ]
    # fused frozen SA levels (gps_sa_mlp_forward); algorithmic bytes = the two group_points calls
    # each launch absorbs (unfused reference API), FLOPs = the three 1x1 convs
    from sceneverse_amd.pointnet2 import pointnet2_modules as M
    torch.manual_seed(0)

    def _packed_pair(cin, chans):
        ws, ss, c = [], [], cin
        for co in chans:
            ws.append(torch.randn(co, c, device=dev) * (2.0 / c) ** 0.5)
            ss.append(torch.randn(co, device=dev) * 0.05)
            c = co
        return {p: hip.sa_mlp_pack(ws, ss, p) for p in ("fp32", "bf16x3")}

    wp1, wp2 = _packed_pair(6, [64, 64, 128]), _packed_pair(131, [128, 128, 256])
    f1 = hip.sa_mlp_forward(xyz, new_xyz, rgb, idx, wp1["fp32"], [64, 64, 128])
    by1 = 2 * b * (3 * 1024 * 4 + 1024 * 4 + 3 * 1024 * 4)
    by2 = b * (3 * 32 * 4 + 512 * 4 + 3 * 512 * 4) + b * (128 * 32 * 4 + 512 * 4 + 128 * 512 * 4)
    fl1 = 2 * b * 1024 * (6 * 64 + 64 * 64 + 64 * 128)
    fl2 = 2 * b * 512 * (131 * 128 + 128 * 128 + 128 * 256)
    fused = []
    for prec in ("fp32", "bf16x3"):
        fused.append((f"sa_mlp SA1 fused (6-64-64-128) {prec}",
                      lambda prec=prec: hip.sa_mlp_forward(xyz, new_xyz, rgb, idx, wp1[prec], [64, 64, 128], prec),
                      by1, fl1))
        fused.append((f"sa_mlp SA2 fused (131-128-128-256) {prec}",
                      lambda prec=prec: hip.sa_mlp_forward(new_xyz, nx2, f1, idx2, wp2[prec], [128, 128, 256], prec),
                      by2, fl2))
    rows = []
    for name, fn, nbytes, flops in fused:
        us = timeit(fn)
        row = {"op": name, "us": round(us, 2), "algorithmic_bytes": nbytes, "GBps": round(nbytes / us / 1e3, 1),
               "frac_8TBps": round(nbytes / us / 1e3 / 8000, 4), "TFLOPs_fp32": round(flops / us / 1e6, 1),
               "frac_fp32_mfma_157TF": round(flops / us / 1e6 / 157.3, 4)}
        rows.append(row)
        print(row, flush=True)
    total_bq_group_us = 0.0
    total_bq_group_bytes = 0
    for name, fn, nbytes in ops:
        us = timeit(lambda: fn(hip))
        row = {"op": name, "us": round(us, 2), "algorithmic_bytes": nbytes,
               "GBps": round(nbytes / us / 1e3, 1), "frac_8TBps": round(nbytes / us / 1e3 / 8000, 4)}
        if ref is not None:
            row["reference_kernel_us"] = round(timeit(lambda: fn(ref)), 2)
        rows.append(row)
        if name.startswith(("ball_query", "group SA")):
            total_bq_group_us += us
            total_bq_group_bytes += nbytes
        print(row, flush=True)
    agg = {"op": "ball_query+group (SA1+SA2, 6 launches)", "us": round(total_bq_group_us, 2),
           "algorithmic_bytes": total_bq_group_bytes,
           "GBps": round(total_bq_group_bytes / total_bq_group_us / 1e3, 1),
           "frac_8TBps": round(total_bq_group_bytes / total_bq_group_us / 1e3 / 8000, 4)}
    print(agg)
    rows.append(agg)
    # loader-side object processing (gps_obj_processing_post): 64 scenes x 80 slots x 1024 points from
    # HBM-resident raw scans (ScanNet-like object sizes: log-uniform 50..20000 points)
    import numpy as np
    from sceneverse_amd.data import gpu_objects as G
    rng = np.random.default_rng(0)
    packed = G.PackedScans(dev)
    for s in range(16):
        n = int(rng.integers(20, 80))
        ks = np.exp(rng.uniform(np.log(50), np.log(20000), size=n)).astype(np.int64)
        pts = rng.normal(size=(int(ks.sum()), 3)).astype(np.float32)
        col = rng.integers(0, 256, size=(int(ks.sum()), 3), dtype=np.uint8)
        packed.add_scan(f"s{s}", pts, col, np.repeat(np.arange(n), ks), list(range(n)))
    packed.finalize()
    slots = G.batch_rows(packed, [f"s{i % 16}" for i in range(args.batch)], 80)
    nbytes = G._algorithmic_bytes(packed, slots, slots.numel(), 1024)
    slots_d = slots.to(dev)
    us = timeit(lambda: G.obj_processing_post(packed, slots_d, 1024, seed=1))
    row = {"op": f"obj_processing_post ({args.batch}x80 slots, 1024 pts, device sampler)", "us": round(us, 2),
           "algorithmic_bytes": nbytes, "GBps": round(nbytes / us / 1e3, 1),
           "frac_8TBps": round(nbytes / us / 1e3 / 8000, 4),
           "raw_points_per_batch": int(packed.sizes_host[slots.reshape(-1).numpy()[slots.reshape(-1).numpy() >= 0]].sum())}
    print(row, flush=True)
    rows.append(row)
    # bias-gradient column sums (gps_colsum_bf16) next to torch's reduction of the same matrix
    from sceneverse_amd.common import colsum as WS
    for T, N in ((19200, 3072), (19200, 768), (8320, 2048), (5120, 2376)):
        dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
        us = timeit(lambda: WS.colsum_bf16(dy))
        us_t = timeit(lambda: dy.sum(0))
        nb = 2 * T * N + 4 * N
        row = {"op": f"colsum_bf16 ({T}x{N})", "us": round(us, 2), "algorithmic_bytes": nb,
               "GBps": round(nb / us / 1e3, 1), "frac_8TBps": round(nb / us / 1e3 / 8000, 4),
               "torch_sum_us": round(us_t, 2)}
        print(row, flush=True)
        rows.append(row)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"batch": args.batch, "objects": b, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
