"""Turn the two rocprofv3 --pmc passes of tools/pmc_workload.py into HBM bytes per launch.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json

Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB-like units derived from
the L2's fabric request counters and are NOT byte-exact on gfx950 (FETCH_SIZE = 1/2 of a wide coalesced
read; WRITE_SIZE uncalibrated) -> both are calibrated on the 512 MiB device copy at the head of the
workload (known bytes / counter), and the factors are recorded in the output."""
import collections
import csv
import json
import re
import sys

CAL_BYTES = 512 << 20

# HIP symbol -> the kernel names bench.py reports
NAMES = [
    (r"sa_mlp_x3_kernel<128, 128, 128, 256,", "sa_mlp_forward(c=128,n=32,np=16,mlp=128-128-256,bf16x3)"),
    (r"sa_mlp_x3_kernel<3, 64, 64, 128,", "sa_mlp_forward(c=3,n=1024,np=32,mlp=64-64-128,bf16x3)"),
    (r"sa_mlp_kernel<128, 128, 128, 256>", "sa_mlp_forward(c=128,n=32,np=16,mlp=128-128-256,fp32)"),
    (r"sa_mlp_kernel<3, 64, 64, 128>", "sa_mlp_forward(c=3,n=1024,np=32,mlp=64-64-128,fp32)"),
    (r"fps_resident_kernel<16>", "furthest_point_sampling(n=1024,m=32)"),
    (r"fps_resident_kernel<1>", "furthest_point_sampling(n=32,m=16)"),
    (r"ball_query_kernel<16", "ball_query(n=1024,m=32,ns=32)"),
    (r"ball_query_kernel<1,", "ball_query(n=32,m=16,ns=32)"),
    (r"wgrad_grouped_kernel", "gemm_tn_grouped(problems=69)"),
    (r"ball_query_small_kernel", "ball_query(n=32,m=16,ns=32)"),
    (r"add_dropout_ln_bwd_kernel", "add_dropout_layernorm_backward"),
    (r"add_dropout_ln_fwd_kernel", "add_dropout_layernorm_forward"),
    (r"attn_bwd_stream_kernel", "attn_backward(L<=300,varlen,seqs=128)"),
    (r"attn_fwd_stream_kernel", "attn_forward(L<=300,varlen,seqs=128)"),
    (r"gemm8p_kernel<false, true, 0>", "gemm_nn(M=22400,N=768,K=3072,epi=0)"),          # two-group 256 x 256 form [r3]
    (r"gemm8p_kernel<true, true, 5>", "gemm_tn(M=3072,N=768,K=22400,epi=5)"),
    (r"gemm_kernel<128, 128, 4, 2, false, false, 1,", "gemm_nt(M=22400,N=3072,K=768,epi=1)"),
    (r"gemm_kernel<128, 128, 4, 2, false, true, 0,", "gemm_nn(M=22400,N=768,K=3072,epi=0)"),
    (r"gemm_kernel<128, 128, 4, 2, true, true, 5,", "gemm_tn(M=3072,N=768,K=22400,epi=5)"),
    (r"adamw_kernel", "adamw_step(pmc workload: 25 165 824 parameters)"),
    (r"attn_bwd_kernel", "attn_backward"),
    (r"attn_fwd_kernel", "attn_forward"),
    (r"colsum_stage1_kernel", "colsum_bf16_stage1"),
    (r"obj_processing_post_kernel", "obj_processing_post"),
    (r"group_points_kernel", "group_points"),
    (r"gather_points_kernel", "gather_points"),
]


def per_kernel(path, counter):
    """kernel symbol [+ grid size for the shape-polymorphic kernels] -> counter values per dispatch"""
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            if "add_dropout_ln_" in k or "gps_attn::" in k or "gps_red::" in k:
                k = f"{k} grid={r.get('Grid_Size', '?')}"
            acc[k].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out):
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    cal = [k for k in fetch if "copy" in k.lower() and max(fetch[k]) > 0]
    cal_k = max(cal, key=lambda k: max(fetch[k]))
    # only the 512 MiB calibration copies: other (tiny) launches of the same copy kernel are ignored
    big_f = [v for v in fetch[cal_k] if v >= 0.5 * max(fetch[cal_k])]
    big_w = [v for v in write[cal_k] if v >= 0.5 * max(write[cal_k])]
    f_unit = CAL_BYTES / (sum(big_f) / len(big_f))
    w_unit = CAL_BYTES / (sum(big_w) / len(big_w))
    res = {"calibration": {"kernel": cal_k[:80], "known_bytes_each_way": CAL_BYTES,
                           "bytes_per_FETCH_SIZE_unit": f_unit, "bytes_per_WRITE_SIZE_unit": w_unit},
           "per_launch_hbm_bytes": {}, "per_launch_detail": {}}
    for k in fetch:
        for pat, name in NAMES:
            if re.search(re.escape(pat), k):
                fb = f_unit * sum(fetch[k]) / len(fetch[k])
                wb = w_unit * sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
                if name in ("group_points", "gather_points") or (" grid=" in k and "stream" not in k):
                    name = f"{name}#{len(res['per_launch_detail'])}" + (k[k.rfind(" grid="):] if " grid=" in k else "")
                if name in res["per_launch_hbm_bytes"] and " grid=" in k:      # the same symbol at another shape
                    # keep the launch with MORE workgroups under the plain name (the step's dominant shape)
                    prev_grid = res["per_launch_detail"][name].get("grid", 0)
                    this_grid = int(k[k.rfind(" grid=") + 6:] or 0)
                    if this_grid <= prev_grid:
                        name = name + k[k.rfind(" grid="):]
                    else:
                        old_name = name + f" grid={prev_grid}"
                        res["per_launch_hbm_bytes"][old_name] = res["per_launch_hbm_bytes"].pop(name)
                        res["per_launch_detail"][old_name] = res["per_launch_detail"].pop(name)
                res["per_launch_hbm_bytes"][name] = int(fb + wb)
                res["per_launch_detail"][name] = {"symbol": k[:100], "read_bytes": int(fb), "write_bytes": int(wb),
                                                  "launches": len(fetch[k]),
                                                  "grid": int(k[k.rfind(" grid=") + 6:]) if " grid=" in k else 0}
                break
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
