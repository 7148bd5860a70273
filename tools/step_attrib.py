"""Where does the glue go?  Runs eager GPS pre-train steps under torch.profiler (with Python stacks) and prints device
time per (kernel name, innermost sceneverse_amd source line of the launching op), for the kernels that are NOT
libgps_hip.so launches unless --all is given.

    python tools/step_attrib.py [--all] [--steps 2] [--out FILE]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--config", default="pretrain")
    args = ap.parse_args()
    import bench
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    from torch.profiler import ProfilerActivity, profile

    preset = bench.WORKLOADS[args.config]
    dev = torch.device("cuda", 0)
    cfg = bench.gps_pretrain_cfg(bench._lang_dir(), num_gpu=1, workload=args.config)
    step = GPSTrainStep(cfg, device=dev, amp_dtype=torch.bfloat16, graph=False)
    batch = synth_batch(preset["batch"], n_obj=preset["n_obj"], n_pts=preset["n_pts"], txt_len=preset["txt_len"],
                        seed=42, device=dev)
    if not preset["scene_cap"]:
        batch.pop("scene_txt_ids"), batch.pop("scene_txt_masks")
    for _ in range(3):
        step.step(dict(batch))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(args.steps):
            step.step(dict(batch))
        torch.cuda.synchronize()
    out = open(args.out, "w") if args.out else sys.stdout
    try:
        print(prof.key_averages(group_by_input_shape=True, group_by_stack_n=5).table(
            sort_by="self_device_time_total", row_limit=120, max_name_column_width=60, max_src_column_width=110,
            max_shapes_column_width=60), file=out)
        out.flush()
    except Exception as e:  # noqa: BLE001
        print(f"built-in table failed: {e}", file=out)
    path = "/tmp/step_trace.json"
    prof.export_chrome_trace(path)
    with open(path) as f:
        tr = json.load(f)
    evs = tr["traceEvents"]
    # kernels by correlation id -> the cpu op that launched them (through the runtime launch event)
    launches = {}      # correlation -> (ts, tid) of the hipLaunchKernel / hipMemcpyAsync call
    for e in evs:
        if e.get("cat") in ("cuda_runtime", "cuda_driver") and "args" in e and "correlation" in e["args"]:
            launches[e["args"]["correlation"]] = (e["ts"], e["tid"], e["pid"])
    cpu_ops = [e for e in evs if e.get("cat") in ("cpu_op", "user_annotation", "python_function") and "dur" in e]
    by_thread = collections.defaultdict(list)
    for e in cpu_ops:
        by_thread[(e["pid"], e["tid"])].append(e)
    for v in by_thread.values():
        v.sort(key=lambda e: e["ts"])

    # sweep per thread: stack of the cpu events open at each launch timestamp
    per_thread_launch = collections.defaultdict(list)
    for corr, (ts, tid, pid) in launches.items():
        per_thread_launch[(pid, tid)].append((ts, corr))
    owners = {}
    for key, ls in per_thread_launch.items():
        ls.sort()
        events = sorted(by_thread.get(key, ()), key=lambda e: (e["ts"], -e["dur"]))
        stack, i = [], 0
        for ts, corr in ls:
            while i < len(events) and events[i]["ts"] <= ts:
                e = events[i]
                while stack and stack[-1]["ts"] + stack[-1]["dur"] < e["ts"]:
                    stack.pop()
                stack.append(e)
                i += 1
            while stack and stack[-1]["ts"] + stack[-1]["dur"] < ts:
                stack.pop()
            best_py, best_op, best_node = None, None, None
            for e in stack:
                if e["ts"] + e["dur"] < ts:
                    continue
                if e["cat"] == "python_function":
                    n = e["name"]
                    if "sceneverse_amd" in n or "bench.py" in n:
                        best_py = n
                elif e["cat"] == "cpu_op":
                    best_op = e
                    if e["name"].startswith("autograd::engine::evaluate_function"):
                        best_node = e["name"].split(":")[-1].strip()
            owners[corr] = (best_py, best_op, best_node)

    agg = collections.defaultdict(lambda: [0.0, 0])
    for e in evs:
        if e.get("cat") not in ("kernel", "gpu_memcpy", "gpu_memset"):
            continue
        name = e["name"]
        if not args.all and "gps" in name:
            continue
        corr = e.get("args", {}).get("correlation")
        py, op, node = (None, None, None)
        if corr in owners:
            py, op, node = owners[corr]
        if py is None and node is not None:
            py = "backward of " + node
        shapes = ""
        opname = ""
        if op is not None:
            opname = op["name"]
            shapes = str(op.get("args", {}).get("Input Dims", ""))[:80]
        short = name
        m = __import__("re").search(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+(?:<[^,>]*)?)", name.split("(")[0] if name.startswith("void at::native::vectorized") is False else name[40:])
        if "at::native::" in name:
            inner = name.split("at::native::")
            short = "..." + "at::native::".join(inner[1:])[:66] if len(inner) > 2 else name[:70]
        key = (short[:70], opname[:40], shapes, (py or "?")[-80:])
        agg[key][0] += e["dur"]
        agg[key][1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    total = sum(v[0] for v in agg.values())
    print(f"# total device time of the listed kernels: {total / 1e3 / args.steps:.3f} ms/step", file=out)
    for (name, opname, shapes, py), (t, n) in rows[:400]:
        print(f"{t / 1e3 / args.steps:8.4f} ms {n / args.steps:6.1f}x  {name:70s} | {opname:40s} | {shapes:80s} | {py}", file=out)


if __name__ == "__main__":
    main()
