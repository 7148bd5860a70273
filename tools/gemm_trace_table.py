"""One line per `gemm_probe trace` record (tools/gpu_r6_trace.sh): where a workgroup's time goes."""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if not line:
        continue
    d = json.loads(line)
    t = d["trace"]
    print(f"{d['form']} e{d['epi']} {d['M']}x{d['N']}x{d['K']} v{d['variant']}: wall {t['wall_us']:.1f} us, {t['workgroups_traced']} workgroups, {t['shader_MHz']:.0f} MHz")
    print("    us", {k: round(v, 2) for k, v in t["us"].items()})
    print("    start", t["start_us"], " end", t["end_us"])
    for k in ("phase_cycles_wave0", "phase_cycles_wave4"):
        if k in t:
            v = t[k]
            names = ["reads", "barrier+lgkm", "mfma", "barrier"]
            print(f"    {k}: " + " | ".join(f"ph{p + 1} " + " ".join(f"{names[i]} {v[4 * p + i]:.0f}" for i in range(4)) for p in range(4)) + f"  (sum {sum(v):.0f})")
