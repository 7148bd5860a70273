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
