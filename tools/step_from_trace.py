"""One steady-state step out of a rocprofv3 --kernel-trace CSV of `bench.py`: the dispatches between the last two
launches of the grouped weight-gradient kernel (one per step) = one whole step period of the replayed graph, start-up
work (parameter copies, optimizer-state fills, warm-up, capture) excluded -- which a whole-run `--stats` table divided
by the step count is not.

    python tools/step_from_trace.py bench_kernel_trace.csv [--json out.json] [--marker wgrad_grouped_kernel]

Prints device time and launch count per kernel, split into libgps_hip.so kernels and everything else."""
import argparse
import collections
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--json", default=None)
    ap.add_argument("--marker", default="wgrad_grouped_kernel")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--index", type=int, default=None,
                    help="which inter-marker window (0-based); default: the middle one of the longest run of windows with "
                         "the same dispatch count (= the timed graph replays; bench.py's own eager timing passes at the "
                         "end of the run launch more)")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < 3:
        raise SystemExit(f"fewer than 3 '{a.marker}' dispatches in the trace")
    counts = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
    if a.index is None:
        best, run_lo, run_len, i = (0, 0), 0, 0, 0
        while i < len(counts):
            j = i
            while j + 1 < len(counts) and counts[j + 1] == counts[i]:
                j += 1
            if j - i + 1 > best[1]:
                best = (i, j - i + 1)
            i = j + 1
        a.index = best[0] + best[1] // 2
    print("dispatches per inter-marker window:", counts, "-> window", a.index)
    lo, hi = marks[a.index], marks[a.index + 1]         # [marker of step k, marker of step k + 1): one step period
    win = rows[lo:hi]
    period_us = (rows[hi][0] - rows[lo][0]) / 1e3
    agg = collections.OrderedDict()
    for s, e, n in win:
        t = agg.setdefault(n, [0.0, 0])
        t[0] += (e - s) / 1e3
        t[1] += 1
    native = lambda n: any(tag in n for tag in ("gps_", "gps::", "_ZN3gps"))  # noqa: E731
    lib = {n: v for n, v in agg.items() if native(n)}
    oth = {n: v for n, v in agg.items() if not native(n)}
    summ = lambda d: (round(sum(v[0] for v in d.values()), 1), sum(v[1] for v in d.values()))  # noqa: E731

    def family(n):
        if "rocclr" in n:
            return "rocclr copy / fill"
        if n.startswith("Cijk_"):
            return "hipBLASLt"
        return "torch"
    fam = collections.defaultdict(lambda: [0.0, 0])
    for n, v in oth.items():
        fam[family(n)][0] += v[0]
        fam[family(n)][1] += v[1]
    out = {"step_period_us": round(period_us, 1), "kernel_time_us": summ(agg)[0], "launches": summ(agg)[1],
           "libgps_hip": {"us": summ(lib)[0], "launches": summ(lib)[1]},
           "outside": {"us": summ(oth)[0], "launches": summ(oth)[1],
                       "by_family": {k: {"us": round(v[0], 1), "launches": v[1]} for k, v in fam.items()}},
           "top_outside": [{"us": round(v[0], 1), "launches": v[1], "name": n[:140]}
                           for n, v in sorted(oth.items(), key=lambda kv: -kv[1][0])[:a.top]],
           "top_libgps": [{"us": round(v[0], 1), "launches": v[1], "name": n[:140]}
                          for n, v in sorted(lib.items(), key=lambda kv: -kv[1][0])[:a.top]]}
    print(json.dumps({k: out[k] for k in ("step_period_us", "kernel_time_us", "launches", "libgps_hip", "outside")}, indent=1))
    for r in out["top_outside"]:
        print(f"{r['us']:8.1f} us {r['launches']:4d}  {r['name'][:120]}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
