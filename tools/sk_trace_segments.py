"""Per-segment timeline of the stream-K GEMM from a raw `gemm_probe trace ... FILE` dump: for every segment kind
(0 whole tile, 1 contribution, 2 owner of a split tile) the median time from segment start to accumulators ready (slab
reads of an owner), the main loop per K tile, and the ending (epilogue / slab store + flag)."""
import struct
import sys
from collections import defaultdict

SLOTS = 96
raw = open(sys.argv[1], "rb").read()
n = len(raw) // (8 * SLOTS)
# shader clock from the two stamp kinds (s_memtime cycles against s_memrealtime's 100 MHz)
_r = []
for w in range(n):
    t = struct.unpack_from(f"{SLOTS}Q", raw, w * 8 * SLOTS)
    if t[0] and t[5] > t[4] + 500:
        _r.append((t[6] - t[0]) / ((t[5] - t[4]) / 100.0))
mhz = sorted(_r)[len(_r) // 2] if _r else (float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0)
acc = defaultdict(lambda: defaultdict(list))
tot = []
for w in range(n):
    t = struct.unpack_from(f"{SLOTS}Q", raw, w * 8 * SLOTS)
    if t[0] == 0:
        continue
    tot.append((t[6] - t[0]) / mhz)
    for sgi in range(4):
        b = 8 + 6 * sgi
        if t[b] == 0 or t[b + 3] == 0:
            continue
        kind, nst = t[b + 5], t[b + 4]
        acc[kind]["ready"].append((t[b + 1] - t[b + 2]) / mhz if t[b + 1] else 0.0)     # owner: flags seen, measured from main loop done
        acc[kind]["loop"].append((t[b + 2] - t[b]) / mhz)
        acc[kind]["per_kt"].append((t[b + 2] - t[b]) / mhz / max(nst, 1))
        acc[kind]["ending"].append((t[b + 3] - t[b + 2]) / mhz)
        acc[kind]["nst"].append(nst)
med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
print(f"workgroups {len(tot)}  total us median {med(tot):.1f} max {max(tot):.1f}  ({mhz:.0f} MHz)")
for kind, name in ((1, "contribution"), (0, "whole tile"), (2, "owner")):
    a = acc[kind]
    if not a:
        continue
    print(f"  {name:13s} n={len(a['loop']):4d}  K tiles med {med(a['nst']):.0f}  ready {med(a['ready']):6.2f}  loop {med(a['loop']):6.2f} ({med(a['per_kt']):.2f}/K tile, "
          f"fixed part ~{med([l - 1.31 * k for l, k in zip(a['loop'], a['nst'])]):.2f})  ending {med(a['ending']):6.2f} (p90 {sorted(a['ending'])[int(0.9 * (len(a['ending']) - 1))]:.2f})")
