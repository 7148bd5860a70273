"""Table of tools/probes/gemm_probe `bench` outputs: per shape the time of every variant (us, TFLOP/s, fraction of the
2.5 PFLOP/s bf16 MFMA peak), cold and warm side by side.   python tools/gemm_probe_table.py cold.json [warm.json]"""
import json
import sys


def load(p):
    with open(p) as f:
        return json.load(f)


def main():
    runs = [load(p) for p in sys.argv[1:]]
    first = runs[0]
    vs = [k for k in first["rows"][0] if k.startswith("v")]
    hdr = f"{'shape':44s}" + "".join(f"{r['mode'] + ' ' + v:>16s}" for r in runs for v in vs)
    print(hdr)
    tot = {(ri, v): 0.0 for ri in range(len(runs)) for v in vs}
    flops = 0.0
    for i, row in enumerate(first["rows"]):
        name = f"{row['form']} e{row['epi']} {row['M']}x{row['N']}x{row['K']} ({row['what']}) d{row['default_variant']}"
        line = f"{name:44s}"
        for ri, r in enumerate(runs):
            rr = r["rows"][i]
            for v in vs:
                c = rr[v]
                bad = "!" if c["diff_words"] else " "
                line += f"{c['us']:8.1f}{bad}{c['TF'] / 2500:6.3f} "
                tot[(ri, v)] += c["us"]
        flops += 2.0 * row["M"] * row["N"] * row["K"]
        print(line)
    line = f"{'sum us / population frac':44s}"
    for ri in range(len(runs)):
        for v in vs:
            line += f"{tot[(ri, v)]:8.1f} {flops / tot[(ri, v)] * 1e-6 / 2500:6.3f} "
    print(line)


if __name__ == "__main__":
    main()
