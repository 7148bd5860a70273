"""Compile every HIP source of libgps_hip.so to gfx950 assembly and list, per kernel, what the source does not show:
VGPRs, scratch bytes and scratch instructions (register arrays indexed at run time, spills), v_readfirstlane counts
(VGPR-resident descriptors -> waterfall loops around buffer loads), full-drain waits (`s_waitcnt vmcnt(0)`) and IEEE
division sequences (`v_div_scale_f32`: `1.f / x` and `__frcp_rn` are ~10 instructions, `__builtin_amdgcn_rcpf` is one).

    python tools/asm_audit.py [--out profiles/r4/asm_audit.txt] [file.hip ...]

Round 4 found three kernels losing 15 - 50 % this way; no GPU needed."""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except OSError:
        return names


def audit(path, tmp):
    asm = os.path.join(tmp, os.path.basename(path) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", "-S",
                    "--cuda-device-only", "-o", asm, path], check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n\s+- \.agpr_count|\n\s+- \.args|\Z)", text, re.S):
        blk = m.group(2)
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) if re.search(rf"\.{k}:\s+(\d+)", blk) else -1
        meta[m.group(1)] = (g("vgpr_count"), g("private_segment_fixed_size"), g("sgpr_spill_count"), g("vgpr_spill_count"))
    rows = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in meta:
            continue
        rows.append((name, *meta[name], len(re.findall(r"\bscratch_", body)), len(re.findall(r"v_readfirstlane", body)),
                     len(re.findall(r"s_waitcnt vmcnt\(0\)", body)), len(re.findall(r"v_mfma", body)),
                     len(re.findall(r"v_div_scale_f32", body)) // 2))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(ROOT, "sceneverse_amd", "csrc", "*.hip")))
    lines = [f"{'vgpr':>5} {'scr_B':>6} {'s_sp':>4} {'v_sp':>4} {'scr_i':>5} {'rfl':>4} {'vm0':>4} {'mfma':>5} {'div':>4}  kernel"]
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            rows = audit(f, tmp)
            names = demangle([r[0] for r in rows])
            lines.append(f"-- {os.path.relpath(f, ROOT)}")
            for r, n in zip(rows, names):
                lines.append(f"{r[1]:5d} {r[2]:6d} {r[3]:4d} {r[4]:4d} {r[5]:5d} {r[6]:4d} {r[7]:4d} {r[8]:5d} {r[9]:4d}  {n[:150]}")
    out = "\n".join(lines) + "\n"
    sys.stdout.write(out)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(out)


if __name__ == "__main__":
    main()
