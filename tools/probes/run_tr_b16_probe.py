"""Run tools/probes/tr_b16_probe.hip for a few address patterns and print what every lane received.

    python tools/probes/run_tr_b16_probe.py            (on a GPU box; compiles the probe with hipcc)

Patterns (byte offsets per lane l, 64 lanes):
  linear8      l * 8                         -- the guide's "lane l, elem j reads lds[(l&15) + j*16 + (l>>4)*64]"
  rows_p72     a [k][m] bf16 tile with a 72-element row pitch: 16-lane group g reads k rows 8g .. 8g+3,
               lane t of the group supplies &tile[8g + t/4][4 * (t & 3)]  (the A-fragment recipe planned for
               the weight-gradient GEMM: expected result lane t <- tile[8g + 0..3][t])
Findings go into DESIGN.md before the kernel is written."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    so = os.path.join(HERE, "tr_b16_probe.hsaco")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "--genco",
                           os.path.join(HERE, "tr_b16_probe.hip"), "-o", so])
    hip = ctypes.CDLL("libamdhip64.so")
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), so.encode()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"tr_b16_probe") == 0
    n_slots = 8192
    pitch = 72
    pats = {
        "linear8": [l * 8 for l in range(64)],
        "rows_p72": [2 * ((8 * (l >> 4) + (l & 15) // 4) * pitch + 4 * (l & 3)) for l in range(64)],
    }
    for name, offs in pats.items():
        off_t = torch.tensor(offs, dtype=torch.int32, device="cuda")
        out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
        args = (ctypes.c_void_p * 3)(
            ctypes.cast(ctypes.pointer(ctypes.c_void_p(off_t.data_ptr())), ctypes.c_void_p),
            ctypes.cast(ctypes.pointer(ctypes.c_void_p(out.data_ptr())), ctypes.c_void_p),
            ctypes.cast(ctypes.pointer(ctypes.c_int(n_slots)), ctypes.c_void_p))
        rc = hip.hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, n_slots * 2, None, args, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        got = out.cpu().view(64, 4).tolist()
        print(f"== {name}")
        for l in range(64):
            slots = got[l]
            if name == "rows_p72":
                slots = [(s // pitch, s % pitch) for s in slots]     # (k row, m column)
            print(f"lane {l:2d} off {offs[l]:5d} -> {slots}")


if __name__ == "__main__":
    sys.exit(main())
