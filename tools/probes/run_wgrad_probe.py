"""Build and run tools/probes/wgrad_gemm_probe.hip on a GPU box: correctness against an fp32 reference of the
same bf16 operands (M != N shapes, so a transposed result cannot pass) and time per call (graph replay)
next to torch.mm and the library split-K bmm for the weight-gradient shapes of the GPS step.

    python tools/probes/run_wgrad_probe.py
"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    so = os.path.join(HERE, "libwgrad_probe.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(HERE, "wgrad_gemm_probe.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.wgrad_probe_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 5
    from sceneverse_amd.common.wgrad_splitk import pick_splits, splitk_wgrad_mm
    dev = "cuda"
    shapes = [(19200, 768, 768), (19200, 2304, 768), (19200, 3072, 768), (19200, 768, 3072),
              (8320, 768, 768), (8320, 2048, 768), (8320, 768, 2048), (5120, 768, 768), (5120, 2376, 768),
              (200, 136, 264)]                                    # ragged tile edges
    for R, M, N in shapes:
        dy = torch.randn(R, M, device=dev).to(torch.bfloat16)
        x = torch.randn(R, N, device=dev).to(torch.bfloat16)
        ref = dy.float().t() @ x.float()
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        for splits in sorted({1, max(1, min(32, 512 // tiles)), max(1, min(32, 1024 // tiles))}):
            ws = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
            out = torch.empty((M, N), dtype=torch.float32, device=dev)

            def run():
                rc = lib.wgrad_probe_launch(R, M, N, splits, dy.data_ptr(), x.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            run()
            torch.cuda.synchronize()
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            us = timeit(run) if R >= 1000 else float("nan")
            print(f"R={R} M={M} N={N} splits={splits:2d}  rel_err={err:.2e}  {us:8.1f} us  "
                  f"{2 * R * M * N / us / 1e6 if us == us else 0:7.0f} TFLOP/s", flush=True)
        if R >= 1000:
            s_lib = pick_splits(R, M, N)
            print(f"    torch.mm {timeit(lambda: torch.mm(dy.t(), x)):8.1f} us   library split-K (S={s_lib}) "
                  f"{timeit(lambda: splitk_wgrad_mm(dy, x, s_lib)):8.1f} us", flush=True)


if __name__ == "__main__":
    main()
