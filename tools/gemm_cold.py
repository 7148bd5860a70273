"""Why are the GEMMs of the step 15 - 35 % slower than the same launches in tools/gemm_bench.py?  Three timings of one
forward GEMM (tokens x 768 -> 3072, bias + GELU epilogue) and one input-gradient GEMM:
  hot      back-to-back launches on the same operands (what the micro-benchmark measures),
  cold     every launch preceded by a 1 GB fill (operands and weights evicted from L2 / MALL), the GEMM alone timed,
  sustained  2 000 back-to-back launches (~150 ms): the average of the last 200 against the first 200 (clock / power).
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sceneverse_amd.modules.layers import gemm as G  # noqa: E402

dev = "cuda"
T, K, N = 12608, 768, 3072
x = torch.randn(T, K, device=dev).to(torch.bfloat16)
w = (0.05 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
b = torch.randn(N, device=dev)
dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GB


def fwd():
    return G.linear_forward(x, w, b, act="gelu", want_pre="factor")


def dgrad():
    return G.linear_dgrad(dy, w)


def timed(fn, n, flush=False):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        if flush:
            big.fill_(1.0)
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return [s.elapsed_time(e) * 1e3 for s, e in ev]


out = {}
for name, fn in (("fwd_gelu", fwd), ("dgrad", dgrad)):
    for _ in range(5):
        fn()
    hot = timed(fn, 50)
    cold = timed(fn, 20, flush=True)
    long = timed(fn, 2000)
    out[name] = {"hot_us": round(sorted(hot)[len(hot) // 2], 1), "cold_us": round(sorted(cold)[len(cold) // 2], 1),
                 "sustained_first200_us": round(sum(long[:200]) / 200, 1), "sustained_last200_us": round(sum(long[-200:]) / 200, 1)}
print(json.dumps(out))
