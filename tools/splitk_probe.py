"""Probe: weight-gradient GEMMs dW = dY^T X (K = tokens) as one library GEMM vs split-K over a batch
dimension (torch.bmm over S token chunks, partial sums added in fp32).  Prints us per variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


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
    dev = "cuda"
    shapes = [(19200, 768, 768), (19200, 2304, 768), (19200, 3072, 768), (19200, 768, 3072),
              (8320, 768, 768), (8320, 2304, 768), (8320, 2048, 768), (8320, 768, 2048),
              (5120, 768, 768), (5120, 2376, 768), (5120, 2048, 768), (5120, 768, 2048),
              (3200, 768, 768), (3200, 2304, 768), (3200, 3072, 768), (3200, 768, 3072)]
    have_out_dtype = True
    try:
        torch.bmm(torch.zeros(2, 8, 8, device=dev, dtype=torch.bfloat16),
                  torch.zeros(2, 8, 8, device=dev, dtype=torch.bfloat16), out_dtype=torch.float32)
    except Exception as ex:  # noqa: BLE001
        have_out_dtype = False
        print("bmm out_dtype unsupported:", type(ex).__name__)
    for T, N, K in shapes:
        dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
        x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
        base = timeit(lambda: torch.mm(dy.t(), x))
        ref = torch.mm(dy.t().float(), x.float())
        row = {"T": T, "N": N, "K": K, "mm_us": round(base, 1),
               "mm_TF": round(2 * T * N * K / base / 1e6, 0)}
        for S in (2, 4, 8, 16):
            if T % S:
                continue
            a = dy.view(S, T // S, N).transpose(1, 2)
            b = x.view(S, T // S, K)
            if have_out_dtype:
                fn = lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)  # noqa: E731
            else:
                fn = lambda: torch.bmm(a, b).sum(0, dtype=torch.float32)  # noqa: E731
            us = timeit(fn)
            err = (fn().float() - ref).abs().max().item() / ref.abs().max().item()
            row[f"S{S}_us"] = round(us, 1)
            row[f"S{S}_err"] = round(err, 5)
        print(row, flush=True)


if __name__ == "__main__":
    main()
