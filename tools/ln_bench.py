"""Times of the fused residual + LayerNorm launches at the step's row counts (forward and backward), each captured 8 times
over rotating buffers into one HIP graph and replayed: us per launch.

    GPS_LN_FWD_GRID=4096 python tools/ln_bench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch import nn  # noqa: E402

from sceneverse_amd.modules.layers.fused_norm import add_dropout_layer_norm  # noqa: E402

DEV = "cuda"
CASES = [("text", 22400, 12608), ("joint", 8320, 5043), ("object", 5120, None), ("text full", 22400, None)]
SETS = 8


def main():
    norm = nn.LayerNorm(768).to(DEV)
    for name, n, live in CASES:
        rows = torch.tensor([live], dtype=torch.int32, device=DEV) if live else None
        xs = [torch.randn(n, 768, device=DEV, requires_grad=True) for _ in range(SETS)]
        hs = [torch.randn(n, 768, device=DEV).to(torch.bfloat16).requires_grad_(True) for _ in range(SETS)]
        gy = torch.randn(n, 768, device=DEV)
        gy16 = torch.randn(n, 768, device=DEV).to(torch.bfloat16)
        res = {}
        for what in ("fwd", "fwd+bwd"):
            def run():
                for x, h in zip(xs, hs):
                    y, y16 = add_dropout_layer_norm(x, h, norm, 0.1, True, want_bf16=True, rows_dev=rows)
                    if what != "fwd":
                        torch.autograd.backward([y, y16], [gy, gy16])
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(2):
                    run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g.replay()
                b.record()
                b.synchronize()
                ts.append(1e3 * a.elapsed_time(b) / SETS)
            res[what] = sorted(ts)[len(ts) // 2]
        print(f"{name:10s} rows {n:6d} live {live or n:6d}: forward {res['fwd']:6.1f} us, backward (+ reduce) {res['fwd+bwd'] - res['fwd']:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
