"""Runs the fused attention core forward + backward N times at one shape (for rocprofv3 --pmc passes).

    python tools/attn_pmc_loop.py --L 300 --p 0.1 [--spatial] [--n 12]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sceneverse_amd.modules.layers.fused_attention import _FusedSelfAttention  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=300)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--spatial", action="store_true")
    ap.add_argument("--n", type=int, default=12)
    a = ap.parse_args()
    dev = "cuda"
    W = 3 * 768 + (72 if a.spatial else 0)
    packed = torch.randn(a.B, a.L, W, device=dev).to(torch.bfloat16).requires_grad_(True)
    pl = torch.rand(a.B, a.L, a.L, 5, device=dev) if a.spatial else None
    mask = torch.zeros(a.B, a.L, dtype=torch.bool, device=dev)
    mask[:, a.L - a.L // 10:] = True
    go = torch.randn(a.B, a.L, 768, device=dev).to(torch.bfloat16)
    for _ in range(a.n):
        out = _FusedSelfAttention.apply(packed, pl, mask, 12, a.p, 1234, None)
        out.backward(go)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
