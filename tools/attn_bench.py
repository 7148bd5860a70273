"""Attention-core microbenchmark at the bench configuration's shapes (B = 64 scenes): every launch family of the step,
forward and backward, GPU time per launch from a replayed HIP graph of `iters` launches.

  spatial   (B, 80, 12 x 64) with the pairwise term: gps_attention_sp.hip (plane form) vs the general kernels
  joint     (B, 130) plain form with a key-padding mask (unified encoder)
  text      128 variable-length sequences packed back to back (64 sentences <= 50, 64 captions <= 300 tokens)

Checks each new path against the one it replaces (max |diff| relative to max |ref|) before timing it.
    python tools/attn_bench.py [--json out.json] [--iters 20]
GPS_ATTN_SP_BWD_OCC=3|4 selects the register budget of the spatial backward kernel (process-wide)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sceneverse_amd.modules.layers import fused_attention as FA  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

H, D = 12, 768
DEV = "cuda"


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def spatial_case(B, L, iters):
    g = torch.Generator().manual_seed(B * 1000 + L)
    W = 3 * D + 6 * H
    packed = torch.randn(B, L, W, generator=g)
    packed[..., 3 * D:] *= 2.0
    packed = packed.to(torch.bfloat16).to(DEV)
    centers = (torch.rand(B, L, 3, generator=g) * 8 - 4).to(DEV)
    from sceneverse_amd.modules.utils import calc_pairwise_locs
    pl = calc_pairwise_locs(centers, None)
    n_real = torch.randint(20, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] >= n_real[:, None]).to(DEV)
    go = torch.randn(B, L, D, generator=g).to(torch.bfloat16).to(DEV)
    res = {}
    outs = {}
    for name, planes in (("general", False), ("planes", True)):
        FA.set_spatial_planes(planes)
        x = packed.clone().requires_grad_(True)
        out = FA._FusedSelfAttention.apply(x, pl, mask, H, 0.0, 0, None)
        out.backward(go)
        outs[name] = (out.detach(), x.grad.detach())
        xf = packed.clone().requires_grad_(True)

        def fwd():
            return FA._FusedSelfAttention.apply(xf, pl, mask, H, 0.0, 0, None)
        o = fwd()

        def bwd():
            xf.grad = None
            o.backward(go, retain_graph=True)
        with torch.no_grad():
            res[name + "_fwd_us"] = timeit(lambda: FA._FusedSelfAttention.apply(packed, pl, mask, H, 0.0, 0, None), iters)
        res[name + "_bwd_us"] = timeit(bwd, iters)
    FA.set_spatial_planes(True)
    valid = ~mask
    res["out_vs_general"] = rel(outs["planes"][0][valid], outs["general"][0][valid])
    res["grad_vs_general"] = rel(outs["planes"][1][valid], outs["general"][1][valid])
    res["dsw_vs_general"] = rel(outs["planes"][1][valid][..., 3 * D:], outs["general"][1][valid][..., 3 * D:])
    return res


def joint_case(B, L, iters):
    g = torch.Generator().manual_seed(7)
    packed = torch.randn(B, L, 3 * D, generator=g).to(torch.bfloat16).to(DEV)
    n_real = torch.randint(L // 2, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] >= n_real[:, None]).to(DEV)
    go = torch.randn(B, L, D, generator=g).to(torch.bfloat16).to(DEV)
    res = {}
    for p in (0.0, 0.1):
        xf = packed.clone().requires_grad_(True)
        o = FA._FusedSelfAttention.apply(xf, None, mask, H, p, 0, FA._next_device_seed(torch.device(DEV)) if p else None)

        def bwd():
            xf.grad = None
            o.backward(go, retain_graph=True)
        with torch.no_grad():
            res[f"fwd_us_p{p}"] = timeit(lambda: FA._FusedSelfAttention.apply(
                packed, None, mask, H, p, 0, FA._next_device_seed(torch.device(DEV)) if p else None), iters)
        res[f"bwd_us_p{p}"] = timeit(bwd, iters)
    return res


def text_case(iters, full=False):
    g = torch.Generator().manual_seed(42)
    lens = torch.cat([torch.randint(6, 51, (64,), generator=g), torch.randint(30, 301, (64,), generator=g)])
    if full:
        lens = torch.cat([torch.full((64,), 50), torch.full((64,), 300)])
    cu = torch.zeros(129, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    T = int(cu[-1])
    order = torch.argsort(lens, descending=True).to(torch.int32).to(DEV)
    packed = torch.randn(T, 3 * D, generator=g).to(torch.bfloat16).to(DEV)
    go = torch.randn(T, D, generator=g).to(torch.bfloat16).to(DEV)
    cu = cu.to(DEV)
    res = {"rows": T, "sum_len_sq": int((lens.double() ** 2).sum())}
    for p in (0.0, 0.1):
        xf = packed.clone().requires_grad_(True)
        o = FA.fused_varlen_self_attention(xf, cu, 128, 300, H, dropout_p=p, training=True, order=order)

        def bwd():
            xf.grad = None
            o.backward(go, retain_graph=True)
        with torch.no_grad():
            res[f"fwd_us_p{p}"] = timeit(lambda: FA.fused_varlen_self_attention(packed, cu, 128, 300, H, dropout_p=p,
                                                                              training=True, order=order), iters)
        res[f"bwd_us_p{p}"] = timeit(bwd, iters)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    out = {"bwd_occ_env": os.environ.get("GPS_ATTN_SP_BWD_OCC", "")}
    out["spatial_L80"] = spatial_case(args.batch, 80, args.iters)
    print("spatial_L80", out["spatial_L80"], flush=True)
    out["spatial_L130"] = spatial_case(8, 130, args.iters)
    print("spatial_L130", out["spatial_L130"], flush=True)
    out["joint_L130"] = joint_case(args.batch, 130, args.iters)
    print("joint_L130", out["joint_L130"], flush=True)
    out["text_varlen"] = text_case(args.iters)
    print("text_varlen", out["text_varlen"], flush=True)
    out["text_full"] = text_case(args.iters, full=True)
    print("text_full", out["text_full"], flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
