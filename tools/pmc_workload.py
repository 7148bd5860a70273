"""Workload for the rocprofv3 --pmc passes (HBM traffic per launch of the native point-path kernels).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o fetch --output-format csv -- python tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <dir> -o write --output-format csv -- python tools/pmc_workload.py

A plain 512 MiB device copy runs first: its byte count is known, so tools/pmc_traffic.py can calibrate
the two counters in this environment (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of a wide
coalesced read; WRITE_SIZE is uncalibrated) before converting the kernels' counters to bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sceneverse_amd.data.synthetic import synth_batch  # noqa: E402
from sceneverse_amd.pointnet2 import _ext as hip  # noqa: E402

CAL_BYTES = 512 << 20


def main():
    dev = "cuda"
    src = torch.randn(CAL_BYTES // 4, device=dev)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)                      # calibration: CAL_BYTES read + CAL_BYTES written
    torch.cuda.synchronize()

    d = synth_batch(64, seed=42)
    pcs = d["obj_fts"].reshape(-1, 1024, 6).to(dev)
    xyz = pcs[..., :3].contiguous()
    rgb = pcs[..., 3:].transpose(1, 2).contiguous()
    xyz_t = xyz.transpose(1, 2).contiguous()
    torch.manual_seed(0)

    def packed(cin, chans):
        ws, ss, c = [], [], cin
        for co in chans:
            ws.append(torch.randn(co, c, device=dev) * (2.0 / c) ** 0.5)
            ss.append(torch.randn(co, device=dev) * 0.05)
            c = co
        return hip.sa_mlp_pack(ws, ss, 'bf16x3')

    wp1, wp2 = packed(6, [64, 64, 128]), packed(131, [128, 128, 256])
    for _ in range(3):
        fps = hip.furthest_point_sampling(xyz, 32)
        new_xyz = hip.gather_points(xyz_t, fps).transpose(1, 2).contiguous()
        idx = hip.ball_query(new_xyz, xyz, 0.2, 32)
        f1 = hip.sa_mlp_forward(xyz, new_xyz, rgb, idx, wp1, [64, 64, 128], 'bf16x3')
        fps2 = hip.furthest_point_sampling(new_xyz, 16)
        nx2 = hip.gather_points(new_xyz.transpose(1, 2).contiguous(), fps2).transpose(1, 2).contiguous()
        idx2 = hip.ball_query(nx2, new_xyz, 0.4, 32)
        hip.sa_mlp_forward(new_xyz, nx2, f1, idx2, wp2, [128, 128, 256], 'bf16x3')
        # the unfused reference API on the same data (what the fused launches absorb)
        hip.group_points(xyz_t, idx)
        hip.group_points(f1, idx2)
    # fused residual+LayerNorm and attention at the bench shapes (forward + backward)
    from sceneverse_amd.modules.layers.fused_attention import _FusedSelfAttention
    from sceneverse_amd.modules.layers.fused_norm import add_dropout_layer_norm
    norm = torch.nn.LayerNorm(768).to(dev)
    for rows in (19200, 8320, 5120, 3200):
        x = torch.randn(rows, 768, device=dev, requires_grad=True)
        h = torch.randn(rows, 768, device=dev).to(torch.bfloat16).requires_grad_(True)
        for _ in range(3):
            add_dropout_layer_norm(x, h, norm, 0.1, True).sum().backward()
    for L, spatial in ((80, True), (130, False)):
        W = 3 * 768 + (72 if spatial else 0)
        packed = torch.randn(64, L, W, device=dev).to(torch.bfloat16).requires_grad_(True)
        pl = torch.rand(64, L, L, 5, device=dev) if spatial else None
        mask = torch.zeros(64, L, dtype=torch.bool, device=dev)
        for _ in range(3):
            _FusedSelfAttention.apply(packed, pl, mask, 12, 0.0, 0, None).float().sum().backward()
    # [r3] the variable-length text attention of the step: 64 sentences (6..50 tokens) + 64 captions (30..300 tokens)
    # packed back to back, the lengths of bench.py's batch (synth_batch(64, seed=42))
    from sceneverse_amd.modules.layers.fused_attention import fused_varlen_self_attention
    lens = torch.cat([d["txt_masks"].sum(1), d["scene_txt_masks"].sum(1)]).to(torch.int32).to(dev)
    cu = torch.zeros(lens.numel() + 1, dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    n_valid = cu[-1:].clone()
    order = torch.argsort(lens, descending=True).to(torch.int32)
    rows_all = 64 * 350
    packed = torch.randn(rows_all, 3 * 768, device=dev).to(torch.bfloat16).requires_grad_(True)
    for _ in range(3):
        fused_varlen_self_attention(packed, cu, 128, 300, 12, order=order).float().sum().backward()
    # the MFMA GEMMs of the largest Linears of the step (forward GELU, input gradient x GELU', weight gradient) and
    # the optimizer pass; tools/pmc_traffic.py tells the launches of one kernel symbol apart by their grid size
    from sceneverse_amd import _native
    from sceneverse_amd.modules.layers import gemm as GM
    T, K, N = 22400, 768, 3072          # token rows of both BERT texts (3 200 + 19 200) in one GEMM
    x = torch.randn(T, K, device=dev).to(torch.bfloat16)
    w = (0.02 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    dyb = torch.randn(T, N, device=dev).to(torch.bfloat16)
    for _ in range(3):                 # with the step's device-side row count (valid text tokens of the bench batch)
        h, pre = GM.linear_forward(x, w, b, act="gelu", want_pre=True, rows_dev=n_valid)
        GM.linear_dgrad(dyb, w, None, rows_dev=n_valid)
        GM.linear_wgrad(dyb, x, rows_dev=n_valid)
    print("valid text rows", int(n_valid), "of", rows_all, flush=True)
    # [r4] the grouped weight-gradient launch of the step: the 69 problems of one backward pass (4 text layers over the
    # live text rows, 4 object layers over 5 120 rows, 4 joint layers over the live joint rows [r6: sentence tokens +
    # real objects of the bench batch, the device-side extent of the compacted unified encoder]), written into fresh buffers
    probs, keep = [], []
    n_joint = (d["txt_masks"].sum() + d["obj_masks"].sum()).to(torch.int32).reshape(1).to(dev)

    def add(T, parts, K_in, ext=None):
        dy = (0.1 * torch.randn(T, sum(parts), device=dev)).to(torch.bfloat16)
        xx = torch.randn(T, K_in, device=dev).to(torch.bfloat16)
        r = 0
        for n in parts:
            q = _native.WgradProblem()
            C, cs = torch.empty(n, K_in, device=dev), torch.empty(n, device=dev)
            q.M, q.N, q.K, q.accumulate = n, K_in, T, 0
            q.A, q.lda, q.B, q.ldb = dy.data_ptr() + 2 * r, dy.stride(0), xx.data_ptr(), K_in
            q.C, q.ldc, q.colsum = C.data_ptr(), K_in, cs.data_ptr()
            q.extent_dev = ext.data_ptr() if ext is not None else None
            probs.append(q)
            keep.extend((dy, xx, C, cs))
            r += n
    for _ in range(4):
        add(rows_all, [768, 768, 768], 768, n_valid), add(rows_all, [768], 768, n_valid)
        add(rows_all, [3072], 768, n_valid), add(rows_all, [768], 3072, n_valid)
        add(5120, [768, 768, 768, 72], 768), add(5120, [768], 768), add(5120, [2048], 768), add(5120, [768], 2048)
        add(8320, [2304], 768, n_joint), add(8320, [768], 768, n_joint), add(8320, [2048], 768, n_joint), add(8320, [768], 2048, n_joint)
    # [r5] + the masked-LM head's transform (dense 768 -> 768 over the labelled rows only, modules/heads/pretrain_head.py)
    n_lm = torch.tensor([480], dtype=torch.int32, device=dev)
    add(3200, [768], 768, n_lm)
    arr = (_native.WgradProblem * len(probs))(*probs)
    for _ in range(3):
        _native.check(_native.load().gps_gemm_wgrad_grouped(arr, len(probs), torch.cuda.current_stream().cuda_stream), "wgrad_grouped")
    print("grouped weight gradients:", len(probs), "problems; live joint rows", int(n_joint), flush=True)
    ps = [torch.nn.Parameter(torch.randn(4096, 768, device=dev)) for _ in range(8)]
    from sceneverse_amd.optim.fused_adamw import GpsAdamW
    opt = GpsAdamW(ps, lr=1e-3)
    for p_ in ps:
        p_.grad = torch.randn_like(p_)
    for _ in range(3):
        opt.step(max_grad_norm=1.0)
    # bias-gradient column sums and the loader-side object kernel
    import numpy as np
    from sceneverse_amd.common import colsum as WS
    from sceneverse_amd.data import gpu_objects as G
    for T, N in ((19200, 3072), (19200, 768), (8320, 2048)):
        dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
        for _ in range(3):
            WS.colsum_bf16(dy)
    rng = np.random.default_rng(0)
    scans = G.PackedScans(dev)
    for s in range(16):
        n = int(rng.integers(20, 80))
        ks = np.exp(rng.uniform(np.log(50), np.log(20000), size=n)).astype(np.int64)
        scans.add_scan(f"s{s}", rng.normal(size=(int(ks.sum()), 3)).astype(np.float32),
                       rng.integers(0, 256, size=(int(ks.sum()), 3), dtype=np.uint8), np.repeat(np.arange(n), ks),
                       list(range(n)))
    scans.finalize()
    slots = G.batch_rows(scans, [f"s{i % 16}" for i in range(64)], 80).to(dev)
    for _ in range(3):
        G.obj_processing_post(scans, slots, 1024, seed=1)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
