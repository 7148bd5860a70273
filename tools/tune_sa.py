"""Within-run A/B of the fused SA level's launch geometry (waves per workgroup).

    python tools/tune_sa.py build      # here: compile one standalone .so per variant (hipcc, no GPU needed)
    python tools/tune_sa.py run        # on the GPU: interleaved timing rounds of every variant

Variants are compile-time (-DGPS_SA1_WAVES / -DGPS_SA2_WAVES); the .so files land in
sceneverse_amd/csrc/tune/ (git-ignored, shipped by gpurun)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TUNE = os.path.join(ROOT, "sceneverse_amd", "csrc", "tune")
SRC = os.path.join(ROOT, "sceneverse_amd", "csrc", "gps_sa_mlp.hip")
VARIANTS = [(8, 4), (8, 8)]


def build():
    os.makedirs(TUNE, exist_ok=True)
    for w1, w2 in VARIANTS:
        out = os.path.join(TUNE, f"libsa_w{w1}_{w2}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               f"-DGPS_SA1_WAVES={w1}", f"-DGPS_SA2_WAVES={w2}", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out]
        print(" ".join(cmd))
        subprocess.check_call(cmd)


def run():
    import torch
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.pointnet2 import _ext as hip
    dev = "cuda"
    d = synth_batch(64, seed=42)
    pcs = d["obj_fts"].reshape(-1, 1024, 6).to(dev)
    xyz = pcs[..., :3].contiguous()
    rgb = pcs[..., 3:].transpose(1, 2).contiguous()
    xyz_t = xyz.transpose(1, 2).contiguous()
    b = xyz.shape[0]
    fps = hip.furthest_point_sampling(xyz, 32)
    new_xyz = hip.gather_points(xyz_t, fps).transpose(1, 2).contiguous()
    idx = hip.ball_query(new_xyz, xyz, 0.2, 32)
    fps2 = hip.furthest_point_sampling(new_xyz, 16)
    nx2 = hip.gather_points(new_xyz.transpose(1, 2).contiguous(), fps2).transpose(1, 2).contiguous()
    idx2 = hip.ball_query(nx2, new_xyz, 0.4, 32)
    torch.manual_seed(0)

    def packed(cin, chans):
        ws, ss, c = [], [], cin
        for co in chans:
            ws.append(torch.randn(co, c, device=dev) * (2.0 / c) ** 0.5)
            ss.append(torch.randn(co, device=dev) * 0.05)
            c = co
        return hip.sa_mlp_pack(ws, ss, "bf16x3")

    # reference outputs from the fp32-MFMA path (independent pack format and kernel)
    def packed32(seed_cin, chans):
        torch.manual_seed(0)
    torch.manual_seed(0)
    ws1, ss1, ws2, ss2 = [], [], [], []
    c = 6
    for co in [64, 64, 128]:
        ws1.append(torch.randn(co, c, device=dev) * (2.0 / c) ** 0.5); ss1.append(torch.randn(co, device=dev) * 0.05); c = co
    c = 131
    for co in [128, 128, 256]:
        ws2.append(torch.randn(co, c, device=dev) * (2.0 / c) ** 0.5); ss2.append(torch.randn(co, device=dev) * 0.05); c = co
    wp1, wp2 = hip.sa_mlp_pack(ws1, ss1, "bf16x3"), hip.sa_mlp_pack(ws2, ss2, "bf16x3")
    f1 = hip.sa_mlp_forward(xyz, new_xyz, rgb, idx, hip.sa_mlp_pack(ws1, ss1, "fp32"), [64, 64, 128], "fp32")
    ref2 = hip.sa_mlp_forward(new_xyz, nx2, f1, idx2, hip.sa_mlp_pack(ws2, ss2, "fp32"), [128, 128, 256], "fp32")
    prod1 = hip.sa_mlp_forward(xyz, new_xyz, rgb, idx, wp1, [64, 64, 128], "bf16x3")
    prod2 = hip.sa_mlp_forward(new_xyz, nx2, f1, idx2, wp2, [128, 128, 256], "bf16x3")
    print("product lib vs fp32 path: SA1", (prod1 - f1).abs().max().item(), "SA2", (prod2 - ref2).abs().max().item(),
          "scale", f1.abs().max().item(), ref2.abs().max().item())
    out1 = torch.empty_like(f1)
    out2 = torch.empty_like(ref2)
    st = torch.cuda.current_stream().cuda_stream
    vp, ci = ctypes.c_void_p, ctypes.c_int
    libs = {}
    for w1, w2 in VARIANTS:
        lib = ctypes.CDLL(os.path.join(TUNE, f"libsa_w{w1}_{w2}.so"))
        lib.gps_sa_mlp_forward_bf16x3.argtypes = [ci] * 8 + [vp] * 7
        lib.gps_sa_mlp_forward_bf16x3.restype = ci
        libs[(w1, w2)] = lib

    def sa1(lib):
        return lib.gps_sa_mlp_forward_bf16x3(b, 1024, 32, 32, 3, 64, 64, 128, xyz.data_ptr(), new_xyz.data_ptr(),
                                             rgb.data_ptr(), idx.data_ptr(), wp1.data_ptr(), out1.data_ptr(), st)

    def sa2(lib):
        return lib.gps_sa_mlp_forward_bf16x3(b, 32, 16, 32, 128, 128, 128, 256, new_xyz.data_ptr(), nx2.data_ptr(),
                                             f1.data_ptr(), idx2.data_ptr(), wp2.data_ptr(), out2.data_ptr(), st)

    for key, lib in libs.items():          # correctness of every variant first
        assert sa1(lib) == 0 and sa2(lib) == 0, key
        torch.cuda.synchronize()
        e1, e2 = (out1 - f1).abs().max().item(), (out2 - ref2).abs().max().item()
        print("variant", key, "vs fp32 path: SA1 err", e1, "SA2 err", e2)
        bad = (out1 - f1).abs() > 1e-3
        if bad.any():
            ix = bad.nonzero()
            print("  SA1 bad entries:", ix.shape[0], "first", ix[:5].tolist(), "objs", ix[:, 0].unique()[:8].tolist(),
                  "chans", ix[:, 1].unique()[:16].tolist(), "groups", ix[:, 2].unique()[:40].tolist())
    res = {k: {"sa1": [], "sa2": []} for k in libs}
    for rnd in range(6):                   # interleaved rounds (DVFS / box noise is correlated within a round)
        for key, lib in libs.items():
            for name, fn in (("sa1", sa1), ("sa2", sa2)):
                for _ in range(3):
                    fn(lib)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn(lib)
                e.record()
                torch.cuda.synchronize()
                res[key][name].append(s.elapsed_time(e) * 100)      # us per launch
    for key, r in res.items():
        for name in ("sa1", "sa2"):
            v = sorted(r[name])
            print(f"waves(SA1,SA2)={key} {name}: median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f} us")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
