"""Generate tests/golden/point_ops_ref_gpu.pt on the MI355X box: outputs of the REFERENCE's own
pointnet2 extension (oracle/_ref/pointnet2_ref_ext.so = its unmodified sources compiled for gfx950
by oracle/build_ref.py) on the seeded inputs of tests/point_cases.py.

    gpurun -- python tests/golden/make_golden_gpu.py      (writes gpurun_out/point_ops_ref_gpu.pt;
                                                           copy it to tests/golden/)

These vectors pin the CPU oracle against the reference itself for the nine native ops
(tests/test_oracle_vs_golden_gpu.py runs without a GPU)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle.build_ref import load_ext  # noqa: E402
from point_cases import BQ_SHAPES, FPS_SHAPES, GROUP_SHAPES, generic_cloud, sa1_cloud  # noqa: E402


def main():
    ref = load_ext()
    dev = "cuda"
    fx = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0)}

    x = sa1_cloud().to(dev)
    fps = ref.furthest_point_sampling(x, 32)
    new_xyz = ref.gather_points(x.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    idx = ref.ball_query(new_xyz, x, 0.2, 32)
    fps2 = ref.furthest_point_sampling(new_xyz, 16)
    nx2 = ref.gather_points(new_xyz.transpose(1, 2).contiguous(), fps2).transpose(1, 2).contiguous()
    idx2 = ref.ball_query(nx2, new_xyz, 0.4, 32)
    fx["sa_chain"] = {k: v.cpu() for k, v in dict(fps=fps, new_xyz=new_xyz, idx=idx, fps2=fps2,
                                                   new_xyz2=nx2, idx2=idx2).items()}

    fx["fps"] = {}
    for n, m in FPS_SHAPES:
        c = generic_cloud(5, n, seed=n * 7 + m).to(dev)
        fx["fps"][(n, m)] = ref.furthest_point_sampling(c, m).cpu()

    fx["ball_query"] = {}
    for n, m, radius, nsample in BQ_SHAPES:
        c = generic_cloud(4, n, seed=n + m).to(dev)
        q = generic_cloud(4, m, seed=99).to(dev)
        fx["ball_query"][(n, m, radius, nsample)] = ref.ball_query(q, c, radius, nsample).cpu()

    fx["group"] = {}
    for c, n, npoint, nsample in GROUP_SHAPES:
        g = torch.Generator().manual_seed(c * 1000 + n)
        pts = torch.randn(3, c, n, generator=g)
        ix = torch.randint(0, n, (3, npoint, nsample), generator=g, dtype=torch.int32)
        go = torch.randn(3, c, npoint, nsample, generator=g)
        out = ref.group_points(pts.to(dev), ix.to(dev)).cpu()
        grad = ref.group_points_grad(go.to(dev), ix.to(dev), n).cpu()
        # store compactly: checksum-like summaries + a slice (full tensors can be large)
        fx["group"][(c, n, npoint, nsample)] = {
            "out_sum": out.double().sum().item(), "out_head": out.flatten()[:512].clone(),
            "grad": grad if grad.numel() <= 65536 else grad.flatten()[:65536].clone()}

    g = torch.Generator().manual_seed(4)
    fx["three"] = []
    for (b, n, m, c) in [(3, 100, 17, 5), (2, 1, 2, 3), (2, 300, 1500, 2), (1, 257, 3, 1)]:
        u = torch.randn(b, n, 3, generator=g)
        k = torch.randn(b, m, 3, generator=g)
        k[:, -1] = k[:, 0]
        feats = torch.randn(b, c, m, generator=g)
        w = torch.rand(b, n, 3, generator=g)
        go = torch.randn(b, c, n, generator=g)
        d2, i3 = ref.three_nn(u.to(dev), k.to(dev))
        out = ref.three_interpolate(feats.to(dev), i3, w.to(dev))
        grad = ref.three_interpolate_grad(go.to(dev), i3, w.to(dev), m)
        fx["three"].append({"shape": (b, n, m, c), "u": u, "k": k, "feats": feats, "w": w, "go": go,
                            "dist2": d2.cpu(), "idx": i3.cpu(), "out": out.cpu(), "grad": grad.cpu()})

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    dst = os.path.join(ROOT, "gpurun_out", "point_ops_ref_gpu.pt")
    torch.save(fx, dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
