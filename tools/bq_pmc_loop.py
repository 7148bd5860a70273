"""ball_query SA1 / SA2 at the bench workload, 30 launches each (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sceneverse_amd.data.synthetic import synth_batch  # noqa: E402
from sceneverse_amd.pointnet2 import _ext as hip  # noqa: E402

batch = synth_batch(64, n_obj=80, n_pts=1024, seed=42, device="cuda")
pcs = batch["obj_fts"].reshape(-1, 1024, 6)
xyz = pcs[..., :3].contiguous()
xyz_t = xyz.transpose(1, 2).contiguous()
fps = hip.furthest_point_sampling(xyz, 32)
new_xyz = hip.gather_points(xyz_t, fps).transpose(1, 2).contiguous()
fps2 = hip.furthest_point_sampling(new_xyz, 16)
nx2 = hip.gather_points(new_xyz.transpose(1, 2).contiguous(), fps2).transpose(1, 2).contiguous()
for _ in range(30):
    hip.ball_query(new_xyz, xyz, 0.2, 32)
    hip.ball_query(nx2, new_xyz, 0.4, 32)
torch.cuda.synchronize()
