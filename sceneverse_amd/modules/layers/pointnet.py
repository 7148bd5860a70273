"""PointNet++ object encoder (reference modules/layers/pointnet.py:6-63): three set-abstraction
levels over each object's (P, 3 + C) cloud, then `fc`.  State-dict names `encoder.{i}.*`, `fc.*`."""
import torch
import torch.nn as nn

from ...pointnet2.pointnet2_modules import PointnetSAModule


def break_up_pc(pc):
    """(…, P, 3 + C) -> xyz (…, P, 3) contiguous, features (…, C, P) contiguous or None."""
    xyz = pc[..., 0:3].contiguous()
    features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
    return xyz, features


class PointNetPP(nn.Module):
    def __init__(self, sa_n_points: list, sa_n_samples: list, sa_radii: list, sa_mlps: list,
                 bn=True, use_xyz=True):
        super().__init__()
        n_sa = len(sa_n_points)
        if not (n_sa == len(sa_n_samples) == len(sa_radii) == len(sa_mlps)):
            raise ValueError('Lens of given hyper-params are not compatible')
        self.encoder = nn.ModuleList(
            PointnetSAModule(npoint=sa_n_points[i], nsample=sa_n_samples[i], radius=sa_radii[i],
                             mlp=sa_mlps[i], bn=bn, use_xyz=use_xyz)
            for i in range(n_sa))
        out_n_points = sa_n_points[-1] if sa_n_points[-1] is not None else 1
        self.fc = nn.Linear(out_n_points * sa_mlps[-1][-1], sa_mlps[-1][-1])

    def forward(self, features):
        """features: (B * N_objects, N_points, 3 + C) -> (B * N_objects, sa_mlps[-1][-1])."""
        pc = features
        first = self.encoder[0]
        out = None
        if pc.size(-1) > 3 and pc.is_cuda and pc.dtype == torch.float32 and hasattr(first, "forward_point_major"):
            # frozen first level: the colour columns are read in place from the interleaved cloud (no (B, C, P) copy)
            out = first.forward_point_major(pc[..., 0:3].contiguous(), pc[..., 3:])
        if out is None:
            xyz, features = break_up_pc(pc)
            out = first(xyz, features)
        xyz, features = out
        for sa in list(self.encoder)[1:]:
            xyz, features = sa(xyz, features)
        return self.fc(features.view(features.size(0), -1))
