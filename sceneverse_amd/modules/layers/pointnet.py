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
        if self._distinct_ok(pc):
            out = self._forward_distinct(pc)
            if out is not None:
                return out
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

    # ---- distinct clouds only -------------------------------------------------------------------------------------------
    def _distinct_ok(self, pc) -> bool:
        """The frozen encoder on a GPU batch: every level is a per-object launch (results of one object do not depend
        on the others), nothing needs a gradient -- the preconditions of running the distinct clouds only."""
        from ...pointnet2 import pointnet2_modules as M
        if not (DISTINCT_CLOUDS and pc.is_cuda and pc.dtype == torch.float32 and pc.dim() == 3 and pc.size(-1) > 3
                and pc.is_contiguous() and not pc.requires_grad and hasattr(self.encoder[0], "forward_point_major")):
            return False
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return False
        first, last, ext = self.encoder[0], self.encoder[-1], M.pointnet2_utils._ext
        if not (all(M._is_frozen(mlp) for sa in self.encoder for mlp in sa.mlps) and last.npoint is None
                and len(self.encoder) >= 2 and self.encoder[-2].npoint is not None and hasattr(ext, "cloud_compact")
                and len(first.groupers) == 1 and hasattr(first.groupers[0], "nsample")):
            return False
        # the first level must take its point-major fused form (else the plan's three launches would be wasted work)
        chans = [l.conv.out_channels for l in first.mlps[0].children() if hasattr(l, "conv")]
        return bool(ext.sa_mlp_point_major_supported(pc.size(-1) - 3, chans, first.groupers[0].nsample, M._SA_PRECISION))

    def _forward_distinct(self, pc):
        """PointNet++ on the objects that are not pads + ONE pad representative (the reference pads scenes with constant
        clouds, data/datasets/dataset_wrapper.py:64-65 `pad=1.0`, and encodes every slot: 37 % of the bench batch); every
        pad slot reads the representative's row.  Which objects are pads is read from the data (gps_cloud_compact), not
        from a mask.  Bit-identical to encoding every slot: the levels are per-object launches.  None when a level cannot
        take its fused form (the caller then runs every object)."""
        from ...pointnet2 import pointnet2_modules as M
        ext = M.pointnet2_utils._ext
        plan = ext.cloud_compact(pc, rows_mult=int(self.encoder[-2].npoint))
        prev = M._OBJECT_ROWS
        M._OBJECT_ROWS = plan.rows16
        try:
            with ext.object_extent(plan.n_work), torch.no_grad():
                out = self.encoder[0].forward_point_major(plan.xyz, plan.feats_pm)
                if out is None:
                    return None
                xyz, features = out
                for sa in list(self.encoder)[1:]:
                    new_xyz = sa._sample_centres(xyz)
                    pooled = sa._forward_frozen(sa.groupers[0], sa.mlps[0], xyz, new_xyz, features) if len(sa.groupers) == 1 else None
                    if pooled is None:
                        return None
                    xyz, features = new_xyz, pooled
        finally:
            M._OBJECT_ROWS = prev
        y = self.fc(features.view(features.size(0), -1))      # rows past the work list are never selected below
        return y.index_select(0, plan.slot_of)


DISTINCT_CLOUDS = True


def set_distinct_clouds(flag: bool) -> None:
    """False: the frozen object encoder runs every object slot, pads included (A/B runs, the conservative bench figure)."""
    global DISTINCT_CLOUDS
    DISTINCT_CLOUDS = bool(flag)
