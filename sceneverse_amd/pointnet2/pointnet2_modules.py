"""PointNet++ set-abstraction / feature-propagation modules, API-compatible with the reference's
modules/third_party/pointnet2/pointnet2_modules.py:

    _PointnetSAModuleBase.forward   ref :34-75   sample (FPS) -> group -> SharedMLP -> max-pool
    PointnetSAModuleMSG             ref :78-124  (note ref :121-122: `mlp_spec[0] += 3` mutates the
                                                  caller's list when use_xyz -- reproduced)
    PointnetSAModule                ref :127-161
    PointnetFPModule                ref :356-416 three_nn + inverse-distance interpolation + MLP

State-dict layout is the reference's (`groupers.i`, `mlps.i.layer{j}.conv|bn.bn`).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def _sample_centres(self, xyz: torch.Tensor) -> Optional[torch.Tensor]:
        """(B,N,3) -> (B,npoint,3) FPS centres, or None for a group-all level."""
        if self.npoint is None:
            return None
        inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        picked = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds)
        return picked.transpose(1, 2).contiguous()

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum mlp[-1],npoint)."""
        new_xyz = self._sample_centres(xyz)
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped = grouper(xyz, new_xyz, features)          # (B, C', npoint, nsample)
            grouped = mlp(grouped)                             # (B, mlp[-1], npoint, nsample)
            # max over nsample (max_pool2d like ref :68-71, so tie routing in backward matches)
            pooled.append(F.max_pool2d(grouped, kernel_size=[1, grouped.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError("radii, nsamples and mlps must have equal length")
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            if npoint is not None:
                self.groupers.append(pointnet2_utils.QueryAndGroup(
                    radius, nsample, use_xyz=use_xyz, sample_uniformly=sample_uniformly))
            else:
                self.groupers.append(pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3  # in place, as the reference does
            self.mlps.append(pt_utils.SharedMLP(mlp_spec, bn=bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction level."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: interpolate `known_feats` onto `unknown` (3-NN, weights
    1/(dist+1e-8) normalised, ref :394-401), concat skip features, SharedMLP."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        new_features = (torch.cat([interpolated, unknow_feats], dim=1)
                        if unknow_feats is not None else interpolated)
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
