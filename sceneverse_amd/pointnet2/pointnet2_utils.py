"""Autograd front-end of the point ops, API-compatible with the reference's
modules/third_party/pointnet2/pointnet2_utils.py (names, argument order, return values,
differentiability), executed by libgps_hip.so.

    furthest_point_sample(xyz, npoint)            ref :48-77   (non-differentiable)
    gather_operation(features, idx)               ref :80-114
    three_nn(unknown, known)                      ref :117-146 (returns sqrt distances)
    three_interpolate(features, idx, weight)      ref :149-203
    grouping_operation(features, idx)             ref :206-254
    ball_query(radius, nsample, xyz, new_xyz)     ref :257-288 (non-differentiable)
    QueryAndGroup / GroupAll                      ref :291-373 / :376-419

`_ext` is a module-level attribute on purpose: the reference resolves its native ops through
`pointnet2_utils._ext` too, which is the seam tests use to swap implementations.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from . import _ext as _ext  # the nine native entry points (C ABI -> HIP)

# The native ops are fp32-only (indices must be bit-exact; the reference asserts fp32 too,
# include/utils.h:21-25).  Under torch.autocast(bf16) floating inputs are cast back to fp32.
_fwd = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


class FurthestPointSampling(Function):
    @staticmethod
    @_fwd
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3) f32 -> (B,npoint) i32 indices; idx[:,0] == 0."""
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    @_fwd
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) i32 -> (B,C,npoint)."""
        ctx.n_src = features.size(2)
        ctx.idx = idx
        return _ext.gather_points(features, idx)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        return _ext.gather_points_grad(grad_out.contiguous(), ctx.idx, ctx.n_src), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    @_fwd
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2 distances, idx (B,n,3) i32)."""
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    @_fwd
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B,c,m), idx/weight (B,n,3) -> (B,c,n)."""
        ctx.m_src = features.size(2)
        ctx.idx, ctx.weight = idx, weight
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        g = _ext.three_interpolate_grad(grad_out.contiguous(), ctx.idx, ctx.weight, ctx.m_src)
        return g, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    @_fwd
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) i32 -> (B,C,npoint,nsample)."""
        ctx.n_src = features.size(2)
        ctx.idx = idx
        return _ext.group_points(features, idx)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        return _ext.group_points_grad(grad_out.contiguous(), ctx.idx, ctx.n_src), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    @_fwd
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) i32."""
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query around `new_xyz`, then gather xyz (centre-relative) and features:
    returns (B, 3 + C, npoint, nsample) -- xyz channels first (ref :314-373)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if ret_unique_cnt and not sample_uniformly:
            raise AssertionError("ret_unique_cnt requires sample_uniformly")

    def _resample_uniformly(self, idx: torch.Tensor):
        # host-side re-draw of the padded slots (ref :335-343); rarely used, kept for API parity
        unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
        host = idx.cpu()
        for bi in range(host.shape[0]):
            for ri in range(host.shape[1]):
                uniq = torch.unique(host[bi, ri, :])
                k = uniq.shape[0]
                unique_cnt[bi, ri] = k
                extra = torch.randint(0, k, (self.nsample - k,), dtype=torch.long)
                host[bi, ri, :] = torch.cat((uniq, uniq[extra]))
        idx.copy_(host)
        return unique_cnt

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = self._resample_uniformly(idx) if self.sample_uniformly else None

        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is None:
            if not self.use_xyz:
                raise AssertionError("Cannot have not features and not use xyz as a feature!")
            new_features = grouped_xyz
        else:
            grouped_features = grouping_operation(features, idx)
            new_features = (torch.cat([grouped_xyz, grouped_features], dim=1)
                            if self.use_xyz else grouped_features)

        outs = [new_features]
        if self.ret_grouped_xyz:
            outs.append(grouped_xyz)
        if self.ret_unique_cnt:
            outs.append(unique_cnt)
        return outs[0] if len(outs) == 1 else tuple(outs)


class GroupAll(nn.Module):
    """One group holding every point: (B, 3 + C, 1, N) (ref :389-419)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features
