"""PointNet++ set-abstraction / feature-propagation modules, API-compatible with the reference's
modules/third_party/pointnet2/pointnet2_modules.py:

    _PointnetSAModuleBase.forward   ref :34-75   sample (FPS) -> group -> SharedMLP -> max-pool
    PointnetSAModuleMSG             ref :78-124  (note ref :121-122: `mlp_spec[0] += 3` mutates the
                                                  caller's list when use_xyz -- reproduced)
    PointnetSAModule                ref :127-161
    PointnetSAModuleVotes           ref :164-274 single-scale level that also returns the sampled indices; max / avg /
                                                  RBF pooling, optional radius-normalised xyz
    PointnetSAModuleMSGVotes        ref :276-353 multi-scale level returning the sampled indices
    PointnetFPModule                ref :356-416 three_nn + inverse-distance interpolation + MLP
    PointnetLFPModuleMSG            ref :418-496 learnable feature propagation (ball-query grouping + post MLP)

The three `*Votes` / `LFP` classes are not used by any GPS configuration (VoteNet heritage of the third-party
package); they are provided so that the package's public names all resolve, on the same native ops.

State-dict layout is the reference's (`groupers.i`, `mlps.i.layer{j}.conv|bn.bn`).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils

# Fused frozen-encoder path (libgps_hip.so gps_sa_mlp_forward): on by default on GPU tensors,
# `set_fused_sa(False)` restores the op-by-op path (used by the A/B parity tests).
_FUSED_SA = True


def set_fused_sa(flag: bool) -> None:
    global _FUSED_SA
    _FUSED_SA = bool(flag)


# Arithmetic of the fused level: "bf16x3" = split-bf16 products on the bf16 MFMA (~2^-16 relative
# error per product, default), "fp32" = v_mfma_f32_32x32x2_f32 (bitwise an fp32 fmaf chain).
_SA_PRECISION = "bf16x3"


def set_sa_precision(name: str) -> None:
    """"bf16x3" (default: split-bf16 triple products, within 1e-4 of the fp32 op-by-op path), "fp32" (fp32 MFMA), or
    "bf16" -- OPT-IN: the bf16x3 kernels with ONE product per multiply-accumulate (bf16 operands, fp32 accumulation: what
    torch's bf16 autocast computes for the reference's Conv2d stacks, pytorch_utils.py:11-36); a third of the matrix work,
    features within 2e-2 of the fp32 path's scale (tests/test_gpu_sa_fused.py; include/gps_hip.h gps_sa_mlp_set_products).
    The split-operand group-all level keeps its three products in every mode."""
    global _SA_PRECISION
    if name not in ("fp32", "bf16x3", "bf16"):
        raise ValueError(name)
    from .. import _native
    _native.load().gps_sa_mlp_set_products(1 if name == "bf16" else 3)
    _SA_PRECISION = "bf16x3" if name == "bf16" else name


def fold_shared_mlp(mlp: "pt_utils.SharedMLP"):
    """SharedMLP of (conv1x1 [bias], BatchNorm2d, ReLU) layers -> ([W' (c_out,c_in)], [shift (c_out)])
    with the batch-norm's RUNNING statistics folded in (valid in eval mode only):
        y = relu(W' x + shift),  W' = diag(g / sqrt(var + eps)) W,  shift = beta + (bias - mean) g / sqrt(var + eps)
    Returns None when the module is not of that form."""
    ws, shifts = [], []
    for layer in mlp.children():
        kids = dict(layer.named_children())
        conv, bnw, act = kids.get("conv"), kids.get("bn"), kids.get("activation")
        if conv is None or not isinstance(act, nn.ReLU) or list(kids) != ["conv"] + (["bn"] if bnw is not None else []) + ["activation"]:
            return None
        if not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1:
            return None
        w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).float()
        shift = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(w[:, 0])
        if bnw is not None:
            bn = bnw.bn
            if bn.training or not bn.track_running_stats:
                return None
            scale = (bn.weight.detach() if bn.affine else 1.0) / torch.sqrt(bn.running_var + bn.eps)
            w = w * scale[:, None]
            shift = (shift - bn.running_mean) * scale + (bn.bias.detach() if bn.affine else 0.0)
        ws.append(w.contiguous())
        shifts.append(shift.contiguous())
    return ws, shifts


# Row extent (device int32) of the group-all level's GEMMs while the object encoder runs on a distinct-cloud work list
# (modules/layers/pointnet.py): rows = work slots x points per object.  None = every row.
_OBJECT_ROWS = None


def _frozen_key(mlp: nn.Module):
    return tuple((t.data_ptr(), t._version) for t in list(mlp.parameters()) + list(mlp.buffers()))


def _is_frozen(mlp: nn.Module) -> bool:
    """No parameter needs a gradient (or autograd is off) and every batch-norm is in eval mode."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in mlp.parameters()):
        return False
    return not any(isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training for m in mlp.modules())


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
        if xyz.is_cuda and not xyz.requires_grad and xyz.dtype == torch.float32 and xyz.is_contiguous() \
                and hasattr(pointnet2_utils._ext, "furthest_point_sampling_xyz"):
            # indices and the points they name from the launch that picks them
            both = pointnet2_utils._ext.furthest_point_sampling_xyz(xyz, self.npoint)
            if both is not None:
                return both[1]
        inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        if xyz.is_cuda and not xyz.requires_grad:
            # the reference's transpose -> gather_operation -> transpose (pointnet2_modules.py:47-54) is a pure row
            # gather: same values without the two full-cloud copies
            return torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
        picked = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds)
        return picked.transpose(1, 2).contiguous()

    def forward_point_major(self, xyz: torch.Tensor, features_pm: torch.Tensor):
        """The frozen single-scale level on POINT-major features -- features_pm (B,N,C), e.g. the view cloud[..., 3:] of
        an interleaved (B,N,3+C) cloud -- without materialising the (B,C,N) transpose; None when that form does not
        apply (the caller then transposes and calls forward)."""
        ext = pointnet2_utils._ext
        if not (_FUSED_SA and xyz.is_cuda and len(self.groupers) == 1 and self.npoint is not None
                and hasattr(ext, "sa_mlp_point_major_supported") and not xyz.requires_grad and not features_pm.requires_grad):
            return None
        grouper, mlp = self.groupers[0], self.mlps[0]
        if not (isinstance(grouper, pointnet2_utils.QueryAndGroup) and _is_frozen(mlp) and grouper.use_xyz
                and not (grouper.sample_uniformly or grouper.normalize_xyz or grouper.ret_grouped_xyz or grouper.ret_unique_cnt)):
            return None
        chans = [l.conv.out_channels for l in mlp.children() if hasattr(l, "conv")]
        if not ext.sa_mlp_point_major_supported(features_pm.shape[2], chans, grouper.nsample, _SA_PRECISION):
            return None
        folded = self._folded(mlp, pack=True)
        if folded is None:
            return None
        new_xyz = self._sample_centres(xyz)
        with torch.no_grad():
            idx = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
            pooled = ext.sa_mlp_forward(xyz, new_xyz.float().contiguous(), features_pm, idx, folded[2], chans,
                                        _SA_PRECISION, point_major=True)
        return new_xyz, pooled

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum mlp[-1],npoint)."""
        new_xyz = self._sample_centres(xyz)
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            fused = self._forward_frozen(grouper, mlp, xyz, new_xyz, features)
            if fused is not None:
                pooled.append(fused)
                continue
            grouped = grouper(xyz, new_xyz, features)          # (B, C', npoint, nsample)
            grouped = mlp(grouped)                             # (B, mlp[-1], npoint, nsample)
            # max over nsample (max_pool2d like ref :68-71, so tie routing in backward matches)
            pooled.append(F.max_pool2d(grouped, kernel_size=[1, grouped.size(3)]).squeeze(-1))
        return new_xyz, (pooled[0] if len(pooled) == 1 else torch.cat(pooled, dim=1))     # cat of one tensor copies it


    # ---- frozen encoder: one native launch per level -------------------------------------------
    def _folded(self, mlp, pack: bool):
        """(weights, shifts[, packed buffer]) of `mlp`, cached until a parameter/buffer changes."""
        key = (_frozen_key(mlp), pack, _SA_PRECISION)
        cache = mlp.__dict__.get("_gps_folded")
        if cache is None or cache[0] != key:
            folded = fold_shared_mlp(mlp)
            if folded is None:
                cache = (key, None)
            else:
                ws, shifts = folded
                packed = pointnet2_utils._ext.sa_mlp_pack(ws, shifts, _SA_PRECISION) if pack else None
                cache = (key, (ws, shifts, packed))
            mlp.__dict__["_gps_folded"] = cache
        return cache[1]

    def _forward_frozen(self, grouper, mlp, xyz, new_xyz, features):
        """Pooled features (B, C_out, npoint) of one (grouper, mlp) pair when the encoder is frozen
        and the tensors live on the GPU, else None (the caller runs the op-by-op path)."""
        if not (_FUSED_SA and xyz.is_cuda and _is_frozen(mlp)):
            return None
        if torch.is_grad_enabled() and (xyz.requires_grad or (features is not None and features.requires_grad)):
            return None          # an earlier (trainable) level wants input gradients: the fused level computes none
        ext = pointnet2_utils._ext
        if isinstance(grouper, pointnet2_utils.QueryAndGroup):
            plain = not (grouper.sample_uniformly or grouper.normalize_xyz or grouper.ret_grouped_xyz
                         or grouper.ret_unique_cnt) and grouper.use_xyz and features is not None
            chans = [l.conv.out_channels for l in mlp.children() if hasattr(l, "conv")]
            if not plain or not hasattr(ext, "sa_mlp_supported") or \
                    not ext.sa_mlp_supported(features.shape[1], chans, grouper.nsample):
                return None
            folded = self._folded(mlp, pack=True)
            if folded is None:
                return None
            with torch.no_grad():
                idx = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
                return ext.sa_mlp_forward(xyz.float().contiguous(), new_xyz.float().contiguous(),
                                          features.float().contiguous(), idx, folded[2], chans, _SA_PRECISION)
        if isinstance(grouper, pointnet2_utils.GroupAll) and features is not None and grouper.use_xyz:
            # group-all level: every object is one 16-column group -> three plain GEMMs over
            # (B * N) rows with the folded weights (hipBLASLt), ReLU, max over N
            folded = self._folded(mlp, pack=False)
            if folded is None:
                return None
            ws, shifts, _ = folded
            with torch.no_grad(), torch.autocast(device_type="cuda", enabled=False):
                from ..modules.layers import gemm as G
                b, n = xyz.shape[0], xyz.shape[1]
                split = _SA_PRECISION == "bf16x3" and n == 16 and G.enabled() and all(w.shape[0] % 8 == 0 for w in ws)
                if not split:
                    x = torch.cat([xyz.float(), features.float().transpose(1, 2)], dim=2)   # (B, N, 3 + C)
                    x = x.reshape(b * n, -1)
                if split:
                    # split-bf16 (hi, lo) operands through libgps_hip.so's MFMA GEMM: ReLU + re-split in the
                    # epilogues of the first layers, ReLU + max over the 16 points of an object in the last
                    cache = mlp.__dict__.get("_gps_split3")
                    if cache is None or cache[0] is not folded:
                        k_pads = [(ws[0].shape[1] + 7) // 8 * 8] + [w.shape[0] for w in ws[:-1]]
                        cache = (folded, [(G.split3_weight(w, kp), s.float().contiguous())
                                          for w, s, kp in zip(ws, shifts, k_pads)])
                        mlp.__dict__["_gps_split3"] = cache
                    a = G.split3_points(xyz, features, cache[1][0][0].shape[1] // 3)
                    return G.split3_mlp_max16(a, cache[1], rows_dev=_OBJECT_ROWS).unsqueeze(-1)
                for w, sft in zip(ws, shifts):
                    x = torch.relu_(torch.addmm(sft, x, w.t()))
                return x.view(b, n, -1).amax(dim=1).unsqueeze(-1)
        return None


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


def _sample_centres(xyz: torch.Tensor, npoint: Optional[int], inds: Optional[torch.Tensor]):
    """FPS (unless the caller brings its own indices) + gather of the sampled centres: (inds, new_xyz (B, npoint, 3)).
    npoint None (group-all): nothing is sampled (the reference would call FPS with npoint = None there and fail)."""
    if npoint is None:
        return inds, None
    if inds is None:
        inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    centres = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds)
    return inds, centres.transpose(1, 2).contiguous()


def _bump_for_xyz(spec: List[int], use_xyz: bool) -> List[int]:
    # the reference adds the 3 xyz channels to the CALLER's list in place (ref :201-203, :311-313, :446-448)
    if use_xyz and len(spec) > 0:
        spec[0] += 3
    return spec


class PointnetSAModuleVotes(nn.Module):
    """Set-abstraction level that also hands back the indices of the sampled points (ref :164-274).

    forward(xyz (B,N,3), features (B,C,N), inds (B,npoint) or None)
        -> (new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds[, unique_cnt])
    pooling: 'max' | 'avg' | 'rbf' (Gaussian of the centre-relative xyz with width sigma, divided by nsample)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = 'max', sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling, self.use_xyz = pooling, use_xyz
        self.sigma = radius / 2 if sigma is None else sigma
        self.normalize_xyz, self.ret_unique_cnt = normalize_xyz, ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                         normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                                                         ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        self.mlp_module = pt_utils.SharedMLP(_bump_for_xyz(mlp, use_xyz), bn=bn)

    def forward(self, xyz, features=None, inds=None):
        if inds is not None:
            assert inds.shape[1] == self.npoint
        inds, new_xyz = _sample_centres(xyz, self.npoint, inds)
        grouped = self.grouper(xyz, new_xyz, features)
        unique_cnt = None
        if self.ret_unique_cnt:
            grouped_features, grouped_xyz, unique_cnt = grouped
        else:
            grouped_features, grouped_xyz = grouped
        new_features = self.mlp_module(grouped_features)                     # (B, mlp[-1], npoint, nsample)
        if self.pooling == 'max':
            new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == 'avg':
            new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == 'rbf':
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            new_features = torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True) / float(self.nsample)
        new_features = new_features.squeeze(-1)
        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleMSGVotes(nn.Module):
    """Multi-scale set abstraction returning the sampled indices (ref :276-353): one (ball query, SharedMLP, max-pool)
    branch per radius, channel-concatenated."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float], nsamples: List[int],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.npoint = npoint
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                               sample_uniformly=sample_uniformly)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))

    def forward(self, xyz, features=None, inds=None):
        inds, new_xyz = _sample_centres(xyz, self.npoint, inds)
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            f = mlp(grouper(xyz, new_xyz, features))
            pooled.append(F.max_pool2d(f, kernel_size=[1, f.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(pooled, dim=1), inds


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation (ref :418-496): for every radius, group the features of set 1 around the points of
    set 2, SharedMLP + max-pool, concatenate set 2's own features, `post_mlp`; branches channel-concatenated."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int], post_mlp: List[int],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                               sample_uniformly=sample_uniformly))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))

    def forward(self, xyz2, xyz1, features2, features1):
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            f = mlp(grouper(xyz1, xyz2, features1))                          # (B, mlp[-1], N2, nsample)
            f = F.max_pool2d(f, kernel_size=[1, f.size(3)]).squeeze(-1)      # (B, mlp[-1], N2)
            if features2 is not None:
                f = torch.cat([f, features2], dim=1)
            outs.append(self.post_mlp(f.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)
