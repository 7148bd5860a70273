"""Helpers shared by the vision / grounding modules (reference modules/utils.py).

    get_activation_fn, get_mlp_head, layer_repeat        ref :12-32
    calc_pairwise_locs                                   ref :38-87  (5-d pairwise geometry)
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


def get_activation_fn(activation_type):
    if activation_type not in ("relu", "gelu", "glu"):
        raise RuntimeError(f"activation function currently support relu/gelu, not {activation_type}")
    return getattr(F, activation_type)


def get_mlp_head(input_size, hidden_size, output_size, dropout=0):
    # Linear -> ReLU -> LayerNorm(eps 1e-12) -> Dropout -> Linear; indices 0..4 as in the reference
    return nn.Sequential(
        nn.Linear(input_size, hidden_size),
        nn.ReLU(),
        nn.LayerNorm(hidden_size, eps=1e-12),
        nn.Dropout(dropout),
        nn.Linear(hidden_size, output_size),
    )


def layer_repeat(module, N, share_layer=False):
    if share_layer:
        return nn.ModuleList([module] * N)
    # N-1 deep copies followed by the original, like the reference (matters for RNG-free init only)
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N - 1)] + [module])


def calc_pairwise_locs(obj_centers, obj_whls, eps=1e-10, pairwise_rel_type='center',
                       spatial_dist_norm=True, spatial_dim=5):
    """(B,L,3) centres [, (B,L,3) sizes] -> (B,L,L,spatial_dim) relative geometry of pair (l,t):
    [ d/d_max, dz/d, d_xy/d, dy/d_xy, dx/d_xy ] with d = sqrt(|c_l - c_t|^2 + eps) and d_max the
    per-scene maximum over ALL L*L pairs (padding slots included) -- ref :38-87."""
    if pairwise_rel_type == 'mlp':
        locs = torch.cat([obj_centers, obj_whls], 2)
        L = locs.size(1)
        return torch.cat([locs.unsqueeze(2).expand(-1, -1, L, -1),
                          locs.unsqueeze(1).expand(-1, L, -1, -1)], dim=3)

    if (obj_centers.is_cuda and obj_centers.dtype == torch.float32 and pairwise_rel_type == 'center'
            and spatial_dist_norm and spatial_dim == 5 and obj_centers.dim() == 3
            and not (torch.is_grad_enabled() and obj_centers.requires_grad) and obj_centers.size(1) <= 2048):
        return _pairwise_locs_native(obj_centers, eps)

    delta = obj_centers.unsqueeze(2) - obj_centers.unsqueeze(1)           # (B,L,L,3): c_l - c_t
    dist = torch.sqrt(torch.sum(delta ** 2, 3) + eps)                      # (B,L,L)
    if spatial_dist_norm:
        dmax = torch.max(dist.view(dist.size(0), -1), dim=1)[0]
        ndist = dist / dmax.view(-1, 1, 1)
    else:
        ndist = dist
    if spatial_dim == 1:
        return ndist.unsqueeze(3)

    dist_xy = torch.sqrt(torch.sum(delta[..., :2] ** 2, 3) + eps)
    if pairwise_rel_type == 'center':
        feats = [ndist, delta[..., 2] / dist, dist_xy / dist,
                 delta[..., 1] / dist_xy, delta[..., 0] / dist_xy]
    elif pairwise_rel_type == 'vertical_bottom':
        bottoms = obj_centers.clone()
        bottoms[:, :, 2] -= obj_whls[:, :, 2]
        bdelta = bottoms.unsqueeze(2) - bottoms.unsqueeze(1)
        bdist = torch.sqrt(torch.sum(bdelta ** 2, 3) + eps)
        bdist_xy = torch.sqrt(torch.sum(bdelta[..., :2] ** 2, 3) + eps)
        feats = [ndist, bdelta[..., 2] / bdist, bdist_xy / bdist,
                 delta[..., 1] / dist_xy, delta[..., 0] / dist_xy]
    else:
        raise NotImplementedError(f"pairwise_rel_type {pairwise_rel_type}")
    out = torch.stack(feats, dim=3)
    return out[..., 1:] if spatial_dim == 4 else out


def _pairwise_locs_native(obj_centers: torch.Tensor, eps: float) -> torch.Tensor:
    """One launch of libgps_hip.so's gps_pairwise_locs_planes (within 1e-6 of the torch formulation above).  The same
    launch writes the tensor a second time as five fp16 planes (b, 5, l, ld) -- the layout the spatial attention kernels
    read (include/gps_hip.h gps_attn_args.pl_planes); it travels as the attribute `_gps_planes` of the result."""
    from .. import _native
    c = obj_centers.contiguous()
    b, l, _ = c.shape
    out = torch.empty((b, l, l, 5), dtype=torch.float32, device=c.device)
    ld = (l + 3) // 4 * 4
    planes = torch.empty((b, 5, l, ld), dtype=torch.float16, device=c.device)
    with torch.cuda.device(c.device):
        st = _native.load().gps_pairwise_locs_planes(b, l, c.data_ptr(), float(eps), out.data_ptr(), planes.data_ptr(), ld,
                                                     torch.cuda.current_stream().cuda_stream)
    _native.check(st, "pairwise_locs")
    out._gps_planes = planes
    return out


def pairwise_planes(pl: torch.Tensor) -> torch.Tensor:
    """The fp16 plane form (B, 5, L, ld) of a (B, L, L, 5) pairwise tensor: the one `calc_pairwise_locs` attached, else one
    conversion launch (gps_pairwise_to_planes) whose result is cached on the tensor object."""
    planes = getattr(pl, "_gps_planes", None)
    if planes is not None and planes.device == pl.device:
        return planes
    from .. import _native
    b, l = pl.shape[0], pl.shape[1]
    src = pl.detach().float().contiguous()
    ld = (l + 3) // 4 * 4
    planes = torch.empty((b, 5, l, ld), dtype=torch.float16, device=pl.device)
    with torch.cuda.device(pl.device):
        st = _native.load().gps_pairwise_to_planes(b, l, src.data_ptr(), planes.data_ptr(), ld,
                                                   torch.cuda.current_stream().cuda_stream)
    _native.check(st, "pairwise_to_planes")
    try:
        pl._gps_planes = planes
    except AttributeError:
        pass
    return planes
