"""Tensor-level mirror of the reference's pybind module `pointnet2._ext`
(/root/reference/modules/third_party/pointnet2/_ext_src/src/bindings.cpp:6-19): the same nine
function names, argument order, dtype/contiguity checks (include/utils.h:5-25 -> RuntimeError)
and return conventions as the host wrappers in sampling.cpp / ball_query.cpp / group_points.cpp /
interpolate.cpp -- but each call goes straight to the C ABI of libgps_hip.so on the current
torch stream.  GPU tensors only: there is deliberately no CPU path (the reference has none
either: `AT_ASSERT(false, "CPU not supported")`).
"""
from __future__ import annotations

import torch

from .. import _native


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-launch timing (bench.py): HIP events on the launch stream ------------------
_PROFILE = None  # None | {op name: [(start_event, end_event, algorithmic_bytes), ...]}


def profile_start() -> None:
    global _PROFILE
    _PROFILE = {}


def profiling() -> bool:
    return _PROFILE is not None


def profile_stop() -> dict:
    """-> {op: {"launches": n, "avg_us": t, "bytes_per_launch": B, "flops_per_launch": F,
    "mfma_dtype": "fp32"|"bf16"|None}} (synchronises)."""
    global _PROFILE
    rec, _PROFILE = _PROFILE or {}, None
    torch.cuda.synchronize()
    out = {}
    for name, evs in rec.items():
        ms = [s.elapsed_time(e) for s, e, *_ in evs]
        # launches with a device-side extent (gps_gemm_args.extent_dev) did only `frac` of the work their static
        # shape names: the snapshot of the extent word taken at launch time is read back here, once
        fr = [ev[5]() if ev[5] is not None else 1.0 for ev in evs]
        out[name] = {"launches": len(evs), "avg_us": 1e3 * sum(ms) / len(ms),
                     "bytes_per_launch": sum((ev[2](f) if callable(ev[2]) else ev[2] * f) for ev, f in zip(evs, fr)) / len(evs),
                     "flops_per_launch": sum(ev[3] * f for ev, f in zip(evs, fr)) / len(evs),
                     "mfma_dtype": evs[0][4]}
        if any(ev[5] is not None for ev in evs):
            out[name]["work_fraction"] = sum(fr) / len(fr)
    return out


class _timed:
    """Brackets one native launch with two events on the current (= launch) stream."""

    def __init__(self, name: str, algo_bytes: int, algo_flops: int = 0, mfma_dtype=None, work_fraction=None):
        """work_fraction: None, or a callable evaluated at profile_stop() -> the share of the static shape's
        algorithmic work this launch really did (device-side extents)."""
        self.name, self.bytes, self.flops, self.mfma = name, algo_bytes, algo_flops, mfma_dtype
        self.work_fraction = work_fraction

    def __enter__(self):
        if _PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.e.record()
            _PROFILE.setdefault(self.name, []).append((self.s, self.e, self.bytes, self.flops, self.mfma,
                                                       self.work_fraction))
        return False


def _chk(t: torch.Tensor, name: str, dtype: torch.dtype) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (libgps_hip has no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {'a float' if dtype == torch.float32 else 'an int'} tensor")


def _same_device(*ts: torch.Tensor) -> None:
    d = ts[0].device
    for t in ts[1:]:
        if t.device != d:
            raise RuntimeError("all tensors must live on the same GPU")


def gather_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(points, idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed("gather_points", 4 * (b * m + 2 * b * c * m)):   # idx + gathered elements read, output written
        st = _native.load().gps_gather_points(b, c, n, m, points.data_ptr(), idx.data_ptr(),
                                              out.data_ptr(), _stream())
    _native.check(st, "gather_points")
    return out


def gather_points_grad(grad_out: torch.Tensor, idx: torch.Tensor, n: int) -> torch.Tensor:
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(grad_out, idx)
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed("gather_points_grad", 4 * (b * c * m + b * m + b * c * int(n))):
        st = _native.load().gps_gather_points_grad(b, c, int(n), m, grad_out.data_ptr(),
                                                   idx.data_ptr(), out.data_ptr(), _stream())
    _native.check(st, "gather_points_grad")
    return out


def furthest_point_sampling(points: torch.Tensor, nsamples: int) -> torch.Tensor:
    _chk(points, "points", torch.float32)
    b, n, _ = points.shape
    m = int(nsamples)
    out = torch.empty((b, m), dtype=torch.int32, device=points.device)
    temp = None
    if n > 2048:  # GPS_FPS_MAX_RESIDENT_N: streaming form keeps running distances in HBM
        temp = torch.empty((b, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed(f"furthest_point_sampling(n={n},m={m})", 4 * (b * n * 3 + b * m)):
        st = _native.load().gps_furthest_point_sampling(
            b, n, m, points.data_ptr(), temp.data_ptr() if temp is not None else None,
            out.data_ptr(), _stream())
    _native.check(st, "furthest_point_sampling")
    return out


def furthest_point_sampling_xyz(points: torch.Tensor, nsamples: int):
    """-> (indices (b, m) int32, sampled points (b, m, 3) fp32) from one launch; None when the cloud is too large for the
    register-resident form (the caller then samples and gathers)."""
    _chk(points, "points", torch.float32)
    b, n, _ = points.shape
    m = int(nsamples)
    if n > 2048:
        return None
    out = torch.empty((b, m), dtype=torch.int32, device=points.device)
    cen = torch.empty((b, m, 3), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed(f"furthest_point_sampling(n={n},m={m})", 4 * (b * n * 3 + b * m * 4)):
        st = _native.load().gps_furthest_point_sampling_xyz(b, n, m, points.data_ptr(), out.data_ptr(), cen.data_ptr(),
                                                            _stream())
    _native.check(st, "furthest_point_sampling_xyz")
    return out, cen


class CloudPlan:
    """Work list of gps_cloud_compact: the objects that are not pads (constant clouds) + one pad representative."""
    __slots__ = ("xyz", "feats_pm", "obj_of", "slot_of", "scal", "n_work", "rows16")


def cloud_compact(cloud: torch.Tensor, rows_mult: int = 16) -> CloudPlan:
    """cloud (b, n, 3 + C) fp32 contiguous -> CloudPlan: xyz (b, n, 3) / feats_pm (b, n, C) of the work slots, obj_of,
    slot_of (int64: `result.index_select(0, slot_of)` is the result of every object), n_work / rows16 device ints."""
    _chk(cloud, "cloud", torch.float32)
    b, n, ld = cloud.shape
    dev = cloud.device
    plan = CloudPlan()
    plan.xyz = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    plan.feats_pm = torch.empty((b, n, max(ld - 3, 0)), dtype=torch.float32, device=dev)
    plan.obj_of = torch.empty(b, dtype=torch.int32, device=dev)
    plan.slot_of = torch.empty(b, dtype=torch.int64, device=dev)
    plan.scal = torch.empty(4, dtype=torch.int32, device=dev)
    flag = torch.empty(2 * b, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev), _timed(f"cloud_compact(n={n},ld={ld})", 4 * b * n * ld * 2):
        st = _native.load().gps_cloud_compact(b, n, ld, cloud.data_ptr(), int(rows_mult), flag.data_ptr(), plan.obj_of.data_ptr(),
                                              plan.slot_of.data_ptr(), plan.scal.data_ptr(), plan.xyz.data_ptr(),
                                              plan.feats_pm.data_ptr() if ld > 3 else None, _stream())
    _native.check(st, "cloud_compact")
    plan.n_work, plan.rows16 = plan.scal[0:1], plan.scal[3:4]
    return plan


class object_extent:
    """with object_extent(n_dev): the per-object launches of this module process objects [0, *n_dev) only."""

    def __init__(self, n_dev):
        self.n_dev = n_dev

    def __enter__(self):
        _native.load().gps_point_set_object_extent(self.n_dev.data_ptr() if self.n_dev is not None else None)
        return self

    def __exit__(self, *exc):
        _native.load().gps_point_set_object_extent(None)
        return False


def three_nn(unknowns: torch.Tensor, knows: torch.Tensor):
    _chk(unknowns, "unknowns", torch.float32)
    _chk(knows, "knows", torch.float32)
    _same_device(unknowns, knows)
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    with torch.cuda.device(unknowns.device), _timed("three_nn", 4 * (b * n * 3 + b * m * 3 + 2 * b * n * 3)):
        st = _native.load().gps_three_nn(b, n, m, unknowns.data_ptr(), knows.data_ptr(),
                                         dist2.data_ptr(), idx.data_ptr(), _stream())
    _native.check(st, "three_nn")
    return [dist2, idx]


def three_interpolate(points: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_device(points, idx, weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed("three_interpolate", 4 * (b * c * m + 2 * b * n * 3 + b * c * n)):
        st = _native.load().gps_three_interpolate(b, c, m, n, points.data_ptr(), idx.data_ptr(),
                                                  weight.data_ptr(), out.data_ptr(), _stream())
    _native.check(st, "three_interpolate")
    return out


def three_interpolate_grad(grad_out: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor,
                           m: int) -> torch.Tensor:
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_device(grad_out, idx, weight)
    b, c, n = grad_out.shape
    out = torch.empty((b, c, int(m)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed("three_interpolate_grad", 4 * (b * c * n + 2 * b * n * 3 + b * c * int(m))):
        st = _native.load().gps_three_interpolate_grad(b, c, n, int(m), grad_out.data_ptr(),
                                                       idx.data_ptr(), weight.data_ptr(),
                                                       out.data_ptr(), _stream())
    _native.check(st, "three_interpolate_grad")
    return out


def ball_query(new_xyz: torch.Tensor, xyz: torch.Tensor, radius: float, nsample: int) -> torch.Tensor:
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    _same_device(new_xyz, xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device), _timed(f"ball_query(n={n},m={m},ns={int(nsample)})", 4 * (b * (n + m) * 3 + b * m * int(nsample))):
        st = _native.load().gps_ball_query(b, n, m, float(radius), int(nsample),
                                           new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(),
                                           _stream())
    _native.check(st, "ball_query")
    return idx


def group_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(points, idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed(f"group_points(c={c},n={n},np={npoints},ns={nsample})", 4 * (b * c * n + b * npoints * nsample + b * c * npoints * nsample)):
        st = _native.load().gps_group_points(b, c, n, npoints, nsample, points.data_ptr(),
                                             idx.data_ptr(), out.data_ptr(), _stream())
    _native.check(st, "group_points")
    return out


def group_points_grad(grad_out: torch.Tensor, idx: torch.Tensor, n: int) -> torch.Tensor:
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(grad_out, idx)
    b, c, npoints, nsample = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed(f"group_points_grad(c={c},n={int(n)},np={npoints},ns={nsample})", 4 * (b * c * npoints * nsample + b * npoints * nsample + b * c * int(n))):
        st = _native.load().gps_group_points_grad(b, c, int(n), npoints, nsample,
                                                  grad_out.data_ptr(), idx.data_ptr(),
                                                  out.data_ptr(), _stream())
    _native.check(st, "group_points_grad")
    return out



# ---- fused set-abstraction level (additions to the nine reference names; include/gps_hip.h) ----
def sa_mlp_supported(c_feat: int, channels, nsample: int) -> bool:
    """True when libgps_hip.so implements the fused level for this MLP (c_feat+3 -> channels)."""
    if nsample != 32 or len(channels) != 3:
        return False
    return (c_feat, *channels) in ((3, 64, 64, 128), (128, 128, 128, 256))


SA_PRECISIONS = ("fp32", "bf16x3")


def sa_mlp_pack(weights, shifts, precision: str = "fp32") -> torch.Tensor:
    """weights[i] (c_out_i, c_in_i) BN-folded fp32 GPU tensors, shifts[i] (c_out_i) -> the packed
    buffer gps_sa_mlp_forward[_bf16x3] reads ([layer 1 | layer 2 | layer 3], K-slot order of the MFMA)."""
    lib = _native.load()
    assert precision in SA_PRECISIONS, precision
    x3 = precision == "bf16x3"
    layer_floats = lib.gps_sa_mlp_layer_floats_bf16x3 if x3 else lib.gps_sa_mlp_layer_floats
    pack_layer = lib.gps_sa_mlp_pack_layer_bf16x3 if x3 else lib.gps_sa_mlp_pack_layer
    sizes = []
    for w in weights:
        n = int(layer_floats(w.shape[1], w.shape[0]))
        if n < 0:
            raise RuntimeError(f"sa_mlp_pack: unsupported layer {w.shape[1]}->{w.shape[0]}")
        sizes.append(n)
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=weights[0].device)
    off = 0
    with torch.cuda.device(buf.device):
        for w, sft, n in zip(weights, shifts, sizes):
            w, sft = w.contiguous(), sft.contiguous()
            _chk(w, "weight", torch.float32)
            _chk(sft, "shift", torch.float32)
            st = pack_layer(w.shape[1], w.shape[0], w.data_ptr(), sft.data_ptr(),
                            buf.data_ptr() + 4 * off, _stream())
            _native.check(st, "sa_mlp_pack_layer")
            off += n
    return buf


def sa_mlp_point_major_supported(c_feat: int, channels, nsample: int, precision: str) -> bool:
    return precision == "bf16x3" and c_feat == 3 and list(channels) == [64, 64, 128] and nsample == 32


def sa_mlp_forward(xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor,
                   idx: torch.Tensor, wpack: torch.Tensor, channels, precision: str = "fp32",
                   point_major: bool = False) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N), idx (B,npoint,32) i32, packed folded
    MLP C+3 -> channels  ->  (B, channels[-1], npoint) pooled features (one launch).
    point_major: features is a (B,N,C) view with unit channel stride and row pitch features.stride(1) -- e.g.
    cloud[..., 3:] of the interleaved (B,N,3+C) cloud -- read in place (gps_sa_mlp_forward_bf16x3_pm)."""
    _chk(xyz, "xyz", torch.float32)
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(wpack, "wpack", torch.float32)
    if point_major:
        b, n, _ = xyz.shape
        if not (features.is_cuda and features.dtype == torch.float32 and features.dim() == 3 and features.stride(2) == 1
                and features.shape[:2] == (b, n) and features.stride(0) == n * features.stride(1)):
            raise ValueError("point-major features: (B, N, C) float32 view with unit channel stride and dense rows")
        c_feat, ld_feat = features.shape[2], features.stride(1)
        _, npoint, nsample = idx.shape
        c1, c2, c3 = (int(c) for c in channels)
        if not sa_mlp_point_major_supported(c_feat, [c1, c2, c3], nsample, precision):
            raise ValueError("point-major features: first level (3 -> 64-64-128, nsample 32, bf16x3) only")
        _same_device(xyz, new_xyz, features, idx, wpack)
        out = torch.empty((b, c3, npoint), dtype=torch.float32, device=xyz.device)
        algo = 4 * (b * 3 * n + b * 3 * npoint + b * c_feat * n + b * npoint * nsample + b * c3 * npoint)
        flops = 2 * b * npoint * nsample * ((3 + c_feat) * c1 + c1 * c2 + c2 * c3)
        with torch.cuda.device(xyz.device), _timed(f"sa_mlp_forward(c={c_feat},n={n},np={npoint},mlp={c1}-{c2}-{c3},{precision})",
                                                    algo, 3 * flops, "bf16"):
            st = _native.load().gps_sa_mlp_forward_bf16x3_pm(b, n, npoint, nsample, c_feat, c1, c2, c3, xyz.data_ptr(),
                                                             new_xyz.data_ptr(), features.data_ptr(), ld_feat, idx.data_ptr(),
                                                             wpack.data_ptr(), out.data_ptr(), _stream())
        _native.check(st, "sa_mlp_forward_pm")
        return out
    _chk(features, "features", torch.float32)
    _same_device(xyz, new_xyz, features, idx, wpack)
    b, n, _ = xyz.shape
    c_feat = features.shape[1]
    _, npoint, nsample = idx.shape
    c1, c2, c3 = (int(c) for c in channels)
    out = torch.empty((b, c3, npoint), dtype=torch.float32, device=xyz.device)
    # algorithmic bytes of THIS launch: every input read once (points, centres, features, neighbour indices) and the
    # pooled output written once.  (The unfused reference API moves 8x more for the same work -- two group_points
    # outputs and three conv / BN / ReLU round trips -- which is what fusing removes, not this kernel's roof.)
    algo = 4 * (b * 3 * n + b * 3 * npoint + b * c_feat * n + b * npoint * nsample + b * c3 * npoint)
    flops = 2 * b * npoint * nsample * ((3 + c_feat) * c1 + c1 * c2 + c2 * c3)
    x3 = precision == "bf16x3"
    fn = _native.load().gps_sa_mlp_forward_bf16x3 if x3 else _native.load().gps_sa_mlp_forward
    with torch.cuda.device(xyz.device), _timed(f"sa_mlp_forward(c={c_feat},n={n},np={npoint},mlp={c1}-{c2}-{c3},{precision})",
                                                algo, 3 * flops if x3 else flops, "bf16" if x3 else "fp32"):
        st = fn(b, n, npoint, nsample, c_feat, c1, c2, c3,
                                               xyz.data_ptr(), new_xyz.data_ptr(), features.data_ptr(),
                                               idx.data_ptr(), wpack.data_ptr(), out.data_ptr(), _stream())
    _native.check(st, "sa_mlp_forward")
    return out
