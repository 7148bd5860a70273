"""ctypes front-end of the CPU oracle (oracle/pointnet2_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from sceneverse_amd/.

`OracleExt` mirrors the nine functions of the reference's pybind module
`pointnet2._ext` (/root/reference/modules/third_party/pointnet2/_ext_src/src/bindings.cpp:6-19)
on CPU torch tensors, with the reference's argument order, dtype/contiguity checks
(_ext_src/include/utils.h:5-25) and return conventions (host wrappers allocate the
outputs: sampling.cpp:15-87, ball_query.cpp:8-32, group_points.cpp:12-62,
interpolate.cpp:14-99).  It can therefore be injected as `pointnet2_utils._ext`
into the reference's own Python -- that is how the golden
fixtures of tests/golden/ are produced and what the CPU baseline times.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpointnet2_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int32)
_i = ctypes.c_int


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_opt_n_threads.argtypes = [_i]
        L.oracle_opt_n_threads.restype = _i
        L.oracle_gather_points.argtypes = [_i, _i, _i, _i, _F, _I, _F]
        L.oracle_gather_points_grad.argtypes = [_i, _i, _i, _i, _F, _I, _F]
        L.oracle_furthest_point_sampling.argtypes = [_i, _i, _i, _F, _I]
        L.oracle_ball_query.argtypes = [_i, _i, _i, ctypes.c_float, _i, _F, _F, _I]
        L.oracle_group_points.argtypes = [_i, _i, _i, _i, _i, _F, _I, _F]
        L.oracle_group_points_grad.argtypes = [_i, _i, _i, _i, _i, _F, _I, _F]
        L.oracle_group_points_grad_f64.argtypes = [_i, _i, _i, _i, _i, _F, _I, _D]
        L.oracle_three_nn.argtypes = [_i, _i, _i, _F, _F, _F, _I]
        L.oracle_three_interpolate.argtypes = [_i, _i, _i, _i, _F, _I, _F, _F]
        L.oracle_three_interpolate_grad.argtypes = [_i, _i, _i, _i, _F, _I, _F, _F]
        _lib = L
    return _lib


def opt_n_threads(work_size: int) -> int:
    return int(lib().oracle_opt_n_threads(int(work_size)))


def _fp(t: torch.Tensor):
    return ctypes.cast(t.data_ptr(), _F)


def _ip(t: torch.Tensor):
    return ctypes.cast(t.data_ptr(), _I)


def _check(t: torch.Tensor, name: str, dtype: torch.dtype):
    # _ext_src/include/utils.h:5-25: AT_ASSERT -> RuntimeError in Python.
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = "a float" if dtype == torch.float32 else "an int"
        raise RuntimeError(f"{name} must be {kind} tensor")
    if t.device.type != "cpu":
        raise RuntimeError(f"{name}: the oracle runs on CPU tensors only")


class OracleExt:
    """Drop-in stand-in for `pointnet2._ext` running on one CPU thread."""

    @staticmethod
    def gather_points(points, idx):
        _check(points, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        b, c, n = points.shape
        m = idx.shape[1]
        out = torch.zeros(b, c, m, dtype=torch.float32)
        lib().oracle_gather_points(b, c, n, m, _fp(points), _ip(idx), _fp(out))
        return out

    @staticmethod
    def gather_points_grad(grad_out, idx, n):
        _check(grad_out, "grad_out", torch.float32)
        _check(idx, "idx", torch.int32)
        b, c, m = grad_out.shape
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().oracle_gather_points_grad(b, c, int(n), m, _fp(grad_out), _ip(idx), _fp(out))
        return out

    @staticmethod
    def furthest_point_sampling(points, nsamples):
        _check(points, "points", torch.float32)
        b, n, _ = points.shape
        out = torch.zeros(b, int(nsamples), dtype=torch.int32)
        lib().oracle_furthest_point_sampling(b, n, int(nsamples), _fp(points), _ip(out))
        return out

    @staticmethod
    def three_nn(unknowns, knows):
        _check(unknowns, "unknowns", torch.float32)
        _check(knows, "knows", torch.float32)
        b, n, _ = unknowns.shape
        m = knows.shape[1]
        idx = torch.zeros(b, n, 3, dtype=torch.int32)
        dist2 = torch.zeros(b, n, 3, dtype=torch.float32)
        lib().oracle_three_nn(b, n, m, _fp(unknowns), _fp(knows), _fp(dist2), _ip(idx))
        return [dist2, idx]

    @staticmethod
    def three_interpolate(points, idx, weight):
        _check(points, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        _check(weight, "weight", torch.float32)
        b, c, m = points.shape
        n = idx.shape[1]
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().oracle_three_interpolate(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
        return out

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m):
        _check(grad_out, "grad_out", torch.float32)
        _check(idx, "idx", torch.int32)
        _check(weight, "weight", torch.float32)
        b, c, n = grad_out.shape
        out = torch.zeros(b, c, int(m), dtype=torch.float32)
        lib().oracle_three_interpolate_grad(b, c, n, int(m), _fp(grad_out), _ip(idx), _fp(weight),
                                            _fp(out))
        return out

    @staticmethod
    def ball_query(new_xyz, xyz, radius, nsample):
        _check(new_xyz, "new_xyz", torch.float32)
        _check(xyz, "xyz", torch.float32)
        b, m, _ = new_xyz.shape
        n = xyz.shape[1]
        idx = torch.zeros(b, m, int(nsample), dtype=torch.int32)
        lib().oracle_ball_query(b, n, m, float(radius), int(nsample), _fp(new_xyz), _fp(xyz),
                                _ip(idx))
        return idx

    @staticmethod
    def group_points(points, idx):
        _check(points, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        b, c, n = points.shape
        _, npoints, nsample = idx.shape
        out = torch.zeros(b, c, npoints, nsample, dtype=torch.float32)
        lib().oracle_group_points(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out))
        return out

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        _check(grad_out, "grad_out", torch.float32)
        _check(idx, "idx", torch.int32)
        b, c, npoints, nsample = grad_out.shape
        out = torch.zeros(b, c, int(n), dtype=torch.float32)
        lib().oracle_group_points_grad(b, c, int(n), npoints, nsample, _fp(grad_out), _ip(idx),
                                       _fp(out))
        return out

    @staticmethod
    def group_points_grad_f64(grad_out, idx, n):
        """Order-free reference sum (double accumulation); not part of `_ext`."""
        b, c, npoints, nsample = grad_out.shape
        out = torch.zeros(b, c, int(n), dtype=torch.float64)
        lib().oracle_group_points_grad_f64(b, c, int(n), npoints, nsample, _fp(grad_out),
                                           _ip(idx), ctypes.cast(out.data_ptr(), _D))
        return out


# ---------------------------------------------------------------------------
# Independent numpy restatements, used by tests to cross-check the C oracle
# (closed-form FPS tie rule of SURVEY.md App. B.1; vectorised ball query).
# ---------------------------------------------------------------------------

def _bitrev(x: np.ndarray, bits: int) -> np.ndarray:
    r = np.zeros_like(x)
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def fps_closed_form(xyz: np.ndarray, m: int) -> np.ndarray:
    """FPS with the closed-form tie rule: among equal maxima pick the point minimising
    (bitreverse_{log2 bs}(k mod bs), k div bs), bs = opt_n_threads(n)
    (sampling_gpu.cu:94-172 analysed in SURVEY.md App. B.1)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    b, n, _ = xyz.shape
    bs = opt_n_threads(n)
    bits = int(np.log2(bs))
    k = np.arange(n)
    key = _bitrev(k % bs, bits).astype(np.int64) * (n // bs + 1) + k // bs
    out = np.zeros((b, m), dtype=np.int32)
    for bi in range(b):
        p = xyz[bi]
        mag = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]
        valid = mag.astype(np.float64) > 1e-3
        temp = np.full(n, 1e10, dtype=np.float32)
        old = 0
        for j in range(1, m):
            d = p - p[old]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            temp = np.where(valid, np.minimum(d2, temp), temp)
            if not valid.any():
                old = 0
            else:
                best = temp[valid].max()
                cand = np.nonzero(valid & (temp == best))[0]
                old = int(cand[np.argmin(key[cand])])
            out[bi, j] = old
    return out


def ball_query_numpy(new_xyz: np.ndarray, xyz: np.ndarray, radius: float, nsample: int) -> np.ndarray:
    new_xyz = np.asarray(new_xyz, dtype=np.float32)
    xyz = np.asarray(xyz, dtype=np.float32)
    b, m, _ = new_xyz.shape
    r2 = np.float32(radius) * np.float32(radius)
    out = np.zeros((b, m, nsample), dtype=np.int32)
    for bi in range(b):
        d = new_xyz[bi][:, None, :] - xyz[bi][None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        hit = d2 < r2
        for j in range(m):
            ks = np.nonzero(hit[j])[0][:nsample]
            if ks.size:
                out[bi, j, :] = ks[0]
                out[bi, j, : ks.size] = ks
    return out
