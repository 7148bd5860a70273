"""Column sums of a bf16 matrix on libgps_hip.so (gps_colsum_bf16): the bias gradient `dY.sum(0)` of a Linear
whose weight gradient does NOT go through gps_gemm_bf16's TN form (that form produces it from the same tiles)."""
from __future__ import annotations

import torch

from .. import _native


def colsum_bf16(dy2: torch.Tensor) -> torch.Tensor:
    """(T, N) bf16 -> (N,) fp32 column sums, fp32 accumulation in a fixed order (deterministic)."""
    lib = _native.load()
    T, N = dy2.shape
    out = torch.empty(N, dtype=torch.float32, device=dy2.device)
    parts = lib.gps_colsum_parts(T, N)
    scratch = torch.empty((max(parts, 1), N), dtype=torch.float32, device=dy2.device)
    from ..pointnet2._ext import _timed
    with torch.cuda.device(dy2.device), _timed(f"colsum_bf16(rows={T},cols={N})", 2 * T * N + 4 * N):
        st = lib.gps_colsum_bf16(T, N, dy2.data_ptr(), dy2.stride(0), scratch.data_ptr(), out.data_ptr(),
                                 torch.cuda.current_stream(dy2.device).cuda_stream)
    if st == _native.GPS_ERR_UNSUPPORTED:            # columns / pitch not a multiple of 8: not this kernel's shape
        return dy2.sum(0, dtype=torch.float32)
    _native.check(st, "colsum_bf16")
    return out
