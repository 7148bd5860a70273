"""Host emulation of the GEMM kernel's LDS layout (sceneverse_amd/csrc/gps_gemm_layout.h): the SAME index
functions the device code uses are compiled with g++ and driven here.

  stage map (global_load_lds: piece base + lane * 16, per-lane source)  ->  LDS image
  fragment reads (ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16 lane semantics)  ->  which (row, k) a lane gets
  checks: every fragment element e of lane (i, g) is  tile[r0 + i][32 ks + frag_k(g, e)]  with ONE frag_k for
  both operand kinds (what the MFMA needs), the images are bijective, and no read is bank-conflicted
  under the gfx950 banking rules of /opt/skills/guides/MI355X_MICROARCH.md (LDS table).
"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    out = os.path.join(tempfile.mkdtemp(prefix="gemm_layout_"), "shim.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "sceneverse_amd", "csrc"),
                           os.path.join(ROOT, "tools", "gemm_layout_shim.cpp"), "-o", out])
    return ctypes.CDLL(out)


def _km_image(shim, rows):
    """LDS image (uint16 element ids) of a K-major [rows][64] tile: element id = row * 64 + k."""
    img = np.full(rows * 64, -1, dtype=np.int64)
    r, c = ctypes.c_int(), ctypes.c_int()
    for q in range(rows // 8):
        for lane in range(64):
            shim.shim_km_stage_src(q, lane, ctypes.byref(r), ctypes.byref(c))
            dst = (q * 1024 + lane * 16) // 2
            assert 0 <= r.value < rows and 0 <= c.value < 8
            img[dst:dst + 8] = r.value * 64 + c.value * 8 + np.arange(8)
    assert sorted(img.tolist()) == list(range(rows * 64))          # bijective
    return img


def _rm_image(shim, cols):
    """LDS image of a reduction-major [64 k][cols] tile: element id = k * cols + col."""
    img = np.full(64 * cols, -1, dtype=np.int64)
    k, c = ctypes.c_int(), ctypes.c_int()
    for q in range(cols // 8):
        for lane in range(64):
            shim.shim_rm_stage_src(cols, q, lane, ctypes.byref(k), ctypes.byref(c))
            dst = (q * 1024 + lane * 16) // 2
            assert 0 <= k.value < 64 and 0 <= c.value < cols // 8
            img[dst:dst + 8] = k.value * cols + c.value * 8 + np.arange(8)
    assert sorted(img.tolist()) == list(range(64 * cols))
    return img


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
HALF_GROUPS = [list(range(32)), list(range(32, 64))]


def _max_conflict(addrs, nbytes, groups):
    """Worst number of DISTINCT addresses on one 4-byte bank (64 banks) inside a lane group."""
    worst = 1
    for g in groups:
        per_bank = {}
        for l in g:
            for b in range(addrs[l] // 4, (addrs[l] + nbytes) // 4):
                per_bank.setdefault(b % 64, set()).add(b)
        worst = max(worst, max(len(s) for s in per_bank.values()))
    return worst


@pytest.mark.parametrize("rows", [64, 128, 256])
def test_k_major_fragments(shim, rows):
    img = _km_image(shim, rows)
    for r0 in range(0, rows, 16):
        for ks in range(2):
            addrs = [shim.shim_km_frag(r0 + (l & 15), ks, l >> 4) for l in range(64)]
            assert all(a % 16 == 0 for a in addrs)
            assert _max_conflict(addrs, 16, B128_GROUPS) == 1
            for l in range(64):
                i, g = l & 15, l >> 4
                got = img[addrs[l] // 2: addrs[l] // 2 + 8]
                want = [(r0 + i) * 64 + 32 * ks + shim.shim_frag_k(g, e) for e in range(8)]
                assert got.tolist() == want


@pytest.mark.parametrize("cols", [128, 256, 64])
def test_reduction_major_transposed_fragments(shim, cols):
    """ds_read_b64_tr_b16: inside a 16-lane group, lane i supplies the address of 4 consecutive bf16 and receives
    element (i & 3) of the lanes 4 j + (i >> 2), j = 0..3 (guide: column i of the 4 x 16 block the group fetched)."""
    img = _rm_image(shim, cols)
    for c0 in range(0, cols, 16):
        for ks in range(2):
            for which in range(2):
                addrs = [shim.shim_rm_frag(cols, c0, ks, l, which) for l in range(64)]
                assert all(a % 8 == 0 for a in addrs)
                conflict = _max_conflict(addrs, 8, HALF_GROUPS)
                assert conflict == (1 if cols >= 128 else 2)
                loaded = [img[a // 2: a // 2 + 4] for a in addrs]
                for l in range(64):
                    i, g = l & 15, l >> 4
                    got = [int(loaded[16 * g + 4 * j + (i >> 2)][i & 3]) for j in range(4)]
                    want = [(32 * ks + shim.shim_frag_k(g, 4 * which + j)) * cols + c0 + i for j in range(4)]
                    assert got == want


def test_xcd_virtual_id_is_a_bijection(shim):
    for total in [1, 7, 8, 9, 36, 255, 256, 1170, 4097]:
        ids = sorted(shim.shim_xcd_virtual_id(b, total) for b in range(total))
        assert ids == list(range(total))
        # the blocks of one XCD form one contiguous range
        for x in range(min(8, total)):
            mine = sorted(shim.shim_xcd_virtual_id(b, total) for b in range(x, total, 8))
            assert mine == list(range(mine[0], mine[0] + len(mine)))
