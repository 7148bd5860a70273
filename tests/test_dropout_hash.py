"""Statistical sanity of the counter-based dropout stream of the attention / LayerNorm / GEMM kernels
(gps_attention.hip `mix32` / `pair_rng`, same function in gps_layernorm.hip and gps_gemm.hip), emulated in numpy:
keep rate = 1 - p, the two 16-bit halves of a pair hash independent, neighbouring indices and neighbouring seeds
uncorrelated.  The kernels' own tests check that forward and backward draw the same mask; this one checks that the
mask is a fair coin."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x21F0AAAD)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x735A2D97)) & M32
    x ^= x >> np.uint64(15)
    return x


def seed_fold(seed):
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    return mix32(np.array([lo], dtype=np.uint64) ^ mix32(np.array([(hi + np.uint64(0x9E3779B9)) & M32], dtype=np.uint64)))[0]


def test_keep_rate_and_independence():
    n = 1 << 20
    idx = np.arange(n, dtype=np.uint64)
    for seed in (0, 1, 123456789, (7 << 32) + 5):
        h = mix32(idx ^ seed_fold(seed))
        lo, hi = h & np.uint64(0xFFFF), h >> np.uint64(16)
        for p in (0.1, 0.3):
            thr = np.uint64(int(p * 65536))
            k_lo, k_hi = (lo >= thr), (hi >= thr)
            sigma = (p * (1 - p) / n) ** 0.5
            assert abs(k_lo.mean() - (1 - p)) < 5 * sigma + 2e-5 and abs(k_hi.mean() - (1 - p)) < 5 * sigma + 2e-5
            # the two keys of a pair, and neighbouring pairs, are independent draws
            c_pair = np.corrcoef(k_lo, k_hi)[0, 1]
            c_next = np.corrcoef(k_lo[:-1], k_lo[1:])[0, 1]
            assert abs(c_pair) < 6e-3 and abs(c_next) < 6e-3, (c_pair, c_next)
        # full 32-bit thresholds (LayerNorm / GEMM element stream)
        keep = h >= np.uint64(int(0.1 * 4294967296.0))
        assert abs(keep.mean() - 0.9) < 5 * (0.09 / n) ** 0.5


def test_streams_of_consecutive_seeds_are_uncorrelated():
    n = 1 << 18
    idx = np.arange(n, dtype=np.uint64)
    thr = np.uint64(int(0.1 * 4294967296.0))
    a = mix32(idx ^ seed_fold(1000)) >= thr
    b = mix32(idx ^ seed_fold(1001)) >= thr
    assert abs(np.corrcoef(a, b)[0, 1]) < 1e-2
    assert (a != b).mean() > 0.15          # two independent p = 0.1 masks differ on 2 * 0.1 * 0.9 = 18 % of the elements
