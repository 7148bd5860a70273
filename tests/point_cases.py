"""Seeded inputs for the point-op parity tests (shared by CPU and GPU tests and by
tests/golden/make_golden_gpu.py)."""
import numpy as np
import torch

from sceneverse_amd.data.synthetic import adversarial_objects, synth_object


def object_clouds(n_obj, n_pts, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(np.stack([synth_object(rng, n_pts)[:, :3] for _ in range(n_obj)]))


def sa1_cloud(seed=0):
    """(b, 1024, 3): adversarial objects + synthetic objects + a padding object."""
    return torch.cat([adversarial_objects(1024), object_clouds(10, 1024, seed)], 0).contiguous()


def generic_cloud(b, n, seed, scale=0.4):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, n, 3, generator=g) * scale
    if n >= 8:
        x[0, n // 2:] = x[0, : n - n // 2].clone()   # duplicates -> ties
        x[-1, ::3] *= 0.02                          # many near-origin points
    return x.contiguous()


# (n, m) FPS shapes: GPS SA1 / SA2, odd sizes, the 32-register path, the streaming path
FPS_SHAPES = [(1024, 32), (32, 16), (1, 1), (2, 2), (5, 5), (31, 7), (33, 9), (63, 20), (64, 64),
              (100, 13), (127, 127), (300, 65), (512, 40), (1000, 130), (1500, 24), (2048, 64),
              (2049, 16), (3000, 20)]

# (n, m, radius, nsample)
BQ_SHAPES = [(1024, 32, 0.2, 32), (32, 16, 0.4, 32), (1, 1, 0.5, 3), (100, 7, 0.3, 5),
             (63, 5, 0.25, 64), (1000, 33, 0.15, 16), (2048, 16, 0.2, 48), (3000, 9, 0.1, 20),
             (130, 4, 1e-6, 8), (130, 4, 100.0, 200)]

# (c, n, npoint, nsample)
GROUP_SHAPES = [(3, 1024, 32, 32), (3, 32, 16, 32), (128, 32, 16, 32), (1, 7, 3, 5), (5, 100, 7, 9),
                (131, 33, 4, 6), (2, 20000, 5, 8), (64, 16, 1, 1)]

# BASELINE configs[4] ("stress"): 2048 points per object.  Kept apart from the lists above, whose entries key the
# committed reference-kernel fixture (tests/golden/point_ops_ref_gpu.pt); these run against the oracle and, on the
# GPU box, against the reference's own kernels (oracle/_ref).
STRESS_FPS_SHAPES = [(2048, 32)]
STRESS_BQ_SHAPES = [(2048, 32, 0.2, 32)]
STRESS_GROUP_SHAPES = [(3, 2048, 32, 32)]
