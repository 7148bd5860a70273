"""TEST INFRASTRUCTURE -- CPU oracle of the reference's per-object input processing (numpy).

Restates, operation for operation and in the reference's dtype, what the data loader does to the
objects of one scene before they reach the GPS model:

    ScanBase._obj_processing_post          data/datasets/base.py:697-740
        rotation draw                       data/data_utils.py:163-178 (build_rotate_mat)
    loader's colour scaling                 data/datasets/base.py:74-76   (colors / 127.5 - 1)
    padding to max_obj_len + obj_masks      data/datasets/dataset_wrapper.py:62-70, data_utils.py:345-353

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product
path (sceneverse_amd/data/gpu_objects.py -> libgps_hip.so gps_obj_processing_post) never does.

Pinned: tests/golden/obj_processing_ref.npz holds outputs of the reference's own
`ScanBase._obj_processing_post` (imported unmodified from /root/reference by
tests/golden/make_golden_objproc.py); tests/test_oracle_obj_processing.py requires this restatement
to reproduce them bit for bit (float64), including the reference's RNG draw order.

dtype: numpy promotes `[points f32 | colors/127.5-1]` to float64 when the stored colours are uint8
(and keeps float32 when they are float32); everything downstream runs in that dtype and the loader
casts to float32 at the very end (`.float()`, dataset_wrapper.py:64-69).
"""
from __future__ import annotations

import numpy as np

ROTATE_ANGLES = [0, np.pi / 2, np.pi, np.pi * 3 / 2]


def scale_colors(colors: np.ndarray) -> np.ndarray:
    """base.py:75 -- uint8 in -> float64 out, float32 in -> float32 out (numpy promotion)."""
    return colors / 127.5 - 1


def build_rotate_mat(split: str, rot_aug: bool = True):
    """data_utils.py:163-178, 'axis' mode: ONE np.random.randint(4) draw, always consumed."""
    theta = ROTATE_ANGLES[np.random.randint(len(ROTATE_ANGLES))]
    if theta != 0 and split == "train" and rot_aug:
        return np.array([[np.cos(theta), -np.sin(theta), 0],
                         [np.sin(theta), np.cos(theta), 0],
                         [0, 0, 1]], dtype=np.float32)
    return None


def draw_like_reference(ks, num_points: int, split: str = "train", rot_aug: bool = True):
    """The reference's RNG consumption for one scene (global numpy RNG, seed it first): one rotation
    draw, then one np.random.choice per object in order (base.py:699, 720-721)."""
    rot = build_rotate_mat(split, rot_aug)
    idxs = [np.random.choice(k, size=num_points, replace=k < num_points) for k in ks]
    return rot, idxs


def obj_processing_post(obj_pcds, num_points: int, rot_matrix, sample_idxs, is_need_bbox: bool = False):
    """base.py:697-740 with the random draws passed in.  obj_pcds: list of (k_i, 6) arrays
    [x y z r g b]; returns (obj_fts (n,num_points,6), obj_locs (n,6), obj_boxes (n,6) or (0,))
    in the input dtype.  Like the reference, the rotation is applied to ALL points of the object
    before centre/size are taken; the inputs are not modified (the reference rotates in place)."""
    fts, locs, boxes = [], [], []
    for obj_pcd, idx in zip(obj_pcds, sample_idxs):
        obj_pcd = obj_pcd.copy()
        if rot_matrix is not None:
            obj_pcd[:, :3] = np.matmul(obj_pcd[:, :3], rot_matrix.transpose())
        center = obj_pcd[:, :3].mean(0)
        size = obj_pcd[:, :3].max(0) - obj_pcd[:, :3].min(0)
        locs.append(np.concatenate([center, size], 0))
        if is_need_bbox:
            boxes.append(np.concatenate([(obj_pcd[:, :3].max(0) + obj_pcd[:, :3].min(0)) / 2, size], 0))
        obj_pcd = obj_pcd[idx]
        obj_pcd[:, :3] = obj_pcd[:, :3] - obj_pcd[:, :3].mean(0)
        max_dist = np.max(np.sqrt(np.sum(obj_pcd[:, :3] ** 2, 1)))
        if max_dist < 1e-6:                      # tiny point clouds, i.e. padding
            max_dist = 1
        obj_pcd[:, :3] = obj_pcd[:, :3] / max_dist
        fts.append(obj_pcd)
    return np.stack(fts, 0), np.array(locs), np.array(boxes)


def pad_scene(obj_fts: np.ndarray, obj_locs: np.ndarray, max_obj_len: int):
    """dataset_wrapper.py:62-70: pad to max_obj_len (features with 1.0, locations with 0.0), cast to
    float32, obj_masks = arange(max_obj_len) < n."""
    n = obj_locs.shape[0]
    assert n <= max_obj_len
    f = np.ones((max_obj_len,) + obj_fts.shape[1:], dtype=np.float32)
    l = np.zeros((max_obj_len, obj_locs.shape[1]), dtype=np.float32)
    f[:n] = obj_fts.astype(np.float32)
    l[:n] = obj_locs.astype(np.float32)
    return f, l, np.arange(max_obj_len) < n


def synth_scene(rng: np.random.Generator, n_obj: int, color_dtype=np.uint8, k_choices=None):
    """A synthetic scene in the loader's raw layout: per object (points f32 (k,3), colors (k,3))."""
    objs = []
    for i in range(n_obj):
        if k_choices is not None:
            k = int(k_choices[i % len(k_choices)])
        else:
            k = int(rng.integers(1, 4000))
        centre = rng.uniform([-4, -4, 0], [4, 4, 2.5])
        pts = (centre + rng.normal(size=(k, 3)) * rng.uniform(0.05, 1.0, size=3)).astype(np.float32)
        if color_dtype == np.uint8:
            col = rng.integers(0, 256, size=(k, 3), dtype=np.uint8)
        else:
            col = rng.uniform(0, 255, size=(k, 3)).astype(np.float32)
        objs.append((pts, col))
    return objs


def to_obj_pcds(objs):
    """base.py:74-76: pcds = concatenate([points, colors / 127.5 - 1], 1) per object."""
    return [np.concatenate([p, scale_colors(c)], 1) for p, c in objs]
