"""Synthetic GPS batches with the value distributions of the reference's data path
(data/datasets/base.py:697-740, data/datasets/dataset_wrapper.py:62-77):

  * per scene n_real ~ U{20..max_obj-1} real objects, remaining slots are PADDING: all-ones
    points (pad value 1.0), zero locs, obj_masks False, label -100;
  * a real object: k ~ U{50..4000} base points ~ N(0, diag(sigma^2)), sigma ~ U(0.05,1)^3,
    1024 sampled WITH replacement iff k < 1024 (exact duplicates -> FPS ties), centred, scaled
    into the unit ball, rgb ~ U(-1,1);
  * obj_locs = centre ~ U([-4,4]^2 x [0,2.5]) + size ~ U(0.1,2)^3;
  * tokens [101] + U{1000..30521}^(len-2) + [102], len ~ U{6..max_len}, 0-padded; 15 % masked-LM
    labels, -1 elsewhere; scene captions likewise with length up to 300.

There is no dataset access on the build/bench machines, so this generator IS the input of
bench.py, smoke() and the parity tests.  Everything is driven by one numpy Generator seed.
"""
from __future__ import annotations

import numpy as np
import torch


def synth_object(rng: np.random.Generator, n_pts: int) -> np.ndarray:
    k = int(rng.integers(50, 4001))
    sigma = rng.uniform(0.05, 1.0, size=3)
    base = rng.normal(size=(k, 3)) * sigma
    sel = rng.choice(k, size=n_pts, replace=k < n_pts)
    xyz = base[sel]
    xyz = xyz - xyz.mean(0, keepdims=True)
    xyz = xyz / max(np.sqrt((xyz ** 2).sum(1)).max(), 1e-6)
    rgb = rng.uniform(-1.0, 1.0, size=(n_pts, 3))
    return np.concatenate([xyz, rgb], axis=1).astype(np.float32)


def _tokens(rng: np.random.Generator, batch: int, max_len: int, min_len: int = 6):
    ids = np.zeros((batch, max_len), dtype=np.int64)
    masks = np.zeros((batch, max_len), dtype=np.int64)
    for i in range(batch):
        n = int(rng.integers(min(min_len, max_len), max_len + 1))
        ids[i, 0] = 101
        if n > 2:
            ids[i, 1:n - 1] = rng.integers(1000, 30522, size=n - 2)
        ids[i, n - 1] = 102
        masks[i, :n] = 1
    return ids, masks


def synth_batch(batch: int, n_obj: int = 80, n_pts: int = 1024, txt_len: int = 50,
                scene_txt_len: int = 300, seed: int = 42, n_cls: int = 607,
                min_real: int | None = None, device: str | torch.device = "cpu") -> dict:
    """One `data_dict` as the reference's dataloader would hand to OpenVocab.forward
    (SURVEY.md App. A)."""
    rng = np.random.default_rng(seed)
    obj_fts = np.ones((batch, n_obj, n_pts, 6), dtype=np.float32)      # pad value 1.0
    obj_locs = np.zeros((batch, n_obj, 6), dtype=np.float32)
    obj_masks = np.zeros((batch, n_obj), dtype=bool)
    obj_labels = np.full((batch, n_obj), -100, dtype=np.int64)
    tgt = np.zeros((batch, 1), dtype=np.int64)
    lo = min(20, max(1, n_obj // 4)) if min_real is None else min_real
    for b in range(batch):
        n_real = int(rng.integers(lo, max(lo + 1, n_obj)))            # U{lo..n_obj-1}
        for o in range(n_real):
            obj_fts[b, o] = synth_object(rng, n_pts)
        obj_locs[b, :n_real, 0:2] = rng.uniform(-4, 4, size=(n_real, 2))
        obj_locs[b, :n_real, 2] = rng.uniform(0, 2.5, size=n_real)
        obj_locs[b, :n_real, 3:] = rng.uniform(0.1, 2.0, size=(n_real, 3))
        obj_masks[b, :n_real] = True
        obj_labels[b, :n_real] = rng.integers(0, n_cls, size=n_real)
        tgt[b, 0] = int(rng.integers(0, n_real))
    txt_ids, txt_masks = _tokens(rng, batch, txt_len)
    mlm = np.full((batch, txt_len), -1, dtype=np.int64)
    pick = (rng.uniform(size=(batch, txt_len)) < 0.15) & (txt_masks == 1)
    pick[:, 0] = False
    mlm[pick] = txt_ids[pick]
    if not pick.any():
        mlm[0, 1] = txt_ids[0, 1]
    txt_ids_in = txt_ids.copy()
    txt_ids_in[mlm != -1] = 103                                        # [MASK]
    scene_ids, scene_masks = _tokens(rng, batch, scene_txt_len, min_len=min(30, scene_txt_len))
    obj_sem_masks = rng.uniform(size=(batch, n_obj)) > 0.25
    d = {
        "obj_fts": torch.from_numpy(obj_fts),
        "obj_locs": torch.from_numpy(obj_locs),
        "obj_masks": torch.from_numpy(obj_masks),
        "obj_sem_masks": torch.from_numpy(obj_sem_masks),
        "obj_labels": torch.from_numpy(obj_labels),
        "txt_ids": torch.from_numpy(txt_ids_in),
        "txt_masks": torch.from_numpy(txt_masks),
        "masked_lm_labels": torch.from_numpy(mlm),
        "scene_txt_ids": torch.from_numpy(scene_ids),
        "scene_txt_masks": torch.from_numpy(scene_masks),
        "tgt_object_id": torch.from_numpy(tgt),
    }
    return {k: v.to(device) for k, v in d.items()}


def adversarial_objects(n_pts: int = 1024, seed: int = 7) -> torch.Tensor:
    """(6, n_pts, 3) xyz clouds that stress the bit-exactness rules of SURVEY.md App. B:
    0 heavy duplicates (FPS ties), 1 many points inside |p|^2 <= 1e-3 (FPS skip rule),
    2 all points identical (padding object), 3 points on a lattice with spacing == radius
    (d2 == r2 boundary, strict <), 4 all points at the origin (everything skipped),
    5 plain Gaussian."""
    rng = np.random.default_rng(seed)
    out = np.zeros((6, n_pts, 3), dtype=np.float32)
    base = (rng.normal(size=(37, 3)) * 0.3).astype(np.float32)
    out[0] = base[rng.integers(0, 37, size=n_pts)]
    near = (rng.normal(size=(n_pts, 3)) * 0.3).astype(np.float32)
    near[::3] *= 0.05
    near[5] = np.float32(np.sqrt(1e-3 / 3.0))        # |p|^2 right at the threshold
    out[1] = near
    out[2] = 1.0
    g = np.stack(np.meshgrid(*[np.arange(-5, 6)] * 3, indexing="ij"), -1).reshape(-1, 3)
    lat = (g[rng.integers(0, g.shape[0], size=n_pts)] * 0.2).astype(np.float32)
    out[3] = lat
    out[4] = 0.0
    out[5] = (rng.normal(size=(n_pts, 3)) * 0.4).astype(np.float32)
    return torch.from_numpy(out)
