"""HBM-resident scenes + the loader's per-object processing as ONE launch of libgps_hip.so.

Reference (what this replaces on the data-loader workers):
    ScanBase._load_scan            data/datasets/base.py:65-142   per scan: pcds = [points | colors/127.5-1],
                                                                  obj_pcds = [pcds[instance_labels == id] ...]
    ScanBase._obj_processing_post  data/datasets/base.py:697-740  rotate, obj_locs, box, subsample, normalise
    dataset wrapper padding        data/datasets/dataset_wrapper.py:62-70  pad to max_obj_len, obj_masks

MI355X form: every scan is uploaded ONCE in its raw on-disk precision (xyz f32 + rgb u8, packed as
16-byte records, instead of the loader's 48 B/point float64 rows), points regrouped so that each kept instance is
contiguous (same within-instance order as `pcds[mask]`), with a CSR offset table.  A training batch
is then described by a (B, max_obj_len) table of object ids; `obj_processing_post` turns it into the
model's `obj_fts / obj_locs / obj_masks` (+ boxes) on the device -- no per-object host work, no
126 MB/step host-to-device copy.  GPU only: there is no CPU path here (the reference's own loader IS
the CPU path; oracle/obj_processing.py restates it for the tests).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _native


class PackedScans:
    """Raw scans resident on one device.  Build with add_scan(...) x n, then finalize()."""

    def __init__(self, device: torch.device | str = "cuda", records: bool = True):
        """records=True (default, uint8 colours only): store points as 16-byte records {x, y, z f32; r, g, b
        u8; pad} -- one aligned vector load per point, colours gathered with the coordinates.
        records=False: xyz (N,3) and rgb (N,3) as two arrays (also the layout for float32 colours)."""
        self.device = torch.device(device)
        self.records = records
        self._xyz: List[np.ndarray] = []
        self._rgb: List[np.ndarray] = []
        self._sizes: List[int] = []
        self._scan_objs: Dict[str, range] = {}
        self._inst_ids: Dict[str, list] = {}
        self.xyz = self.rgb = self.obj_offsets = None
        self.sizes_host: Optional[np.ndarray] = None

    def add_scan(self, scan_id: str, points: np.ndarray, colors: np.ndarray, instance_labels: np.ndarray,
                 inst_ids: Sequence[int]) -> None:
        """points (N,3) float32, colors (N,3) uint8 or float32 in 0..255 (the `.pth` contents of
        base.py:70-73), instance_labels (N,), inst_ids: the instances the loader keeps, in its order
        (base.py:82-89; instances without points are skipped exactly like `np.sum(mask) == 0`)."""
        assert self.xyz is None, "finalize() was already called"
        points = np.ascontiguousarray(points, dtype=np.float32)
        if colors.dtype != np.uint8:
            colors = np.ascontiguousarray(colors, dtype=np.float32)
        if self._rgb and self._rgb[0].dtype != colors.dtype:
            raise ValueError("all scans must store colours in the same dtype")
        first = len(self._sizes)
        kept = []
        for inst in inst_ids:
            sel = np.flatnonzero(instance_labels == inst)
            if sel.size == 0:
                continue
            self._xyz.append(points[sel])
            self._rgb.append(np.ascontiguousarray(colors[sel]))
            self._sizes.append(int(sel.size))
            kept.append(inst)
        self._scan_objs[scan_id] = range(first, len(self._sizes))
        self._inst_ids[scan_id] = kept

    def finalize(self) -> "PackedScans":
        sizes = np.asarray(self._sizes, dtype=np.int64)
        off = np.zeros(len(sizes) + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        self.sizes_host = sizes
        xyz = np.concatenate(self._xyz, 0) if self._xyz else np.zeros((0, 3), np.float32)
        rgb = np.concatenate(self._rgb, 0) if self._rgb else np.zeros((0, 3), np.uint8)
        if self.records and rgb.dtype == np.uint8:
            rec = np.zeros((xyz.shape[0], 4), dtype=np.float32)
            rec[:, :3] = xyz
            rec.view(np.uint8).reshape(-1, 16)[:, 12:15] = rgb
            self.xyz, self.rgb = torch.from_numpy(rec).to(self.device), None
        else:
            self.records = False
            self.xyz, self.rgb = torch.from_numpy(xyz).to(self.device), torch.from_numpy(rgb).to(self.device)
        self.obj_offsets = torch.from_numpy(off).to(self.device)
        self._xyz, self._rgb = [], []
        return self

    def scan_objects(self, scan_id: str) -> range:
        """Global object ids of a scan, in the loader's object order."""
        return self._scan_objs[scan_id]

    def scan_inst_ids(self, scan_id: str) -> list:
        return self._inst_ids[scan_id]

    @property
    def n_objects(self) -> int:
        return 0 if self.sizes_host is None else int(self.sizes_host.shape[0])


def batch_rows(packed: PackedScans, scan_ids: Sequence[str], max_obj_len: int,
               obj_select: Optional[Sequence[Sequence[int]]] = None) -> torch.Tensor:
    """(B, max_obj_len) int32 table of global object ids, -1 = padding slot.  obj_select[b] optionally
    lists scene-local object positions (the loader's selected_obj_idxs, base.py:236-246)."""
    rows = np.full((len(scan_ids), max_obj_len), -1, dtype=np.int32)
    for b, sid in enumerate(scan_ids):
        objs = list(packed.scan_objects(sid))
        if obj_select is not None:
            objs = [objs[i] for i in obj_select[b]]
        if len(objs) > max_obj_len:
            raise ValueError(f"scan {sid}: {len(objs)} objects > max_obj_len {max_obj_len}")
        rows[b, :len(objs)] = objs
    return torch.from_numpy(rows)


def obj_processing_post(packed: PackedScans, row_obj: torch.Tensor, num_points: int = 1024,
                        rot: Optional[torch.Tensor] = None, sample_idx: Optional[torch.Tensor] = None,
                        seed: int = 0, need_boxes: bool = False, out: Optional[dict] = None) -> dict:
    """row_obj (B, O) int32 object ids (-1 = padding) -> dict with obj_fts (B,O,num_points,6) f32,
    obj_locs (B,O,6) f32, obj_masks (B,O) bool [, obj_boxes (B,O,6) f32].

    rot: per-scene rotations (build_rotate_mat's matrix): a (B,3,3) tensor/array, or a length-B list
    with None for the scenes it left unrotated; None = no rotation at all.
    sample_idx: (B,O,num_points) int32 object-local indices (np.random.choice draws, for bit-compatible
    replays of the reference loader) or None = drawn on the device from `seed`.
    out: optional {"obj_fts", "obj_locs", "obj_masks"} of preallocated contiguous tensors of those shapes (obj_masks bool
    or uint8) the kernel writes IN PLACE -- the static input buffers of a captured training step
    (`GPSTrainStep.static_inputs()`), so that no per-step copy of the 126 MB of object points is needed."""
    if packed.xyz is None:
        raise RuntimeError("PackedScans.finalize() has not been called")
    dev = packed.xyz.device
    if dev.type != "cuda":
        raise RuntimeError("obj_processing_post runs on libgps_hip.so: scenes must live on a GPU (no CPU path)")
    B, O = row_obj.shape
    n_rows = B * O
    nbytes = _algorithmic_bytes(packed, row_obj if row_obj.device.type == "cpu" else None, n_rows, num_points)
    row_obj = row_obj.to(device=dev, dtype=torch.int32).contiguous()
    if out is not None:
        fts, locs, masks = out["obj_fts"], out["obj_locs"], out["obj_masks"]
        ok = (fts.shape == (B, O, num_points, 6) and fts.dtype == torch.float32 and locs.shape == (B, O, 6)
              and locs.dtype == torch.float32 and masks.shape == (B, O) and masks.dtype in (torch.bool, torch.uint8)
              and all(t.is_contiguous() and t.device == dev for t in (fts, locs, masks)))
        if not ok:
            raise ValueError("obj_processing_post: `out` tensors must be contiguous (B,O,P,6) f32 / (B,O,6) f32 / (B,O) bool on the scans' GPU")
    else:
        fts = torch.empty((B, O, num_points, 6), dtype=torch.float32, device=dev)
        locs = torch.empty((B, O, 6), dtype=torch.float32, device=dev)
        masks = torch.empty((B, O), dtype=torch.uint8, device=dev)
    boxes = torch.empty((B, O, 6), dtype=torch.float32, device=dev) if need_boxes else None
    rot_ptr = row_rot_ptr = None
    keep = []
    if rot is not None:
        rot_t, row_rot = rot_rows(rot, B, O, dev)
        keep = [rot_t, row_rot]
        rot_ptr, row_rot_ptr = rot_t.data_ptr(), row_rot.data_ptr()
    if sample_idx is not None:
        sample_idx = sample_idx.to(device=dev, dtype=torch.int32).contiguous()
        assert sample_idx.shape == (B, O, num_points), sample_idx.shape
    from ..pointnet2._ext import _timed
    with torch.cuda.device(dev), _timed(f"obj_processing_post(rows={n_rows},P={num_points})", nbytes):
        st = _native.load().gps_obj_processing_post(
            n_rows, num_points, packed.xyz.data_ptr(), packed.rgb.data_ptr() if packed.rgb is not None else None,
            int(packed.rgb is None or packed.rgb.dtype == torch.uint8),
            packed.obj_offsets.data_ptr(), row_obj.data_ptr(),
            sample_idx.data_ptr() if sample_idx is not None else None, int(seed) & ((1 << 64) - 1),
            rot_ptr, row_rot_ptr, fts.data_ptr(), locs.data_ptr(), boxes.data_ptr() if need_boxes else None,
            masks.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _native.check(st, "obj_processing_post")
    del keep
    res = {"obj_fts": fts, "obj_locs": locs, "obj_masks": masks if masks.dtype == torch.bool else masks.bool()}
    if need_boxes:
        res["obj_boxes"] = boxes
    return res


def rot_rows(rot, B: int, O: int, dev):
    """Per-scene rotations -> (matrices (n,3,3) f32 on dev, row_rot (B*O) int32 with -1 = unrotated).
    `rot` is a (B,3,3) tensor/array, or a list with None for scenes build_rotate_mat left unrotated."""
    mats, row_rot = [], np.full((B, O), -1, dtype=np.int32)
    for b in range(B):
        r = rot[b]
        if r is None:
            continue
        row_rot[b, :] = len(mats)
        mats.append(np.asarray(r.cpu() if torch.is_tensor(r) else r, dtype=np.float32).reshape(3, 3))
    if not mats:
        mats = [np.eye(3, dtype=np.float32)]
    return (torch.from_numpy(np.stack(mats, 0)).to(dev).contiguous(),
            torch.from_numpy(row_rot.reshape(-1)).to(dev))


def _algorithmic_bytes(packed: PackedScans, row_obj_host, n_rows: int, num_points: int) -> int:
    """Each object's raw points read once (16-byte records; 15 or 24 B/point as two arrays) + the sampled
    points gathered + the f32 feature rows written.  With the row table already on the device the object
    sizes of THIS batch are not known on the host without a sync: the mean object size stands in."""
    per_pt = 16 if packed.rgb is None else 12 + 3 * packed.rgb.element_size()
    if row_obj_host is not None:
        ids = row_obj_host.reshape(-1).numpy()
        k_total = int(packed.sizes_host[ids[ids >= 0]].sum())
    else:
        k_total = int(packed.sizes_host.mean() * n_rows) if packed.n_objects else 0
    return k_total * per_pt + n_rows * num_points * (per_pt + 24) + n_rows * (4 + 24 + 1)
