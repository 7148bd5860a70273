"""gps_obj_processing_post (sceneverse_amd/data/gpu_objects.py -> libgps_hip.so) against
  * the reference's own ScanBase._obj_processing_post outputs (tests/golden/obj_processing_ref.npz),
  * the numpy oracle (oracle/obj_processing.py) at the training shapes,
  * size-independent properties at the full batch size (unit ball, zero mean, padding, distinct draws).
Tolerance: the kernel computes in float64 like the reference and rounds to float32 at the end; only
the summation order of the two means differs -> at most 1 float32 ulp (1.2e-7 relative, 1e-7 absolute
for the all-identical-object residue), and > 99.9 % of the elements bit-equal.  For float32-stored
colours the REFERENCE runs in float32 (numpy promotion) and the kernel is the more accurate one: 1e-5
against the reference, 1 ulp against a float64 evaluation of the same formulas."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

from oracle import obj_processing as O  # noqa: E402
from make_golden_objproc import CASES  # noqa: E402
from sceneverse_amd.data import gpu_objects as G  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj_processing_ref.npz"))


def pack(objs, shuffle_seed=0, scan_id="s0", packed=None, records=True):
    """Objects -> one raw scan (points interleaved across instances, as on disk) -> PackedScans."""
    pts = np.concatenate([p for p, _ in objs], 0)
    col = np.concatenate([c for _, c in objs], 0)
    lab = np.concatenate([np.full(len(p), 100 + i) for i, (p, _) in enumerate(objs)])
    # scatter the instances over the "disk" order while keeping each instance's own point order
    # (which is what `pcds[instance_labels == id]` preserves)
    lab_p = lab[np.argsort(np.random.default_rng(shuffle_seed).random(len(lab)), kind="stable")]
    out_idx = np.empty(len(lab), dtype=np.int64)
    for i in range(len(objs)):
        out_idx[np.flatnonzero(lab_p == 100 + i)] = np.flatnonzero(lab == 100 + i)
    own = packed is None
    packed = packed or G.PackedScans(DEV, records=records)
    packed.add_scan(scan_id, pts[out_idx], col[out_idx], lab[out_idx], [100 + i for i in range(len(objs))] + [999])
    return packed.finalize() if own else packed


def case_objs(case):
    name, scene_seed, np_seed, n_obj, num_points, cdt, split, ks = case
    objs = O.synth_scene(np.random.default_rng(scene_seed), n_obj, np.dtype(cdt).type, ks)
    objs[1] = (np.repeat(objs[1][0][:1], len(objs[1][0]), 0), objs[1][1])
    return objs


def close_f32(got, ref64, rel=1.2e-7, ab=1e-7):
    ref = ref64.astype(np.float32)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert np.all(err <= ab + rel * np.abs(ref)), float(err.max())
    big = np.abs(ref) > 1e-9          # the all-identical object's xyz is a 1e-16 rounding residue: not counted
    return float(np.mean(got[big] == ref[big]))


@pytest.mark.parametrize("records", [True, False], ids=["rec16", "two_arrays"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_kernel_matches_reference_loader_outputs(case, records):
    name, scene_seed, np_seed, n_obj, num_points, cdt, split, ks = case
    objs = case_objs(case)
    packed = pack(objs, shuffle_seed=scene_seed, records=records)
    assert packed.records == (records and cdt == "uint8")
    assert packed.n_objects == n_obj and list(packed.sizes_host) == [len(p) for p, _ in objs]
    np.random.seed(np_seed)
    rot, idxs = O.draw_like_reference(ks, num_points, split, True)
    O_pad = n_obj + 3
    rows = G.batch_rows(packed, ["s0"], O_pad)
    sidx = torch.zeros((1, O_pad, num_points), dtype=torch.int32)
    sidx[0, :n_obj] = torch.from_numpy(np.stack(idxs, 0).astype(np.int32))
    out = G.obj_processing_post(packed, rows, num_points, rot=[rot], sample_idx=sidx, need_boxes=True)
    fts, locs, boxes = (out[k][0].cpu().numpy() for k in ("obj_fts", "obj_locs", "obj_boxes"))
    # float32-stored colours: the reference's own float32 rounding (|x| ~ 4 -> 5e-7 per subtraction, divided
    # by a max_dist of ~0.1) is worth up to ~5e-6 -- measured 4.6e-6 between the reference and a float64
    # evaluation of the same formulas; the kernel is additionally held to 1 ulp of that float64 evaluation
    tol = dict(rel=1.2e-7, ab=1e-7) if cdt == "uint8" else dict(rel=1e-5, ab=1e-5)
    ref_fts = GOLD[f"{name}/fts"].copy()
    if cdt != "uint8":
        # float32 colours make the reference run in float32, where the all-identical object's x - mean is
        # rounding noise of ~1 ulp(|x|) that can exceed the 1e-6 threshold and then gets scaled to the unit
        # ball; in float64 it is ~1e-16 and stays (correctly) ~0.  Not a parity target: compare to 0.
        assert np.all(np.abs(fts[1, :, :3]) < 1e-6)
        ref_fts[1, :, :3] = fts[1, :, :3]
    eq = close_f32(fts[:n_obj], ref_fts, **tol)
    close_f32(locs[:n_obj], GOLD[f"{name}/locs"], **tol)
    close_f32(boxes[:n_obj], GOLD[f"{name}/boxes"], **tol)
    if cdt == "uint8":
        assert eq > 0.999, eq
    else:
        pc64 = [np.concatenate([p.astype(np.float64), (c / np.float32(127.5) - np.float32(1)).astype(np.float64)], 1)
                for p, c in objs]
        f64, l64, b64 = O.obj_processing_post(pc64, num_points, rot, idxs, True)
        assert close_f32(fts[:n_obj], f64) > 0.999
        close_f32(locs[:n_obj], l64)
        close_f32(boxes[:n_obj], b64)
    # padding slots: dataset_wrapper.py:62-70
    assert np.all(fts[n_obj:] == 1.0) and np.all(locs[n_obj:] == 0.0) and np.all(boxes[n_obj:] == 0.0)
    assert out["obj_masks"][0].tolist() == [True] * n_obj + [False] * 3


def test_kernel_matches_oracle_at_training_shapes():
    """4 scenes x 80 slots x 1024 points, per-scene rotations (one unrotated), host-drawn indices."""
    rng = np.random.default_rng(5)
    packed = G.PackedScans(DEV)
    scenes, n_objs = [], [80, 37, 61, 20]
    for s, n in enumerate(n_objs):
        objs = O.synth_scene(rng, n, np.uint8)
        scenes.append(objs)
        pack(objs, shuffle_seed=s, scan_id=f"s{s}", packed=packed)
    packed.finalize()
    rows = G.batch_rows(packed, [f"s{s}" for s in range(4)], 80)
    rots, sidx, want = [], torch.zeros((4, 80, 1024), dtype=torch.int32), []
    for s, objs in enumerate(scenes):
        np.random.seed(50 + s)
        rot, idxs = O.draw_like_reference([len(p) for p, _ in objs], 1024, "train" if s != 2 else "val", True)
        rots.append(rot)
        sidx[s, :len(objs)] = torch.from_numpy(np.stack(idxs, 0).astype(np.int32))
        f, l, _ = O.obj_processing_post(O.to_obj_pcds(objs), 1024, rot, idxs)
        want.append(O.pad_scene(f, l, 80))
    assert any(r is not None for r in rots) and rots[2] is None
    out = G.obj_processing_post(packed, rows, 1024, rot=rots, sample_idx=sidx)
    for s in range(4):
        f, l, m = want[s]
        n = n_objs[s]
        eq = close_f32(out["obj_fts"][s, :n].cpu().numpy(), f[:n].astype(np.float64))
        close_f32(out["obj_locs"][s, :n].cpu().numpy(), l[:n].astype(np.float64), rel=2.4e-7)
        assert eq > 0.999
        assert torch.equal(out["obj_masks"][s].cpu(), torch.from_numpy(m))
        assert np.all(out["obj_fts"][s, n:].cpu().numpy() == 1.0) and np.all(out["obj_locs"][s, n:].cpu().numpy() == 0)


def _index_coded_objects(ks):
    """Objects whose colours encode the point index (r = i & 255, g = i >> 8), so that the indices a
    device-side draw used can be read back from obj_fts[..., 3:5]."""
    objs = []
    for k in ks:
        i = np.arange(k)
        pts = np.stack([i * 0.01, np.sin(i * 0.37), np.cos(i * 0.11)], 1).astype(np.float32)
        col = np.stack([i & 255, i >> 8, np.zeros_like(i)], 1).astype(np.uint8)
        objs.append((pts, col))
    return objs


def _decode(fts):
    c = np.rint((fts[..., 3:5].astype(np.float64) + 1.0) * 127.5).astype(np.int64)
    return c[..., 0] + 256 * c[..., 1]


def test_device_sampler_properties():
    ks = [1, 5, 255, 256, 257, 1000, 4097, 30000]
    P = 256
    packed = pack(_index_coded_objects(ks))
    rows = G.batch_rows(packed, ["s0", "s0"], len(ks))          # the same scene twice in one batch
    a = G.obj_processing_post(packed, rows, P, seed=7)
    b = G.obj_processing_post(packed, rows, P, seed=7)
    c = G.obj_processing_post(packed, rows, P, seed=8)
    assert torch.equal(a["obj_fts"], b["obj_fts"])               # a function of (seed, row) only
    assert not torch.equal(a["obj_fts"], c["obj_fts"])
    idx = _decode(a["obj_fts"].cpu().numpy())
    for o, k in enumerate(ks):
        for r in range(2):
            got = idx[r, o]
            assert got.min() >= 0 and got.max() < k
            if k >= P:
                assert len(set(got.tolist())) == P, (k, "np.random.choice(replace=False): distinct")
            else:
                assert len(set(got.tolist())) <= k
        if k > 1:
            assert not np.array_equal(idx[0, o], idx[1, o])      # two draws of the same object differ
    # uniformity of the without-replacement draw: over many rows every point of a k=1000 object is
    # picked with probability P/k; pooled counts must sit within 5 sigma of a binomial
    rows = torch.full((64, len(ks)), 5, dtype=torch.int32)
    rows[:] = torch.tensor(list(packed.scan_objects("s0")))[5]
    idx = _decode(G.obj_processing_post(packed, rows, P, seed=11)["obj_fts"].cpu().numpy()).reshape(-1)
    n_draw = 64 * len(ks)
    cnt = np.bincount(idx, minlength=1000)
    p = P / 1000
    assert np.all(np.abs(cnt - n_draw * p) < 5 * np.sqrt(n_draw * p * (1 - p))), (cnt.min(), cnt.max())
    # with replacement (k=5 < P): each index ~ P/5 per row
    rows[:] = torch.tensor(list(packed.scan_objects("s0")))[1]
    cnt = np.bincount(_decode(G.obj_processing_post(packed, rows, P, seed=3)["obj_fts"].cpu().numpy()).reshape(-1),
                      minlength=5)
    assert np.all(np.abs(cnt - n_draw * P / 5) < 5 * np.sqrt(n_draw * P * 0.2 * 0.8)), cnt


def test_full_batch_properties_and_f32_colours():
    """B=64 x 80 slots x 1024 points (the bench batch), device-drawn samples, float32-stored colours:
    every real object is centred (|mean| <= 1e-6) with its farthest sample on the unit sphere, padding
    slots are exactly the wrapper's pad values, obj_locs size >= 0."""
    rng = np.random.default_rng(9)
    packed = G.PackedScans(DEV)
    n_objs = rng.integers(20, 80, size=8)
    for s, n in enumerate(n_objs):
        pack(O.synth_scene(rng, int(n), np.float32), shuffle_seed=s, scan_id=f"s{s}", packed=packed)
    packed.finalize()
    ids = [f"s{i % 8}" for i in range(64)]
    rows = G.batch_rows(packed, ids, 80)
    out = G.obj_processing_post(packed, rows, 1024, seed=123, need_boxes=True)
    fts, locs, masks = out["obj_fts"], out["obj_locs"], out["obj_masks"]
    assert fts.shape == (64, 80, 1024, 6) and masks.sum().item() == sum(int(n_objs[i % 8]) for i in range(64))
    real = fts[masks]
    assert real[..., :3].mean(1).abs().max().item() <= 1e-6
    far = real[..., :3].double().norm(dim=-1).max(1).values
    assert (far - 1).abs().max().item() <= 1e-6
    assert real[..., 3:].min().item() >= -1.0 and real[..., 3:].max().item() <= 1.0
    assert torch.all(fts[~masks] == 1.0) and torch.all(locs[~masks] == 0.0)
    assert torch.all(locs[masks][:, 3:] >= 0) and torch.allclose(locs[masks][:, 3:], out["obj_boxes"][masks][:, 3:])


def test_argument_errors_and_empty_inputs():
    packed = pack(_index_coded_objects([10, 2000]))
    rows = G.batch_rows(packed, ["s0"], 4)
    with pytest.raises(RuntimeError):
        G.obj_processing_post(packed, rows, 4096)                 # > 2048 points per object: unsupported
    with pytest.raises(ValueError):
        G.batch_rows(packed, ["s0"], 1)
    out = G.obj_processing_post(packed, rows[:0], 64)
    assert out["obj_fts"].shape == (0, 4, 64, 6)
    with pytest.raises(RuntimeError):
        G.obj_processing_post(G.PackedScans("cpu").finalize(), rows, 64)
