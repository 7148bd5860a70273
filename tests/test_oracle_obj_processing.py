"""oracle/obj_processing.py against outputs of the reference's own ScanBase._obj_processing_post
(tests/golden/obj_processing_ref.npz, made by tests/golden/make_golden_objproc.py): bit-exact in the
reference's dtype, including the RNG draw order and the rotation branch."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

from oracle import obj_processing as O  # noqa: E402
from make_golden_objproc import CASES  # noqa: E402  (the case table only; nothing from /root/reference)

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj_processing_ref.npz"))


def case_inputs(case):
    name, scene_seed, np_seed, n_obj, num_points, cdt, split, ks = case
    objs = O.synth_scene(np.random.default_rng(scene_seed), n_obj, np.dtype(cdt).type, ks)
    objs[1] = (np.repeat(objs[1][0][:1], len(objs[1][0]), 0), objs[1][1])
    return objs


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_reference_outputs_bit_for_bit(case):
    name, scene_seed, np_seed, n_obj, num_points, cdt, split, ks = case
    pcds = O.to_obj_pcds(case_inputs(case))
    np.random.seed(np_seed)
    rot, idxs = O.draw_like_reference([len(p) for p in pcds], num_points, split, True)
    fts, locs, boxes = O.obj_processing_post(pcds, num_points, rot, idxs, is_need_bbox=True)
    for got, key in ((fts, "fts"), (locs, "locs"), (boxes, "boxes")):
        ref = GOLD[f"{name}/{key}"]
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref), (name, key, np.abs(got - ref).max())


def test_cases_cover_the_branches():
    rots = []
    for case in CASES:
        name, scene_seed, np_seed, n_obj, num_points, cdt, split, ks = case
        np.random.seed(np_seed)
        rot, idxs = O.draw_like_reference(ks, num_points, split, True)
        rots.append(rot is not None)
        for k, idx in zip(ks, idxs):
            assert idx.shape == (num_points,) and idx.min() >= 0 and idx.max() < k
            if k >= num_points:
                assert len(set(idx.tolist())) == num_points          # without replacement
        assert any(k < num_points for k in ks) and any(k >= num_points for k in ks)
    assert any(rots) and not all(rots)          # rotated and unrotated scenes both present
    # the all-identical object really takes the max_dist < 1e-6 branch: xyz = rounding residue of x - mean
    assert np.all(np.abs(GOLD["u8_train/fts"][1][:, :3]) < 1e-12)


def test_pad_scene_matches_the_wrapper_contract():
    rng = np.random.default_rng(0)
    fts, locs = rng.normal(size=(5, 16, 6)), rng.normal(size=(5, 6))
    f, l, m = O.pad_scene(fts, locs, 8)
    assert f.dtype == np.float32 and l.dtype == np.float32 and m.tolist() == [True] * 5 + [False] * 3
    assert np.all(f[5:] == 1.0) and np.all(l[5:] == 0.0) and np.array_equal(f[:5], fts.astype(np.float32))
