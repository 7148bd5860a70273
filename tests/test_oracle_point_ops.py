"""CPU checks of the C oracle (oracle/pointnet2_oracle.c): against independent numpy
restatements, hand-derived known answers for the semantics SURVEY.md App. B singles out, and the
reference's only own test (pointnet2_test.py:18-30, a gradcheck of three_interpolate)."""
import numpy as np
import pytest
import torch

from oracle.pointnet2_oracle import OracleExt, ball_query_numpy, fps_closed_form, opt_n_threads
from point_cases import BQ_SHAPES, FPS_SHAPES, generic_cloud, sa1_cloud


def test_opt_n_threads_matches_reference_formula():
    # include/cuda_utils.h:15-19: clamp(2^floor(log(n)/log(2)), 1, 512) evaluated in double
    expect = {1: 1, 2: 2, 3: 2, 4: 4, 7: 4, 8: 8, 31: 16, 32: 32, 33: 32, 63: 32, 64: 64, 100: 64,
              511: 256, 512: 512, 1000: 512, 1024: 512, 2048: 512, 100000: 512}
    for n, t in expect.items():
        assert opt_n_threads(n) == t, n


@pytest.mark.parametrize("n,m", [s for s in FPS_SHAPES if s[0] <= 2048])
def test_fps_literal_tree_equals_closed_form_tie_rule(n, m):
    x = generic_cloud(3, n, seed=n * 7 + m)
    lit = OracleExt.furthest_point_sampling(x, m).numpy()
    assert np.array_equal(lit, fps_closed_form(x.numpy(), m))


def test_fps_known_answers():
    # all points identical and outside the skip ball: every distance ties at 0 -> key order wins:
    # first pick after 0 is the point with the smallest (bitrev(k mod bs), k div bs) = index 0 again
    x = torch.ones(1, 8, 3)
    assert OracleExt.furthest_point_sampling(x, 4).tolist() == [[0, 0, 0, 0]]
    # every point inside |p|^2 <= 1e-3: all skipped -> index 0 every round
    x = torch.full((1, 16, 3), 0.01)
    assert OracleExt.furthest_point_sampling(x, 5).tolist() == [[0] * 5]
    # two exactly tied farthest points at indices 1 and 2 with bs = 4: bitrev2(1)=2, bitrev2(2)=1
    # -> index 2 wins although index 1 comes first
    x = torch.tensor([[[1., 0., 0.], [1., 2., 0.], [1., -2., 0.], [1., 0.5, 0.]]])
    assert OracleExt.furthest_point_sampling(x, 2).tolist() == [[0, 2]]
    # the skip rule compares against the DOUBLE literal 1e-3: |p|^2 == 1e-3f (> 1e-3) is kept
    s = np.float32(1e-3)
    x = torch.tensor([[[1., 0., 0.], [float(np.sqrt(s)), 0., 0.], [0.02, 0., 0.]]])
    mag = np.float32(x[0, 1, 0].item()) ** 2
    out = OracleExt.furthest_point_sampling(x, 3).tolist()[0]
    assert out[0] == 0 and 2 not in out[1:]      # point 2 (|p|^2 = 4e-4) can never be selected
    assert (float(mag) > 1e-3) == (out[1] == 1)


@pytest.mark.parametrize("n,m,radius,nsample", BQ_SHAPES)
def test_ball_query_c_equals_numpy(n, m, radius, nsample):
    x = generic_cloud(2, n, seed=n + m)
    c = generic_cloud(2, m, seed=99)
    assert np.array_equal(OracleExt.ball_query(c, x, radius, nsample).numpy(),
                          ball_query_numpy(c.numpy(), x.numpy(), radius, nsample))


def test_ball_query_known_answers():
    x = torch.tensor([[[0., 0., 0.], [0.1, 0., 0.], [0.2, 0., 0.], [0.05, 0., 0.], [5., 5., 5.]]])
    c = torch.tensor([[[0., 0., 0.], [9., 9., 9.]]])
    out = OracleExt.ball_query(c, x, 0.2, 4).tolist()[0]
    # d2 < r2 strict: the point at distance exactly 0.2 is excluded (0.2f*0.2f == (0.2f)^2);
    # hits 0,1,3 in index order, padded with the first hit; no hit -> zeros
    assert out == [[0, 1, 3, 0], [0, 0, 0, 0]]
    out = OracleExt.ball_query(c, x, 0.2, 2).tolist()[0]
    assert out[0] == [0, 1]                      # stops after nsample hits


def test_group_gather_interpolate_against_torch_indexing():
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(2, 5, 40, generator=g)
    idx = torch.randint(0, 40, (2, 6, 7), generator=g, dtype=torch.int32)
    exp = torch.gather(pts[:, :, None].expand(-1, -1, 6, -1), 3, idx.long()[:, None].expand(-1, 5, -1, -1))
    assert torch.equal(OracleExt.group_points(pts, idx), exp)
    go = torch.randn(2, 5, 6, 7, generator=g)
    exp_g = torch.zeros(2, 5, 40).scatter_add_(2, idx.long().view(2, 1, 42).expand(-1, 5, -1), go.view(2, 5, 42))
    torch.testing.assert_close(OracleExt.group_points_grad(go, idx, 40), exp_g, rtol=1e-5, atol=1e-5)
    i1 = torch.randint(0, 40, (2, 9), generator=g, dtype=torch.int32)
    assert torch.equal(OracleExt.gather_points(pts, i1),
                       torch.gather(pts, 2, i1.long()[:, None].expand(-1, 5, -1)))
    u, k = torch.randn(2, 11, 3, generator=g), torch.randn(2, 23, 3, generator=g)
    d2, i3 = OracleExt.three_nn(u, k)
    full = ((u[:, :, None] - k[:, None]) ** 2).sum(-1)
    top = full.topk(3, dim=2, largest=False)
    assert torch.equal(i3.long(), top.indices)
    torch.testing.assert_close(d2, top.values, rtol=1e-5, atol=1e-6)
    d2_small, _ = OracleExt.three_nn(u, k[:, :2].contiguous())
    assert torch.isinf(d2_small[..., 2]).all()   # (float)1e40 -> +inf when m < 3


def test_reference_own_test_three_interpolate_gradcheck():
    """pointnet2_test.py:18-30 of the reference, restated for the oracle: finite-difference check
    of d(three_interpolate)/d(features) at B=1,c=2,m=4,n=2 with its fixed idx/weights, tol 1e-1."""
    feats = torch.randn(1, 2, 4, dtype=torch.float32)
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32)
    weight = torch.tensor([[[1., 1., 1.], [2., 2., 2.]]])
    go = torch.randn(1, 2, 2)
    analytic = OracleExt.three_interpolate_grad(go, idx, weight, 4)
    eps = 1e-2
    numeric = torch.zeros_like(feats)
    for i in range(feats.numel()):
        fp, fm = feats.clone().view(-1), feats.clone().view(-1)
        fp[i] += eps
        fm[i] -= eps
        d = OracleExt.three_interpolate(fp.view_as(feats), idx, weight) - \
            OracleExt.three_interpolate(fm.view_as(feats), idx, weight)
        numeric.view(-1)[i] = (d * go).sum() / (2 * eps)
    torch.testing.assert_close(analytic, numeric, atol=1e-1, rtol=1e-1)


def test_oracle_rejects_wrong_dtype_and_layout():
    with pytest.raises(RuntimeError):
        OracleExt.ball_query(torch.zeros(1, 2, 3).double(), torch.zeros(1, 4, 3), 0.2, 4)
    with pytest.raises(RuntimeError):
        OracleExt.group_points(torch.zeros(1, 3, 8).transpose(1, 2), torch.zeros(1, 2, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        OracleExt.gather_points(torch.zeros(1, 3, 8), torch.zeros(1, 2, dtype=torch.int64))


def test_sa1_chain_runs_on_adversarial_clouds():
    x = sa1_cloud()
    fps = OracleExt.furthest_point_sampling(x, 32)
    assert (fps[:, 0] == 0).all()
    assert fps[2].eq(0).all() and fps[4].eq(0).all()      # all-identical / all-origin objects
    assert np.array_equal(fps.numpy(), fps_closed_form(x.numpy(), 32))
