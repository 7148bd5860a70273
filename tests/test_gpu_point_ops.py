"""Parity of the gfx950 point ops (through the C ABI of libgps_hip.so) with the CPU oracle:
bit-exact for indices and copies, bit-exact for the deterministic group gradient, tolerance for
the atomic scatter-adds.  Plus size-independent properties at the full GPS workload size."""
import pytest
import torch

from oracle.pointnet2_oracle import OracleExt
from point_cases import (BQ_SHAPES, FPS_SHAPES, GROUP_SHAPES, STRESS_BQ_SHAPES, STRESS_FPS_SHAPES,
                         STRESS_GROUP_SHAPES, generic_cloud, sa1_cloud)
from sceneverse_amd.pointnet2 import _ext as hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mismatch(a, b):
    bad = (a != b).nonzero()
    return f"{bad.shape[0]} mismatches, first at {bad[:5].tolist()}: {a[tuple(bad[0])].item()} vs {b[tuple(bad[0])].item()}"


def test_library_is_the_hip_one():
    from sceneverse_amd import _native
    assert _native.load().gps_abi_version() == 11


def test_fps_sa1_adversarial_and_synthetic():
    x = sa1_cloud()
    ref = OracleExt.furthest_point_sampling(x, 32)
    got = hip.furthest_point_sampling(x.to(DEV), 32).cpu()
    assert torch.equal(got, ref), _mismatch(got, ref)


@pytest.mark.parametrize("n,m", FPS_SHAPES + STRESS_FPS_SHAPES)
def test_fps_shapes(n, m):
    x = generic_cloud(5, n, seed=n * 7 + m)
    ref = OracleExt.furthest_point_sampling(x, m)
    got = hip.furthest_point_sampling(x.to(DEV), m).cpu()
    assert torch.equal(got, ref), _mismatch(got, ref)


@pytest.mark.parametrize("n,m", [(1024, 32), (32, 16), (2048, 130), (100, 64), (7, 7), (1500, 1)])
def test_fps_with_the_sampled_points_from_the_same_launch(n, m):
    """gps_furthest_point_sampling_xyz: the oracle's indices (bit-exact) and xyz[b, idx[b, j]] for every (b, j) -- what the
    reference gets from transpose -> gather_operation -> transpose (pointnet2_modules.py:47-54)."""
    x = generic_cloud(5, n, seed=n * 11 + m)
    ref = OracleExt.furthest_point_sampling(x, m)
    idx, cen = hip.furthest_point_sampling_xyz(x.to(DEV), m)
    assert torch.equal(idx.cpu(), ref), _mismatch(idx.cpu(), ref)
    want = torch.gather(x, 1, ref.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(cen.cpu(), want)
    assert hip.furthest_point_sampling_xyz(torch.zeros(1, 4096, 3, device=DEV), 8) is None     # streaming form: two calls


def test_ball_query_sa1_sa2_chain():
    x = sa1_cloud()
    fps = OracleExt.furthest_point_sampling(x, 32)
    new_xyz = OracleExt.gather_points(x.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    ref = OracleExt.ball_query(new_xyz, x, 0.2, 32)
    got = hip.ball_query(new_xyz.to(DEV), x.to(DEV), 0.2, 32).cpu()
    assert torch.equal(got, ref), _mismatch(got, ref)
    fps2 = OracleExt.furthest_point_sampling(new_xyz, 16)
    nx2 = OracleExt.gather_points(new_xyz.transpose(1, 2).contiguous(), fps2).transpose(1, 2).contiguous()
    ref2 = OracleExt.ball_query(nx2, new_xyz, 0.4, 32)
    got2 = hip.ball_query(nx2.to(DEV), new_xyz.to(DEV), 0.4, 32).cpu()
    assert torch.equal(got2, ref2), _mismatch(got2, ref2)
    assert torch.equal(hip.furthest_point_sampling(new_xyz.to(DEV), 16).cpu(), fps2)


@pytest.mark.parametrize("n,m,radius,nsample", BQ_SHAPES + STRESS_BQ_SHAPES)
def test_ball_query_shapes(n, m, radius, nsample):
    x = generic_cloud(4, n, seed=n + m)
    centres = x[:, torch.randperm(n, generator=torch.Generator().manual_seed(1))[:m]].contiguous()
    if centres.shape[1] < m:
        centres = generic_cloud(4, m, seed=99)
    ref = OracleExt.ball_query(centres, x, radius, nsample)
    got = hip.ball_query(centres.to(DEV), x.to(DEV), radius, nsample).cpu()
    assert torch.equal(got, ref), _mismatch(got, ref)


def test_ball_query_points_exactly_on_the_radius():
    # lattice with spacing 0.2 == radius: d2 == r2 must NOT count (strict <)
    import numpy as np
    g = np.stack(np.meshgrid(*[np.arange(-3, 4)] * 3, indexing="ij"), -1).reshape(-1, 3)
    x = torch.from_numpy((g * np.float32(0.2)).astype(np.float32))[None].contiguous()
    c = x[:, ::17].contiguous()
    ref = OracleExt.ball_query(c, x, 0.2, 16)
    got = hip.ball_query(c.to(DEV), x.to(DEV), 0.2, 16).cpu()
    assert torch.equal(got, ref), _mismatch(got, ref)


@pytest.mark.parametrize("c,n,npoint,nsample", GROUP_SHAPES + STRESS_GROUP_SHAPES)
def test_group_points_and_grad(c, n, npoint, nsample):
    g = torch.Generator().manual_seed(c * 1000 + n)
    b = 3
    pts = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, npoint, nsample), generator=g, dtype=torch.int32)
    idx[0] = idx[0, 0, 0]                      # degenerate: every slot the same target
    ref = OracleExt.group_points(pts, idx)
    got = hip.group_points(pts.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, ref)
    go = torch.randn(b, c, npoint, nsample, generator=g)
    ref_g = OracleExt.group_points_grad(go, idx, n)
    got_g = hip.group_points_grad(go.to(DEV), idx.to(DEV), n)
    again = hip.group_points_grad(go.to(DEV), idx.to(DEV), n)
    f64 = OracleExt.group_points_grad_f64(go, idx, n)
    torch.testing.assert_close(got_g.cpu().double(), f64, rtol=1e-5, atol=1e-4)
    if npoint * nsample * 4 * 2 + n * 8 < 60000:   # deterministic CSR path (else atomics)
        assert torch.equal(got_g.cpu(), ref_g), "deterministic group grad must equal the oracle's order"
        assert torch.equal(got_g, again), "group grad must be run-to-run reproducible"


def test_gather_points_and_grad():
    g = torch.Generator().manual_seed(3)
    for (b, c, n, m) in [(4, 3, 1024, 32), (2, 7, 33, 50), (1, 1, 1, 1)]:
        pts = torch.randn(b, c, n, generator=g)
        idx = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32)
        assert torch.equal(hip.gather_points(pts.to(DEV), idx.to(DEV)).cpu(), OracleExt.gather_points(pts, idx))
        go = torch.randn(b, c, m, generator=g)
        torch.testing.assert_close(hip.gather_points_grad(go.to(DEV), idx.to(DEV), n).cpu(),
                                   OracleExt.gather_points_grad(go, idx, n), rtol=1e-5, atol=1e-5)


def test_three_nn_and_interpolate():
    g = torch.Generator().manual_seed(4)
    for (b, n, m, c) in [(3, 100, 17, 5), (2, 1, 2, 3), (2, 300, 1500, 2), (1, 257, 3, 1)]:
        u = torch.randn(b, n, 3, generator=g)
        k = torch.randn(b, m, 3, generator=g)
        k[:, -1] = k[:, 0]                      # duplicate known point -> distance ties
        d_ref, i_ref = OracleExt.three_nn(u, k)
        d_got, i_got = hip.three_nn(u.to(DEV), k.to(DEV))
        assert torch.equal(i_got.cpu(), i_ref)
        assert torch.equal(d_got.cpu(), d_ref)  # incl. +inf when m < 3
        feats = torch.randn(b, c, m, generator=g)
        w = torch.rand(b, n, 3, generator=g)
        o_ref = OracleExt.three_interpolate(feats, i_ref, w)
        assert torch.equal(hip.three_interpolate(feats.to(DEV), i_ref.to(DEV), w.to(DEV)).cpu(), o_ref)
        go = torch.randn(b, c, n, generator=g)
        torch.testing.assert_close(
            hip.three_interpolate_grad(go.to(DEV), i_ref.to(DEV), w.to(DEV), m).cpu(),
            OracleExt.three_interpolate_grad(go, i_ref, w, m), rtol=1e-5, atol=1e-5)


def test_empty_and_error_behaviour():
    z = torch.zeros(0, 16, 3, device=DEV)
    assert hip.furthest_point_sampling(z, 4).shape == (0, 4)
    assert hip.ball_query(torch.zeros(2, 0, 3, device=DEV), torch.zeros(2, 5, 3, device=DEV), 0.1, 4).shape == (2, 0, 4)
    with pytest.raises(RuntimeError):
        hip.group_points(torch.zeros(1, 2, 3, device=DEV).double(), torch.zeros(1, 1, 1, dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError):
        hip.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 4, 3), 0.2, 4)   # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        hip.gather_points(torch.zeros(1, 3, 8, device=DEV).transpose(1, 2), torch.zeros(1, 2, dtype=torch.int32, device=DEV))


def test_full_size_properties():
    """B=64 scenes x 80 objects x 1024 points (BASELINE config 2): properties that do not need the
    CPU oracle at full size + a spot-check of 64 objects against it."""
    from sceneverse_amd.data.synthetic import synth_batch
    d = synth_batch(64, seed=5)
    xyz = d["obj_fts"][..., :3].reshape(-1, 1024, 3).contiguous().to(DEV)
    b = xyz.shape[0]
    fps = hip.furthest_point_sampling(xyz, 32)
    assert fps.shape == (b, 32) and int(fps.min()) >= 0 and int(fps.max()) < 1024
    assert bool((fps[:, 0] == 0).all())
    new_xyz = hip.gather_points(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    assert torch.equal(new_xyz, torch.gather(xyz, 1, fps.long()[..., None].expand(-1, -1, 3)))
    idx = hip.ball_query(new_xyz, xyz, 0.2, 32)
    # every returned index is inside the ball, hits are strictly ascending then padded with the first
    picked = torch.gather(xyz[:, None].expand(-1, 32, -1, -1), 2, idx.long()[..., None].expand(-1, -1, -1, 3))
    d2 = ((picked - new_xyz[:, :, None]) ** 2).sum(-1)
    assert bool((d2 < 0.2 * 0.2 + 1e-6).all())
    inc = idx[:, :, 1:] > idx[:, :, :-1]
    pad = idx[:, :, 1:] == idx[:, :, :1]
    assert bool((inc | pad).all())
    grouped = hip.group_points(xyz.transpose(1, 2).contiguous(), idx)
    expect = torch.gather(xyz.transpose(1, 2)[:, :, None].expand(-1, -1, 32, -1), 3,
                          idx.long()[:, None].expand(-1, 3, -1, -1))
    assert torch.equal(grouped, expect)
    sel = torch.arange(0, b, b // 64)[:64]
    assert torch.equal(fps[sel].cpu(), OracleExt.furthest_point_sampling(xyz[sel].cpu(), 32))
    assert torch.equal(idx[sel].cpu(), OracleExt.ball_query(new_xyz[sel].cpu(), xyz[sel].cpu(), 0.2, 32))
    # SA2-shaped feature grouping at full size: (b,128,32) -> (b,128,16,32)
    feats = torch.randn(b, 128, 32, device=DEV)
    idx2 = torch.randint(0, 32, (b, 16, 32), device=DEV, dtype=torch.int32)
    g2 = hip.group_points(feats, idx2)
    assert torch.equal(g2, torch.gather(feats[:, :, None].expand(-1, -1, 16, -1), 3,
                                        idx2.long()[:, None].expand(-1, 128, -1, -1)))
