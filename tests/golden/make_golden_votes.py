"""Generate tests/golden/votes_reference_cpu.pt: outputs of the REFERENCE's own PointnetSAModuleVotes /
PointnetSAModuleMSGVotes / PointnetLFPModuleMSG (/root/reference/modules/third_party/pointnet2/pointnet2_modules.py:
164-353, 418-496), imported unmodified and run on CPU with the CPU oracle injected as `_ext` (recipe of
tests/golden/make_golden.py).  Weights are not stored: oracle/param_fill.fill_params derives them from names.

    python tests/golden/make_golden_votes.py
"""
import builtins
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
builtins.__POINTNET2_SETUP__ = True

import torch  # noqa: E402

from oracle.param_fill import fill_params  # noqa: E402
from oracle.pointnet2_oracle import OracleExt  # noqa: E402

SEED = 13


def inputs():
    g = torch.Generator().manual_seed(SEED)
    xyz = torch.rand(2, 96, 3, generator=g) * 2 - 1
    feats = torch.randn(2, 5, 96, generator=g)
    xyz2 = xyz[:, :24].contiguous() + 0.01
    feats2 = torch.randn(2, 7, 24, generator=g)
    return xyz, feats, xyz2, feats2


CASES = {
    "votes_max": dict(kind="votes", kw=dict(mlp=[5, 16, 32], npoint=24, radius=0.6, nsample=16, pooling="max")),
    "votes_avg_norm": dict(kind="votes", kw=dict(mlp=[5, 16, 32], npoint=24, radius=0.6, nsample=16, pooling="avg", normalize_xyz=True)),
    "votes_rbf": dict(kind="votes", kw=dict(mlp=[5, 16, 32], npoint=24, radius=0.6, nsample=16, pooling="rbf", sigma=0.4)),
    "msg_votes": dict(kind="msg", kw=dict(mlps=[[5, 8, 16], [5, 12, 24]], npoint=24, radii=[0.4, 0.8], nsamples=[8, 16])),
    "lfp_msg": dict(kind="lfp", kw=dict(mlps=[[5, 8, 16], [5, 12, 16]], radii=[0.4, 0.8], nsamples=[8, 16], post_mlp=[16 + 7, 20])),
}


def build(mod, name):
    import copy
    c = CASES[name]
    cls = {"votes": mod.PointnetSAModuleVotes, "msg": mod.PointnetSAModuleMSGVotes, "lfp": mod.PointnetLFPModuleMSG}[c["kind"]]
    m = cls(**copy.deepcopy(c["kw"])).eval()
    fill_params(m, SEED)
    return m


def run(mod, name, dev="cpu"):
    xyz, feats, xyz2, feats2 = (t.to(dev) for t in inputs())
    m = build(mod, name).to(dev)
    with torch.no_grad():
        if CASES[name]["kind"] == "lfp":
            return {"out": m(xyz2, xyz, feats2, feats)}
        new_xyz, new_feats, inds = m(xyz, feats)
        return {"new_xyz": new_xyz, "new_features": new_feats, "inds": inds}


def main():
    import modules.third_party.pointnet2.pointnet2_modules as ref_mod
    import pointnet2_utils
    pointnet2_utils._ext = OracleExt
    out = {"seed": SEED}
    for name in CASES:
        out[name] = {k: v.clone() for k, v in run(ref_mod, name).items()}
        print(name, {k: tuple(v.shape) for k, v in out[name].items()})
    torch.save(out, os.path.join(HERE, "votes_reference_cpu.pt"))


if __name__ == "__main__":
    main()
