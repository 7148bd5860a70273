"""Split-K weight gradients (sceneverse_amd/common/wgrad_splitk.py): same forward, same dX and db bit for
bit, dW equal to the single-GEMM autocast gradient within bf16 rounding of that gradient (the split form
keeps fp32 partials, i.e. it is the MORE accurate of the two -- checked against an fp32 reference)."""
import os
import sys

import pytest
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneverse_amd.common import wgrad_splitk as W  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tokens,n_in,n_out,shape3d", [(8320, 768, 2048, True), (19200, 768, 768, True),
                                                       (5120, 2048, 768, False), (6000, 512, 256, False),
                                                       (3200, 768, 3072, True)])
def test_splitk_path_matches_autocast_linear(tokens, n_in, n_out, shape3d):
    torch.manual_seed(0)
    lin = nn.Linear(n_in, n_out).to(DEV)
    x = torch.randn(tokens, n_in, device=DEV)
    if shape3d:
        x = x.view(64, tokens // 64, n_in)
    g = torch.randn(*x.shape[:-1], n_out, device=DEV)
    splits = W.pick_splits(tokens, n_out, n_in)
    assert (splits > 1 and tokens % splits == 0) or tokens == 3200      # 3 200 tokens: bias path only

    def run(ctx):
        lin.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16), ctx:
            y = lin(xi)
        (y.float() * g).sum().backward()
        return y.detach(), xi.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()

    import contextlib
    y0, dx0, dw0, db0 = run(contextlib.nullcontext())
    y1, dx1, dw1, db1 = run(W.splitk_wgrad())
    assert torch.nn.functional.linear is W._ORIG_LINEAR                 # patch removed on exit
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    assert dw1.dtype == torch.float32 and dw1.shape == dw0.shape
    # fp32 reference from the same bf16-rounded operands
    x16 = x.reshape(-1, n_in).to(torch.bfloat16).float()
    g16 = g.reshape(-1, n_out).to(torch.bfloat16).float()
    ref = g16.t() @ x16
    scale = ref.abs().max().item()
    e_split = (dw1 - ref).abs().max().item() / scale
    e_single = (dw0 - ref).abs().max().item() / scale
    if splits > 1:
        assert e_split <= 1e-4, e_split              # fp32 partials: only accumulation-order noise
    assert e_split <= e_single + 1e-6, (e_split, e_single)
    assert e_single < 1e-2
    # bias gradient: fp32 column sums of the bf16 dY (gps_colsum_bf16) vs autocast's bf16-rounded reduction
    db_ref = g16.sum(0)
    bscale = db_ref.abs().max().item()
    assert db1.dtype == torch.float32 and (db1 - db_ref).abs().max().item() <= 1e-5 * bscale
    assert (db0 - db_ref).abs().max().item() <= 1e-2 * bscale


@pytest.mark.parametrize("rows,cols,ld", [(19200, 3072, 3072), (5120, 2376, 2376), (1000, 768, 2304), (70, 8, 8),
                                          (1, 256, 256), (4097, 264, 264), (100, 20, 20)])
def test_colsum_kernel(rows, cols, ld):
    """gps_colsum_bf16 vs an fp64 column sum of the same bf16 values: fp32 accumulation error only;
    deterministic (two calls bit-equal); a strided view (ld > cols) and a shape outside the kernel's
    8-column granularity (falls to torch's sum inside the wrapper) included."""
    torch.manual_seed(rows + cols)
    base = torch.randn(rows, ld, device=DEV).to(torch.bfloat16)
    x = base[:, :cols]
    a, b = W.colsum_bf16(x), W.colsum_bf16(x)
    ref = x.double().sum(0)
    assert a.shape == (cols,) and a.dtype == torch.float32 and torch.equal(a, b)
    assert (a.double() - ref).abs().max().item() <= 2e-6 * max(1.0, x.double().abs().sum(0).max().item())


def test_small_or_odd_calls_are_left_alone():
    lin = nn.Linear(768, 768).to(DEV)
    x = torch.randn(16, 50, 768, device=DEV, requires_grad=True)       # 800 tokens: below both thresholds
    with torch.autocast("cuda", dtype=torch.bfloat16), W.splitk_wgrad():
        y = lin(x)
        assert type(y.grad_fn).__name__ != "_AttachWGradBackward"
        y2 = torch.nn.functional.linear(torch.randn(8000, 768, device=DEV), lin.weight.detach())   # no grad wanted
        assert y2.grad_fn is None
    with W.splitk_wgrad():                                               # no autocast: untouched
        y = lin(torch.randn(8192, 768, device=DEV))
        assert type(y.grad_fn).__name__ != "_AttachWGradBackward"
    assert W.pick_splits(19200, 768, 6) == 1 and W.pick_splits(3200, 3072, 768) == 1


def test_engine_uses_it_and_losses_agree():
    """The bench-sized step with and without the split-K path: same loss trajectory (dropout off)."""
    from bench import gps_pretrain_cfg, _lang_dir
    from sceneverse_amd.data.synthetic import synth_batch
    from sceneverse_amd.engine import GPSTrainStep
    from sceneverse_amd.modules.layers.transformers import MultiheadSelfAttention
    batch = synth_batch(64, n_obj=80, seed=5, device=DEV)
    traj = {}
    for flag in (False, True):
        st = GPSTrainStep(gps_pretrain_cfg(_lang_dir()), device=DEV, ddp=False, graph=False, seed=11,
                          splitk_wgrad=flag)
        for m in st.model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, MultiheadSelfAttention):
                m.dropout = 0.0
            if hasattr(m, "attention_probs_dropout_prob"):
                m.attention_probs_dropout_prob = 0.0
            if hasattr(m, "dropout_prob"):
                m.dropout_prob = 0.0
        traj[flag] = [st.step(dict(batch))[0].item() for _ in range(4)]
        del st
        torch.cuda.empty_cache()
    for a, b in zip(traj[False], traj[True]):
        assert abs(a - b) <= 3e-3 * abs(a), traj
    assert traj[True][-1] < traj[True][0]
