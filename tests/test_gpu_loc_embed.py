"""gps_loc_embed_forward / backward (sceneverse_amd/modules/layers/fused_loc.py) against the Sequential it replaces --
nn.Linear(6, 768) + nn.LayerNorm(768), reference modules/vision/pcd_openvocab_encoder.py:64-66 -- in fp64: output and
the four parameter gradients, deterministic between calls."""
import os
import sys

import pytest
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneverse_amd.modules.layers import fused_loc as FL  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _seq(k_in=6, seed=0):
    torch.manual_seed(seed)
    seq = nn.Sequential(nn.Linear(k_in, 768), nn.LayerNorm(768))
    with torch.no_grad():
        seq[0].weight.mul_(3.0)
        seq[1].weight.add_(torch.randn(768) * 0.2)
        seq[1].bias.add_(torch.randn(768) * 0.2)
    return seq.to(DEV)


@pytest.mark.parametrize("shape,k_in", [((64, 80), 6), ((5120,), 6), ((1,), 6), ((3, 7), 3), ((1030,), 8)])
def test_loc_embed_matches_linear_layernorm_in_fp64(shape, k_in):
    seq = _seq(k_in)
    g = torch.Generator().manual_seed(sum(shape) + k_in)
    x = (torch.randn(*shape, k_in, generator=g) * 2.0).to(DEV)
    x[..., 0, :] = 0                                            # a padded box: all zeros (constant row -> LN of the bias)
    assert FL.supported(seq, x)
    y = FL.loc_embed(seq, x)
    assert y.shape == shape + (768,) and y.dtype == torch.float32
    seq64 = nn.Sequential(nn.Linear(k_in, 768), nn.LayerNorm(768)).to(DEV).double()
    seq64.load_state_dict({k: v.double() for k, v in seq.state_dict().items()})
    ref = seq64(x.double())
    assert (y.double() - ref).abs().max().item() <= 2e-5
    wy = torch.randn(*shape, 768, generator=g).to(DEV)
    for p in seq.parameters():
        p.grad = None
    y.backward(wy)
    first = [p.grad.clone() for p in seq.parameters()]
    ref.backward(wy.double())
    for (name, p), q in zip(seq.named_parameters(), seq64.parameters()):
        scale = max(1.0, q.grad.abs().max().item())
        assert (p.grad.double() - q.grad).abs().max().item() <= 3e-5 * scale, name
    for p in seq.parameters():
        p.grad = None
    FL.loc_embed(seq, x).backward(wy)
    assert all(torch.equal(a, p.grad) for a, p in zip(first, seq.parameters()))     # deterministic


def test_loc_embed_falls_back_when_the_input_needs_a_gradient():
    seq = _seq()
    x = torch.randn(4, 6, device=DEV, requires_grad=True)
    assert not FL.supported(seq, x)
    FL.loc_embed(seq, x).sum().backward()
    assert x.grad is not None
