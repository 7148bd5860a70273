"""gps_adamw_step (sceneverse_amd/optim/fused_adamw.py) against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW
on identical parameters, gradients and hyper-parameters: same update rule, so agreement to fp32 rounding (1e-6
relative to the parameter scale after several steps); bf16 shadows / packed bias mirrors of the GEMM layer follow
the masters without a cast launch; state dicts are interchangeable."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneverse_amd.optim.fused_adamw import GpsAdamW  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(768, 768), (768,), (30, 7), (5,), (1,), (2304, 768), (8193,), (3, 3, 3)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]


@pytest.mark.parametrize("max_norm", [None, 5.0, 0.05])
def test_matches_torch_adamw_and_clip(max_norm):
    a, b = _params(0), _params(0)
    groups = lambda ps: [{"params": ps[:4], "weight_decay": 0.01, "lr": 5e-4}, {"params": ps[4:], "weight_decay": 0.0, "lr": 1e-3}]
    ref = torch.optim.AdamW(groups(a), betas=(0.9, 0.98), eps=1e-8)
    opt = GpsAdamW(groups(b), betas=(0.9, 0.98), eps=1e-8)
    for step in range(5):
        g = torch.Generator().manual_seed(100 + step)
        for i, (pa, pb) in enumerate(zip(a, b)):
            if i == 3 and step % 2 == 1:            # a parameter that sometimes gets no gradient
                pa.grad = pb.grad = None
                continue
            gr = (torch.randn(pa.shape, generator=g) * (10.0 if step == 2 else 1.0)).to(DEV)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if max_norm is not None:
            want_norm = torch.nn.utils.clip_grad_norm_(a, max_norm)
        ref.step()
        opt.step(max_grad_norm=max_norm)
        if max_norm is not None:
            assert abs(float(opt.last_grad_norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        for pa, pb in zip(a, b):
            scale = pa.abs().max().item() + 1e-6
            assert (pa - pb).abs().max().item() <= 2e-6 * scale
    sa, sb = ref.state[a[0]], opt.state[b[0]]
    assert float(sb["step"]) == 5.0 and float(opt.state[b[3]]["step"]) == float(ref.state[a[3]]["step"]) == 3.0 and (sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max().item() < 1e-5 * sa["exp_avg_sq"].abs().max().item()


def test_scheduler_drives_the_device_learning_rate():
    (p,) = [torch.nn.Parameter(torch.ones(64, device=DEV))]
    opt = GpsAdamW([{"params": [p], "lr": 1.0, "weight_decay": 0.0}], betas=(0.0, 0.0), eps=1e-12)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 0.1 if s >= 1 else 1.0)
    p.grad = torch.ones_like(p)
    opt.step()                       # betas = 0: the update is lr * sign(g)
    assert torch.allclose(p, torch.zeros_like(p), atol=1e-6)
    sched.step()
    assert abs(float(opt.param_groups[0]["lr"]) - 0.1) < 1e-7 and torch.is_tensor(opt.param_groups[0]["lr"])
    p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(p, torch.full_like(p, -0.1), atol=1e-6)


def test_shadows_follow_the_masters_without_a_cast():
    from sceneverse_amd.modules.layers import gemm as G
    G.clear_shadows()
    lins = [torch.nn.Linear(64, n).to(DEV) for n in (64, 32, 8)]
    single = torch.nn.Linear(64, 64).to(DEV)
    x = torch.randn(16, 64, device=DEV)
    y = G.packed_linear(x, lins).float().sum() + G.linear(x, single.weight, single.bias).float().sum()
    y.backward()
    params = [p for m in lins + [single] for p in m.parameters()]
    opt = GpsAdamW(params, lr=1e-2)
    sh = G.shadow_entry([m.weight for m in lins])
    v_before = sh.w16._version
    opt.step(max_grad_norm=1.0)
    w16, b32 = G.shadow_of([m.weight for m in lins], [m.bias for m in lins])
    assert w16._version == v_before                                   # no refresh copy ran ...
    assert torch.equal(w16, torch.cat([m.weight for m in lins]).to(torch.bfloat16))   # ... yet the shadow is current
    assert torch.equal(b32, torch.cat([m.bias for m in lins]))
    s16, _ = G.shadow_of([single.weight], [single.bias])
    assert torch.equal(s16, single.weight.to(torch.bfloat16))
    G.clear_shadows()


def test_state_dict_roundtrip_with_torch_adamw():
    a, b = _params(1), _params(1)
    opt = GpsAdamW(a, lr=1e-3, betas=(0.9, 0.98))
    for p in a:
        p.grad = torch.ones_like(p)
    opt.step()
    sd = opt.state_dict()
    ref = torch.optim.AdamW(b, lr=1e-3, betas=(0.9, 0.98), capturable=True)
    ref.load_state_dict(sd)
    opt2 = GpsAdamW(_params(1), lr=1e-3, betas=(0.9, 0.98))
    opt2.load_state_dict(sd)
    assert float(opt2.state[opt2.param_groups[0]["params"][0]]["step"]) == 1.0


def test_capturable_in_a_hip_graph():
    ps = _params(2)
    opt = GpsAdamW(ps, lr=1e-3)
    static_g = [torch.randn_like(p) for p in ps]
    for p, g in zip(ps, static_g):
        p.grad = g
    opt.step(max_grad_norm=1.0)                  # builds the tables outside the capture
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in ps]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step(max_grad_norm=1.0)
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    assert float(opt.state[ps[0]]["step"]) == 3.0          # 1 eager step + 2 replays (the capture itself does not execute)
    assert any((a - b).abs().max().item() > 0 for a, b in zip(before, ps))
