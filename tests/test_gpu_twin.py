"""Grouped forward / input-gradient launches (gps_gemm_bf16_grouped) and the lock-step execution of the text and the
object stack that feeds them (modules/layers/gemm.py drive_pair, model/openvocab.py).

* C ABI: n products in one launch == the same products launched one by one with variant 12 (same kernel body, same tiles:
  bit-equal), for every epilogue the entry admits, with device-side row extents and a ragged K.
* host: two layer generators driven in lock-step == driven one after the other (outputs and every gradient);
* model: the whole GPS pre-train forward + backward with the stacks paired == unpaired (dropout off)."""
import ctypes

import pytest
import torch

from sceneverse_amd import _native
from sceneverse_amd._native import (EPI_BIAS, EPI_BIAS_GELU_FACTOR, EPI_BIAS_RELU, EPI_DRELU, EPI_MUL_AUX, GEMM_NN, GEMM_NT,
                                    GemmArgs)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _args(form, epi, M, N, K, seed, extent=None, variant=12):
    g = torch.Generator(device=DEV).manual_seed(seed)
    A = (torch.rand(M, K, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
    B = ((torch.rand((N, K) if form == GEMM_NT else (K, N), device=DEV, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    bias = torch.rand(N, device=DEV, generator=g) if epi in (EPI_BIAS, EPI_BIAS_GELU_FACTOR, EPI_BIAS_RELU) else None
    aux = (torch.rand(M, N, device=DEV, generator=g) - 0.3).to(torch.bfloat16) if epi in (EPI_MUL_AUX, EPI_DRELU) else None
    aux_out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16) if epi == EPI_BIAS_GELU_FACTOR else None
    ext = torch.tensor([extent], dtype=torch.int32, device=DEV) if extent is not None else None
    a = GemmArgs()
    a.form, a.epilogue, a.M, a.N, a.K, a.splits, a.variant = form, epi, M, N, K, 1, variant
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = A.data_ptr(), K, B.data_ptr(), (K if form == GEMM_NT else N), C.data_ptr(), N
    a.bias = bias.data_ptr() if bias is not None else None
    a.aux, a.ldaux = (aux.data_ptr(), N) if aux is not None else (None, 0)
    a.aux_out, a.ldaux_out = (aux_out.data_ptr(), N) if aux_out is not None else (None, 0)
    a.p_drop, a.seed = (0.1 if epi in (EPI_BIAS_GELU_FACTOR, EPI_BIAS_RELU, EPI_DRELU) else 0.0), 7 + seed
    a.extent_dev = ext.data_ptr() if ext is not None else None
    return a, dict(A=A, B=B, C=C, bias=bias, aux=aux, aux_out=aux_out, ext=ext, rows=extent if extent is not None else M)


CASES = [
    (GEMM_NT, EPI_BIAS, [(1000, 768, 768, None), (520, 2376, 768, None)]),
    (GEMM_NT, EPI_BIAS_GELU_FACTOR, [(1500, 1024, 768, 1111), (640, 2048, 768, None), (300, 512, 256, None)]),
    (GEMM_NT, EPI_BIAS_RELU, [(900, 512, 384, None), (260, 768, 768, 250)]),
    (GEMM_NN, EPI_BIAS, [(1000, 768, 2304, 777), (512, 768, 2376, None)]),           # ragged K (2376 % 64 = 8)
    (GEMM_NN, EPI_MUL_AUX, [(1300, 768, 1024, None), (400, 768, 512, None), (256, 256, 128, None), (70, 264, 64, None)]),
    (GEMM_NN, EPI_DRELU, [(800, 768, 512, 640), (300, 520, 768, None)]),
]


@pytest.mark.parametrize("form,epi,shapes", CASES, ids=[f"form{c[0]}-epi{c[1]}-n{len(c[2])}" for c in CASES])
def test_grouped_launch_equals_single_launches(form, epi, shapes):
    lib = _native.load()
    st = torch.cuda.current_stream().cuda_stream
    singles = [_args(form, epi, M, N, K, 10 + i, ext) for i, (M, N, K, ext) in enumerate(shapes)]
    for a, _ in singles:
        _native.check(lib.gps_gemm_bf16(ctypes.byref(a), st), "single")
    grouped = [_args(form, epi, M, N, K, 10 + i, ext) for i, (M, N, K, ext) in enumerate(shapes)]
    arr = (GemmArgs * len(grouped))(*[a for a, _ in grouped])
    _native.check(lib.gps_gemm_bf16_grouped(arr, len(grouped), st), "grouped")
    torch.cuda.synchronize()
    for (_, s), (_, g) in zip(singles, grouped):
        r = s["rows"]
        assert torch.equal(s["C"][:r].view(torch.int16), g["C"][:r].view(torch.int16))
        assert not torch.isnan(g["C"][:r].float()).any()
        if s["aux_out"] is not None:
            assert torch.equal(s["aux_out"][:r].view(torch.int16), g["aux_out"][:r].view(torch.int16))


def test_grouped_launch_argument_checks():
    lib = _native.load()
    st = torch.cuda.current_stream().cuda_stream
    a, _ka = _args(GEMM_NT, EPI_BIAS, 256, 256, 128, 1)
    b, _kb = _args(GEMM_NT, EPI_BIAS_RELU, 256, 256, 128, 2)
    arr = (GemmArgs * 2)(a, b)
    assert lib.gps_gemm_bf16_grouped(arr, 2, st) == _native.GPS_ERR_INVALID_ARGUMENT      # epilogues differ
    c, _kc = _args(GEMM_NT, EPI_BIAS_RELU, 256, 256, 200, 3)                             # ragged K, not EPI_BIAS
    arr = (GemmArgs * 2)(b, c)
    assert lib.gps_gemm_bf16_grouped(arr, 2, st) == _native.GPS_ERR_UNSUPPORTED
    assert lib.gps_gemm_bf16_grouped(arr, 0, st) == _native.GPS_OK
    five = (GemmArgs * 5)(a, a, a, a, a)
    assert lib.gps_gemm_bf16_grouped(five, 5, st) == _native.GPS_ERR_UNSUPPORTED
    torch.cuda.synchronize()


def _stack(d, ff, layers, seed):
    torch.manual_seed(seed)
    mods = torch.nn.ModuleList([torch.nn.ModuleDict({
        "qkv": torch.nn.Linear(d, 3 * d), "out": torch.nn.Linear(d, d), "l1": torch.nn.Linear(d, ff), "l2": torch.nn.Linear(ff, d)})
        for _ in range(layers)]).to(DEV)
    return mods


def _stack_gen(mods, x, act, rows_dev=None):
    from sceneverse_amd.modules.layers import gemm
    for m in mods:
        p = yield gemm.LinearOp.of(x, [m["qkv"]], rows_dev=rows_dev)
        h = torch.tanh(p[..., :x.shape[-1]].float()).to(torch.bfloat16)
        o = yield gemm.LinearOp.of(h, [m["out"]], rows_dev=rows_dev)
        f = yield gemm.FFNOp(o, m["l1"], m["l2"], act, 0.0, False, rows_dev=rows_dev)
        x = (o.float() + f.float()).to(torch.bfloat16)
    return x


@pytest.mark.parametrize("twin_backward", [True, False], ids=["paired-backward", "forward-only"])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_lock_step_stacks_equal_sequential_stacks(act, twin_backward):
    from sceneverse_amd.modules.layers import gemm
    gemm.set_twin_backward(twin_backward)
    sa, sb = _stack(256, 512, 2, 1), _stack(256, 768, 3, 2)      # different depths: the longer stack finishes alone
    xa0 = torch.randn(700, 256, device=DEV).to(torch.bfloat16)
    xb0 = torch.randn(3, 100, 256, device=DEV).to(torch.bfloat16)
    res = {}
    for mode in ("sequential", "paired"):
        for m in list(sa.parameters()) + list(sb.parameters()):
            m.grad = None
        xa, xb = xa0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
        if mode == "paired":
            ya, yb = gemm.drive_pair(_stack_gen(sa, xa, act), _stack_gen(sb, xb, act))
        else:
            ya, yb = gemm.drive(_stack_gen(sa, xa, act)), gemm.drive(_stack_gen(sb, xb, act))
        (ya.float().square().mean() + yb.float().square().mean()).backward()
        res[mode] = [ya.detach(), yb.detach(), xa.grad, xb.grad] + [p.grad.clone() for p in list(sa.parameters()) + list(sb.parameters())]
    gemm.set_twin_backward(True)
    for i, (s, p) in enumerate(zip(res["sequential"], res["paired"])):
        torch.testing.assert_close(p.float(), s.float(), rtol=2e-2, atol=2e-3 * float(s.float().abs().max()) + 1e-8,
                                   msg=lambda m, i=i: f"tensor {i}: {m}")


@pytest.mark.parametrize("twin_backward", [True, False], ids=["paired-node", "separate-nodes"])
def test_lock_step_backward_with_one_side_only(twin_backward):
    """The split-graph data-parallel step runs the two encoders' backward passes as two graphs (and pairs the forward
    launches only: separate nodes, no retained graph needed); a paired node, too, serves a backward call that carries one
    side's gradient only."""
    from sceneverse_amd.modules.layers import gemm
    gemm.set_twin_backward(twin_backward)
    sa, sb = _stack(256, 512, 1, 3), _stack(256, 512, 1, 4)
    xa = torch.randn(300, 256, device=DEV).to(torch.bfloat16).requires_grad_(True)
    xb = torch.randn(500, 256, device=DEV).to(torch.bfloat16).requires_grad_(True)
    ya, yb = gemm.drive_pair(_stack_gen(sa, xa, "gelu"), _stack_gen(sb, xb, "gelu"))
    gemm.set_twin_backward(True)
    torch.autograd.backward(ya.float().sum(), inputs=[xa] + list(sa.parameters()), retain_graph=twin_backward)
    ga = xa.grad.clone()
    assert xb.grad is None and all(p.grad is None for p in sb.parameters())
    torch.autograd.backward(yb.float().sum(), inputs=[xb] + list(sb.parameters()))
    xa2 = xa.detach().clone().requires_grad_(True)
    gemm.drive(_stack_gen(sa, xa2, "gelu")).float().sum().backward()
    torch.testing.assert_close(ga.float(), xa2.grad.float(), rtol=2e-2, atol=2e-3 * float(xa2.grad.float().abs().max()))
    assert xb.grad is not None


def test_gps_model_paired_stacks_equal_unpaired(golden_cpu):
    from oracle.param_fill import fill_params
    from sceneverse_amd.model.build import build_model
    from sceneverse_amd.modules.layers import gemm
    from sceneverse_amd.optim.loss import Loss
    from util import clone_batch, gps_cfg, lang_dir
    fx = golden_cpu
    model = build_model(gps_cfg(lang_dir(fx["seed"]), freeze=True))
    fill_params(model, fx["seed"])
    model = model.to(DEV).eval()                     # dropout off: the two modes draw their masks in different orders
    loss_mod = Loss(model.cfg).to(DEV)
    res = {}
    for paired in (False, True):
        gemm.set_twin_stacks(paired)
        try:
            for p in model.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(clone_batch(fx["batch"], DEV))
                total, _ = loss_mod(out)
            total.backward()
        finally:
            gemm.set_twin_stacks(True)
        res[paired] = (total.detach().float(), {k: out[k].detach().float() for k in ("og3d_logits", "intra_text_embed", "intra_obj_embeds")},
                       {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert abs(res[True][0].item() - res[False][0].item()) <= 2e-3 * abs(res[False][0].item())
    for k, v in res[False][1].items():
        torch.testing.assert_close(res[True][1][k], v, rtol=2e-2, atol=2e-2 * float(v.abs().max()), msg=lambda m, k=k: f"{k}: {m}")
    assert res[True][2].keys() == res[False][2].keys()
    for n, g in res[False][2].items():
        ref = float(g.norm())
        assert float((res[True][2][n] - g).norm()) <= 3e-2 * ref + 1e-7, n
