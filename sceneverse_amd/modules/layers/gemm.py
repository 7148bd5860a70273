"""Host side of libgps_hip.so's bf16 MFMA GEMMs (include/gps_hip.h gps_gemm_bf16): the nn.Linear
contractions of the reference's transformer layers

    w_qs / w_ks / w_vs / lang_cond_fc / fc      modules/layers/transformers.py:173-186, 193-197
    nn.MultiheadAttention in_proj / out_proj    modules/layers/transformers.py:120-121, 141
    linear1 -> activation -> dropout -> linear2 modules/layers/transformers.py:123-125, 148-152, 301-316

as explicit autograd functions (no global F.linear patching):

    linear(x, weight, bias)                       y = x W^T + b
    packed_linear(x, [(W_i, b_i), ...])           y = x [W_0; W_1; ...]^T + [b_0; b_1; ...]   (one GEMM)
    ffn(x, W1, b1, W2, b2, act, p_drop, training) y = dropout(act(x W1^T + b1)) W2^T + b2

Parameters stay fp32 masters (checkpoint- and optimizer-compatible with the reference); the GEMMs read a
persistent bf16 SHADOW of every weight that is refreshed only when the master changed (`_version`) -- or
written by the optimizer kernel itself (optim/fused_adamw.py) -- instead of autocast's per-call casts.
Forward / input-gradient / weight-gradient all run on the hand-written kernels; bias, GELU / ReLU, FFN
dropout and their derivatives live in the GEMM epilogues, weight and bias gradients come back in fp32.
GPU only: callers keep the torch formulation for CPU tensors (tests, gloo runs).
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Optional, Sequence

import torch

from ... import _native
from ..._native import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_FACTOR, EPI_BIAS_RELU, EPI_DGELU, EPI_DRELU, EPI_F32, EPI_MUL_AUX,
                        EPI_RELU_MAX16, EPI_RELU_SPLIT,
                        GEMM_NN, GEMM_NT,
                        GEMM_TN, GemmArgs)

_ENABLED = True
_VARIANT = {GEMM_NT: -1, GEMM_NN: -1, GEMM_TN: -1}       # -1 = the library's shape heuristic


def set_gemm_backend(enabled: bool) -> None:
    """False: layers fall back to F.linear (hipBLASLt) -- for A/B runs only."""
    global _ENABLED
    _ENABLED = bool(enabled)


def set_gemm_variant(form: int, variant: int) -> None:
    _VARIANT[form] = int(variant)


def enabled() -> bool:
    return _ENABLED


def usable(x: torch.Tensor, in_features: int, out_features: int) -> bool:
    """The kernels serve GPU tensors under bf16 execution with 8-multiple feature counts."""
    if not (_ENABLED and x.is_cuda and in_features % 8 == 0 and out_features % 8 == 0):
        return False
    if x.dtype == torch.bfloat16:
        return True
    return x.dtype == torch.float32 and torch.is_autocast_enabled("cuda") and \
        torch.get_autocast_dtype("cuda") == torch.bfloat16


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_FORM_NAME = {GEMM_NT: "nt", GEMM_NN: "nn", GEMM_TN: "tn"}


class _Product:
    """One NT / NN / TN product ready to launch: the C argument record + what bench.py's per-launch accounting needs.
    `keep` pins the tensors whose addresses the record holds."""
    __slots__ = ("args", "name", "nbytes", "flops", "work_fraction", "device", "keep", "form", "epilogue")


def _product(form: int, epilogue: int, M: int, N: int, K: int, A: torch.Tensor, lda: int, B: torch.Tensor, ldb: int,
             C: torch.Tensor, ldc: int, bias: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None,
             ldaux: int = 0, aux_out: Optional[torch.Tensor] = None, ldaux_out: int = 0,
             workspace: Optional[torch.Tensor] = None, colsum: Optional[torch.Tensor] = None, p_drop: float = 0.0,
             seed_dev: Optional[torch.Tensor] = None, splits: int = 1, variant: Optional[int] = None,
             extent_dev: Optional[torch.Tensor] = None) -> _Product:
    a = GemmArgs()
    a.form, a.epilogue, a.M, a.N, a.K = form, epilogue, M, N, K
    a.splits = splits
    a.variant = _VARIANT[form] if variant is None else variant
    a.reserved = 0
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc
    a.bias = _ptr(bias)
    a.aux, a.ldaux, a.aux_out, a.ldaux_out = _ptr(aux), ldaux, _ptr(aux_out), ldaux_out
    a.workspace, a.colsum, a.seed_dev = _ptr(workspace), _ptr(colsum), _ptr(seed_dev)
    a.extent_dev = _ptr(extent_dev)
    a.seed, a.p_drop = 0, float(p_drop)
    from ...pointnet2._ext import profiling
    nbytes = 2 * (M * K + N * K) + {EPI_F32: 4 * M * N, EPI_RELU_SPLIT: 6 * M * N, EPI_RELU_MAX16: M * N // 4}.get(epilogue, 2 * M * N)
    work_fraction = None
    if extent_dev is not None and profiling():
        # the kernel stops at *extent_dev rows of the token dimension (M for NT / NN, K for TN): charge that share
        # of the static shape's work.  Snapshot now (the word may be rewritten by the next step), read at profile_stop.
        snap = extent_dev.detach().clone()
        full = K if form == GEMM_TN else M
        work_fraction = lambda: min(1.0, max(0.0, float(snap.item()) / full))  # noqa: E731
        out_b = {EPI_F32: 4}.get(epilogue, 2)
        if form == GEMM_TN:      # both operands lose token rows, the (M, N) result is written whole
            nbytes = lambda f: int(2 * f * K * (M + N)) + out_b * M * N  # noqa: E731
        else:                    # A and C lose rows, the weight operand is read whole
            nbytes = lambda f: int(f * M * (2 * K + out_b * N)) + 2 * N * K  # noqa: E731
    q = _Product()
    q.args, q.form, q.epilogue, q.device = a, form, epilogue, A.device
    q.name = f"gemm_{_FORM_NAME[form]}(M={M},N={N},K={K},epi={epilogue})"
    q.nbytes, q.flops, q.work_fraction = nbytes, 2 * M * N * K, work_fraction
    q.keep = (A, B, C, bias, aux, aux_out, workspace, colsum, seed_dev, extent_dev)
    return q


def _launch(q: _Product) -> None:
    from ...pointnet2._ext import _timed
    with torch.cuda.device(q.device), _timed(q.name, q.nbytes, q.flops, "bf16", q.work_fraction):
        st = _native.load().gps_gemm_bf16(ctypes.byref(q.args), _stream())
    _native.check(st, q.name)


_GROUPED = True          # False: the products of a twin call leave one by one (A/B, tests)


def set_grouped_launches(flag: bool) -> None:
    global _GROUPED
    _GROUPED = bool(flag)


def _launch_together(products: Sequence[_Product]) -> None:
    """Independent products: those that share form and epilogue leave as ONE launch over the union of their output tiles
    (gps_gemm_bf16_grouped -- e.g. the same Linear of the text stack and of the object stack, which alone fill 23 - 59 %
    of the chip); whatever the library declines (GPS_ERR_UNSUPPORTED) leaves one by one."""
    products = [q for q in products if q is not None]
    if len(products) < 2 or not _GROUPED:
        for q in products:
            _launch(q)
        return
    lib = _native.load()
    pending = list(products)
    while pending:
        head = pending[0]
        same = [q for q in pending if q.form == head.form and q.epilogue == head.epilogue and q.device == head.device
                and head.form != GEMM_TN][:4]
        pending = [q for q in pending if all(q is not t for t in same)] if len(same) > 1 else pending[1:]
        if len(same) < 2:
            _launch(head)
            continue
        arr = (GemmArgs * len(same))(*[q.args for q in same])
        from ...pointnet2._ext import _timed, profiling
        args = [q.args for q in same]
        name = (f"gemm_{_FORM_NAME[head.form]}_grouped(M={'+'.join(str(a.M) for a in args)},N={'+'.join(str(a.N) for a in args)},"
                f"K={'+'.join(str(a.K) for a in args)},epi={head.epilogue})")
        flops = sum(q.flops for q in same)
        work_fraction, nbytes = None, None
        if profiling():
            fr = [q.work_fraction for q in same]
            live = lambda: [f() if f is not None else 1.0 for f in fr]  # noqa: E731
            work_fraction = lambda: sum(q.flops * f for q, f in zip(same, live())) / max(1, flops)  # noqa: E731
            nbytes = lambda _f: sum((q.nbytes(f) if callable(q.nbytes) else q.nbytes) for q, f in zip(same, live()))  # noqa: E731
        else:
            nbytes = 0
        with torch.cuda.device(head.device), _timed(name, nbytes, flops, "bf16", work_fraction):
            st = lib.gps_gemm_bf16_grouped(arr, len(same), _stream())
        if st == _native.GPS_ERR_UNSUPPORTED:
            for q in same:
                _launch(q)
        else:
            _native.check(st, name)


def gemm(form: int, epilogue: int, M: int, N: int, K: int, A: torch.Tensor, lda: int, B: torch.Tensor, ldb: int,
         C: torch.Tensor, ldc: int, **kw) -> None:
    """Thin checked call of gps_gemm_bf16 on the current stream (shapes in the header's convention; keyword arguments
    as `_product`)."""
    _launch(_product(form, epilogue, M, N, K, A, lda, B, ldb, C, ldc, **kw))


# ---- the three contractions of a Linear -------------------------------------------------------------------
def _forward_product(x16: torch.Tensor, w16: torch.Tensor, bias: Optional[torch.Tensor], act: Optional[str] = None,
                     p_drop: float = 0.0, seed_dev: Optional[torch.Tensor] = None, want_pre=False,
                     rows_dev: Optional[torch.Tensor] = None):
    """-> (product, y, pre or None): the forward GEMM of `linear_forward`, not yet launched."""
    T, K = x16.shape
    N = w16.shape[0]
    y = torch.empty((T, N), dtype=torch.bfloat16, device=x16.device)
    pre = torch.empty_like(y) if (want_pre and act == "gelu") else None
    epi = {None: EPI_BIAS, "gelu": EPI_BIAS_GELU, "relu": EPI_BIAS_RELU}[act]
    if want_pre == "factor" and act == "gelu":
        epi = EPI_BIAS_GELU_FACTOR
    q = _product(GEMM_NT, epi, T, N, K, x16, x16.stride(0), w16, w16.stride(0), y, N, bias=bias, aux_out=pre, ldaux_out=N,
                 p_drop=p_drop if act else 0.0, seed_dev=seed_dev, extent_dev=rows_dev)
    return q, y, pre


def linear_forward(x16: torch.Tensor, w16: torch.Tensor, bias: Optional[torch.Tensor], act: Optional[str] = None,
                   p_drop: float = 0.0, seed_dev: Optional[torch.Tensor] = None, want_pre: bool = False,
                   rows_dev: Optional[torch.Tensor] = None):
    """x16 (T, K) bf16, w16 (N, K) bf16, bias (N) fp32 -> y (T, N) bf16 [, pre-activation (T, N) bf16].
    want_pre = "factor" (gelu): the second output is gelu'(pre) x dropout-mask / (1 - p) instead -- what the backward
    pass multiplies by (`linear_dgrad(act="factor")`), from the erf terms the activation computes anyway.
    rows_dev (all three contractions): int32 device word, the number of leading token rows that carry work; tiles of
    rows past it are skipped (their output rows stay unwritten), the weight gradient sums the live rows only."""
    q, y, pre = _forward_product(x16, w16, bias, act, p_drop, seed_dev, want_pre, rows_dev)
    _launch(q)
    return (y, pre) if want_pre else y


def _dgrad_product(dy16: torch.Tensor, w16: torch.Tensor, act: Optional[str] = None, aux: Optional[torch.Tensor] = None,
                   p_drop: float = 0.0, seed_dev: Optional[torch.Tensor] = None,
                   rows_dev: Optional[torch.Tensor] = None):
    """-> (product, dx): the input-gradient GEMM of `linear_dgrad`, not yet launched."""
    T, N = dy16.shape
    K = w16.shape[1]
    dx = torch.empty((T, K), dtype=torch.bfloat16, device=dy16.device)
    epi = {None: EPI_BIAS, "gelu": EPI_DGELU, "relu": EPI_DRELU, "factor": EPI_MUL_AUX}[act]
    q = _product(GEMM_NN, epi, T, K, N, dy16, dy16.stride(0), w16, w16.stride(0), dx, K, aux=aux,
                 ldaux=aux.stride(0) if aux is not None else 0, p_drop=p_drop if act in ("gelu", "relu") else 0.0,
                 seed_dev=seed_dev, extent_dev=rows_dev)
    return q, dx


def linear_dgrad(dy16: torch.Tensor, w16: torch.Tensor, act: Optional[str] = None, aux: Optional[torch.Tensor] = None,
                 p_drop: float = 0.0, seed_dev: Optional[torch.Tensor] = None,
                 rows_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy16 (T, N) bf16, w16 (N, K) bf16 -> dx (T, K) bf16 = dy W, optionally times the derivative of the
    activation that PRODUCED this layer's input (aux = its saved pre-activation (gelu) / output (relu)), or -- act
    "factor" -- times aux itself (the factor `linear_forward(want_pre="factor")` saved: derivative x dropout mask)."""
    q, dx = _dgrad_product(dy16, w16, act, aux, p_drop, seed_dev, rows_dev)
    _launch(q)
    return dx


_WS = {}


def _workspace(device, floats: int) -> torch.Tensor:
    """Split-K scratch, one growing buffer per device and stream (safe: launches on a stream are ordered)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < floats:
        buf = _WS[key] = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
    return buf


def linear_wgrad(dy16: torch.Tensor, x16: torch.Tensor, want_bias: bool = True, rows_dev: Optional[torch.Tensor] = None):
    """dy16 (T, N) bf16, x16 (T, K) bf16 -> dW (N, K) fp32 = dy^T x, db (N) fp32 = column sums of dy."""
    T, N = dy16.shape
    K = x16.shape[1]
    dw = torch.empty((N, K), dtype=torch.float32, device=dy16.device)
    db = torch.empty(N, dtype=torch.float32, device=dy16.device) if want_bias else None
    lib = _native.load()
    splits = int(lib.gps_gemm_pick_splits(GEMM_TN, N, K, T))
    ws = None
    if splits > 1:
        ws = _workspace(dy16.device, int(lib.gps_gemm_workspace_floats(GEMM_TN, N, K, splits)))
    gemm(GEMM_TN, EPI_F32, N, K, T, dy16, dy16.stride(0), x16, x16.stride(0), dw, K, workspace=ws, colsum=db,
         splits=splits, extent_dev=rows_dev)
    return dw, db


# ---- weight gradients on a side stream ---------------------------------------------------------------------------
# In the backward pass of a Linear, dX = dY W feeds the next backward op while dW = dY^T X (+ db) is only needed by the
# optimizer.  Both GEMM forms leave most of the matrix pipe idle (stage-copy and epilogue stalls), so the
# weight gradients are issued to a SECOND HIP stream and run beside the input-gradient chain; their results go
# straight into `param.grad` (no AccumulateGrad copy) and the streams are joined once, after backward and before the
# optimizer (`deferred_wgrads()` context, entered by sceneverse_amd/engine.py around `backward()`; a HIP-graph capture
# records the fork / join as graph edges).  Off by default: plain `loss.backward()` callers get ordinary autograd
# gradients.  Not used under torch DDP (its reducer hooks live on autograd's accumulation nodes).
_DEFER = {"on": False, "side": {}, "keep": [], "joined": True}


def _side_stream(device) -> "torch.cuda.Stream":
    st = _DEFER["side"].get(device)
    if st is None:
        st = _DEFER["side"][device] = torch.cuda.Stream(device=device)
    return st


class deferred_wgrads:
    """with deferred_wgrads(): loss.backward()   -- weight / bias gradients of the native Linears are computed on a
    side stream into `param.grad`; leaving the block makes the current stream wait for them."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _DEFER["on"]
        _DEFER["on"] = self.enabled
        return self

    def __exit__(self, *exc):
        _DEFER["on"] = self.prev
        join_wgrads()
        return False


def join_wgrads() -> None:
    for dev, side in _DEFER["side"].items():
        torch.cuda.current_stream(dev).wait_stream(side)
    _DEFER["keep"].clear()


# ---- grouped weight gradients ---------------------------------------------------------------------------------------
# The weight / bias gradients of a backward pass are independent of its input-gradient chain.  One at a time each of
# them is a small output (768 x 768 ... 3072 x 768) with a reduction over 5 000 - 22 000 token rows: split over K to
# fill the chip, every split dumping fp32 partial tiles that a second launch sums.  `grouped_wgrads()` DEFERS them --
# the backward functions below hand (dY, X, parameters) to `_GROUP` instead of launching -- and issues all of them in
# one persistent launch when the block is left (gps_gemm_wgrad_grouped: every 256 x 256 tile of every gradient walks its
# whole reduction in one workgroup, no partial tiles, results written straight into `param.grad`, or added to it when it
# already exists: the flat gradient buffer of the split-graph data-parallel step).  Deterministic.  Not used under torch
# DDP, whose reducer hooks live on autograd's accumulation nodes (sceneverse_amd/engine.py decides).
_GROUP = {"on": False, "only": None, "items": [], "seen": set(), "written": set(), "stale": set(), "ln_items": [], "ln_scratch": {}}


def grouped_written_ids() -> set:
    """ids of the parameters whose gradients the grouped launches have written so far (the engine reads this after its
    eager warm-up steps to learn which gradient buffers never need zeroing: see `mark_stale_grads`)."""
    return set(_GROUP["written"])


def reset_grouped_bookkeeping() -> None:
    """Forget which parameters were written (a new engine / model starts from scratch: ids of dead tensors are recycled)."""
    _GROUP["written"].clear()
    _GROUP["stale"].clear()


def mark_stale_grads(params) -> None:
    """The `.grad` buffers of these parameters hold LAST step's values: the next grouped write to each of them is a plain
    store (no zero-fill before, no read-modify-write), a second write in the same pass accumulates as usual."""
    _GROUP["stale"] = {id(p) for p in params}


def stale_grads_left() -> set:
    """ids marked by `mark_stale_grads` that no write has consumed yet (their buffers still hold last step's values)."""
    return set(_GROUP["stale"])


class grouped_wgrads:
    """with grouped_wgrads(): loss.backward()   -- weight / bias gradients of the native Linears are collected and
    computed by ONE grouped launch into `param.grad` when the block is left."""

    def __init__(self, enabled: bool = True, only=None):
        """only: optional iterable of parameters -- gradients of other parameters are NOT written by the deferred
        launches (they go back to autograd, which keeps or drops them as its `inputs=` say).  A backward pass restricted
        with `inputs=` must pass the same set here: the deferred writes are side effects autograd does not filter."""
        self.enabled = bool(enabled)
        self.only = None if only is None else {id(p) for p in only}

    def __enter__(self):
        self.prev = (_GROUP["on"], _GROUP.get("only"))
        _GROUP["on"], _GROUP["only"] = self.enabled, self.only
        return self

    def __exit__(self, exc_type, *exc):
        _GROUP["on"], _GROUP["only"] = self.prev
        if exc_type is None:
            flush_grouped_wgrads()
        else:
            _GROUP["items"].clear()
            _GROUP["ln_items"].clear()
            _GROUP["seen"].clear()
        return False


def _grad_buffer(p: torch.nn.Parameter, force_existing: bool):
    """-> (fp32 gradient buffer of p, accumulate flag).  A parameter without a gradient gets a fresh buffer (plain
    store); an existing one is added to in place when the kernel can address it (contiguous fp32, 16-byte aligned)."""
    g = p.grad
    if g is None:
        if force_existing:
            g = p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
            return g, 1
        g = p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
        return g, 0
    if g.dtype != torch.float32 or not g.is_contiguous() or (g.data_ptr() & 15):
        return None, 0
    if id(p) in _GROUP["stale"] and not force_existing:      # last step's values: overwrite
        _GROUP["stale"].discard(id(p))
        return g, 0
    return g, 1


def defer_ln_param_grads(part: torch.Tensor, parts: int, d: int, gamma, beta, needs=(True, True)) -> bool:
    """fused_norm's backward hands the (2, parts, d) partial rows of a LayerNorm's dgamma / dbeta over: inside
    `grouped_wgrads()` they are reduced with all the others of the pass by ONE launch into gamma.grad / beta.grad.
    Returns False when that does not apply (the caller then reduces them itself and returns them to autograd)."""
    if not _GROUP["on"] or not all(needs):
        return False
    if not all(isinstance(t, torch.nn.Parameter) and t.is_leaf and t.requires_grad and t.dtype == torch.float32 and t.is_contiguous()
               for t in (gamma, beta)):
        return False
    ids = {id(gamma), id(beta)}
    if _GROUP.get("only") is not None and not ids <= _GROUP["only"]:
        return False
    if ids & _GROUP["seen"]:
        flush_grouped_wgrads()
    _GROUP["seen"] |= ids
    _GROUP["ln_items"].append((part, int(parts), int(d), gamma, beta))
    return True


def _flush_ln_param_grads() -> None:
    items, _GROUP["ln_items"] = _GROUP["ln_items"], []
    if not items:
        return
    from ..._native import LnReduceProblem
    lib = _native.load()
    by_d = {}
    for it in items:
        by_d.setdefault((it[2], it[0].device), []).append(it)
    for (d, dev), group in by_d.items():
        key = (dev, d)
        scratch = _GROUP["ln_scratch"].get(key)
        if scratch is None:
            nbytes = int(lib.gps_ln_reduce_grouped_scratch_bytes(d))
            scratch = _GROUP["ln_scratch"][key] = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=dev)
        probs = []
        for part, parts, _, gamma, beta in group:
            gg, acc_g = _grad_buffer(gamma, force_existing=beta.grad is not None and id(beta) not in _GROUP["stale"])
            gb, acc_b = _grad_buffer(beta, force_existing=bool(acc_g))
            if gg is None or gb is None or bool(acc_g) != bool(acc_b):      # buffers the kernel cannot address: torch sums
                sums = part.sum(1)
                for t, v in ((gamma, sums[0]), (beta, sums[1])):
                    if t.grad is None:
                        t.grad = v.clone()
                    elif id(t) in _GROUP["stale"]:
                        _GROUP["stale"].discard(id(t))
                        t.grad.copy_(v)
                    else:
                        t.grad.add_(v)
                continue
            q = LnReduceProblem()
            q.part, q.out_gamma, q.out_beta, q.parts, q.accumulate = part.data_ptr(), gg.data_ptr(), gb.data_ptr(), parts, int(acc_g)
            probs.append(q)
            _GROUP["written"].update((id(gamma), id(beta)))
        if probs:
            arr = (LnReduceProblem * len(probs))(*probs)
            with torch.cuda.device(dev):
                st = lib.gps_ln_reduce_partials_grouped(arr, len(probs), d, scratch.data_ptr(), _stream())
            _native.check(st, f"ln_reduce_partials_grouped({len(probs)} problems)")
    del items


def flush_grouped_wgrads() -> None:
    """Issue every deferred weight gradient (one gps_gemm_wgrad_grouped call) and forget the operands."""
    _flush_ln_param_grads()
    items, _GROUP["items"] = _GROUP["items"], []
    _GROUP["seen"].clear()
    if not items:
        return
    from ..._native import WgradProblem
    probs, prob_rows, keep, fallback = [], [], [], []
    for dy16, x16, weights, biases, rows, rows_dev in items:
        T, K_in = x16.shape
        r = 0
        for w, b, n in zip(weights, biases, rows):
            stale_pair = id(w) in _GROUP["stale"] and (b is None or id(b) in _GROUP["stale"])
            gw, acc = _grad_buffer(w, force_existing=b is not None and b.grad is not None and not stale_pair)
            gb = None
            if gw is not None and b is not None:
                gb, acc_b = _grad_buffer(b, force_existing=bool(acc))
                if gb is None or bool(acc_b) != bool(acc):
                    gw = None
            if gw is None or (n & 7) or (r & 7) or (K_in & 7):   # a block the kernel cannot address (buffer, or a 16-byte
                #                                               misaligned row offset / pitch): classic path
                fallback.append((dy16, x16, w, b, r, n, rows_dev))
            else:
                q = WgradProblem()
                q.M, q.N, q.K, q.accumulate = n, K_in, T, int(acc)
                q.A, q.lda = dy16.data_ptr() + 2 * r, dy16.stride(0)
                q.B, q.ldb = x16.data_ptr(), x16.stride(0)
                q.C, q.ldc = gw.data_ptr(), K_in
                q.colsum = gb.data_ptr() if gb is not None else None
                q.extent_dev = _ptr(rows_dev)
                probs.append(q)
                prob_rows.append((dy16, x16, rows_dev))
                _GROUP["written"].add(id(w))
                if b is not None:
                    _GROUP["written"].add(id(b))
            r += n
        keep.append((dy16, x16, rows_dev))
    dev = items[0][0].device
    if probs:
        arr = (WgradProblem * len(probs))(*probs)
        from ...pointnet2._ext import _timed, profiling
        flops = sum(2 * q.M * q.N * q.K for q in probs)
        nbytes = sum(2 * q.K * (q.M + q.N) + 4 * q.M * q.N for q in probs)
        work_fraction = None
        if profiling():
            # bench accounting: a problem with a device-side row extent does extent / K of its static work; snapshots of
            # the extent words now, read back at profile_stop (the words may be rewritten by the next step)
            shapes = [(q.M, q.N, q.K) for q in probs]
            exts = [(i, rd.detach().clone()) for i, (_, _, rd) in enumerate(prob_rows) if rd is not None]

            def live():
                k = [kk for _, _, kk in shapes]
                for i, snap in exts:
                    k[i] = min(k[i], max(0, int(snap.item())))
                return k

            work_fraction = lambda: sum(2 * m * n * kk for (m, n, _), kk in zip(shapes, live())) / max(1, flops)  # noqa: E731
            nbytes = lambda f: sum(2 * kk * (m + n) + 4 * m * n for (m, n, _), kk in zip(shapes, live()))       # noqa: E731
        with torch.cuda.device(dev), _timed(f"gemm_tn_grouped(problems={len(probs)})", nbytes, flops, "bf16", work_fraction):
            st = _native.load().gps_gemm_wgrad_grouped(arr, len(probs), _stream())
        _native.check(st, f"gemm_wgrad_grouped({len(probs)} problems)")
    for dy16, x16, w, b, r, n, rows_dev in fallback:
        dw, db = linear_wgrad(dy16[:, r:r + n], x16, want_bias=b is not None, rows_dev=rows_dev)
        for t, g in ((w, dw), (b, db)):
            if t is None:
                continue
            if t.grad is None:
                t.grad = g
            elif id(t) in _GROUP["stale"]:
                _GROUP["stale"].discard(id(t))
                t.grad.copy_(g)
            else:
                t.grad.add_(g)
    del keep


def _wgrad_to_params(dy16: torch.Tensor, x16: torch.Tensor, weights, biases, rows, rows_dev=None) -> bool:
    """Deferred forms of `linear_wgrad` for a (packed) Linear whose parameters are leaf tensors: dW / db land in
    (or are added to) `param.grad` -- collected for one grouped launch (`grouped_wgrads`) or computed on a side stream
    (`deferred_wgrads`).  Returns False when neither applies (the caller then returns ordinary gradients to autograd)."""
    if not (_DEFER["on"] or _GROUP["on"]):
        return False
    params = [w for w in weights] + [b for b in biases if b is not None]
    if not all(isinstance(t, torch.nn.Parameter) and t.is_leaf and t.requires_grad and t.dtype == torch.float32 for t in params):
        return False
    if _GROUP["on"]:
        ids = {id(t) for t in params}
        if _GROUP.get("only") is not None and not ids <= _GROUP["only"]:
            return False                      # outside this pass's `inputs=`: autograd decides what happens to them
        if ids & _GROUP["seen"]:          # a parameter used twice in one pass: its two gradients must not share a launch
            flush_grouped_wgrads()
        _GROUP["seen"] |= ids
        _GROUP["items"].append((dy16, x16, tuple(weights), tuple(biases), tuple(rows), rows_dev))
        return True
    dev = dy16.device
    cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
    want_bias = any(b is not None for b in biases)
    side.wait_stream(cur)                                   # dy16 / x16 (and a zeroed flat gradient buffer) are ready
    with torch.cuda.stream(side):
        single = len(weights) == 1 and weights[0].grad is None and (biases[0] is None or biases[0].grad is None)
        dw, db = linear_wgrad(dy16, x16, want_bias=want_bias, rows_dev=rows_dev)
        r = 0
        for w, b, n in zip(weights, biases, rows):
            gw = dw if single else dw[r:r + n]
            if w.grad is None:
                w.grad = gw if single else gw.clone()
            else:
                w.grad.add_(gw)
            if b is not None:
                gb = db if single else db[r:r + n]
                if b.grad is None:
                    b.grad = gb if single else gb.clone()
                else:
                    b.grad.add_(gb)
            r += n
    # operands and results stay referenced until the join: the caching allocator must not hand their memory to a
    # later allocation of the main stream while the side stream still reads / writes it
    _DEFER["keep"].append((dy16, x16, dw, db))
    return True


# ---- fp32-accurate MLP chains on the bf16 MFMA path (frozen point-encoder heads) ----------------------------------
def split3_rows(x: torch.Tensor, k_pad: int) -> torch.Tensor:
    """fp32 (M, K) -> bf16 (M, 3 k_pad) = [hi | lo | hi], hi = bf16(x), lo = bf16(x - hi), zero padding to k_pad."""
    M, K = x.shape
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    out = torch.zeros((M, 3 * k_pad), dtype=torch.bfloat16, device=x.device) if k_pad != K else \
        torch.empty((M, 3 * k_pad), dtype=torch.bfloat16, device=x.device)
    out[:, :K] = hi
    out[:, k_pad:k_pad + K] = lo
    out[:, 2 * k_pad:2 * k_pad + K] = hi
    return out


def split3_weight(w: torch.Tensor, k_pad: int) -> torch.Tensor:
    """fp32 (N, K) -> bf16 (N, 3 k_pad) = [W_hi | W_hi | W_lo]: against [x_hi | x_lo | x_hi] the K-sum is
    x_hi W_hi + x_lo W_hi + x_hi W_lo = x W^T up to the dropped lo x lo term (2^-16 relative)."""
    N, K = w.shape
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    out = torch.zeros((N, 3 * k_pad), dtype=torch.bfloat16, device=w.device)
    out[:, :K] = hi
    out[:, k_pad:k_pad + K] = hi
    out[:, 2 * k_pad:2 * k_pad + K] = lo
    return out


def split3_points(xyz: torch.Tensor, feats: torch.Tensor, k_pad: int) -> torch.Tensor:
    """xyz (B, n, 3) fp32, feats (B, C, n) fp32 -> bf16 (B * n, 3 k_pad) = [hi | lo | hi] of the rows
    [xyz | feats^T] (one launch instead of cat + casts + three strided copies)."""
    B, n, _ = xyz.shape
    C = feats.shape[1]
    xyz, feats = xyz.float().contiguous(), feats.float().contiguous()
    out = torch.empty((B * n, 3 * k_pad), dtype=torch.bfloat16, device=xyz.device)
    with torch.cuda.device(xyz.device):
        st = _native.load().gps_split3_points(B, n, C, xyz.data_ptr(), feats.data_ptr(), k_pad, out.data_ptr(), _stream())
    _native.check(st, "split3_points")
    return out


def split3_mlp_max16(x: torch.Tensor, layers, rows_dev=None) -> torch.Tensor:
    """relu(...relu(x W1^T + s1)... Wn^T + sn) followed by the max over every 16 consecutive rows, fp32-accurate,
    as n MFMA GEMMs: x fp32 (M, K) (or the bf16 operand split3_points made), M % 16 == 0; layers = [(split3_weight(W_i, k_pad_i), shift_i fp32), ...] with
    k_pad_1 = K rounded up to 8 and k_pad_i = N_(i-1) (multiples of 8).  -> fp32 (M / 16, N_n).
    rows_dev (device int32, a multiple of 16): only the first *rows_dev rows carry work, the rest is not written."""
    if x.dtype == torch.bfloat16:           # already the [hi | lo | hi] operand of the first layer
        a, M = x, x.shape[0]
    else:
        M = x.shape[0]
        a = split3_rows(x, layers[0][0].shape[1] // 3)
    for li, (w3, shift) in enumerate(layers):
        N, K3 = w3.shape
        assert a.shape[1] == K3
        if li + 1 < len(layers):
            c = torch.empty((M, 3 * N), dtype=torch.bfloat16, device=x.device)
            gemm(GEMM_NT, EPI_RELU_SPLIT, M, N, K3, a, K3, w3, K3, c, 3 * N, bias=shift, extent_dev=rows_dev)
            a = c
        else:
            out = torch.empty((M // 16, N), dtype=torch.float32, device=x.device)
            gemm(GEMM_NT, EPI_RELU_MAX16, M, N, K3, a, K3, w3, K3, out, N, bias=shift, extent_dev=rows_dev)
    return out


# ---- bf16 shadows of the fp32 master weights ---------------------------------------------------------------
class _Shadow:
    __slots__ = ("w16", "b32", "versions", "weight_ids", "bias_ids", "row_offsets", "row_counts", "owners", "b32_is_copy")

    def __init__(self):
        self.w16 = None
        self.b32 = None
        self.versions = None
        self.owners = None          # weak references to the masters: an id() can be recycled after a module dies
        self.weight_ids = ()
        self.bias_ids = ()
        self.row_offsets = ()
        self.row_counts = ()
        self.b32_is_copy = False


_SHADOWS = {}
_REGISTRY_VERSION = 0          # bumped whenever a shadow buffer is (re)allocated: the optimizer re-reads the map


def registry_version() -> int:
    return _REGISTRY_VERSION


def shadow_targets() -> dict:
    """id(parameter) -> (bf16 shadow view or None, fp32 mirror view or None): where the optimizer kernel writes
    the updated value of a parameter besides its fp32 master."""
    out = {}
    for sh in _SHADOWS.values():
        if sh.w16 is None or sh.owners is None:
            continue
        alive = {id(t) for t in (r() for r in sh.owners) if t is not None}      # ids of dead masters may be recycled
        for wid, bid, off, cnt in zip(sh.weight_ids, sh.bias_ids, sh.row_offsets, sh.row_counts):
            if wid in alive:
                out[wid] = (sh.w16[off:off + cnt], None)
            if bid is not None and bid in alive and sh.b32 is not None and sh.b32_is_copy:
                out[bid] = (None, sh.b32[off:off + cnt])
    return out


def _versions(params) -> tuple:
    return tuple(p._version for p in params if p is not None)


def shadow_of(weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], pad_rows: int = 1):
    """(w16, b32): bf16 copy of the row-wise concatenation of `weights` and fp32 concatenation of `biases`,
    rebuilt only when one of the masters changed since the last call.  A single weight returns its own bias
    tensor (no copy).  The optimizer kernel may write these buffers itself and call `mark_fresh`.
    pad_rows > 1: the row count is rounded up to that multiple (zero rows / zero bias entries behind the data),
    for outputs whose width must be a multiple of 8 (the 30 522-row vocabulary decoder)."""
    key = tuple(id(w) for w in weights) + ((("pad", pad_rows),) if pad_rows > 1 else ())
    sh = _SHADOWS.get(key)
    if sh is None:
        _prune_dead_shadows()
        sh = _SHADOWS[key] = _Shadow()
    ver = _versions(list(weights) + list(biases))
    dev = weights[0].device
    masters = [t for t in list(weights) + list(biases) if t is not None]
    if sh.owners is None or len(sh.owners) != len(masters) or any(r() is not t for r, t in zip(sh.owners, masters)):
        sh.owners = tuple(weakref.ref(t) for t in masters)      # first use, or the ids now name other tensors
        sh.versions = None
    rows = sum(w.shape[0] for w in weights)
    rows = (rows + pad_rows - 1) // pad_rows * pad_rows
    padded = rows != sum(w.shape[0] for w in weights)
    if sh.w16 is None or sh.w16.device != dev or tuple(sh.w16.shape) != (rows, weights[0].shape[1]):
        global _REGISTRY_VERSION
        _REGISTRY_VERSION += 1
        sh.w16 = (torch.zeros if padded else torch.empty)((rows, weights[0].shape[1]), dtype=torch.bfloat16, device=dev)
        sh.b32 = None
        sh.versions = None
        sh.weight_ids = tuple(id(w) for w in weights)
        sh.bias_ids = tuple(id(b) if b is not None else None for b in biases)
        offs, r = [], 0
        for w in weights:
            offs.append(r)
            r += w.shape[0]
        sh.row_offsets = tuple(offs)
        sh.row_counts = tuple(w.shape[0] for w in weights)
    if sh.versions != ver:
        with torch.no_grad():
            r = 0
            for w in weights:
                sh.w16[r:r + w.shape[0]].copy_(w)
                r += w.shape[0]
            if any(b is not None for b in biases):
                if len(biases) == 1 and not padded:
                    sh.b32 = biases[0].detach()
                    sh.b32_is_copy = False
                else:
                    if sh.b32 is None or sh.b32.data_ptr() in [b.data_ptr() for b in biases]:
                        sh.b32 = torch.zeros(sh.w16.shape[0], dtype=torch.float32, device=dev)
                        sh.b32_is_copy = True
                        _REGISTRY_VERSION += 1
                    r = 0
                    for w, b in zip(weights, biases):      # a bias-free member of a packed group contributes zeros
                        if b is not None:
                            sh.b32[r:r + w.shape[0]].copy_(b)
                        else:
                            sh.b32[r:r + w.shape[0]].zero_()
                        r += w.shape[0]
            else:
                sh.b32 = None
        sh.versions = ver
    return sh.w16, sh.b32


def _prune_dead_shadows() -> None:
    """Drop the records whose masters are gone (a model that was deleted): their bf16 copies would otherwise stay
    resident for the life of the process (two models live in A/B benches and in the tests)."""
    global _REGISTRY_VERSION
    dead = [k for k, sh in _SHADOWS.items() if sh.owners is not None and any(r() is None for r in sh.owners)]
    for k in dead:
        del _SHADOWS[k]
    if dead:
        _REGISTRY_VERSION += 1


def shadow_entry(weights: Sequence[torch.Tensor]):
    """The shadow record of a weight group (None before its first use): the optimizer writes through it."""
    return _SHADOWS.get(tuple(id(w) for w in weights))


def mark_fresh(weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]]) -> None:
    sh = shadow_entry(weights)
    if sh is not None:
        sh.versions = _versions(list(weights) + list(biases))


def invalidate_shadows() -> None:
    """Force a rebuild of every shadow at its next use (masters changed behind Python's back: graph replays)."""
    for sh in _SHADOWS.values():
        sh.versions = None


def clear_shadows() -> None:
    global _REGISTRY_VERSION
    _SHADOWS.clear()
    _REGISTRY_VERSION += 1


def _as_rows16(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != torch.bfloat16:
        x2 = x2.to(torch.bfloat16)
    if x2.stride(-1) != 1 or (x2.stride(0) % 8) or (x2.data_ptr() % 16):
        x2 = x2.contiguous()
    return x2


# forward results computed ahead of the autograd node that owns them (`_twin_*_forward_only`: the two stacks' forward
# products leave paired, their backward passes stay separate nodes): id(input tensor) -> what the node's forward would compute
_PRECOMPUTED = {}


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b over the concatenation of n weight blocks (n = 1: a plain Linear)."""

    @staticmethod
    def forward(ctx, x, n, *wb):
        rows_dev = None
        if len(wb) == 2 * n + 1:                             # optional trailing device-side row count
            rows_dev, wb = wb[-1], wb[:-1]
        weights, biases = wb[:n], wb[n:]
        ahead = _PRECOMPUTED.pop(id(x), None)
        if ahead is not None:
            x16, w16, y = ahead
        else:
            w16, b32 = shadow_of(weights, biases)
            x16 = _as_rows16(x)
            y = linear_forward(x16, w16, b32, rows_dev=rows_dev)
        ctx.rows_dev = rows_dev
        ctx.save_for_backward(x16, w16)
        ctx.meta = (x.shape, x.dtype, n, [w.shape[0] for w in weights], [b is not None for b in biases])
        ctx.params = (weights, biases)                       # the parameter OBJECTS (deferred weight gradients)
        return y.view(*x.shape[:-1], w16.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x16, w16 = ctx.saved_tensors
        x_shape, x_dtype, n, rows, has_b = ctx.meta
        dy16 = _as_rows16(dy)
        need_w = any(ctx.needs_input_grad[2:2 + n])
        need_b = any(ctx.needs_input_grad[2 + n:])
        dws, dbs = [None] * n, [None] * n
        deferred = False
        if need_w and all(ctx.needs_input_grad[2:2 + n]) and all((not hb) or g for hb, g in zip(has_b, ctx.needs_input_grad[2 + n:])):
            deferred = _wgrad_to_params(dy16, x16, ctx.params[0], ctx.params[1], rows, ctx.rows_dev)      # side stream, first
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dy16, w16, rows_dev=ctx.rows_dev).view(x_shape)
            if dx.dtype != x_dtype:
                dx = dx.to(x_dtype)
        if (need_w or need_b) and not deferred:
            dw, db = linear_wgrad(dy16, x16, want_bias=need_b, rows_dev=ctx.rows_dev)
            r = 0
            for i in range(n):
                if ctx.needs_input_grad[2 + i]:
                    dws[i] = dw[r:r + rows[i]]
                if has_b[i] and ctx.needs_input_grad[2 + n + i]:
                    dbs[i] = db[r:r + rows[i]]
                r += rows[i]
        return (dx, None, *dws, *dbs) + ((None,) if ctx.rows_dev is not None else ())


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
           rows_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    if rows_dev is not None:
        return _LinearFn.apply(x, 1, weight, bias, rows_dev)
    return _LinearFn.apply(x, 1, weight, bias)


def packed_linear(x: torch.Tensor, layers: Sequence[torch.nn.Linear], rows_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One GEMM for several Linears that read the same input; output columns in the order given."""
    extra = (rows_dev,) if rows_dev is not None else ()
    return _LinearFn.apply(x, len(layers), *[m.weight for m in layers], *[m.bias for m in layers], *extra)


class _LinearGeluFn(torch.autograd.Function):
    """y = gelu(x W^T + b) for the rows below a device-side count: one forward GEMM whose epilogue also saves gelu'(pre)
    (EPI_BIAS_GELU_FACTOR), backward = one elementwise multiply + the input- and weight-gradient GEMMs (the masked-LM
    head's transform on the labelled rows only: optim/loss/fused_lm_loss.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rows_dev):
        w16, b32 = shadow_of((weight,), (bias,))
        x16 = _as_rows16(x)
        y, fac = linear_forward(x16, w16, b32, act="gelu", want_pre="factor", rows_dev=rows_dev)
        ctx.rows_dev = rows_dev
        ctx.save_for_backward(x16, w16, fac)
        ctx.meta = (x.shape, x.dtype, bias is not None)
        ctx.params = (weight, bias)
        return y.view(*x.shape[:-1], w16.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x16, w16, fac = ctx.saved_tensors
        x_shape, x_dtype, has_b = ctx.meta
        weight, bias = ctx.params
        rd = ctx.rows_dev
        dpre = _as_rows16(dy) * fac               # rows past the count hold garbage on both sides: never read below
        dw = db = None
        if not (ctx.needs_input_grad[1] and (not has_b or ctx.needs_input_grad[2])
                and _wgrad_to_params(dpre, x16, (weight,), (bias,), (weight.shape[0],), rd)):
            dw, db = linear_wgrad(dpre, x16, want_bias=has_b, rows_dev=rd)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dpre, w16, rows_dev=rd).view(x_shape)
            if dx.dtype != x_dtype:
                dx = dx.to(x_dtype)
        return dx, dw, db, None


def linear_gelu(x: torch.Tensor, linear: torch.nn.Linear, rows_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _LinearGeluFn.apply(x, linear.weight, linear.bias, rows_dev)


_GELU_FACTOR = True          # False: save the pre-activation and recompute gelu' and the mask in the backward epilogue (A/B, tests)


def set_gelu_factor(flag: bool) -> None:
    global _GELU_FACTOR
    _GELU_FACTOR = bool(flag)


class _FFNFn(torch.autograd.Function):
    """y = dropout(act(x W1^T + b1)) W2^T + b2 : two forward GEMMs, four backward GEMMs, no elementwise launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, p_drop, seed_dev, rows_dev=None):
        ahead = _PRECOMPUTED.pop(id(x), None)
        ctx.gelu_factor = act == "gelu" and _GELU_FACTOR
        if ahead is not None:
            x16, w1_16, w2_16, h, pre, y = ahead
        else:
            w1_16, b1_32 = shadow_of((w1,), (b1,))
            w2_16, b2_32 = shadow_of((w2,), (b2,))
            x16 = _as_rows16(x)
            # gelu: the forward epilogue saves gelu'(pre) x dropout-mask / (1 - p) (one multiply per element in the backward
            # epilogue instead of the erf terms and the mask hash: gps_gemm.hip EPI_BIAS_GELU_FACTOR / EPI_MUL_AUX)
            h, pre = linear_forward(x16, w1_16, b1_32, act=act, p_drop=p_drop, seed_dev=seed_dev,
                                    want_pre="factor" if ctx.gelu_factor else True, rows_dev=rows_dev)
            y = linear_forward(h, w2_16, b2_32, rows_dev=rows_dev)
        ctx.rows_dev = rows_dev
        ctx.save_for_backward(x16, w1_16, w2_16, h, pre, seed_dev)
        ctx.meta = (x.shape, x.dtype, act, float(p_drop), b1 is not None, b2 is not None)
        ctx.params = (w1, b1, w2, b2)
        return y.view(*x.shape[:-1], w2_16.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x16, w1_16, w2_16, h, pre, seed_dev = ctx.saved_tensors
        x_shape, x_dtype, act, p_drop, has_b1, has_b2 = ctx.meta
        dy16 = _as_rows16(dy)
        w1, b1, w2, b2 = ctx.params
        all_w = all(ctx.needs_input_grad[1:5][i] for i in (0, 2)) and (not has_b1 or ctx.needs_input_grad[2]) and \
            (not has_b2 or ctx.needs_input_grad[4])
        dw1 = db1 = dw2 = db2 = None
        rd = ctx.rows_dev
        if not (all_w and _wgrad_to_params(dy16, h, (w2,), (b2,), (w2.shape[0],), rd)):
            dw2, db2 = linear_wgrad(dy16, h, want_bias=has_b2, rows_dev=rd)
        dpre = linear_dgrad(dy16, w2_16, act="factor" if ctx.gelu_factor else act, aux=pre if act == "gelu" else h, p_drop=p_drop,
                            seed_dev=seed_dev, rows_dev=rd)
        if not (all_w and _wgrad_to_params(dpre, x16, (w1,), (b1,), (w1.shape[0],), rd)):
            dw1, db1 = linear_wgrad(dpre, x16, want_bias=has_b1, rows_dev=rd)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dpre, w1_16, rows_dev=rd).view(x_shape)
            if dx.dtype != x_dtype:
                dx = dx.to(x_dtype)
        return dx, dw1, db1, dw2, db2, None, None, None, None


def ffn(x: torch.Tensor, linear1: torch.nn.Linear, linear2: torch.nn.Linear, act: str, p_drop: float,
        training: bool, rows_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act in {"gelu", "relu"}; dropout between activation and linear2 as the reference's layers apply it."""
    p = float(p_drop) if training else 0.0
    seed_dev = None
    if p > 0.0:
        from .fused_attention import _next_device_seed
        seed_dev = _next_device_seed(x.device)
    return _FFNFn.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, act, p, seed_dev, rows_dev)


# ---- two independent stacks in lock-step: paired products --------------------------------------------------------------------
# The text encoder and the object encoder of the GPS model do not see each other until the joint layers (reference
# model/openvocab.py:41-63) and their layers run the same op sequence: packed projection -> attention core -> output
# projection -> residual LayerNorm -> FFN -> residual LayerNorm.  Launched one stack after the other their 768-wide GEMMs
# are 150 (text, 12 608 live rows) and 60 (objects) tiles of 256 x 256 on 256 CUs.  Here the layer code of either stack is a
# GENERATOR that yields its GEMM calls (`LinearOp`, `FFNOp`) instead of making them; `drive(gen)` executes them one by one
# (the ordinary forward), `drive_pair(gen_a, gen_b)` advances both stacks in step and issues the two products of a pair as
# one launch (`_TwinLinearFn`, `_TwinFFNFn`: ONE autograd node per pair, so that the backward pass pairs the input-gradient
# GEMMs the same way; weight gradients already leave grouped -- `grouped_wgrads`).  tools/probes/gemm_probe group: the
# eight forward / input-gradient launches of one layer pair 849 -> 616 us.
class LinearOp:
    """y = x [W_0; W_1; ...]^T + [b_0; ...] (`packed_linear`; one Linear: `linear`)."""
    __slots__ = ("x", "weights", "biases", "rows_dev")

    def __init__(self, x, weights, biases, rows_dev=None):
        self.x, self.weights, self.biases, self.rows_dev = x, tuple(weights), tuple(biases), rows_dev

    @staticmethod
    def of(x, layers: Sequence[torch.nn.Linear], rows_dev=None) -> "LinearOp":
        return LinearOp(x, [m.weight for m in layers], [m.bias for m in layers], rows_dev)

    def run(self):
        extra = (self.rows_dev,) if self.rows_dev is not None else ()
        return _LinearFn.apply(self.x, len(self.weights), *self.weights, *self.biases, *extra)


class FFNOp:
    """y = dropout(act(x W1^T + b1)) W2^T + b2 (`ffn`)."""
    __slots__ = ("x", "linear1", "linear2", "act", "p_drop", "training", "rows_dev")

    def __init__(self, x, linear1, linear2, act, p_drop, training, rows_dev=None):
        self.x, self.linear1, self.linear2, self.act = x, linear1, linear2, act
        self.p_drop, self.training, self.rows_dev = float(p_drop), bool(training), rows_dev

    def run(self):
        return ffn(self.x, self.linear1, self.linear2, self.act, self.p_drop, self.training, rows_dev=self.rows_dev)


def _step(gen, value, first):
    """Advance a layer generator: -> (op, None) when it yields its next GEMM call, (None, result) when it returns."""
    try:
        return (next(gen) if first else gen.send(value)), None
    except StopIteration as stop:
        return None, (stop.value,)


def drive(gen):
    """Run a layer generator alone: every yielded op is executed on the spot."""
    op, done = _step(gen, None, True)
    while done is None:
        op, done = _step(gen, op.run(), False)
    return done[0]


_TWIN = True             # False: `drive_pair` runs its two generators one after the other (A/B, tests)


def set_twin_stacks(flag: bool) -> None:
    global _TWIN
    _TWIN = bool(flag)


def twin_stacks() -> bool:
    return _TWIN and _ENABLED and _GROUPED


def drive_pair(gen_a, gen_b):
    """Run two INDEPENDENT layer generators in lock-step; ops of the same kind that both have pending go out paired.
    -> (result_a, result_b).  Results equal `drive(gen_a), drive(gen_b)` (the products are the same, tile for tile)."""
    if not twin_stacks():
        return drive(gen_a), drive(gen_b)
    op_a, done_a = _step(gen_a, None, True)
    op_b, done_b = _step(gen_b, None, True)
    while done_a is None or done_b is None:
        if done_a is not None:
            op_b, done_b = _step(gen_b, op_b.run(), False)
        elif done_b is not None:
            op_a, done_a = _step(gen_a, op_a.run(), False)
        elif isinstance(op_a, LinearOp) and isinstance(op_b, LinearOp):
            ya, yb = _twin_linear(op_a, op_b)
            op_a, done_a = _step(gen_a, ya, False)
            op_b, done_b = _step(gen_b, yb, False)
        elif isinstance(op_a, FFNOp) and isinstance(op_b, FFNOp) and op_a.act == op_b.act:
            ya, yb = _twin_ffn(op_a, op_b)
            op_a, done_a = _step(gen_a, ya, False)
            op_b, done_b = _step(gen_b, yb, False)
        else:
            # different kinds: the FFN waits (it is the longer op); the sequences re-align at the next matching pair
            if isinstance(op_a, LinearOp):
                op_a, done_a = _step(gen_a, op_a.run(), False)
            else:
                op_b, done_b = _step(gen_b, op_b.run(), False)
    return done_a[0], done_b[0]


def _dx_out(dx, shape, dtype):
    dx = dx.view(shape)
    return dx if dx.dtype == dtype else dx.to(dtype)


class _TwinLinearFn(torch.autograd.Function):
    """(y_a, y_b) = (`_LinearFn` of side a, `_LinearFn` of side b) with the two forward products -- and, when both output
    gradients arrive together, the two input-gradient products -- issued as one launch each.  A backward call that brings
    only one side's gradient (the split-graph data-parallel step runs the text and the object encoder's backward as two
    graphs) computes that side alone."""

    @staticmethod
    def forward(ctx, xa, xb, na, nb, rows_a, rows_b, *wb):
        ctx.set_materialize_grads(False)
        sides, at = [], 0
        for x, n, rows_dev in ((xa, na, rows_a), (xb, nb, rows_b)):
            weights, biases = wb[at:at + n], wb[at + n:at + 2 * n]
            at += 2 * n
            w16, b32 = shadow_of(weights, biases)
            x16 = _as_rows16(x)
            q, y, _ = _forward_product(x16, w16, b32, rows_dev=rows_dev)
            sides.append((q, y, x16, w16, x.shape, x.dtype, weights, biases, rows_dev))
        _launch_together([sd[0] for sd in sides])
        ctx.save_for_backward(sides[0][2], sides[0][3], sides[1][2], sides[1][3])
        ctx.meta = [(sd[4], sd[5], [w.shape[0] for w in sd[6]], [b is not None for b in sd[7]]) for sd in sides]
        ctx.params = [(sd[6], sd[7]) for sd in sides]
        ctx.rows_dev = [sd[8] for sd in sides]
        ctx.n = (na, nb)
        return tuple(sd[1].view(*sd[4][:-1], sd[3].shape[0]) for sd in sides)

    @staticmethod
    def backward(ctx, dya, dyb):
        saved = ctx.saved_tensors
        na, nb = ctx.n
        need = ctx.needs_input_grad
        grads_x, grads_wb, products, finish = [None, None], [], [], []
        at = 6
        for side, (dy, n) in enumerate(((dya, na), (dyb, nb))):
            x16, w16 = saved[2 * side], saved[2 * side + 1]
            x_shape, x_dtype, rows, has_b = ctx.meta[side]
            need_ws, need_bs = need[at:at + n], need[at + n:at + 2 * n]
            at += 2 * n
            dws, dbs = [None] * n, [None] * n
            if dy is not None:
                dy16 = _as_rows16(dy)
                rd = ctx.rows_dev[side]
                deferred = False
                if any(need_ws) and all(need_ws) and all((not hb) or g for hb, g in zip(has_b, need_bs)):
                    deferred = _wgrad_to_params(dy16, x16, ctx.params[side][0], ctx.params[side][1], rows, rd)
                if need[side]:
                    q, dx = _dgrad_product(dy16, w16, rows_dev=rd)
                    products.append(q)
                    finish.append((side, dx, x_shape, x_dtype))
                if (any(need_ws) or any(need_bs)) and not deferred:
                    dw, db = linear_wgrad(dy16, x16, want_bias=any(need_bs), rows_dev=rd)
                    r = 0
                    for i in range(n):
                        if need_ws[i]:
                            dws[i] = dw[r:r + rows[i]]
                        if has_b[i] and need_bs[i]:
                            dbs[i] = db[r:r + rows[i]]
                        r += rows[i]
            grads_wb += dws + dbs
        _launch_together(products)
        for side, dx, x_shape, x_dtype in finish:
            grads_x[side] = _dx_out(dx, x_shape, x_dtype)
        return (grads_x[0], grads_x[1], None, None, None, None, *grads_wb)


_TWIN_BACKWARD = True    # False: paired forward products, SEPARATE autograd nodes (a backward pass that is cut between the
#                          two stacks -- the split-graph data-parallel step runs them as two graphs -- must not meet a node
#                          that belongs to both: it would run twice and drag the other stack's nodes along with undefined
#                          gradients)


def set_twin_backward(flag: bool) -> None:
    global _TWIN_BACKWARD
    _TWIN_BACKWARD = bool(flag)


def _twin_linear_forward_only(op_a: LinearOp, op_b: LinearOp):
    ahead = []
    with torch.no_grad():
        for op in (op_a, op_b):
            w16, b32 = shadow_of(op.weights, op.biases)
            x16 = _as_rows16(op.x)
            q, y, _ = _forward_product(x16, w16, b32, rows_dev=op.rows_dev)
            ahead.append((q, (x16, w16, y)))
        _launch_together([q for q, _ in ahead])
    outs = []
    for op, (_, res) in zip((op_a, op_b), ahead):
        _PRECOMPUTED[id(op.x)] = res
        try:
            outs.append(op.run())
        finally:
            _PRECOMPUTED.pop(id(op.x), None)
    return tuple(outs)


def _twin_ffn_forward_only(op_a: FFNOp, op_b: FFNOp):
    from .fused_attention import _next_device_seed
    factor = op_a.act == "gelu" and _GELU_FACTOR
    st = []
    with torch.no_grad():
        for op in (op_a, op_b):
            p = op.p_drop if op.training else 0.0
            seed = _next_device_seed(op.x.device) if p > 0.0 else None
            w1_16, b1_32 = shadow_of((op.linear1.weight,), (op.linear1.bias,))
            w2_16, b2_32 = shadow_of((op.linear2.weight,), (op.linear2.bias,))
            x16 = _as_rows16(op.x)
            q1, h, pre = _forward_product(x16, w1_16, b1_32, act=op.act, p_drop=p, seed_dev=seed,
                                          want_pre="factor" if factor else True, rows_dev=op.rows_dev)
            st.append(dict(op=op, p=p, seed=seed, x16=x16, w1_16=w1_16, w2_16=w2_16, b2_32=b2_32, q1=q1, h=h, pre=pre))
        _launch_together([d["q1"] for d in st])
        for d in st:
            d["q2"], d["y"], _ = _forward_product(d["h"], d["w2_16"], d["b2_32"], rows_dev=d["op"].rows_dev)
        _launch_together([d["q2"] for d in st])
    outs = []
    for d in st:
        op = d["op"]
        _PRECOMPUTED[id(op.x)] = (d["x16"], d["w1_16"], d["w2_16"], d["h"], d["pre"], d["y"])
        try:
            outs.append(_FFNFn.apply(op.x, op.linear1.weight, op.linear1.bias, op.linear2.weight, op.linear2.bias, op.act,
                                     d["p"], d["seed"], op.rows_dev))
        finally:
            _PRECOMPUTED.pop(id(op.x), None)
    return tuple(outs)


def _twin_linear(op_a: LinearOp, op_b: LinearOp):
    if not _TWIN_BACKWARD:
        return _twin_linear_forward_only(op_a, op_b)
    return _TwinLinearFn.apply(op_a.x, op_b.x, len(op_a.weights), len(op_b.weights), op_a.rows_dev, op_b.rows_dev,
                               *op_a.weights, *op_a.biases, *op_b.weights, *op_b.biases)


class _TwinFFNFn(torch.autograd.Function):
    """(y_a, y_b) = (`_FFNFn` of side a, `_FFNFn` of side b): both first GEMMs as one launch, both second GEMMs as one
    launch; backward likewise when both gradients arrive (else the side that did alone)."""

    @staticmethod
    def forward(ctx, xa, xb, act, pa, pb, seed_a, seed_b, rows_a, rows_b, w1a, b1a, w2a, b2a, w1b, b1b, w2b, b2b):
        ctx.set_materialize_grads(False)
        factor = act == "gelu" and _GELU_FACTOR
        st = []
        for x, p, seed, rd, w1, b1, w2, b2 in ((xa, pa, seed_a, rows_a, w1a, b1a, w2a, b2a), (xb, pb, seed_b, rows_b, w1b, b1b, w2b, b2b)):
            w1_16, b1_32 = shadow_of((w1,), (b1,))
            w2_16, b2_32 = shadow_of((w2,), (b2,))
            x16 = _as_rows16(x)
            q1, h, pre = _forward_product(x16, w1_16, b1_32, act=act, p_drop=p, seed_dev=seed,
                                          want_pre="factor" if factor else True, rows_dev=rd)
            st.append(dict(x=x, x16=x16, w1_16=w1_16, w2_16=w2_16, b2_32=b2_32, q1=q1, h=h, pre=pre, p=p, seed=seed, rd=rd,
                           params=(w1, b1, w2, b2)))
        _launch_together([d["q1"] for d in st])
        ys = []
        for d in st:
            d["q2"], y, _ = _forward_product(d["h"], d["w2_16"], d["b2_32"], rows_dev=d["rd"])
            ys.append(y)
        _launch_together([d["q2"] for d in st])
        ctx.save_for_backward(*[t for d in st for t in (d["x16"], d["w1_16"], d["w2_16"], d["h"], d["pre"], d["seed"])])
        ctx.meta = [(d["x"].shape, d["x"].dtype, float(d["p"]), d["params"][1] is not None, d["params"][3] is not None) for d in st]
        ctx.params = [d["params"] for d in st]
        ctx.rows_dev = [d["rd"] for d in st]
        ctx.act, ctx.gelu_factor = act, factor
        return tuple(y.view(*d["x"].shape[:-1], d["w2_16"].shape[0]) for y, d in zip(ys, st))

    @staticmethod
    def backward(ctx, dya, dyb):
        saved, need, act = ctx.saved_tensors, ctx.needs_input_grad, ctx.act
        live = []
        grads = {}
        for side, dy in enumerate((dya, dyb)):
            if dy is None:
                continue
            x16, w1_16, w2_16, h, pre, seed = saved[6 * side:6 * side + 6]
            x_shape, x_dtype, p, has_b1, has_b2 = ctx.meta[side]
            w1, b1, w2, b2 = ctx.params[side]
            nw = need[9 + 4 * side:13 + 4 * side]
            all_w = nw[0] and nw[2] and (not has_b1 or nw[1]) and (not has_b2 or nw[3])
            live.append(dict(side=side, dy16=_as_rows16(dy), x16=x16, w1_16=w1_16, w2_16=w2_16, h=h, pre=pre, seed=seed,
                             x_shape=x_shape, x_dtype=x_dtype, p=p, has_b1=has_b1, has_b2=has_b2, w1=w1, b1=b1, w2=w2, b2=b2,
                             all_w=all_w, rd=ctx.rows_dev[side], dw1=None, db1=None, dw2=None, db2=None))
        for d in live:
            if not (d["all_w"] and _wgrad_to_params(d["dy16"], d["h"], (d["w2"],), (d["b2"],), (d["w2"].shape[0],), d["rd"])):
                d["dw2"], d["db2"] = linear_wgrad(d["dy16"], d["h"], want_bias=d["has_b2"], rows_dev=d["rd"])
            d["q"], d["dpre"] = _dgrad_product(d["dy16"], d["w2_16"], act="factor" if ctx.gelu_factor else act,
                                               aux=d["pre"] if act == "gelu" else d["h"], p_drop=d["p"], seed_dev=d["seed"],
                                               rows_dev=d["rd"])
        _launch_together([d["q"] for d in live])
        products = []
        for d in live:
            if not (d["all_w"] and _wgrad_to_params(d["dpre"], d["x16"], (d["w1"],), (d["b1"],), (d["w1"].shape[0],), d["rd"])):
                d["dw1"], d["db1"] = linear_wgrad(d["dpre"], d["x16"], want_bias=d["has_b1"], rows_dev=d["rd"])
            d["dx"] = None
            if need[d["side"]]:
                q, d["dx"] = _dgrad_product(d["dpre"], d["w1_16"], rows_dev=d["rd"])
                products.append(q)
        _launch_together(products)
        out = [None] * 17
        for d in live:
            if d["dx"] is not None:
                out[d["side"]] = _dx_out(d["dx"], d["x_shape"], d["x_dtype"])
            out[9 + 4 * d["side"]:13 + 4 * d["side"]] = [d["dw1"], d["db1"], d["dw2"], d["db2"]]
        return tuple(out)


def _twin_ffn(op_a: FFNOp, op_b: FFNOp):
    from .fused_attention import _next_device_seed
    if not _TWIN_BACKWARD:
        return _twin_ffn_forward_only(op_a, op_b)
    ps, seeds = [], []
    for op in (op_a, op_b):
        p = op.p_drop if op.training else 0.0
        ps.append(p)
        seeds.append(_next_device_seed(op.x.device) if p > 0.0 else None)
    return _TwinFFNFn.apply(op_a.x, op_b.x, op_a.act, ps[0], ps[1], seeds[0], seeds[1], op_a.rows_dev, op_b.rows_dev,
                            op_a.linear1.weight, op_a.linear1.bias, op_a.linear2.weight, op_a.linear2.bias,
                            op_b.linear1.weight, op_b.linear1.bias, op_b.linear2.weight, op_b.linear2.bias)


def activation_name(fn) -> Optional[str]:
    """Maps the layer's activation callable (modules/utils.get_activation_fn) to an epilogue name."""
    import torch.nn.functional as F
    if fn is F.gelu:
        return "gelu"
    if fn is F.relu:
        return "relu"
    name = getattr(fn, "__name__", "")
    return name if name in ("gelu", "relu") else None
