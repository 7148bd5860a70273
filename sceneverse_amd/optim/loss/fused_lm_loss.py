"""Masked-LM head + cross-entropy over the LABELLED rows only (fused linear-cross-entropy).

Reference: `BertLMPredictionHead.decoder` + bias (modules/heads/pretrain_head.py:22-30) produces (B, L, 30522) logits
for every token, and `lm_cls_loss` (optim/loss/loss.py:56-61) then averages the cross-entropy over the ~15 % of tokens
whose label is not `ignore_index`.  The other rows contribute nothing to the loss or to any gradient, so here they are
never computed -- with static shapes, because the step is replayed as one HIP graph and the number of labelled tokens
is only known on the device:

  * the token rows are permuted so that the labelled ones come first (stable: original order inside each class);
  * the three GEMMs of the head run on libgps_hip.so with a DEVICE-side row count (`gps_gemm_args.extent_dev`): forward
    tiles past it exit at once, the weight gradient stops its reduction there, the input gradient (reduction over the
    vocabulary, few rows) runs split-K into fp32;
  * the vocabulary is padded to a multiple of 8 columns in a bf16 shadow of the decoder weight (zero rows) -- the
    cross-entropy kernels (`gps_masked_ce_*`, row-sparse already) read the first V columns of each row.

Loss value and every gradient equal the reference formulation's (to bf16 rounding of the logits); full logits exist
only when somebody asks for them (`LazyLMLogits.materialize()`: evaluation, accuracy metrics).
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import _native
from ...modules.layers import gemm as G

_TICKETS = {}


def _ticket(device: torch.device) -> torch.Tensor:
    """The zeroed arrival counter of gps_masked_ce_forward_rows' mean (left at zero by every launch)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def _padded_shadow(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """bf16 copy of `weight` (V, D) with the rows padded to a multiple of 8 (zeros) + fp32 padded bias, from the
    shadow registry of modules/layers/gemm.py (the clip + AdamW kernel writes it with the masters)."""
    return G.shadow_of([weight], [bias], pad_rows=8)


def row_plan(labels: torch.Tensor, vocab: int, ignore_index: int):
    """-> (perm, labels in permuted order, n_valid): the token rows permuted so that the labelled ones come first
    (stable), the device-side count of labelled rows.  One launch (gps_lm_row_plan)."""
    labels = labels.reshape(-1).to(torch.int64).contiguous()
    n = labels.numel()
    perm = torch.empty(n, dtype=torch.int64, device=labels.device)
    lp = torch.empty(n, dtype=torch.int64, device=labels.device)
    n_valid = torch.empty(1, dtype=torch.int32, device=labels.device)
    with torch.cuda.device(labels.device):
        st = _native.load().gps_lm_row_plan(n, int(vocab), labels.data_ptr(), int(ignore_index), perm.data_ptr(),
                                            lp.data_ptr(), n_valid.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _native.check(st, "lm_row_plan")
    return perm, lp, n_valid


class _PermuteLiveRows(torch.autograd.Function):
    """x (n, D) -> x[perm] (labelled rows first); backward scatters the gradient rows back and ZEROES the rows past the
    device-side count (the extent-aware kernels downstream never write them: undefined memory)."""

    @staticmethod
    def forward(ctx, x, perm, n_valid):
        ctx.save_for_backward(perm, n_valid)
        return x.index_select(0, perm)

    @staticmethod
    def backward(ctx, dy):
        perm, n_valid = ctx.saved_tensors
        if _rows_move_ok(dy, perm):
            # one launch: dx[perm[r]] = r < *n_valid ? dy[r] : 0 (perm is a bijection: every row of dx is written)
            dy = dy.contiguous()
            dx = torch.empty_like(dy)
            with torch.cuda.device(dy.device):
                st = _native.load().gps_rows_move(dy.shape[0], dy.shape[0], dy.shape[0], dy.shape[1] * dy.element_size(),
                                                  dy.data_ptr(), None, dx.data_ptr(), perm.data_ptr(), n_valid.data_ptr(), 1,
                                                  torch.cuda.current_stream().cuda_stream)
            _native.check(st, "rows_move")
            return dx, None, None
        live = (torch.arange(dy.shape[0], device=dy.device) < n_valid)[:, None]
        dy = torch.where(live, dy, torch.zeros((), dtype=dy.dtype, device=dy.device))
        dx = torch.empty_like(dy)
        dx.index_copy_(0, perm, dy)                        # perm is a bijection: every row is written
        return dx, None, None


def _rows_move_ok(x: torch.Tensor, idx: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and (x.shape[1] * x.element_size()) % 16 == 0
            and idx.dtype == torch.int64 and idx.is_contiguous())


_ZERO_ROWS = {}


def _zero_rows(n: int, d: int, device: torch.device) -> torch.Tensor:
    """A persistent all-zero (n, d) fp32 tensor (read-only by contract: the zero residual of the head's LayerNorm) -- one
    allocation + fill per (device, shape) instead of a fill launch per step."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), n, d)
    z = _ZERO_ROWS.get(key)
    if z is None:
        with torch.no_grad():
            z = torch.zeros((n, d), dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():     # (a fill captured into one graph would not have run for the others)
            # never evicted: a captured step graph reads the buffer by address on every replay (one entry per batch shape)
            _ZERO_ROWS[key] = z
    return z


class _SparseLMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias, labels, ignore_index: int, plan=None):
        n, D = h.shape
        V = weight.shape[0]
        w16, b32 = _padded_shadow(weight, bias)
        Vp = w16.shape[0]
        lib = _native.load()
        dev = h.device
        if plan is None:
            perm, lp, n_valid = row_plan(labels, V, ignore_index)
            hp = h.detach().to(torch.bfloat16).index_select(0, perm).contiguous()
        else:                                               # h arrives permuted already (rows past n_valid undefined)
            perm, lp, n_valid = None, plan[1], plan[2]
            hp = h.detach().to(torch.bfloat16).contiguous()
        logits = torch.empty((n, Vp), dtype=torch.bfloat16, device=dev)
        G.gemm(_native.GEMM_NT, _native.EPI_BIAS, n, Vp, D, hp, D, w16, D, logits, Vp, bias=b32, extent_dev=n_valid)
        rows = torch.empty(n, dtype=torch.float32, device=dev)
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        mean = torch.empty(2, dtype=torch.float32, device=dev)           # [mean over the labelled rows, their number]
        with torch.cuda.device(dev):
            st = lib.gps_masked_ce_forward_rows(n, V, 1, logits.data_ptr(), Vp, lp.data_ptr(), int(ignore_index),
                                                n_valid.data_ptr(), rows.data_ptr(), lse.data_ptr(), mean.data_ptr(),
                                                _ticket(dev).data_ptr(), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "masked_ce_forward")
        ctx.save_for_backward(hp, w16, logits, lp, lse, perm, n_valid, mean)
        ctx.meta = (n, D, V, Vp, int(ignore_index), h.dtype, bias is not None)
        ctx.prepermuted = plan is not None
        return mean[0]

    @staticmethod
    def backward(ctx, g):
        hp, w16, logits, lp, lse, perm, n_valid, mean = ctx.saved_tensors
        n, D, V, Vp, ignore_index, h_dtype, has_bias = ctx.meta
        lib = _native.load()
        dev = hp.device
        stream = torch.cuda.current_stream().cuda_stream
        g32 = g.reshape(1).float().contiguous()                             # (no launch for the fp32 scalar autograd hands over)
        dlogits = torch.empty((n, Vp), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            # rows past n_valid stay unwritten: the two GEMMs below stop at the extent (their ragged tail reads zeros)
            st = lib.gps_masked_ce_backward_rows(n, V, 1, logits.data_ptr(), Vp, lp.data_ptr(), ignore_index,
                                                 n_valid.data_ptr(), lse.data_ptr(), None, g32.data_ptr(),
                                                 mean[1:].data_ptr(), dlogits.data_ptr(), Vp, stream)
        _native.check(st, "masked_ce_backward")
        dh = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX = dlogits . W: a reduction over the vocabulary for a few hundred rows -> split-K into fp32
            splits = 16
            ws = G._workspace(dev, int(lib.gps_gemm_workspace_floats(_native.GEMM_NN, n, D, splits)))
            dxp = torch.empty((n, D), dtype=torch.float32, device=dev)
            G.gemm(_native.GEMM_NN, _native.EPI_F32, n, D, Vp, dlogits, Vp, w16, D, dxp, D, workspace=ws, splits=splits,
                   extent_dev=n_valid)
            if ctx.prepermuted:                              # the caller's _PermuteLiveRows zeroes and scatters
                dh = dxp.to(h_dtype)
            else:
                live = (torch.arange(n, device=dev) < n_valid)[:, None]      # rows past the extent were never written
                dxp = torch.where(live, dxp, torch.zeros((), dtype=torch.float32, device=dev))
                dh = torch.empty((n, D), dtype=h_dtype, device=dev)
                dh.index_copy_(0, perm, dxp.to(h_dtype))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwp = torch.empty((Vp, D), dtype=torch.float32, device=dev)
            dbp = torch.empty(Vp, dtype=torch.float32, device=dev)
            G.gemm(_native.GEMM_TN, _native.EPI_F32, Vp, D, n, dlogits, Vp, hp, D, dwp, D, colsum=dbp, splits=1,
                   extent_dev=n_valid)
            dw = dwp[:V]
            db = dbp[:V] if has_bias else None
        return dh, dw, db, None, None, None


def sparse_lm_loss(h: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], labels: torch.Tensor,
                   ignore_index: int = -1) -> torch.Tensor:
    """mean over labelled tokens of CE(h W^T + b, labels); h (..., D), labels (...)."""
    D = h.shape[-1]
    return _SparseLMLoss.apply(h.reshape(-1, D), weight, bias, labels.reshape(-1), int(ignore_index))


def usable(h: torch.Tensor, weight: torch.Tensor) -> bool:
    return G.usable(h, weight.shape[1], (weight.shape[0] + 7) // 8 * 8) and weight.shape[1] % 8 == 0


def transform_supported(transform, hidden: torch.Tensor) -> bool:
    """Can the head's transform (dense -> gelu -> LayerNorm, modules/heads/pretrain_head.py:8-20) run on the labelled rows
    only, on the native kernels?"""
    from ...modules.layers import fused_norm
    d = hidden.shape[-1]
    dense, norm = transform.dense, transform.LayerNorm
    probe = torch.empty(0, dtype=torch.bfloat16, device=hidden.device)
    return (getattr(transform.transform_act_fn, "__name__", "") == "gelu" and dense.in_features == d == dense.out_features
            and G.usable(probe, d, d) and fused_norm.supported(hidden.reshape(-1, d)[:1].float(), probe.new_empty((1, d)), norm))


class LazyLMLogits:
    """What the masked-LM head hands to the loss when the fused path is on: the hidden states and the head's parameters.
    `lm_cls_loss` turns it into the loss without full logits; anything else that wants the (B, L, V) tensor calls
    `materialize()`.  With `transform` (the head's dense -> gelu -> LayerNorm, row-wise) given, `hidden` is the head's
    INPUT and the transform itself runs behind the row selection, on the ~15 % of token rows that carry a label -- the
    other rows' transform outputs reach no loss (reference: pretrain_head.py:22-30 computes them for every token)."""

    def __init__(self, hidden: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], transform=None):
        self.hidden, self.weight, self.bias, self.transform = hidden, weight, bias, transform

    @property
    def shape(self):
        return (*self.hidden.shape[:-1], self.weight.shape[0])

    def materialize(self) -> torch.Tensor:
        h = self.hidden if self.transform is None else self.transform(self.hidden)
        return torch.nn.functional.linear(h, self.weight, self.bias)

    def loss(self, labels: torch.Tensor, ignore_index: int = -1) -> torch.Tensor:
        if self.transform is None:
            return sparse_lm_loss(self.hidden, self.weight, self.bias, labels, ignore_index)
        from ...modules.layers.fused_norm import add_dropout_layer_norm
        D = self.hidden.shape[-1]
        x = self.hidden.reshape(-1, D)
        plan = row_plan(labels, self.weight.shape[0], int(ignore_index))
        perm, _, n_valid = plan
        xs = _PermuteLiveRows.apply(x, perm, n_valid)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            y = G.linear_gelu(xs, self.transform.dense, rows_dev=n_valid)
        # LayerNorm(y) through the residual kernel with a zero residual: y fp32 = LN(0 + y) on the live rows
        zero = _zero_rows(x.shape[0], D, x.device)
        h = add_dropout_layer_norm(zero, y, self.transform.LayerNorm, 0.0, False, rows_dev=n_valid)
        return _SparseLMLoss.apply(h, self.weight, self.bias, labels.reshape(-1), int(ignore_index), plan)
