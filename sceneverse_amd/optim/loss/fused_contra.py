"""The two contrastive losses of the pre-training step on libgps_hip.so: one launch per direction each
(csrc/gps_contrastive.hip) instead of ~55 torch launches of 3 - 10 us per step.

Reference: optim/loss/contra_loss.py:22-43 (TextObjWithinBatch, cross-entropy branch) and :11-17 + :57-60 / :79-83
(_symmetric_clip_loss over F.normalize'd rows with the clamped logit scale).  fp32 GPU tensors; anything else keeps
the torch composition in contra_loss.py (CPU tensors, the BCE branch)."""
from __future__ import annotations

import torch

from ... import _native

_TICKETS = {}


def _ticket(device: torch.device, site: str) -> torch.Tensor:
    """One zeroed arrival counter per device and call site (the kernels leave it at zero)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), site)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def _rows_ok(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.shape[-1] % 4 == 0 and t.shape[-1] <= 8192 for t in ts)


class _TextObjCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, text, labels, masks, eps: float, ignore_index: int):
        B, O, D = obj.shape
        obj, text = obj.contiguous(), text.contiguous()
        labels = labels.reshape(-1).to(torch.int64).contiguous()
        masks = masks.to(torch.uint8).contiguous()
        dev = obj.device
        f = dict(dtype=torch.float32, device=dev)
        cosv, prob, inv_o = torch.empty((B, O), **f), torch.empty((B, O), **f), torch.empty((B, O), **f)
        inv_t, loss_rows, scal = torch.empty(B, **f), torch.empty(B, **f), torch.empty(2, **f)
        with torch.cuda.device(dev):
            st = _native.load().gps_text_obj_ce_forward(
                B, O, D, obj.data_ptr(), text.data_ptr(), labels.data_ptr(), masks.data_ptr(), float(eps), int(ignore_index),
                cosv.data_ptr(), prob.data_ptr(), inv_o.data_ptr(), inv_t.data_ptr(), loss_rows.data_ptr(), scal.data_ptr(),
                _ticket(dev, "text_obj").data_ptr(), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "text_obj_ce_forward")
        ctx.save_for_backward(obj, text, labels, cosv, prob, inv_o, inv_t, scal)
        ctx.meta = (float(eps), int(ignore_index))
        return scal[0]

    @staticmethod
    def backward(ctx, g):
        obj, text, labels, cosv, prob, inv_o, inv_t, scal = ctx.saved_tensors
        eps, ignore_index = ctx.meta
        B, O, D = obj.shape
        g = g.reshape(1).float().contiguous()
        dobj = torch.empty_like(obj) if ctx.needs_input_grad[0] else None
        dtext = torch.empty_like(text) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(obj.device):
            st = _native.load().gps_text_obj_ce_backward(
                B, O, D, obj.data_ptr(), text.data_ptr(), labels.data_ptr(), eps, ignore_index, cosv.data_ptr(),
                prob.data_ptr(), inv_o.data_ptr(), inv_t.data_ptr(), scal.data_ptr(), g.data_ptr(),
                dobj.data_ptr() if dobj is not None else None, dtext.data_ptr() if dtext is not None else None,
                torch.cuda.current_stream().cuda_stream)
        _native.check(st, "text_obj_ce_backward")
        return dobj, dtext, None, None, None, None


_LDS_LIMIT = 64 * 1024          # the launches request dynamic LDS without a MaxDynamicSharedMemorySize grant (gps_contrastive.hip)


def text_obj_ce_usable(obj, text, labels, masks) -> bool:
    return (obj.dim() == 3 and text.dim() == 2 and _rows_ok(obj, text) and obj.shape[1] <= 4096
            and 4 * (obj.shape[2] + obj.shape[1] + 16) <= _LDS_LIMIT
            and labels.numel() == obj.shape[0] and masks.shape == obj.shape[:2] and obj.shape[0] > 0)


def text_obj_ce(obj, text, labels, masks, eps: float = 1e-12, ignore_index: int = -100):
    """mean_b CE(masked <normalize(obj[b]), normalize(text[b])>, labels[b]) -- TextObjWithinBatch's cross-entropy branch."""
    return _TextObjCE.apply(obj, text, labels, masks, eps, ignore_index)


class _ClipLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, scale, normalize: bool, max_scale: float, eps: float):
        n, D = a.shape
        a, b = a.contiguous(), b.contiguous()
        scale32 = scale.detach().reshape(1).float().contiguous()
        dev = a.device
        f = dict(dtype=torch.float32, device=dev)
        M = torch.empty((n, n), **f)
        lse_row, lse_col, inv_a, inv_b, rows = (torch.empty(n, **f) for _ in range(5))
        loss = torch.empty(1, **f)
        with torch.cuda.device(dev):
            st = _native.load().gps_clip_loss_forward(
                n, D, int(bool(normalize)), a.data_ptr(), b.data_ptr(), scale32.data_ptr(), float(max_scale), float(eps),
                M.data_ptr(), lse_row.data_ptr(), lse_col.data_ptr(), inv_a.data_ptr(), inv_b.data_ptr(), rows.data_ptr(),
                loss.data_ptr(), _ticket(dev, "clip_fwd").data_ptr(), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "clip_loss_forward")
        ctx.save_for_backward(a, b, scale32, M, lse_row, lse_col, inv_a, inv_b)
        ctx.meta = (bool(normalize), float(max_scale), float(eps), scale.dtype, scale.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        a, b, scale32, M, lse_row, lse_col, inv_a, inv_b = ctx.saved_tensors
        normalize, max_scale, eps, s_dtype, s_shape = ctx.meta
        n, D = a.shape
        g = g.reshape(1).float().contiguous()
        feats = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        da = torch.empty_like(a) if feats else None
        db = torch.empty_like(b) if feats else None
        dscale = torch.empty(1, dtype=torch.float32, device=a.device)
        rows = torch.empty(n, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            st = _native.load().gps_clip_loss_backward(
                n, D, int(normalize), a.data_ptr(), b.data_ptr(), scale32.data_ptr(), max_scale, eps, M.data_ptr(),
                lse_row.data_ptr(), lse_col.data_ptr(), inv_a.data_ptr(), inv_b.data_ptr(), g.data_ptr(),
                da.data_ptr() if feats else None, db.data_ptr() if feats else None, dscale.data_ptr(), rows.data_ptr(),
                _ticket(a.device, "clip_bwd").data_ptr(), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "clip_loss_backward")
        ds = dscale.reshape(s_shape).to(s_dtype) if ctx.needs_input_grad[2] else None
        return (da if ctx.needs_input_grad[0] else None), (db if ctx.needs_input_grad[1] else None), ds, None, None, None


def clip_loss_usable(a, b, scale) -> bool:
    return (a.dim() == 2 and a.shape == b.shape and _rows_ok(a, b) and 0 < a.shape[0] <= 8192 and torch.is_tensor(scale)
            and 4 * (2 * a.shape[1] + 2 * a.shape[0] + 16) <= _LDS_LIMIT
            and scale.is_cuda and scale.numel() == 1)


def clip_loss(a, b, scale, normalize: bool, max_scale: float = 100.0, eps: float = 1e-12):
    """(CE(s a_n b_n^T, arange) + CE(s b_n a_n^T, arange)) / 2 with s = min(scale, max_scale); a_n / b_n = the rows
    normalised (normalize=True) or as given."""
    return _ClipLoss.apply(a, b, scale, normalize, max_scale, eps)
