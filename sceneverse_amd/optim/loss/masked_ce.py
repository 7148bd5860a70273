"""Row-sparse cross-entropy on libgps_hip.so (gps_masked_ce_forward/backward): the masked-LM loss of
the reference (optim/loss/loss.py:56-61) without touching the > 90 % of rows whose label is
`ignore_index`.  GPU tensors only; `lm_cls_loss` keeps F.cross_entropy for CPU tensors.
Deviation from F.cross_entropy: a label outside [0, V) that is not `ignore_index` is treated as ignored (the kernels
must not index outside the row) where torch raises a device assert; the data pipeline only produces -1 or valid ids."""
from __future__ import annotations

import torch

from ... import _native


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, labels: torch.Tensor, ignore_index: int) -> torch.Tensor:
        assert logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1
        assert logits.dtype in (torch.bfloat16, torch.float32)
        n, v = logits.shape
        labels = labels.contiguous().to(torch.int64)
        loss = torch.empty(n, dtype=torch.float32, device=logits.device)
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        from ...pointnet2._ext import _timed
        with torch.cuda.device(logits.device), _timed("masked_ce_forward", n * v * logits.element_size() // 8):
            st = _native.load().gps_masked_ce_forward(
                n, v, int(logits.dtype == torch.bfloat16), logits.data_ptr(), logits.stride(0),
                labels.data_ptr(), int(ignore_index), loss.data_ptr(), lse.data_ptr(),
                torch.cuda.current_stream().cuda_stream)
        _native.check(st, "masked_ce_forward")
        ctx.save_for_backward(logits, labels, lse)
        ctx.ignore_index = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, grad_rows: torch.Tensor):
        logits, labels, lse = ctx.saved_tensors
        n, v = logits.shape
        grad_rows = grad_rows.contiguous().float()
        dlogits = torch.empty((n, v), dtype=logits.dtype, device=logits.device)
        from ...pointnet2._ext import _timed
        with torch.cuda.device(logits.device), _timed("masked_ce_backward", n * v * logits.element_size()):
            st = _native.load().gps_masked_ce_backward(
                n, v, int(logits.dtype == torch.bfloat16), logits.data_ptr(), logits.stride(0),
                labels.data_ptr(), ctx.ignore_index, lse.data_ptr(), grad_rows.data_ptr(),
                dlogits.data_ptr(), dlogits.stride(0), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "masked_ce_backward")
        return dlogits, None, None


def masked_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -1) -> torch.Tensor:
    """logits (..., V), labels (...) -> mean over the positions whose label != ignore_index."""
    v = logits.shape[-1]
    flat = logits.reshape(-1, v)
    if flat.stride(1) != 1:
        flat = flat.contiguous()
    lab = labels.reshape(-1)
    rows = _MaskedCE.apply(flat, lab, ignore_index)
    valid = ((lab != ignore_index) & (lab >= 0) & (lab < v)).sum()
    return rows.sum() / valid
