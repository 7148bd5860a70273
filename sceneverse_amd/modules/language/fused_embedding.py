"""Embedding block of the BERT text encoder with a deterministic, sort-free word-table gradient.

Reference: HF `BertEmbeddings.forward` behind modules/language/bert.py:21-26 -- word + token-type + position
lookups, LayerNorm, dropout.  Same values in the same order here; what changes is the backward:
  * word table (30 522 x 768): `gps_embedding_grad` (libgps_hip.so): first-occurrence marking + one wave per
    distinct id adding its duplicates in ascending token order, instead of torch's sort / segment /
    scatter pipeline (~60 launches, 1.27 ms per GPS step for the two BERT passes);
  * token types: the encoder is only ever called without token_type_ids (bert.py:24), i.e. every token has
    type 0 -- the lookup is a broadcast of row 0 and its gradient a plain reduction, not a 19 200-way
    duplicate scatter;
  * positions: rows [0, L) broadcast over the batch, as in HF.
GPU only; CPU tensors keep the HF module (modules/language/bert.py decides).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ... import _native


class _WordLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids: torch.Tensor, weight: torch.Tensor, padding_idx: int):
        ctx.save_for_backward(ids)
        ctx.meta = (weight.shape[0], weight.shape[1], int(padding_idx))
        return F.embedding(ids, weight, padding_idx if padding_idx >= 0 else None)

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        (ids,) = ctx.saved_tensors
        num_rows, d, padding_idx = ctx.meta
        return None, embedding_grad(ids, dy, num_rows, padding_idx), None


def embedding_grad(ids: torch.Tensor, dy: torch.Tensor, num_rows: int, padding_idx: int = -1) -> torch.Tensor:
    """ids (...) int64, dy (..., d) fp32 -> dense (num_rows, d) fp32 table gradient (gps_embedding_grad)."""
    d = dy.shape[-1]
    dy2 = dy.reshape(-1, d)
    if dy2.dtype != torch.float32:
        dy2 = dy2.float()
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    ids1 = ids.reshape(-1)
    if ids1.dtype != torch.int64 or not ids1.is_contiguous():
        ids1 = ids1.to(torch.int64).contiguous()
    n = ids1.numel()
    out = torch.empty((num_rows, d), dtype=torch.float32, device=dy2.device)
    scratch = torch.empty(2 * num_rows, dtype=torch.int32, device=dy2.device)
    from ...pointnet2._ext import _timed
    with torch.cuda.device(dy2.device), _timed(f"embedding_grad(n={n},rows={num_rows},d={d})",
                                               4 * n * d + 8 * n + 4 * num_rows * d):
        st = _native.load().gps_embedding_grad(n, d, num_rows, ids1.data_ptr(), dy2.data_ptr(), dy2.stride(0),
                                               int(padding_idx), scratch.data_ptr(), out.data_ptr(),
                                               torch.cuda.current_stream(dy2.device).cuda_stream)
    _native.check(st, "embedding_grad")
    return out


def supported(emb, input_ids: torch.Tensor) -> bool:
    w = emb.word_embeddings.weight
    return (input_ids.is_cuda and input_ids.dim() == 2 and w.is_cuda and w.dtype == torch.float32
            and w.shape[1] % 4 == 0 and w.shape[1] <= 2048
            and input_ids.shape[1] <= emb.position_embeddings.weight.shape[0]
            and emb.word_embeddings.max_norm is None and not emb.word_embeddings.scale_grad_by_freq
            and not emb.word_embeddings.sparse)


def bert_embeddings(emb, input_ids: torch.Tensor) -> torch.Tensor:
    """HF BertEmbeddings.forward(input_ids) for token_type_ids = None, position_ids = None."""
    L = input_ids.shape[1]
    pad = emb.word_embeddings.padding_idx
    x = _WordLookup.apply(input_ids, emb.word_embeddings.weight, -1 if pad is None else int(pad))
    x = x + emb.token_type_embeddings.weight[0]          # every token has type 0 (the HF buffer of zeros)
    x = x + emb.position_embeddings.weight[:L]
    return emb.dropout(emb.LayerNorm(x))


def bert_embeddings_multi(emb, texts) -> list:
    """The same block for several id tensors [(B_i, L_i), ...] with ONE word-table lookup over all their tokens, so
    that the backward pass fills the (30 522 x 768) table gradient once (one zero-fill + one gps_embedding_grad launch
    over every token of the step) instead of once per text; positions, type row, LayerNorm and dropout per text."""
    pad = emb.word_embeddings.padding_idx
    flat = torch.cat([ids.reshape(-1) for ids in texts]) if len(texts) > 1 else texts[0].reshape(-1)
    words = _WordLookup.apply(flat, emb.word_embeddings.weight, -1 if pad is None else int(pad))
    outs, r0 = [], 0
    for ids in texts:
        B, L = ids.shape
        x = words[r0:r0 + B * L].view(B, L, -1)
        r0 += B * L
        x = x + emb.token_type_embeddings.weight[0]
        x = x + emb.position_embeddings.weight[:L]
        outs.append(emb.dropout(emb.LayerNorm(x)))
    return outs
