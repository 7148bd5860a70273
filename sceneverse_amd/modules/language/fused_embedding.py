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
    scratch = torch.empty(int(_native.load().gps_embedding_grad_scratch_ints(n, num_rows, d)), dtype=torch.int32, device=dy2.device)
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


class _BertEmbedRows(torch.autograd.Function):
    """y, y16 = dropout(LayerNorm(word[ids] + type[0] + pos_table[pos])) for a flat list of token rows, one launch per
    direction (gps_bert_embed_forward / backward); rows past *rows_dev are neither computed nor written."""

    @staticmethod
    def forward(ctx, ids, pos, word_w, pos_w, type_w, gamma, beta, eps: float, p_drop: float, seed_dev, rows_dev,
                padding_idx: int, cu_rows=None, poison_dev=None):
        n, d = ids.numel(), word_w.shape[1]
        dev = word_w.device
        ids = ids.reshape(-1).to(torch.int64).contiguous()
        pos = pos.reshape(-1).to(torch.int64).contiguous()
        type0 = type_w[0].contiguous()
        y = torch.empty((n, d), dtype=torch.float32, device=dev)
        y16 = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
        mean = torch.empty(n, dtype=torch.float32, device=dev)
        rstd = torch.empty(n, dtype=torch.float32, device=dev)
        from ...pointnet2._ext import _timed
        from ..layers.fused_norm import _ptr, _row_fraction
        with torch.cuda.device(dev), _timed(f"bert_embed_forward(rows={n},d={d})", n * (d * (4 + 4 + 2) + 16),
                                            work_fraction=_row_fraction(rows_dev, n)):
            st = _native.load().gps_bert_embed_forward(
                n, d, ids.data_ptr(), pos.data_ptr(), word_w.data_ptr(), pos_w.data_ptr(), type0.data_ptr(),
                gamma.data_ptr(), beta.data_ptr(), float(eps), float(p_drop), 0, _ptr(seed_dev), y.data_ptr(),
                y16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(rows_dev), _ptr(poison_dev),
                torch.cuda.current_stream(dev).cuda_stream)
        _native.check(st, "bert_embed_forward")
        ctx.save_for_backward(ids, pos, word_w, pos_w, type0, gamma, mean, rstd, seed_dev, rows_dev, cu_rows)
        ctx.meta = (float(p_drop), int(padding_idx), type_w.shape[0])
        return y, y16

    @staticmethod
    def backward(ctx, dy, dy16):
        ids, pos, word_w, pos_w, type0, gamma, mean, rstd, seed_dev, rows_dev, cu_rows = ctx.saved_tensors
        p_drop, padding_idx, n_types = ctx.meta
        n, d = ids.numel(), word_w.shape[1]
        dev = word_w.device
        if dy is None:
            dy = torch.zeros((n, d), dtype=torch.float32, device=dev)
        dy = dy.reshape(n, d).float().contiguous()
        dy16 = dy16.reshape(n, d).to(torch.bfloat16).contiguous() if dy16 is not None else None
        lib = _native.load()
        parts = int(lib.gps_bert_embed_partial_rows(n))
        part = torch.empty((2, parts, d), dtype=torch.float32, device=dev)
        dz = torch.empty((n, d), dtype=torch.float32, device=dev)
        from ...pointnet2._ext import _timed
        from ..layers.fused_norm import _ptr, _reduce_scratch, _row_fraction
        with torch.cuda.device(dev), _timed(f"bert_embed_backward(rows={n},d={d})", n * (d * (4 + 2 + 4 + 4) + 24),
                                            work_fraction=_row_fraction(rows_dev, n)):
            st = lib.gps_bert_embed_backward(
                n, d, dy.data_ptr(), _ptr(dy16), ids.data_ptr(), pos.data_ptr(), word_w.data_ptr(), pos_w.data_ptr(),
                type0.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_drop, 0, _ptr(seed_dev),
                dz.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), _ptr(rows_dev), torch.cuda.current_stream(dev).cuda_stream)
        _native.check(st, "bert_embed_backward")
        sums = torch.empty((2, d), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = lib.gps_ln_reduce_partials(parts, d, part.data_ptr(), sums.data_ptr(), _reduce_scratch(dev, d).data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream)
        _native.check(st, "ln_reduce_partials")
        d_word = embedding_grad(ids, dz, word_w.shape[0], padding_idx)
        if cu_rows is not None:      # whole sequences end to end, position = offset inside the sequence
            d_pos = torch.empty((pos_w.shape[0], d), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev), _timed(f"bert_position_grad(seqs={cu_rows.numel() - 1},d={d})", n * d * 4,
                                                work_fraction=_row_fraction(rows_dev, n)):
                st = lib.gps_bert_position_grad(cu_rows.numel() - 1, pos_w.shape[0], d, cu_rows.data_ptr(), dz.data_ptr(),
                                                d_pos.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            _native.check(st, "bert_position_grad")
        else:
            d_pos = embedding_grad(pos, dz, pos_w.shape[0], -1)
        d_type = torch.zeros((n_types, d), dtype=torch.float32, device=dev)
        d_type[0] = d_pos.sum(0)                 # every row has exactly one position: sum of all rows' dz
        return None, None, d_word, d_pos, d_type, sums[0], sums[1], None, None, None, None, None, None, None


def rows_supported(emb) -> bool:
    w, ln = emb.word_embeddings.weight, emb.LayerNorm
    d = w.shape[1]
    return (w.is_cuda and w.dtype == torch.float32 and d % 256 == 0 and d <= 1024
            and emb.position_embeddings.weight.dtype == torch.float32 and emb.token_type_embeddings.weight.dtype == torch.float32
            and ln.elementwise_affine and ln.bias is not None and ln.weight.dtype == torch.float32
            and emb.word_embeddings.max_norm is None and not emb.word_embeddings.scale_grad_by_freq
            and not emb.word_embeddings.sparse)


def bert_embeddings_rows(emb, ids: torch.Tensor, pos: torch.Tensor, rows_dev=None, training: bool = False, cu_rows=None,
                         poison_dev=None):
    """HF BertEmbeddings for a flat list of (token id, position) rows -> (y fp32 (n, d), y bf16 (n, d)); rows at or past
    the device-side count `rows_dev` are left unwritten.  cu_rows (int32, n_seq + 1): promise that the live rows are whole
    sequences laid end to end (sequence s = rows cu_rows[s] .. cu_rows[s + 1] - 1) with pos = offset inside the sequence
    -- the position-table gradient then takes the per-position form (gps_bert_position_grad).  poison_dev (int32 device
    word, the plan's `violation`): non-zero turns every output row into NaN."""
    p = float(emb.dropout.p) if training else 0.0
    seed_dev = None
    if p > 0.0:
        from ..layers.fused_attention import _next_device_seed
        seed_dev = _next_device_seed(ids.device)
    pad = emb.word_embeddings.padding_idx
    ln = emb.LayerNorm
    return _BertEmbedRows.apply(ids, pos, emb.word_embeddings.weight, emb.position_embeddings.weight,
                                emb.token_type_embeddings.weight, ln.weight, ln.bias, ln.eps, p, seed_dev, rows_dev,
                                -1 if pad is None else int(pad), cu_rows, poison_dev)


class VarlenPlan:
    """Index tensors of the variable-length text path (gps_varlen_plan, one launch): views into three allocations."""
    __slots__ = ("lens", "cu", "order", "q_limit", "n_valid", "n_live_full", "rows_tail", "ids", "pos", "inv", "sel",
                 "valid", "violation")


def varlen_plan_supported(texts) -> bool:
    return (0 < len(texts) <= 8 and sum(ids.shape[0] for ids, _ in texts) <= 8192
            and all(ids.dtype == torch.int64 and ids.dim() == 2 and m.shape == ids.shape and ids.is_cuda and m.is_cuda
                    and m.element_size() in (1, 2, 4, 8) and not m.is_complex() for ids, m in texts))


def varlen_plan(texts, n_seq_full: int = 0) -> VarlenPlan:
    """texts = [(ids (B_i, L_i) int64, mask (B_i, L_i)), ...] with masks that are non-empty PREFIXES of their rows;
    n_seq_full = number of leading sequences (whole texts) that are read at every token (0: no [CLS]-only tail).
    Everything `_fast_forward_varlen` needs from the masks, element for element what the torch formulation gives."""
    dev = texts[0][0].device
    S = sum(ids.shape[0] for ids, _ in texts)
    T = sum(ids.numel() for ids, _ in texts)
    T_full, seq = 0, 0
    for ids, _ in texts:
        if seq == n_seq_full:
            break
        seq += ids.shape[0]
        T_full += ids.numel()
    tail = 0 < n_seq_full < S
    n_sel = (S - n_seq_full) + T_full if tail else 0
    i32 = torch.empty(4 * S + 5, dtype=torch.int32, device=dev)
    i64 = torch.empty(3 * T + n_sel, dtype=torch.int64, device=dev)
    valid = torch.empty(T, dtype=torch.bool, device=dev)
    arr = (_native.VarlenText * len(texts))()
    keep = []
    for i, (ids, m) in enumerate(texts):
        ids_c = ids if ids.is_contiguous() else ids.contiguous()
        m_c = m if m.is_contiguous() else m.contiguous()
        keep += [ids_c, m_c]
        arr[i].ids, arr[i].mask = ids_c.data_ptr(), m_c.data_ptr()
        arr[i].mask_elem_bytes, arr[i].mask_is_float = m_c.element_size(), int(m_c.is_floating_point())
        arr[i].n_seq, arr[i].len = ids.shape[0], ids.shape[1]
    with torch.cuda.device(dev):
        st = _native.load().gps_varlen_plan(arr, len(texts), int(n_seq_full), i32.data_ptr(), i64.data_ptr(),
                                            valid.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _native.check(st, "varlen_plan")
    p = VarlenPlan()
    p.lens, p.cu, p.order, p.q_limit = i32[:S], i32[S:2 * S + 1], i32[2 * S + 1:3 * S + 1], i32[3 * S + 1:4 * S + 1]
    p.n_valid, p.n_live_full, p.rows_tail = i32[4 * S + 1:4 * S + 2], i32[4 * S + 2:4 * S + 3], i32[4 * S + 3:4 * S + 4]
    p.violation = i32[4 * S + 4:4 * S + 5]
    p.ids, p.pos, p.inv = i64[:T], i64[T:2 * T], i64[2 * T:3 * T]
    p.sel = i64[3 * T:] if tail else None
    p.valid = valid
    return p
