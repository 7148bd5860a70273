"""BERT text encoder wrapper (reference modules/language/bert.py:7-26).  The arithmetic lives in
HuggingFace `transformers` (out of scope); this wrapper keeps the
registry name, constructor arguments, `self.model` attribute (checkpoint keys `model.*`) and
forward contract.  Extra, for offline machines: `weights=None` (or `random_init=True`) builds
`BertModel(BertConfig(...))` with random weights instead of calling `from_pretrained`."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..build import LANGUAGE_REGISTRY

# GPU fast path of the encoder stack: the HuggingFace PARAMETERS and embedding
# block are used as they are, but each BertLayer runs as
#   one packed QKV GEMM -> fused attention core (libgps_hip.so, up to 512 tokens; torch SDPA beyond)
#   -> dense -> fused residual+dropout+LayerNorm -> dense+GELU -> dense -> fused residual+dropout+LN
# instead of HF's op-by-op formulation (3 projection GEMMs, separate dropout/add/LayerNorm/casts).
# Same mathematics as transformers.models.bert.modeling_bert.BertLayer (post-norm, exact GELU,
# key-padding mask from the attention mask, dropout on probabilities and on both branches).
_FAST = True
_FAST_EMB = True
# Variable-length form of the fast path: the VALID tokens of all texts are compacted to the front of one row batch
# (stable: every sequence stays contiguous), and every kernel of the stack works on those rows only -- GEMMs and
# LayerNorms through a device-side row count, attention through per-sequence row offsets.  The reference runs BERT on
# the padded (B, L) batch (modules/language/bert.py:21-26) and then only ever reads valid positions (padded keys are
# masked in every attention, padded text rows carry label -1, the caption is read at [CLS] only), so nothing
# observable changes; at the bench workload 45 % of the text rows are padding (sentence 6..50 of 50, caption 30..300
# of 300 tokens).  Shapes stay static (the counts live on the device): the step remains one replayable HIP graph.
_VARLEN = True
_PREFIX_CHECKS = 4      # eager forwards per encoder whose masks are verified to be non-empty prefixes (see below)


def set_fast_bert(flag: bool) -> None:
    global _FAST
    _FAST = bool(flag)


_CLS_TAIL = True
# the index plan of the variable-length path (lens, row offsets, compaction, dispatch order, tail selection) from ONE
# launch of gps_varlen_plan instead of ~30 torch launches; False = the torch formulation (kept as the test's reference)
_PLAN_KERNEL = True


def set_cls_tail(flag: bool) -> None:
    """Last-layer tail on the rows that reach an output only (texts read at [CLS]); off = every live row (A/B, tests)."""
    global _CLS_TAIL
    _CLS_TAIL = bool(flag)


def set_varlen(flag: bool) -> None:
    """False: the fast path keeps the padded (B, L) row batch (A/B runs, parity tests against the padded form)."""
    global _VARLEN
    _VARLEN = bool(flag)


class _ZeroDeadRows(torch.autograd.Function):
    """Identity on (T, D) rows whose backward zeroes the rows >= *n_valid: gradients of rows no kernel ever wrote
    (undefined memory) must not reach the embedding tables."""

    @staticmethod
    def forward(ctx, x, n_valid):
        ctx.save_for_backward(n_valid)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        (n_valid,) = ctx.saved_tensors
        live = torch.arange(dy.shape[0], device=dy.device, dtype=torch.int32)[:, None] < n_valid
        return torch.where(live, dy, torch.zeros((), dtype=dy.dtype, device=dy.device)), None


class _UnpadRows(torch.autograd.Function):
    """out[e] = x[src[e]] where live[e], else 0 -- the compact token rows back in a text's (B x L) layout.  The map from
    live positions to compact rows is injective, so the backward pass is a row COPY (dead positions into a dump row), not
    the zero-fill + atomic index_add_ autograd derives for index_select (39 us on the 3 264 x 768 tail batch)."""

    @staticmethod
    def forward(ctx, x, src, live):
        ctx.rows = x.shape[0]
        ctx.native = (x.is_cuda and x.dim() == 2 and (x.shape[1] * x.element_size()) % 16 == 0 and src.dtype == torch.int64)
        if ctx.native:
            # dead positions address row -1: gps_rows_move writes zeros for them (forward) and drops them (backward)
            idx = torch.where(live[:, 0], src, torch.full((), -1, dtype=src.dtype, device=src.device))
            ctx.save_for_backward(idx)
            x = x.contiguous()
            out = torch.empty((src.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
            _SelectRows._move(src.shape[0], x, idx, out, None, None, False)
            return out
        ctx.save_for_backward(src, live)
        rows = x.index_select(0, src)
        return torch.where(live, rows, torch.zeros((), dtype=rows.dtype, device=rows.device))

    @staticmethod
    def backward(ctx, dout):
        if ctx.native:
            (idx,) = ctx.saved_tensors
            dout = dout.contiguous()
            dx = torch.zeros((ctx.rows, dout.shape[1]), dtype=dout.dtype, device=dout.device)
            _SelectRows._move(idx.shape[0], dout, None, dx, idx, None, False)
            return dx, None, None
        src, live = ctx.saved_tensors
        n = ctx.rows
        dx = torch.zeros((n + 1, dout.shape[1]), dtype=dout.dtype, device=dout.device)     # row n: dump
        dst = torch.where(live[:, 0], src, torch.full_like(src, n))
        dx.index_copy_(0, dst, dout.contiguous())
        return dx[:n], None, None


class _SelectRows(torch.autograd.Function):
    """out[r] = x[sel[r]] for r < *rows_live, zeros past it (rows of any dtype; gps_rows_move); the live entries of `sel` are
    distinct, so the gradient is a zero-filled buffer + a row SCATTER of the live rows -- instead of index_select + a
    dead-row mask whose autograd gradient is where + zero-fill + atomic index_add_ (the [CLS]-tail selection: two
    tensors, ~ 75 us per step)."""

    @staticmethod
    def _move(n, src, src_idx, dst, dst_idx, rows_live, zero_dead):
        from ... import _native
        with torch.cuda.device(src.device):
            st = _native.load().gps_rows_move(n, src.shape[0], dst.shape[0], src.shape[1] * src.element_size(), src.data_ptr(),
                                              None if src_idx is None else src_idx.data_ptr(), dst.data_ptr(),
                                              None if dst_idx is None else dst_idx.data_ptr(),
                                              None if rows_live is None else rows_live.data_ptr(),
                                              int(zero_dead), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "rows_move")

    @staticmethod
    def forward(ctx, x, sel, rows_live):
        x = x.contiguous()
        out = torch.empty((sel.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
        _SelectRows._move(sel.shape[0], x, sel, out, None, rows_live, True)
        ctx.save_for_backward(sel, rows_live)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        sel, rows_live = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.zeros((ctx.rows, dout.shape[1]), dtype=dout.dtype, device=dout.device)
        _SelectRows._move(sel.shape[0], dout, None, dx, sel, rows_live, False)
        return dx, None, None


def select_rows(x: torch.Tensor, sel: torch.Tensor, rows_live: torch.Tensor) -> torch.Tensor:
    """x.index_select(0, sel) with the rows at or past *rows_live zeroed (forward) and ignored (backward)."""
    if x.is_cuda and x.dim() == 2 and (x.shape[1] * x.element_size()) % 16 == 0 and sel.dtype == torch.int64:
        return _SelectRows.apply(x, sel.contiguous(), rows_live)
    return _ZeroDeadRows.apply(x.index_select(0, sel), rows_live)


def set_fused_embedding(flag: bool) -> None:
    """Word-table gradient through libgps_hip.so (fused_embedding.py) inside the fast path; off = HF module."""
    global _FAST_EMB
    _FAST_EMB = bool(flag)


def _returning(value):
    """A generator that yields nothing and returns `value` (a stack without GEMM calls to pair)."""
    return value
    yield       # noqa: unreachable -- makes this a generator function


@LANGUAGE_REGISTRY.register()
class BERTLanguageEncoder(nn.Module):
    def __init__(self, cfg, weights="bert-base-uncased", hidden_size=768, num_hidden_layers=4,
                 num_attention_heads=12, type_vocab_size=2, random_init=False):
        super().__init__()
        from transformers import BertConfig, BertModel
        self.bert_config = BertConfig(hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                                      num_attention_heads=num_attention_heads,
                                      type_vocab_size=type_vocab_size)
        if weights is None or random_init:
            self.tokenizer = None
            self.model = BertModel(self.bert_config)
        else:
            from transformers import BertTokenizer
            self.tokenizer = BertTokenizer.from_pretrained(weights, do_lower_case=True)
            self.model = BertModel.from_pretrained(weights, config=self.bert_config)
        # which formulation the most recent forward took: "varlen" | "padded" (both on libgps_hip.so) | "hf" (the
        # HuggingFace module itself); bench.py refuses to report a number measured on a silent fallback
        self.last_path = None
        self._varlen_masks_ok = True
        self._prefix_checks_left = _PREFIX_CHECKS
        self._varlen_violation = None        # device word of the last variable-length forward (see check_varlen_masks)

    def _fast_ok(self, txt_ids) -> bool:
        from ..layers.transformers import _bf16_mode
        cfg = self.bert_config
        probe = self.model.embeddings.word_embeddings.weight
        return (_FAST and txt_ids.is_cuda and _bf16_mode(probe) and cfg.hidden_act == "gelu"
                and cfg.hidden_size == cfg.num_attention_heads * 64 and not cfg.is_decoder
                and getattr(cfg, "position_embedding_type", "absolute") == "absolute")

    def _fast_forward(self, txt_ids, txt_masks):
        return self._fast_forward_multi([(txt_ids, txt_masks)])[0]

    def _varlen_ok(self, texts) -> bool:
        from ..layers import gemm
        from ..layers.fused_attention import MAX_LEN
        from . import fused_embedding
        D = self.bert_config.hidden_size
        probe = torch.empty(0, dtype=torch.bfloat16, device=texts[0][0].device)
        return (gemm.enabled() and gemm.usable(probe, D, D) and self.bert_config.intermediate_size % 8 == 0
                and _FAST_EMB and all(ids.dim() == 2 and ids.shape[1] <= MAX_LEN
                                      and fused_embedding.supported(self.model.embeddings, ids) for ids, _ in texts))

    def _masks_are_prefixes(self, texts) -> bool:
        """The variable-length form promises `gps_bert_position_grad` that every text's valid tokens are a non-empty
        PREFIX of its row (position = offset inside the compacted sequence; the [CLS] row exists).  Right-padded
        tokenizer output (the reference: BertTokenizer(..., padding='max_length')) always is.  The property is data.
        EVERY batch is checked on the device: the plan kernel writes a violation word (gps_varlen_plan) that turns the
        embedding block's output -- hence the loss -- into NaN (`poison_dev`), inside captured graphs too, and that
        `check_varlen_masks()` reads at logging / evaluation points.  In addition the first `_PREFIX_CHECKS` eager
        forwards of an encoder (the warm-up steps that precede any graph capture) check on the host -- one sync -- and a
        batch with holes, left padding or an empty text switches this encoder to the padded row batch for good, which
        handles any mask like HF does."""
        if not self._varlen_masks_ok:
            return False
        if self._prefix_checks_left > 0 and not torch.cuda.is_current_stream_capturing():
            self._prefix_checks_left -= 1
            ok = torch.stack([((m[:, 1:] != 0) <= (m[:, :-1] != 0)).all() & (m[:, 0] != 0).all() for _, m in texts]).all()
            if not bool(ok.item()):
                import logging
                logging.getLogger("sceneverse_amd").warning(
                    "BERT: attention masks are not non-empty prefixes (holes / left padding / empty text): the "
                    "variable-length path is disabled for this encoder, the padded row batch is used instead")
                self._varlen_masks_ok = False
        return self._varlen_masks_ok

    def check_varlen_masks(self) -> None:
        """Raise if the last variable-length forward (eager or a graph replay) saw an attention mask that is not a
        non-empty prefix of its row; its outputs are NaN by construction.  One host sync: call at logging / evaluation
        points, not per step."""
        v = self._varlen_violation
        if v is not None and bool(v.item()):
            raise RuntimeError("BERT variable-length path: an attention mask has a hole, left padding or an empty row; "
                               "the outputs of that forward are NaN.  Use right-padded masks or set_varlen(False).")

    def _fast_forward_varlen(self, texts, cls_only=()):
        from ..layers import gemm
        return gemm.drive(self._fast_forward_varlen_gen(texts, cls_only))

    def _fast_forward_varlen_gen(self, texts, cls_only=()):
        """The encoder stack over the VALID tokens only (see _VARLEN above).  texts = [(ids (B_i, L_i), masks), ...];
        -> per text the last hidden state as (B_i, L_i, D) with zeros at padded positions, or (B_i, 1, D) = the
        [CLS] rows for the texts listed in `cls_only`.  A generator: the layers' GEMM calls are yielded (gemm.LinearOp /
        gemm.FFNOp), `gemm.drive` runs them one by one, `gemm.drive_pair` beside the object encoder's."""
        from ..layers import gemm
        from ..layers.fused_attention import fused_varlen_self_attention
        from ..layers.fused_norm import add_dropout_layer_norm
        from . import fused_embedding
        from .fused_embedding import _WordLookup
        m, H = self.model, self.bert_config.num_attention_heads
        emb = m.embeddings
        dev = texts[0][0].device
        T, S, cap = sum(ids.numel() for ids, _ in texts), sum(ids.shape[0] for ids, _ in texts), max(ids.shape[1] for ids, _ in texts)
        # Texts that are only read at [CLS] (`cls_only`, listed after the fully-read ones): in the LAST layer, everything
        # behind the attention core is row-wise, and of those texts only the first row of each sequence reaches an output
        # -- the other rows' results are never read and their gradients are exactly zero in the reference as well.  The
        # tail of the last layer (output projection, LayerNorm, FFN, LayerNorm) therefore runs on the rows
        #     [ first row of every [CLS]-only sequence | the live rows of the fully-read texts ]
        # (the compact buffer holds the fully-read texts' live rows first, so the second part is a prefix of it).
        n_full = sum(1 for ti in range(len(texts)) if ti not in cls_only)
        cls_tail = (_CLS_TAIL and 0 < n_full < len(texts) and all(ti >= n_full for ti in cls_only)
                    and len(m.encoder.layer) > 0)
        T_full = sum(texts[ti][0].numel() for ti in range(n_full)) if cls_tail else 0
        S_full = sum(texts[ti][0].shape[0] for ti in range(n_full)) if cls_tail else 0
        plan = None
        if _PLAN_KERNEL and fused_embedding.varlen_plan_supported(texts):
            # one launch instead of ~30 (gps_varlen_plan; masks are prefixes -- _masks_are_prefixes -- so the stable
            # "valid rows first" permutation has a closed form)
            plan = fused_embedding.varlen_plan(texts, S_full)
            lens, cu, order, n_valid, valid = plan.lens, plan.cu, plan.order, plan.n_valid, plan.valid
            violation = plan.violation
            ids_c, pos_c = plan.ids, plan.pos
            if cls_tail:
                sel, rows_tail, q_limit = plan.sel, plan.rows_tail, plan.q_limit
        else:
            ids_all = torch.cat([ids.reshape(-1) for ids, _ in texts])
            valid = torch.cat([(masks != 0).reshape(-1) for _, masks in texts])
            violation = torch.stack([(((m[:, 1:] != 0) > (m[:, :-1] != 0)).any() | (m[:, 0] == 0).any())
                                     for _, m in texts]).any().to(torch.int32).reshape(1)
            lens = torch.cat([(masks != 0).sum(dim=1) for _, masks in texts]).to(torch.int32)
            pos = torch.cat([torch.arange(ids.shape[1], device=dev).repeat(ids.shape[0]) for ids, _ in texts])
            n_valid = valid.sum(dtype=torch.int32).reshape(1)
            perm = torch.argsort(valid.logical_not().to(torch.uint8), stable=True)      # compact row r <- flat row perm[r]
            cu = torch.zeros(S + 1, dtype=torch.int32, device=dev)
            cu[1:] = torch.cumsum(lens, 0)
            order = torch.argsort(lens, descending=True).to(torch.int32)               # longest sequences dispatched first
            ids_c, pos_c = ids_all.index_select(0, perm), pos.index_select(0, perm)
            if cls_tail:
                n_live_full = valid[:T_full].sum(dtype=torch.int32).reshape(1)
                sel = torch.cat([cu[S_full:S].long(), torch.arange(T_full, device=dev)])
                rows_tail = n_live_full + (S - S_full)
                # last-layer attention: every query of the fully-read sequences, the first one of the [CLS]-only ones
                q_limit = torch.cat([lens[:S_full], torch.ones(S - S_full, dtype=torch.int32, device=dev)])
        self._varlen_violation = violation
        # embedding block on the compacted tokens (HF BertEmbeddings with token type 0): rows past n_valid hold pad ids
        training = self.training
        if fused_embedding.rows_supported(emb):
            # one launch: lookups + LayerNorm + dropout, fp32 and bf16 outputs, live rows only (gps_bert_embed_forward)
            x, x16 = fused_embedding.bert_embeddings_rows(emb, ids_c, pos_c, rows_dev=n_valid, training=training, cu_rows=cu,
                                                          poison_dev=violation)
        else:
            pad = emb.word_embeddings.padding_idx
            x = _WordLookup.apply(ids_c, emb.word_embeddings.weight, -1 if pad is None else int(pad))
            x = x + emb.token_type_embeddings.weight[0]
            x = x + emb.position_embeddings.weight.index_select(0, pos_c)
            x = emb.dropout(emb.LayerNorm(x))
            x = _ZeroDeadRows.apply(x, n_valid)
            x = x + torch.where(violation != 0, float("nan"), 0.0).to(x.dtype)      # same poison as the fused launch
            x16 = x
        last = len(m.encoder.layer) - 1
        for li, layer in enumerate(m.encoder.layer):
            sa, so = layer.attention.self, layer.attention.output
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                packed = yield gemm.LinearOp.of(x16, [sa.query, sa.key, sa.value], rows_dev=n_valid)
                ctx = fused_varlen_self_attention(packed, cu, S, cap, H, dropout_p=sa.dropout.p, training=training,
                                                  order=order, q_limit=q_limit if (cls_tail and li == last) else None)
                rows = n_valid
                if cls_tail and li == last:
                    # rows past `rows_tail` of the tail batch are never written by the extent-aware kernels, forward or
                    # backward: their (undefined) gradients must not be scattered back into the full row batch
                    ctx = select_rows(ctx, sel, rows_tail)
                    x, rows = select_rows(x, sel, rows_tail), rows_tail
                attn_out = yield gemm.LinearOp.of(ctx, [so.dense], rows_dev=rows)
                x, x16 = add_dropout_layer_norm(x, attn_out, so.LayerNorm, so.dropout.p, training, want_bf16=True,
                                                rows_dev=rows)
                ffn_out = yield gemm.FFNOp(x16, layer.intermediate.dense, layer.output.dense, "gelu", 0.0, training,
                                           rows_dev=rows)
                x, x16 = add_dropout_layer_norm(x, ffn_out, layer.output.LayerNorm, layer.output.dropout.p, training,
                                                want_bf16=True, rows_dev=rows)
        # back to the callers' layouts
        if plan is not None:
            inv = plan.inv
        else:
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(T, device=dev)
        outs, r0, s0 = [], 0, 0
        for ti, (ids, masks) in enumerate(texts):
            B, L = ids.shape
            if ti in cls_only:
                if cls_tail:                                                 # rows [0, S - S_full) of the tail batch
                    outs.append(x[s0 - S_full:s0 - S_full + B].view(B, 1, -1))
                else:
                    first = cu[s0:s0 + B].long()                             # compact row of every sequence's first token
                    outs.append(x.index_select(0, first).view(B, 1, -1))
            else:
                src = inv[r0:r0 + B * L]
                if cls_tail:      # tail batch: compact row c of a fully-read text sits at (S - S_full) + c; padded tokens
                    src = (src + (S - S_full)).clamp(max=x.shape[0] - 1)      # (masked below) may point past the batch
                outs.append(_UnpadRows.apply(x, src, valid[r0:r0 + B * L, None]).view(B, L, -1))
            r0 += B * L
            s0 += B
        return outs

    def _fast_forward_multi(self, texts, cls_only=(), as_gen: bool = False):
        """texts = [(ids (B_i, L_i), masks (B_i, L_i)), ...] -> [last hidden state (B_i, L_i, D), ...]
        (as_gen: a generator of the stack's GEMM calls that returns them, see _fast_forward_varlen_gen).
        Every row-wise operation of a layer (QKV / output / FFN GEMMs, residual + LayerNorm) runs ONCE over the
        token rows of all texts together -- the weights are the same, the rows independent -- and only the attention
        core runs per text (its own length and padding mask).  One text: the reference's call.  Two (the sentence
        and the 300-token scene caption of the pre-train step): the 3 200-row GEMMs of the short text, which alone
        fill a fraction of the chip, ride along with the 19 200-row ones, and each parameter gets ONE gradient
        instead of two that autograd then adds."""
        from ..layers import gemm
        from ..layers.fused_attention import fused_self_attention, supported as attn_supported
        from ..layers.fused_norm import add_dropout_layer_norm
        from . import fused_embedding
        m, H = self.model, self.bert_config.num_attention_heads
        if _VARLEN and self._varlen_ok(texts) and self._masks_are_prefixes(texts):
            self.last_path = "varlen"
            g = self._fast_forward_varlen_gen(texts, cls_only)
            return g if as_gen else gemm.drive(g)
        if as_gen:
            return _returning(self._fast_forward_multi(texts, cls_only))
        self.last_path = "padded"
        xs, shapes, pads = [], [], []
        fused_emb = _FAST_EMB and all(fused_embedding.supported(m.embeddings, ids) for ids, _ in texts)
        embs = fused_embedding.bert_embeddings_multi(m.embeddings, [ids for ids, _ in texts]) if fused_emb else None
        for ti, (ids, masks) in enumerate(texts):
            if fused_emb:
                e = embs[ti]                                    # same values, ONE sort-free table gradient for all texts
            else:
                e = m.embeddings(input_ids=ids)                 # (B, L, D) fp32 under autocast
            shapes.append(e.shape)
            xs.append(e.reshape(-1, e.shape[-1]))
            pads.append(masks == 0)
        D = xs[0].shape[-1]
        rows = [t.shape[0] for t in xs]
        x = xs[0] if len(xs) == 1 else torch.cat(xs, 0)         # (sum of B_i L_i, D): the row batch of every GEMM / LN
        x16 = x                                                 # bf16 copy of x once a fused LN made one
        training = self.training
        for layer in m.encoder.layer:
            sa, so = layer.attention.self, layer.attention.output
            native = gemm.usable(x16, D, D) and layer.intermediate.dense.out_features % 8 == 0
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                if native:
                    packed = gemm.packed_linear(x16, [sa.query, sa.key, sa.value])
                else:
                    w = torch.cat([sa.query.weight, sa.key.weight, sa.value.weight], 0)
                    b = torch.cat([sa.query.bias, sa.key.bias, sa.value.bias], 0)
                    packed = F.linear(x16, w, b)
                ctxs, r0 = [], 0
                for (B, L, _), pad, n in zip(shapes, pads, rows):
                    pk = packed[r0:r0 + n].view(B, L, 3 * D)
                    r0 += n
                    if attn_supported(D, H, L):
                        c = fused_self_attention(pk, H, None, pad, dropout_p=sa.dropout.p, training=training)
                    else:   # beyond the fused core's length limit: torch SDPA on views of the packed projection
                        q, k, v = pk.view(B, L, 3, H, D // H).permute(2, 0, 3, 1, 4)
                        c = F.scaled_dot_product_attention(
                            q, k, v, attn_mask=pad.logical_not()[:, None, None, :],
                            dropout_p=sa.dropout.p if training else 0.0)
                        c = c.transpose(1, 2).reshape(B, L, D)
                    ctxs.append(c.reshape(n, D))
                ctx = ctxs[0] if len(ctxs) == 1 else torch.cat(ctxs, 0)
                attn_out = gemm.linear(ctx, so.dense.weight, so.dense.bias) if native else so.dense(ctx)
                x, x16 = add_dropout_layer_norm(x, attn_out, so.LayerNorm, so.dropout.p, training, want_bf16=True)
                if native:      # dense + GELU + dense as two GEMMs with fused epilogues (HF: no dropout in between)
                    ffn_out = gemm.ffn(x16, layer.intermediate.dense, layer.output.dense, "gelu", 0.0, training)
                else:
                    inter = layer.intermediate.intermediate_act_fn(layer.intermediate.dense(x16))
                    ffn_out = layer.output.dense(inter)
                x, x16 = add_dropout_layer_norm(x, ffn_out, layer.output.LayerNorm,
                                                layer.output.dropout.p, training, want_bf16=True)
        outs, r0 = [], 0
        for shp, n in zip(shapes, rows):
            outs.append(x[r0:r0 + n].view(shp))
            r0 += n
        return outs

    def forward_pair(self, ids_a, masks_a, ids_b, masks_b, cls_second: bool = False):
        """Both texts of a pre-train pair through ONE walk of the encoder stack (see _fast_forward_multi); without
        the fast path: two plain calls, as the reference makes them (model/openvocab.py:34-40)."""
        if self._fast_ok(ids_a) and self._fast_ok(ids_b):
            # the second text (the scene caption) is only ever read at [CLS] (model/openvocab.py: scene_txt[:, 0]):
            # the variable-length form hands back just those rows, as a (B, 1, D) tensor
            a, b = self._fast_forward_multi([(ids_a, masks_a), (ids_b, masks_b)], cls_only=(1,) if cls_second else ())
            return a, b
        self.last_path = "hf"
        return self.forward(ids_a, masks_a), self.forward(ids_b, masks_b)

    def forward(self, txt_ids, txt_masks, **kwargs):
        if self._fast_ok(txt_ids):
            return self._fast_forward(txt_ids, txt_masks)
        self.last_path = "hf"
        return self.model(txt_ids, txt_masks).last_hidden_state

    # ---- generator forms (the model runs this stack in lock-step with the object encoder's: gemm.drive_pair) ----
    def forward_gen(self, txt_ids, txt_masks):
        if self._fast_ok(txt_ids):
            return (yield from self._fast_forward_multi([(txt_ids, txt_masks)], as_gen=True))[0]
        return self.forward(txt_ids, txt_masks)

    def forward_pair_gen(self, ids_a, masks_a, ids_b, masks_b, cls_second: bool = False):
        if self._fast_ok(ids_a) and self._fast_ok(ids_b):
            a, b = yield from self._fast_forward_multi([(ids_a, masks_a), (ids_b, masks_b)], cls_only=(1,) if cls_second else (),
                                                       as_gen=True)
            return a, b
        return self.forward_pair(ids_a, masks_a, ids_b, masks_b, cls_second)
