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


def set_fast_bert(flag: bool) -> None:
    global _FAST
    _FAST = bool(flag)


def set_fused_embedding(flag: bool) -> None:
    """Word-table gradient through libgps_hip.so (fused_embedding.py) inside the fast path; off = HF module."""
    global _FAST_EMB
    _FAST_EMB = bool(flag)


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

    def _fast_ok(self, txt_ids) -> bool:
        from ..layers.transformers import _bf16_mode
        cfg = self.bert_config
        probe = self.model.embeddings.word_embeddings.weight
        return (_FAST and txt_ids.is_cuda and _bf16_mode(probe) and cfg.hidden_act == "gelu"
                and cfg.hidden_size == cfg.num_attention_heads * 64 and not cfg.is_decoder
                and getattr(cfg, "position_embedding_type", "absolute") == "absolute")

    def _fast_forward(self, txt_ids, txt_masks):
        return self._fast_forward_multi([(txt_ids, txt_masks)])[0]

    def _fast_forward_multi(self, texts):
        """texts = [(ids (B_i, L_i), masks (B_i, L_i)), ...] -> [last hidden state (B_i, L_i, D), ...].
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

    def forward_pair(self, ids_a, masks_a, ids_b, masks_b):
        """Both texts of a pre-train pair through ONE walk of the encoder stack (see _fast_forward_multi); without
        the fast path: two plain calls, as the reference makes them (model/openvocab.py:34-40)."""
        if self._fast_ok(ids_a) and self._fast_ok(ids_b):
            a, b = self._fast_forward_multi([(ids_a, masks_a), (ids_b, masks_b)])
            return a, b
        return self.forward(ids_a, masks_a), self.forward(ids_b, masks_b)

    def forward(self, txt_ids, txt_masks, **kwargs):
        if self._fast_ok(txt_ids):
            return self._fast_forward(txt_ids, txt_masks)
        return self.model(txt_ids, txt_masks).last_hidden_state
