"""Transformer layers of the GPS hot path, API- and checkpoint-compatible with the reference's
modules/layers/transformers.py (class names, constructor arguments, forward signatures, return
tuples, parameter names), re-built around one attention core:

    attention_core()                  softmax(QK^T/sqrt(d) + bias) V, the op behind every layer
    MultiHeadAttentionSpatial         ref :157-239  language-conditioned pairwise-spatial attention
    MultiheadSelfAttention            stands in for torch.nn.MultiheadAttention as the reference
                                      uses it (batch_first, key_padding_mask, attn dropout), with
                                      nn.MultiheadAttention's parameter names
    TransformerEncoderLayer           ref :115-154
    TransformerSpatialEncoderLayer    ref :285-316
    TransformerDecoderLayer           ref :66-112
    TransformerSpatialDecoderLayer    ref :242-282
    CrossAttentionLayer               ref :12-63

Backends of the attention core (see `set_attention_backend`):
    "hip"    fused gfx950 kernels from libgps_hip.so (GPU tensors; the default on a GPU): self- and cross-attention,
             bf16 (<= 512 tokens) and fp32 operands (<= 256 tokens, fp32 MFMA),
    "torch"  plain PyTorch ops; chosen automatically only for CPU tensors (tests / gloo runs), when attention
             probabilities are requested (`need_weights=True`), for an explicit `attn_mask`, for the spatial
             fusions other than 'cond', or for fp32 rows above 256 tokens.

Deviations from the reference, on purpose (DESIGN.md "reference quirks"):
  * no `assert torch.sum(torch.isnan(fused_attn) == 0)` host sync (ref :234);
  * attention probabilities are only materialised when `need_weights` is set on the module
    (every caller in the reference discards them).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from ..utils import get_activation_fn

_BACKEND = "auto"  # "auto" | "hip" | "torch"


def set_attention_backend(name: str) -> None:
    global _BACKEND
    if name not in ("auto", "hip", "torch"):
        raise ValueError(name)
    _BACKEND = name


def get_attention_backend() -> str:
    return _BACKEND


def _bf16_mode(x: Tensor) -> bool:
    """True when this op would run in bf16: a bf16 tensor, or fp32 under torch.autocast(bf16)."""
    if x.dtype == torch.bfloat16:
        return True
    return x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled() and \
        torch.get_autocast_dtype('cuda') == torch.bfloat16


def _use_hip(x: Tensor, d_model: int, n_head: int, kv_len: Optional[int] = None) -> bool:
    """The fused gfx950 attention core (libgps_hip.so) serves GPU self- and cross-attention: bf16 (a bf16 tensor, or
    fp32 under bf16 autocast) up to 512 tokens on the bf16 MFMA, plain fp32 up to 256 tokens on the fp32 MFMA."""
    if _BACKEND == "torch" or not x.is_cuda:
        return False
    from . import fused_attention
    dtype = torch.bfloat16 if _bf16_mode(x) else x.dtype
    ok = x.dim() == 3 and dtype in (torch.bfloat16, torch.float32) and \
        fused_attention.supported(d_model, n_head, x.shape[1], dtype, kv_len)
    if _BACKEND == "hip" and not ok:
        raise RuntimeError("attention backend 'hip' requested for an unsupported call "
                           f"(shape {tuple(x.shape)}, dtype {x.dtype}, d_model {d_model}, heads {n_head}, keys {kv_len})")
    return ok


def _proj_ctx(x: Tensor):
    """Projections around the fused core when they go through F.linear: bf16 autocast in bf16 mode, plain fp32 else."""
    import contextlib
    if _bf16_mode(x):
        return torch.autocast(device_type="cuda", dtype=torch.bfloat16)
    return contextlib.nullcontext()


def _split_heads(x: Tensor, n_head: int) -> Tensor:
    """(B, L, H*dh) -> (B, H, L, dh)"""
    b, l, d = x.shape
    return x.view(b, l, n_head, d // n_head).transpose(1, 2)


def _merge_heads(x: Tensor) -> Tensor:
    """(B, H, L, dh) -> (B, L, H*dh)"""
    b, h, l, dh = x.shape
    return x.transpose(1, 2).reshape(b, l, h * dh)


def attention_core(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor] = None,
                   key_padding_mask: Optional[Tensor] = None, dropout_p: float = 0.0,
                   training: bool = False, need_weights: bool = False):
    """q (B,H,L,dh), k/v (B,H,T,dh), bias (B,H,L,T) additive or None, key_padding_mask (B,T)
    True = ignore.  Returns (out (B,H,L,dh), probs (B,H,L,T) or None).  fp32 softmax."""
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.size(-1))
    scores = scores.float()
    if bias is not None:
        scores = scores + bias.float()
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    p = probs
    if dropout_p > 0.0 and training:
        p = F.dropout(p, dropout_p, training=True)
    out = torch.matmul(p.to(v.dtype), v)
    return out, (probs if need_weights else None)


class MultiHeadAttentionSpatial(nn.Module):
    """Self-attention over objects whose logits are modulated by pairwise geometry.

    fusion 'cond' (the one GPS uses): per query token l and head h a 6-vector
    (bias, w_1..w_5) = lang_cond_fc(x_l); loc[h,b,l,t] = sigmoid(w . pairwise[b,l,t,:] + bias);
    probs = softmax(log(clamp(loc, 1e-6)) + q k^T / sqrt(d_h)) with padded keys removed.
    Other fusions ('mul', 'bias', 'add', 'ctx') follow ref :199-231.
    The `dropout` argument is accepted and unused, as in the reference.
    """

    def __init__(self, d_model, n_head, dropout=0.1, spatial_multihead=True, spatial_dim=5,
                 spatial_attn_fusion='mul'):
        super().__init__()
        assert d_model % n_head == 0, 'd_model: %d, n_head: %d' % (d_model, n_head)
        self.n_head = n_head
        self.d_model = d_model
        self.d_per_head = d_model // n_head
        self.spatial_multihead = spatial_multihead
        self.spatial_dim = spatial_dim
        self.spatial_attn_fusion = spatial_attn_fusion
        self.need_weights = False

        self.w_qs = nn.Linear(d_model, d_model)
        self.w_ks = nn.Linear(d_model, d_model)
        self.w_vs = nn.Linear(d_model, d_model)
        self.fc = nn.Linear(d_model, d_model)

        self.spatial_n_head = n_head if spatial_multihead else 1
        if spatial_attn_fusion in ('mul', 'bias', 'add'):
            self.pairwise_loc_fc = nn.Linear(spatial_dim, self.spatial_n_head)
        elif spatial_attn_fusion == 'ctx':
            self.pairwise_loc_fc = nn.Linear(spatial_dim, d_model)
        elif spatial_attn_fusion == 'cond':
            self.lang_cond_fc = nn.Linear(d_model, self.spatial_n_head * (spatial_dim + 1))
        else:
            raise NotImplementedError('unsupported spatial_attn_fusion %s' % spatial_attn_fusion)

    # ---- geometry term, (B,H,L,T), already in the form that is ADDED to the scaled logits ----
    def _spatial_logits(self, x_in: Tensor, q: Tensor, pairwise_locs: Tensor,
                        key_padding_mask: Optional[Tensor]):
        fusion = self.spatial_attn_fusion
        H = self.n_head
        if fusion == 'cond':
            b, l, _ = x_in.shape
            sw = self.lang_cond_fc(x_in).float().view(b, l, self.spatial_n_head, self.spatial_dim + 1)
            sw = sw.permute(0, 2, 1, 3)                                     # (B,h,L,1+D)
            if self.spatial_n_head == 1:
                sw = sw.expand(-1, H, -1, -1)
            loc = torch.einsum('bhld,bltd->bhlt', sw[..., 1:], pairwise_locs.float()) + sw[..., :1]
            loc = torch.sigmoid(loc)
        elif fusion in ('mul', 'bias', 'add'):
            loc = self.pairwise_loc_fc(pairwise_locs).float().permute(0, 3, 1, 2)  # (B,h,L,T)
            if fusion == 'mul':
                loc = F.relu(loc)
            if not self.spatial_multihead:
                loc = loc.expand(-1, H, -1, -1)
        else:  # 'ctx'
            b, l, t, _ = pairwise_locs.shape
            ctx = self.pairwise_loc_fc(pairwise_locs).view(b, l, t, H, self.d_per_head)
            loc = torch.einsum('bhlk,blthk->bhlt', q.float(), ctx.float()) / math.sqrt(self.d_per_head)
        if fusion in ('mul', 'cond'):
            if key_padding_mask is not None:
                loc = loc.masked_fill(key_padding_mask[:, None, None, :], 0)
            return torch.log(torch.clamp(loc, min=1e-6))
        return loc

    def _forward_fused_gen(self, x, pairwise_locs, key_padding_mask):
        """fusion 'cond', self-attention, bf16 on the GPU: ONE projection GEMM producing
        [q | k | v | per-head (bias, w_1..w_5)] (libgps_hip.so's MFMA GEMM on a persistent packed bf16
        copy of the four weights) and one fused attention launch.  A generator: the two GEMMs are YIELDED
        (gemm.LinearOp) so that a caller may pair them with another stack's (gemm.drive / gemm.drive_pair)."""
        from . import gemm
        from .fused_attention import fused_self_attention
        if _bf16_mode(x) and gemm.usable(x, self.d_model, self.d_model):
            packed = yield gemm.LinearOp.of(_gemm_input(x), [self.w_qs, self.w_ks, self.w_vs, self.lang_cond_fc])
            out = fused_self_attention(packed, self.n_head, pairwise_locs, key_padding_mask)
            return (yield gemm.LinearOp.of(out, [self.fc])), None
        # fp32 operands (the fp32 master path: fp32 MFMA core) or bf16 without the native GEMMs (A/B runs)
        w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight, self.lang_cond_fc.weight], 0)
        bias = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias, self.lang_cond_fc.bias], 0)
        with _proj_ctx(x):
            packed = F.linear(x, w, bias)
        out = fused_self_attention(packed, self.n_head, pairwise_locs, key_padding_mask)
        with _proj_ctx(x):
            return self.fc(out), None

    def forward(self, q, k, v, pairwise_locs, key_padding_mask=None, txt_embeds=None):
        from . import gemm
        return gemm.drive(self.forward_gen(q, k, v, pairwise_locs, key_padding_mask, txt_embeds))

    def forward_gen(self, q, k, v, pairwise_locs, key_padding_mask=None, txt_embeds=None):
        if (self.spatial_attn_fusion == 'cond' and k is q and v is q and not self.need_weights
                and self.spatial_n_head == self.n_head and self.spatial_dim == 5
                and _use_hip(q, self.d_model, self.n_head)):
            return (yield from self._forward_fused_gen(q, pairwise_locs, key_padding_mask))
        x_in = q
        qh = _split_heads(self.w_qs(q), self.n_head)
        kh = _split_heads(self.w_ks(k), self.n_head)
        vh = _split_heads(self.w_vs(v), self.n_head)
        if self.spatial_attn_fusion == 'add':
            # average of two softmaxes (ref :226-227)
            scores = torch.matmul(qh, kh.transpose(-1, -2)).float() / math.sqrt(self.d_per_head)
            loc = self._spatial_logits(x_in, qh, pairwise_locs, key_padding_mask)
            if key_padding_mask is not None:
                m = key_padding_mask[:, None, None, :]
                scores = scores.masked_fill(m, float('-inf'))
                loc = loc.masked_fill(m, float('-inf'))
            probs = (torch.softmax(scores, 3) + torch.softmax(loc, 3)) / 2
            out = torch.matmul(probs.to(vh.dtype), vh)
        else:
            bias = self._spatial_logits(x_in, qh, pairwise_locs, key_padding_mask)
            out, probs = attention_core(qh, kh, vh, bias=bias, key_padding_mask=key_padding_mask,
                                        need_weights=self.need_weights)
        out = self.fc(_merge_heads(out))
        if self.need_weights and probs is not None:
            probs = probs.transpose(0, 1)  # reference layout (head, B, L, T)
        else:
            probs = None
        return out, probs


class MultiheadSelfAttention(nn.Module):
    """torch.nn.MultiheadAttention as the reference instantiates it (batch_first=True, optional
    kdim/vdim, dropout on the attention probabilities), with the same parameter names
    (`in_proj_weight`, `in_proj_bias`, `out_proj.*`, or `q/k/v_proj_weight` when kdim/vdim differ)
    so reference checkpoints load.  Returns (out, head-averaged probs or None)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, batch_first=True, kdim=None, vdim=None):
        super().__init__()
        if not batch_first:
            raise NotImplementedError("the GPS path only uses batch_first=True")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.kdim = embed_dim if kdim is None else kdim
        self.vdim = embed_dim if vdim is None else vdim
        self._same = self.kdim == embed_dim and self.vdim == embed_dim
        self.need_weights = False
        if self._same:
            self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
            self.register_parameter('q_proj_weight', None)
            self.register_parameter('k_proj_weight', None)
            self.register_parameter('v_proj_weight', None)
        else:
            self.q_proj_weight = nn.Parameter(torch.empty(embed_dim, embed_dim))
            self.k_proj_weight = nn.Parameter(torch.empty(embed_dim, self.kdim))
            self.v_proj_weight = nn.Parameter(torch.empty(embed_dim, self.vdim))
            self.register_parameter('in_proj_weight', None)
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self._reset_parameters()

    def _reset_parameters(self):
        for w in (self.in_proj_weight, self.q_proj_weight, self.k_proj_weight, self.v_proj_weight):
            if w is not None:
                nn.init.xavier_uniform_(w)
        nn.init.constant_(self.in_proj_bias, 0.)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None, need_weights=None):
        want = self.need_weights if need_weights is None else need_weights
        if (self._same and key is query and value is query and attn_mask is None and not want
                and _use_hip(query, self.embed_dim, self.num_heads)):
            from . import gemm
            from .fused_attention import fused_self_attention
            native = _bf16_mode(query) and gemm.usable(query, self.embed_dim, self.embed_dim)
            if native:
                packed = gemm.linear(_gemm_input(query), self.in_proj_weight, self.in_proj_bias)
            else:
                with _proj_ctx(query):
                    packed = F.linear(query, self.in_proj_weight, self.in_proj_bias)
            out = fused_self_attention(packed, self.num_heads, None, key_padding_mask,
                                       dropout_p=self.dropout, training=self.training)
            if native:
                return gemm.linear(out, self.out_proj.weight, self.out_proj.bias), None
            with _proj_ctx(query):
                return self.out_proj(out), None
        E = self.embed_dim
        if (key is not query and attn_mask is None and not want and key.dim() == 3 and key.shape[1] == value.shape[1]
                and _use_hip(query, self.embed_dim, self.num_heads, kv_len=key.shape[1])):
            # cross-attention (decoder / cross layers, ref :12-112, 242-282) on the fused core: q from `query`,
            # [k | v] from `key` / `value` (one GEMM when they are the same tensor, as every reference caller passes)
            from .fused_attention import fused_cross_attention
            bq_, bkv = self.in_proj_bias[:E], self.in_proj_bias[E:]
            with _proj_ctx(query):
                if self._same:
                    q = F.linear(query, self.in_proj_weight[:E], bq_)
                    if key is value:
                        kv = F.linear(key, self.in_proj_weight[E:], bkv)
                    else:
                        kv = torch.cat([F.linear(key, self.in_proj_weight[E:2 * E], bkv[:E]),
                                        F.linear(value, self.in_proj_weight[2 * E:], bkv[E:])], dim=-1)
                else:
                    q = F.linear(query, self.q_proj_weight, bq_)
                    kv = torch.cat([F.linear(key, self.k_proj_weight, bkv[:E]),
                                    F.linear(value, self.v_proj_weight, bkv[E:])], dim=-1)
            if kv.dtype != q.dtype:
                kv = kv.to(q.dtype)
            out = fused_cross_attention(q, kv, self.num_heads, key_padding_mask, dropout_p=self.dropout,
                                        training=self.training)
            with _proj_ctx(query):
                return self.out_proj(out), None
        bq, bk, bv = self.in_proj_bias[:E], self.in_proj_bias[E:2 * E], self.in_proj_bias[2 * E:]
        if self._same:
            if key is query and value is query:
                q, k, v = F.linear(query, self.in_proj_weight, self.in_proj_bias).chunk(3, dim=-1)
            else:
                wq, wk, wv = self.in_proj_weight[:E], self.in_proj_weight[E:2 * E], self.in_proj_weight[2 * E:]
                q, k, v = F.linear(query, wq, bq), F.linear(key, wk, bk), F.linear(value, wv, bv)
        else:
            q = F.linear(query, self.q_proj_weight, bq)
            k = F.linear(key, self.k_proj_weight, bk)
            v = F.linear(value, self.v_proj_weight, bv)
        H = self.num_heads
        bias = None
        if attn_mask is not None:
            bias = attn_mask
            if bias.dtype == torch.bool:
                bias = torch.zeros_like(bias, dtype=torch.float32).masked_fill(bias, float('-inf'))
            while bias.dim() < 4:
                bias = bias.unsqueeze(0)
        want = self.need_weights if need_weights is None else need_weights
        out, probs = attention_core(_split_heads(q, H), _split_heads(k, H), _split_heads(v, H),
                                    bias=bias, key_padding_mask=key_padding_mask,
                                    dropout_p=self.dropout, training=self.training,
                                    need_weights=want)
        out = self.out_proj(_merge_heads(out))
        return out, (probs.mean(dim=1) if probs is not None else None)


def _ffn_gen(layer, x):
    """linear2(dropout(activation(linear1(x)))) -- on the GPU in bf16: two MFMA GEMMs with the bias, activation
    and dropout in their epilogues (libgps_hip.so), backward likewise (modules/layers/gemm.py).  Generator: the native
    form is yielded as one gemm.FFNOp."""
    from . import gemm
    if x.is_cuda:
        act = gemm.activation_name(layer.activation)
        if act is not None and gemm.usable(x, layer.linear1.in_features, layer.linear1.out_features) \
                and layer.linear2.out_features % 8 == 0:
            return (yield gemm.FFNOp(x, layer.linear1, layer.linear2, act, layer.dropout.p, layer.training))
    return layer.linear2(layer.dropout(layer.activation(layer.linear1(x))))


def _ffn(layer, x):
    from . import gemm
    return gemm.drive(_ffn_gen(layer, x))


def _res_norm(norm: nn.LayerNorm, x: Tensor, h: Tensor, drop: nn.Dropout, want_bf16: bool = False, post=None, post_share=None):
    """norm(x + drop(h)) [+ post]: one fused launch on the GPU (fused_norm), the plain ops elsewhere.
    want_bf16 (only meaningful under bf16 autocast): -> (y, bf16 copy of y for the next GEMM)."""
    from .fused_norm import add_dropout_layer_norm
    return add_dropout_layer_norm(x, h, norm, drop.p, drop.training, want_bf16=want_bf16, post=post, post_share=post_share)


# Folding the next layer's `x + embedding` into this layer's last LayerNorm launch (fused_norm `post`): the sum and its
# bf16 copy leave the LayerNorm launch, the explicit add, the cast in front of the next projection GEMM and the gradient
# accumulation of the addend's four uses go away.  [r3] measured +0.7 % on the one-graph step but kept off: with it the
# split-graph data-parallel step returned wrong text-encoder gradients.  [r4] root-caused -- not this code: stale
# AccumulateGrad nodes forked the captured bottom-backward graph and ROCm ran its nodes out of order; ANY change of the
# allocation pattern moved the damage (sceneverse_amd/engine.py `_work_stream`, tests/test_gpu_graph_chain.py,
# DESIGN.md section 9).  On by default since.
_FUSE_POST_ADD = True


def set_fuse_post_add(flag: bool) -> None:
    global _FUSE_POST_ADD
    _FUSE_POST_ADD = bool(flag)


def _gemm_input(x: Tensor) -> Tensor:
    """The tensor a projection GEMM should read for `x`: the bf16 copy the producing fused LayerNorm wrote next to it
    (`_layer_output`), so that neither a cast nor a separate gradient accumulation is launched; else x itself."""
    alt = getattr(x, "_gps_bf16", None)
    return alt if (alt is not None and alt.shape == x.shape) else x


def _layer_output_gen(layer, tgt: Tensor, ffn_in: Tensor, post_add, post_share=None):
    """Last step of a post-norm layer: norm2(tgt + dropout2(ffn)) [+ post_add].  With `post_add` (the addend the next
    layer would apply to its input: the re-added location / type embeddings) on the bf16 GPU path, the sum and its bf16
    copy leave the same launch; the copy travels as an attribute of the fp32 result for the next layer's `_gemm_input`."""
    h = yield from _ffn_gen(layer, ffn_in)
    if post_add is None:
        return _res_norm(layer.norm2, tgt, h, layer.dropout2)
    if _FUSE_POST_ADD and _bf16_mode(tgt) and tgt.dtype == torch.float32 and tgt.is_cuda:
        # post_share: (fused_norm.SharedPostGrad, final) when the caller passes the SAME addend to several layers
        y, y16 = _res_norm(layer.norm2, tgt, h, layer.dropout2, want_bf16=True, post=post_add, post_share=post_share)
        if y16 is not y:
            y._gps_bf16 = y16
        return y
    return _res_norm(layer.norm2, tgt, h, layer.dropout2) + post_add


def _layer_output(layer, tgt: Tensor, ffn_in: Tensor, post_add, post_share=None):
    from . import gemm
    return gemm.drive(_layer_output_gen(layer, tgt, ffn_in, post_add, post_share))


class TransformerEncoderLayer(nn.Module):
    """Self-attention + FFN, post-norm unless `prenorm` (ref :115-154).  forward -> (x, attn)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, batch_first=True, dropout=0.1,
                 activation="relu", prenorm=False):
        super().__init__()
        self.self_attn = MultiheadSelfAttention(d_model, nhead, dropout=dropout, batch_first=batch_first)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = get_activation_fn(activation)
        self.prenorm = prenorm

    def forward(self, tgt, tgt_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None, post_add: Optional[Tensor] = None, post_share=None):
        """post_add: optional tensor of tgt's shape added to the layer's OUTPUT (callers that re-add an embedding to
        every layer's input pass it to the previous layer instead: same sums, one launch less per layer)."""
        h = self.norm1(tgt) if self.prenorm else tgt
        h, attn = self.self_attn(query=h, key=h, value=h, attn_mask=tgt_mask,
                                 key_padding_mask=tgt_key_padding_mask)
        if not self.prenorm:
            b16 = _bf16_mode(tgt) and tgt.dtype == torch.float32
            tgt, ffn_in = _res_norm(self.norm1, tgt, h, self.dropout1, want_bf16=True) if b16 else \
                (lambda t: (t, t))(_res_norm(self.norm1, tgt, h, self.dropout1))
            return _layer_output(self, tgt, ffn_in, post_add, post_share), attn
        tgt = tgt + self.dropout1(h)
        # ref :147-153: the pre-norm variant normalises the residual stream itself before the FFN
        tgt = self.norm2(tgt)
        tgt = tgt + self.dropout2(_ffn(self, tgt))
        return (tgt if post_add is None else tgt + post_add), attn


class TransformerSpatialEncoderLayer(TransformerEncoderLayer):
    """Post-norm encoder layer whose attention is MultiHeadAttentionSpatial (ref :285-316)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 spatial_multihead=True, spatial_dim=5, spatial_attn_fusion='mul'):
        super().__init__(d_model, nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                         activation=activation)
        del self.self_attn
        self.self_attn = MultiHeadAttentionSpatial(
            d_model, nhead, dropout=dropout, spatial_multihead=spatial_multihead,
            spatial_dim=spatial_dim, spatial_attn_fusion=spatial_attn_fusion)

    def forward(self, tgt, tgt_pairwise_locs, tgt_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None, post_add: Optional[Tensor] = None, post_share=None):
        from . import gemm
        return gemm.drive(self.forward_gen(tgt, tgt_pairwise_locs, tgt_mask, tgt_key_padding_mask, post_add, post_share))

    def forward_gen(self, tgt, tgt_pairwise_locs, tgt_mask: Optional[Tensor] = None,
                    tgt_key_padding_mask: Optional[Tensor] = None, post_add: Optional[Tensor] = None, post_share=None):
        """The layer as a generator of its GEMM calls (gemm.drive runs it alone, gemm.drive_pair beside another stack)."""
        h, attn = yield from self.self_attn.forward_gen(tgt, tgt, tgt, tgt_pairwise_locs,
                                                        key_padding_mask=tgt_key_padding_mask)
        b16 = _bf16_mode(tgt) and tgt.dtype == torch.float32
        tgt, ffn_in = _res_norm(self.norm1, tgt, h, self.dropout1, want_bf16=True) if b16 else \
            (lambda t: (t, t))(_res_norm(self.norm1, tgt, h, self.dropout1))
        return (yield from _layer_output_gen(self, tgt, ffn_in, post_add, post_share)), attn


class TransformerDecoderLayer(nn.Module):
    """Pre-norm self-attention, cross-attention to `memory`, FFN (ref :66-112).
    forward -> (x, self_attn, cross_attn)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadSelfAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.multihead_attn = MultiheadSelfAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = get_activation_fn(activation)

    def _self_block(self, h, pairwise_locs, tgt_mask, tgt_key_padding_mask):
        return self.self_attn(query=h, key=h, value=h, attn_mask=tgt_mask,
                              key_padding_mask=tgt_key_padding_mask)

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, _pairwise_locs=None):
        h, self_attn = self._self_block(self.norm1(tgt), _pairwise_locs, tgt_mask, tgt_key_padding_mask)
        tgt = tgt + self.dropout1(h)
        h, cross_attn = self.multihead_attn(query=self.norm2(tgt), key=memory, value=memory,
                                            attn_mask=memory_mask,
                                            key_padding_mask=memory_key_padding_mask)
        tgt = tgt + self.dropout2(h)
        tgt = tgt + self.dropout3(_ffn(self, self.norm3(tgt)))
        return tgt, self_attn, cross_attn


class TransformerSpatialDecoderLayer(TransformerDecoderLayer):
    """Decoder layer with spatial self-attention (ref :242-282)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 spatial_multihead=True, spatial_dim=5, spatial_attn_fusion='mul'):
        super().__init__(d_model, nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                         activation=activation)
        del self.self_attn
        self.self_attn = MultiHeadAttentionSpatial(
            d_model, nhead, dropout=dropout, spatial_multihead=spatial_multihead,
            spatial_dim=spatial_dim, spatial_attn_fusion=spatial_attn_fusion)

    def _self_block(self, h, pairwise_locs, tgt_mask, tgt_key_padding_mask):
        return self.self_attn(h, h, h, pairwise_locs, key_padding_mask=tgt_key_padding_mask)

    def forward(self, tgt, memory, tgt_pairwise_locs: Optional[Tensor] = None,
                tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None):
        return super().forward(tgt, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                               tgt_key_padding_mask=tgt_key_padding_mask,
                               memory_key_padding_mask=memory_key_padding_mask,
                               _pairwise_locs=tgt_pairwise_locs)


class CrossAttentionLayer(nn.Module):
    """Cross-attention + FFN block (ref :12-63).  forward -> (x, cross_attn)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 k_dim=None, v_dim=None, prenorm=True):
        super().__init__()
        self.prenorm = prenorm
        self.multihead_attn = MultiheadSelfAttention(
            d_model, nhead, dropout=dropout, batch_first=True,
            kdim=d_model if k_dim is None else k_dim, vdim=d_model if v_dim is None else v_dim)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = get_activation_fn(activation)

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None):
        h = self.norm1(tgt) if self.prenorm else tgt
        h, cross_attn = self.multihead_attn(query=h, key=memory, value=memory,
                                            attn_mask=memory_mask,
                                            key_padding_mask=memory_key_padding_mask)
        tgt = tgt + self.dropout2(h)
        if not self.prenorm:
            tgt = self.norm1(tgt)
        # ref :56-58: in post-norm mode the FFN is fed the raw attention output, not the
        # normalised residual stream -- reproduced as is
        h = self.norm3(tgt) if self.prenorm else h
        tgt = tgt + self.dropout3(_ffn(self, h))
        if not self.prenorm:
            tgt = self.norm3(tgt)
        return tgt, cross_attn
