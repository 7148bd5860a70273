"""Joint language/object encoders (reference modules/grounding/unified_encoder.py).

    UnifiedSpatialCrossEncoderV2   ref :121-177  the one every shipped config uses: N x
                                   { obj += LN(Linear(locs)) + type_emb[1]; txt += type_emb[0];
                                     post-norm ReLU TransformerEncoderLayer over cat(txt, obj) }
                                   (location and token-type embeddings are re-added EVERY layer)
    UnifiedSpatialCrossEncoderV1   ref :60-118
    EntitySpatialCrossEncoder      ref :12-57

Unlike the reference (ref :157,162: `.cuda()` on freshly built index tensors every layer) the
type embeddings are read straight from the embedding table on whatever device the module is."""
import torch
import torch.nn as nn

from ..build import GROUNDING_REGISTRY
from ..layers.transformers import (TransformerDecoderLayer, TransformerEncoderLayer,
                                   TransformerSpatialDecoderLayer)
from ..utils import calc_pairwise_locs, layer_repeat
from ..weights import _init_weights_bert
from ..layers.fused_loc import loc_embed
from ..layers.fused_norm import add_row, broadcast_row


_COMPACT_ROWS = True         # False: the joint layers run every padded row, as the reference does (A/B, tests)


def set_compact_joint_rows(flag: bool) -> None:
    global _COMPACT_ROWS
    _COMPACT_ROWS = bool(flag)


def _loc_layer(dim_loc, hidden_size):
    return layer_repeat(nn.Sequential(nn.Linear(dim_loc, hidden_size), nn.LayerNorm(hidden_size)), 1)


@GROUNDING_REGISTRY.register()
class EntitySpatialCrossEncoder(nn.Module):
    def __init__(self, cfg, hidden_size=768, num_attention_heads=12, spatial_dim=5, num_layers=4,
                 dim_loc=6, pairwise_rel_type='center'):
        super().__init__()
        layer = TransformerSpatialDecoderLayer(
            hidden_size, num_attention_heads, dim_feedforward=2048, dropout=0.1, activation='gelu',
            spatial_dim=spatial_dim, spatial_multihead=True, spatial_attn_fusion='cond')
        self.layers = layer_repeat(layer, num_layers)
        self.loc_layers = _loc_layer(dim_loc, hidden_size)
        self.pairwise_rel_type = pairwise_rel_type
        self.spatial_dim = spatial_dim
        self.spatial_dist_norm = True
        self.apply(_init_weights_bert)

    def forward(self, txt_embeds, txt_masks, obj_embeds, obj_locs, obj_masks,
                output_attentions=False, output_hidden_states=False, **kwargs):
        pairwise_locs = calc_pairwise_locs(obj_locs[:, :, :3], obj_locs[:, :, 3:],
                                           pairwise_rel_type=self.pairwise_rel_type)
        obj_pad, txt_pad = obj_masks.logical_not(), txt_masks.logical_not()
        out = obj_embeds
        for layer in self.layers:
            out = out + loc_embed(self.loc_layers[0], obj_locs)
            out, _, _ = layer(out, txt_embeds, pairwise_locs, tgt_key_padding_mask=obj_pad,
                              memory_key_padding_mask=txt_pad)
        return txt_embeds, out


@GROUNDING_REGISTRY.register()
class UnifiedSpatialCrossEncoderV1(nn.Module):
    def __init__(self, cfg, hidden_size=768, num_attention_heads=12, spatial_dim=5, num_layers=4,
                 dim_loc=6, pairwise_rel_type='center'):
        super().__init__()
        pc_layer = TransformerSpatialDecoderLayer(
            hidden_size, num_attention_heads, dim_feedforward=2048, dropout=0.1, activation='gelu',
            spatial_dim=spatial_dim, spatial_multihead=True, spatial_attn_fusion='cond')
        lang_layer = TransformerDecoderLayer(hidden_size, num_attention_heads)
        self.pc_encoder = layer_repeat(pc_layer, num_layers)
        self.lang_encoder = layer_repeat(lang_layer, num_layers)
        self.loc_layers = _loc_layer(dim_loc, hidden_size)
        self.pairwise_rel_type = pairwise_rel_type
        self.spatial_dim = spatial_dim
        self.spatial_dist_norm = True
        self.apply(_init_weights_bert)

    def forward(self, txt_embeds, txt_masks, obj_embeds, obj_locs, obj_masks,
                output_attentions=False, output_hidden_states=False, **kwargs):
        pairwise_locs = calc_pairwise_locs(obj_locs[:, :, :3], obj_locs[:, :, 3:],
                                           pairwise_rel_type=self.pairwise_rel_type)
        obj_pad, txt_pad = obj_masks.logical_not(), txt_masks.logical_not()
        for pc_layer, lang_layer in zip(self.pc_encoder, self.lang_encoder):
            obj_embeds = obj_embeds + loc_embed(self.loc_layers[0], obj_locs)
            # both streams read the PRE-update state of the other (ref :100-115)
            obj_next, _, _ = pc_layer(obj_embeds, txt_embeds, pairwise_locs,
                                      tgt_key_padding_mask=obj_pad, memory_key_padding_mask=txt_pad)
            txt_next, _, _ = lang_layer(txt_embeds, obj_embeds, tgt_key_padding_mask=txt_pad,
                                        memory_key_padding_mask=obj_pad)
            obj_embeds, txt_embeds = obj_next, txt_next
        return txt_embeds, obj_embeds


def _rows_call(name, *args):
    from ... import _native
    st = getattr(_native.load(), name)(*args, torch.cuda.current_stream().cuda_stream)
    _native.check(st, name)


class _JointEmbed(torch.autograd.Function):
    """The packed input of the first joint layer from the text rows a (B, La, D), the object rows b (B, Lb, D) and their
    per-layer addends ea, eb (token-type / location embeddings): x = joint + extra, e = extra, x16 = bf16(x), row r = flat row
    perm[r] of the (B, La + Lb) layout that is never materialised, zeros for r >= *n_live (gps_joint_embed_forward: replaces two
    cats, the add, two gathers and the bf16 cast in front of the first projection).  One gradient launch turns dx, the bf16
    gradient that comes back through x16 and de into the four flat-side gradients (gps_joint_embed_backward)."""

    @staticmethod
    def forward(ctx, a, b, ea, eb, perm, inv, valid8, n_live):
        B, La, D = a.shape
        Lb = b.shape[1]
        a, b, ea, eb = (t.float().contiguous() for t in (a, b, ea, eb))
        n = B * (La + Lb)
        x = torch.empty((n, D), dtype=torch.float32, device=a.device)
        e = torch.empty((n, D), dtype=torch.float32, device=a.device)
        x16 = torch.empty((n, D), dtype=torch.bfloat16, device=a.device)
        with torch.cuda.device(a.device):
            _rows_call("gps_joint_embed_forward", B, La, Lb, D, a.data_ptr(), b.data_ptr(), ea.data_ptr(), eb.data_ptr(),
                       perm.data_ptr(), n_live.data_ptr(), x.data_ptr(), e.data_ptr(), x16.data_ptr())
        ctx.save_for_backward(inv, valid8)
        ctx.dims = (B, La, Lb, D)
        return x, e, x16

    @staticmethod
    def backward(ctx, dx, de, dx16):
        inv, valid8 = ctx.saved_tensors
        B, La, Lb, D = ctx.dims
        dev = inv.device
        dx = None if dx is None else dx.float().contiguous()
        de = None if de is None else de.float().contiguous()
        dx16 = None if dx16 is None else dx16.to(torch.bfloat16).contiguous()
        outs = [torch.empty((B, L, D), dtype=torch.float32, device=dev) for L in (La, Lb, La, Lb)]
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(dev):
            _rows_call("gps_joint_embed_backward", B, La, Lb, D, ptr(dx), ptr(dx16), ptr(de), inv.data_ptr(), valid8.data_ptr(),
                       *(t.data_ptr() for t in outs))
        return outs[0], outs[1], outs[2], outs[3], None, None, None, None


class _UnpackJoint(torch.autograd.Function):
    """The packed rows back as the text part (B, La, D) and the object part (B, Lb, D), both contiguous, zeros at invalid
    positions (gps_rows_unpack2); gradient = gps_rows_pack2 of the two parts' gradients."""

    @staticmethod
    def forward(ctx, x, perm, inv, valid8, n_live, B, La, Lb):
        D = x.shape[1]
        x = x.float().contiguous()
        a = torch.empty((B, La, D), dtype=torch.float32, device=x.device)
        b = torch.empty((B, Lb, D), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _rows_call("gps_rows_unpack2", B, La, Lb, D, x.data_ptr(), inv.data_ptr(), valid8.data_ptr(), a.data_ptr(), b.data_ptr())
        ctx.save_for_backward(perm, n_live)
        ctx.dims = (B, La, Lb, D)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        perm, n_live = ctx.saved_tensors
        B, La, Lb, D = ctx.dims
        dev = perm.device
        da = torch.zeros((B, La, D), dtype=torch.float32, device=dev) if da is None else da.float().contiguous()
        db = torch.zeros((B, Lb, D), dtype=torch.float32, device=dev) if db is None else db.float().contiguous()
        dx = torch.empty((B * (La + Lb), D), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _rows_call("gps_rows_pack2", B, La, Lb, D, da.data_ptr(), db.data_ptr(), perm.data_ptr(), n_live.data_ptr(),
                       dx.data_ptr(), None)
        return dx, None, None, None, None, None, None, None


@GROUNDING_REGISTRY.register()
class UnifiedSpatialCrossEncoderV2(nn.Module):
    def __init__(self, cfg, hidden_size=768, dim_feedforward=2048, num_attention_heads=12,
                 num_layers=4, dim_loc=6):
        super().__init__()
        layer = TransformerEncoderLayer(hidden_size, num_attention_heads, dim_feedforward=dim_feedforward)
        self.unified_encoder = layer_repeat(layer, num_layers)
        self.loc_layers = _loc_layer(dim_loc, hidden_size)
        self.token_type_embeddings = nn.Embedding(2, hidden_size)
        self.apply(_init_weights_bert)

    def _compact_ok(self, txt_embeds, obj_embeds) -> bool:
        from ..layers import gemm
        from ..layers.fused_attention import supported as attn_supported
        from ..layers.fused_norm import supported as norm_supported
        from ..layers.transformers import _bf16_mode
        if not (_COMPACT_ROWS and txt_embeds.is_cuda and _bf16_mode(txt_embeds) and txt_embeds.dtype == torch.float32
                and obj_embeds.dtype == torch.float32 and len(self.unified_encoder) > 0):
            return False
        layer = self.unified_encoder[0]
        D, H = layer.self_attn.embed_dim, layer.self_attn.num_heads
        T = txt_embeds.shape[1] + obj_embeds.shape[1]
        probe = txt_embeds.reshape(-1, D)[:1]
        return (not layer.prenorm and layer.self_attn._same and D == H * 64 and gemm.usable(probe, D, D)
                and gemm.activation_name(layer.activation) is not None and layer.linear1.out_features % 8 == 0
                and attn_supported(D, H, T) and norm_supported(probe, probe.to(torch.bfloat16), layer.norm1))

    def _forward_compact(self, txt_embeds, txt_masks, obj_embeds, obj_locs, obj_masks, txt_extra, obj_extra):
        """The four layers over the VALID rows only.  The reference runs every padded text position and every padded
        object slot of the joint (B, T, D) sequence through its layers (ref :147-177); padded rows are masked as attention
        keys and no head or loss reads them as outputs, so the valid rows' results do not depend on them.  Here the joint
        rows are compacted (gps_rows_plan: every scene's valid rows contiguous), the row-wise ops (projection / FFN GEMMs,
        residual LayerNorms) stop at the device-side row count, attention is the variable-length core, and the result goes
        back into the (B, T, D) layout with ZEROS at the padded positions (the reference leaves unspecified values there).
        At the bench workload 60 % of the 8 320 joint rows are valid."""
        from ... import _native
        from ..layers import gemm
        from ..layers.fused_attention import fused_varlen_self_attention
        from ..layers.fused_norm import add_dropout_layer_norm
        B, Lt, D = txt_embeds.shape
        T = Lt + obj_embeds.shape[1]
        n = B * T
        dev = txt_embeds.device
        if txt_masks.dtype == torch.bool and obj_masks.dtype == torch.bool:
            valid = torch.cat((txt_masks, obj_masks), dim=1).reshape(n)                 # one launch
        else:
            valid = torch.cat((txt_masks != 0, obj_masks != 0), dim=1).reshape(n)
        i64 = torch.empty(2 * n, dtype=torch.int64, device=dev)
        i32 = torch.empty(B + 2, dtype=torch.int32, device=dev)
        perm, inv, cu, n_live = i64[:n], i64[n:], i32[:B + 1], i32[B + 1:]
        with torch.cuda.device(dev):
            st = _native.load().gps_rows_plan(B, T, valid.view(torch.uint8).data_ptr(), perm.data_ptr(), inv.data_ptr(),
                                              cu.data_ptr(), n_live.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _native.check(st, "rows_plan")
        # packed rows straight from the two parts (no concatenated (B, T, D) tensor, no gather of one): zeros in the dead rows,
        # whose (undefined) gradients the reverse launch never reads
        valid8 = valid.view(torch.uint8)
        x, extra_c, x16 = _JointEmbed.apply(txt_embeds, obj_embeds, txt_extra, obj_extra, perm, inv, valid8, n_live)
        n_layers = len(self.unified_encoder)
        from ..layers.fused_norm import SharedPostGrad
        share = SharedPostGrad()         # gradient of extra_c: one buffer for the layers' backward launches
        for li, layer in enumerate(self.unified_encoder):
            sa = layer.self_attn
            training = layer.training
            packed = gemm.linear(x16, sa.in_proj_weight, sa.in_proj_bias, rows_dev=n_live)
            ctx = fused_varlen_self_attention(packed, cu, B, T, sa.num_heads, dropout_p=sa.dropout, training=training)
            attn_out = gemm.linear(ctx, sa.out_proj.weight, sa.out_proj.bias, rows_dev=n_live)
            x, x16 = add_dropout_layer_norm(x, attn_out, layer.norm1, layer.dropout1.p, training, want_bf16=True, rows_dev=n_live)
            ffn_out = gemm.ffn(x16, layer.linear1, layer.linear2, gemm.activation_name(layer.activation), layer.dropout.p,
                               training, rows_dev=n_live)
            x, x16 = add_dropout_layer_norm(x, ffn_out, layer.norm2, layer.dropout2.p, training, want_bf16=True, rows_dev=n_live,
                                            post=extra_c if li + 1 < n_layers else None, post_share=(share, li == 0))
        return _UnpackJoint.apply(x, perm, inv, valid8, n_live, B, Lt, T - Lt)      # (text part, object part), contiguous

    def forward(self, txt_embeds, txt_masks, obj_embeds, obj_locs, obj_masks,
                output_attentions=False, output_hidden_states=False, **kwargs):
        txt_len, obj_len = txt_embeds.shape[1], obj_embeds.shape[1]
        type_txt = self.token_type_embeddings.weight[0]
        # the same deterministic embeddings are re-added to both streams every layer (ref :154-164) and the
        # streams are concatenated right after: build the joint (B, T, D) addend once and keep the sequence joint
        # across layers -- the same elementwise sums, one add per layer instead of two adds + cat + split
        # (row-vector additions through add_row: their gradient is a column sum over 3 200 / 5 120 rows, which torch's
        # generic reduction takes 45 - 60 us for)
        obj_extra = add_row(loc_embed(self.loc_layers[0], obj_locs), self.token_type_embeddings.weight[1])
        txt_extra = broadcast_row(type_txt.to(obj_extra.dtype), (txt_embeds.shape[0], txt_len))
        if self._compact_ok(txt_embeds, obj_embeds):
            return self._forward_compact(txt_embeds, txt_masks, obj_embeds, obj_locs, obj_masks, txt_extra, obj_extra)
        joint_pad = torch.cat((txt_masks, obj_masks), dim=1).logical_not()
        extra = torch.cat((txt_extra, obj_extra), dim=1)
        joint = torch.cat((txt_embeds, obj_embeds), dim=1)
        # layer l reads joint_l + extra: the first sum is explicit, every later one leaves the previous layer's last
        # LayerNorm launch together with its bf16 copy (post_add)
        joint = joint + extra
        n_layers = len(self.unified_encoder)
        from ..layers.fused_norm import SharedPostGrad
        share = SharedPostGrad()
        for li, layer in enumerate(self.unified_encoder):
            joint, _ = layer(joint, tgt_key_padding_mask=joint_pad, post_add=extra if li + 1 < n_layers else None,
                             post_share=(share, li == 0))
        txt_embeds, obj_embeds = torch.split(joint, [txt_len, obj_len], dim=1)
        return txt_embeds, obj_embeds
