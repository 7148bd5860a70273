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
from ..layers.fused_norm import add_row


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

    def forward(self, txt_embeds, txt_masks, obj_embeds, obj_locs, obj_masks,
                output_attentions=False, output_hidden_states=False, **kwargs):
        txt_len, obj_len = txt_embeds.shape[1], obj_embeds.shape[1]
        joint_pad = torch.cat((txt_masks, obj_masks), dim=1).logical_not()
        type_txt = self.token_type_embeddings.weight[0]
        # the same deterministic embeddings are re-added to both streams every layer (ref :154-164) and the
        # streams are concatenated right after: build the joint (B, T, D) addend once and keep the sequence joint
        # across layers -- the same elementwise sums, one add per layer instead of two adds + cat + split
        # (row-vector additions through add_row: their gradient is a column sum over 3 200 / 5 120 rows, which torch's
        # generic reduction takes 45 - 60 us for)
        obj_extra = add_row(loc_embed(self.loc_layers[0], obj_locs), self.token_type_embeddings.weight[1])
        txt_extra = add_row(obj_extra.new_zeros((txt_embeds.shape[0], txt_len, obj_extra.shape[-1])), type_txt.to(obj_extra.dtype))
        extra = torch.cat((txt_extra, obj_extra), dim=1)
        joint = torch.cat((txt_embeds, obj_embeds), dim=1)
        # layer l reads joint_l + extra: the first sum is explicit, every later one leaves the previous layer's last
        # LayerNorm launch together with its bf16 copy (post_add)
        joint = joint + extra
        n_layers = len(self.unified_encoder)
        for li, layer in enumerate(self.unified_encoder):
            joint, _ = layer(joint, tgt_key_padding_mask=joint_pad, post_add=extra if li + 1 < n_layers else None)
        txt_embeds, obj_embeds = torch.split(joint, [txt_len, obj_len], dim=1)
        return txt_embeds, obj_embeds
