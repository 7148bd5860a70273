"""PointOpenVocabEncoder -- object tokeniser of GPS (reference
modules/vision/pcd_openvocab_encoder.py:16-184):

    PointNet++ per object  ->  dropout  ->  softmax(emb @ text_features^T) (607-way, detached)
    ->  num_layers x { emb += LN(Linear(obj_locs)) ; TransformerSpatialEncoderLayer }

Constructor arguments, buffers (`text_features`), sub-module names (`point_feature_extractor`,
`sem_cls_embed_layer`, `sem_mask_embeddings`, `spatial_encoder`, `loc_layers`) and the three
return values are the reference's.  Reference quirks kept: `freeze=True` only freezes what exists
when the loop runs, i.e. the PointNet++ (ref :54-57); frozen BatchNorm is forced to eval at every
forward (ref :121-129); `obj_sem_cls` are detached softmax probabilities (ref :142)."""
import glob
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..build import VISION_REGISTRY
from ..layers.pointnet import PointNetPP
from ..layers.transformers import TransformerSpatialEncoderLayer
from ..utils import calc_pairwise_locs, layer_repeat
from ..weights import _init_weights_bert
from ..layers.fused_loc import loc_embed


@VISION_REGISTRY.register()
class PointOpenVocabEncoder(nn.Module):
    def __init__(self, cfg, backbone='pointnet++', hidden_size=768, path=None, freeze=False,
                 dim_feedforward=2048, num_attention_heads=12, spatial_dim=5, num_layers=4,
                 dim_loc=6, pairwise_rel_type='center', use_matmul_label=False,
                 mixup_strategy=None, mixup_stage1=None, mixup_stage2=None, lang_type='bert',
                 lang_path=None, attn_type='spatial'):
        super().__init__()
        assert backbone in ['pointnet++']
        self.point_feature_extractor = PointNetPP(
            sa_n_points=[32, 16, None],
            sa_n_samples=[32, 32, None],
            sa_radii=[0.2, 0.4, None],
            sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]],
        )

        # open-vocabulary class head: fixed text embeddings of the 607 ScanNet categories
        vocab = 'bert-base-uncased' if lang_type == 'bert' else 'clip-ViT-B16'
        self.register_buffer(
            "text_features", torch.load(os.path.join(lang_path, f"scannet_607_{vocab}_id.pth")))
        self.point_cls_head = lambda x: x @ self.text_features.t()
        self.dropout = nn.Dropout(0.1)
        self.attn_type = attn_type
        # the reference's point path is fp32-only (its ext rejects anything else); keep PointNet++
        # out of an enclosing bf16 autocast unless explicitly allowed
        self.pointnet_autocast = False

        self.freeze = freeze
        if freeze:
            for p in self.parameters():
                p.requires_grad = False

        # present in checkpoints, unused in forward (ref :60-62, :74)
        self.sem_cls_embed_layer = nn.Sequential(nn.Linear(hidden_size, hidden_size),
                                                 nn.LayerNorm(hidden_size), nn.Dropout(0.1))
        self.use_matmul_label = use_matmul_label
        self.sem_mask_embeddings = nn.Embedding(1, 768)

        if self.attn_type == 'spatial':
            layer = TransformerSpatialEncoderLayer(
                hidden_size, num_attention_heads, dim_feedforward=dim_feedforward, dropout=0.1,
                activation='gelu', spatial_dim=spatial_dim, spatial_multihead=True,
                spatial_attn_fusion='cond')
            self.spatial_encoder = layer_repeat(layer, num_layers)
            self.loc_layers = layer_repeat(
                nn.Sequential(nn.Linear(dim_loc, hidden_size), nn.LayerNorm(hidden_size)), 1)
            self.pairwise_rel_type = pairwise_rel_type
            self.spatial_dim = spatial_dim

        self.apply(_init_weights_bert)
        if path is not None:
            self._load_pretrained(path)

    def _load_pretrained(self, path):
        ckpts = glob.glob(os.path.join(path, '*.bin'))
        if ckpts:
            for ckpt in ckpts:
                self.load_state_dict(torch.load(ckpt, map_location='cpu'), strict=False)
            print("loaded checkpoint files")
        elif path.endswith('.pth'):
            self.load_state_dict(torch.load(path), strict=False)
            print("loaded checkpoint file")

    def freeze_bn(self, m):
        for layer in m.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def forward(self, obj_pcds, obj_locs, obj_masks, obj_sem_masks, obj_labels=None, cur_step=None,
                max_steps=None, **kwargs):
        """obj_pcds (B,O,P,6), obj_locs (B,O,6), obj_masks (B,O) bool ->
        (obj_embeds (B,O,768) after spatial layers, obj_embeds_pre (B,O,768), obj_sem_cls (B,O,607))."""
        from ..layers import gemm
        return gemm.drive(self.forward_gen(obj_pcds, obj_locs, obj_masks, obj_sem_masks, obj_labels, cur_step, max_steps))

    def forward_gen(self, obj_pcds, obj_locs, obj_masks, obj_sem_masks, obj_labels=None, cur_step=None,
                    max_steps=None, **kwargs):
        """`forward` as a generator that yields the GEMM calls of its spatial layers (modules/layers/gemm.py drive /
        drive_pair: the model runs this stack in lock-step with the text encoder's)."""
        if self.freeze:
            self.freeze_bn(self.point_feature_extractor)
        B, O = obj_pcds.shape[:2]
        pcs = obj_pcds.reshape(B * O, *obj_pcds.shape[2:])
        if not self.pointnet_autocast:
            with torch.autocast(device_type=obj_pcds.device.type, enabled=False):
                obj_embeds = self.point_feature_extractor(pcs.float())
        else:
            obj_embeds = self.point_feature_extractor(pcs)
        obj_embeds = self.dropout(obj_embeds.view(B, O, -1))
        if self.freeze:
            obj_embeds = obj_embeds.detach()
        obj_sem_cls = F.softmax(self.point_cls_head(obj_embeds), dim=2).detach()
        obj_embeds_pre = obj_embeds

        if self.attn_type == 'spatial':
            pairwise_locs = calc_pairwise_locs(
                obj_locs[:, :, :3], obj_locs[:, :, 3:], pairwise_rel_type=self.pairwise_rel_type,
                spatial_dist_norm=True, spatial_dim=self.spatial_dim)
            pad = obj_masks.logical_not()
            loc_embeds = loc_embed(self.loc_layers[0], obj_locs)      # re-added every layer (ref :176-178); evaluated once
            obj_embeds = obj_embeds + loc_embeds
            n_layers = len(self.spatial_encoder)
            from ..layers.fused_norm import SharedPostGrad
            share = SharedPostGrad()     # the addend's gradient: one buffer for all layers' launches (layer 0's runs last)
            for li, layer in enumerate(self.spatial_encoder):     # later re-adds ride on the previous layer's last LayerNorm
                obj_embeds, _ = yield from layer.forward_gen(obj_embeds, pairwise_locs, tgt_key_padding_mask=pad,
                                                             post_add=loc_embeds if li + 1 < n_layers else None,
                                                             post_share=(share, li == 0))
        return obj_embeds, obj_embeds_pre, obj_sem_cls
