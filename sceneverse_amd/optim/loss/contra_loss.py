"""Contrastive losses (reference optim/loss/contra_loss.py:11-98).  The two between-batch losses
all-gather their features across data-parallel ranks WITHOUT autograd (as the reference does), so
under DDP they only train `logit_scale`."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...common.dist_utils import all_gather
from ...modules.layers.fused_norm import l2_normalize
from .loss import LOSS_REGISTRY


_FUSED = True        # tests: False = the torch composition on GPU tensors too


def _clip(a_raw, b_raw, logit_scale_param, gathered):
    """The loss of the two between-batch modules: s = clamp(logit_scale, max=100); rows = `gathered` (already normalised,
    no autograd history: data-parallel runs) or the normalised a_raw / b_raw.  GPU fp32 rows: two launches
    (fused_contra.clip_loss); otherwise the reference's composition."""
    from . import fused_contra as FC
    if gathered is not None:
        a, b = gathered
        if _FUSED and FC.clip_loss_usable(a, b, logit_scale_param):
            return FC.clip_loss(a.detach(), b.detach(), logit_scale_param, normalize=False)
        return _symmetric_clip_loss(a, b, torch.clamp(logit_scale_param, max=100))
    if _FUSED and FC.clip_loss_usable(a_raw, b_raw, logit_scale_param):
        return FC.clip_loss(a_raw, b_raw, logit_scale_param, normalize=True)
    return _symmetric_clip_loss(l2_normalize(a_raw), l2_normalize(b_raw), torch.clamp(logit_scale_param, max=100))


def _symmetric_clip_loss(a, b, logit_scale):
    labels = torch.arange(a.shape[0], device=a.device)
    a2b = logit_scale * a @ b.t()
    b2a = logit_scale * b @ a.t()
    return (F.cross_entropy(a2b, labels) + F.cross_entropy(b2a, labels)) / 2


@LOSS_REGISTRY.register()
class TextObjWithinBatch(nn.Module):
    """Cross-entropy over the objects of each scene between the sentence CLS and every object."""

    def __init__(self, cfg):
        super().__init__()
        self.distributed = cfg.num_gpu > 1
        self.bce = cfg.task in ["ScanQA"]

    def forward(self, data_dict):
        obj_feats = data_dict["intra_obj_embeds"]      # (B,O,D)
        text_feats = data_dict["intra_text_embed"]     # (B,D)
        labels = data_dict["tgt_object_id"]            # (B,1)
        masks = data_dict["obj_masks"]
        if obj_feats.shape[0] != masks.shape[0]:       # per-scene variant: B*L rows
            rep = int(obj_feats.shape[0] / masks.shape[0])
            masks = masks.unsqueeze(1).repeat(1, rep, 1).view(-1, masks.shape[1])
            labels = labels.view(-1, 1)
        if not self.bce and _FUSED:
            from . import fused_contra as FC
            if FC.text_obj_ce_usable(obj_feats, text_feats, labels, masks):
                return FC.text_obj_ce(obj_feats, text_feats, labels, masks)
        obj_feats = l2_normalize(obj_feats)
        text_feats = l2_normalize(text_feats)
        logits = torch.einsum('bod,bd->bo', obj_feats, text_feats)
        labels = labels.squeeze(-1)
        if self.bce:
            return F.binary_cross_entropy_with_logits(logits, labels.float(), reduction="sum",
                                                      weight=masks) / float(labels.shape[0])
        logits.masked_fill_(masks.logical_not(), -float('inf'))
        return F.cross_entropy(logits, labels)


@LOSS_REGISTRY.register()
class TextObjBetweenBatch(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.distributed = cfg.num_gpu > 1
        self.logit_scale = nn.Parameter((torch.ones([]) * np.log(1 / 0.07)).exp())

    _gathered = None   # engine hook, see TextSceneBetweenBatch

    def gather_inputs(self, data_dict):
        obj_feats = data_dict["inter_obj_embeds"]
        labels = data_dict["tgt_object_id"]
        if obj_feats.shape[0] != labels.shape[0]:
            labels = labels.view(-1, 1)
        tgt = obj_feats[torch.arange(labels.size(0)), labels[:, 0], :]
        return [l2_normalize(tgt), l2_normalize(data_dict["inter_text_embed"])]

    def raw_inputs(self, data_dict):
        obj_feats = data_dict["inter_obj_embeds"]
        labels = data_dict["tgt_object_id"]
        if obj_feats.shape[0] != labels.shape[0]:
            labels = labels.view(-1, 1)
        return obj_feats[torch.arange(labels.size(0)), labels[:, 0], :], data_dict["inter_text_embed"]

    def forward(self, data_dict):
        if self.distributed:
            tgt, text_feats = self._gathered if self._gathered is not None else all_gather(self.gather_inputs(data_dict))
            return _clip(None, None, self.logit_scale, (text_feats, tgt))
        tgt, text_feats = self.raw_inputs(data_dict)
        return _clip(text_feats, tgt, self.logit_scale, None)


@LOSS_REGISTRY.register()
class TextSceneBetweenBatch(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.distributed = cfg.num_gpu > 1
        self.logit_scale = nn.Parameter((torch.ones([]) * np.log(1 / 0.07)).exp())

    # Engine hook (sceneverse_amd.engine, HIP-graph data parallelism): when set, `_gathered` holds
    # the already all-gathered [scene, text] features (static buffers the engine fills with an
    # eager RCCL all-gather of `gather_inputs(...)` between two graph replays); like the reference's
    # gather (common/dist_utils.py:131-149) they carry no autograd history.
    _gathered = None

    def gather_inputs(self, data_dict):
        return [l2_normalize(data_dict["scene_embed"]),
                l2_normalize(data_dict["scene_text_embed"])]

    def forward(self, data_dict):
        if self.distributed:
            scene_feats, text_feats = (self._gathered if self._gathered is not None
                                       else all_gather(self.gather_inputs(data_dict)))
            return _clip(None, None, self.logit_scale, (text_feats, scene_feats))
        return _clip(data_dict["scene_text_embed"], data_dict["scene_embed"], self.logit_scale, None)
