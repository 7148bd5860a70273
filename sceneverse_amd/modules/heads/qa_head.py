"""Question-answering head (reference modules/heads/qa_head.py:8-91): attention-flatten both
streams, fuse, classify.  Not on the pre-train / grounding path; kept so `QAHeadV1` resolves."""
import torch
import torch.nn.functional as F
from torch import nn

from ..build import HEADS_REGISTRY


class FC(nn.Module):
    def __init__(self, in_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.pdrop, self.use_gelu = pdrop, use_gelu
        self.linear = nn.Linear(in_size, out_size)
        if use_gelu:
            self.gelu = nn.GELU()
        if pdrop > 0:
            self.dropout = nn.Dropout(pdrop)

    def forward(self, x):
        x = self.linear(x)
        if self.use_gelu:
            x = self.gelu(x)
        return self.dropout(x) if self.pdrop > 0 else x


class MLP(nn.Module):
    def __init__(self, in_size, mid_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.fc = FC(in_size, mid_size, pdrop=pdrop, use_gelu=use_gelu)
        self.linear = nn.Linear(mid_size, out_size)

    def forward(self, x):
        return self.linear(self.fc(x))


class AttFlat(nn.Module):
    """Learned soft pooling over tokens: `flat_glimpses` attention maps -> concat -> linear."""

    def __init__(self, hidden_size, flat_mlp_size=512, flat_glimpses=1, flat_out_size=1024, pdrop=0.1):
        super().__init__()
        self.mlp = MLP(hidden_size, flat_mlp_size, flat_glimpses, pdrop=pdrop, use_gelu=True)
        self.flat_glimpses = flat_glimpses
        self.linear_merge = nn.Linear(hidden_size * flat_glimpses, flat_out_size)

    def forward(self, x, x_mask):
        att = self.mlp(x)
        if x_mask is not None:
            att = att.masked_fill(x_mask.unsqueeze(2), -1e9)
        att = F.softmax(att, dim=1)                      # (B, T, G)
        pooled = torch.einsum('btg,btd->bgd', att, x)    # glimpse-major, as the reference's cat
        return self.linear_merge(pooled.flatten(1))


@HEADS_REGISTRY.register()
class QAHeadV1(nn.Module):
    def __init__(self, cfg, hidden_size=768, mlp_size=256, glimpse=1, flat_out_size=512, num_answers=8864):
        super().__init__()
        self.attflat_visual = AttFlat(hidden_size, mlp_size, glimpse, flat_out_size, 0.1)
        self.attflat_lang = AttFlat(hidden_size, mlp_size, glimpse, flat_out_size, 0.1)
        self.answer_cls = nn.Sequential(nn.Linear(flat_out_size, hidden_size), nn.GELU(),
                                        nn.Dropout(0.3), nn.Linear(hidden_size, num_answers))
        self.fusion_norm = nn.LayerNorm(flat_out_size)

    def forward(self, obj_embeds, obj_masks, txt_embeds, txt_masks, **kwargs):
        object_feat = self.attflat_visual(obj_embeds, obj_masks.logical_not())
        lang_feat = self.attflat_lang(txt_embeds, txt_masks.logical_not())
        return self.answer_cls(self.fusion_norm(lang_feat + object_feat))
