"""CLIP text encoder wrapper (reference modules/language/clip.py:9-28); HuggingFace arithmetic,
kept so the registry name resolves.  Needs locally available weights."""
from contextlib import nullcontext

import torch
import torch.nn as nn

from ..build import LANGUAGE_REGISTRY
from ..utils import get_mlp_head


@LANGUAGE_REGISTRY.register()
class CLIPLanguageEncoder(nn.Module):
    def __init__(self, cfg, weights="openai/clip-vit-large-patch14", output_dim=768,
                 freeze_backbone=True, use_projection=False, dropout=0.1):
        super().__init__()
        from transformers import CLIPTextModelWithProjection
        self.context = torch.no_grad if freeze_backbone else nullcontext
        self.model = CLIPTextModelWithProjection.from_pretrained(weights)
        self.use_projection = use_projection
        if use_projection:
            self.projection = get_mlp_head(self.model.config.hidden_size, output_dim, output_dim,
                                           dropout=dropout)

    def forward(self, txt_ids, txt_masks):
        with self.context():
            txt = self.model(txt_ids, txt_masks).last_hidden_state
            txt = torch.nn.functional.normalize(self.model.text_projection(txt), p=2, dim=2)
        return self.projection(txt) if self.use_projection else txt
