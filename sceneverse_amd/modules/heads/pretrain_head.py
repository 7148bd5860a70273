"""Masked-LM style pre-training heads (reference modules/heads/pretrain_head.py:8-56)."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..build import HEADS_REGISTRY
from ..utils import get_activation_fn


_FUSED_LM_LOSS = False


@contextlib.contextmanager
def fused_lm_loss(flag: bool = True):
    """Inside this context a TRAINING-mode `BertLMPredictionHead` on the GPU hands the loss a `LazyLMLogits`
    (optim/loss/fused_lm_loss.py) instead of the (B, L, vocab) logits: the engine's train step turns it on,
    everything else (evaluation, tests reading the logits) keeps the reference's tensor."""
    global _FUSED_LM_LOSS
    prev, _FUSED_LM_LOSS = _FUSED_LM_LOSS, bool(flag)
    try:
        yield
    finally:
        _FUSED_LM_LOSS = prev


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, hidden_size, hidden_act='gelu'):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.transform_act_fn = get_activation_fn(hidden_act)
        self.LayerNorm = nn.LayerNorm(hidden_size)

    def forward(self, hidden_states):
        return self.LayerNorm(self.transform_act_fn(self.dense(hidden_states)))


class BertLMPredictionHead(nn.Module):
    """dense -> gelu -> LayerNorm -> bias-free decoder + separate bias parameter."""

    def __init__(self, hidden_size, vocab_size):
        super().__init__()
        self.transform = BertPredictionHeadTransform(hidden_size=hidden_size, hidden_act='gelu')
        self.decoder = nn.Linear(hidden_size, vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab_size))

    def forward(self, hidden_states):
        if _FUSED_LM_LOSS and self.training and hidden_states.is_cuda:
            from ...optim.loss import fused_lm_loss as F_lm
            if F_lm.usable(hidden_states, self.decoder.weight):
                if F_lm.transform_supported(self.transform, hidden_states):
                    # dense -> gelu -> LayerNorm is row-wise: it runs behind the loss's row selection, on the labelled rows
                    return F_lm.LazyLMLogits(hidden_states, self.decoder.weight, self.bias, transform=self.transform)
                return F_lm.LazyLMLogits(self.transform(hidden_states), self.decoder.weight, self.bias)
            return F.linear(self.transform(hidden_states), self.decoder.weight, self.bias)
        # decoder(h) + bias (ref :29) as ONE GEMM with the bias in its epilogue: under bf16 autocast the separate
        # add promoted the (tokens x vocab) logits to fp32 (a 390 MB round trip forward, the same again backward)
        return F.linear(self.transform(hidden_states), self.decoder.weight, self.bias)


@HEADS_REGISTRY.register()
class PretrainHeadV1(nn.Module):
    def __init__(self, cfg, hidden_size=768, vocab_size=30522):
        super().__init__()
        self.lm_pred_head = BertLMPredictionHead(hidden_size, vocab_size)

    def forward(self, txt_embeds, **kwargs):
        return self.lm_pred_head(txt_embeds)


@HEADS_REGISTRY.register()
class OVPretrainHead(nn.Module):
    def __init__(self, cfg, hidden_size=768, vocab_size=30522, obj_vocab_size=607):
        super().__init__()
        self.lm_pred_head = BertLMPredictionHead(hidden_size, vocab_size)
        self.obj_pred_head = BertLMPredictionHead(hidden_size, obj_vocab_size)

    def forward(self, txt_embeds, obj_embeds, **kwargs):
        return (self.lm_pred_head(txt_embeds), self.obj_pred_head(obj_embeds))
