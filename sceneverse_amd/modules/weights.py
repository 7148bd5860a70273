import torch.nn as nn


def _init_weights_bert(module, std=0.02):
    """BERT-style init applied with `Module.apply` (reference modules/weights.py:3-20):
    Linear/Embedding weights ~ N(0, std), biases 0, padding row 0, LayerNorm (1, 0)."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=std)
        if isinstance(module, nn.Linear):
            if module.bias is not None:
                module.bias.data.zero_()
        elif module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    elif isinstance(module, nn.LayerNorm):
        module.weight.data.fill_(1.0)
        module.bias.data.zero_()
