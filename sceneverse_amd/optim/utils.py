def no_decay_param_group(parameters, lr):
    """Two AdamW groups (weight decay 0.01 / 0.0) split BY PARAMETER NAME, exactly like the
    reference's optim/utils.py:1-18: no decay iff the name contains 'bias', 'LayerNorm.bias' or
    'LayerNorm.weight' (so `norm1.weight` IS decayed -- parameter names are part of the contract)."""
    no_decay = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')
    decay_params, no_decay_params = [], []
    for n, p in parameters:
        if not p.requires_grad:
            continue
        (no_decay_params if any(nd in n for nd in no_decay) else decay_params).append(p)
    return [{'params': decay_params, 'weight_decay': 0.01, 'lr': lr},
            {'params': no_decay_params, 'weight_decay': 0.0, 'lr': lr}]
