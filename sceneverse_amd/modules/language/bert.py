"""BERT text encoder wrapper (reference modules/language/bert.py:7-26).  The arithmetic lives in
HuggingFace `transformers` (out of scope); this wrapper keeps the
registry name, constructor arguments, `self.model` attribute (checkpoint keys `model.*`) and
forward contract.  Extra, for offline machines: `weights=None` (or `random_init=True`) builds
`BertModel(BertConfig(...))` with random weights instead of calling `from_pretrained`."""
import torch.nn as nn

from ..build import LANGUAGE_REGISTRY


@LANGUAGE_REGISTRY.register()
class BERTLanguageEncoder(nn.Module):
    def __init__(self, cfg, weights="bert-base-uncased", hidden_size=768, num_hidden_layers=4,
                 num_attention_heads=12, type_vocab_size=2, random_init=False):
        super().__init__()
        from transformers import BertConfig, BertModel
        self.bert_config = BertConfig(hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                                      num_attention_heads=num_attention_heads,
                                      type_vocab_size=type_vocab_size)
        if weights is None or random_init:
            self.tokenizer = None
            self.model = BertModel(self.bert_config)
        else:
            from transformers import BertTokenizer
            self.tokenizer = BertTokenizer.from_pretrained(weights, do_lower_case=True)
            self.model = BertModel.from_pretrained(weights, config=self.bert_config)

    def forward(self, txt_ids, txt_masks, **kwargs):
        return self.model(txt_ids, txt_masks).last_hidden_state
