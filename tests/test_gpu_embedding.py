"""gps_embedding_grad (sceneverse_amd/modules/language/fused_embedding.py) against torch's own embedding
backward and an fp64 reference: same dense table gradient (fp32 summation-order noise only), bit-identical
between calls (deterministic), padding row and unreferenced rows zero; and the whole BERT embedding block
against HF's BertEmbeddings (values equal, gradients close)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneverse_amd.modules.language import fused_embedding as FE  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_grad(ids, dy, num_rows, padding_idx):
    out = torch.zeros(num_rows, dy.shape[-1], dtype=torch.float64, device=dy.device)
    flat, rows = ids.reshape(-1), dy.reshape(-1, dy.shape[-1]).double()
    keep = (flat != padding_idx) if padding_idx >= 0 else torch.ones_like(flat, dtype=torch.bool)
    out.index_add_(0, flat[keep], rows[keep])
    return out


@pytest.mark.parametrize("shape,num_rows,d,padding_idx,kind", [
    ((64, 300), 30522, 768, 0, "bert"),          # the scene-caption pass: CLS/SEP x 64, pads, random words
    ((64, 50), 30522, 768, 0, "bert"),
    ((7, 13), 50, 64, -1, "dense"),              # few rows: every id many times
    ((3000,), 4, 256, -1, "dense"),              # long duplicate runs (750 per row)
    ((33, 17), 1000, 1028, 5, "random"),         # d not a multiple of 256, padding row in the middle
    ((1, 1), 10, 4, -1, "random"),
])
def test_embedding_grad_matches_torch_and_fp64(shape, num_rows, d, padding_idx, kind):
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + num_rows)
    if kind == "bert":
        B, L = shape
        ids = torch.randint(1000, num_rows, shape, generator=g)
        lens = torch.randint(6, L + 1, (B,), generator=g)
        ids[:, 0] = 101
        for b in range(B):
            ids[b, lens[b] - 1] = 102
            ids[b, lens[b]:] = 0
    else:
        ids = torch.randint(0, num_rows, shape, generator=g)
    ids = ids.to(DEV)
    dy = torch.randn(*shape, d, generator=g).to(DEV)
    a = FE.embedding_grad(ids, dy, num_rows, padding_idx)
    b = FE.embedding_grad(ids, dy, num_rows, padding_idx)
    assert a.shape == (num_rows, d) and a.dtype == torch.float32 and torch.equal(a, b)
    ref = ref_grad(ids, dy, num_rows, padding_idx)
    scale = max(1.0, ref.abs().max().item())
    assert (a.double() - ref).abs().max().item() <= 2e-6 * scale * max(1, (ids.numel() // max(1, num_rows)) ** 0.5)
    if padding_idx >= 0:
        assert torch.all(a[padding_idx] == 0)
    untouched = torch.ones(num_rows, dtype=torch.bool, device=DEV)
    untouched[ids.reshape(-1)] = False
    assert torch.all(a[untouched] == 0)
    # torch's own backward of F.embedding on the same inputs
    w = torch.zeros(num_rows, d, device=DEV, requires_grad=True)
    F.embedding(ids, w, padding_idx if padding_idx >= 0 else None).backward(dy)
    assert torch.allclose(a, w.grad, rtol=1e-5, atol=1e-5 * scale)


def test_bert_embedding_block_matches_huggingface():
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    emb = BertModel(BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12,
                               type_vocab_size=2)).embeddings.to(DEV)
    emb.train()
    emb.dropout.p = 0.0                                   # compare values, not masks
    ids = torch.randint(1000, 30522, (16, 50), device=DEV)
    ids[:, 0] = 101
    ids[:, 40:] = 0
    assert FE.supported(emb, ids) and not FE.supported(emb, ids.cpu())
    g = torch.randn(16, 50, 768, device=DEV)
    y0 = emb(input_ids=ids)
    y0.backward(g)
    want = {n: p.grad.clone() for n, p in emb.named_parameters()}
    emb.zero_grad(set_to_none=True)
    y1 = FE.bert_embeddings(emb, ids)
    y1.backward(g)
    assert torch.equal(y0, y1)
    for n, p in emb.named_parameters():
        ref = want[n]
        assert p.grad is not None and p.grad.shape == ref.shape, n
        assert torch.allclose(p.grad, ref, rtol=1e-4, atol=1e-5 * max(1.0, ref.abs().max().item())), n


def test_bert_embedding_block_for_two_texts_with_one_table_gradient():
    """bert_embeddings_multi: the sentence and the scene caption share ONE word lookup (one table gradient launch);
    values and every gradient equal two separate HF embedding calls."""
    from transformers import BertConfig, BertModel
    torch.manual_seed(1)
    emb = BertModel(BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12,
                               type_vocab_size=2)).embeddings.to(DEV)
    emb.train()
    emb.dropout.p = 0.0
    a = torch.randint(1000, 30522, (8, 50), device=DEV)
    b = torch.randint(1000, 30522, (8, 300), device=DEV)
    b[:, :20] = a[:, :20]                                  # tokens shared between the two texts: duplicates across them
    ga, gb = torch.randn(8, 50, 768, device=DEV), torch.randn(8, 300, 768, device=DEV)
    ya, yb = emb(input_ids=a), emb(input_ids=b)
    (ya * ga).sum().backward()
    (yb * gb).sum().backward()
    want = {n: p.grad.clone() for n, p in emb.named_parameters()}
    emb.zero_grad(set_to_none=True)
    za, zb = FE.bert_embeddings_multi(emb, [a, b])
    ((za * ga).sum() + (zb * gb).sum()).backward()
    assert torch.equal(ya, za) and torch.equal(yb, zb)
    for n, p in emb.named_parameters():
        ref = want[n]
        assert torch.allclose(p.grad, ref, rtol=1e-4, atol=1e-5 * max(1.0, ref.abs().max().item())), n


def test_argument_errors():
    from sceneverse_amd import _native
    ids = torch.zeros(4, dtype=torch.int64, device=DEV)
    with pytest.raises(_native.GpsNativeError):
        FE.embedding_grad(ids, torch.randn(4, 6, device=DEV), 10)          # d not a multiple of 4
    out = FE.embedding_grad(ids[:0], torch.randn(0, 8, device=DEV), 10)    # no tokens: all-zero table
    assert out.shape == (10, 8) and torch.all(out == 0)
