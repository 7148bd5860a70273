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
    ((7000,), 100, 1028, -1, "dense"),           # ~70 tokens per row: more heavy ids (>= 64 tokens) than the 64 the parallel path takes
    ((64, 350), 30522, 768, 0, "masked"),        # the bench text: 15 % [MASK] (one id ~2 000 times), CLS / SEP x 64
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
    elif kind == "masked":
        B, L = shape
        ids = torch.randint(1000, num_rows, shape, generator=g)
        ids[torch.rand(shape, generator=g) < 0.15] = 103
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


class _Emb(torch.nn.Module):
    """The parameters of HF BertEmbeddings (bert-base sizes) without the HF class."""

    def __init__(self, vocab=30522, d=768, n_pos=512, p=0.1, seed=3):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.word_embeddings = torch.nn.Embedding(vocab, d, padding_idx=0)
        self.position_embeddings = torch.nn.Embedding(n_pos, d)
        self.token_type_embeddings = torch.nn.Embedding(2, d)
        self.LayerNorm = torch.nn.LayerNorm(d, eps=1e-12)
        self.dropout = torch.nn.Dropout(p)
        with torch.no_grad():
            for prm in self.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.3 if prm.dim() == 2 else 1.0))
            self.LayerNorm.weight.add_(1.0)


@pytest.mark.parametrize("live", [None, 1500, 0, 2048])
def test_bert_embed_rows_forward_and_backward_match_the_torch_chain(live):
    """gps_bert_embed_forward / backward (lookups + LayerNorm, no dropout) against word + type + pos -> layer_norm in
    fp64: outputs (fp32 and the bf16 copy), and the gradients of all five parameters, with and without a device-side
    row count (rows past it: not written forward, no contribution backward)."""
    n, d = 2048, 768
    emb = _Emb().to(DEV)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(1, 30522, (n,), generator=g)
    ids[::7] = 101                                             # duplicates
    ids[5::31] = 0                                             # the padding row: no gradient
    pos = torch.randint(0, 300, (n,), generator=g)
    ids, pos = ids.to(DEV), pos.to(DEV)
    rows_dev = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
    m = n if live is None else live
    y, y16 = FE.bert_embeddings_rows(emb, ids, pos, rows_dev=rows_dev, training=False)
    # fp64 reference on the live rows
    prm64 = {k: v.detach().double().requires_grad_(True) for k, v in emb.named_parameters()}
    e = prm64["word_embeddings.weight"][ids[:m]] + prm64["token_type_embeddings.weight"][0] + prm64["position_embeddings.weight"][pos[:m]]
    ref = F.layer_norm(e, (d,), prm64["LayerNorm.weight"], prm64["LayerNorm.bias"], 1e-12)
    if m:
        assert (y[:m].double() - ref).abs().max().item() <= 2e-5
        assert (y16[:m].double() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item()
    wy = torch.randn(n, d, generator=g).to(DEV)
    wy16 = (torch.randn(n, d, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    if live is not None:                                        # dead rows carry garbage gradients in the real step
        wy[m:] = float("nan")
        wy16[m:] = float("nan")
    for prm in emb.parameters():
        prm.grad = None
    torch.autograd.backward([y, y16], [wy, wy16])
    (ref * (wy[:m].double() + wy16[:m].double())).sum().backward()
    for name, prm in emb.named_parameters():
        got, want = prm.grad.double(), prm64[name].grad.clone()
        if name == "word_embeddings.weight":
            want[0] = 0                                          # nn.Embedding(padding_idx=0): that row gets no gradient
        assert torch.isfinite(got).all(), name
        scale = max(1.0, want.abs().max().item())
        assert (got - want).abs().max().item() <= 3e-5 * scale, (name, (got - want).abs().max().item(), scale)
    assert torch.all(emb.word_embeddings.weight.grad[0] == 0)     # padding row
    assert torch.all(emb.token_type_embeddings.weight.grad[1] == 0)


def test_bert_embed_rows_dropout_mask_is_the_same_in_both_directions():
    n, d, p = 1024, 768, 0.1
    emb = _Emb(p=p).to(DEV)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1, 30522, (n,), generator=g).to(DEV)
    pos = torch.randint(0, 300, (n,), generator=g).to(DEV)
    torch.manual_seed(0)
    y, y16 = FE.bert_embeddings_rows(emb, ids, pos, training=True)
    y0, _ = FE.bert_embeddings_rows(emb, ids, pos, training=False)
    dropped = y == 0
    frac = dropped.float().mean().item()
    assert abs(frac - p) < 0.01, frac
    keep = ~dropped
    assert torch.allclose(y[keep], y0[keep] / (1 - p), rtol=1e-6, atol=1e-6)
    assert torch.equal(y16, y.to(torch.bfloat16))
    # backward: a cotangent that only touches DROPPED elements produces exactly zero gradients everywhere
    for prm in emb.parameters():
        prm.grad = None
    torch.autograd.backward([y, y16], [dropped.float(), torch.zeros_like(y16)])
    for name, prm in emb.named_parameters():
        assert torch.all(prm.grad == 0), name


def test_position_gradient_per_position_form_equals_the_scatter_form():
    """Rows = whole sequences end to end, pos = offset inside the sequence: gps_bert_position_grad (cu_rows given) and
    gps_embedding_grad on the position ids must give the same table gradient (fp32 summation order aside), with a
    device-side row count that leaves a dead tail."""
    d = 768
    emb = _Emb().to(DEV)
    g = torch.Generator().manual_seed(21)
    lens = torch.randint(1, 301, (128,), generator=g)
    lens[3] = 300
    n_live = int(lens.sum())
    n = n_live + 777                                             # dead tail (capacity rows)
    cu = torch.zeros(129, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    pos = torch.cat([torch.arange(int(L)) for L in lens] + [torch.zeros(777, dtype=torch.long)]).to(DEV)
    ids = torch.randint(1, 30522, (n,), generator=g).to(DEV)
    rows_dev = torch.tensor([n_live], dtype=torch.int32, device=DEV)
    wy = torch.randn(n, d, generator=g).to(DEV)
    grads = []
    for cu_rows in (None, cu.to(DEV)):
        for prm in emb.parameters():
            prm.grad = None
        y, y16 = FE.bert_embeddings_rows(emb, ids, pos, rows_dev=rows_dev, training=False, cu_rows=cu_rows)
        torch.autograd.backward([y[:n_live]], [wy[:n_live]])
        grads.append({k: v.grad.clone() for k, v in emb.named_parameters()})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item()), k
    assert torch.all(grads[1]["position_embeddings.weight"][300:] == 0)


def test_embedding_grad_inside_a_replayed_graph():
    """The whole call (init, mark, heavy / duplicate lists, fill, sums) captured into a HIP graph over poisoned pool
    memory and replayed: every replay equals the eager result bit for bit.  (Round 5: the two adjacent hipMemsetAsync
    calls this entry point used to make left `count` with a wrong pattern in REPLAYS of the training step's graph -- not
    in eager runs and not in a small capture like this one -- which the old kernel survived slowly and the new heavy
    path did not; the tables are now initialised by a kernel.)"""
    g0 = torch.Generator(device="cpu").manual_seed(3)
    ids = torch.randint(1000, 30522, (64, 350), generator=g0)
    ids[torch.rand(64, 350, generator=g0) < 0.15] = 103
    ids[:, 0] = 101
    ids[:, 200:] = 0
    ids = ids.to(DEV)
    dy = torch.randn(64 * 350, 768, generator=g0).to(DEV)
    ref = FE.embedding_grad(ids, dy, 30522, 0).clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            poison = torch.empty(16 << 20, dtype=torch.int32, device=DEV).fill_(0x7F7F7F7F)
            del poison
            out = FE.embedding_grad(ids, dy, 30522, 0)
        for _ in range(4):
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
