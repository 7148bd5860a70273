"""Gradients of the FIRST replay of the segmented graph step, saved per parameter; `diff A B` compares two saved runs.
FUSE_POST=1 turns the LayerNorm post-addend on (GPS_POST_ONLY=spatial|plain restricts it to one encoder)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

if sys.argv[1] == "diff":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for n in a:
        x, y = a[n].float(), b[n].float()
        rel = ((x - y).norm() / (x.norm() + 1e-20)).item()
        if rel > 5e-3 or not torch.isfinite(y).all():
            bad += 1
            print(f"differs {n}: rel {rel:.3e} |a| {x.norm().item():.3e} |b| {y.norm().item():.3e}")
    print("bad", bad, "of", len(a))
    sys.exit(0)

from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep
from sceneverse_amd.modules.layers.transformers import MultiheadSelfAttention, set_fuse_post_add

DEV = "cuda"
if os.environ.get("FUSE_POST"):
    set_fuse_post_add(True, os.environ.get("GPS_POST_ONLY"))
if os.environ.get("NO_CLS_TAIL"):
    from sceneverse_amd.modules.language import bert as _B
    _B.set_cls_tail(False)
junk = [torch.full((256, 1024, 1024), float("nan"), device=DEV) for _ in range(40)]
del junk
cfg = gps_pretrain_cfg(_lang_dir())
st = GPSTrainStep(cfg, device=DEV, ddp=False, graph=("dp" if len(sys.argv) <= 3 or sys.argv[3] == "dp" else True), graph_warmup=2, seed=7)
for m in st.model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if isinstance(m, MultiheadSelfAttention):
        m.dropout = 0.0
    if hasattr(m, "attention_probs_dropout_prob"):
        m.attention_probs_dropout_prob = 0.0
BS = int(os.environ.get("PROBE_B", "4"))
NO = int(os.environ.get("PROBE_OBJ", "16"))
batches = [synth_batch(BS, n_obj=NO, seed=20 + i, min_real=5, device=DEV) for i in range(3)]
for b in batches:
    total, _ = st.step(dict(b))
torch.cuda.synchronize()
print("loss", total.item(), "graph", st._graph is not None)
torch.save({n: p.grad.detach().to(torch.bfloat16).cpu() for n, p in st.model.named_parameters() if p.grad is not None}, sys.argv[2])
