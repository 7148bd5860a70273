"""Eager check of the two-segment backward (engine._graph_dp_step) against one-pass backward on the GPU model."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep

DEV = "cuda"
cfg = gps_pretrain_cfg(_lang_dir())
st = GPSTrainStep(cfg, device=DEV, ddp=False, graph=False, seed=7)
for m in st.model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
        m.dropout = 0.0
st.net.train()
b = synth_batch(4, n_obj=16, seed=20, min_real=5, device=DEV)
params = [p for p in st.model.parameters() if p.requires_grad]
names = {id(p): n for n, p in st.model.named_parameters()}


def run(staged):
    for p in params:
        p.grad = None
    out, total, losses = st.forward_loss(dict(b))
    if not staged:
        total.backward()
    else:
        bottom_ids = set()
        for name in ("lang_encoder", "point_encoder"):
            bottom_ids.update(id(p) for p in getattr(st.model, name).parameters())
        top = [p for p in params if id(p) not in bottom_ids]
        bottom = [p for p in params if id(p) in bottom_ids]
        boundary = list(st.model._stage_boundary)
        print("boundary", [tuple(t.shape) for t in boundary], [type(t.grad_fn).__name__ for t in boundary])
        torch.autograd.backward(total, inputs=top + boundary, retain_graph=True)
        live = [t for t in boundary if t.grad is not None]
        torch.autograd.backward(live, grad_tensors=[t.grad for t in live], inputs=bottom)
    return {names[id(p)]: (None if p.grad is None else p.grad.detach().clone()) for p in params}, total.item()


ga, la = run(False)
gb, lb = run(True)
print("loss", la, lb)
bad = 0
for n in ga:
    a, c = ga[n], gb[n]
    if (a is None) != (c is None):
        print("presence differs", n, a is None, c is None); bad += 1; continue
    if a is None:
        continue
    if not torch.isfinite(c).all():
        print("NONFINITE staged", n); bad += 1; continue
    rel = ((a - c).norm() / (a.norm() + 1e-20)).item()
    if rel > 1e-3:
        print("differs", n, rel, a.norm().item(), c.norm().item()); bad += 1
print("bad", bad, "of", len(ga))
