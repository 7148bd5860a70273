"""Time gps_embedding_grad on the ids of the bench batch (two BERT passes: 64 x 50 + 64 x 300 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.modules.language import fused_embedding as FE

dev = torch.device("cuda", 0)
b = synth_batch(64, n_obj=80, n_pts=1024, txt_len=50, seed=42, device=dev)
ids = torch.cat([b["txt_ids"].reshape(-1), b["scene_txt_ids"].reshape(-1)])
print("tokens", ids.numel(), "non-pad", int((ids != 0).sum()), "distinct", int(ids.unique().numel()))
dy = torch.randn(ids.numel(), 768, device=dev)
for _ in range(3):
    out = FE.embedding_grad(ids, dy, 30522, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    e0.record(); out = FE.embedding_grad(ids, dy, 30522, 0); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("embedding_grad us: min %.1f median %.1f max %.1f" % (ts[0], ts[len(ts) // 2], ts[-1]))
ref = torch.zeros(30522, 768, device=dev, dtype=torch.float64)
keep = ids != 0
ref.index_add_(0, ids[keep], dy[keep].double())
print("max abs err vs fp64", (out.double() - ref).abs().max().item())
