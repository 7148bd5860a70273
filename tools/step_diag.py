"""Is the training step GPU-bound or host(launch)-bound?  Times the host side of each step (issue
only, no sync) next to the synchronised wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep

dev = torch.device("cuda", 0)
step = GPSTrainStep(gps_pretrain_cfg(_lang_dir()), device=dev)
batch = synth_batch(64, seed=42, device=dev)
for _ in range(3):
    step.step(dict(batch))
torch.cuda.synchronize()
issue = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    step.step(dict(batch))
    issue.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue per step: {1e3 * t_issue / 10:.2f} ms   wall per step: {1e3 * t_all / 10:.2f} ms")
print("per-step issue ms:", [round(1e3 * x, 1) for x in issue])
# forward only / backward only split of the host time
import contextlib
torch.cuda.synchronize()
a = time.perf_counter()
out, total, losses = step.forward_loss(dict(batch, cur_step=0, total_steps=1 << 30))
b = time.perf_counter()
total.backward()
c = time.perf_counter()
torch.nn.utils.clip_grad_norm_(step.model.parameters(), 5.0)
step.optimizer.step()
d = time.perf_counter()
torch.cuda.synchronize()
print(f"host: forward+loss {1e3*(b-a):.1f} ms, backward {1e3*(c-b):.1f} ms, clip+opt {1e3*(d-c):.1f} ms")
