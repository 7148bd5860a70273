import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep
from sceneverse_amd.modules.layers import pointnet as PN
from sceneverse_amd.pointnet2 import _ext
dev = torch.device("cuda", 0)
cfg = bench.gps_pretrain_cfg(bench._lang_dir(), num_gpu=1, workload="pretrain")
step = GPSTrainStep(cfg, device=dev, amp_dtype=torch.bfloat16, graph=False)
batch = synth_batch(64, n_obj=80, n_pts=1024, txt_len=50, seed=42, device=dev)
pcs = batch["obj_fts"].reshape(-1, 1024, 6)
plan = _ext.cloud_compact(pcs)
print("plan scal", plan.scal.tolist())
enc = step.model.point_encoder.point_feature_extractor
step.model.point_encoder.freeze_bn(enc)
print("freeze", step.model.point_encoder.freeze, "training", enc.training)
print("distinct ok:", enc._distinct_ok(pcs), "grad enabled", torch.is_grad_enabled(), "any requires_grad", any(p.requires_grad for p in enc.parameters()))
for flag in (True, False, True):
    PN.set_distinct_clouds(flag)
    for _ in range(2):
        y = enc(pcs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        y = enc(pcs)
    rec = None
    torch.cuda.synchronize(); print("distinct", flag, "encoder ms", (time.perf_counter() - t0) / 5 * 1e3, float(y.float().abs().sum()))

_ext.profile_start()
PN.set_distinct_clouds(True)
y = enc(pcs)
rec = _ext.profile_stop()
print(sorted(rec.keys()))
