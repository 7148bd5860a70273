"""Eager pre-train steps with every tensor in its own hipMalloc (PYTORCH_NO_CUDA_MEMORY_CACHING=1) and blocking launches:
an out-of-bounds access of any kernel then tends to fault at the offending op instead of landing in a neighbour's block.
FUSE_POST=1 switches the LayerNorm post-addend on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep

if os.environ.get("FUSE_POST"):
    from sceneverse_amd.modules.layers.transformers import set_fuse_post_add
    set_fuse_post_add(True)
DEV = "cuda"
B, O = int(os.environ.get("PROBE_B", "16")), int(os.environ.get("PROBE_OBJ", "80"))
st = GPSTrainStep(gps_pretrain_cfg(_lang_dir()), device=DEV, ddp=False, graph=False, seed=7)
for i in range(2):
    total, _ = st.step(dict(synth_batch(B, n_obj=O, seed=30 + i, device=DEV)))
    torch.cuda.synchronize()
    print("step", i, "loss", total.item(), flush=True)
print("clean")
