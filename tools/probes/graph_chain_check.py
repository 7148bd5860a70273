"""Captures the split-graph data-parallel step once and exits right after the captures (no replay).  Run with
DEBUG_HIP_GRAPH_DOT_PRINT=1 in an empty working directory: ROCm writes one DOT file per instantiated graph there;
tests/test_gpu_graph_chain.py checks that every one of them is a linear chain.

    --legacy   re-create round 3's conditions: a fresh side stream per warm-up step, the previous step's stage boundary
               (= the autograd graph of the text / object encoders with their AccumulateGrad nodes) kept alive into the
               capture, torch's stream-mismatch warning silenced.  The bottom-backward graph then FORKS (the root cause of
               the corrupted gradients / memory-aperture violations of DESIGN.md section 9)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep

legacy = "--legacy" in sys.argv
classic = "--classic-wgrad" in sys.argv
cfg = gps_pretrain_cfg(_lang_dir())
for sec in (cfg.model.language, cfg.model.vision, cfg.model.grounding):
    if "num_hidden_layers" in sec.args:
        sec.args.num_hidden_layers = 1
    if "num_layers" in sec.args:
        sec.args.num_layers = 1
st = GPSTrainStep(cfg, device="cuda", ddp=False, graph="dp", graph_warmup=2, seed=7, wgrad_group=not classic)
st._debug_joint_bottom = legacy                                          # round 3: both encoders in one backward call
if legacy:
    import warnings
    warnings.filterwarnings("ignore", message=".*AccumulateGrad node's stream does not match.*")
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    st._stream = lambda: torch.cuda.Stream(device=st.device)           # a new stream per use, as round 3 did
    st._drop_previous_graph = lambda: None                              # the stage boundary survives into the next forward


def hook(stage, step, **kw):
    if stage == "captured_g2b":
        torch.cuda.synchronize()
        print("captured", flush=True)
        os._exit(0)


st.stage_hook = hook
for i in range(3):
    st.step(synth_batch(2, n_obj=8, seed=20 + i, min_real=3, device="cuda"))
print("no capture happened", flush=True)
sys.exit(1)
