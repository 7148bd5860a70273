"""Child process of tests/test_gpu_graph_chain.py: captures the split-graph data-parallel step once and prints the shape
of every captured hipGraph (nodes, edges, nodes with more than one successor) as one JSON line, read straight from the
runtime with hipGraphGetNodes / hipGraphGetEdges (no dependence on DEBUG_HIP_GRAPH_DOT_PRINT).

    --legacy   re-create round 3's conditions in a SUBCLASS of the engine: a fresh side stream per use, the previous
               step's stage boundary (= the autograd graph of the text / object encoders with their AccumulateGrad nodes)
               kept alive into the capture, both encoders in one backward call, torch's stream-mismatch warning silenced.
               The bottom-backward graph then FORKS (root cause of the corrupted gradients of DESIGN.md section 9a)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep

legacy = "--legacy" in sys.argv
classic = "--classic-wgrad" in sys.argv


def graph_shape(raw_graph: int):
    hip = ctypes.CDLL("libamdhip64.so")
    g = ctypes.c_void_p(raw_graph)
    n_nodes, n_edges = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, ctypes.byref(n_nodes)) == 0
    assert hip.hipGraphGetEdges(g, None, None, ctypes.byref(n_edges)) == 0
    src = (ctypes.c_void_p * max(n_edges.value, 1))()
    dst = (ctypes.c_void_p * max(n_edges.value, 1))()
    assert hip.hipGraphGetEdges(g, src, dst, ctypes.byref(n_edges)) == 0
    fan_out = {}
    for i in range(n_edges.value):
        fan_out[src[i]] = fan_out.get(src[i], 0) + 1
    return {"nodes": n_nodes.value, "edges": n_edges.value, "forks": sum(1 for v in fan_out.values() if v > 1)}


class Inspectable(GPSTrainStep):
    def _new_graph(self):
        g = torch.cuda.CUDAGraph(keep_graph=True)          # the hipGraph_t stays alive after capture_end
        self.__dict__.setdefault("_made", []).append(g)
        return g


class Round3(Inspectable):
    def _stream(self):                                      # a new stream per use
        return torch.cuda.Stream(device=self.device)

    def _drop_previous_graph(self):                         # the stage boundary survives into the next forward
        pass

    @staticmethod
    def _bottom_groups(live, boundary, bottom_segs):        # both encoders in ONE backward call
        return [live]

    def _strict_accumulate_grad(self):
        import contextlib
        return contextlib.nullcontext()


cfg = gps_pretrain_cfg(_lang_dir())
for sec in (cfg.model.language, cfg.model.vision, cfg.model.grounding):
    if "num_hidden_layers" in sec.args:
        sec.args.num_hidden_layers = 1
    if "num_layers" in sec.args:
        sec.args.num_layers = 1
st = (Round3 if legacy else Inspectable)(cfg, device="cuda", ddp=False, graph="dp", graph_warmup=2, seed=7,
                                         wgrad_group=not classic)
if legacy:
    import warnings
    warnings.filterwarnings("ignore", message=".*AccumulateGrad node's stream does not match.*")
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def hook(stage, step, **kw):
    if stage == "captured_g2b":
        torch.cuda.synchronize()
        shapes = []
        for g in step._made:
            try:
                raw = g.raw_cuda_graph()
            except RuntimeError:                   # created but not captured yet (the clip + AdamW graph)
                continue
            shapes.append(graph_shape(raw))
        print("captured " + json.dumps(shapes), flush=True)
        os._exit(0)


st.stage_hook = hook
for i in range(3):
    st.step(synth_batch(2, n_obj=8, seed=20 + i, min_real=3, device="cuda"))
print("no capture happened", flush=True)
sys.exit(1)
