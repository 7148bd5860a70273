"""After the first replay of the segmented data-parallel graph step: which gradient views hold non-finite values?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep

DEV = "cuda"
if os.environ.get("FUSE_POST"):
    from sceneverse_amd.modules.layers.transformers import set_fuse_post_add
    set_fuse_post_add(True, os.environ.get("GPS_POST_ONLY"))
if os.environ.get("POISON"):
    junk = [torch.full((256, 1024, 1024), float("nan"), device=DEV) for _ in range(int(os.environ["POISON"]))]
    del junk
cfg = gps_pretrain_cfg(_lang_dir())
from sceneverse_amd.modules.layers.transformers import MultiheadSelfAttention
cfg.solver.sched.args.warmup_steps = 4
st = GPSTrainStep(cfg, device=DEV, ddp=False, graph="dp", graph_warmup=2, seed=7)
for m in st.model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if isinstance(m, MultiheadSelfAttention):
        m.dropout = 0.0
    if hasattr(m, "attention_probs_dropout_prob"):
        m.attention_probs_dropout_prob = 0.0
    if hasattr(m, "dropout_prob"):
        m.dropout_prob = 0.0
batches = [synth_batch(4, n_obj=16, seed=20 + i, min_real=5, device=DEV) for i in range(5)]
for i, b in enumerate(batches):
    total, _ = st.step(dict(b))
    torch.cuda.synchronize()
    print("step", i, "loss", total.item(), "graph" if st._graph is not None else "eager")
    if st._graph is not None:
        flat = st._flat_grad
        print("  flat grad finite:", bool(torch.isfinite(flat).all()), "n_top", st._n_top, "numel", flat.numel())
        for n, p in st.model.named_parameters():
            if p.grad is not None and not torch.isfinite(p.grad).all():
                print("  NONFINITE grad", n, int((~torch.isfinite(p.grad)).sum()), "of", p.grad.numel())
        bad = [n for n, p in st.model.named_parameters() if not torch.isfinite(p).all()]
        print("  nonfinite params:", bad[:10], len(bad))
        for t in st.model._stage_boundary:
            print("  boundary grad finite", tuple(t.shape), None if t.grad is None else bool(torch.isfinite(t.grad).all()))
