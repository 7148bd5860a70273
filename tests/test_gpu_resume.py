"""Checkpoint resume in the graph modes: `Optimizer.load_state_dict` replaces every param group's
`lr` tensor with a copy; the captured AdamW and the scheduler's one-copy update must keep talking about the SAME device
word afterwards (sceneverse_amd/engine.py `_rebind_lr`).  Reference behaviour: trainer/build.py:160-187 restores
optimizer + scheduler through Accelerate and the learning rate then follows the schedule."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _engine(graph):
    from bench import gps_pretrain_cfg, _lang_dir
    from sceneverse_amd.engine import GPSTrainStep
    cfg = gps_pretrain_cfg(_lang_dir())
    for sec in (cfg.model.language, cfg.model.vision, cfg.model.grounding):
        if "num_hidden_layers" in sec.args:
            sec.args.num_hidden_layers = 1
        if "num_layers" in sec.args:
            sec.args.num_layers = 1
    return GPSTrainStep(cfg, device=DEV, ddp=False, graph=graph, graph_warmup=2, seed=3, total_steps=50)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("graph", [True, "dp"])
def test_learning_rate_follows_the_schedule_after_a_resume_into_a_graph_engine(graph):
    from sceneverse_amd.data.synthetic import synth_batch
    batches = [synth_batch(2, n_obj=8, seed=40 + i, min_real=3, device=DEV) for i in range(8)]
    a = _engine(graph)
    for i in range(3):
        a.step(dict(batches[i]))
    opt_sd, sch_sd = copy.deepcopy(a.optimizer.state_dict()), copy.deepcopy(a.scheduler.state_dict())
    model_sd = copy.deepcopy(a.model.state_dict())
    want = []                                     # the uninterrupted run's learning rates of steps 3..7
    for i in range(3, 8):
        a.step(dict(batches[i]))
        want.append([float(g["lr"]) for g in a.optimizer.param_groups])
    del a
    b = _engine(graph)
    b.model.load_state_dict(model_sd)
    b.optimizer.load_state_dict(opt_sd)           # before the first captured step, as GpsAdamW documents
    b.scheduler.load_state_dict(sch_sd)
    got = []
    for i in range(3, 8):                         # two eager warm-up steps, the capture, two replays
        b.step(dict(batches[i]))
        got.append([float(g["lr"]) for g in b.optimizer.param_groups])
        for gi, g in enumerate(b.optimizer.param_groups):      # the live lr IS the word the scheduler writes
            assert g["lr"].data_ptr() == b._lr_dev[gi].data_ptr()
    assert b._graph is not None
    assert len({tuple(r) for r in got}) == len(got), got       # it moves (warm-up ramp of the cosine schedule)
    for w, g in zip(want, got):
        assert w == pytest.approx(g, rel=1e-6), (want, got)
