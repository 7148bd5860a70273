"""The upper drop-in boundary, proven with the reference's OWN trainer: `trainer/default_trainer.py` and
`trainer/build.py` are imported UNMODIFIED from /root/reference (they are never copied), with the `model` and
`modules` packages shadowed by sceneverse_amd's as INTEGRATION.md section 2 prescribes; everything else the trainer
touches (Accelerate, the reference's own `optim` (Loss / AdamW / LambdaLR), `evaluator` (PretrainEval), `data.build`
DataLoader construction) is the reference's.  A synthetic dataset is registered in the reference's DATASET_REGISTRY
(that is configuration, not a trainer change).

    DefaultTrainer(cfg).train_step(0)          /root/reference/trainer/default_trainer.py:15-48
    BaseTrainer.__init__                       /root/reference/trainer/build.py:46-131   (build_model, build_optim, prepare)

Checked: the reference trainer runs its steps on our model classes, and the total loss it logs at every step equals
what `sceneverse_amd.engine.GPSTrainStep` computes from the same initial weights on the same batches (dropout
probabilities zeroed on both sides so that the comparison is deterministic).  CPU variant: point ops on the CPU oracle
(tests only); `-m gpu` variant: libgps_hip.so.  Where /root/reference does not exist (the GPU box) the same files are imported
from oracle/_ref/ref_python.zip (oracle/stage_ref_python.py; git-ignored, travels like oracle/_ref/*.so).
"""
import builtins
import copy
import os
import sys
import tempfile
from pathlib import Path

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STUBS = os.path.join(HERE, "golden", "ref_stubs")
# the reference's packages: the tree itself in the build container; on the GPU box (no /root/reference) the archive
# oracle/stage_ref_python.py packed from it -- the same files, byte for byte, imported through zipimport
_REF_DIR = "/root/reference"
_REF_ZIP = os.path.join(ROOT, "oracle", "_ref", "ref_python.zip")
if os.path.isdir(os.path.join(_REF_DIR, "trainer")) and os.environ.get("GPS_REF_FROM_ZIP") != "1":
    REF = _REF_DIR
else:
    REF = _REF_ZIP

pytestmark = pytest.mark.skipif(not os.path.exists(REF),
                                reason="neither /root/reference nor oracle/_ref/ref_python.zip (python oracle/stage_ref_python.py) exists")

N_SCENES, BATCH, N_OBJ = 4, 2, 6


def _shadow_and_import():
    """shadow `model` / `modules` before `trainer` is imported."""
    for p in (ROOT, REF, STUBS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [STUBS, REF, ROOT]
    builtins.__POINTNET2_SETUP__ = True
    import sceneverse_amd.model as model
    import sceneverse_amd.modules as modules
    sys.modules["model"], sys.modules["modules"] = model, modules
    sys.modules["model.build"], sys.modules["modules.build"] = model.build, modules.build
    import accelerate.utils as au                        # accelerate 1.14 renamed TPU -> XLA
    if not hasattr(au.DistributedType, "TPU"):
        au.DistributedType.TPU = au.DistributedType.XLA
    import trainer.build as tb                           # the reference's files, unmodified
    import trainer.default_trainer as dt
    assert os.path.realpath(dt.__file__).startswith(os.path.realpath(REF)), dt.__file__
    assert os.path.realpath(tb.__file__).startswith(os.path.realpath(REF)), tb.__file__
    import data.build as db
    import evaluator  # noqa: F401  registers PretrainEval
    return tb, dt, db


def _cfg(lang_path, exp_dir):
    from util import gps_cfg
    cfg = gps_cfg(lang_path)
    cfg.model.language.args.num_hidden_layers = 1
    cfg.model.vision.args.num_layers = 1
    cfg.model.grounding.args.num_layers = 1
    extra = {
        "name": "dropin", "rng_seed": 42, "mode": "train", "resume": False, "trainer": "DefaultTrainer",
        "exp_dir": Path(exp_dir) / "dropin_run" / "x", "debug": {"flag": True, "hard_debug": True},
        "logger": {"name": None, "entity": None, "run_id": None, "autoname": True},
        "dataloader": {"batchsize": BATCH, "num_workers": 0},
        "data_wrapper": "PassThroughWrapper",
        "eval": {"name": "PretrainEval", "save": False},
    }
    for k, v in extra.items():                      # item assignment wraps nested dicts into ConfigNodes
        cfg[k] = v
    for k in ("train", "val", "test"):
        cfg.data[k] = ["SynthGPS"]
    for k, v in {"epochs": 1, "grad_norm": 5.0, "epochs_per_eval": 0, "epochs_per_save": 0}.items():
        cfg.solver[k] = v
    return cfg


def _register_synthetic(db):
    from torch.utils.data import Dataset
    from data.datasets.dataset_wrapper import DATASETWRAPPER_REGISTRY
    from sceneverse_amd.data.synthetic import synth_batch

    if "SynthGPS" not in db.DATASET_REGISTRY:
        class SynthGPS(Dataset):
            def __init__(self, cfg, split):
                b = synth_batch(N_SCENES, n_obj=N_OBJ, n_pts=1024, scene_txt_len=40, seed=77, min_real=3)
                self.rows = [{k: v[i] for k, v in b.items()} for i in range(N_SCENES)]

            def __len__(self):
                return len(self.rows)

            def __getitem__(self, i):
                return self.rows[i]

        class PassThroughWrapper(Dataset):
            def __init__(self, cfg, dataset, split="train"):
                self.dataset = dataset

            def __len__(self):
                return len(self.dataset)

            def __getitem__(self, i):
                return self.dataset[i]

        db.DATASET_REGISTRY.register(SynthGPS)
        DATASETWRAPPER_REGISTRY.register(PassThroughWrapper)


def _zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(getattr(m, "dropout", None), float):      # nn.MultiheadAttention keeps a float
            m.dropout = 0.0
        for name in ("attention_probs_dropout_prob", "hidden_dropout_prob"):
            if hasattr(getattr(m, "config", None), name):
                setattr(m.config, name, 0.0)


def _run(device: str):
    tb, dt, db = _shadow_and_import()
    _register_synthetic(db)
    from util import lang_dir
    from sceneverse_amd.engine import GPSTrainStep
    lp = lang_dir(0)
    tmp = tempfile.mkdtemp()
    cfg = _cfg(lp, tmp)
    trainer = dt.DefaultTrainer(cfg)                               # reference class; build_model -> our registry
    import sceneverse_amd.model.openvocab as ours
    inner = trainer.accelerator.unwrap_model(trainer.model)
    assert type(inner) is ours.OpenVocab, type(inner)
    assert str(trainer.accelerator.device).startswith(device)
    _zero_dropout(inner)
    init = copy.deepcopy(inner.state_dict())
    seen, logged = [], []
    handle = inner.register_forward_pre_hook(
        lambda m, a: seen.append({k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in a[0].items()}))
    trainer.log = lambda results, mode="train": logged.append({k: float(v) for k, v in results.items()})
    trainer.train_step(0)                                          # the reference's loop body, 2 optimisation steps
    handle.remove()
    assert len(logged) == N_SCENES // BATCH == len(seen)
    assert trainer.global_step == len(logged)
    for row in logged:
        assert {"total_loss", "lm_cls_loss", "TextObjWithinBatch", "TextSceneBetweenBatch", "og_acc"} <= row.keys(), row
        assert all(v == v for v in row.values()), row

    step = GPSTrainStep(_cfg(lp, tmp), device=device, amp_dtype=None, ddp=False, total_steps=len(logged),
                        native_optimizer=(device != "cpu"), fused_lm_loss=False)
    step.model.load_state_dict(init, strict=True)
    _zero_dropout(step.model)
    from sceneverse_amd.modules.layers import gemm as _gemm
    _gemm.invalidate_shadows()
    tol = 1e-5 if device == "cpu" else 2e-4
    for i, (batch, row) in enumerate(zip(seen, logged)):
        total, losses = step.step({k: v for k, v in batch.items() if torch.is_tensor(v)})
        for k, v in losses.items():
            assert abs(float(v) - row[k]) <= tol * max(1.0, abs(row[k])), (i, k, float(v), row[k])
    # after the same two updates the weights agree too (reference AdamW + clip_grad_norm_ vs ours)
    worst = 0.0
    mine = step.model.state_dict()
    for k, v in inner.state_dict().items():
        if torch.is_floating_point(v):
            worst = max(worst, float((v - mine[k]).abs().max()))
    assert worst <= (1e-6 if device == "cpu" else 5e-5), worst


@pytest.mark.timeout(900)
def test_reference_default_trainer_runs_on_the_shadowed_packages_cpu(monkeypatch):
    from util import use_oracle_ext
    monkeypatch.setenv("ACCELERATE_USE_CPU", "true")
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self, raising=False)
    with use_oracle_ext():
        _run("cpu")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_reference_default_trainer_runs_on_the_shadowed_packages_gpu():
    _run("cuda")
