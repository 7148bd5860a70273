"""CPU baseline of bench.py, kind "reference": the REFERENCE's OWN Python -- `model.openvocab.OpenVocab`, its `modules/`
layers, `optim.loss.loss.Loss` -- imported unmodified and run on the host cores in fp32, with the CPU oracle
(oracle/pointnet2_oracle.py) injected as `pointnet2_utils._ext` (the reference's native ops assert "CPU not supported":
its Python can only run on a CPU on top of the oracle).  Recipe = SURVEY.md App. G / tests/golden/make_golden.py.

TEST / MEASUREMENT INFRASTRUCTURE ONLY: imported by bench.py's `cpu_baseline` worker process and by nothing under
sceneverse_amd/.  The reference's files come from /root/reference where it exists (the build container) and from
oracle/_ref/ref_python.zip (oracle/stage_ref_python.py: the same files, byte for byte, through zipimport) on the GPU box.

One step = what trainer/default_trainer.py:30-48 runs: forward, Loss, backward, clip_grad_norm_, AdamW step
(optim/optimizer/optim.py:9-14 builds torch.optim.AdamW by name), on the batch sceneverse_amd.data.synthetic makes
(the reference's loaders need datasets that are not available here: SURVEY.md section 8(d))."""
from __future__ import annotations

import builtins
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STUBS = os.path.join(ROOT, "tests", "golden", "ref_stubs")
_REF_DIR = "/root/reference"
_REF_ZIP = os.path.join(HERE, "_ref", "ref_python.zip")


def reference_root() -> str | None:
    if os.path.isdir(os.path.join(_REF_DIR, "modules")) and os.environ.get("GPS_REF_FROM_ZIP") != "1":
        return _REF_DIR
    if os.path.exists(_REF_ZIP):
        import zipfile
        with zipfile.ZipFile(_REF_ZIP) as z:
            if any(n.startswith("modules/") for n in z.namelist()):
                return _REF_ZIP
    return None


def _import_reference(ref: str):
    """App. G steps 1 - 5.  Must run in a process that has not imported sceneverse_amd's `model` / `modules` shadows."""
    import torch
    import torch.nn as nn
    for p in (STUBS, ref, ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [STUBS, ref, ROOT]
    builtins.__POINTNET2_SETUP__ = True
    torch.Tensor.cuda = lambda self, *a, **k: self                     # unified_encoder.py:157,162 hard-codes .cuda()
    from oracle.pointnet2_oracle import OracleExt
    import modules.third_party.pointnet2.pointnet2_modules  # noqa: F401  (appends its directory to sys.path)
    import pointnet2_utils                                   # the top-level alias the SA modules really use
    pointnet2_utils._ext = OracleExt
    import modules.build as mb
    from transformers import BertConfig, BertModel

    class OfflineBert(nn.Module):                           # BertTokenizer / BertModel.from_pretrained need the network
        def __init__(self, cfg, weights=None, hidden_size=768, num_hidden_layers=4, num_attention_heads=12,
                     type_vocab_size=2, **kw):
            super().__init__()
            self.model = BertModel(BertConfig(hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                                              num_attention_heads=num_attention_heads, type_vocab_size=type_vocab_size))

        def forward(self, txt_ids, txt_masks, **kw):
            return self.model(txt_ids, txt_masks).last_hidden_state

    import modules  # noqa: F401  registers everything
    mb.LANGUAGE_REGISTRY._obj_map['BERTLanguageEncoder'] = OfflineBert
    import model  # noqa: F401
    import optim.loss.contra_loss  # noqa: F401
    from model.build import build_model
    from optim.loss.loss import Loss
    ref_real = os.path.realpath(ref)
    for mod in (sys.modules["model.openvocab"], sys.modules["modules.layers.transformers"], sys.modules["optim.loss.loss"]):
        assert os.path.realpath(mod.__file__).startswith(ref_real), mod.__file__      # the reference's files, not ours
    return build_model, Loss


def run(batch_size: int, steps: int, n_obj: int, n_pts: int, threads: int = 0, budget_s: float = 30.0) -> dict:
    ref = reference_root()
    if ref is None:
        raise RuntimeError("neither /root/reference nor oracle/_ref/ref_python.zip (with modules/ and model/) exists")
    import torch
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if threads > 0:
        cores = max(1, min(cores, threads))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    build_model, Loss = _import_reference(ref)
    from bench import N_CLS, gps_pretrain_cfg
    from sceneverse_amd.data.synthetic import synth_batch
    tmp = tempfile.mkdtemp(prefix="gps_txt_")
    torch.save(0.02 * torch.randn(N_CLS, 768), os.path.join(tmp, "scannet_607_bert-base-uncased_id.pth"))
    cfg = gps_pretrain_cfg(tmp)
    model = build_model(cfg).train()
    for m in model.modules():                  # same as the port: dropout off (the timing does not depend on it)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    loss_mod = Loss(cfg)
    params = [p for p in model.parameters() if p.requires_grad] + [p for p in loss_mod.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.98))
    data = synth_batch(batch_size, n_obj=n_obj, n_pts=n_pts, seed=123)

    def one_step():
        d = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
        d["cur_step"], d["total_steps"] = 1, 1000
        out = model(d)
        loss, _ = loss_mod(out)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return float(loss)

    one_step()  # warm-up
    t0 = time.perf_counter()
    done, last = 0, 0.0
    for _ in range(steps):
        last = one_step()
        done += 1
        if done >= 3 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    src = "/root/reference" if ref == _REF_DIR else "oracle/_ref/ref_python.zip"
    return {"value": batch_size * done / dt, "unit": "pairs/s", "cores": cores, "kind": "reference",
            "sample": f"{done} steps at B={batch_size} ({n_obj} obj x {n_pts} pts, 50+300 tokens) of the reference's own "
                      f"model.openvocab.OpenVocab + optim.loss.loss.Loss ({src}), point ops on the CPU oracle, fp32 torch "
                      f"on {cores} threads, 1 warm-up; {dt:.1f} s; final loss {last:.3f}"}


if __name__ == "__main__":
    import json
    a = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [2, 3, 8, 256, 8]
    print(json.dumps(run(*a)))
