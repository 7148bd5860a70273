"""Host-side mirror of the reference interface (registries, modules, models, losses) checked on
CPU against the golden outputs of the reference's own Python.  The point ops are routed to the
CPU oracle here (tests only) -- on a GPU the same modules run libgps_hip.so, see test_gpu_*.py."""
import torch
import torch.nn as nn

from oracle.param_fill import fill_params
from sceneverse_amd.common.config import ConfigNode
from sceneverse_amd.model.build import MODEL_REGISTRY, build_model
from sceneverse_amd.optim.loss import Loss
from sceneverse_amd.optim.loss.loss import obj_cls_loss
from util import clone_batch, gps_cfg, lang_dir, use_oracle_ext


def _build(fx, **kw):
    with use_oracle_ext():
        model = build_model(gps_cfg(lang_dir(fx["seed"]), **kw))
    fill_params(model, fx["seed"])
    return model


def test_state_dict_keys_and_param_groups_match_reference(golden_cpu):
    g = golden_cpu["gps_pretrain"]
    model = _build(golden_cpu)
    assert sorted(model.state_dict().keys()) == g["state_dict_keys"]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == g["n_trainable"]
    assert len(model.get_opt_params()) == g["n_opt_groups"]


def test_gps_pretrain_forward_loss_backward(golden_cpu):
    fx = golden_cpu
    g = fx["gps_pretrain"]
    model = _build(fx).eval()
    cfg = model.cfg
    loss_mod = Loss(cfg)
    with use_oracle_ext():
        out = model(clone_batch(fx["batch"]))
        total, losses = loss_mod(out)
        total.backward()
    for k in ("og3d_logits", "intra_text_embed", "intra_obj_embeds", "inter_obj_embeds",
              "scene_embed", "scene_text_embed", "obj_cls_post_logits"):
        torch.testing.assert_close(out[k], g[k], rtol=1e-4, atol=1e-4, msg=lambda m, k=k: f"{k}: {m}")
    torch.testing.assert_close(torch.logsumexp(out["txt_lm_cls_logits"], 2), g["txt_lm_cls_logits_lse"],
                               rtol=1e-4, atol=1e-4)
    for k, v in g["losses"].items():
        assert abs(float(losses[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(losses[k]), v)
    params = dict(model.named_parameters())
    for name, ref in g["grads"].items():
        grad = params[name].grad
        assert abs(grad.norm().item() - ref["norm"]) <= 2e-3 * ref["norm"] + 1e-7, name
        torch.testing.assert_close(grad.flatten()[:256], ref["head"], rtol=2e-3,
                                   atol=2e-4 * ref["norm"] + 1e-8, msg=lambda m, n=name: f"{n}: {m}")
    # frozen PointNet++ receives no gradient; 13 trainable tensors are never used
    assert all(p.grad is None for n, p in params.items() if n.startswith("point_encoder.point_feature_extractor"))
    unused = [n for n, p in params.items() if p.requires_grad and p.grad is None]
    assert len(unused) == 13, unused


def test_grounding_finetune_forward(golden_cpu):
    fx = golden_cpu
    g = fx["gps_ground"]
    model = _build(fx, heads="ground", use_scene_cap=False).eval()
    with use_oracle_ext(), torch.no_grad():
        out = model(clone_batch(fx["batch"]))
    torch.testing.assert_close(out["og3d_logits"], g["og3d_logits"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["obj_cls_pre_logits"], g["obj_cls_pre_logits"], rtol=1e-4, atol=1e-4)
    assert torch.equal(out["og3d_logits"].argmax(-1), g["pred"])
    assert abs(float(Loss(model.cfg)(out)[0]) - g["og3d_loss"]) < 1e-4


def test_objcls_train_step_matches_reference(golden_cpu):
    """BASELINE config 1: ObjCls, unfrozen PointNet++ (train-mode BatchNorm), backward through
    SharedMLPs and group_points_grad."""
    g = golden_cpu["objcls"]
    cfg = ConfigNode({"num_gpu": 1, "solver": {"lr": 1e-3},
                      "model": {"name": "ObjCls", "model_name": "pointnet++", "language_type": "bert",
                                "open_vocab": False, "num_classes": 607, "cls_hidden": 1024}})
    model = MODEL_REGISTRY.get("ObjCls")(cfg).train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    fill_params(model, golden_cpu["seed"])
    with use_oracle_ext():
        out = model(clone_batch(g["batch"]))
        loss = obj_cls_loss(out)
        loss.backward()
    torch.testing.assert_close(out["obj_logits"], g["obj_logits"], rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - g["loss"]) < 1e-4
    params = dict(model.named_parameters())
    for name, ref in g["grads"].items():
        torch.testing.assert_close(params[name].grad, ref, rtol=2e-3, atol=1e-5 + 1e-4 * ref.abs().max().item(),
                                   msg=lambda m, n=name: f"{n}: {m}")


def test_registry_api_surface():
    from sceneverse_amd.modules import build as mb
    for reg, names in [(mb.VISION_REGISTRY, ["PointOpenVocabEncoder", "ObjClsEncoder"]),
                       (mb.LANGUAGE_REGISTRY, ["BERTLanguageEncoder", "CLIPLanguageEncoder"]),
                       (mb.GROUNDING_REGISTRY, ["EntitySpatialCrossEncoder", "UnifiedSpatialCrossEncoderV1",
                                                "UnifiedSpatialCrossEncoderV2"]),
                       (mb.HEADS_REGISTRY, ["GroundHeadV1", "GroundHead", "PretrainHeadV1", "OVPretrainHead",
                                            "QAHeadV1"]),
                       (MODEL_REGISTRY, ["ObjCls", "OpenVocab", "OpenVocabPerScene"])]:
        for n in names:
            assert n in reg and reg.get(n).__name__ == n
    try:
        mb.build_module("nope", ConfigNode({"name": "x", "args": {}}))
    except NotImplementedError:
        pass
    else:
        raise AssertionError("unknown module type must raise NotImplementedError")
