"""Generate tests/golden/gps_reference_cpu.pt by running the REFERENCE's own Python
(/root/reference, imported unmodified) on CPU.

Only runnable where /root/reference exists (the build container); the produced fixture is
committed and is what travels.  Recipe = SURVEY.md App. G:
  * stub packages (tests/golden/ref_stubs) for fvcore / omegaconf / wandb / hydra / jsonlines /
    clip / submitit, none of which is installable here;
  * builtins.__POINTNET2_SETUP__ so pointnet2_utils imports without its CUDA extension, then the
    CPU oracle (oracle/pointnet2_oracle.py) injected as `pointnet2_utils._ext` -- the reference's
    native ops assert "CPU not supported", so its Python can only run on CPU on top of the oracle;
    the native ops themselves are pinned separately on the GPU (tests/golden/point_ops_ref_gpu.pt);
  * torch.Tensor.cuda -> identity (unified_encoder.py:157,162 hard-codes .cuda());
  * BERTLanguageEncoder swapped for an offline random-init BertModel of the same config.
Weights are NOT stored: oracle/param_fill.fill_params derives them from parameter names.

    python tests/golden/make_golden.py
"""
import builtins
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
builtins.__POINTNET2_SETUP__ = True

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self

from oracle.param_fill import fill_params  # noqa: E402
from oracle.pointnet2_oracle import OracleExt  # noqa: E402
from sceneverse_amd.common.config import ConfigNode  # noqa: E402
from sceneverse_amd.data.synthetic import adversarial_objects, synth_batch  # noqa: E402

SEED = 3
N_CLS = 607


def import_reference():
    import modules.third_party.pointnet2.pointnet2_modules  # noqa: F401  (appends its dir to sys.path)
    import pointnet2_utils  # the top-level alias the SA modules really use
    pointnet2_utils._ext = OracleExt
    import modules.build as mb
    from transformers import BertConfig, BertModel

    class OfflineBert(nn.Module):
        def __init__(self, cfg, weights=None, hidden_size=768, num_hidden_layers=4,
                     num_attention_heads=12, type_vocab_size=2, **kw):
            super().__init__()
            self.model = BertModel(BertConfig(hidden_size=hidden_size,
                                              num_hidden_layers=num_hidden_layers,
                                              num_attention_heads=num_attention_heads,
                                              type_vocab_size=type_vocab_size))

        def forward(self, txt_ids, txt_masks, **kw):
            return self.model(txt_ids, txt_masks).last_hidden_state

    import modules  # noqa: F401  registers everything
    mb.LANGUAGE_REGISTRY._obj_map['BERTLanguageEncoder'] = OfflineBert
    import model  # noqa: F401
    import optim.loss.contra_loss  # noqa: F401
    return mb


def gps_cfg(lang_path, heads="pretrain", freeze=True, use_scene_cap=True):
    head_cfg = {
        "pretrain": {"head_list": ["pretrain_head"],
                     "pretrain_head": {"name": "OVPretrainHead",
                                       "args": {"hidden_size": 768, "vocab_size": 30522}}},
        "ground": {"head_list": ["ground_head"],
                   "ground_head": {"name": "GroundHeadV1",
                                   "args": {"hidden_size": 384, "input_size": 768,
                                            "sem_cls_size": 607, "dropout": 0.3,
                                            "detach_all_aux_loss": True}}},
    }[heads]
    losses = (["lm_cls_loss", "TextObjWithinBatch", "TextSceneBetweenBatch"] if heads == "pretrain"
              else ["og3d_loss"])
    return ConfigNode({
        "num_gpu": 1, "task": "Pretrain",
        "data": {"args": {"use_scene_cap": use_scene_cap}},
        "solver": {"lr": 5e-4},
        "model": {
            "name": "OpenVocab",
            "language": {"name": "BERTLanguageEncoder",
                         "args": {"weights": None, "hidden_size": 768, "num_hidden_layers": 4,
                                  "num_attention_heads": 12, "type_vocab_size": 2}, "lr": 1e-5},
            "vision": {"name": "PointOpenVocabEncoder",
                       "args": {"backbone": "pointnet++", "hidden_size": 768, "freeze": freeze,
                                "path": None, "num_attention_heads": 12, "spatial_dim": 5,
                                "num_layers": 4, "dim_loc": 6, "dim_feedforward": 2048,
                                "attn_type": "spatial", "pairwise_rel_type": "center",
                                "use_matmul_label": False, "lang_type": "bert",
                                "lang_path": lang_path}, "lr": 1e-4},
            "grounding": {"name": "UnifiedSpatialCrossEncoderV2",
                          "args": {"hidden_size": 768, "num_attention_heads": 12, "num_layers": 4,
                                   "dim_feedforward": 2048, "dim_loc": 6}, "lr": 1e-4},
            "inter": "before",
            "heads": head_cfg,
            "loss_list": losses, "vis_loss_list": losses,
        },
    })


def text_features(seed=SEED):
    g = torch.Generator().manual_seed(1234 + seed)
    return 0.02 * torch.randn(N_CLS, 768, generator=g)


def grad_summary(model, names):
    out = {}
    params = dict(model.named_parameters())
    for n in names:
        g = params[n].grad
        out[n] = {"norm": g.norm().item(), "head": g.flatten()[:256].clone()}
    return out


GRAD_NAMES = [
    "unified_encoder.unified_encoder.3.self_attn.in_proj_weight",
    "unified_encoder.unified_encoder.0.linear1.weight",
    "unified_encoder.loc_layers.0.0.weight",
    "unified_encoder.token_type_embeddings.weight",
    "point_encoder.spatial_encoder.0.self_attn.lang_cond_fc.weight",
    "point_encoder.spatial_encoder.0.self_attn.w_qs.weight",
    "point_encoder.spatial_encoder.3.linear2.bias",
    "point_encoder.loc_layers.0.0.weight",
    "lang_encoder.model.embeddings.word_embeddings.weight",
]


def main():
    mb = import_reference()
    from model.build import build_model
    from modules.layers.pointnet import PointNetPP
    from modules.layers.transformers import TransformerEncoderLayer, TransformerSpatialEncoderLayer
    from modules.utils import calc_pairwise_locs
    from optim.loss.loss import Loss

    fx = {"seed": SEED}
    tmp = tempfile.mkdtemp()
    torch.save(text_features(), os.path.join(tmp, "scannet_607_bert-base-uncased_id.pth"))

    # ---- 1. PointNet++ encoder on adversarial + synthetic objects (eval BN) -------------------
    torch.manual_seed(0)
    pn = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None],
                    sa_radii=[0.2, 0.4, None],
                    sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]]).eval()
    fill_params(pn, SEED)
    adv = adversarial_objects()
    g = torch.Generator().manual_seed(11)
    adv_pcs = torch.cat([adv, torch.rand(adv.shape, generator=g) * 2 - 1], dim=2)   # + rgb
    with torch.no_grad():
        fx["pointnetpp_adv_in"] = adv_pcs
        fx["pointnetpp_adv_out"] = pn(adv_pcs.clone())

    # ---- 2. geometry + single layers ----------------------------------------------------------
    batch = synth_batch(2, n_obj=6, n_pts=1024, seed=SEED, min_real=3)
    fx["batch"] = batch
    with torch.no_grad():
        fx["pairwise_locs"] = calc_pairwise_locs(batch["obj_locs"][:, :, :3], batch["obj_locs"][:, :, 3:],
                                                 pairwise_rel_type='center', spatial_dist_norm=True,
                                                 spatial_dim=5)
        sl = TransformerSpatialEncoderLayer(768, 12, dim_feedforward=2048, dropout=0.1,
                                            activation='gelu', spatial_dim=5, spatial_multihead=True,
                                            spatial_attn_fusion='cond').eval()
        fill_params(sl, SEED)
        gx = torch.Generator().manual_seed(5)
        x = torch.randn(2, 6, 768, generator=gx)
        fx["spatial_layer_in"] = x
        y, probs = sl(x, fx["pairwise_locs"], tgt_key_padding_mask=batch["obj_masks"].logical_not())
        fx["spatial_layer_out"], fx["spatial_layer_probs"] = y, probs
        jl = TransformerEncoderLayer(768, 12, dim_feedforward=2048).eval()
        fill_params(jl, SEED)
        xj = torch.randn(2, 11, 768, generator=gx)
        padj = torch.zeros(2, 11, dtype=torch.bool)
        padj[0, 4:6] = True
        padj[1, 9:] = True
        fx["joint_layer_in"], fx["joint_layer_pad"] = xj, padj
        yj, pj = jl(xj, tgt_key_padding_mask=padj)
        fx["joint_layer_out"], fx["joint_layer_probs"] = yj, pj

    # ---- 3. full GPS pre-train model, eval semantics, forward + loss + backward ---------------
    cfg = gps_cfg(tmp, heads="pretrain")
    model = build_model(cfg).eval()
    loss_mod = Loss(cfg)
    fill_params(model, SEED)
    data = {k: v.clone() for k, v in batch.items()}
    out = model(data)
    total, losses = loss_mod(out)
    total.backward()
    fx["gps_pretrain"] = {
        "og3d_logits": out["og3d_logits"].detach(),
        "intra_text_embed": out["intra_text_embed"].detach(),
        "intra_obj_embeds": out["intra_obj_embeds"].detach(),
        "inter_obj_embeds": out["inter_obj_embeds"].detach(),
        "scene_embed": out["scene_embed"].detach(),
        "scene_text_embed": out["scene_text_embed"].detach(),
        "obj_cls_raw_logits_top": out["obj_cls_raw_logits"].detach()[:, :, :32].clone(),
        "txt_lm_cls_logits_lse": torch.logsumexp(out["txt_lm_cls_logits"].detach(), dim=2),
        "txt_lm_cls_logits_head": out["txt_lm_cls_logits"].detach()[:, :, :64].clone(),
        "obj_cls_post_logits": out["obj_cls_post_logits"].detach(),
        "losses": {k: float(v) for k, v in losses.items()},
        "grads": grad_summary(model, GRAD_NAMES),
        "n_trainable": sum(p.numel() for p in model.parameters() if p.requires_grad),
        "n_opt_groups": len(model.get_opt_params()),
        "state_dict_keys": sorted(model.state_dict().keys()),
    }

    # ---- 4. grounding fine-tune head (ScanRefer config: GroundHeadV1 hidden 384, og3d_loss) ----
    cfg_g = gps_cfg(tmp, heads="ground", use_scene_cap=False)
    model_g = build_model(cfg_g).eval()
    fill_params(model_g, SEED)
    data = {k: v.clone() for k, v in batch.items()}
    with torch.no_grad():
        out = model_g(data)
    fx["gps_ground"] = {
        "og3d_logits": out["og3d_logits"],
        "txt_cls_logits": out["txt_cls_logits"],
        "obj_cls_pre_logits": out["obj_cls_pre_logits"],
        "og3d_loss": float(Loss(cfg_g)(out)[0]),
        "pred": torch.argmax(out["og3d_logits"], dim=-1),
    }

    # ---- 5. ObjCls (BASELINE config 1): unfrozen PointNet++, train-mode BN, dropout p=0 --------
    from model.objcls import ObjCls
    cfg_o = ConfigNode({"num_gpu": 1, "solver": {"lr": 1e-3},
                        "model": {"name": "ObjCls", "model_name": "pointnet++",
                                  "language_type": "bert", "open_vocab": False,
                                  "num_classes": N_CLS, "cls_hidden": 1024}})
    oc = ObjCls(cfg_o).train()
    for m in oc.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    fill_params(oc, SEED)
    ob = synth_batch(1, n_obj=8, n_pts=1024, seed=SEED + 1, min_real=6)
    d = {"obj_fts": ob["obj_fts"].clone(), "obj_labels": ob["obj_labels"].clamp(min=0),
         "obj_masks": ob["obj_masks"]}
    from optim.loss.loss import obj_cls_loss
    out = oc(d)
    loss = obj_cls_loss(out)
    loss.backward()
    pg = dict(oc.named_parameters())
    fx["objcls"] = {
        "batch": {k: v for k, v in d.items() if k != "obj_logits"},
        "obj_logits": out["obj_logits"].detach(),
        "loss": float(loss),
        "grads": {n: pg[n].grad.clone() for n in [
            "point_feature_extractor.encoder.0.mlps.0.layer0.conv.weight",
            "point_feature_extractor.encoder.1.mlps.0.layer0.conv.weight",
            "point_feature_extractor.encoder.2.mlps.0.layer2.bn.bn.weight",
            "point_feature_extractor.fc.bias"]},
    }

    dst = os.path.join(HERE, "gps_reference_cpu.pt")
    torch.save(fx, dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")
    print("losses", fx["gps_pretrain"]["losses"], "trainable", fx["gps_pretrain"]["n_trainable"])
    print("ground", fx["gps_ground"]["og3d_loss"], fx["gps_ground"]["pred"])
    print("objcls loss", fx["objcls"]["loss"])


if __name__ == "__main__":
    main()
