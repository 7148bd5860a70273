"""Shared helpers for the test-suite (imports oracle/ -- test infrastructure)."""
import os
import tempfile

import numpy as np
import torch

from sceneverse_amd.common.config import ConfigNode

N_CLS = 607


def text_features(seed):
    g = torch.Generator().manual_seed(1234 + seed)
    return 0.02 * torch.randn(N_CLS, 768, generator=g)


def lang_dir(seed):
    d = tempfile.mkdtemp()
    torch.save(text_features(seed), os.path.join(d, "scannet_607_bert-base-uncased_id.pth"))
    return d


def gps_cfg(lang_path, heads="pretrain", freeze=True, use_scene_cap=True, num_gpu=1):
    """all_pretrain.yaml:198-258 model section (heads='pretrain') or the ScanRefer fine-tune
    head (finetune/scanrefer_finetune.yaml:245-251, heads='ground')."""
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
        "num_gpu": num_gpu, "task": "Pretrain",
        "data": {"args": {"use_scene_cap": use_scene_cap}},
        "solver": {"lr": 5e-4, "optim": {"name": "AdamW", "args": {"betas": [0.9, 0.98]}},
                   "sched": {"name": "warmup_cosine", "args": {"warmup_steps": 500, "minimum_ratio": 0.1}}},
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


class use_oracle_ext:
    """Context manager: route sceneverse_amd's point ops to the CPU oracle (tests only), so the
    host-side model logic can be exercised on a machine without a GPU."""

    def __enter__(self):
        from oracle.pointnet2_oracle import OracleExt
        from sceneverse_amd.pointnet2 import pointnet2_utils
        self._mod, self._old = pointnet2_utils, pointnet2_utils._ext
        pointnet2_utils._ext = OracleExt
        return self

    def __exit__(self, *exc):
        self._mod._ext = self._old
        return False


def clone_batch(batch, device="cpu"):
    return {k: (v.clone().to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def _ulp_diff(a, b):
    return abs(int(np.float32(a).view(np.int32)) - int(np.float32(b).view(np.int32)))


def fps_divergence_is_rounding_tie(cloud, mine, theirs):
    """First round where two FPS index sequences differ: replay the pinned (individually rounded)
    running distances up to there and check that both candidates are within 2 ulp -- i.e. the
    disagreement is an FMA-contraction artefact on a geometrically exact tie."""
    j = int((mine != theirs).nonzero()[0])
    p = cloud.cpu().numpy()
    temp = np.full(p.shape[0], 1e10, dtype=np.float32)
    for old in mine[:j].tolist():
        d = p - p[old]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        temp = np.minimum(temp, d2)
    return _ulp_diff(temp[int(mine[j])], temp[int(theirs[j])]) <= 2
