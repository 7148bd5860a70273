"""Object-encoder pre-training model (reference model/objcls.py:16-97): PointNet++ over every
object, then a closed-set MLP head or open-vocabulary logits against fixed text embeddings.
Only the `model_name == "pointnet++"` branch of the reference is runnable with its shipped
PointNetPP; that is the branch implemented.  Building text embeddings on the
fly from CLIP/BERT weights (ref :45-62) needs downloads and is replaced by `pre_extract_path`."""
from pathlib import Path

import torch
import torch.nn as nn

from ..modules.layers.pointnet import PointNetPP
from ..modules.utils import get_mlp_head
from .build import MODEL_REGISTRY, BaseModel


@MODEL_REGISTRY.register()
class ObjCls(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.cfg = cfg
        self.model_name = cfg.model.get("model_name", "pointnext")
        self.language_type = cfg.model.get("language_type", "clip")
        self.pre_extract_path = cfg.model.get("pre_extract_path", None)
        width = 512 if self.language_type == "clip" else 768
        self.point_feature_extractor = PointNetPP(
            sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
            sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, width]])
        if cfg.num_gpu > 1:
            self.point_feature_extractor = nn.SyncBatchNorm.convert_sync_batchnorm(
                self.point_feature_extractor)
        if not cfg.model.open_vocab:
            self.cls_head = get_mlp_head(width, cfg.model.get("cls_hidden", 1024), cfg.model.num_classes)
        else:
            if self.pre_extract_path is None:
                raise NotImplementedError(
                    "open_vocab ObjCls needs model.pre_extract_path (pre-extracted 607-class text "
                    "embeddings); on-the-fly CLIP/BERT extraction requires network access")
            vocab = 'clip-ViT-B16' if self.language_type == 'clip' else 'bert-base-uncased'
            self.register_buffer(
                "text_embeds", torch.load(Path(self.pre_extract_path) / f"scannet_607_{vocab}_id.pth").float())
        self.dropout = nn.Dropout(0.1)

    def forward(self, data_dict):
        if 'cur_step' not in data_dict:
            data_dict['cur_step'] = 1
            data_dict['total_steps'] = 1
        obj_pcds = data_dict["obj_fts"]
        B, O = obj_pcds.shape[:2]
        if self.model_name not in ("pointnet++", "pointmlp"):
            raise NotImplementedError(f"ObjCls model_name={self.model_name!r}: only the PointNet++ "
                                      "encoder exists (as in the reference's shipped code)")
        emb = self.point_feature_extractor(obj_pcds.reshape(B * O, *obj_pcds.shape[2:]))
        emb = self.dropout(emb)
        logits = emb @ self.text_embeds.t() if self.cfg.model.open_vocab else self.cls_head(emb)
        data_dict["obj_logits"] = logits.view(B, O, -1)
        return data_dict

    def get_opt_params(self):
        return [{"params": self.parameters(),
                 "weight_decay": self.cfg.solver.get("weight_decay", 0.0),
                 "lr": self.cfg.solver.lr}]
