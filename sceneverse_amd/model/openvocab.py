"""GPS models (reference model/openvocab.py): `OpenVocab` (:11-126) and `OpenVocabPerScene`
(:129-255).  Same `data_dict` contract, same sub-module attribute names
(`lang_encoder`, `point_encoder`, `unified_encoder`, heads by config name), same optimiser groups.
Reference quirk kept: `scene_embed` averages all object slots, padding included (:24,53)."""
import torch

from ..modules.build import build_module
from ..optim.utils import no_decay_param_group
from .build import MODEL_REGISTRY, BaseModel


def _lr_of(node, default_lr):
    lr = node.get("lr")
    return default_lr if lr is None else lr


class _GPSBase(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.cfg = cfg
        self.lang_encoder = build_module("language", cfg.model.language)
        self.point_encoder = build_module("vision", cfg.model.vision)
        self.unified_encoder = build_module("grounding", cfg.model.grounding)
        self.head_list = cfg.model.heads.head_list
        for head in self.head_list:
            setattr(self, head, build_module("heads", getattr(cfg.model.heads, head)))

    def _encode_objects(self, data_dict):
        if "Scene" in self.cfg.model.vision.name:
            return self.point_encoder(data_dict)
        return self.point_encoder(data_dict['obj_fts'].float(), data_dict['obj_locs'],
                                  data_dict['obj_masks'], data_dict['obj_sem_masks'],
                                  data_dict['obj_labels'], data_dict['cur_step'],
                                  data_dict['total_steps'])

    def _pretrain_outputs(self, data_dict, txt_fused, obj_fused):
        if getattr(self, "pretrain_head", None) is None:
            return
        output = self.pretrain_head(txt_fused, obj_fused)
        if isinstance(output, tuple):
            data_dict['txt_lm_cls_logits'], data_dict['obj_cls_post_logits'] = output
        else:
            data_dict['txt_lm_cls_logits'] = output

    def get_opt_params(self):
        base_lr = self.cfg.solver.lr
        groups = []
        groups += no_decay_param_group(self.lang_encoder.named_parameters(),
                                       _lr_of(self.cfg.model.language, base_lr))
        groups += no_decay_param_group(self.point_encoder.named_parameters(),
                                       _lr_of(self.cfg.model.vision, base_lr))
        groups += no_decay_param_group(self.unified_encoder.named_parameters(),
                                       _lr_of(self.cfg.model.grounding, base_lr))
        for head in ("ground_head", "qa_head", "pretrain_head"):
            if head in self.head_list:
                groups += no_decay_param_group(getattr(self, head).named_parameters(),
                                               _lr_of(getattr(self.cfg.model.heads, head), base_lr))
        return groups


_OBJ_FIRST = False           # tests: run the object encoder before the text encoder (order of the bottom backward graph)


@MODEL_REGISTRY.register()
class OpenVocab(_GPSBase):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.use_scene_cap = cfg.data.args.get("use_scene_cap", False)
        if self.use_scene_cap:
            self.object_pool = lambda x: x.mean(dim=1)

    def _twin_generators(self, data_dict):
        """The text encoder and the object encoder as two generators of their GEMM calls (modules/layers/gemm.py
        drive_pair), or None when either side has no generator form / the inputs are not on the GPU / pairing is off.
        The two stacks are independent until `unified_encoder` (reference model/openvocab.py:41-63)."""
        from ..modules.layers import gemm
        lang, pts = self.lang_encoder, self.point_encoder
        if not (gemm.twin_stacks() and data_dict['txt_ids'].is_cuda and torch.is_autocast_enabled("cuda")
                and hasattr(lang, "forward_pair_gen") and hasattr(pts, "forward_gen")
                and "Scene" not in self.cfg.model.vision.name):
            return None
        if self.use_scene_cap:
            g_txt = lang.forward_pair_gen(data_dict['txt_ids'], data_dict['txt_masks'], data_dict['scene_txt_ids'],
                                          data_dict['scene_txt_masks'], cls_second=True)
        else:
            g_txt = lang.forward_gen(data_dict['txt_ids'], data_dict['txt_masks'])
        g_obj = pts.forward_gen(data_dict['obj_fts'].float(), data_dict['obj_locs'], data_dict['obj_masks'],
                                data_dict['obj_sem_masks'], data_dict['obj_labels'], data_dict['cur_step'],
                                data_dict['total_steps'])
        return g_txt, g_obj

    def forward(self, data_dict):
        if 'cur_step' not in data_dict:
            data_dict['cur_step'] = 1
            data_dict['total_steps'] = 1

        scene_txt = None
        pre = self._encode_objects(data_dict) if _OBJ_FIRST else None
        twin = self._twin_generators(data_dict) if pre is None else None
        if twin is not None:
            # the two bottom stacks in lock-step: their layers' GEMMs leave pairwise as single launches
            from ..modules.layers import gemm
            txt_out, pre = gemm.drive_pair(*twin)
            if self.use_scene_cap:
                txt, scene_txt = txt_out
                data_dict['scene_text_embed'] = scene_txt[:, 0]
            else:
                txt = txt_out
        elif self.use_scene_cap and hasattr(self.lang_encoder, "forward_pair"):
            # the sentence and the scene caption go through the text encoder's layers as one row batch
            txt, scene_txt = self.lang_encoder.forward_pair(data_dict['txt_ids'], data_dict['txt_masks'],
                                                            data_dict['scene_txt_ids'], data_dict['scene_txt_masks'],
                                                            cls_second=True)
            data_dict['scene_text_embed'] = scene_txt[:, 0]
        else:
            txt = self.lang_encoder(data_dict['txt_ids'], data_dict['txt_masks'])
            if self.use_scene_cap:
                scene_txt = self.lang_encoder(data_dict['scene_txt_ids'], data_dict['scene_txt_masks'])
                data_dict['scene_text_embed'] = scene_txt[:, 0]

        obj, obj_pre, obj_cls_raw = pre if pre is not None else self._encode_objects(data_dict)
        if torch.is_tensor(obj) and obj.requires_grad:
            # autograd EXECUTES the grad_fn of a non-leaf tensor named in backward(inputs=...) (it is how the gradient is
            # captured), and executes it again when the staged backward continues from that tensor: everything above reads
            # a trivial view of the encoder output, so that the node run twice is a ViewBackward and not the object
            # encoder's last LayerNorm (sceneverse_amd/engine.py cuts the backward pass at this tensor)
            obj = obj.view_as(obj)
        if self.use_scene_cap:
            data_dict["scene_embed"] = self.object_pool(obj)
        # outputs of the two bottom encoders (text, objects): where sceneverse_amd.engine cuts the backward pass of the
        # split-graph data-parallel step (gradients of everything above are exchanged while the part below still runs).
        # A valid cut holds no tensor that is an ancestor of another one: `obj_pre` feeds `obj`, so the cut exists only
        # when nothing above reads `obj_pre` (the grounding head does; the pre-train heads do not).
        reads_obj_pre = getattr(self, "ground_head", None) is not None
        self._stage_boundary = [] if reads_obj_pre else \
            [t for t in (txt, scene_txt, obj) if torch.is_tensor(t) and t.requires_grad]

        before = self.cfg.model.inter == "before"
        if before:
            data_dict["inter_text_embed"] = txt[:, 0]
            data_dict["inter_obj_embeds"] = obj

        txt_fused, obj_fused = self.unified_encoder(txt, data_dict['txt_masks'], obj,
                                                    data_dict['obj_locs'], data_dict['obj_masks'])
        if not before:
            data_dict["inter_text_embed"] = txt_fused[:, 0]
            data_dict["inter_obj_embeds"] = obj_fused

        cls_tok = txt_fused[:, 0]
        data_dict["intra_text_embed"] = cls_tok
        data_dict["intra_obj_embeds"] = obj_fused
        data_dict['obj_cls_raw_logits'] = obj_cls_raw
        data_dict['og3d_logits'] = torch.einsum('bod,bd->bo', obj_fused, cls_tok)

        if getattr(self, "ground_head", None) is not None:
            (data_dict['txt_cls_logits'], data_dict['obj_cls_post_logits'],
             data_dict['obj_cls_pre_logits'], data_dict['og3d_logits']) = self.ground_head(
                txt_fused, obj_fused, obj_pre, data_dict['obj_masks'])
        if getattr(self, "qa_head", None) is not None:
            data_dict['answer_scores'] = self.qa_head(obj_fused, data_dict['obj_masks'], txt_fused,
                                                      data_dict['txt_masks'])
        self._pretrain_outputs(data_dict, txt_fused, obj_fused)
        return data_dict


@MODEL_REGISTRY.register()
class OpenVocabPerScene(_GPSBase):
    """Variant fed L sentences per scene: txt_ids (B,L,T); objects are encoded once per scene and
    repeated per sentence (reference :141-232)."""

    def forward(self, data_dict):
        if 'cur_step' not in data_dict:
            data_dict['cur_step'] = 1
            data_dict['total_steps'] = 1
        per_scene = data_dict['txt_ids'].dim() == 3
        txt_ids, txt_masks = data_dict['txt_ids'], data_dict['txt_masks']
        if per_scene:
            B, L, _ = txt_ids.shape
            O = data_dict['obj_masks'].shape[1]
            txt_ids = txt_ids.view(B * L, -1)
            txt_masks = txt_masks.view(B * L, -1)

        txt = self.lang_encoder(txt_ids, txt_masks)
        obj, obj_pre, obj_cls_raw = self._encode_objects(data_dict)
        obj_locs, obj_masks = data_dict['obj_locs'], data_dict['obj_masks']
        if per_scene:
            obj = obj.unsqueeze(1).repeat(1, L, 1, 1).view(B * L, O, obj.shape[-1])
            obj_locs = obj_locs.unsqueeze(1).repeat(1, L, 1, 1).view(B * L, O, obj_locs.shape[-1])
            obj_masks = obj_masks.unsqueeze(1).repeat(1, L, 1).view(B * L, O)

        before = self.cfg.model.inter == "before"
        if before:
            data_dict["inter_text_embed"] = txt[:, 0]
            data_dict["inter_obj_embeds"] = obj
        txt_fused, obj_fused = self.unified_encoder(txt, txt_masks, obj, obj_locs, obj_masks)
        if not before:
            data_dict["inter_text_embed"] = txt_fused[:, 0]
            data_dict["inter_obj_embeds"] = obj_fused

        cls_tok = txt_fused[:, 0]
        data_dict["intra_text_embed"] = cls_tok
        data_dict["intra_obj_embeds"] = obj_fused
        data_dict['obj_cls_raw_logits'] = obj_cls_raw
        og3d = torch.einsum('bod,bd->bo', obj_fused, cls_tok)
        data_dict['og3d_logits'] = og3d.view(B, L, O) if per_scene else og3d

        if getattr(self, "qa_head", None) is not None:
            data_dict['answer_scores'] = self.qa_head(obj_fused, data_dict['obj_masks'], txt_fused,
                                                      data_dict['txt_masks'])
        self._pretrain_outputs(data_dict, txt_fused, obj_fused)
        return data_dict
