"""Loss functions over `data_dict` and the list-of-losses container (reference
optim/loss/loss.py).  Module-level functions are looked up by name exactly like the reference's
`globals()` dispatch (:121-129); registered nn.Module losses come from LOSS_REGISTRY."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...common.registry import Registry

LOSS_REGISTRY = Registry("loss")


def _masked_obj_ce(logits_key, data_dict, weight_fn, **ce_kwargs):
    w = weight_fn(data_dict)
    ce = F.cross_entropy(data_dict[logits_key].permute(0, 2, 1), data_dict["obj_labels"],
                         reduction='none', **ce_kwargs)
    return (ce * w).sum() / w.sum()


def _w_all(d):
    return d["obj_masks"]


def _w_masked(d):
    return d["obj_masks"] * d["obj_sem_masks"].logical_not()


def _w_unmasked(d):
    return d["obj_masks"] * d["obj_sem_masks"]


def og3d_loss(data_dict):                                   # ref :8-9
    return F.cross_entropy(data_dict["og3d_logits"], data_dict["tgt_object_id"].squeeze(1))


def og3d_multi_loss(data_dict):                             # ref :12-16
    tgt = data_dict["tgt_object_id"]
    return F.binary_cross_entropy_with_logits(data_dict["og3d_logits"], tgt.float(),
                                              reduction="sum") / float(tgt.shape[0])


def txt_cls_multi_loss(data_dict):                          # ref :19-23
    tgt = data_dict["tgt_object_label"]
    return F.binary_cross_entropy_with_logits(data_dict["txt_cls_logits"], tgt.float(),
                                              reduction='sum') / float(tgt.shape[0])


def obj_cls_raw_loss(data_dict):                            # ref :26-31
    return _masked_obj_ce("obj_cls_raw_logits", data_dict, _w_all)


def obj_cls_pre_loss(data_dict):                            # ref :34-39
    return _masked_obj_ce("obj_cls_pre_logits", data_dict, _w_all)


def obj_cls_post_loss(data_dict):                           # ref :42-47
    return _masked_obj_ce("obj_cls_post_logits", data_dict, _w_all)


def answer_loss(data_dict):                                 # ref :50-53
    return F.binary_cross_entropy_with_logits(
        data_dict["answer_scores"], data_dict["answer_label"].float(),
        reduction='sum') / data_dict["answer_scores"].shape[0]


def lm_cls_loss(data_dict):                                 # ref :56-61
    labels = data_dict["masked_lm_labels"]
    if labels.dim() == 3:
        labels = labels.view(-1, labels.size(-1))
    logits = data_dict["txt_lm_cls_logits"]
    if hasattr(logits, "materialize"):          # LazyLMLogits: loss over the labelled rows, no (B, L, vocab) tensor
        return logits.loss(labels, ignore_index=-1)
    if logits.is_cuda and logits.dtype in (torch.bfloat16, torch.float32):
        # row-sparse kernel: ignored positions (> 90 % of the rows) are never read
        from .masked_ce import masked_cross_entropy
        return masked_cross_entropy(logits, labels, ignore_index=-1)
    return F.cross_entropy(logits.permute(0, 2, 1), labels, ignore_index=-1)


def obj_cls_pre_loss_mask(data_dict):                       # ref :64-69
    return _masked_obj_ce("obj_cls_pre_logits", data_dict, _w_masked)


def obj_cls_pre_loss_unmask(data_dict):                     # ref :72-77
    return _masked_obj_ce("obj_cls_pre_logits", data_dict, _w_unmasked)


def obj_cls_post_loss_mask(data_dict):                      # ref :80-85
    return _masked_obj_ce("obj_cls_post_logits", data_dict, _w_masked)


def obj_cls_post_loss_unmask(data_dict):                    # ref :88-93
    return _masked_obj_ce("obj_cls_post_logits", data_dict, _w_unmasked)


def obj_cls_loss(data_dict, smoothing=0.3):                 # ref :96-102
    return _masked_obj_ce("obj_logits", data_dict, _w_all, label_smoothing=smoothing)


def mse_loss(data_dict):                                    # ref :105-108
    return ((data_dict["pred_images"] - data_dict["target_images"]) ** 2).mean()


class Loss(nn.Module):
    """Evaluates every loss in vis_loss_list + loss_list, sums the ones in loss_list
    (reference :111-148).  forward -> (total_loss, dict of all losses incl. 'total_loss')."""

    def __init__(self, cfg):
        super().__init__()
        self.all_keys = list(set(cfg.model.vis_loss_list + cfg.model.loss_list))
        self.selected_keys = cfg.model.loss_list
        self.loss_fn = {}
        for k in self.all_keys:
            if k in globals():
                self.loss_fn[k] = globals()[k]
            else:
                self.loss_fn[k] = LOSS_REGISTRY.get(k)(cfg)
                setattr(self, k, self.loss_fn[k])  # register: parameters follow .to(device)

    def forward(self, data_dict):
        all_losses = {}
        for k, fn in self.loss_fn.items():
            if k == 'txt_cls_loss' and 'txt_cls_label' not in data_dict:
                data_dict['txt_cls_label'] = data_dict["tgt_object_label"].squeeze(1)
            cur = fn(data_dict)
            if isinstance(cur, list):
                all_losses[k] = cur[0]
                for ck, cv in cur[1].items():
                    all_losses[k + "_" + ck] = cv
            else:
                all_losses[k] = cur
        total_loss = sum(all_losses[k] for k in self.selected_keys)
        all_losses["total_loss"] = total_loss
        return total_loss, all_losses
