"""fp32 PyTorch restatement of the floating-point part of the GPS hot path.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  Functional style:
every function takes a reference-layout `state_dict` (+ key prefix) and plain tensors, so it can
check ANY implementation that keeps the reference's parameter names.  Point ops come from
oracle/pointnet2_oracle.py (the C restatement); everything runs on CPU in fp32.

Reference files restated (all under /root/reference/):
  modules/third_party/pointnet2/pytorch_utils.py:11-36,87-120   SharedMLP = (conv1x1 no-bias, BN, ReLU)*
  modules/third_party/pointnet2/pointnet2_utils.py:314-373,389-419  QueryAndGroup / GroupAll
  modules/third_party/pointnet2/pointnet2_modules.py:34-75      SA level: FPS, group, MLP, max
  modules/layers/pointnet.py:6-63                               PointNetPP
  modules/utils.py:38-87                                        calc_pairwise_locs
  modules/layers/transformers.py:188-239,301-316                spatial attention / encoder layer
  modules/layers/transformers.py:115-154                        joint encoder layer (nn.MultiheadAttention)
  modules/vision/pcd_openvocab_encoder.py:126-184               PointOpenVocabEncoder.forward
  modules/grounding/unified_encoder.py:147-177                  UnifiedSpatialCrossEncoderV2.forward
  modules/heads/pretrain_head.py:8-56, grounding_head.py:29-39  heads
  model/openvocab.py:26-101                                     OpenVocab.forward
  optim/loss/loss.py:8-9,56-61, contra_loss.py:11-98            losses
Dropout is the identity here (eval semantics): parity is defined with dropout off.

Pinning: this restatement is checked against outputs of the reference's own Python, imported from
/root/reference and run on CPU (tests/golden/make_golden.py -> tests/golden/gps_reference_cpu.pt,
tests/test_oracle_vs_golden.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .pointnet2_oracle import OracleExt

BN_EPS = 1e-5   # nn.BatchNorm2d default
LN_EPS = 1e-5   # nn.LayerNorm default


# --------------------------------------------------------------------------------------------
# point-op wrappers with autograd where the reference has it
# --------------------------------------------------------------------------------------------
class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx):
        ctx.idx, ctx.n = idx, feats.size(2)
        return OracleExt.group_points(feats.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        return OracleExt.group_points_grad(g.contiguous(), ctx.idx, ctx.n), None


def _ln(sd, p, x, eps=LN_EPS):
    return F.layer_norm(x, (x.size(-1),), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# --------------------------------------------------------------------------------------------
# PointNet++
# --------------------------------------------------------------------------------------------
def shared_mlp(sd, prefix, x, bn_training=False):
    """x (B,C,np,ns).  Layers `prefix.layer{i}.conv.weight` (Cout,Cin,1,1) + `.bn.bn.*`."""
    i = 0
    while f"{prefix}.layer{i}.conv.weight" in sd:
        w = sd[f"{prefix}.layer{i}.conv.weight"]
        x = torch.einsum('oc,bcps->bops', w[:, :, 0, 0], x)
        bn = f"{prefix}.layer{i}.bn.bn"
        if bn_training:
            mean = x.mean(dim=(0, 2, 3), keepdim=True)
            var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        else:
            mean = sd[bn + ".running_mean"].view(1, -1, 1, 1)
            var = sd[bn + ".running_var"].view(1, -1, 1, 1)
        x = (x - mean) / torch.sqrt(var + BN_EPS) * sd[bn + ".weight"].view(1, -1, 1, 1) \
            + sd[bn + ".bias"].view(1, -1, 1, 1)
        x = torch.relu(x)
        i += 1
    return x


def sa_level(sd, prefix, xyz, feats, npoint, radius, nsample, bn_training=False):
    """One set-abstraction level.  xyz (B,N,3), feats (B,C,N) or None ->
    (new_xyz (B,npoint,3) | None, new_feats (B,C',npoint|1))."""
    if npoint is not None:
        fps = OracleExt.furthest_point_sampling(xyz.contiguous(), npoint)
        xyz_t = xyz.transpose(1, 2).contiguous()
        new_xyz = OracleExt.gather_points(xyz_t, fps).transpose(1, 2).contiguous()
        idx = OracleExt.ball_query(new_xyz, xyz.contiguous(), radius, nsample)
        grouped_xyz = OracleExt.group_points(xyz_t, idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if feats is not None:
            grouped = torch.cat([grouped_xyz, _Group.apply(feats, idx)], dim=1)
        else:
            grouped = grouped_xyz
    else:
        new_xyz = None
        grouped = xyz.transpose(1, 2).unsqueeze(2)
        if feats is not None:
            grouped = torch.cat([grouped, feats.unsqueeze(2)], dim=1)
    out = shared_mlp(sd, f"{prefix}.mlps.0", grouped, bn_training)
    return new_xyz, out.max(dim=3)[0]


GPS_SA = dict(npoints=[32, 16, None], nsamples=[32, 32, None], radii=[0.2, 0.4, None])


def pointnetpp(sd, prefix, pcs, bn_training=False, sa=GPS_SA):
    """pcs (b, P, 3+C) -> (b, width)."""
    xyz = pcs[..., :3].contiguous()
    feats = pcs[..., 3:].transpose(1, 2).contiguous() if pcs.size(-1) > 3 else None
    for i, (npnt, ns, r) in enumerate(zip(sa["npoints"], sa["nsamples"], sa["radii"])):
        xyz_next, feats = sa_level(sd, f"{prefix}.encoder.{i}", xyz, feats, npnt, r, ns, bn_training)
        if xyz_next is not None:
            xyz = xyz_next
    return _lin(sd, f"{prefix}.fc", feats.reshape(feats.size(0), -1))


# --------------------------------------------------------------------------------------------
# geometry + attention layers
# --------------------------------------------------------------------------------------------
def calc_pairwise_locs(centers, eps=1e-10):
    """(B,L,3) -> (B,L,L,5): [d/dmax, dz/d, dxy/d, dy/dxy, dx/dxy], 'center' type, normalised."""
    diff = centers[:, :, None, :] - centers[:, None, :, :]
    d = torch.sqrt((diff ** 2).sum(3) + eps)
    dmax = d.reshape(d.size(0), -1).max(dim=1)[0]
    dxy = torch.sqrt((diff[..., :2] ** 2).sum(3) + eps)
    return torch.stack([d / dmax[:, None, None], diff[..., 2] / d, dxy / d,
                        diff[..., 1] / dxy, diff[..., 0] / dxy], dim=3)


def spatial_attention(sd, prefix, x, pairwise, key_pad, n_head):
    """MultiHeadAttentionSpatial, fusion 'cond'.  x (B,L,D); returns (out (B,L,D), probs (H,B,L,L))."""
    B, L, D = x.shape
    dh = D // n_head

    def heads(t):
        return t.view(B, L, n_head, dh).permute(2, 0, 1, 3)          # (H,B,L,dh)

    q, k, v = (heads(_lin(sd, f"{prefix}.{n}", x)) for n in ("w_qs", "w_ks", "w_vs"))
    attn = torch.einsum('hblk,hbtk->hblt', q, k) / math.sqrt(dh)
    sw = _lin(sd, f"{prefix}.lang_cond_fc", x).view(B, L, n_head, 6).permute(2, 0, 1, 3)
    loc = torch.sigmoid(torch.einsum('hbld,bltd->hblt', sw[..., 1:], pairwise) + sw[..., :1])
    if key_pad is not None:
        m = key_pad[None, :, None, :].expand(n_head, B, L, L)
        attn = attn.masked_fill(m, float('-inf'))
        loc = loc.masked_fill(m, 0)
    probs = torch.softmax(torch.log(torch.clamp(loc, min=1e-6)) + attn, dim=3)
    out = torch.einsum('hblt,hbtv->hblv', probs, v).permute(1, 2, 0, 3).reshape(B, L, D)
    return _lin(sd, f"{prefix}.fc", out), probs


def _ffn(sd, prefix, x, act):
    return _lin(sd, f"{prefix}.linear2", act(_lin(sd, f"{prefix}.linear1", x)))


def spatial_encoder_layer(sd, prefix, x, pairwise, key_pad, n_head):
    """TransformerSpatialEncoderLayer: post-norm, GELU(erf) FFN."""
    a, probs = spatial_attention(sd, f"{prefix}.self_attn", x, pairwise, key_pad, n_head)
    x = _ln(sd, f"{prefix}.norm1", x + a)
    x = _ln(sd, f"{prefix}.norm2", x + _ffn(sd, prefix, x, F.gelu))
    return x, probs


def mha_self_attention(sd, prefix, x, key_pad, n_head):
    """nn.MultiheadAttention(batch_first) self-attention with key_padding_mask; returns
    (out, head-averaged probs (B,T,T))."""
    B, T, D = x.shape
    dh = D // n_head
    qkv = F.linear(x, sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"])
    q, k, v = (t.view(B, T, n_head, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    scores = (q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2)       # torch scales q first
    if key_pad is not None:
        scores = scores.masked_fill(key_pad[:, None, None, :], float('-inf'))
    probs = torch.softmax(scores, dim=-1)
    out = (probs @ v).transpose(1, 2).reshape(B, T, D)
    return _lin(sd, f"{prefix}.out_proj", out), probs.mean(dim=1)


def joint_encoder_layer(sd, prefix, x, key_pad, n_head):
    """TransformerEncoderLayer as V2 uses it: post-norm, ReLU FFN."""
    a, probs = mha_self_attention(sd, f"{prefix}.self_attn", x, key_pad, n_head)
    x = _ln(sd, f"{prefix}.norm1", x + a)
    x = _ln(sd, f"{prefix}.norm2", x + _ffn(sd, prefix, x, F.relu))
    return x, probs


def _loc_embed(sd, prefix, obj_locs):
    return _ln(sd, f"{prefix}.loc_layers.0.1", _lin(sd, f"{prefix}.loc_layers.0.0", obj_locs))


def point_open_vocab_encoder(sd, prefix, obj_pcds, obj_locs, obj_masks, n_head=12, n_layers=4,
                             bn_training=False):
    """-> (obj_embeds (B,O,D), obj_embeds_pre (B,O,D), obj_sem_cls (B,O,607))."""
    B, O = obj_pcds.shape[:2]
    emb = pointnetpp(sd, f"{prefix}.point_feature_extractor",
                     obj_pcds.reshape(B * O, *obj_pcds.shape[2:]), bn_training).view(B, O, -1)
    sem = torch.softmax(emb @ sd[f"{prefix}.text_features"].t(), dim=2)
    pre = emb
    pairwise = calc_pairwise_locs(obj_locs[:, :, :3])
    pad = obj_masks.logical_not()
    for i in range(n_layers):
        emb = emb + _loc_embed(sd, prefix, obj_locs)
        emb, _ = spatial_encoder_layer(sd, f"{prefix}.spatial_encoder.{i}", emb, pairwise, pad, n_head)
    return emb, pre, sem


def unified_encoder_v2(sd, prefix, txt, txt_masks, obj, obj_locs, obj_masks, n_head=12, n_layers=4):
    Lt, Lo = txt.shape[1], obj.shape[1]
    pad = torch.cat([txt_masks.bool(), obj_masks.bool()], dim=1).logical_not()
    tt = sd[f"{prefix}.token_type_embeddings.weight"]
    for i in range(n_layers):
        obj = obj + _loc_embed(sd, prefix, obj_locs) + tt[1]
        txt = txt + tt[0]
        joint, _ = joint_encoder_layer(sd, f"{prefix}.unified_encoder.{i}",
                                       torch.cat([txt, obj], dim=1), pad, n_head)
        txt, obj = joint[:, :Lt], joint[:, Lt:Lt + Lo]
    return txt, obj


def bert_lm_head(sd, prefix, x):
    h = _ln(sd, f"{prefix}.transform.LayerNorm", F.gelu(_lin(sd, f"{prefix}.transform.dense", x)))
    return F.linear(h, sd[f"{prefix}.decoder.weight"]) + sd[f"{prefix}.bias"]


def mlp_head(sd, prefix, x):
    h = torch.relu(_lin(sd, f"{prefix}.0", x))
    h = F.layer_norm(h, (h.size(-1),), sd[f"{prefix}.2.weight"], sd[f"{prefix}.2.bias"], 1e-12)
    return _lin(sd, f"{prefix}.4", h)


# --------------------------------------------------------------------------------------------
# whole model + losses
# --------------------------------------------------------------------------------------------
def openvocab_forward(sd, data, lang_encoder, use_scene_cap=True, heads=("pretrain_head",),
                      inter="before"):
    """OpenVocab.forward in eval semantics.  `lang_encoder(ids, masks) -> (B,L,D)` is the
    HuggingFace BERT (out-of-scope arithmetic, shared with the implementation under test)."""
    out = {}
    txt = lang_encoder(data['txt_ids'], data['txt_masks'])
    if use_scene_cap:
        out['scene_text_embed'] = lang_encoder(data['scene_txt_ids'], data['scene_txt_masks'])[:, 0]
    obj, pre, sem = point_open_vocab_encoder(sd, "point_encoder", data['obj_fts'].float(),
                                             data['obj_locs'], data['obj_masks'])
    if use_scene_cap:
        out['scene_embed'] = obj.mean(dim=1)
    if inter == "before":
        out['inter_text_embed'], out['inter_obj_embeds'] = txt[:, 0], obj
    txt_f, obj_f = unified_encoder_v2(sd, "unified_encoder", txt, data['txt_masks'], obj,
                                      data['obj_locs'], data['obj_masks'])
    if inter != "before":
        out['inter_text_embed'], out['inter_obj_embeds'] = txt_f[:, 0], obj_f
    out['intra_text_embed'], out['intra_obj_embeds'] = txt_f[:, 0], obj_f
    out['obj_cls_raw_logits'] = sem
    out['og3d_logits'] = torch.einsum('bod,bd->bo', obj_f, txt_f[:, 0])
    if "ground_head" in heads:
        og = mlp_head(sd, "ground_head.og3d_head", obj_f).squeeze(2)
        out['og3d_logits'] = og.masked_fill(data['obj_masks'].logical_not(), float('-inf'))
        out['txt_cls_logits'] = mlp_head(sd, "ground_head.txt_clf_head", txt_f[:, 0])
        out['obj_cls_post_logits'] = mlp_head(sd, "ground_head.obj3d_clf_head", obj_f)
        out['obj_cls_pre_logits'] = mlp_head(sd, "ground_head.obj3d_clf_pre_head", pre)
    if "pretrain_head" in heads:
        out['txt_lm_cls_logits'] = bert_lm_head(sd, "pretrain_head.lm_pred_head", txt_f)
        if "pretrain_head.obj_pred_head.decoder.weight" in sd:
            out['obj_cls_post_logits'] = bert_lm_head(sd, "pretrain_head.obj_pred_head", obj_f)
    return out


def lm_cls_loss(out, data):
    return F.cross_entropy(out['txt_lm_cls_logits'].permute(0, 2, 1), data['masked_lm_labels'],
                           ignore_index=-1)


def og3d_loss(out, data):
    return F.cross_entropy(out['og3d_logits'], data['tgt_object_id'].squeeze(1))


def text_obj_within_batch(out, data):
    o = F.normalize(out['intra_obj_embeds'], dim=-1, p=2)
    t = F.normalize(out['intra_text_embed'], dim=-1, p=2)
    logits = torch.einsum('bod,bd->bo', o, t).masked_fill(data['obj_masks'].logical_not(), float('-inf'))
    return F.cross_entropy(logits, data['tgt_object_id'].squeeze(-1))


def _clip_loss(a, b, scale):
    lab = torch.arange(a.shape[0])
    return (F.cross_entropy(scale * a @ b.t(), lab) + F.cross_entropy(scale * b @ a.t(), lab)) / 2


def text_scene_between_batch(out, logit_scale):
    s = F.normalize(out['scene_embed'], dim=-1, p=2)
    t = F.normalize(out['scene_text_embed'], dim=-1, p=2)
    return _clip_loss(t, s, torch.clamp(logit_scale, max=100))


def pretrain_losses(out, data, logit_scale):
    losses = {
        'lm_cls_loss': lm_cls_loss(out, data),
        'TextObjWithinBatch': text_obj_within_batch(out, data),
        'TextSceneBetweenBatch': text_scene_between_batch(out, logit_scale),
    }
    losses['total_loss'] = sum(losses.values())
    return losses
