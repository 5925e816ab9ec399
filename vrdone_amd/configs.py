"""model_config / inference_config of the reference's four shipped YAML files
(configs/vidvrd.yaml, vidor.yaml, vidor_local.yaml, vidor_x.yaml), as Python dicts so the bench
and smoke paths need no file from the reference.  `MaskVRD(model_config(name), device)` accepts
either these or the `model_config` section parsed from the reference's YAML (train.py:42-49)."""
import copy

_VIDVRD = dict(
    visual_dim=1024, bbox_entity_dim=8, bbox_so_dim=5, embd_dim=512, num_classes=132,
    backbone_arch=[2, 2, 3], scale_factor=2, fpn_start_level=0, max_seq_len=96, n_mha_win_size=7,
    use_abs_pe=False, use_rel_pe=False, use_local=False, max_so_pair=200,
    fuse_ks=1, fuse_head=4, fuse_qx_stride=1, fuse_kv_stride=1, fuse_path_drop=0.1,
    n_head=4, embd_kernel_size=3, embd_with_ln=True, dropattn=0.0, dropout=0.0, droppath=0.1,
    fpn_dim=256, fpn_with_ln=True, fpn_norm_first=True, loss_types=['labels', 'masks'],
    with_fuzzy=True, scale_range=0.85,
    predictor=dict(n_input=512, n_embd=256, n_head=4, n_hidden=1024, num_queries=9, num_classes=132,
                   attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.1, cls_prior_prob=0.01, n_qx_stride=0,
                   n_kv_stride=1, num_layers=4, deep_supervision=True, enforce_input_project=False),
    cost_coeff_dict=dict(cost_class=1.0, cost_mask=5.0, cost_dice=5.0),
    loss_coeff_dict=dict(eos_coef=0.1, loss_class=1.0, loss_mask=5.0, loss_dice=5.0),
)


def _vidor(**over):
    c = copy.deepcopy(_VIDVRD)
    c.update(num_classes=50, max_seq_len=512, n_mha_win_size=9, fuse_head=8, n_head=8)
    c.pop("with_fuzzy"), c.pop("scale_range")
    c["predictor"].update(n_head=8, num_classes=50)
    c["cost_coeff_dict"].update(cost_class=2.0, cost_mask=2.0)
    c["loss_coeff_dict"].update(loss_class=2.0, loss_mask=2.0)
    c["with_clip_feature"] = False
    c.update(over)
    return c


_MODEL = {
    "vidvrd": _VIDVRD,
    "vidor": _vidor(clip_dim=512),
    "vidor_local": _vidor(use_local=True, clip_dim=512),     # the YAML carries clip_dim but no CLIP features
    "vidor_x": _vidor(clip_dim=512, with_clip_feature=True),
}
_MODEL["vidor_x"]["predictor"]["num_queries"] = 10

_INFER = {
    "vidvrd": dict(topk=8, feat_stride=1, pred_min_frames=2, n_max_pair=200, viou_th=0.5),
    "vidor": dict(topk=6, feat_stride=4, pred_min_frames=5, n_max_pair=200, viou_th=0.5),
    "vidor_local": dict(topk=1, feat_stride=4, pred_min_frames=5, n_max_pair=200, viou_th=0.5),
    "vidor_x": dict(topk=6, feat_stride=4, pred_min_frames=5, n_max_pair=200, viou_th=0.5),
}


def model_config(name):
    return copy.deepcopy(_MODEL[name])


def inference_config(name):
    return copy.deepcopy(_INFER[name])


def input_channels(cfg):
    clip = cfg["clip_dim"] if cfg.get("with_clip_feature", False) else 0
    return 2 * cfg["visual_dim"] + 2 * clip + cfg["bbox_so_dim"] + 2 * cfg["bbox_entity_dim"]
