"""Deterministic synthetic weights and inputs for smoke / bench runs (there are no datasets or
checkpoints on the box).  Name-seeded so every process regenerates identical tensors:
O(1) drop-path scales (the 1e-4 init would hide every attention / MLP branch), LayerNorm affine
near identity, conv weights N(0,1)/sqrt(fan_in) so activations stay O(1)."""
import hashlib
import math

import torch


def synth_tensor(name, shape, ln_bias_std=0.1):
    shape = tuple(shape)
    if name == "empty_weight":
        return None
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
    g = torch.Generator().manual_seed(seed)
    leaf = name.rsplit(".", 1)[-1]
    is_ln = len(shape) == 3 and shape[0] == 1 and shape[2] == 1
    if leaf == "scale":
        return torch.rand(shape, generator=g) + 0.5
    if is_ln:
        return 1.0 + 0.1 * torch.randn(shape, generator=g) if leaf == "weight" else ln_bias_std * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.02 * torch.randn(shape, generator=g)
    fan_in = 1 if "query_embed" in name else max(1, math.prod(shape[1:]))
    return torch.randn(shape, generator=g) / math.sqrt(fan_in)


def load_synthetic_weights(model):
    """Overwrite every parameter of `model` in place (buffers such as empty_weight are kept)."""
    sd = model.state_dict()
    new = {}
    for name, t in sd.items():
        s = synth_tensor(name, t.shape)
        new[name] = t if s is None else s.to(t.dtype)
    model.load_state_dict(new, strict=True)
    return model


def synth_pairs(n_pairs, c_in, t_pad, lengths=None, seed=1234, device="cpu"):
    """x ~ N(0,1) * mask of shape (B, C_in, T_pad); mask (B, 1, T_pad) bool with mask[b, t] = t < len_b."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(n_pairs, c_in, t_pad, generator=g, device=device)
    if lengths is None:
        lengths = torch.full((n_pairs,), t_pad, dtype=torch.long)
    lengths = torch.as_tensor(lengths).to(device)
    mask = (torch.arange(t_pad, device=device)[None, :] < lengths[:, None])[:, None, :]
    return x * mask.to(x.dtype), mask


def synth_video(n_tracklets, c_in, min_len, max_len, seed=7, device="cpu"):
    """One synthetic video for MaskVRD.forward_test: n tracklets with random durations inside a (max_len + 8)-frame
    video and every ordered pair whose overlap is >= 2 frames (46 tracklets -> up to 2070 pairs), in the dataloader's
    eval format (dataloaders/vidvrd.py:706-715): per-pair features as an (L, C_in) matrix viewed (C_in, L)."""
    g = torch.Generator().manual_seed(seed)
    video_len = max_len + 8
    span, boxes = [], []
    for _ in range(n_tracklets):
        n = int(torch.randint(min_len, max_len + 1, (1,), generator=g))
        t0 = int(torch.randint(0, video_len - n + 1, (1,), generator=g))
        span.append((t0, t0 + n))
        corner = torch.rand(n, 2, generator=g) * 100.0
        boxes.append(torch.cat([corner, corner + torch.rand(n, 2, generator=g) * 50.0 + 1.0], dim=1).to(device))
    sids, oids, feats = [], [], []
    for s in range(n_tracklets):
        for o in range(n_tracklets):
            overlap = min(span[s][1], span[o][1]) - max(span[s][0], span[o][0])
            if s != o and overlap >= 2:
                sids.append(s)
                oids.append(o)
                feats.append(torch.randn(overlap, c_in, generator=g).to(device).permute(1, 0))
    return {"sids": torch.tensor(sids, device=device), "oids": torch.tensor(oids, device=device),
            "so_features_list": feats, "bboxes_list": boxes,
            "cat_ids": torch.randint(1, 36, (n_tracklets,), generator=g).to(device),
            "cat_scores": torch.rand(n_tracklets, generator=g).to(device),
            "traj_durations": torch.tensor(span, device=device), "so_offset": torch.zeros(len(sids), dtype=torch.long, device=device)}


def synth_raw_video(n_tracklets, n_visual, min_len, max_len, seed=7, n_clip=0, wh=(1280, 720)):
    """One synthetic video in the form the reference's `_prepare_test` hands to `_test_getitem`
    (dataloaders/vidvrd.py:459-550) and vrdone_amd.proposals.prepare_test_proposal takes: per-TRACKLET visual (and CLIP)
    features and boxes on the host, durations [start, end), distinct categories, every ordered pair of tracklets that
    share a frame."""
    g = torch.Generator().manual_seed(seed)
    video_len = max_len + 8
    spans, boxes, vis, clip = [], [], [], []
    for _ in range(n_tracklets):
        n = int(torch.randint(min_len, max_len + 1, (1,), generator=g))
        t0 = int(torch.randint(0, video_len - n + 1, (1,), generator=g))
        spans.append((t0, t0 + n))
        corner = torch.rand(n, 2, generator=g) * torch.tensor([wh[0] * 0.6, wh[1] * 0.6])
        boxes.append(torch.cat([corner, corner + torch.rand(n, 2, generator=g) * torch.tensor([wh[0] * 0.3, wh[1] * 0.3]) + 8.0], dim=1))
        vis.append(torch.randn(n, n_visual, generator=g))
        if n_clip:
            clip.append(torch.randn(n, n_clip, generator=g))
    sids, oids = [], []
    for s in range(n_tracklets):
        for o in range(n_tracklets):
            if s != o and min(spans[s][1], spans[o][1]) > max(spans[s][0], spans[o][0]):
                sids.append(s)
                oids.append(o)
    out = {"sids": torch.tensor(sids), "oids": torch.tensor(oids), "cat_ids": torch.arange(1, n_tracklets + 1),
           "cat_scores": torch.rand(n_tracklets, generator=g), "bboxes_list": boxes,
           "traj_durations": torch.tensor(spans, dtype=torch.int64), "visual_features_list": vis, "video_wh": wh}
    if n_clip:
        out["clip_features_list"] = clip
    return out
