"""Matching costs and losses of the training criterion (reference models/losses.py:4-354).

The reference prices every prediction of the batch against every ground-truth relation of the batch -- a
(B*Q, sum N_gt) matrix per term -- and then reads only the diagonal blocks (maskvrd.py:487-491).  Here the
block-diagonal is what is computed: a relation is priced against the Q queries of its own pair only
(`pair_costs`), as one gather + reduction over T per term.  The `batch_masked_*` / `masked_*` functions keep the
reference's names and argument meaning for callers that use them directly; all of them are thin wrappers around
the three helpers below.  Tensors stay on the device of the predictions; arithmetic is fp32 like the reference.

`device_criterion` is the fused form the training step uses on the GPU: matching costs, assignment, losses and their
gradients for all decoder layers as four kernel launches (csrc/vrd_criterion.hip, vrd_assign); the tensor functions above
stay as its CPU / fallback form and as what the tests compare it with.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- helpers
def focal_sides(logits, alpha: float = 0.25, gamma: float = 2.0):
    """Focal loss of every element against target 1 and against target 0 (losses.py:21-31):
    alpha (1-p)^gamma softplus(-x)  and  (1-alpha) p^gamma softplus(x)."""
    p = torch.sigmoid(logits)
    pos = (1 - p) ** gamma * F.softplus(-logits)
    neg = p ** gamma * F.softplus(logits)
    if alpha >= 0:
        pos, neg = pos * alpha, neg * (1 - alpha)
    return pos, neg


def fuzzy_targets(targets, segs, valid, scale_range: float):
    """Soft relation masks (losses.py:214-227): 1 inside the segment shrunk by scale_range around its centre,
    sqrt(cos) ramp between the shrunk and the 1/scale_range-widened segment (valid frames only), 0 outside.
    targets (G, T) 0/1, segs (G, 2) integer [start, end), valid (G, T) bool."""
    assert scale_range <= 1.0
    G, T = valid.shape
    lo, hi = segs[:, 0], segs[:, 1]
    centre = (hi - 1 + lo).float() / 2
    off = torch.arange(T, device=valid.device, dtype=torch.float32)[None, :] - centre[:, None]
    length = (hi - lo)[:, None]
    core = off.abs() < (length / 2 * scale_range)
    wide = (off.abs() < (length / 2 / scale_range)) & valid
    ramp = (wide ^ core) & valid
    w = torch.cos(math.pi / (length / scale_range) * off)
    w = (w * (w > 0)) ** 0.5
    return w * ramp + targets * core


def _soft_or_hard(targets, segs, valid, scale_range):
    return targets if segs is None else fuzzy_targets(targets, segs, valid, scale_range)


def pair_costs(pred_logits, pred_masks, out_valid, tgt_ids, tgt_masks, owner, segs=None, scale_range=None,
               alpha: float = 0.25, gamma: float = 2.0):
    """Costs of giving relation g to query q of its own pair, for all g at once.
    pred_logits (B, Q, K+1), pred_masks (B, Q, T), out_valid (B, T) bool, tgt_ids (G,), tgt_masks (G, T),
    owner (G,) pair index of each relation -> cost_class, cost_mask, cost_dice, each (G, Q).
    Entry [g, q] equals entry [owner[g]*Q + q, g] of the reference's matrices (maskvrd.py:447-478)."""
    valid = out_valid[owner]                                    # (G, T): frames of the relation's pair
    vf = valid.to(pred_masks.dtype)
    tp = _soft_or_hard(tgt_masks, segs, valid, scale_range)
    cost_class = -F.log_softmax(pred_logits, dim=-1)[owner, :, tgt_ids]          # (G, Q)
    pos, neg = focal_sides(pred_masks, alpha, gamma)
    of = out_valid.to(pred_masks.dtype)[:, None, :]
    pos, neg = (pos * of)[owner], (neg * of)[owner]                               # (G, Q, T)
    cost_mask = (pos * (tp * vf)[:, None, :]).sum(-1) + (neg * ((1 - tp) * vf)[:, None, :]).sum(-1)
    cost_mask = cost_mask / out_valid.sum(-1)[owner][:, None]
    sig = torch.sigmoid(pred_masks) * of
    tpv = tp * vf
    num = 2 * (sig[owner] * tpv[:, None, :]).sum(-1)
    den = sig.sum(-1)[owner] + tpv.sum(-1)[:, None]
    cost_dice = 1 - (num + 1) / (den + 1)
    return cost_class, cost_mask, cost_dice


def matched_losses(inputs, targets, num_masks, loss_mask, segs=None, scale_range=None,
                   alpha: float = 0.25, gamma: float = 2.0):
    """Focal and dice loss of matched (prediction, relation) rows (losses.py:98-129, 151-172, 271-354).
    inputs, targets, loss_mask: (G, T); returns (focal, dice) scalars already divided by num_masks."""
    mf = loss_mask.to(inputs.dtype)
    tp = _soft_or_hard(targets, segs, loss_mask, scale_range)
    p = torch.sigmoid(inputs)
    # the fuzzy variant evaluates the cross-entropy against the target restricted to valid frames but the
    # modulating factor against the unrestricted one (losses.py:303-310); the hard variant uses targets for both
    ce_t = tp * loss_mask if segs is not None else tp
    ce = F.binary_cross_entropy_with_logits(inputs, ce_t, reduction="none")
    p_t = p * tp + (1 - p) * (1 - tp)
    focal = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        focal = (alpha * tp + (1 - alpha) * (1 - tp)) * focal
    focal = (mf * focal).mean(1).sum() / num_masks
    pm, tm = p * mf, tp * mf
    dice = 1 - (2 * (pm * tm).sum(-1) + 1) / (pm.sum(-1) + tm.sum(-1) + 1)
    return focal, dice.sum() / num_masks


def _all_pairs(fn):
    """Reference-shaped wrapper: every row of inputs (N, T) against every row of targets (M, T) -> (N, M)."""
    def run(inputs, targets, batch_out_mask, batch_tgt_mask, batch_tgt_seg=None, scale_range=None, **kw):
        N, M = inputs.shape[0], targets.shape[0]
        tp = _soft_or_hard(targets, batch_tgt_seg, batch_tgt_mask, scale_range) * batch_tgt_mask.to(inputs.dtype)
        return fn(inputs, tp, batch_out_mask.to(inputs.dtype), batch_tgt_mask.to(inputs.dtype), **kw).reshape(N, M)
    return run


@_all_pairs
def _focal_matrix(x, tpv, of, tf, alpha: float = 0.25, gamma: float = 2.0):
    pos, neg = focal_sides(x, alpha, gamma)
    return ((pos * of) @ tpv.T + (neg * of) @ (tf - tpv).T) / of.sum(-1, keepdim=True)


@_all_pairs
def _ce_matrix(x, tpv, of, tf):
    return ((F.softplus(-x) * of) @ tpv.T + (F.softplus(x) * of) @ (tf - tpv).T) / of.sum(-1, keepdim=True)


@_all_pairs
def _dice_matrix(x, tpv, of, tf):
    s = torch.sigmoid(x) * of
    return 1 - (2 * s @ tpv.T + 1) / (s.sum(-1)[:, None] + tpv.sum(-1)[None, :] + 1)


# ------------------------------------------------------------------- the reference's names (models/losses.py)
def batch_masked_sigmoid_focal_loss(inputs, targets, batch_out_mask, batch_tgt_mask, alpha: float = 0.25, gamma: float = 2):
    return _focal_matrix(inputs, targets, batch_out_mask, batch_tgt_mask, alpha=alpha, gamma=gamma)


def batch_masked_sigmoid_ce_loss(inputs, targets, batch_out_mask, batch_tgt_mask):
    return _ce_matrix(inputs, targets, batch_out_mask, batch_tgt_mask)


def batch_masked_dice_loss(inputs, targets, batch_out_mask, batch_tgt_mask):
    return _dice_matrix(inputs, targets, batch_out_mask, batch_tgt_mask)


def batch_masked_sigmoid_focal_fuzzy_loss(inputs, targets, batch_out_mask, batch_tgt_mask, batch_tgt_seg,
                                          scale_range: float = 0.8, alpha: float = 0.25, gamma: float = 2):
    return _focal_matrix(inputs, targets, batch_out_mask, batch_tgt_mask, batch_tgt_seg, scale_range,
                         alpha=alpha, gamma=gamma)


def batch_masked_dice_fuzzy_loss(inputs, targets, batch_out_mask, batch_tgt_mask, batch_tgt_seg, scale_range: float = 0.8):
    return _dice_matrix(inputs, targets, batch_out_mask, batch_tgt_mask, batch_tgt_seg, scale_range)


def masked_sigmoid_focal_loss(inputs, targets, num_masks, loss_mask, alpha: float = 0.25, gamma: float = 2):
    return matched_losses(inputs, targets, num_masks, loss_mask, alpha=alpha, gamma=gamma)[0]


def masked_dice_loss(inputs, targets, num_masks, loss_mask):
    return matched_losses(inputs, targets, num_masks, loss_mask)[1]


def masked_sigmoid_ce_loss(inputs, targets, num_masks, loss_mask):
    mf = loss_mask.to(inputs.dtype)
    ce = mf * F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return (ce.sum(1) / mf.sum(1)).sum() / num_masks


def masked_sigmoid_focal_fuzzy_loss(inputs, targets, num_masks, loss_mask, tgt_segs, scale_range: float = 0.8,
                                    alpha: float = 0.25, gamma: float = 2):
    return matched_losses(inputs, targets, num_masks, loss_mask, tgt_segs, scale_range, alpha, gamma)[0]


def masked_dice_fuzzy_loss(inputs, targets, num_masks, loss_mask, tgt_segs, scale_range: float = 0.8):
    return matched_losses(inputs, targets, num_masks, loss_mask, tgt_segs, scale_range)[1]


# ------------------------------------------------------------------------------------- fused device criterion
class _Criterion(torch.autograd.Function):
    """(n_layers, 3) losses [class, focal, dice] of the given assignment; backward = vrd_criterion_backward."""

    @staticmethod
    def forward(ctx, pack, q_of, *preds):
        from .. import _hip
        from ..ops import _stream
        args, keep, class_weight, num_masks = pack
        L = args.n_layers
        out = torch.empty(L, 4, device=q_of.device, dtype=torch.float32)
        _hip.check(_hip.lib.vrd_criterion_losses(_hip.C.byref(args), q_of.data_ptr(), class_weight.data_ptr(), num_masks, out.data_ptr(),
                                                 _stream()), "vrd_criterion_losses")
        ctx.pack, ctx.shapes = pack, [p.shape for p in preds]
        ctx.save_for_backward(q_of, out, *preds)
        return out[:, :3].clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        from .. import _hip
        from ..ops import _stream
        args, keep, class_weight, num_masks = ctx.pack
        q_of, out, *preds = ctx.saved_tensors
        grads = [torch.empty_like(p) for p in preds]
        gr = _hip.CriterionGrads()
        L = args.n_layers
        for l in range(L):
            gr.logits[l], gr.masks[l] = grads[l].data_ptr(), grads[L + l].data_ptr()
        gout = gout.contiguous().float()
        _hip.check(_hip.lib.vrd_criterion_backward(_hip.C.byref(args), q_of.data_ptr(), class_weight.data_ptr(), num_masks, out.data_ptr(),
                                                   gout.data_ptr(), _hip.C.byref(gr), _stream()), "vrd_criterion_backward")
        return (None, None) + tuple(grads)


def device_criterion(layers, out_valid, sizes, tgt_ids, tgt_masks, segs, scale_range, class_weight, cost_w, alpha=0.25, gamma=2.0):
    """Matching + losses of up to four decoder layers on the device.
    layers: [(pred_logits (B, Q, K1), pred_masks (B, Q, T))], final head first; out_valid (B, T) bool; sizes [N_p] relations
    per pair; tgt_ids (G,) int64, tgt_masks (G, T) f32, segs (G, 2) or None (fuzzy targets with scale_range);
    class_weight (K1,); cost_w = (w_class, w_mask, w_dice).
    Returns (losses (n_layers, 3) [class, focal, dice] -- differentiable with respect to the predictions --,
    query_of (n_layers, G) int32, failed: 0-d bool tensor, true when some pair's costs were NaN / infinite)."""
    from .. import _hip, ops
    from ..ops import _stream
    L = len(layers)
    assert 1 <= L <= 4
    logits = [lg.contiguous().float() for lg, _ in layers]
    masks = [mk.contiguous().float() for _, mk in layers]
    B, Q, K1 = logits[0].shape
    T = masks[0].shape[-1]
    dev = logits[0].device
    G = sum(sizes)
    a = _hip.CriterionArgs()
    for l in range(L):
        assert logits[l].shape == (B, Q, K1) and masks[l].shape == (B, Q, T)
        a.logits[l], a.masks[l] = logits[l].data_ptr(), masks[l].data_ptr()
    valid_u8 = out_valid.contiguous().view(torch.uint8) if out_valid.dtype == torch.bool else out_valid.contiguous()
    owner = torch.repeat_interleave(torch.arange(len(sizes), device=dev, dtype=torch.int32),
                                    torch.tensor(sizes, device=dev), output_size=G)
    tgt_ids = tgt_ids.contiguous().long()
    tgt_masks = tgt_masks.contiguous().float()
    segs32 = None if segs is None else segs.to(torch.int32).contiguous()
    assert tgt_masks.shape == (G, T) and valid_u8.shape == (B, T)
    a.n_layers, a.B, a.Q, a.K1, a.T, a.G = L, B, Q, K1, T, G
    a.out_valid, a.tgt_ids, a.tgt_masks, a.owner = valid_u8.data_ptr(), tgt_ids.data_ptr(), tgt_masks.data_ptr(), owner.data_ptr()
    a.segs = None if segs32 is None else segs32.data_ptr()
    a.scale_range = float(scale_range) if segs32 is not None else 1.0
    a.alpha, a.gamma = float(alpha), float(gamma)
    a.w_class, a.w_mask, a.w_dice = (float(w) for w in cost_w)
    keep = (valid_u8, owner, tgt_ids, tgt_masks, segs32, logits, masks)          # what the argument struct points into
    with torch.no_grad():
        cost = torch.empty(L, G, Q, device=dev, dtype=torch.float32)
        _hip.check(_hip.lib.vrd_criterion_costs(_hip.C.byref(a), cost.data_ptr(), _stream()), "vrd_criterion_costs")
        q_of = ops.assign(cost.view(L * G, Q), list(sizes) * L)                    # every layer's pairs in one launch
        failed = (q_of < 0).any()
    class_weight = class_weight.to(device=dev, dtype=torch.float32).contiguous()
    num_masks = float(max(G, 1))
    losses = _Criterion.apply((a, keep, class_weight, num_masks), q_of, *logits, *masks)
    return losses, q_of.view(L, G), failed
