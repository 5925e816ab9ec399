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
