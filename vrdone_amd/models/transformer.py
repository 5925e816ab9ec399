"""The two helpers of the reference's models/transformer.py that the hot path uses
(reference models/transformer.py:439-460); the DETR transformer in that file is dead code."""
import copy

from torch import nn
from torch.nn import functional as F


def _get_clones(module, n):
    return nn.ModuleList(copy.deepcopy(module) for _ in range(n))


def _get_activation_fn(activation, in_module=False):
    table = {"relu": (F.relu, nn.ReLU), "gelu": (F.gelu, nn.GELU), "glu": (F.glu, nn.GLU)}
    if activation not in table:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
    return table[activation][1 if in_module else 0]
