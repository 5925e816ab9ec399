"""Top-down 1-D feature pyramid that fuses the four backbone levels into one full-resolution
mask-feature map.  Same constructor and parameter tree as the reference's FPN1D_Fuse
(models/fpns.py:141-257); every level is one or two fused HIP kernels:
    top:    LN_in -> [grouped k3 conv * mask -> LN]
    others: LN_in -> 1x1 lateral GEMM * mask -> LN -> [(+ nearest x2 upsample of the coarser level)
            -> depthwise k3 conv * mask -> LN]
    out:    depthwise k3 conv (+bias) * mask
"""
from torch import nn

from .blocks import LayerNorm, MaskedConv1D, _from_cl, _mask2d, _ops, _to_cl


class FPN1D_Fuse(nn.Module):
    def __init__(self, in_channels, out_channel, scale_factor=2.0, start_level=0, end_level=-1, with_ln=True,
                 norm_first=False):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        assert with_ln and norm_first and start_level == 0 and int(scale_factor) == 2, \
            "built for the shipped configs: fpn_with_ln, fpn_norm_first, start level 0, scale factor 2"
        self.in_channels, self.out_channel, self.scale_factor = in_channels, out_channel, scale_factor
        self.start_level = start_level
        self.end_level = len(in_channels) if end_level == -1 else end_level
        assert self.start_level < self.end_level <= len(in_channels)

        self.input_norms = nn.ModuleList()
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        self.lateral_norms = nn.ModuleList()
        self.fpn_norms = nn.ModuleList()
        for i in range(self.start_level, self.end_level):
            top = i == self.end_level - 1
            self.input_norms.append(LayerNorm(in_channels[i]))
            self.lateral_convs.append(None if top else MaskedConv1D(in_channels[i], out_channel, 1, bias=False))
            self.lateral_norms.append(None if top else LayerNorm(out_channel))
            self.fpn_convs.append(MaskedConv1D(in_channels[i] if top else out_channel, out_channel, 3, padding=1,
                                               bias=False, groups=out_channel))
            self.fpn_norms.append(LayerNorm(out_channel))
        self.mask_features = MaskedConv1D(out_channel, out_channel, 3, padding=1, groups=out_channel)

    def cl(self, feats, masks):
        ops = _ops()
        y = None
        for l in range(len(self.lateral_convs) - 1, -1, -1):
            # below the top level the normalised input feeds only the lateral 1x1 GEMM
            x = self.input_norms[l].cl(feats[l], pair=ops.pair_mode() and self.lateral_convs[l] is not None)
            fpn = dict(weight=self.fpn_convs[l].conv.weight, gamma=self.fpn_norms[l].weight, beta=self.fpn_norms[l].bias)
            if self.lateral_convs[l] is None:
                y, = ops.dwconv_ln(x, [fpn], mask_out=masks[l])
            else:
                c = ops.conv_gemm(x, self.lateral_convs[l].conv.weight, None, row_mask=masks[l])
                c = self.lateral_norms[l].cl(c)
                y, = ops.dwconv_ln(c, [fpn], mask_out=masks[l], x_up=y)
        mf = self.mask_features.conv
        out, = ops.dwconv_ln(y, [dict(weight=mf.weight, bias=mf.bias)], mask_out=masks[0])
        return out, masks[0]

    def forward(self, inputs, fpn_masks):
        assert len(inputs) == len(self.in_channels) == len(fpn_masks)
        out, m = self.cl([_to_cl(x) for x in inputs], [_mask2d(m) for m in fpn_masks])
        return _from_cl(out), m[:, None, :]
