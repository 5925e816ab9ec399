"""MaskFormer-style 1-D mask-segmentation head: query decoder over the coarsest level, class
logits per query and a per-query temporal mask = <mask embedding, mask features>.  Same
constructor and parameter tree as the reference's models/predictor.py."""
import math

from torch import nn

from .blocks import ConvMLP, LayerNorm, _mask2d, _ops, _to_cl
from .local_transformer import MaskedConvTransformerDecoderOnly


class MaskedTransformerPredictor(nn.Module):
    def __init__(self, n_input, n_embd, n_head, n_hidden, num_queries, num_classes, attn_pdrop=0.0, proj_pdrop=0.0,
                 path_pdrop=0.1, cls_prior_prob=0.01, n_qx_stride=0, n_kv_stride=1, num_layers=4,
                 deep_supervision=False, enforce_input_project=False):
        super().__init__()
        self.transformer = MaskedConvTransformerDecoderOnly(
            n_embd, n_head, n_hidden, attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop, path_pdrop=path_pdrop,
            n_qx_stride=n_qx_stride, n_kv_stride=n_kv_stride, num_layers=num_layers,
            return_intermediate=deep_supervision)
        self.num_queries = num_queries
        self.query_embed = nn.Embedding(num_queries, n_embd)
        self.input_norm = LayerNorm(n_input)
        self.input_proj = None
        if n_input != n_embd or enforce_input_project:
            self.input_proj = nn.Conv1d(n_input, n_embd, kernel_size=1)
            nn.init.zeros_(self.input_proj.bias)
        self.aux_loss = deep_supervision
        self.class_embed = nn.Conv1d(n_embd, num_classes + 1, 1)      # + background
        nn.init.constant_(self.class_embed.bias, -math.log((1 - cls_prior_prob) / cls_prior_prob))
        self.mask_embed = ConvMLP(n_embd, n_embd, n_embd, 3)

    def _heads(self, hs, mask_features, output_mask, fill):
        ops = _ops()
        logits = ops.conv_gemm(hs, self.class_embed.weight, self.class_embed.bias)      # (B, Q, K+1)
        seg = ops.mask_head(self.mask_embed.cl(hs), mask_features, output_mask, fill)   # (B, Q, T)
        return logits, seg

    def cl(self, x, mask_features, mask, output_mask, with_aux=None, non_attn_const=-10):
        """x (B, T/8, D), mask_features (B, T, Dp), mask (B, T/8), output_mask (B, T).
        with_aux=None follows the reference (all decoder layers' heads when deep_supervision);
        with_aux=False computes the last layer only (what forward_test reads, maskvrd.py:206)."""
        ops = _ops()
        if with_aux is None:
            with_aux = self.aux_loss
        src = self.input_norm.cl(x, pair=ops.pair_mode() and self.input_proj is not None)
        if self.input_proj is not None:
            src = ops.conv_gemm(src, self.input_proj.weight, self.input_proj.bias, row_mask=mask)
        hs = self.transformer.cl(src, mask, self.query_embed.weight, all_layers=with_aux and self.aux_loss)
        logits, seg = self._heads(hs[-1], mask_features, output_mask, float(non_attn_const))
        out = {"pred_logits": logits, "pred_masks": seg}
        if with_aux and self.aux_loss:
            out["aux_outputs"] = [dict(zip(("pred_logits", "pred_masks"),
                                           self._heads(h, mask_features, output_mask, float(non_attn_const))))
                                  for h in hs[:-1]]
        out["output_mask"] = output_mask[:, None, :]
        return out

    def forward(self, x, mask_features, mask, output_mask, non_attn_const=(-10)):
        return self.cl(_to_cl(x), _to_cl(mask_features), _mask2d(mask), _mask2d(output_mask),
                       non_attn_const=non_attn_const)
