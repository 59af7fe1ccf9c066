"""TransMIL with the reference's interface, computed by libacmil_hip.so (eval forward).

Mirrors `architecture/transMIL.py` of dazhangyu123/ACMIL: `TransLayer` (:8-28), `PPEG` (:31-45), `TransMIL`
(:48-91) and the `NystromAttention` parameter layout of the pip package it imports (vendored fork:
`architecture/nystrom_attention.py:29-65`) -- same constructor `TransMIL(conf)`, `forward(input [B,N,D_feat]) ->
logits [B,C]`, same `state_dict()` keys.  All arithmetic runs in `acmil_transmil_forward` (acmil_amd/csrc/
transmil.hip); the nn.Modules below are parameter containers.  Not implemented: training (the Dropout(0.1) of
`to_out` and the backward) and B > 1 (the reference's pinv couples batch rows; its shipped configs use B = 1).
"""
import torch
import torch.nn as nn

from .. import ops


class NystromAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6, residual=True,
                 residual_conv_kernel=33, eps=1e-8, dropout=0.0):
        super().__init__()
        if heads != 8 or pinv_iterations != 6 or residual_conv_kernel != 33 or not residual or dim_head * heads != dim \
                or num_landmarks != dim // 2:
            raise NotImplementedError("acmil_amd: only TransMIL's fixed Nystrom configuration is built (transMIL.py:13-23)")
        inner = heads * dim_head
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        self.res_conv = nn.Conv2d(heads, heads, (residual_conv_kernel, 1), padding=(residual_conv_kernel // 2, 0), groups=heads, bias=False)


class TransLayer(nn.Module):
    def __init__(self, norm_layer=nn.LayerNorm, dim=512):
        super().__init__()
        self.norm = norm_layer(dim)
        self.attn = NystromAttention(dim=dim, dim_head=dim // 8, heads=8, num_landmarks=dim // 2, pinv_iterations=6,
                                     residual=True, dropout=0.1)


class PPEG(nn.Module):
    def __init__(self, dim=512):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 7, 1, 7 // 2, groups=dim)
        self.proj1 = nn.Conv2d(dim, dim, 5, 1, 5 // 2, groups=dim)
        self.proj2 = nn.Conv2d(dim, dim, 3, 1, 3 // 2, groups=dim)


class TransMIL(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.pos_layer = PPEG(dim=conf.D_inner)
        self._fc1 = nn.Sequential(nn.Linear(conf.D_feat, conf.D_inner), nn.ReLU())
        self.cls_token = nn.Parameter(torch.randn(1, 1, conf.D_inner))
        self.n_classes = conf.n_class
        self.layer1 = TransLayer(dim=conf.D_inner)
        self.layer2 = TransLayer(dim=conf.D_inner)
        self.norm = nn.LayerNorm(conf.D_inner)
        self._fc2 = nn.Linear(conf.D_inner, conf.n_class)

    def forward(self, input, debug=False):
        """input [B=1, N, D_feat] -> logits [1, C]  (transMIL.py:60-91)."""
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("acmil_amd: TransMIL training (dropout + backward) is not built yet; use .eval()")
        if input.dim() != 3 or input.shape[0] != 1:
            raise RuntimeError("acmil_amd: TransMIL expects input [1, N, D_feat]")
        sd = dict(self.named_parameters())
        out = ops.transmil_forward(input[0], sd, self.n_classes, debug=debug)
        self._last = out
        return out["logits"].unsqueeze(0)
