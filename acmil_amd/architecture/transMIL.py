"""TransMIL with the reference's interface, computed by libacmil_hip.so (eval forward).

Mirrors `architecture/transMIL.py` of dazhangyu123/ACMIL: `TransLayer` (:8-28), `PPEG` (:31-45), `TransMIL`
(:48-91) and the `NystromAttention` parameter layout of the pip package it imports (vendored fork:
`architecture/nystrom_attention.py:29-65`) -- same constructor `TransMIL(conf)`, `forward(input [B,N,D_feat]) ->
logits [B,C]`, same `state_dict()` keys.  All arithmetic runs in `acmil_transmil_forward` (acmil_amd/csrc/
transmil.hip) in eval mode; the nn.Modules below are parameter containers.  With gradients enabled the module runs the same
mathematics op by op through `acmil_amd.autograd` (HIP forward + backward kernels per op: GEMMs, LayerNorm, row softmax,
sequence conv, depth-wise 7x7, landmark means), in the reference's association, so `loss.backward()` works.
B > 1 (never used by the reference's scripts) follows the reference exactly -- its pinv initialisation couples the bags of a batch
through one global maximum -- on the op-by-op path with the bags in lockstep.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as AG
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

    # ------------------------------------------------------------------------------------------ training path
    def _attention(self, xs, layer, precision):
        """NystromAttention.forward (nystrom_attention.py:67-149) on a LIST of bags x [n, Di] (the batch of the reference's [B, n, Di]
        tensor), op by op with autograd.  Every stage acts per bag except one scalar: the Moore-Penrose initialisation divides by the
        maxima of the row / column sums over the WHOLE attn2 tensor, i.e. over batch and heads (nystrom_attention.py:16-18), which
        couples the bags of a batch -- so the bags advance in lockstep up to attn2, share that scalar, and finish one by one."""
        a = layer.attn
        n, di = xs[0].shape
        h, m = 8, di // 2
        d = di // h
        scale = d ** -0.5
        rem = n % m
        l = math.ceil(n / m)
        staged = []
        for x in xs:
            if rem > 0:
                x = F.pad(x, (0, 0, m - rem, 0), value=0.0)                   # FRONT zero padding (:72-75)
            npad = x.shape[0]
            qkv = AG.linear(x, a.to_qkv.weight, None, precision=precision)    # [npad, 3 Di]
            q, k, v = qkv[:, :di], qkv[:, di:2 * di], qkv[:, 2 * di:]
            heads = lambda t: t.reshape(npad, h, d).permute(1, 0, 2)         # [h, npad, d] strided views of qkv
            qh, kh, vh = heads(q), heads(k), heads(v)
            q_l, k_l = AG.landmark_mean(q, l), AG.landmark_mean(k, l)         # [h, m, d]   (scale folded into the products)
            attn2 = AG.softmax_rows(AG.matmul(q_l, k_l, trans_b=True, alpha=scale))       # the reference scales q once (:91)
            staged.append((npad, qh, kh, vh, v, q_l, k_l, attn2))
        # Moore-Penrose iteration (nystrom_attention.py:12-27): tiny [h, m, m] tensors, products on the exact GEMM
        col = torch.stack([st[7].abs().sum(dim=-1).max() for st in staged]).max()
        row = torch.stack([st[7].abs().sum(dim=-2).max() for st in staged]).max()
        eye = torch.eye(m, device=xs[0].device, dtype=xs[0].dtype).unsqueeze(0)
        outs = []
        for npad, qh, kh, vh, v, q_l, k_l, attn2 in staged:
            attn1 = AG.softmax_rows(AG.matmul(qh, k_l, trans_b=True, alpha=scale))        # [h, npad, m]
            attn3 = AG.softmax_rows(AG.matmul(q_l, kh, trans_b=True, alpha=scale))        # [h, m, npad]
            z = attn2.transpose(-1, -2) / (col * row)
            for _ in range(6):
                xz = AG.matmul(attn2, z)
                z = 0.25 * AG.matmul(z, 13 * eye - AG.matmul(xz, 15 * eye - AG.matmul(xz, 7 * eye - xz)))
            out = AG.matmul(AG.matmul(attn1, z), AG.matmul(attn3, vh))                      # reference association (:133)
            out = out.permute(1, 0, 2).reshape(npad, di) + AG.seq_conv(v, a.res_conv.weight)
            out = AG.linear(out, a.to_out[0].weight, a.to_out[0].bias, precision=precision)
            out = F.dropout(out, a.to_out[1].p, self.training)
            outs.append(out[-n:])
        return outs

    def _forward_train(self, xs, precision="f16x3"):
        """transMIL.py:60-91 with autograd-capable ops; xs = list of B bags [N, D_feat] (same N) -> logits [B, C]"""
        di = self._fc1[0].out_features
        hs = []
        for x in xs:
            h = AG.linear(x.float().contiguous(), self._fc1[0].weight, self._fc1[0].bias, relu=True, precision=precision)
            n0 = h.shape[0]
            side = int(math.ceil(math.sqrt(n0)))
            hs.append(torch.cat([self.cls_token.reshape(1, di), h, h[:side * side - n0]], dim=0))   # cls + tokens + wrap-around padding (:64-72)
        att = self._attention([AG.layer_norm(h, self.layer1.norm.weight, self.layer1.norm.bias) for h in hs], self.layer1, precision)
        hs = [h + o for h, o in zip(hs, att)]
        p = self.pos_layer                                                    # PPEG: one folded depth-wise 7x7 (:38-45)
        weff = p.proj.weight[:, 0] + F.pad(p.proj1.weight[:, 0], (1, 1, 1, 1)) + F.pad(p.proj2.weight[:, 0], (2, 2, 2, 2))
        ident = torch.zeros(7, 7, device=xs[0].device); ident[3, 3] = 1.0
        weff = (weff + ident).reshape(di, 49).t().contiguous()                # [49, C] tap-major
        beff = p.proj.bias + p.proj1.bias + p.proj2.bias
        hs = [torch.cat([h[:1], AG.dwconv7(h[1:], weff, beff, side)], dim=0) for h in hs]
        att = self._attention([AG.layer_norm(h, self.layer2.norm.weight, self.layer2.norm.bias) for h in hs], self.layer2, precision)
        cls = torch.cat([AG.layer_norm((h + o)[:1], self.norm.weight, self.norm.bias) for h, o in zip(hs, att)], dim=0)
        return AG.linear(cls, self._fc2.weight, self._fc2.bias, precision="fp32")

    def forward(self, input, debug=False):
        """input [B, N, D_feat] -> logits [B, C]  (transMIL.py:60-91).  B = 1 (what the reference's scripts use) is the one-call HIP
        forward.  B > 1 keeps the reference's semantics INCLUDING the batch-global scalar of the pinv initialisation
        (nystrom_attention.py:16-18 takes its maxima over batch and heads, so the bags of a batch are not independent): it runs the
        op-by-op HIP path with the bags in lockstep (`_attention`), not B independent B = 1 forwards."""
        if input.dim() != 3 or input.shape[0] < 1:
            raise RuntimeError("acmil_amd: TransMIL expects input [B, N, D_feat]")
        if not input.is_cuda:
            raise RuntimeError("acmil_amd: TransMIL runs on an MI355X only (no CPU fallback)")
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())):
            return self._forward_train([input[b] for b in range(input.shape[0])])
        if self.training:
            raise NotImplementedError("acmil_amd: train-mode TransMIL draws dropout masks; call it with gradients enabled or use .eval()")
        if input.shape[0] > 1:
            if debug:
                raise NotImplementedError("acmil_amd: debug intermediates are a B = 1 feature")
            return self._forward_train([input[b] for b in range(input.shape[0])])
        sd = dict(self.named_parameters())
        out = ops.transmil_forward(input[0], sd, self.n_classes, debug=debug)
        self._last = out
        return out["logits"].unsqueeze(0)
