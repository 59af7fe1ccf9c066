"""DTFD-style attention blocks on an already projected bag, same names / ctor signatures / state_dict keys as the
reference's `architecture/Attention.py` (Attention_Gated :29-57, Attention_with_Classifier :60-70).
Arithmetic: acmil_gated_scores / acmil_attn_pool / acmil_softmax_rows / acmil_gemm (csrc/attn_generic.hip) under
torch.no_grad(); with gradients enabled the same mathematics runs op by op through acmil_amd.autograd (Linear GEMMs, gate
kernel, row softmax, pooling GEMM, each with a HIP backward), so the modules can be trained."""
import torch
import torch.nn as nn

from .. import autograd as AG
from .. import ops
from .network import Classifier_1fc


def _needs_grad(x, *params):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))


class Attention_Gated(nn.Module):
    def __init__(self, L=512, D=128, K=1, *, precision="f16x3"):
        super().__init__()
        self.L, self.D, self.K = L, D, K
        self.attention_V = nn.Sequential(nn.Linear(L, D), nn.Tanh())
        self.attention_U = nn.Sequential(nn.Linear(L, D), nn.Sigmoid())
        self.attention_weights = nn.Linear(D, K)
        self.precision = precision

    def scores(self, x):
        v, u, w = self.attention_V[0], self.attention_U[0], self.attention_weights
        if not x.is_cuda:
            raise RuntimeError("acmil_amd: Attention_Gated runs on an MI355X only (no CPU fallback)")
        if _needs_grad(x, v.weight, u.weight, w.weight):
            return AG.gated_scores(x.float(), v.weight, v.bias, u.weight, u.bias, w.weight, w.bias, self.precision)
        return ops.gated_scores(x.float().contiguous(), v.weight, v.bias, u.weight, u.bias, w.weight, w.bias, self.precision)

    def forward(self, x, isNorm=True):   # x: N x L -> K x N   (Attention.py:47-57)
        A = self.scores(x)
        if not isNorm:
            return A
        return AG.softmax_rows(A.contiguous()) if A.requires_grad else ops.softmax_rows(A)


class Attention_with_Classifier(nn.Module):
    def __init__(self, L=512, D=128, K=1, num_cls=2, droprate=0, *, precision="f16x3"):
        super().__init__()
        if droprate != 0:
            raise NotImplementedError("acmil_amd: classifier dropout is a train-time feature; the eval forward has none")
        self.attention = Attention_Gated(L, D, K, precision=precision)
        self.classifier = Classifier_1fc(L, num_cls, droprate)

    def forward(self, x):   # x: N x L -> K x num_cls   (Attention.py:66-70)
        x = x.float().contiguous()
        A = self.attention.scores(x)
        if A.requires_grad or _needs_grad(x, self.classifier.fc.weight):
            afeat = AG.attn_pool(x, A)
            return AG.linear(afeat, self.classifier.fc.weight, self.classifier.fc.bias, precision="fp32")
        afeat = ops.attn_pool(x, A)                                  # softmax over N fused into the pooling pass
        return ops.gemm(afeat, self.classifier.fc.weight.detach(), trans_b=True, bias=self.classifier.fc.bias.detach())
