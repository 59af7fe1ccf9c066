"""IBMIL (interventional bag MIL): same ctor / parameter names / return contract as the reference's
`architecture/ibmil.py:38-113`.  Without a confounder dictionary it is structurally ABMIL (DimReduction +
Attention_Gated(Di, 128, 1) + Classifier_1fc) and runs on the fully fused forward kernel; only the returned attention map
needs one extra row softmax.  With `conf.c_path` (the deconfounding stage) the pooled bag feature attends over the loaded
dictionary (W_q, W_k, softmax over its entries) and is merged ('cat' / 'add' / 'sub') ahead of an nn.Linear classifier."""
import torch
import torch.nn as nn

from .. import autograd as AG
from .. import ops
from .network import Classifier_1fc, DimReduction
from .transformer import Attention_Gated, _GatedBase


class IBMIL(_GatedBase):
    def __init__(self, conf, confounder_dim=128, confounder_merge='cat', *, precision="f16x3"):
        super().__init__()
        assert confounder_merge in ['cat', 'add', 'sub']
        self.confounder_merge, self.confounder_path = confounder_merge, None
        self.dimreduction = DimReduction(conf.D_feat, conf.D_inner)
        self.attention = Attention_Gated(conf.D_inner, 128, 1)
        self.classifier = Classifier_1fc(conf.D_inner, conf.n_class, 0)
        if getattr(conf, "c_path", None):                 # deconfounding branch, ibmil.py:45-67
            import numpy as np
            self.confounder_path = conf.c_path
            conf_tensor = torch.cat([torch.from_numpy(np.load(i)).view(-1, conf.D_inner).float() for i in conf.c_path], 0)
            if getattr(conf, "c_learn", False):
                self.confounder_feat = nn.Parameter(conf_tensor, requires_grad=True)
            else:
                self.register_buffer("confounder_feat", conf_tensor)
            self.W_q = nn.Linear(conf.D_inner, confounder_dim)
            self.W_k = nn.Linear(conf_tensor.shape[-1], confounder_dim)
            self.classifier = nn.Linear(conf.D_inner + conf_tensor.shape[-1] if confounder_merge == 'cat' else conf.D_inner, conf.n_class)
            self.dropout = nn.Dropout(0.5)                # defined by the reference, never applied in its forward
        self.precision = precision

    def _heads(self):
        return [self.classifier.fc.weight], [self.classifier.fc.bias], None, None

    def _deconfound(self, M, lin, mm, softmax_rows):
        """ibmil.py:93-107 on the pooled bag feature M [1, Di]: attention of the bag query over the confounder dictionary
        (softmax over the dictionary entries), weighted confounder feature, merge, classifier.  lin / mm / softmax_rows are the
        differentiable (acmil_amd.autograd) or the plain (acmil_amd.ops) HIP products -- all of it is O(dictionary x Di)."""
        cf = self.confounder_feat
        bag_q = lin(M, self.W_q.weight, self.W_q.bias)                       # [1, J]
        conf_k = lin(cf, self.W_k.weight, self.W_k.bias)                     # [n_conf, J]
        scores = mm(bag_q, conf_k, True) * (1.0 / float(conf_k.shape[1]) ** 0.5)      # [1, n_conf] = (conf_k bag_q^T)^T / sqrt(J)
        dA = softmax_rows(scores.contiguous())                               # softmax over the dictionary
        conf_feats = mm(dA, cf, False)                                       # [1, Dc]
        if self.confounder_merge == 'cat':
            M = torch.cat((M, conf_feats), dim=1)
        elif self.confounder_merge == 'add':
            M = M + conf_feats
        else:
            M = M - conf_feats
        return lin(M, self.classifier.weight, self.classifier.bias), M, dA.t()        # deconf_A [n_conf, 1] as the reference

    def forward(self, x):   # x [1,N,D_feat] -> (Y_prob [1,C], M [1,Di], A [1,N] softmax over N)   (ibmil.py:69-113)
        xb = self._bag(x)
        a = self.attention
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: op by op through acmil_amd.autograd (all three outputs are differentiable)
            h = AG.linear(xb.float(), self.dimreduction.fc1.weight, None, relu=True, precision=self.precision)
            A = AG.softmax_rows(AG.gated_scores(h, a.attention_V[0].weight, a.attention_V[0].bias, a.attention_U[0].weight,
                                                a.attention_U[0].bias, a.attention_weights.weight, a.attention_weights.bias,
                                                self.precision).contiguous())
            M = AG.matmul(A, h)
            if self.confounder_path:
                return self._deconfound(M, lambda t, w, b: AG.linear(t, w, b, precision="fp32"),
                                        lambda p, q, tb: AG.matmul(p, q, trans_b=tb, precision="fp32"), AG.softmax_rows)
            return AG.linear(M, self.classifier.fc.weight, self.classifier.fc.bias, precision="fp32"), M, A
        if self.confounder_path:
            # eval with the deconfounding branch: its classifier has another input width than the fused kernel's packed head,
            # so the bag feature comes from the score + pooling kernels and the O(dictionary) tail runs as small HIP GEMMs
            x32 = xb if xb.dtype == torch.float32 else xb.float()
            h = ops.gemm(x32, self.dimreduction.fc1.weight.detach(), trans_b=True, act=1, precision=self.precision)
            A = ops.gated_scores(h, a.attention_V[0].weight.detach(), a.attention_V[0].bias.detach(), a.attention_U[0].weight.detach(),
                                 a.attention_U[0].bias.detach(), a.attention_weights.weight.detach(), a.attention_weights.bias.detach(),
                                 self.precision)
            M = ops.attn_pool(h, A)
            return self._deconfound(M, lambda t, w, b: ops.gemm(t.contiguous(), w.detach(), trans_b=True, bias=b.detach()),
                                    lambda p, q, tb: ops.gemm(p.contiguous(), q.detach().contiguous(), trans_b=tb), ops.softmax_rows)
        packed, dims = self._packed()
        out = ops.ga_forward(xb, packed, dims, self.precision, want_afeat=True)
        return out["sub_preds"], out["afeat"], ops.softmax_rows(out["A_out"])
