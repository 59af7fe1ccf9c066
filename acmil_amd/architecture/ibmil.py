"""IBMIL (interventional bag MIL) without the confounder branch: same ctor / parameter names / return contract as the
reference's `architecture/ibmil.py:38-113`.  Structurally it is ABMIL (DimReduction + Attention_Gated(Di, 128, 1) +
Classifier_1fc), so it runs on the fully fused forward kernel; only the returned attention map needs one extra row softmax."""
import torch

from .. import autograd as AG
from .. import ops
from .network import Classifier_1fc, DimReduction
from .transformer import Attention_Gated, _GatedBase


class IBMIL(_GatedBase):
    def __init__(self, conf, confounder_dim=128, confounder_merge='cat', *, precision="f16x3"):
        super().__init__()
        assert confounder_merge in ['cat', 'add', 'sub']
        if getattr(conf, "c_path", None):
            raise NotImplementedError("acmil_amd: the IBMIL confounder branch (ibmil.py:45-67,93-107) is not built")
        self.confounder_merge, self.confounder_path = confounder_merge, None
        self.dimreduction = DimReduction(conf.D_feat, conf.D_inner)
        self.attention = Attention_Gated(conf.D_inner, 128, 1)
        self.classifier = Classifier_1fc(conf.D_inner, conf.n_class, 0)
        self.precision = precision

    def _heads(self):
        return [self.classifier.fc.weight], [self.classifier.fc.bias], None, None

    def forward(self, x):   # x [1,N,D_feat] -> (Y_prob [1,C], M [1,Di], A [1,N] softmax over N)   (ibmil.py:69-113)
        xb = self._bag(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: op by op through acmil_amd.autograd (all three outputs are differentiable)
            a = self.attention
            h = AG.linear(xb.float(), self.dimreduction.fc1.weight, None, relu=True, precision=self.precision)
            A = AG.softmax_rows(AG.gated_scores(h, a.attention_V[0].weight, a.attention_V[0].bias, a.attention_U[0].weight,
                                                a.attention_U[0].bias, a.attention_weights.weight, a.attention_weights.bias,
                                                self.precision).contiguous())
            M = AG.matmul(A, h)
            return AG.linear(M, self.classifier.fc.weight, self.classifier.fc.bias, precision="fp32"), M, A
        packed, dims = self._packed()
        out = ops.ga_forward(xb, packed, dims, self.precision, want_afeat=True)
        return out["sub_preds"], out["afeat"], ops.softmax_rows(out["A_out"])
