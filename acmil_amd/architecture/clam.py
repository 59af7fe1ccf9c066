"""CLAM_SB with the gated attention net: same ctor / parameter names (`attention_net.0`, `attention_net.3.attention_{a,b}.0`,
`attention_net.3.attention_c`, `classifiers`, `instance_classifiers.{i}`) as the reference's `architecture/clam.py:83-197`.
Eval forward (`logits`, `attention_only`, `return_features`): fc WITH bias + ReLU as one GEMM, gated scores with a free
attention width (128 "small" / 384 "big"), softmax-pooling, bag classifier.  Instance-level evaluation (`instance_eval=True`,
clam.py:166-188: a training-time loss on top-k patches) is not built and raises."""
import torch
import torch.nn as nn

from .. import ops


class Attn_Net_Gated(nn.Module):
    def __init__(self, L=1024, D=256, dropout=False, n_classes=1):
        super().__init__()
        a, b = [nn.Linear(L, D), nn.Tanh()], [nn.Linear(L, D), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(0.25)); b.append(nn.Dropout(0.25))
        self.attention_a, self.attention_b = nn.Sequential(*a), nn.Sequential(*b)
        self.attention_c = nn.Linear(D, n_classes)


class CLAM_SB(nn.Module):
    def __init__(self, conf, gate=True, size_arg="small", k_sample=8, dropout=True, instance_loss_fn=None, *, precision="f16x3"):
        super().__init__()
        if not gate:
            raise NotImplementedError("acmil_amd: CLAM_SB is built with the gated attention net (gate=True, the reference default)")
        n_classes = conf.n_class
        self.size_dict = {"small": [conf.D_feat, conf.D_inner, 128], "big": [conf.D_feat, 512, 384]}
        size = self.size_dict[size_arg]
        fc = [nn.Linear(size[0], size[1]), nn.ReLU()]
        if dropout:
            fc.append(nn.Dropout(0.25))
        fc.append(Attn_Net_Gated(L=size[1], D=size[2], dropout=dropout, n_classes=1))
        self.attention_net = nn.Sequential(*fc)
        self.classifiers = nn.Linear(size[1], n_classes)
        self.instance_classifiers = nn.ModuleList([nn.Linear(size[1], 2) for _ in range(n_classes)])
        self.k_sample, self.instance_loss_fn, self.n_classes = k_sample, instance_loss_fn or nn.CrossEntropyLoss(), n_classes
        self.subtyping = conf.n_class > 2
        for m in self.modules():            # utils/utils.py:519-523 initialize_weights
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)
                m.bias.data.zero_()
        self.precision = precision

    @torch.no_grad()
    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        if instance_eval:
            raise NotImplementedError("acmil_amd: CLAM instance-level evaluation (clam.py:166-188) is not built")
        if self.training:
            raise NotImplementedError("acmil_amd: CLAM_SB has an eval forward only (train mode draws Dropout(0.25) masks)")
        x = h[0]
        if not x.is_cuda:
            raise RuntimeError("acmil_amd: CLAM_SB runs on an MI355X only (no CPU fallback)")
        fc1, net = self.attention_net[0], self.attention_net[-1]
        h1 = ops.gemm(x.float().contiguous(), fc1.weight, trans_b=True, bias=fc1.bias, act=1, precision=self.precision)
        a, b, c = net.attention_a[0], net.attention_b[0], net.attention_c
        A = ops.gated_scores(h1, a.weight, a.bias, b.weight, b.bias, c.weight, c.bias, self.precision)   # [1, N] raw
        if attention_only:
            return A
        M = ops.attn_pool(h1, A)
        logits = ops.gemm(M, self.classifiers.weight, trans_b=True, bias=self.classifiers.bias)
        return (logits, M) if return_features else logits
