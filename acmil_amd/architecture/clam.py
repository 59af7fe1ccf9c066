"""CLAM_SB with the gated attention net: same ctor / parameter names (`attention_net.0`, `attention_net.3.attention_{a,b}.0`,
`attention_net.3.attention_c`, `classifiers`, `instance_classifiers.{i}`) as the reference's `architecture/clam.py:83-197`.
Eval forward (`logits`, `attention_only`, `return_features`): fc WITH bias + ReLU as one GEMM, gated scores with a free
attention width (128 "small" / 384 "big"), softmax-pooling, bag classifier -- fused HIP ops under torch.no_grad().
With gradients enabled the same mathematics runs op by op through acmil_amd.autograd (Linear GEMMs, gate kernel, row softmax,
pooling GEMM, each with a HIP backward), including the instance-level clustering loss of `instance_eval=True`
(clam.py:130-157, :166-188: the k_sample highest / lowest-attention patches against the per-class instance classifiers;
the top-k is acmil_stkim_select) and the Dropout(0.25) layers of the `dropout=True` configuration (masks drawn by torch)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as AG
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

    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        x = h[0]
        if not x.is_cuda:
            raise RuntimeError("acmil_amd: CLAM_SB runs on an MI355X only (no CPU fallback)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(x, label, instance_eval, return_features, attention_only)
        if instance_eval:
            raise NotImplementedError("acmil_amd: CLAM instance-level evaluation needs gradients enabled (it is a training loss)")
        if self.training and any(isinstance(m, nn.Dropout) for m in self.attention_net.modules()):
            raise NotImplementedError("acmil_amd: train-mode CLAM_SB draws Dropout(0.25) masks; call it with gradients enabled or use .eval()")
        with torch.no_grad():
            fc1, net = self.attention_net[0], self.attention_net[-1]
            h1 = ops.gemm(x.float().contiguous(), fc1.weight, trans_b=True, bias=fc1.bias, act=1, precision=self.precision)
            a, b, c = net.attention_a[0], net.attention_b[0], net.attention_c
            A = ops.gated_scores(h1, a.weight, a.bias, b.weight, b.bias, c.weight, c.bias, self.precision)   # [1, N] raw
            if attention_only:
                return A
            M = ops.attn_pool(h1, A)
            logits = ops.gemm(M, self.classifiers.weight, trans_b=True, bias=self.classifiers.bias)
            return (logits, M) if return_features else logits

    # ---- training: clam.py:159-197 op by op, every product and reduction a HIP kernel with a HIP backward
    def _inst_loss(self, P, h1, classifier, positive_and_negative):
        """inst_eval (in-the-class: k_sample top + k_sample bottom patches, targets 1 / 0) or inst_eval_out (top patches, target 0)."""
        k = self.k_sample
        top_p, _ = ops.stkim_select(P.detach().contiguous(), k, 0, None)                 # [1, k] sorted by descending attention
        ids, targets = top_p[0], torch.ones(k, dtype=torch.long, device=h1.device)
        if positive_and_negative:
            top_n, _ = ops.stkim_select((-P.detach()).contiguous(), k, 0, None)
            ids = torch.cat([ids, top_n[0]])
            targets = torch.cat([targets, torch.zeros(k, dtype=torch.long, device=h1.device)])
        else:
            targets = torch.zeros(k, dtype=torch.long, device=h1.device)
        inst = h1.index_select(0, ids)                                                    # 8 - 16 rows: torch gather / scatter-add
        logits = AG.linear(inst, classifier.weight, classifier.bias, precision="fp32")
        return self.instance_loss_fn(logits, targets)

    def _forward_train(self, x, label, instance_eval, return_features, attention_only):
        fc1, net = self.attention_net[0], self.attention_net[-1]
        drop = self.training and any(isinstance(m, nn.Dropout) for m in self.attention_net)
        drop_gate = self.training and len(net.attention_a) > 2
        h1 = AG.linear(x.float().contiguous(), fc1.weight, fc1.bias, relu=True, precision=self.precision)
        if drop:
            h1 = F.dropout(h1, 0.25, True)
        a, b, c = net.attention_a[0], net.attention_b[0], net.attention_c
        G = torch.cat([AG.linear(h1, a.weight, a.bias, precision=self.precision),
                       AG.linear(h1, b.weight, b.bias, precision=self.precision)], dim=1)
        y = AG._Gate.apply(G)                                  # tanh(.) * sigmoid(.)
        if drop_gate:                                          # Dropout(0.25) on each branch = two independent masks on the product
            y = y * F.dropout(torch.ones_like(y), 0.25, True) * F.dropout(torch.ones_like(y), 0.25, True)
        A = AG.linear(y, c.weight, c.bias, precision="fp32").t()          # [1, N] raw scores
        if attention_only:
            return A
        P = AG.softmax_rows(A.contiguous())
        total_inst_loss = 0.0
        if instance_eval:
            inst_labels = F.one_hot(label, num_classes=self.n_classes).reshape(-1).tolist()
            for i, clf in enumerate(self.instance_classifiers):
                if inst_labels[i] == 1:
                    total_inst_loss = total_inst_loss + self._inst_loss(P, h1, clf, True)
                elif self.subtyping:
                    total_inst_loss = total_inst_loss + self._inst_loss(P, h1, clf, False)
            if self.subtyping:
                total_inst_loss = total_inst_loss / len(self.instance_classifiers)
        M = AG.matmul(P, h1)
        logits = AG.linear(M, self.classifiers.weight, self.classifiers.bias, precision="fp32")
        if instance_eval:
            return logits, total_inst_loss
        return (logits, M) if return_features else logits
