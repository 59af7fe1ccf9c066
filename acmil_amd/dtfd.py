"""DTFD-MIL's double-tier loop on the MI355X ops: this repo's counterpart of `Step3_WSI_classification_DTFD.py`
(train_one_epoch :61-160, evaluate :163-237, main :306-324 of the reference).

Tier 1 = DimReduction + Attention_Gated + Classifier_1fc on `numGroup` random pseudo-bags of a slide; each pseudo-bag hands
`total_instance // numGroup` highest- (and lowest-) scoring patches -- ranked by the class-activation map `get_cam_1d`
(utils/utils.py:48-51) -- or its attention feature to tier 2 = Attention_with_Classifier.  Two Adam optimizers, gradient-norm
clipping per module, as the reference.

Arithmetic: every O(n) product / reduction runs as a HIP kernel (acmil_amd.autograd: Linear GEMMs, gate kernel, row softmax,
pooling GEMM, each with a HIP backward; acmil_amd.ops for the no-gradient CAM ranking and the evaluation).  torch supplies the
random permutation, the sort of the n CAM scores, the gather of the 2 x instance_per_group selected rows, the cross-entropy
of numGroup logits, clip_grad_norm_ and Adam -- host-side glue on small tensors.

One deliberate difference, without effect on any parameter update: the reference back-propagates the tier-2 loss into the
tier-1 modules as well (`slide_pseudo_feat` keeps its graph, :129,143-146) but never uses those gradients -- optimizer1 only
holds the tier-2 parameters and optimizer0.zero_grad() clears them before the next tier-1 backward.  Here the pseudo-bag
features are detached before tier 2.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import autograd as AG
from . import ops
from .architecture.Attention import Attention_Gated, Attention_with_Classifier
from .architecture.network import Classifier_1fc, DimReduction


def build_dtfd(conf, precision: str = "f16x3"):
    """(classifier, attention, dimReduction, attCls) as Step3_WSI_classification_DTFD.py:306-310."""
    classifier = Classifier_1fc(conf.D_inner, conf.n_class, 0)
    attention = Attention_Gated(conf.D_inner, precision=precision)
    dim_reduction = DimReduction(conf.D_feat, conf.D_inner)
    att_cls = Attention_with_Classifier(L=conf.D_inner, num_cls=conf.n_class, droprate=0, precision=precision)
    return classifier, attention, dim_reduction, att_cls


def make_optimizers(classifier, attention, dim_reduction, att_cls, conf):
    """torch.optim.Adam (L2 weight decay, not AdamW) over tier 1 and over tier 2, :314-320."""
    tier1 = list(classifier.parameters()) + list(attention.parameters()) + list(dim_reduction.parameters())
    return (torch.optim.Adam(tier1, lr=conf.lr, weight_decay=conf.wd),
            torch.optim.Adam(att_cls.parameters(), lr=conf.lr, weight_decay=conf.wd))


def _cam_rank(feat: torch.Tensor, att: torch.Tensor, wc: torch.Tensor) -> torch.Tensor:
    """Patch order by the class-activation score of the LAST class, descending (:108-112, :200-204):
    get_cam_1d(classifier, att[:, None] * feat) = att[n] * (feat[n] . Wc[c]) ; softmax over classes ; sort."""
    logits = ops.gemm(feat.detach().contiguous(), wc.detach(), trans_b=True) * att.detach().reshape(-1, 1)
    return torch.sort(torch.softmax(logits, dim=1)[:, -1], descending=True)[1]


def _select(feat, att_feat, order, ipg: int, distill: str):
    if distill == "MaxMinS":
        return feat.index_select(0, torch.cat([order[:ipg], order[-ipg:]]))
    if distill == "MaxS":
        return feat.index_select(0, order[:ipg])
    if distill == "AFS":
        return att_feat
    raise ValueError("distill must be MaxMinS, MaxS or AFS")


def train_step(classifier, attention, dim_reduction, att_cls, x: torch.Tensor, label: torch.Tensor, optimizer0, optimizer1, conf,
               perm: Optional[torch.Tensor] = None, distill: str = "MaxMinS", precision: str = "f16x3"):
    """One slide of the double-tier training loop (:82-149).  x [N, D_feat] on the GPU (fp32 / fp16), label [1];
    perm: the random permutation of the patches (drawn here when omitted).  Returns (loss0, loss1) as device scalars."""
    n_group = conf.numGroup
    ipg = conf.total_instance // n_group
    if perm is None:
        perm = torch.randperm(x.shape[0], device=x.device)
    pseudo, preds = [], []
    for idx in torch.tensor_split(perm.to(x.device), n_group):
        sub = x.index_select(0, idx).float()
        mid = AG.linear(sub, dim_reduction.fc1.weight, None, relu=True, precision=precision)       # DimReduction, network.py:49-57
        att = attention(mid)                                                                       # [1, n], softmax over the pseudo-bag
        att_feat = AG.matmul(att, mid)                                                             # sum_n att[n] mid[n]  -> [1, Di]
        preds.append(AG.linear(att_feat, classifier.fc.weight, classifier.fc.bias, precision="fp32"))
        order = _cam_rank(mid, att[0], classifier.fc.weight)
        pseudo.append(_select(mid, att_feat, order, ipg, distill))
    sub_preds = torch.cat(preds, 0)
    loss0 = F.cross_entropy(sub_preds, label.repeat(n_group))
    optimizer0.zero_grad()
    loss0.backward()
    for m in (dim_reduction, attention, classifier):
        torch.nn.utils.clip_grad_norm_(m.parameters(), conf.grad_clipping)
    optimizer0.step()
    slide_feat = torch.cat(pseudo, 0).detach()                # see the module docstring
    loss1 = F.cross_entropy(att_cls(slide_feat), label)
    optimizer1.zero_grad()
    loss1.backward()
    torch.nn.utils.clip_grad_norm_(att_cls.parameters(), conf.grad_clipping)
    optimizer1.step()
    return loss0.detach(), loss1.detach()


@torch.no_grad()
def predict(classifier, attention, dim_reduction, att_cls, x: torch.Tensor, conf, perm: Optional[torch.Tensor] = None,
            distill: str = "MaxMinS", precision: str = "f16x3") -> torch.Tensor:
    """Slide-level class probabilities [1, C] as the reference's evaluate() forms them (:178-214): raw attention scores over the
    whole bag, softmax inside each random pseudo-bag, CAM ranking, tier 2 on the selected features."""
    n_group = conf.numGroup
    ipg = conf.total_instance // n_group
    x32 = x if x.dtype == torch.float32 else x.float()
    mid = ops.gemm(x32.contiguous(), dim_reduction.fc1.weight, trans_b=True, act=1, precision=precision)
    a = attention
    raw = ops.gated_scores(mid, a.attention_V[0].weight, a.attention_V[0].bias, a.attention_U[0].weight, a.attention_U[0].bias,
                           a.attention_weights.weight, a.attention_weights.bias, precision)[0]               # [N]
    if perm is None:
        perm = torch.randperm(x.shape[0], device=x.device)
    feats = []
    for idx in torch.tensor_split(perm.to(x.device), n_group):
        sub_mid = mid.index_select(0, idx)
        att = ops.softmax_rows(raw.index_select(0, idx).reshape(1, -1).contiguous())
        att_feat = ops.gemm(att, sub_mid)
        order = _cam_rank(sub_mid, att[0], classifier.fc.weight)
        feats.append(_select(sub_mid, att_feat, order, ipg, distill))
    return torch.softmax(att_cls(torch.cat(feats, 0)), dim=1)


def train_one_epoch(modules: Sequence, data, optimizers, device, epoch: int, conf, distill: str = "MaxMinS"):
    from .train import adjust_learning_rate, epoch_order
    classifier, attention, dim_reduction, att_cls = modules
    for m in modules:
        m.train()
    order = epoch_order(len(data), epoch, conf.seed, True, 0, 1)
    tot0 = torch.zeros((), device=device); tot1 = torch.zeros((), device=device)
    for it, i in enumerate(order):
        item = data[i]
        for opt in optimizers:
            adjust_learning_rate(opt, epoch + it / len(order), conf)
        x = torch.as_tensor(item["input"]).to(device)
        y = torch.tensor([int(item["label"])], device=device)
        l0, l1 = train_step(classifier, attention, dim_reduction, att_cls, x, y, optimizers[0], optimizers[1], conf, distill=distill,
                            precision=getattr(conf, "precision", "f16x3"))
        tot0 += l0; tot1 += l1
    n = max(1, len(order))
    return {"loss0": float(tot0) / n, "loss1": float(tot1) / n}


@torch.no_grad()
def evaluate(modules: Sequence, data, device, conf, distill: str = "MaxMinS"):
    """(auroc, acc, f1, loss) as :163-237 (the loss is the reference's CE on the softmaxed prediction)."""
    from .train import micro_f1, multiclass_auroc
    classifier, attention, dim_reduction, att_cls = modules
    for m in modules:
        m.eval()
    probs, labels = [], []
    for i in range(len(data)):
        item = data[i]
        probs.append(predict(classifier, attention, dim_reduction, att_cls, torch.as_tensor(item["input"]).to(device), conf,
                             distill=distill, precision=getattr(conf, "precision", "f16x3")))
        labels.append(int(item["label"]))
    prob = torch.cat(probs, 0).float().cpu()
    target = torch.tensor(labels)
    loss = float(F.cross_entropy(prob, target))
    acc = float((prob.argmax(1) == target).float().mean()) * 100.0
    return multiclass_auroc(prob, target, conf.n_class), acc, micro_f1(prob, target), loss


def main(argv=None):
    import argparse
    from . import train as T
    p = argparse.ArgumentParser("DTFD-MIL double-tier training (MI355X ops)")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--numGroup", type=int, default=4)
    p.add_argument("--total_instance", type=int, default=4)
    p.add_argument("--grad_clipping", type=float, default=5.0)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--wd", type=float, default=1e-5)
    p.add_argument("--train_epoch", type=int, default=2)
    p.add_argument("--n_class", type=int, default=2)
    p.add_argument("--D_feat", type=int, default=384)
    p.add_argument("--D_inner", type=int, default=128)
    p.add_argument("--distill", default="MaxMinS", choices=["MaxMinS", "MaxS", "AFS"])
    p.add_argument("--precision", default="f16x3", choices=["f16x3", "fp32"])
    p.add_argument("--synthetic_slides", type=int, default=16)
    p.add_argument("--synthetic_patches", type=int, default=600)
    a = p.parse_args(argv)
    conf = T.Struct(**vars(a), warmup_epoch=0, min_lr=0)
    dev = torch.device("cuda", 0)
    T.set_seed(conf.seed)
    modules = [m.to(dev) for m in build_dtfd(conf, conf.precision)]
    opts = make_optimizers(*modules, conf)
    train = T.SyntheticBags(a.synthetic_slides, a.synthetic_patches, a.D_feat, a.n_class, seed=conf.seed)
    val = T.SyntheticBags(max(4, a.synthetic_slides // 4), a.synthetic_patches, a.D_feat, a.n_class, seed=conf.seed + 1)
    best = {"epoch": -1, "val_auc": 0.0, "val_f1": 0.0}
    for epoch in range(conf.train_epoch):
        stats = train_one_epoch(modules, train, opts, dev, epoch, conf, a.distill)
        auc, acc, f1, loss = evaluate(modules, val, dev, conf, a.distill)
        print("epoch %d loss0 %.4f loss1 %.4f | val auc %.3f acc %.1f f1 %.3f loss %.3f" % (epoch, stats["loss0"], stats["loss1"], auc, acc, f1, loss))
        if f1 + auc > best["val_f1"] + best["val_auc"]:
            best = {"epoch": epoch, "val_auc": auc, "val_f1": f1}
    print("Results on best epoch:", best)
    return best


if __name__ == "__main__":
    main()
