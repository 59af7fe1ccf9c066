"""One slide through the REAL reference's double-tier DTFD training loop (Step3_WSI_classification_DTFD.py:61-160, dev container
only): parameters before, the patch permutation it drew, both losses and every parameter after the two Adam steps.
The loop is driven with a one-item loader; MetricLogger / wandb are the reference's own (wandb mocked, disabled)."""
import os, sys
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "timm.utils", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets", "yaml"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
from torch import nn
import Step3_WSI_classification_DTFD as S
from architecture.Attention import Attention_Gated as Attention, Attention_with_Classifier
from architecture.network import Classifier_1fc, DimReduction

OUT = os.path.dirname(os.path.abspath(__file__))


class Conf:
    D_feat, D_inner, n_class = 384, 128, 3
    numGroup, total_instance, grad_clipping = 4, 8, 5.0
    lr, wd, min_lr, warmup_epoch, train_epoch = 1e-3, 1e-5, 0.0, 0, 10
    wandb_mode = "disabled"


torch.manual_seed(60)
classifier = Classifier_1fc(Conf.D_inner, Conf.n_class, 0)
attention = Attention(Conf.D_inner)
dimReduction = DimReduction(Conf.D_feat, Conf.D_inner)
attCls = Attention_with_Classifier(L=Conf.D_inner, num_cls=Conf.n_class, droprate=0)
mods = {"classifier": classifier, "attention": attention, "dimReduction": dimReduction, "attCls": attCls}
before = {"%s.%s" % (m, k): v.detach().numpy().copy() for m, mod in mods.items() for k, v in mod.state_dict().items()}
opt0 = torch.optim.Adam(list(classifier.parameters()) + list(attention.parameters()) + list(dimReduction.parameters()), lr=Conf.lr, weight_decay=Conf.wd)
opt1 = torch.optim.Adam(attCls.parameters(), lr=Conf.lr, weight_decay=Conf.wd)
x = torch.randn(1, 900, Conf.D_feat, generator=torch.Generator().manual_seed(900))
label = torch.tensor([2])
loader = [{"input": x, "label": label}]
torch.manual_seed(61)
perm = torch.randperm(900)                 # the permutation train_one_epoch will draw first from this RNG state
torch.manual_seed(61)
losses = {}
real_update = S.MetricLogger.update
def spy(self, **kw):
    for k, v in kw.items():
        if k.startswith("loss"):
            losses[k] = float(v)
    return real_update(self, **kw)
S.MetricLogger.update = spy
clip_calls = []
real_clip = torch.nn.utils.clip_grad_norm_
def clip_spy(params, max_norm, *a, **k):
    params = list(params)
    clip_calls.append([p.grad.detach().clone().numpy() for p in params])      # gradients as backward left them, before clipping
    return real_clip(params, max_norm, *a, **k)
torch.nn.utils.clip_grad_norm_ = clip_spy
S.train_one_epoch(classifier, attention, dimReduction, attCls, nn.CrossEntropyLoss(), loader, opt0, opt1, torch.device("cpu"), 0, Conf)
after = {"%s.%s" % (m, k): v.detach().numpy().copy() for m, mod in mods.items() for k, v in mod.state_dict().items()}
grads = {}
for mname, mod, rec in zip(("dimReduction", "attention", "classifier", "attCls"), (dimReduction, attention, classifier, attCls), clip_calls):
    for (k, _), g in zip(mod.named_parameters(), rec):
        grads["grad.%s.%s" % (mname, k)] = g
np.savez(os.path.join(OUT, "train_dtfd_step_n900_d384_c3.npz"), x=x[0].numpy(), label=label.numpy(), perm=perm.numpy(),
         loss0=np.array(losses["loss0"]), loss1=np.array(losses["loss1"]), **grads,
         **{"before." + k: v for k, v in before.items()}, **{"after." + k: v for k, v in after.items()})
print("loss0 %.5f loss1 %.5f" % (losses["loss0"], losses["loss1"]), len(before), "tensors; max |delta| %.3e" % max(np.abs(after[k] - before[k]).max() for k in before))
