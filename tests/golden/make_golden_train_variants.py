"""Gradient fixtures for the trainable N3 / N4 modules from the REAL reference (dev container only): ACMIL_MHA (dropout p = 0,
no mask-drop, harness-side), Attention_with_Classifier (DTFD) and IBMIL without confounder; one forward + backward each."""
import os, sys
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
import torch.nn.functional as F
from architecture.transformer import ACMIL_MHA
from architecture.Attention import Attention_with_Classifier
from architecture.ibmil import IBMIL

OUT = os.path.dirname(os.path.abspath(__file__))


def perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or "layer_norm" in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)


def save(name, model, x, outs):
    wname = "weights_" + name
    np.savez(os.path.join(OUT, wname + ".npz"), **{k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    np.savez(os.path.join(OUT, name + ".npz"), weights=np.array(wname), x=x.detach().numpy(),
             **{k: v.detach().numpy() for k, v in outs.items()}, **{"grad." + k: p.grad.numpy().copy() for k, p in model.named_parameters()})
    print(name, {k: tuple(v.shape) for k, v in outs.items()}, "max|grad| %.3e" % max(p.grad.abs().max().item() for p in model.parameters()))


class Conf:
    D_feat, D_inner, n_class, n_token, c_path = 384, 128, 2, 3, None


torch.manual_seed(31)
m = ACMIL_MHA(Conf, n_token=3, n_masked_patch=0, mask_drop=0.0).train(); perturb(m, 1)
for att in list(m.sub_attention) + [m.bag_attention]:
    att.dropout.p = 0.0
with torch.no_grad():
    m.q.copy_(torch.randn(m.q.shape, generator=torch.Generator().manual_seed(3)) * 0.5)
x = torch.randn(1, 400, 384, generator=torch.Generator().manual_seed(400))
sub, slide, attns = m(x)
label = torch.tensor([1])
loss = F.cross_entropy(sub, label.repeat(3)) + F.cross_entropy(slide, label) + 0.5 * attns.pow(2).mean()
loss.backward()
save("train_mha_n400_d384_k3_c2", m, x, {"sub_preds": sub, "slide_pred": slide, "attns": attns, "loss": loss})

torch.manual_seed(32)
m = Attention_with_Classifier(L=256, D=128, K=3, num_cls=4).train(); perturb(m, 2)
x = torch.randn(500, 256, generator=torch.Generator().manual_seed(500)).relu()
pred = m(x)
loss = F.cross_entropy(pred, torch.tensor([0, 3, 1]))
loss.backward()
save("train_dtfd_n500_l256_k3_c4", m, x, {"pred": pred, "loss": loss})

torch.manual_seed(33)
m = IBMIL(Conf).train(); perturb(m, 3)
x = torch.randn(1, 600, 384, generator=torch.Generator().manual_seed(600))
y, mm, a = m(x)
loss = F.cross_entropy(y, label) + 0.01 * mm.sum() + 10.0 * (a * a).sum()
loss.backward()
save("train_ibmil_n600_d384_c2", m, x, {"Y_prob": y, "M": mm, "A": a, "loss": loss})
