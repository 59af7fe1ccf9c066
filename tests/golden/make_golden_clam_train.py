"""Gradient fixtures for CLAM_SB's training forward (instance-level clustering loss included) from the REAL reference (dev
container only): dropout=False so that the step is deterministic; binary task (in-the-class branch only) and a 3-class
task (subtyping: out-of-the-class branches too).  One forward + backward each."""
import os, sys
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from architecture.clam import CLAM_SB

OUT = os.path.dirname(os.path.abspath(__file__))


def perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)


for tag, ncls, size_arg, d, di, n, label in (("bin", 2, "small", 384, 128, 600, 1), ("sub", 3, "small", 512, 256, 500, 2)):
    class Conf:
        D_feat, D_inner, n_class = d, di, ncls
    torch.manual_seed(40 + ncls)
    m = CLAM_SB(Conf, size_arg=size_arg, k_sample=8, dropout=False, instance_loss_fn=nn.CrossEntropyLoss()).train()
    perturb(m, ncls)
    x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(n))
    y = torch.tensor([label])
    logits, inst_loss = m(x, label=y, instance_eval=True)
    loss = 0.7 * F.cross_entropy(logits, y) + 0.3 * inst_loss          # bag_weight = 0.7 as in the CLAM trainer
    loss.backward()
    name = "train_clam_%s_n%d_d%d_c%d" % (tag, n, d, ncls)
    wname = "weights_" + name
    np.savez(os.path.join(OUT, wname + ".npz"), **{k: v.detach().numpy().copy() for k, v in m.state_dict().items()})
    grads = {"grad." + k: (p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in m.named_parameters()}
    np.savez(os.path.join(OUT, name + ".npz"), weights=np.array(wname), x=x.numpy(), label=y.numpy(), logits=logits.detach().numpy(),
             inst_loss=np.array(float(inst_loss)), loss=np.array(float(loss)), **grads)
    print(name, "loss %.5f inst %.5f" % (float(loss), float(inst_loss)), "max|grad| %.3e" % max(np.abs(g).max() for g in grads.values()))
