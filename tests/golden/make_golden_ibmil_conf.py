"""Fixtures for IBMIL WITH the confounder branch (ibmil.py:45-67, :93-107) from the REAL reference (dev container only): eval
forward and one forward + backward, confounder dictionary learnable (c_learn) with merge 'cat', fixed with merge 'sub'."""
import os, sys, tempfile
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
import torch.nn.functional as F
from architecture.ibmil import IBMIL

OUT = os.path.dirname(os.path.abspath(__file__))
tmp = tempfile.mkdtemp()
rng = np.random.RandomState(5)
paths = []
for i, k in enumerate((8, 4)):                      # two cluster files, concatenated by the module: 12 confounders x D_inner
    p = os.path.join(tmp, "conf%d.npy" % i)
    np.save(p, rng.randn(k, 128).astype(np.float32) * 0.5)
    paths.append(p)

for tag, merge, learn in (("cat_learn", "cat", True), ("sub_fixed", "sub", False)):
    class Conf:
        D_feat, D_inner, n_class, c_path, c_learn = 384, 128, 3, paths, learn
    torch.manual_seed(50)
    m = IBMIL(Conf, confounder_merge=merge).train()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    x = torch.randn(1, 700, 384, generator=torch.Generator().manual_seed(700))
    y, mm, da = m(x)
    loss = F.cross_entropy(y, torch.tensor([2])) + 0.01 * mm.sum() + 3.0 * (da * da).sum()
    loss.backward()
    name = "train_ibmil_conf_%s_n700_d384_c3" % tag
    wname = "weights_" + name
    np.savez(os.path.join(OUT, wname + ".npz"), **{k: v.detach().numpy().copy() for k, v in m.state_dict().items()})
    grads = {"grad." + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    np.savez(os.path.join(OUT, name + ".npz"), weights=np.array(wname), x=x.numpy(), Y_prob=y.detach().numpy(), M=mm.detach().numpy(),
             deconf_A=da.detach().numpy(), loss=np.array(float(loss.detach())), **grads)
    print(name, tuple(y.shape), tuple(mm.shape), tuple(da.shape), "loss %.5f" % float(loss), sorted(m.state_dict().keys()))
