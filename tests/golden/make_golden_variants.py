"""Golden vectors for the N4 modules from the REAL reference (dev container only; /root/reference imported, never copied):
Attention_with_Classifier (DTFD), IBMIL without confounder, CLAM_SB small / big, eval mode.  Module parameters are
perturbed after construction (biases are zero-initialised in CLAM) so that every term of the forward is exercised."""
import os, sys
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
from architecture.Attention import Attention_with_Classifier
from architecture.ibmil import IBMIL
from architecture.clam import CLAM_SB

OUT = os.path.dirname(os.path.abspath(__file__))


def perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)


def save(name, model, x, outs):
    wname = "weights_" + name
    np.savez(os.path.join(OUT, wname + ".npz"), **{k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    np.savez(os.path.join(OUT, name + ".npz"), weights=np.array(wname), x=x.numpy(), **{k: v.detach().numpy() for k, v in outs.items()})
    print(name, {k: tuple(v.shape) for k, v in outs.items()})


torch.manual_seed(11)
m = Attention_with_Classifier(L=256, D=128, K=3, num_cls=4).eval(); perturb(m, 1)
x = torch.randn(700, 256, generator=torch.Generator().manual_seed(700)).relu()
with torch.no_grad():
    save("variants_dtfd_n700_l256_k3_c4", m, x, {"pred": m(x), "A_norm": m.attention(x), "A_raw": m.attention(x, isNorm=False)})


class Conf:
    D_feat, D_inner, n_class, c_path = 384, 128, 3, None
torch.manual_seed(12)
m = IBMIL(Conf).eval(); perturb(m, 2)
x = torch.randn(1, 900, 384, generator=torch.Generator().manual_seed(900))
with torch.no_grad():
    y, mm, a = m(x)
save("variants_ibmil_n900_d384_c3", m, x, {"Y_prob": y, "M": mm, "A": a})

for size_arg, d, di in (("small", 384, 128), ("big", 256, 128)):
    class Conf2:
        D_feat, D_inner, n_class = d, di, 2
    torch.manual_seed(13)
    m = CLAM_SB(Conf2, size_arg=size_arg).eval(); perturb(m, 3)
    x = torch.randn(1, 600, d, generator=torch.Generator().manual_seed(600))
    with torch.no_grad():
        save("variants_clam_%s_n600_d%d" % (size_arg, d), m, x, {"logits": m(x), "A_raw": m(x, attention_only=True)})
