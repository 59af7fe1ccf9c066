"""Gradient fixture for the TransMIL training path from the REAL reference (dev container only): one forward + backward of
`architecture.transMIL.TransMIL` in train mode with the to_out Dropout probability set to 0 (harness-side, so the result is
deterministic), loss = CrossEntropy(logits, label) as in engine.py:19-21.  Same shims as make_golden_transmil.py."""
import os, sys
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
import architecture.nystrom_attention as fork
sys.modules["nystrom_attention"] = fork
torch.Tensor.cuda = lambda self, *a, **k: self
from architecture.transMIL import TransMIL

OUT = os.path.dirname(os.path.abspath(__file__))


class Conf:
    D_feat, D_inner, n_class = 384, 128, 2


torch.manual_seed(21)
model = TransMIL(Conf).train()
for layer in (model.layer1, model.layer2):
    layer.attn.to_out[1].p = 0.0
with torch.no_grad():   # LayerNorm affine / biases away from their trivial init so their gradients are exercised non-trivially
    g = torch.Generator().manual_seed(5)
    for n, p in model.named_parameters():
        if "norm" in n:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
n = 300
x = torch.randn(1, n, 384, generator=torch.Generator().manual_seed(n))
label = torch.tensor([1])
logits = model(x)
loss = torch.nn.functional.cross_entropy(logits, label)
loss.backward()
np.savez(os.path.join(OUT, "weights_transmil_train_d384_c2.npz"), **{k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
np.savez(os.path.join(OUT, "transmil_train_n300_d384_c2.npz"), weights=np.array("weights_transmil_train_d384_c2"), x=x.numpy(),
         label=label.numpy(), logits=logits.detach().numpy(), loss=np.array(loss.item()),
         **{"grad." + k: p.grad.numpy().copy() for k, p in model.named_parameters()})
print("loss", loss.item(), "logits", logits.detach().numpy(), "max |grad|", max(p.grad.abs().max().item() for p in model.parameters()))
