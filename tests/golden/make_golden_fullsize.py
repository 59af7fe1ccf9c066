#!/usr/bin/env python
"""The GA path at BASELINE's FULL sizes against the REAL reference (development container only), as compact fixtures: weights
and bags come from seeds (acmil_amd.synthetic, shared with bench.py), the fixture stores what the reference returned --
logits, top-k order, score statistics and strided samples; for the training step the uniforms it drew, the masked indices,
the three losses and every parameter gradient as norm + a 1-in-997 sample.
  * north star : ACMIL_GA eval, N = 50 000, D = 512, D_inner = 256, n_token 5, C 2
  * configs[1] : ACMIL_GA one training iteration of the reference's own train_one_epoch, N = 10 000, n_masked_patch 10, mask_drop 0.6
  * configs[2] : ACMIL_GA eval, N = 50 000, D = 384, D_inner = 128, bag values rounded to bf16"""
import os
import sys
from unittest import mock

for name in ("wandb", "timm", "timm.models", "timm.models.layers", "timm.utils", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets", "yaml"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)
import numpy as np
import torch
import Step3_WSI_classification_ACMIL as step3
from architecture.transformer import ACMIL_GA
from acmil_amd import synthetic as S

torch.set_num_threads(16)
out = {}


class Conf:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def model_for(d, di, k, c, n_masked=0, mask_drop=0.0):
    conf = Conf(D_feat=d, D_inner=di, n_class=c, n_token=k, wd=1e-5, lr=1e-4, min_lr=0.0, warmup_epoch=0, train_epoch=10, wandb_mode="disabled")
    m = ACMIL_GA(conf, n_token=k, n_masked_patch=n_masked, mask_drop=mask_drop)
    m.load_state_dict(S.ga_state_dict(d, di, c, k))
    return m, conf


def stats(t):
    t = t.double()
    return np.array([float(t.mean()), float(t.abs().mean()), float(t.max()), float(t.min())])


# ---- eval, north star and cfg3
for key, n, d, di, k, c, slide, bf16 in (("eval_n50000_d512", 50000, 512, 256, 5, 2, 3, False), ("eval_n50000_d384_bf16", 50000, 384, 128, 5, 2, 4, True)):
    m, _ = model_for(d, di, k, c)
    m.eval()
    x = S.synthetic_bag(n, d, slide_idx=slide)[0]
    if bf16:
        x = x.bfloat16().float()
    with torch.no_grad():
        sub, slide_pred, a = m(x.unsqueeze(0))
    out[key + ".meta"] = np.array([n, d, di, k, c, slide, int(bf16)])
    out[key + ".sub_preds"] = sub.numpy(); out[key + ".slide_pred"] = slide_pred.numpy()
    out[key + ".topk"] = torch.topk(a[0], 10, dim=-1).indices.numpy()
    out[key + ".A_sample"] = a[0][:, ::997].numpy()
    out[key + ".A_stats"] = np.stack([stats(a[0][i]) for i in range(k)])
    print(key, sub.numpy().ravel()[:4], slide_pred.numpy())

# ---- cfg2: one iteration of the reference's own training loop
n, d, di, k, c, slide, label, seed = 10000, 512, 256, 5, 2, 5, 1, 77
m, conf = model_for(d, di, k, c, n_masked=10, mask_drop=0.6)
m.train()
x = S.synthetic_bag(n, d, slide_idx=slide)[0]
torch.manual_seed(seed)
uniforms = torch.rand(k, 10)                      # what forward() will draw first from this RNG state (transformer.py:314)
ce_vals = []
class RecCE(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.ce = torch.nn.CrossEntropyLoss()
    def forward(self, a, b):
        v = self.ce(a, b); ce_vals.append(float(v.detach())); return v
div_vals = []
real_cos = torch.cosine_similarity
opt = torch.optim.AdamW(m.parameters(), lr=conf.lr, weight_decay=conf.wd)
torch.manual_seed(seed)
sub, slide_pred, a = m(x.unsqueeze(0))            # the forward the loop is about to repeat under the same seed
a_np = a.detach().numpy()
masked = np.stack([np.nonzero(a_np[0, i] == np.float32(-1e9))[0] for i in range(k)])
torch.manual_seed(seed)
step3.train_one_epoch(m, RecCE(), [{"input": x.unsqueeze(0), "label": torch.tensor([label])}], opt, torch.device("cpu"), 0, conf)
key = "train_n10000_d512"
out[key + ".meta"] = np.array([n, d, di, k, c, slide, label, seed])
out[key + ".uniforms"] = uniforms.numpy()
out[key + ".masked_idx"] = masked
out[key + ".sub_preds"] = sub.detach().numpy(); out[key + ".slide_pred"] = slide_pred.detach().numpy()
out[key + ".loss0"] = np.array(ce_vals[0]); out[key + ".loss1"] = np.array(ce_vals[-1])
pn = [nm for nm, _ in m.named_parameters()]
out[key + ".param_names"] = np.array(pn)
for nm, p in m.named_parameters():
    g = p.grad.detach().double().reshape(-1)
    out[key + ".gnorm." + nm] = np.array([float(g.norm()), float(g.abs().max())])
    out[key + ".gsample." + nm] = g[::997].float().numpy()
print(key, "loss0 %.6f loss1 %.6f" % (ce_vals[0], ce_vals[-1]), masked[0])
np.savez(os.path.join(OUT, "ga_fullsize_reference.npz"), **out)
print(os.path.getsize(os.path.join(OUT, "ga_fullsize_reference.npz")), "bytes")
