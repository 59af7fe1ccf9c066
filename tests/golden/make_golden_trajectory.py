#!/usr/bin/env python
"""Golden fixture of a MULTI-STEP training trajectory, produced by RUNNING THE REFERENCE's own loop.

Run in the development container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trajectory.py

The reference's `train_one_epoch` (Step3_WSI_classification_ACMIL.py:175-227, imported unmodified; its absent third-party imports are
stubbed exactly as in make_golden.py) runs TWO epochs over a five-slide loader: ten optimizer steps with the per-iteration schedule of
utils/utils.py:250-262 (epoch 0 = linear warm-up from lr 0, epoch 1 = the first cosine steps), torch.optim.AdamW moments and bias
corrections evolving, STKIM masks drawn per step.  Stored (no reference text, only inputs and what the reference computed):
  seeds / shapes of the five fp16 bags + a checksum of each (the bags are regenerated from the seeds by the test), labels,
  the loader order of each epoch, the [K, k] uniforms every forward drew, per-step (loss0, loss1), per-step lr,
  the FINAL parameters, and per parameter element the smallest |gradient| any step saw (the test compares where the Adam update
  is a stable function of the gradient: sign(g) steps on elements whose gradient is rounding noise differ between any two fp32
  implementations, e.g. attention_weights.bias, whose gradient is analytically zero), and per hidden unit of dimreduction.fc1 the
  smallest |pre-activation| any patch of any step had (`minpre`: d relu / d pre jumps at 0, so a unit that came within the arithmetic's
  own error of zero has a gradient row that is not a function of the data at fp32 precision).
Weights: weights_d512_k5_c7.npz (make_golden.py: the reference module under manual_seed(0)).
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
for name in ["wandb", "timm", "timm.utils", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "torchvision.datasets", "datasets", "datasets.datasets"]:
    sys.modules[name] = mock.MagicMock(name=name)

from architecture.transformer import ACMIL_GA  # noqa: E402
import Step3_WSI_classification_ACMIL as step3  # noqa: E402


class Conf:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class RecordingCE(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ce = torch.nn.CrossEntropyLoss()
        self.values = []

    def forward(self, a, b):
        v = self.ce(a, b)
        self.values.append(float(v.detach()))
        return v


BAGS = [(300, 501), (640, 502), (900, 503), (450, 504), (1200, 505)]       # (patches, generator seed)
LABELS = [3, 0, 6, 2, 5]
ORDERS = [[0, 1, 2, 3, 4], [3, 0, 4, 2, 1]]
D, DI, K, C = 512, 256, 5, 7


def bag(n, seed):
    return torch.randn(1, n, D, generator=torch.Generator().manual_seed(seed)).half()


def main():
    torch.set_num_threads(1)
    conf = Conf(D_feat=D, D_inner=DI, n_class=C, n_token=K, lr=2e-4, min_lr=0, warmup_epoch=1, train_epoch=3, wd=1e-2,
                wandb_mode="disabled")
    torch.manual_seed(0)
    model = ACMIL_GA(conf, n_token=K, n_masked_patch=10, mask_drop=0.6)
    w = np.load(os.path.join(OUT, "weights_d512_k5_c7.npz"))
    for n, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), w[n]), n             # the same seeded module make_golden.py stored
    bags = [bag(n, s) for n, s in BAGS]
    opt = torch.optim.AdamW(model.parameters(), lr=0.001, weight_decay=conf.wd)
    crit = RecordingCE()
    uniforms, lrs = [], []
    min_abs = {n: torch.full_like(p, float("inf")) for n, p in model.named_parameters()}
    real_rand = torch.rand
    real_step = opt.step

    def rand_spy(*a, **kw):
        u = real_rand(*a, **kw)
        uniforms.append(u.clone())
        return u

    def step_spy(*a, **kw):
        lrs.append(opt.param_groups[0]["lr"])
        for n, p in model.named_parameters():
            min_abs[n] = torch.minimum(min_abs[n], p.grad.detach().abs())
        return real_step(*a, **kw)

    minpre = torch.full((DI,), float("inf"))

    def pre_spy(mod, inp, out):        # fc1 is Linear(D, DI, bias=False): its output is the pre-activation (ReLU follows, network.py:49-57)
        nonlocal minpre
        minpre = torch.minimum(minpre, out.detach().abs().reshape(-1, DI).min(0)[0])

    hook = model.dimreduction.fc1.register_forward_hook(pre_spy)
    opt.step = step_spy
    torch.manual_seed(77)
    with mock.patch.object(torch, "rand", rand_spy):
        for epoch, order in enumerate(ORDERS):
            loader = [{"input": bags[i], "label": torch.tensor([LABELS[i]])} for i in order]
            step3.train_one_epoch(model, crit, loader, opt, torch.device("cpu"), epoch, conf)
    hook.remove()
    steps = sum(len(o) for o in ORDERS)
    assert len(uniforms) == steps and len(lrs) == steps and len(crit.values) == 2 * steps
    final = {"final." + n: v.detach().numpy().copy() for n, v in model.state_dict().items()}
    mins = {"mingrad." + n: v.numpy().astype(np.float32) for n, v in min_abs.items()}
    checks = np.array([[float(b.float().sum()), float(b.float().abs().sum()), float(b[0, -1, -1])] for b in bags], np.float64)
    path = os.path.join(OUT, "ga_trajectory_d512_k5_c7.npz")
    np.savez_compressed(path, weights=np.array("weights_d512_k5_c7"), bag_shapes=np.array(BAGS), bag_checks=checks, labels=np.array(LABELS),
                        orders=np.array(ORDERS), uniforms=torch.stack(uniforms).numpy(), lrs=np.array(lrs, np.float64),
                        losses=np.array(crit.values, np.float32).reshape(steps, 2), minpre=minpre.numpy(), lr=np.array(conf.lr), wd=np.array(conf.wd),
                        warmup_epoch=np.array(conf.warmup_epoch), train_epoch=np.array(conf.train_epoch), **final, **mins)
    print("wrote %s %.2f MB; lrs %s; units with |pre| < 2e-6: %s" % (path, os.path.getsize(path) / 1e6, ["%.2e" % v for v in lrs],
                                                                       np.nonzero(minpre.numpy() < 2e-6)[0].tolist()))


if __name__ == "__main__":
    main()
