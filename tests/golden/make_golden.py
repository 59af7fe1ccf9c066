#!/usr/bin/env python
"""Generate the golden fixtures for the gated-attention (GA) path by RUNNING THE REFERENCE.

Run in the development container only (needs /root/reference, which does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports `architecture.transformer.{ACMIL_GA, ABMIL}` unmodified from /root/reference, and the
reference's own `train_one_epoch` from `Step3_WSI_classification_ACMIL.py` (its unrelated
third-party imports -- wandb, timm, torchmetrics, h5py, torchvision -- are absent from this image and
are stubbed with empty modules; none of them is touched by the code that runs here).  Nothing from
the reference is copied: only inputs and the outputs the reference computes are stored, as `.npz`.

Fixtures (all fp32 unless noted):
  weights_<cfg>.npz     every state_dict tensor of the reference module built under manual_seed(0)
  ga_eval_*.npz         x, sub_preds, slide_pred, A_out, bag_feat (forward_feature)
  abmil_eval_*.npz      x, logits
  ga_train_*.npz        x, label, rand seed, the [K,k] uniforms the forward drew, topk indices,
                        masked indices, sub_preds, slide_pred, A_out (with -1e9), loss0, loss1,
                        per-parameter grads and post-AdamW-step parameters from ONE real
                        reference `train_one_epoch` iteration.
Bags that stand for on-disk features are stored as float16 (what Step2 writes,
Step2_feature_extract.py:165) and up-cast exactly as Step3 does (:193).
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

# ---- stub the absent third-party modules the reference scripts import but this path never calls
# (`datasets` is stubbed too: the reference's data package has no __init__.py and would lose to the
# unrelated HuggingFace `datasets` wheel in this image; the loader is not on the path under test.)
for name in ["wandb", "timm", "timm.utils", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "torchvision.datasets", "datasets", "datasets.datasets"]:
    sys.modules[name] = mock.MagicMock(name=name)

from architecture.transformer import ACMIL_GA, ABMIL  # noqa: E402
import Step3_WSI_classification_ACMIL as step3  # noqa: E402


class Conf:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def npify(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **arrays)
    print("wrote %-40s %.2f MB" % (name + ".npz", os.path.getsize(path) / 1e6))


def bag(n, d, seed, fp16=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, n, d, generator=g)
    return x.half() if fp16 else x


def build(cls, conf, **kw):
    torch.manual_seed(0)
    return cls(conf, **kw)


def eval_case(name, wname, model, x_store):
    model.eval()
    x = x_store.float()
    with torch.no_grad():
        sub, slide, a = model(x)
        feat = model.forward_feature(x)
    save(name, weights=np.array(wname), x=x_store.numpy(), sub_preds=sub.numpy(), slide_pred=slide.numpy(),
         A_out=a.numpy(), bag_feat=feat.numpy())


class RecordingCE(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ce = torch.nn.CrossEntropyLoss()
        self.values = []

    def forward(self, a, b):
        v = self.ce(a, b)
        self.values.append(float(v.detach()))
        return v


def train_case(name, wname, model, conf, x_store, label, seed):
    """One real iteration of the reference train_one_epoch on a one-slide 'loader'."""
    model.train()
    x = x_store.float()
    k = min(model.n_masked_patch, x.shape[1])
    # (1) pure forward under the seed: capture what forward returns and what it drew
    torch.manual_seed(seed)
    sub, slide, a = model(x)
    torch.manual_seed(seed)
    uniforms = torch.rand(conf.n_token, k)
    a_np = a.detach().numpy()
    masked_idx = np.stack([np.nonzero(a_np[0, i] == np.float32(-1e9))[0] for i in range(conf.n_token)])
    # un-masked scores -> reference top-k (same call the forward makes)
    model.eval()
    with torch.no_grad():
        _, _, a_raw = model(x)
    model.train()
    _, topk_idx = torch.topk(a_raw[0], k, dim=-1)
    # (2) the reference's own training iteration (loss assembly, backward, AdamW) under the same seed
    opt = torch.optim.AdamW(model.parameters(), lr=0.001, weight_decay=conf.wd)
    crit = RecordingCE()
    loader = [{"input": x_store, "label": torch.tensor([label])}]
    before = npify(model.state_dict())
    torch.manual_seed(seed)
    step3.train_one_epoch(model, crit, loader, opt, torch.device("cpu"), 0, conf)
    grads = {"grad." + n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
    after = {"after." + n: v for n, v in npify(model.state_dict()).items()}
    loss0 = crit.values[0] if conf.n_token > 1 else 0.0
    loss1 = crit.values[-1]
    save(name, weights=np.array(wname), x=x_store.numpy(), label=np.array([label]), seed=np.array(seed),
         uniforms=uniforms.numpy(), topk_idx=topk_idx.numpy(), masked_idx=masked_idx,
         sub_preds=sub.detach().numpy(), slide_pred=slide.detach().numpy(), A_out=a_np,
         A_raw=a_raw.numpy(), loss0=np.array(loss0, np.float32), loss1=np.array(loss1, np.float32),
         lr=np.array(opt.param_groups[0]["lr"]), wd=np.array(conf.wd), **grads, **after)
    # restore weights so later cases sharing this module see the seeded init
    model.load_state_dict({k2: torch.from_numpy(v) for k2, v in before.items()})


def main():
    torch.set_num_threads(1)  # deterministic reductions
    # ---- cfg A: D 512/256, K=5, C=2 (north-star / cfg2 shape)
    cA = Conf(D_feat=512, D_inner=256, n_class=2, n_token=5)
    mA = build(ACMIL_GA, cA, n_token=5, n_masked_patch=10, mask_drop=0.6)
    save("weights_d512_k5_c2", **npify(mA.state_dict()))
    eval_case("ga_eval_n257_d512_k5_c2", "weights_d512_k5_c2", mA, bag(257, 512, 11))
    eval_case("ga_eval_n1_d512_k5_c2", "weights_d512_k5_c2", mA, bag(1, 512, 12))
    eval_case("ga_eval_n33_d512_k5_c2", "weights_d512_k5_c2", mA, bag(33, 512, 13))
    tconf = Conf(D_feat=512, D_inner=256, n_class=2, n_token=5, lr=1e-4, min_lr=0, warmup_epoch=0,
                 train_epoch=50, wd=1e-5, wandb_mode="disabled")
    train_case("ga_train_n7_d512_k5_c2", "weights_d512_k5_c2", mA, tconf, bag(7, 512, 14, fp16=True), 1, 123)
    train_case("ga_train_n640_d512_k5_c2", "weights_d512_k5_c2", mA, tconf, bag(640, 512, 15, fp16=True), 0, 124)

    # ---- cfg B: cfg1 shape, K=1 (ABMIL-equivalent ACMIL_GA) + the ABMIL class itself
    cB = Conf(D_feat=512, D_inner=256, n_class=2, n_token=1)
    mB = build(ACMIL_GA, cB, n_token=1, n_masked_patch=0, mask_drop=0.0)
    save("weights_d512_k1_c2", **npify(mB.state_dict()))
    xB = bag(1000, 512, 21)
    eval_case("ga_eval_n1000_d512_k1_c2", "weights_d512_k1_c2", mB, xB)
    mAB = build(ABMIL, cB)
    save("weights_abmil_d512_c2", **npify(mAB.state_dict()))
    mAB.eval()
    with torch.no_grad():
        logits = mAB(xB)
    save("abmil_eval_n1000_d512_c2", weights=np.array("weights_abmil_d512_c2"), x_from=np.array("ga_eval_n1000_d512_k1_c2"),
         logits=logits.numpy())

    # ---- cfg C: Camelyon16 / SSL ViT-S shape D 384/128, K=5, C=7, fp16-stored bag
    cC = Conf(D_feat=384, D_inner=128, n_class=7, n_token=5)
    mC = build(ACMIL_GA, cC, n_token=5, n_masked_patch=10, mask_drop=0.6)
    save("weights_d384_k5_c7", **npify(mC.state_dict()))
    eval_case("ga_eval_n1000_d384_k5_c7", "weights_d384_k5_c7", mC, bag(1000, 384, 31, fp16=True))

    # ---- cfg D: BRACS-shape training (cfg5): D 512/256, K=5, C=7, k=10, drop 0.6
    cD = Conf(D_feat=512, D_inner=256, n_class=7, n_token=5, lr=1e-4, min_lr=0, warmup_epoch=0,
              train_epoch=50, wd=1e-5, wandb_mode="disabled")
    mD = build(ACMIL_GA, cD, n_token=5, n_masked_patch=10, mask_drop=0.6)
    save("weights_d512_k5_c7", **npify(mD.state_dict()))
    train_case("ga_train_n2048_d512_k5_c7", "weights_d512_k5_c7", mD, cD, bag(2048, 512, 41, fp16=True), 3, 125)


if __name__ == "__main__":
    main()
