"""Generate the ACMIL_MHA golden vectors from the REAL reference (run in the dev container only; /root/reference is
imported, never copied).  Eval mode (the reference's train mode draws Dropout(0.1) masks).  Two parameter sets per
shape: the reference's own init (q ~ N(0, 1e-6)) and the same module with q re-drawn at std 0.5 (a trained-like query,
so the k_proj path actually matters in the fixture)."""
import os, sys
from unittest import mock
for name in ("wandb", "timm", "timm.models", "timm.models.layers", "torchmetrics", "h5py", "torchvision", "torchvision.transforms",
             "datasets", "datasets.datasets"):
    sys.modules.setdefault(name, mock.MagicMock())
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
from architecture.transformer import ACMIL_MHA

OUT = os.path.dirname(os.path.abspath(__file__))
for tag, n, d, di, k, c in (("n1000_d384_k5_c2", 1000, 384, 128, 5, 2), ("n257_d512_k1_c7", 257, 512, 256, 1, 7)):
    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k
    for variant in ("init", "q05"):
        torch.manual_seed(7)
        model = ACMIL_MHA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6).eval()
        if variant == "q05":
            with torch.no_grad():
                model.q.copy_(torch.randn(model.q.shape, generator=torch.Generator().manual_seed(3)) * 0.5)
        x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(n))
        with torch.no_grad():
            sub, slide, attns = model(x)
        wname = "weights_mha_%s" % tag                   # one weight file per shape; the q05 case carries its own q
        if variant == "init":
            np.savez(os.path.join(OUT, wname + ".npz"), **{kk: v.detach().numpy().copy() for kk, v in model.state_dict().items()})
        np.savez(os.path.join(OUT, "mha_eval_%s_%s.npz" % (tag, variant)), weights=np.array(wname), x=x.numpy(), sub_preds=sub.numpy(),
                 slide_pred=slide.numpy(), attns=attns.numpy(), n_token=np.array(k), q=model.q.detach().numpy().copy())
        print(tag, variant, sub.shape, slide.shape, attns.shape, float(attns.abs().max()))
