#!/usr/bin/env python
"""Golden fixtures for the GA path at OTHER branch counts than the shipped K = 5, by RUNNING THE REFERENCE
(`--n_token` is a free integer: Step3_WSI_classification_ACMIL.py:39; architecture/transformer.py:292-301 builds K branches),
plus the reference's masked `forward_feature` at the fused width 512 / 256, K = 5 (transformer.py:338-347).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ntoken.py        # dev container only (/root/reference)

Same method and file format as make_golden_wide.py (helpers of make_golden.py reused): the reference's ACMIL_GA built under
manual_seed(0), an eval forward, forward_feature with and without the attention mask, and ONE real `train_one_epoch`
iteration per family.  The two D_inner x D_feat tensors of a train case are stored every `w1_row_stride`-th row.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (stubs the absent third-party modules, imports the reference)

# tag, D_feat, D_inner, n_class, n_token, W1 row stride
FAMILIES = [("d512_k8_c2", 512, 256, 2, 8, 2), ("d384_k10_c7", 384, 128, 7, 10, 1), ("d1024_k16_c2", 1024, 512, 2, 16, 4),
            ("d512_k5_c2m", 512, 256, 2, 5, 2)]


def main():
    torch.set_num_threads(1)
    for tag, d, di, c, k, stride in FAMILIES:
        conf = G.Conf(D_feat=d, D_inner=di, n_class=c, n_token=k, lr=1e-4, min_lr=0, warmup_epoch=0, train_epoch=50, wd=1e-5,
                      wandb_mode="disabled")
        m = G.build(G.ACMIL_GA, conf, n_token=k, n_masked_patch=10, mask_drop=0.6)
        wname = "weights_" + tag.rstrip("m")          # the "m" family IS make_golden.py's d512_k5_c2 module (same seed, same shapes)
        if not tag.endswith("m"):
            G.save(wname, **G.npify(m.state_dict()))
        x = G.bag(300, d, 700 + d + k, fp16=True)
        G.eval_case("ga_eval_n300_" + tag, wname, m, x)
        # forward_feature(use_attention_mask=True) draws rand(K,k) itself (transformer.py:338-347): capture the draw
        m.train()
        torch.manual_seed(78)
        with torch.no_grad():
            feat_m = m.forward_feature(x.float(), use_attention_mask=True)
        torch.manual_seed(78)
        u = torch.rand(k, 10)
        z = dict(np.load(os.path.join(G.OUT, "ga_eval_n300_" + tag + ".npz")))
        G.save("ga_eval_n300_" + tag, **z, bag_feat_masked=feat_m.numpy(), bag_feat_masked_uniforms=u.numpy())
        if tag.endswith("m"):
            continue          # only the masked forward_feature capture is new at the shipped K = 5
        name = "ga_train_n200_" + tag
        G.train_case(name, wname, m, conf, G.bag(200, d, 800 + d + k, fp16=True), 1, 400 + di + k)
        z = dict(np.load(os.path.join(G.OUT, name + ".npz")))
        for key in ("grad.dimreduction.fc1.weight", "after.dimreduction.fc1.weight"):
            z[key] = z[key][::stride].copy()
        z["w1_row_stride"] = np.array(stride)
        G.save(name, **z)


if __name__ == "__main__":
    main()
