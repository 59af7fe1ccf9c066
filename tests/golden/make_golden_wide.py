#!/usr/bin/env python
"""Golden fixtures for the GA path at the reference's wider feature families, by RUNNING THE REFERENCE
(Step3_WSI_classification_ACMIL.py:78-87: path-clip-L-336 768/384, UNI 1024/512, GigaPath 1536/768).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_wide.py        # dev container only (/root/reference)

Same method as make_golden.py (whose helpers are reused): the reference's ACMIL_GA built under manual_seed(0), an eval
forward (+ forward_feature with and without the attention mask) and ONE real `train_one_epoch` iteration per family.
To keep the files small the two D_inner x D_feat tensors of a train case (gradient and post-AdamW value of
dimreduction.fc1.weight) are stored every `w1_row_stride`-th row.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (stubs the absent third-party modules, imports the reference)

FAMILIES = [("d768_k5_c2", 768, 384, 2, 1), ("d1024_k5_c7", 1024, 512, 7, 2), ("d1536_k5_c2", 1536, 768, 2, 8)]


def main():
    torch.set_num_threads(1)
    for tag, d, di, c, stride in FAMILIES:
        conf = G.Conf(D_feat=d, D_inner=di, n_class=c, n_token=5, lr=1e-4, min_lr=0, warmup_epoch=0, train_epoch=50, wd=1e-5,
                      wandb_mode="disabled")
        m = G.build(G.ACMIL_GA, conf, n_token=5, n_masked_patch=10, mask_drop=0.6)
        wname = "weights_" + tag
        G.save(wname, **G.npify(m.state_dict()))
        x = G.bag(300, d, 500 + d, fp16=True)
        G.eval_case("ga_eval_n300_" + tag, wname, m, x)
        # forward_feature(use_attention_mask=True) draws rand(K,k) itself (transformer.py:338-347): capture the draw
        m.train()
        torch.manual_seed(77)
        with torch.no_grad():
            feat_m = m.forward_feature(x.float(), use_attention_mask=True)
        torch.manual_seed(77)
        u = torch.rand(5, 10)
        z = dict(np.load(os.path.join(G.OUT, "ga_eval_n300_" + tag + ".npz")))
        G.save("ga_eval_n300_" + tag, **z, bag_feat_masked=feat_m.numpy(), bag_feat_masked_uniforms=u.numpy())
        # one real training iteration; thin the two big tensors afterwards
        name = "ga_train_n200_" + tag
        G.train_case(name, wname, m, conf, G.bag(200, d, 600 + d, fp16=True), 1, 300 + di)
        z = dict(np.load(os.path.join(G.OUT, name + ".npz")))
        for key in ("grad.dimreduction.fc1.weight", "after.dimreduction.fc1.weight"):
            z[key] = z[key][::stride].copy()
        z["w1_row_stride"] = np.array(stride)
        G.save(name, **z)


if __name__ == "__main__":
    main()
