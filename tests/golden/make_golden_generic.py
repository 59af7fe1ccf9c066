#!/usr/bin/env python
"""Golden fixtures for the constructor arguments of the reference's gated-attention modules that the fused kernels do not cover and
that acmil_amd serves on its op-by-op ("generic") path, by RUNNING THE REFERENCE:

  * attention hidden width D != 128 -- `ACMIL_GA(conf, D=...)` / `ABMIL(conf, D=...)` (architecture/transformer.py:240,270,292)
  * `DimReduction(numLayer_Res > 0)` -- the residual blocks of architecture/network.py:22-34,44-56 (stand-alone module)
  * `MHA(conf)` -- the one-branch multi-head module of architecture/transformer.py:86-104 (eval forward at its own init and with a
    trained-size query, and the gradients of the cross-entropy in eval mode = Dropout(0.1) off)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_generic.py        # dev container only (/root/reference)

Same method and file format as make_golden.py (its helpers are reused): modules built under manual_seed(0), an eval forward,
forward_feature, ONE real `train_one_epoch` iteration per ACMIL_GA family (losses, gradients, post-AdamW parameters; the two large
gradient tensors stored every `w1_row_stride`-th row).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (stubs the absent third-party modules, imports the reference)
from architecture.network import DimReduction  # noqa: E402
from architecture.transformer import MHA, MutiHeadAttention, MutiHeadAttention_modify  # noqa: E402

# tag, D_feat, D_inner, n_class, n_token, attention width, W1 row stride
FAMILIES = [("d512_a64_k5_c2", 512, 256, 2, 5, 64, 4), ("d384_a256_k3_c7", 384, 128, 7, 3, 256, 2)]


def main():
    torch.set_num_threads(1)
    for tag, d, di, c, k, da, stride in FAMILIES:
        conf = G.Conf(D_feat=d, D_inner=di, n_class=c, n_token=k, lr=1e-4, min_lr=0, warmup_epoch=0, train_epoch=50, wd=1e-5,
                      wandb_mode="disabled")
        m = G.build(G.ACMIL_GA, conf, D=da, n_token=k, n_masked_patch=10, mask_drop=0.6)
        wname = "weights_" + tag
        G.save(wname, **G.npify(m.state_dict()))
        G.eval_case("ga_eval_n300_" + tag, wname, m, G.bag(300, d, 900 + da, fp16=True))
        name = "ga_train_n200_" + tag
        G.train_case(name, wname, m, conf, G.bag(200, d, 950 + da, fp16=True), 1, 500 + da)
        z = dict(np.load(os.path.join(G.OUT, name + ".npz")))
        for key in ("grad.dimreduction.fc1.weight", "after.dimreduction.fc1.weight"):
            z[key] = z[key][::stride].copy()
        z["w1_row_stride"] = np.array(stride)
        G.save(name, **z)
    # ABMIL with a 64-wide attention
    cB = G.Conf(D_feat=512, D_inner=256, n_class=2, n_token=1)
    mB = G.build(G.ABMIL, cB, D=64)
    G.save("weights_abmil_a64_d512_c2", **G.npify(mB.state_dict()))
    xB = G.bag(400, 512, 977, fp16=True)
    mB.eval()
    with torch.no_grad():
        logits = mB(xB.float())
    G.save("abmil_eval_n400_a64_d512_c2", weights=np.array("weights_abmil_a64_d512_c2"), x=xB.numpy(), logits=logits.numpy())
    # DimReduction with two residual blocks: forward + the gradients of sum(out^2) / N
    torch.manual_seed(0)
    dr = DimReduction(384, 128, numLayer_Res=2)
    x = torch.randn(500, 384, generator=torch.Generator().manual_seed(31)).half()
    out = dr(x.float())
    (out.square().sum() / out.shape[0]).backward()
    G.save("dimreduction_res2_n500_d384", x=x.numpy(), out=out.detach().numpy(),
           **{"w." + n: p.detach().numpy().copy() for n, p in dr.named_parameters()},
           **{"grad." + n: p.grad.numpy().copy() for n, p in dr.named_parameters()})
    # MHA: logits at the constructor's q (std 1e-6) and at a trained-size q; gradients of CE(logits, label) with dropout off
    cM = G.Conf(D_feat=384, D_inner=128, n_class=3, n_token=1)
    torch.manual_seed(0)
    mh = MHA(cM)
    mh.eval()
    xM = G.bag(350, 384, 988, fp16=True)
    with torch.no_grad():
        logits0 = mh(xM.float())
        mh.q.copy_(torch.randn(1, 1, 128, generator=torch.Generator().manual_seed(5)) * 0.5)
    logits1 = mh(xM.float())
    torch.nn.functional.cross_entropy(logits1, torch.tensor([2])).backward()
    G.save("mha_single_n350_d384_c3", x=xM.numpy(), logits_init=logits0.numpy(), logits=logits1.detach().numpy(), label=np.array([2]),
           **{"w." + n: p.detach().numpy().copy() for n, p in mh.named_parameters()},
           **{"grad." + n: p.grad.numpy().copy() for n, p in mh.named_parameters()})
    # the attention layers stand-alone with 4 heads and down-sampling 2 (transformer.py:107-236): eval mode, gradients of sum(out^2)
    torch.manual_seed(0)
    att = MutiHeadAttention(128, 4, downsample_rate=2)
    att.eval()
    g = torch.Generator().manual_seed(17)
    q = (torch.randn(1, 3, 128, generator=g) * 0.7).requires_grad_(True)
    kv = torch.randn(1, 500, 128, generator=g)
    out, attn = att(q, kv, kv)
    out.square().sum().backward()
    mod = MutiHeadAttention_modify(128, 4, downsample_rate=2)
    mod.eval()
    pa = torch.softmax(attn.detach()[:, :1], dim=-1)                     # [H, 1, N] normalised weights, as ACMIL_MHA hands them over
    with torch.no_grad():
        out_m = mod(kv, pa.unsqueeze(0))
    G.save("mha_layer_h4_ds2_n500_e128", q=q.detach().numpy(), kv=kv.numpy(), out=out.detach().numpy(), attn=attn.detach().numpy(),
           grad_q=q.grad.numpy(), pa=pa.numpy(), out_modify=out_m.numpy(),
           **{"w." + n: p.detach().numpy().copy() for n, p in att.named_parameters()},
           **{"grad." + n: p.grad.numpy().copy() for n, p in att.named_parameters()},
           **{"wm." + n: p.detach().numpy().copy() for n, p in mod.named_parameters()})


if __name__ == "__main__":
    main()
