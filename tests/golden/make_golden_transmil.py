#!/usr/bin/env python
"""Generate TransMIL golden fixtures by RUNNING THE REFERENCE (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transmil.py

`architecture/transMIL.py:5` imports the pip package `nystrom_attention` (pinned 0.0.12 in requirements.txt:42,
NOT vendored, not installed, no network).  The repo carries a fork, `architecture/nystrom_attention.py`, with the
same algorithm for return_attn=False; the harness aliases it (sys.modules) and makes `Tensor.cuda` a no-op
(transMIL.py:71 hard-codes .cuda()).  Neither shim touches the reference tree.  Fixtures are eval-mode only
(train mode draws Dropout(0.1) masks); B=1 except one B=2 case (the pinv init couples batch rows through a global max).
Stored: weights (reference ctor under manual_seed(0)), x, logits and the intermediates after layer1, PPEG, layer2,
plus one moore_penrose_iter_pinv in/out pair.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import architecture.nystrom_attention as vendored  # noqa: E402

sys.modules["nystrom_attention"] = vendored
torch.Tensor.cuda = lambda self, *a, **k: self
from architecture.transMIL import TransMIL  # noqa: E402


class Conf:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def capture(model, x):
    feats = {}
    hooks = [model.layer1.register_forward_hook(lambda m, i, o: feats.__setitem__("h1", o.detach().clone())),
             model.pos_layer.register_forward_hook(lambda m, i, o: feats.__setitem__("hp", o.detach().clone())),
             model.layer2.register_forward_hook(lambda m, i, o: feats.__setitem__("h2", o.detach().clone()))]
    with torch.no_grad():
        logits = model(x)
    for h in hooks:
        h.remove()
    return logits, feats


def main():
    torch.set_num_threads(1)
    conf = Conf(D_feat=384, D_inner=128, n_class=2)
    torch.manual_seed(0)
    model = TransMIL(conf).eval()
    np.savez(os.path.join(OUT, "weights_transmil_d384_c2.npz"), **{k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    for n, seed in [(1, 51), (50, 52), (129, 53), (1000, 54)]:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(1, n, 384, generator=g)
        logits, feats = capture(model, x)
        np.savez(os.path.join(OUT, "transmil_eval_n%d_d384_c2.npz" % n), weights=np.array("weights_transmil_d384_c2"),
                 x=x.numpy(), logits=logits.numpy(), h1=feats["h1"].numpy(), hp=feats["hp"].numpy(), h2=feats["h2"].numpy())
        print("n=%d logits=%s" % (n, logits.numpy()))
    # B = 2: the pinv initialisation couples the bags of a batch (global max over batch and heads, nystrom_attention.py:16-18)
    g = torch.Generator().manual_seed(55)
    x = torch.randn(2, 200, 384, generator=g)
    x[1] *= 2.5
    logits, feats = capture(model, x)
    np.savez(os.path.join(OUT, "transmil_eval_b2_n200_d384_c2.npz"), weights=np.array("weights_transmil_d384_c2"), x=x.numpy(),
             logits=logits.numpy(), h2=feats["h2"].numpy())
    print("B=2 logits=%s" % logits.numpy())
    g = torch.Generator().manual_seed(60)
    a = torch.softmax(torch.randn(1, 8, 64, 64, generator=g), dim=-1)
    np.savez(os.path.join(OUT, "pinv_h8_m64.npz"), x=a.numpy(), z=vendored.moore_penrose_iter_pinv(a, 6).numpy())


if __name__ == "__main__":
    main()
