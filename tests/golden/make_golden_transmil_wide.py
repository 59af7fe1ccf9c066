#!/usr/bin/env python
"""TransMIL at the BASELINE configs[3] width (D = 768, D_inner = 384) and at D_inner = 256 from the REAL reference (development
container only; same shims as make_golden_transmil.py: the vendored nystrom_attention fork aliased, Tensor.cuda a no-op).
Weights and bags are NOT stored: both come from seeds (oracle.transmil_oracle.default_state_dict, torch.randn), so a fixture
is a few hundred bytes of reference outputs -- logits and the per-stage means / absolute means / cls rows."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
import architecture.nystrom_attention as vendored  # noqa: E402

sys.modules["nystrom_attention"] = vendored
torch.Tensor.cuda = lambda self, *a, **k: self
from architecture.transMIL import TransMIL  # noqa: E402
from oracle import transmil_oracle as TO  # noqa: E402


class Conf:
    def __init__(self, **kw):
        self.__dict__.update(kw)


cases = {}
torch.set_num_threads(8)
for n, d, di, c, wseed, xseed in [(700, 768, 384, 2, 21, 701), (3000, 768, 384, 2, 21, 702), (900, 512, 256, 3, 22, 703), (100000, 768, 384, 2, 21, 704)]:
    sd = TO.default_state_dict(d, di, c, seed=wseed)
    model = TransMIL(Conf(D_feat=d, D_inner=di, n_class=c)).eval()
    model.load_state_dict(sd)
    x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(xseed))
    feats = {}
    hooks = [model.layer1.register_forward_hook(lambda m, i, o: feats.__setitem__("h1", o.detach().clone())),
             model.pos_layer.register_forward_hook(lambda m, i, o: feats.__setitem__("hp", o.detach().clone())),
             model.layer2.register_forward_hook(lambda m, i, o: feats.__setitem__("h2", o.detach().clone()))]
    with torch.no_grad():
        logits = model(x)
    for h in hooks:
        h.remove()
    key = "n%d_d%d_di%d_c%d" % (n, d, di, c)
    cases[key + ".meta"] = np.array([n, d, di, c, wseed, xseed])
    cases[key + ".logits"] = logits.numpy()
    for k, v in feats.items():
        cases[key + "." + k + "_cls"] = v[0, 0].numpy()                       # the class-token row after the stage
        cases[key + "." + k + "_stat"] = np.array([float(v.mean()), float(v.abs().mean()), float(v.abs().max())])
    print(key, logits.numpy())
np.savez(os.path.join(OUT, "transmil_eval_wide.npz"), **cases)
