"""Error table of the TransMIL HIP path vs the CPU oracle (h1 / hp / h2 / logits), fused and unfused attention legs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import transmil_oracle as TO
from acmil_amd.architecture.transMIL import TransMIL

for n, d, di, c in [(777, 768, 384, 7), (3000, 512, 256, 2), (5000, 1024, 512, 2), (129, 384, 128, 2)]:
    sd = TO.default_state_dict(d, di, c, seed=3)
    x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(n))
    ref = TO.transmil_forward(x, sd)
    class Conf: D_feat, D_inner, n_class = d, di, c
    m = TransMIL(Conf); m.load_state_dict(sd); m = m.cuda().eval()
    with torch.no_grad():
        lg = m(x.cuda(), debug=True)
    errs = {k: (m._last[k].cpu() - ref[k][0]).abs().max().item() for k in ("h1", "hp", "h2")}
    worst = (m._last["h2"].cpu() - ref["h2"][0]).abs()
    idx = worst.argmax().item()
    print(n, d, di, "err", {k: "%.2e" % v for k, v in errs.items()}, "logits %.2e" % (lg.cpu() - ref["logits"]).abs().max().item(),
          "worst h2 at row %d col %d (ref %.4f)" % (idx // di, idx % di, ref["h2"][0].flatten()[idx].item()))
