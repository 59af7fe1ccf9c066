"""Time the ACMIL_MHA eval forward (N=50000, D=512, Di=256, K=5) and the CPU oracle (reference association)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mha_oracle as MO
from acmil_amd.architecture.transformer import ACMIL_MHA
n, d, di, k, c = 50000, 512, 256, 5, 2
class Conf: D_feat, D_inner, n_class, n_token = d, di, c, k
sd = MO.default_state_dict(d, di, c, k, seed=2)
m = ACMIL_MHA(Conf, n_token=k); m.load_state_dict(sd); m = m.cuda().eval()
xs = [torch.randn(1, n, d, device="cuda") for _ in range(4)]
with torch.no_grad():
    for i in range(5): m(xs[i % 4])
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(50): out = m(xs[i % 4])
    torch.cuda.synchronize()
dt = (time.time() - t0) / 50
print("ACMIL_MHA N=%d: %.3f ms/slide (%.0f slides/s)" % (n, dt * 1e3, 1 / dt))
x = xs[0].cpu(); torch.set_num_threads(16)
MO.acmil_mha_forward(x, sd, k); t0 = time.time(); ref = MO.acmil_mha_forward(x, sd, k); el = time.time() - t0
with torch.no_grad(): sub, slide, attns = m(xs[0])
print("CPU oracle (16 threads): %.1f ms/slide; max|d attns| %.2e, max|d logits| %.2e" % (
    el * 1e3, (attns.cpu() - ref["attns"]).abs().max().item(), (sub.cpu() - ref["sub_preds"]).abs().max().item()))
