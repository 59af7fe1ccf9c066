"""Grouped eval of the composed GigaPath family (ACMIL_GA.forward_group): per-slide time of a 16 x 50 000-row group against `model(x)`
per slide, and bitwise equality of the group's outputs with the per-slide ones at full size -- run through gpurun.
ACMIL_HIP_LIB=.../libacmil_hip_ab.so ACMIL_LIN64=0|1 forces the projection kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import synthetic as S
from acmil_amd.architecture.transformer import ACMIL_GA

N, D, Di, K, C, G = 50000, 1536, 768, 5, 2, int(os.environ.get("G", "16"))
dev = torch.device("cuda", 0)


class Conf:
    D_feat, D_inner, n_class, n_token = D, Di, C, K


model = ACMIL_GA(Conf, n_token=K, n_masked_patch=10, mask_drop=0.6)
model.load_state_dict(S.ga_state_dict(D, Di, C, K, seed=0))
model = model.to(dev).eval()
bags = [S.synthetic_bag(N, D, slide_idx=i)[0].to(dev) for i in range(8)]
groups = [(torch.cat([bags[(gi * 3 + j) % 8] for j in range(G)], 0), [N] * G) for gi in range(2)]
with torch.no_grad():
    t = model.forward_group(*groups[0])
    for j in (0, 7 % G, G - 1):
        s1, l1, a1 = model(bags[j % 8].unsqueeze(0))
        print("bag %d bitwise equal:" % j, torch.equal(t[j][2], a1) and torch.equal(t[j][0], s1) and torch.equal(t[j][1], l1),
              "max|dA| %.2e" % (t[j][2] - a1).abs().max().item())

    def timed(fn, n):
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(n):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    tg = timed(lambda i: model.forward_group(*groups[i % 2]), 20)
    ts = timed(lambda i: model(bags[i % 8].unsqueeze(0)), 60)
    print("group of %d: %.1f us per slide (%.0f slides/s);  per slide: %.1f us (%.0f slides/s)" % (G, tg / G, 1e6 * G / tg, ts, 1e6 / ts))
