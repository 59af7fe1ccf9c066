"""Host enqueue time vs wall time of the module's eval paths (run through gpurun): model(x) per slide and model.forward_batch x64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import synthetic as S
from acmil_amd.architecture.transformer import ACMIL_GA
from acmil_amd.train import EVAL_BATCH as EB

dev = torch.device("cuda")


class Conf:
    D_feat, D_inner, n_class, n_token = 512, 256, 2, 5


torch.manual_seed(0)
model = ACMIL_GA(Conf, n_token=5, n_masked_patch=10, mask_drop=0.6).to(dev).eval()
bags = [S.synthetic_bag(50000, 512, slide_idx=i)[0].to(dev) for i in range(64)]
with torch.no_grad():
    for rnd in range(3):
        for i in range(5):
            model(bags[i].unsqueeze(0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            model(bags[i % 64].unsqueeze(0))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("per slide: host %.1f us / call, wall %.1f us / call" % ((t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6))
        for i in range(2):
            model.forward_batch(bags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = None
        host = 0.0
        for i in range(10):
            h0 = time.perf_counter()
            _, status = model.forward_batch(bags, defer_guard=True)
            host += time.perf_counter() - h0
            if pend is not None:
                int(pend)
            pend = status
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("batch x%d: host %.2f ms / call, wall %.2f ms / call = %.0f slides/s" % (EB, host / 10 * 1e3, (t2 - t0) / 10 * 1e3, 10 * EB / (t2 - t0)))
