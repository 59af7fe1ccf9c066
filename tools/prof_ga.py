"""Profiling driver: run the fused GA forward repeatedly over rotating resident bags (for rocprofv3)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd.architecture.transformer import ACMIL_GA

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--d", type=int, default=512)
ap.add_argument("--di", type=int, default=256)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--c", type=int, default=2)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--bags", type=int, default=8)
ap.add_argument("--xdtype", default="float32")
args = ap.parse_args()


class Conf:
    D_feat, D_inner, n_class, n_token = args.d, args.di, args.c, args.k


torch.manual_seed(0)
m = ACMIL_GA(Conf, n_token=args.k, precision=args.precision).cuda().eval()
xs = [torch.randn(1, args.n, args.d, device="cuda").to(getattr(torch, args.xdtype)) for _ in range(args.bags)]
with torch.no_grad():
    for i in range(10):
        m(xs[i % args.bags])
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.iters):
        m(xs[i % args.bags])
    torch.cuda.synchronize()
print("%s N=%d: %.1f us/slide wall" % (args.precision, args.n, (time.time() - t0) / args.iters * 1e6))
