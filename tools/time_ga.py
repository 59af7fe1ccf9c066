"""Kernel-level timing of the GA forward through the C ABI (torch events on the launch stream)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops, _lib
from acmil_amd import synthetic as SY

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--d", type=int, default=512)
ap.add_argument("--di", type=int, default=256)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--c", type=int, default=2)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--modes", default="fp32,f16x3,f16")
ap.add_argument("--xdtype", default="float32")
args = ap.parse_args()
sd = {k: v.cuda() for k, v in SY.ga_state_dict(args.d, args.di, args.c, args.k).items()}
xs = [torch.randn(args.n, args.d, device="cuda").to(getattr(torch, args.xdtype)) for _ in range(8)]
lib = _lib.load()
for mode in args.modes.split(","):
    packed, dims = ops.ga_pack_weights(sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
        sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"], sd["attention.attention_weights.weight"],
        sd["attention.attention_weights.bias"], [sd["classifier.%d.fc.weight" % i] for i in range(args.k)],
        [sd["classifier.%d.fc.bias" % i] for i in range(args.k)], sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], mode)
    for i in range(5):
        ops.ga_forward(xs[i % 8], packed, dims, mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.iters):
        ops.ga_forward(xs[i % 8], packed, dims, mode)
    e1.record(); torch.cuda.synchronize()
    print("ABLATE=%s %-6s N=%d: %.1f us/slide (3 kernels, back-to-back)" % (os.environ.get("ACMIL_ABLATE", "0"), mode, args.n, e0.elapsed_time(e1) / args.iters * 1e3), flush=True)
