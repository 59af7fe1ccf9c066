"""W_hi x_lo skip of the Linear kernels on fp32 rows of f16-exact values (run through gpurun): projection time for randn fp32, fp32 of
fp16 values, fp16 storage at the GigaPath / UNI / TransMIL-fc1 shapes; and the grouped GigaPath forward on fp16-valued bags."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops

dev = torch.device("cuda", 0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (m, k, n_out, name) in [(50000, 1536, 768, "GigaPath projection, one slide (lin64)"), (400000, 1536, 768, "GigaPath projection, 8 slides (lin_kernel)"),
                            (50000, 1024, 512, "UNI projection (lin64)"), (100000, 768, 384, "TransMIL _fc1 (lin_kernel)")]:
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(n_out, k, generator=g) * 0.03).to(dev)
    packed = ops.linear_pack(w)
    x = torch.randn(m, k, device=dev)
    x16 = x.half()
    xe = x16.float()
    out = torch.empty(m, n_out, device=dev)
    t = [timed(lambda v=v: ops.linear_f16x3(v, packed, n_out, relu=True, out=out)) for v in (x, xe, x16)]
    print("%-50s randn fp32 %.1f us | fp32 of fp16 values %.1f us | fp16 storage %.1f us" % (name, *t))
    del x, x16, xe, out
