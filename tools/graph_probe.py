"""Probe: does HIP graph replay (torch.cuda.CUDAGraph around the ctypes launches) shorten launch-bound pipelines?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import synthetic as S
from acmil_amd.architecture.transMIL import TransMIL

class Conf: D_feat, D_inner, n_class = 768, 384, 2
m = TransMIL(Conf); m.load_state_dict(S.transmil_state_dict(768, 384, 2, seed=1)); m = m.cuda().eval()
x = torch.randn(1, 100000, 768, device="cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("eager  : %.3f ms" % timeit(lambda: m(x)))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): m(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(x)
    print("graph  : %.3f ms" % timeit(lambda: g.replay()))
    ref = m(x)
    g.replay(); torch.cuda.synchronize()
    print("max diff graph vs eager:", (out - ref).abs().max().item())
