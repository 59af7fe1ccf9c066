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

# ---- the one-call training step + optimizer launch (9 launches) as a graph: static bag, static (seed, offset) -- a probe of the
# launch path only (a real loop changes the bag pointer and the draw every step)
from acmil_amd import train as T
for N in (10000, 50000):
    conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=7, n_token=5, n_masked_patch=10,
                    mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=512, D_inner=256)
    torch.manual_seed(0)
    model = T.build_model(conf).cuda().train()
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, torch.device("cuda"), bucket, lr=conf.lr)
    bag = S.synthetic_bag(N, 512, slide_idx=0)[0].half().cuda().unsqueeze(0)
    y = torch.tensor([1], device="cuda")
    u = torch.rand(5, 10, device="cuda")

    def step():
        model.train_step(bag, y, uniforms=u, guard_flag=opt.guard_flag)
        opt.step()

    print("train step N=%d eager : %.4f ms" % (N, timeit(step, 100)))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        print("train step N=%d graph : %.4f ms" % (N, timeit(lambda: g.replay(), 100)))
    except Exception as e:
        print("train step N=%d graph capture failed: %r" % (N, e))
