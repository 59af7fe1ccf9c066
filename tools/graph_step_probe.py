"""Timing probe: does replaying the 7-launch training step as ONE hipGraph shrink the per-launch floors?  (Same bag, same scalars every
replay -- a measurement of the launch path, not a training loop.)  python tools/graph_step_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from acmil_amd import train as T, synthetic as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=7, n_token=5, n_masked_patch=10, mask_drop=0.6, arch="ga",
                precision="f16x3", seed=1, D_feat=512, D_inner=256)
torch.manual_seed(0)
model = T.build_model(conf).to(dev).train()
bucket = T.GradBucket(list(model.parameters()))
opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
bag = S.synthetic_bag(N, 512, slide_idx=0)[0].half().to(dev).unsqueeze(0)
label = torch.tensor([3], device=dev)


def step():
    model.train_step(bag, label, guard_flag=opt.guard_flag, optimizer=opt, track_flag=False)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 300
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 300
print("N=%d  eager %.4f ms/step   graph replay %.4f ms/step" % (N, eager * 1e3, graph * 1e3))
