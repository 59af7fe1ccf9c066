"""Time one TransMIL training step (op-by-op autograd path: forward + CE + backward + fused AdamW)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from acmil_amd.architecture.transMIL import TransMIL
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=10000); ap.add_argument("--d", type=int, default=768)
ap.add_argument("--di", type=int, default=384); ap.add_argument("--iters", type=int, default=10); args = ap.parse_args()
class Conf: D_feat, D_inner, n_class = args.d, args.di, 2
torch.manual_seed(0)
m = TransMIL(Conf).cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
xs = [torch.randn(1, args.n, args.d, device="cuda") for _ in range(2)]
y = torch.tensor([1], device="cuda")
def step(i):
    opt.zero_grad(set_to_none=True)
    F.cross_entropy(m(xs[i % 2]), y).backward()
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.time()
for i in range(args.iters): step(i)
torch.cuda.synchronize()
print("TransMIL train step N=%d D=%d Di=%d: %.2f ms/step (peak mem %.2f GB)" % (args.n, args.d, args.di, (time.time() - t0) / args.iters * 1e3, torch.cuda.max_memory_allocated() / 2**30))
