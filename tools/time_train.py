"""Time one full ACMIL training step (forward + losses + backward + AdamW) on resident synthetic bags."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import train as T
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=50000); ap.add_argument("--c", type=int, default=7)
ap.add_argument("--iters", type=int, default=30); ap.add_argument("--precision", default="f16x3"); args = ap.parse_args()
conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=args.c, n_token=5, n_masked_patch=10,
                mask_drop=0.6, arch="ga", precision=args.precision, seed=1, D_feat=512, D_inner=256)
dev = torch.device("cuda", 0)
model = T.build_model(conf).to(dev).train()
opt = T.make_optimizer(model, conf, dev, None, lr=1e-4)
xs = [torch.randn(1, args.n, 512, device=dev).half() for _ in range(8)]
y = torch.tensor([1], device=dev)
import os as _os
FUSED = _os.environ.get("FUSED", "1") == "1"
def step(i):
    if FUSED:
        model.train_step(xs[i % 8], y); opt.step(); return
    sub, slide, attn = model(xs[i % 8])
    l0, l1, d = T.acmil_losses(sub, slide, attn, y, 5)
    opt.zero_grad(set_to_none=False); (l0 + l1 + d).backward(); opt.step()
for i in range(5): step(i)
torch.cuda.synchronize(); t0 = time.time()
for i in range(args.iters): step(i)
torch.cuda.synchronize()
print("train step (fused=%s) N=%d C=%d %s: %.3f ms/step" % (FUSED, args.n, args.c, args.precision, (time.time() - t0) / args.iters * 1e3))
