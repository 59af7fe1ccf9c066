"""Race hunt for the single-launch kernels of the training step (arrival tickets, write-through publishes, self-resetting
counters): the same step -- same bag, same uniforms, same parameters -- repeated many times must give BITWISE identical losses,
scores, indices and gradients (every reduction is fixed-order; atomics only count arrivals).  Run through gpurun."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import synthetic as S, train as T

dev = torch.device("cuda")
bad = 0
for N, reps in ((64, 300), (130, 300), (1000, 300), (10000, 300), (50000, 100)):
    conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=7, n_token=5, n_masked_patch=10,
                    mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=512, D_inner=256)
    torch.manual_seed(0)
    model = T.build_model(conf).to(dev).train()
    x = S.synthetic_bag(N, 512, slide_idx=N % 7)[0].half().to(dev).unsqueeze(0)
    y = torch.tensor([N % 7], device=dev)
    u = torch.rand(5, min(10, N), generator=torch.Generator().manual_seed(N)).to(dev)
    ref = None
    for r in range(reps):
        losses, out = model.train_step(x, y, uniforms=u)
        cur = [losses.clone(), out["A_out"].clone(), out["sub_preds"].clone(), out["masked_idx"].clone(), out["topk_idx"].clone()] + \
              [p.grad.clone() for p in model.parameters()]
        if ref is None:
            ref = cur
        else:
            for i, (a, b) in enumerate(zip(ref, cur)):
                if not torch.equal(a, b):
                    bad += 1
                    print("MISMATCH N=%d rep=%d item=%d max|d|=%.3e" % (N, r, i, (a.double() - b.double()).abs().max().item()))
                    break
    torch.cuda.synchronize()
    print("N=%d: %d repeats checked" % (N, reps))
print("STRESS", "FAIL" if bad else "OK", bad)
sys.exit(1 if bad else 0)
