"""Same-process A/B of the training step's host-side options (run through gpurun): the one-call step (csrc/ga_step.hip)
with and without the range-guard read-back, and the op-by-op step.  Prints ms per step (median of rounds) for each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import synthetic as S, train as T

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=7, n_token=5, n_masked_patch=10,
                mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=512, D_inner=256)
torch.manual_seed(0)
model = T.build_model(conf).to(dev).train()
bucket = T.GradBucket(list(model.parameters()))
opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
bags = [S.synthetic_bag(N, 512, slide_idx=i)[0].half().to(dev).unsqueeze(0) for i in range(8)]
labels = [torch.tensor([i % 7], device=dev) for i in range(8)]


LAGGED = False


def step(i):
    if LAGGED:
        model.train_step(bags[i % 8], labels[i % 8], guard_flag=opt.guard_flag)
        bucket.sync_from_grads()
        opt.step(track_flag=True)
        opt.poll_skipped(2)
        return
    model.train_step(bags[i % 8], labels[i % 8])
    bucket.sync_from_grads()
    opt.step()


def timeit(steps=200):
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {}
for rnd in range(3):
    for name, guard, fused, lag in (("one-call/lagged", True, True, True), ("one-call/readback", True, True, False),
                                    ("one-call/noguard", False, True, False), ("op-by-op", True, False, False)):
        model.range_guard, model.fused_step, LAGGED = guard, fused, lag
        res.setdefault(name, []).append(timeit())
        opt.poll_skipped(0)
for k, v in res.items():
    print("%-18s ms/step %s  median %.4f" % (k, ["%.4f" % t for t in v], sorted(v)[1]))
