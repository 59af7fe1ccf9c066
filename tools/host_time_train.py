"""Host-side cost of one training step (run through gpurun): wall time of the enqueue alone (no synchronisation) against the
GPU time of the same steps, one-call step vs op-by-op."""
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
model.range_guard = False
for fused in (True, False):
    model.fused_step = fused
    for i in range(20):
        model.train_step(bags[i % 8], labels[i % 8]); opt.step()
    torch.cuda.synchronize()
    host = []
    for i in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_step(bags[i % 8], labels[i % 8])
        t1 = time.perf_counter()
        opt.step()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        host.append((t1 - t0, t2 - t1, t3 - t0))
    host.sort()
    h = host[len(host) // 2]
    print("fused=%s  host enqueue train_step %.1f us, opt.step %.1f us, enqueue->idle %.1f us" % (fused, h[0] * 1e6, h[1] * 1e6, h[2] * 1e6))
