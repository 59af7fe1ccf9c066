"""Where does the host time of a training step go?  Tiny bag (N=256) so every phase is launch/host-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import train as T, ops
conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=7, n_token=5, n_masked_patch=10,
                mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=512, D_inner=256)
dev = torch.device("cuda", 0)
model = T.build_model(conf).to(dev).train()
bucket = T.GradBucket(list(model.parameters()))
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
opt_f = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
x = torch.randn(1, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 512, device=dev).half()
y = torch.tensor([1], device=dev)

def timeit(name, fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-40s host %.1f us/iter, with drain %.1f us/iter" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))

packed, dims = model._packed()
xb = model._bag(x)
timeit("train_step only", lambda: model.train_step(x, y))
timeit("train_step + AdamW(foreach)", lambda: (model.train_step(x, y), bucket.sync_from_grads(), opt.step()))
timeit("train_step + AdamW(fused=True)", lambda: (model.train_step(x, y), bucket.sync_from_grads(), opt_f.step()))
timeit("AdamW(foreach) alone", lambda: opt.step())
timeit("AdamW(fused) alone", lambda: opt_f.step())
timeit("_packed() (repack after param update)", lambda: (model.dimreduction.fc1.weight.data.add_(0.0), model._packed()))
timeit("ga_scores", lambda: ops.ga_scores(xb, packed, dims, "f16x3"))
A, h = ops.ga_scores(xb, packed, dims, "f16x3")
u = torch.rand(5, 10, device=dev)
timeit("torch.rand", lambda: torch.rand(5, 10, device=dev))
timeit("stkim_select", lambda: ops.stkim_select(A, 10, 6, u))
topk, midx = ops.stkim_select(A, 10, 6, u)
timeit("ga_pool", lambda: ops.ga_pool(h, A, packed, dims, "f16x3", midx, want_afeat=True))
out = ops.ga_pool(h, A, packed, dims, "f16x3", midx, want_afeat=True)
timeit("ga_loss", lambda: ops.ga_loss(out["sub_preds"], out.get("slide_pred"), out["A_out"], y))
