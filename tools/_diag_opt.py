import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_train_opt_gpu import _setup, _bags
T, conf, dev, ref_model, ref_bucket, ref_opt = _setup(512, 256, 5, 7)
_, _, _, model, bucket, opt = _setup(512, 256, 5, 7)
model.load_state_dict(ref_model.state_dict())
bags = _bags(2, 3000, 512)
for i, x in enumerate(bags):
    y = torch.tensor([i], device=dev); xb = x.to(dev).unsqueeze(0)
    ref_model.train_step(xb, y, guard_flag=ref_opt.guard_flag); ref_opt.step(track_flag=True)
    _, o = model.train_step(xb, y, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True)
    print("step", i, o["opt_step_id"])
    off = 0
    for (n, p) in model.named_parameters():
        k = p.numel()
        for nm, a, b in (("g", bucket.flat, ref_bucket.flat), ("p", opt.flat, ref_opt.flat), ("m", opt.exp_avg, ref_opt.exp_avg), ("v", opt.exp_avg_sq, ref_opt.exp_avg_sq)):
            d = (a[off:off + k] - b[off:off + k]).abs().max().item()
            if d != 0: print("  ", n, nm, d, "scale", b[off:off+k].abs().max().item(), "nbad", int((a[off:off + k] != b[off:off + k]).sum()), "of", k)
        off += k
