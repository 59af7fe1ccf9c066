"""GPU end-to-end test of the Step3-style trainer on synthetic bags (HIP forward + backward, AdamW, eval, checkpoint)."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_training_reduces_loss_and_checkpoints():
    from acmil_amd import train as T
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-5, lr=2e-3, min_lr=0, n_class=3, n_token=5, n_masked_patch=10,
                    mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=384, D_inner=128)
    T.set_seed(1)
    device = torch.device("cuda", 0)
    train = T.SyntheticBags(24, (300, 700), 384, 3, seed=1)
    val = T.SyntheticBags(12, 500, 384, 3, seed=2)
    model = T.build_model(conf).to(device)
    opt = torch.optim.AdamW(model.parameters(), lr=0.001, weight_decay=conf.wd)
    first = T.train_one_epoch(model, train, opt, device, 0, conf, log_every=0)
    for epoch in (1, 2):
        last = T.train_one_epoch(model, train, opt, device, epoch, conf, log_every=0)
    assert last["slide_loss"] < first["slide_loss"] - 0.05, (first, last)
    auroc, acc, f1, loss = T.evaluate(model, val, device, conf, "Val")
    assert 0.0 <= auroc <= 1.0 and 0.0 <= acc <= 100.0 and loss == loss
    assert auroc > 0.6          # the synthetic class signal is learnable
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "checkpoint-best.pth")
        T.save_model(conf, 2, model, opt, path)
        ck = torch.load(path, weights_only=False)
        m2 = T.build_model(conf)
        m2.load_state_dict(ck["model"])
        m2 = m2.to(device).eval()
        x = val[0]["input"].to(device).unsqueeze(0)
        with torch.no_grad():
            a = model.eval()(x)[1]
            b = m2(x)[1]
        assert torch.equal(a, b)


@pytest.mark.parametrize("arch", ["transmil", "abmil"])
def test_single_head_archs_train(arch):
    """`--arch transmil` / `--arch abmil`: criterion(output, label) loop (Step3_WSI_classification.py / engine.py:19-21)."""
    from acmil_amd import train as T
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-5, lr=1e-3, min_lr=0, n_class=2, n_token=1, n_masked_patch=0,
                    mask_drop=0.0, arch=arch, precision="f16x3", seed=1, D_feat=384, D_inner=128)
    T.set_seed(2)
    device = torch.device("cuda", 0)
    train = T.SyntheticBags(16, (200, 400), 384, 2, seed=3)
    model = T.build_model(conf).to(device)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=conf.wd)
    first = T.train_one_epoch(model, train, opt, device, 0, conf, log_every=0, fused=False)
    for epoch in (1, 2):
        last = T.train_one_epoch(model, train, opt, device, epoch, conf, log_every=0, fused=False)
    assert last["slide_loss"] == last["slide_loss"] and last["slide_loss"] < first["slide_loss"], (first, last)
    auroc, acc, f1, loss = T.evaluate(model, train, device, conf, "Train")
    assert 0.0 <= auroc <= 1.0 and loss == loss


def test_flat_adamw_matches_torch_adamw():
    """acmil_amd.optim.FlatAdamW (one launch over flat buffers) against torch.optim.AdamW over 25 steps with a varying lr."""
    from acmil_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(256, 512), (128,), (5, 128), (7, 256), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=1e-3, weight_decay=1e-2)
    mine = FlatAdamW(my_p, lr=1e-3, weight_decay=1e-2)
    for step in range(25):
        lr = 1e-3 * (0.5 + 0.5 * (step % 5) / 5)
        ref.param_groups[0]["lr"] = lr; mine.param_groups[0]["lr"] = lr
        for a, b in zip(ref_p, my_p):
            grad = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (-(step % 4)))
            a.grad = grad.clone(); b.grad.copy_(grad)
        ref.step(); mine.step()
    for a, b in zip(ref_p, my_p):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
    # the checkpoint entry has torch.optim.AdamW's layout: a torch optimizer loads it and continues identically
    sd = mine.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(my_p) and float(sd["state"][0]["step"]) == 25.0
    twin_p = [torch.nn.Parameter(p.detach().clone()) for p in my_p]
    twin = torch.optim.AdamW(twin_p, lr=1e-3, weight_decay=1e-2)
    twin.load_state_dict(sd)
    back = FlatAdamW([torch.nn.Parameter(p.detach().clone()) for p in my_p], lr=1e-3, weight_decay=1e-2)
    back.load_state_dict(ref.state_dict())                      # and the reverse: a torch (reference) checkpoint into FlatAdamW
    for a, b, c in zip(ref_p, twin_p, back.params):
        grad = torch.randn(a.shape, generator=g).cuda()
        a.grad = grad.clone(); b.grad = grad.clone(); c.grad.copy_(grad)
    ref.step(); twin.step(); back.step()
    for a, b, c in zip(ref_p, twin_p, back.params):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
        assert (a - c).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
    # frozen ranges (parameters without gradient) are left untouched, as torch does for grad=None
    fz = FlatAdamW([torch.nn.Parameter(p.detach().clone()) for p in my_p], lr=1e-2, weight_decay=0.5)
    fz.set_frozen([fz.params[1]])
    before = fz.params[1].detach().clone()
    fz.grad.zero_(); fz.step()
    assert torch.equal(fz.params[1], before) and not torch.equal(fz.params[0], my_p[0])
    assert 1 not in fz.state_dict()["state"]


@pytest.mark.parametrize("arch,extra", [("ga", ["--n_token", "5", "--n_masked_patch", "10", "--mask_drop", "0.6"]), ("transmil", []),
                                        ("mha", ["--n_token", "5", "--n_masked_patch", "10", "--mask_drop", "0.6"])])
def test_trainer_main_end_to_end(arch, extra, tmp_path):
    """python -m acmil_amd.train equivalent: synthetic bags, two epochs, checkpoints written (Step3-style main)."""
    from acmil_amd import train as T
    out = str(tmp_path / arch)
    T.main(["--arch", arch, "--synthetic_slides", "16", "--synthetic_patches", "600", "--train_epoch", "2", "--out_dir", out] + extra)
    ck = torch.load(os.path.join(out, "checkpoint-last.pth"), weights_only=False)
    assert set(ck) >= {"model", "optimizer", "epoch", "config"} and ck["epoch"] == 1


def test_fused_step_shares_bucket_with_flat_adamw():
    """The wiring of bench.py --workload train / train.main on one GPU: GradBucket first, FlatAdamW on the SAME flat buffer
    (make_optimizer(..., bucket)), ACMIL_GA.train_step writing its gradients straight into the bucket views, one optimizer
    launch.  Checked against torch.optim.AdamW stepping a twin model on the same gradients."""
    from acmil_amd import train as T
    from acmil_amd.optim import FlatAdamW
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-2, lr=1e-3, min_lr=0, n_class=3, n_token=5, n_masked_patch=10,
                    mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=384, D_inner=128)
    dev = torch.device("cuda", 0)
    T.set_seed(3)
    model = T.build_model(conf).to(dev).train()
    twin = T.build_model(conf).to(dev).train()
    twin.load_state_dict(model.state_dict())
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
    assert isinstance(opt, FlatAdamW) and opt.grad.data_ptr() == bucket.flat.data_ptr()
    ref = torch.optim.AdamW(twin.parameters(), lr=conf.lr, weight_decay=conf.wd)
    bags = T.SyntheticBags(4, 700, 384, 3, seed=5)
    for i in range(4):
        x = bags[i]["input"].to(dev).unsqueeze(0)
        y = torch.tensor([bags[i]["label"]], device=dev)
        model.train_step(x, y)
        bucket.sync_from_grads()
        assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() and
                   p.grad.data_ptr() < bucket.flat.data_ptr() + 4 * bucket.numel for p in model.parameters() if p.requires_grad)
        bucket.allreduce_mean(1)
        for pt, pm in zip(twin.parameters(), model.parameters()):      # the twin steps on the very same gradients
            pt.grad = pm.grad.detach().clone()
        assert float(bucket.flat.abs().sum()) > 0.0
        opt.step()
        ref.step()
    for (n, pm), pt in zip(model.named_parameters(), twin.parameters()):
        assert (pm - pt).abs().max().item() <= 3e-6 * max(1.0, pt.abs().max().item()), n


class _ListBags:
    """A dataset of given (bag, label) pairs with the interface train_one_epoch reads (item -> {'input', 'label'})."""
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        x, y = self.items[i]
        return {"input": x, "label": y}


def _guard_setup(seed=11):
    from acmil_amd import train as T
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-2, lr=1e-3, min_lr=0, n_class=3, n_token=5, n_masked_patch=0,
                    mask_drop=0.0, arch="ga", precision="f16x3", seed=1, D_feat=384, D_inner=128)
    dev = torch.device("cuda", 0)
    T.set_seed(seed)
    model = T.build_model(conf).to(dev).train()
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
    return T, conf, dev, model, bucket, opt


def test_lagged_range_guard_skips_on_device_and_reports_late():
    """The no-read-back guard: a bag outside the f16 range leaves flag = 1 on the device, the optimizer launch changes NOTHING
    for that step (parameters, moments), poll_skipped names the step afterwards and puts the bias-correction count back; an
    in-range bag leaves flag = 0 and is applied."""
    T, conf, dev, model, bucket, opt = _guard_setup()
    g = torch.Generator().manual_seed(0)
    good = torch.randn(500, 384, generator=g).half()
    bad = good.float().clone(); bad[17, 3] = 1.0e5
    y = torch.tensor([1], device=dev)
    before = opt.flat.clone()
    model.train_step(bad.to(dev).unsqueeze(0), y, guard_flag=opt.guard_flag)
    assert float(opt.guard_flag) == 1.0
    sid = opt.step(track_flag=True)
    assert torch.equal(opt.flat, before) and float(opt.exp_avg.abs().sum()) == 0.0
    assert opt.poll_skipped(0) == [sid] and opt.step_count == 0 and opt.skipped_steps == 1
    model.train_step(good.to(dev).unsqueeze(0), y, guard_flag=opt.guard_flag)
    assert float(opt.guard_flag) == 0.0
    opt.step(track_flag=True)
    assert not torch.equal(opt.flat, before) and opt.poll_skipped(0) == [] and opt.step_count == 1
    # the fp32 repeat of the flagged bag gives exactly what the read-back guard's fp32 fallback gives
    T2, _, _, twin, bucket2, opt2 = _guard_setup()
    twin.load_state_dict(model.state_dict())
    model.train_step(bad.to(dev).unsqueeze(0), y, guard_flag=opt.guard_flag, precision="fp32")
    twin.train_step(bad.to(dev).unsqueeze(0), y)
    assert twin.range_fallbacks == 1 and float(opt.guard_flag) == 0.0
    assert torch.equal(bucket.flat[:bucket.numel], bucket2.flat[:bucket2.numel])


def test_evaluate_batched_matches_per_slide_and_guards_lagged():
    """evaluate() groups staged bags into acmil_ga_forward_batch launches of up to 16 and reads the split-f16 range word one batch
    late: per-slide probabilities / losses / div_loss equal the one-slide-per-call loop (Step3_WSI_classification_ACMIL.py:253-268),
    also with ragged N, more bags than one batch (EVAL_BATCH = 64) and ONE bag outside the f16 range (its batch is repeated in fp32)."""
    T, conf, dev, model, bucket, opt = _guard_setup(seed=3)
    g = torch.Generator().manual_seed(4)
    bags = [(torch.randn(300 + 23 * i, 384, generator=g).half(), i % 3) for i in range(T.EVAL_BATCH + 9)]
    bad = bags[20][0].float().clone(); bad[11, 5] = 2.0e5
    bags[20] = (bad, bags[20][1])
    data = _ListBags(bags)
    d_b, d_s = {}, {}
    model.range_fallbacks = 0
    res_b = T.evaluate(model, data, dev, conf, "Val", batched=True, detail=d_b)
    assert model.range_fallbacks == 1                      # exactly the batch that holds bag 20
    res_s = T.evaluate(model, data, dev, conf, "Val", batched=False, detail=d_s)
    assert model.range_fallbacks == 2                      # the per-slide loop repeats just that slide
    assert torch.isfinite(d_b["prob"]).all() and torch.isfinite(d_b["div"]).all()
    # batches of a flagged bag run in fp32 arithmetic, the per-slide loop only that bag: same results to the parity bound
    assert (d_b["prob"] - d_s["prob"]).abs().max().item() < 1e-5
    assert (d_b["loss"] - d_s["loss"]).abs().max().item() < 1e-5
    assert (d_b["div"] - d_s["div"]).abs().max().item() < 1e-4 * max(1.0, d_s["div"].abs().max().item())
    for a, b in zip(res_b, res_s):
        assert abs(a - b) < 1e-5
    # bags of batches without a flagged member are bit-identical to the per-slide launch (same kernel, same tile order per bag)
    keep = [i for i in range(len(bags)) if i // T.EVAL_BATCH != 20 // T.EVAL_BATCH]
    assert len(keep) == 9 and torch.equal(d_b["prob"][keep], d_s["prob"][keep])


def test_train_one_epoch_repeats_a_flagged_bag_in_fp32():
    """train_one_epoch with the lagged guard: the out-of-range bag is skipped by the device, found two steps later and trained
    in fp32.  The model ends where a run ends that sees the same bags with the flagged one moved to where its repeat happened
    (same arithmetic per step; only the order of one bag differs from the reference's loop)."""
    T, conf, dev, model, bucket, opt = _guard_setup(seed=21)
    g = torch.Generator().manual_seed(1)
    bags = [(torch.randn(400 + 37 * i, 384, generator=g).half(), i % 3) for i in range(6)]
    bad = bags[2][0].float().clone(); bad[5, 7] = 3.0e5
    data = _ListBags(bags[:2] + [(bad, bags[2][1])] + bags[3:])
    conf.seed = 0
    order = T.epoch_order(len(data), 0, conf.seed, True, 0, 1)
    stats = T.train_one_epoch(model, data, opt, dev, 0, conf, bucket=bucket, log_every=0)
    assert opt.skipped_steps == 1 and all(v == v for v in stats.values())           # losses finite: the skipped step's are not summed
    # replica: same order and learning-rate schedule, the flagged bag trained (fp32) right after the step at which the lagged
    # loop found it (two iterations later, with that iteration's learning rate)
    T2, conf2, _, twin, bucket2, opt2 = _guard_setup(seed=21)
    pos = list(order).index(2)
    found = min(pos + 2, len(order) - 1)
    for it, i in enumerate(order):
        T.adjust_learning_rate(opt2, 0 + it / len(order), conf2)
        if i != 2:
            twin.train_step(data[i]["input"].to(dev).unsqueeze(0), torch.tensor([data[i]["label"]], device=dev))
            opt2.step()
        if it == found:
            twin.train_step(data[2]["input"].to(dev).unsqueeze(0), torch.tensor([data[2]["label"]], device=dev), precision="fp32")
            opt2.step()
    for (n, pm), pt in zip(model.named_parameters(), twin.parameters()):
        assert (pm - pt).abs().max().item() <= 2e-5 * max(1.0, pt.abs().max().item()), n


def test_train_one_epoch_with_bags_per_step_runs_group_steps_and_repeats_a_flagged_group():
    """conf.bags_per_step = 4: staged groups (rows of a group copied back to back into one ring slot), one group step + AdamW per 4
    slides with the optimizer inside the step's closing launch; the group that holds an out-of-range bag is skipped on the device, found
    two steps later and trained again in exact fp32 (bag by bag, mean gradient).  A replica built from direct train_step_batch calls
    in the same order ends on the same parameters."""
    T, conf, dev, model, bucket, opt = _guard_setup(seed=31)
    conf.n_masked_patch, conf.mask_drop = 10, 0.6
    model.n_masked_patch, model.mask_drop = 10, 0.6
    g = torch.Generator().manual_seed(2)
    bags = [(torch.randn(300 + 41 * i, 384, generator=g).half(), i % 3) for i in range(14)]      # 14 slides: 3 groups of 4 + one of 2
    bad = bags[5][0].float().clone(); bad[9, 2] = 4.0e5
    bags[5] = (bad, bags[5][1])                        # (fp32 storage: 4e5 is not an fp16 value; its group is widened to fp32)
    data = _ListBags(bags)
    conf.seed, conf.bags_per_step = 0, 4
    order = T.epoch_order(len(data), 0, conf.seed, True, 0, 1)
    groups = [order[i:i + 4] for i in range(0, len(order), 4)]
    stats = T.train_one_epoch(model, data, opt, dev, 0, conf, bucket=bucket, log_every=0)
    assert opt.skipped_steps == 1 and all(v == v for v in stats.values())
    assert model.__dict__.get("_opt_in_step_refused") is None
    T2, conf2, _, twin, bucket2, opt2 = _guard_setup(seed=31)
    twin.n_masked_patch, twin.mask_drop = 10, 0.6
    twin._rng_seed, twin._rng_count = model._rng_seed, 0            # the device draws are keyed on (seed, masked-forward count)
    gbad = [gi for gi, grp in enumerate(groups) if 5 in grp][0]
    found = min(gbad + 2, len(groups) - 1)

    def step(idx, precision=None):
        xs = [data[i]["input"].to(dev) for i in idx]
        ys = torch.tensor([data[i]["label"] for i in idx], device=dev)
        if any(x.dtype != xs[0].dtype for x in xs):
            xs = [x.float() for x in xs]
        twin.train_step_batch(xs, ys, precision=precision)
        opt2.step()

    for it, grp in enumerate(groups):
        T.adjust_learning_rate(opt2, 0 + it / len(groups), conf2)
        if it != gbad:
            step(grp)
        else:
            twin._next_rng()                                          # the skipped step consumed one draw
        if it == found:
            step(groups[gbad], precision="fp32")
    for (n, pm), pt in zip(model.named_parameters(), twin.parameters()):
        assert (pm - pt).abs().max().item() <= 2e-5 * max(1.0, pt.abs().max().item()), n


def test_rccl_single_rank_bucket_allreduce_and_broadcast():
    """The collective calls of the data-parallel path on a real RCCL communicator (one rank: a multi-GPU node is not part of
    the test hardware): `torch.distributed` backend nccl = RCCL, parameter broadcast, the flat gradient bucket INCLUDING its
    range-flag slot through all_reduce, then the optimizer launch on the same buffer."""
    import socket
    import torch.distributed as dist
    from acmil_amd import train as T
    if dist.is_initialized():
        pytest.skip("a process group already exists in this interpreter")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        T2, conf, dev, model, bucket, opt = _guard_setup(seed=5)
        T.broadcast_parameters(model, world=2)                     # world > 1 takes the collective branch; the group has one rank
        x = torch.randn(1, 600, 384, device=dev).half()
        y = torch.tensor([2], device=dev)
        model.train_step(x, y, guard_flag=opt.guard_flag)
        bucket.sync_from_grads()
        before = bucket.flat.clone()
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)         # what GradBucket.allreduce_mean issues
        torch.cuda.synchronize()
        assert torch.equal(bucket.flat, before) and float(bucket.flag) == 0.0
        p0 = opt.flat.clone()
        opt.step(track_flag=True)
        assert opt.poll_skipped(0) == [] and not torch.equal(opt.flat, p0)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (switches itself on where they exist)")
def test_two_rank_rccl_training_steps():
    """Two real ranks on RCCL: train_step -> GradBucket.allreduce_mean -> FlatAdamW.step for three steps incl. one range-flagged bag
    on one rank; identical parameters on both ranks, equal to one process on the averaged gradients (tests/dist_worker_nccl.py)."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker_nccl.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:]


def test_two_rank_direct_reduce_on_one_gpu():
    """The one-shot direct gradient reduction fused into the optimizer launch (acmil_amd/peer.py, csrc/peer.hip): two ranks -- sharing
    this box's GPU when it has only one -- map each other's gradient slots through CUDA IPC; parameters bit-identical across ranks and
    equal to one process on the averaged gradients, flagged step skipped on both, 40 unsynchronised steps (tests/dist_worker_peer.py)."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker_peer.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "PEER_OK" in r.stdout, r.stdout[-3000:]


def test_peer_wait_times_out_cleanly_and_stays_out():
    """acmil_adamw_step_peer with a peer whose flag never arrives: after timeout_s the launch sets the error word and changes NOTHING;
    every later launch returns at once (one timeout per dead peer, not one per step); a published peer makes the same call succeed."""
    import ctypes
    from acmil_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n = 1000
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g).to(dev); p0 = p.clone()
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    slots = torch.randn(2, n + 1, generator=g).to(dev)
    slots[:, n] = 0.0                                         # range flag element
    flags = torch.zeros(8, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    skipped = torch.zeros(1, dtype=torch.int32, device=dev)
    sp = (ctypes.c_void_p * 2)(slots[0].data_ptr(), slots[1].data_ptr())
    st = torch.cuda.current_stream().cuda_stream

    def step(step_id, timeout):
        rc = lib.acmil_adamw_step_peer(p.data_ptr(), m.data_ptr(), v.data_ptr(), n, sp, flags.data_ptr(), 2, 0, step_id, timeout, err.data_ptr(),
                                       1e-2, 0.9, 0.999, 1e-8, 0.0, 1, 1, skipped.data_ptr(), None, None, st)
        assert rc == 0
        torch.cuda.synchronize()
    import time
    t0 = time.time(); step(1, 0.05); t_first = time.time() - t0          # rank 0 waits for rank 1's flag: never raised
    assert int(err.item()) == 1 and torch.equal(p, p0) and 0.04 < t_first < 2.0
    t0 = time.time(); step(2, 0.05); t_second = time.time() - t0         # error word set: returns at once
    assert torch.equal(p, p0) and t_second < 0.03
    err.zero_(); flags[1] = 1                                            # the peer has published step 1
    step(1, 0.05)
    assert int(err.item()) == 0 and not torch.equal(p, p0)
    gavg = (slots[0, :n] + slots[1, :n]) / 2
    ref = torch.optim.AdamW([torch.nn.Parameter(p0.clone())], lr=1e-2, weight_decay=0.0)
    ref.param_groups[0]["params"][0].grad = gavg.clone()
    ref.step()
    assert (ref.param_groups[0]["params"][0].data - p).abs().max() < 1e-6

