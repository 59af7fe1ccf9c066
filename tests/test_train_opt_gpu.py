"""The training step with the optimizer inside (acmil_ga_train_step_adamw, csrc/ga_opt_step.hip): its closing launch -- split-K finish
of the weight gradients + AdamW + re-pack of the updated weights -- against the sequence it replaces (acmil_ga_train_step_rng ->
acmil_adamw_step_report -> acmil_ga_pack_weights), BIT FOR BIT: gradients, parameters, moments, losses, and the packed buffer the next
step reads.  The loop is Step3_WSI_classification_ACMIL.py:189-227 (forward, three losses, backward, optimizer.step())."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(D=512, Di=256, K=5, C=7, n_mask=10, seed=3, plain=False):
    from acmil_amd import train as T
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-2, lr=1e-3, min_lr=0, n_class=C, n_token=K, n_masked_patch=n_mask,
                    mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=D, D_inner=Di)
    dev = torch.device("cuda", 0)
    T.set_seed(seed)
    model = T.build_model(conf).to(dev).train()
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
    if plain:
        opt.pack_hook = None      # the reference sequence: acmil_ga_train_step_rng (packs first) -> acmil_adamw_step_report
    return T, conf, dev, model, bucket, opt


def _bags(n, N, D, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(N + 37 * i, D, generator=g) * 0.5).to(dtype) for i in range(n)]


def _frag_off(row, k, nks, plane):
    return ((((row >> 5) * nks + (k >> 4)) * 2 + plane) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7)


def _layout(D, Di, K, C):
    """Byte ranges the pack kernel defines in the f16x3 packed buffer, the offset of wT16 and the total (csrc/ga_common.h ga_layout,
    restated; the gaps are alignment padding nobody writes)."""
    ND, off, regions = Di // 32, 0, []

    def take(n, pad_to=None):
        nonlocal off
        regions.append((off, n))
        off += n if pad_to is None else pad_to
    take((D // 64) * 8 * ND * 1024 + ND * 32 * 1024)
    take((2 + K) * 128 * 4)
    take(16 * 4)
    take(K * C * Di * 4)
    take(K * C * 4, ((K * C + 3) // 4) * 16)
    take(C * Di * 4)
    take(C * 4, ((C + 3) // 4) * 16)
    off = (off + 255) & ~255
    take(2 * 128 * Di * 4 + 2 * 128 * 4 + 2 * 128 * Di * 4 + 2 * 2 * 128 * Di * 2)
    wT = off
    take(2 * Di * 288 * 2)
    return regions, wT, (off + 255) & ~255


def _packed_equal_where_the_pack_kernel_writes(model, dev):
    """The step's private packed buffer (maintained by the closing launch) against a fresh acmil_ga_pack_weights of the current
    parameters, byte for byte -- except the d_afeat columns of the backward tile kernel's operand, which the step's tail kernel owns."""
    (kept, dims), = [v for k, v in model._step_packed.items() if k[0] == "f16x3"]
    fresh, _ = model._pack_now("f16x3")
    a = kept.view(torch.uint8).cpu().numpy()
    b = fresh.view(torch.uint8).cpu().numpy()
    regions, off, total = _layout(dims.D, dims.Di, dims.K, dims.C)
    assert a.shape == b.shape == (total,)
    m = np.zeros(total, dtype=bool)
    for o, n in regions:
        m[o:o + n] = True
    rows = np.arange(dims.Di)[:, None]
    cols = np.arange(256, 288)[None, :]
    for plane in (0, 1):
        idx = (off + 2 * _frag_off(rows, cols, 18, plane)).ravel()
        m[idx] = False
        m[idx + 1] = False
    bad = np.nonzero((a != b) & m)[0]
    assert bad.size == 0, "packed buffer of the in-step optimizer differs from a fresh pack at %d bytes, first at byte %d" % (bad.size, bad[0])


@pytest.mark.parametrize("D,Di,K,C,N", [(512, 256, 5, 7, 3000), (384, 128, 5, 2, 1500), (512, 256, 1, 2, 900), (512, 256, 3, 4, 700)])
def test_in_step_optimizer_is_bit_identical_to_the_three_launch_sequence(D, Di, K, C, N):
    T, conf, dev, ref_model, ref_bucket, ref_opt = _setup(D, Di, K, C, plain=True)
    _, _, _, model, bucket, opt = _setup(D, Di, K, C)
    model.load_state_dict(ref_model.state_dict())
    if not opt.can_run_in_step():
        pytest.skip("n_token = 1 freezes the branch head (torch skips grad=None parameters): the in-step path is not taken")
    bags = _bags(4, N, D)
    for i, x in enumerate(bags):
        y = torch.tensor([i % C], device=dev)
        xb = x.to(dev).unsqueeze(0)
        for g in ref_opt.param_groups + opt.param_groups:
            g["lr"] = 1e-3 * (1.0 + 0.25 * i)
        l_ref, o_ref = ref_model.train_step(xb, y, guard_flag=ref_opt.guard_flag)
        sid_ref = ref_opt.step(track_flag=True)
        l_new, o_new = model.train_step(xb, y, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True)
        assert o_new["opt_step_id"] == sid_ref, "the step did not apply the optimizer itself"
        assert torch.equal(l_ref, l_new)
        assert torch.equal(o_ref["A_out"], o_new["A_out"])
        assert torch.equal(ref_bucket.flat, bucket.flat), "gradients"
        assert torch.equal(ref_opt.flat, opt.flat), "parameters after AdamW (step %d)" % i
        assert torch.equal(ref_opt.exp_avg, opt.exp_avg) and torch.equal(ref_opt.exp_avg_sq, opt.exp_avg_sq)
        _packed_equal_where_the_pack_kernel_writes(model, dev)
    assert opt.poll_skipped(0) == [] and ref_opt.poll_skipped(0) == []
    assert opt.step_count == ref_opt.step_count == 4
    # only the first step packed from scratch
    assert model._step_pack_key is not None


def test_in_step_optimizer_skips_a_flagged_step_on_the_device_and_carries_on():
    T, conf, dev, model, bucket, opt = _setup(384, 128, 5, 3, n_mask=0)
    _, _, _, ref_model, ref_bucket, ref_opt = _setup(384, 128, 5, 3, n_mask=0, plain=True)
    ref_model.load_state_dict(model.state_dict())
    good = _bags(3, 600, 384)
    bad = good[1].float().clone(); bad[17, 3] = 1.0e5
    y = torch.tensor([1], device=dev)
    seq = [good[0], bad, good[2]]
    ids, ref_ids = [], []
    for x in seq:
        before = opt.flat.clone()
        _, out = model.train_step(x.to(dev).unsqueeze(0), y, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True)
        ids.append(out["opt_step_id"])
        ref_model.train_step(x.to(dev).unsqueeze(0), y, guard_flag=ref_opt.guard_flag)
        ref_ids.append(ref_opt.step(track_flag=True))
        if x is bad:
            assert float(opt.guard_flag) == 1.0 and torch.equal(opt.flat, before)
        assert torch.equal(opt.flat, ref_opt.flat) and torch.equal(opt.exp_avg, ref_opt.exp_avg)
    assert ids == ref_ids and None not in ids
    assert opt.poll_skipped(0) == [ids[1]] and ref_opt.poll_skipped(0) == [ref_ids[1]]
    assert opt.step_count == ref_opt.step_count == 2 and opt.skipped_steps == 1
    _packed_equal_where_the_pack_kernel_writes(model, dev)


def test_anything_else_touching_the_parameters_forces_a_repack():
    """The step's packed buffer is trusted only while the in-step update is the sole writer: a plain optimizer.step(), a
    load_state_dict or an in-place edit in between must be seen (the next step packs from scratch) -- checked through the results."""
    T, conf, dev, model, bucket, opt = _setup(384, 128, 5, 3)
    _, _, _, ref_model, ref_bucket, ref_opt = _setup(384, 128, 5, 3, plain=True)
    ref_model.load_state_dict(model.state_dict())
    bags = _bags(5, 500, 384, seed=4)
    y = torch.tensor([2], device=dev)

    def both(i, disturb=None):
        xb = bags[i].to(dev).unsqueeze(0)
        if disturb:
            disturb(model); disturb(ref_model)
        ref_model.train_step(xb, y, guard_flag=ref_opt.guard_flag)
        ref_opt.step(track_flag=True)
        _, out = model.train_step(xb, y, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True)
        assert out["opt_step_id"] is not None
        assert torch.equal(opt.flat, ref_opt.flat), "step %d" % i

    both(0)
    both(1)

    def edit(m):
        with torch.no_grad():
            m.dimreduction.fc1.weight.mul_(1.01)
    both(2, edit)

    def reload(m):
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["attention.attention_weights.weight"] *= 0.5
        m.load_state_dict(sd)
    both(3, reload)
    # a plain optimizer step between two in-step ones (what train_one_epoch's fp32 repeat of a skipped bag does)
    for o in (opt, ref_opt):
        o.grad.fill_(1e-3)
        o.guard_flag.zero_()      # (the flag is the bucket's last slot)
        o.step()
    both(4)
    opt.poll_skipped(0); ref_opt.poll_skipped(0)


def test_train_one_epoch_takes_the_in_step_path_on_one_gpu_and_matches_the_old_loop():
    from acmil_amd import train as T
    results = []
    for in_step in (True, False):
        _, conf, dev, model, bucket, opt = _setup(384, 128, 5, 3, seed=9)
        if not in_step:
            model.supports_in_step_optimizer = False
        data = T.SyntheticBags(6, 700, 384, 3, seed=5)
        calls = {"n": 0}
        orig = opt.step

        def counting(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        opt.step = counting
        stats = T.train_one_epoch(model, data, opt, dev, 0, conf, bucket, 0, 1, log_every=0)
        assert calls["n"] == (0 if in_step else 6)
        results.append((opt.flat.clone(), stats))
    assert torch.equal(results[0][0], results[1][0])
    assert results[0][1] == results[1][1]


@pytest.mark.parametrize("D,Di,K,C", [(512, 256, 5, 7), (384, 128, 3, 2)])
def test_optimizer_launch_that_also_repacks_matches_the_plain_sequence(D, Di, K, C):
    """The data-parallel shape of the step (the update cannot ride in the step's closing launch: an all-reduce sits in between):
    FlatAdamW.step() goes through ACMIL_GA.adamw_pack_hook -> acmil_ga_adamw_pack (AdamW + re-pack, one launch) and the next
    train_step(optimizer=opt, in_step=False) skips its pack launch.  Bit-identical to pack -> step -> plain AdamW, incl. a skipped
    (range-flagged) step and a foreign edit of the parameters in between."""
    T, conf, dev, ref_model, ref_bucket, ref_opt = _setup(D, Di, K, C, plain=True)
    _, _, _, model, bucket, opt = _setup(D, Di, K, C)
    assert opt.pack_hook is not None
    model.load_state_dict(ref_model.state_dict())
    bags = _bags(5, 800, D, seed=2)
    bad = bags[2].float().clone(); bad[11, 5] = 2.0e5
    seq = [bags[0], bags[1], bad, bags[3], bags[4]]
    packs = {"n": 0}
    from acmil_amd import ops
    orig = ops.ga_train_step

    def spy(*a, **k):
        packs["n"] += int(bool(k.get("repack", True)))
        return orig(*a, **k)
    ops.ga_train_step = spy
    try:
        for i, x in enumerate(seq):
            y = torch.tensor([i % C], device=dev)
            xb = x.to(dev).unsqueeze(0)
            if i == 3:
                for m in (model, ref_model):
                    with torch.no_grad():
                        m.attention.attention_V[0].weight.mul_(0.99)
            ref_model.train_step(xb, y, guard_flag=ref_opt.guard_flag)
            ref_opt.step(track_flag=True)
            n0 = packs["n"]
            _, out = model.train_step(xb, y, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True, in_step=False)
            repacked = packs["n"] - n0
            assert out["opt_step_id"] is None
            opt.step(track_flag=True)
            assert torch.equal(ref_bucket.flat, bucket.flat), "gradients (step %d)" % i
            assert torch.equal(ref_opt.flat, opt.flat) and torch.equal(ref_opt.exp_avg, opt.exp_avg) and torch.equal(ref_opt.exp_avg_sq, opt.exp_avg_sq)
            assert repacked == (1 if i in (0, 3) else 0), "step %d: %d pack launches" % (i, repacked)      # first step, and after the foreign edit
            _packed_equal_where_the_pack_kernel_writes(model, dev)
    finally:
        ops.ga_train_step = orig
    assert opt.poll_skipped(0) == ref_opt.poll_skipped(0) and opt.skipped_steps == 1


@pytest.mark.parametrize("extra_numel", [3, 16])
def test_in_step_optimizer_refuses_a_flat_buffer_it_cannot_serve_and_the_caller_steps_as_before(extra_numel):
    """acmil_ga_train_step_adamw needs the module's parameters to tile the optimizer's flat buffer, 16-byte aligned: an optimizer that
    also owns a foreign parameter in front of them (3 floats: misaligned -> ACMIL_ERR_UNSUPPORTED; 16 floats: aligned but not covered
    -> ACMIL_ERR_SHAPE) is refused BEFORE anything is launched; train_step then returns opt_step_id None, the plain sequence runs and
    gives the same parameters as an optimizer over the module alone (the foreign parameter has zero gradient: untouched but for decay)."""
    from acmil_amd.optim import FlatAdamW
    T, conf, dev, ref_model, ref_bucket, ref_opt = _setup(384, 128, 5, 3, plain=True)
    _, _, _, model, _, _ = _setup(384, 128, 5, 3)
    model.load_state_dict(ref_model.state_dict())
    extra = torch.nn.Parameter(torch.zeros(extra_numel, device=dev))
    opt = FlatAdamW([extra] + list(model.parameters()), lr=conf.lr, weight_decay=conf.wd, on_step=model.invalidate_packed)
    opt.pack_hook = model.adamw_pack_hook(opt)
    bags = _bags(3, 500, 384, seed=7)
    for i, x in enumerate(bags):
        y = torch.tensor([i % 3], device=dev)
        xb = x.to(dev).unsqueeze(0)
        ref_model.train_step(xb, y, guard_flag=ref_opt.guard_flag)
        ref_opt.step(track_flag=True)
        _, out = model.train_step(xb, y, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True)
        assert out["opt_step_id"] is None
        opt.step(track_flag=True)
        assert torch.equal(opt.flat[extra_numel:], ref_opt.flat), "step %d" % i
    assert model._opt_in_step_refused
    assert opt.poll_skipped(0) == [] and float(extra.detach().abs().sum()) == 0.0


@pytest.mark.parametrize("in_step", [True, False])
def test_ten_step_trajectory_matches_the_reference_loop(in_step):
    """train.train_one_epoch on the GPU -- cosine / warm-up schedule per iteration, FlatAdamW (its update inside the step's closing launch,
    or as its own launch), lagged range guard -- driven over the slide order and the STKIM draws of the fixture that the REFERENCE's
    own train_one_epoch produced (tests/golden/make_golden_trajectory.py: ten steps, two epochs over five slides): per-step losses and
    the FINAL parameters (Step3_WSI_classification_ACMIL.py:175-235, utils/utils.py:250-262)."""
    from conftest import load_golden, trajectory_bags, trajectory_stable
    from acmil_amd import train as T
    case, sd = load_golden("ga_trajectory_d512_k5_c7")
    bags = trajectory_bags(case)
    dev = torch.device("cuda", 0)
    conf = T.Struct(train_epoch=int(case["train_epoch"]), warmup_epoch=int(case["warmup_epoch"]), wd=float(case["wd"]), lr=float(case["lr"]),
                    min_lr=0, n_class=7, n_token=5, n_masked_patch=10, mask_drop=0.6, arch="ga", precision="f16x3", seed=0,
                    D_feat=512, D_inner=256)
    model = T.build_model(conf)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    opt = T.make_optimizer(model, conf, dev, None, lr=conf.lr)
    if not in_step:
        model.supports_in_step_optimizer = False
    data = [{"input": b[0], "label": int(l)} for b, l in zip(bags, case["labels"])]
    uni = torch.from_numpy(case["uniforms"]).to(dev)
    step0 = 0
    per_epoch = []
    for epoch, order in enumerate(case["orders"].tolist()):
        base = step0
        stats = T.train_one_epoch(model, data, opt, dev, epoch, conf, log_every=0, order=order,
                                  uniforms_fn=lambda e, it, base=base: uni[base + it])
        per_epoch.append(stats)
        step0 += len(order)
    assert opt.skipped_steps == 0 and opt.step_count == 10
    if in_step:
        assert model.__dict__.get("_opt_in_step_refused") is None
    n = len(case["orders"][0])
    for e, stats in enumerate(per_epoch):      # epoch means of the two cross-entropies the reference's criterion saw
        ref = case["losses"][e * n:(e + 1) * n]
        assert stats["sub_loss"] == pytest.approx(float(ref[:, 0].mean()), abs=2e-5)
        assert stats["slide_loss"] == pytest.approx(float(ref[:, 1].mean()), abs=2e-5)
    # final parameters.  The device's gradients are within ~1e-5 of each tensor's LARGEST gradient entry (split-bf16 products, other
    # summation orders); Adam divides by |g|, so an element's step error is ~ lr * (gradient error / its own |g|): 1e-6 where every
    # step's |g| is within 100x of the tensor's largest (70 % of W1, 82 - 100 % elsewhere), 1e-5 within 1000x (92 - 100 %), and the mean
    # over ALL elements -- noise-gradient elements included, attention_weights.bias (analytically zero gradient) excepted -- 1e-7.
    # W1 rows of the hidden units that came within 2e-6 of the ReLU kink on some patch of some step (the fixture's `minpre`, from the
    # reference run: 7 of 256) are compared apart: on which side of zero such a pre-activation falls is decided by the last bits of
    # a 512-term dot product, i.e. differs between any two fp32 summation orders, and it switches that patch's whole contribution to the row
    kink = case["minpre"] < 2e-6
    assert kink.sum() <= 8
    for name, p in model.named_parameters():
        diff = np.abs(p.detach().cpu().numpy() - case["final." + name])
        a, b = trajectory_stable(case, name, rel=1e-2), trajectory_stable(case, name, rel=1e-3)
        if name == "dimreduction.fc1.weight":
            assert diff[kink].max(initial=0.0) <= float(case["lrs"].sum())        # (cannot move further than the learning rates add up to)
            diff, a, b = diff[~kink], a[~kink], b[~kink]
        if name != "attention.attention_weights.bias":
            assert a.mean() >= 0.65 and b.mean() >= 0.9, (name, a.mean(), b.mean())
            assert diff.mean() <= 1e-7, (name, float(diff.mean()))
        assert diff[a].max(initial=0.0) <= 1e-6, (name, float(diff[a].max(initial=0.0)), float(a.mean()))
        assert diff[b].max(initial=0.0) <= 1e-5, (name, float(diff[b].max(initial=0.0)), float(b.mean()))
