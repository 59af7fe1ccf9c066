"""GPU parity tests of the GROUP training step (acmil_ga_train_step_group / ACMIL_GA.train_step_batch): G slides per step, the
parameter gradients = the MEAN of the per-slide gradients -- what G data-parallel ranks compute per step, on one GPU.
Checked against the oracle's torch-CPU autograd run slide by slide (Step3_WSI_classification_ACMIL.py:189-221 per slide) and
averaged, against the single-slide step, and for run-to-run bit-stability."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ga(sd, k, c, d, di, precision="f16x3", **kw):
    from acmil_amd.architecture.transformer import ACMIL_GA

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    m = ACMIL_GA(Conf, n_token=k, precision=precision, **kw)
    m.load_state_dict(sd)
    return m.cuda()


def _oracle_step(sd, x, u, label, k, n_masked=10, dtype=torch.float32):
    from oracle import ga_oracle as O
    sdg = {n: v.clone().to(dtype).requires_grad_(True) for n, v in sd.items()}
    ref = O.acmil_ga_forward(x.to(dtype).unsqueeze(0), sdg, n_token=k, n_masked_patch=n_masked, mask_drop=0.6,
                             uniforms=None if u is None else u.to(dtype), training=True)
    l0, l1, dl = O.acmil_losses(ref["sub_preds"], ref["slide_pred"], ref["A_out"], label, k)
    (l0 + l1 + dl).backward()
    grads = {n: (v.grad.double() if v.grad is not None else torch.zeros_like(v).double()) for n, v in sdg.items()}
    return ref, (float(l0.detach()), float(l1.detach()), float(dl.detach())), grads


def _bags(rows, d, seed, dtype=torch.float16, sd=None):
    """Synthetic bags.  sd: make the gradient comparison WELL-POSED -- d relu / d pre is discontinuous at 0, and a pre-activation within
    the arithmetic's own error of zero (|pre| < 1e-5; the split-f16 product is 3e-7 from fp32) may land on either side: one such element
    in a heavily attended patch moves dW1 by 1e-2 of its largest entry (measured: seed 101, rows >= 1280).  Patches that hold one are
    replaced by fresh draws, so that the oracle's and the device's ReLU masks are the same function of the data."""
    from oracle import ga_oracle as O
    out = []
    for i, n in enumerate(rows):
        x = O.synthetic_bag(n, d, seed + i)[0].to(dtype)
        if sd is not None:
            w1 = sd["dimreduction.fc1.weight"].double()
            for attempt in range(20):
                bad = ((x.double() @ w1.T).abs() < 1e-5).any(dim=1)
                if not bool(bad.any()):
                    break
                fresh = torch.randn(int(bad.sum()), d, generator=torch.Generator().manual_seed(seed * 1000 + i * 20 + attempt)).to(dtype)
                x[bad] = fresh
        out.append(x)
    return out


CASES = [
    # rows, D, Di, K, C
    ([700, 1300], 512, 256, 5, 7),
    ([257, 4101, 128, 3000, 64, 999, 2048, 1500], 512, 256, 5, 7),        # sum 12 097: 32-row backward tiles
    ([3000, 2900, 3100, 2800, 3200, 2700, 3300, 2600], 512, 256, 5, 2),    # sum 23 600: 64-row backward tiles
    ([900, 333, 1200], 384, 128, 5, 2),
    ([500, 800], 512, 256, 1, 2),
]


@pytest.mark.parametrize("rows,D,Di,K,C", CASES)
def test_group_step_matches_oracle_mean_of_per_slide_gradients(rows, D, Di, K, C):
    from oracle import ga_oracle as O
    G = len(rows)
    sd = O.default_state_dict(D, Di, C, K)
    bags = _bags(rows, D, 100, sd=sd)
    labels = torch.tensor([i % C for i in range(G)])
    gen = torch.Generator().manual_seed(9)
    us = torch.rand(G, K, 10, generator=gen)
    refs, losses_ref, gsum, gsum64 = [], [], None, None
    for b in range(G):
        ref, ls, g = _oracle_step(sd, bags[b].float(), us[b], labels[b:b + 1], K)
        _, _, g64 = _oracle_step(sd, bags[b].float(), us[b], labels[b:b + 1], K, dtype=torch.float64)
        refs.append(ref); losses_ref.append(ls)
        gsum = g if gsum is None else {n: gsum[n] + g[n] for n in g}
        gsum64 = g64 if gsum64 is None else {n: gsum64[n] + g64[n] for n in g64}
    gmean = {n: v / G for n, v in gsum.items()}
    gmean64 = {n: v / G for n, v in gsum64.items()}
    model = _ga(sd, K, C, D, Di, n_masked_patch=10, mask_drop=0.6).train()
    losses, out = model.train_step_batch([b.cuda() for b in bags], labels.cuda(), uniforms=us.cuda())
    grads1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    offs = out["offsets"]
    for b in range(G):
        assert np.array_equal(out["topk_idx"][b].cpu().numpy(), refs[b]["topk_idx"].numpy()), b
        assert np.array_equal(np.sort(out["masked_idx"][b].cpu().numpy(), 1), np.sort(refs[b]["masked_idx"].numpy(), 1)), b
        a = out["A_out"][:, offs[b]:offs[b + 1]].cpu()
        a_ref = refs[b]["A_out"].detach().reshape(K, rows[b])
        assert (a - a_ref).abs().max().item() < 1e-4
        assert int((a == -1e9).sum()) == K * 6
        assert (out["sub_preds"][b].cpu() - refs[b]["sub_preds"].detach()).abs().max().item() < 1e-4
        assert (out["slide_pred"][b].cpu() - refs[b]["slide_pred"].detach().reshape(-1)).abs().max().item() < 1e-4
        for got, want in zip(losses[b, :3].tolist(), losses_ref[b]):
            assert got == pytest.approx(want, abs=2e-5), (b, got, want)
        assert losses[b, 3].item() == pytest.approx(sum(losses_ref[b]), abs=5e-5)
    for name, p in model.named_parameters():
        if K == 1 and name.startswith("classifier."):
            continue          # n_token = 1: the branch-head loss is not built (Step3_WSI_classification_ACMIL.py:201-204)
        # two-sided bound of tests/test_full_size_gpu.py: fp32 autograd is itself up to ~1e-2 (relative to the largest entry) from the
        # same computation in fp64 for dW1 -- the ReLU mask of pre-activations near zero hangs on the last bit of h -- so: as close to
        # the fp64 oracle as the fp32 oracle is (x 1.5), or 2e-4
        scale = gmean64[name].abs().max().item()
        if scale < 1e-9:
            continue
        g = p.grad.cpu().double()
        e64 = (g - gmean64[name]).abs().max().item() / scale
        ref_e = (gmean[name] - gmean64[name]).abs().max().item() / scale
        assert e64 <= max(2e-4, 1.5 * ref_e), (name, e64, ref_e)
    # the same group bag by bag on the GPU (single-slide steps, gradients averaged): 3e-5 of each gradient's largest entry
    ls, _ = model._train_step_group_serial(torch.cat([b.cuda() for b in bags]), rows, labels.cuda(), us.cuda(), model._all_params(), None, None)
    assert torch.equal(ls, losses)
    for n, p in model.named_parameters():
        sc = grads1[n].abs().max().item()
        if sc >= 1e-6:
            assert (p.grad - grads1[n]).abs().max().item() <= 3e-5 * sc, n      # (other split-K chunks of the split-bf16 products)
    # run-to-run: bitwise
    losses2, out2 = model.train_step_batch([b.cuda() for b in bags], labels.cuda(), uniforms=us.cuda())
    assert torch.equal(losses, losses2)
    for n, p in model.named_parameters():
        assert torch.equal(p.grad, grads1[n]), n
    # pre-concatenated rows (no copy) == list of bags
    xcat = torch.cat([b.cuda() for b in bags], 0)
    losses3, _ = model.train_step_batch((xcat, rows), labels.cuda(), uniforms=us.cuda())
    assert torch.equal(losses, losses3)
    for n, p in model.named_parameters():
        assert torch.equal(p.grad, grads1[n]), n


@pytest.mark.parametrize("N", [900, 30000])
def test_group_of_one_is_the_single_slide_step_bit_for_bit(N):
    from oracle import ga_oracle as O
    D, Di, K, C = 512, 256, 5, 7
    sd = O.default_state_dict(D, Di, C, K)
    x = O.synthetic_bag(N, D, 3)[0].half().cuda()
    u = torch.rand(K, 10, generator=torch.Generator().manual_seed(1)).cuda()
    y = torch.tensor([3]).cuda()
    model = _ga(sd, K, C, D, Di, n_masked_patch=10, mask_drop=0.6).train()
    l1, o1 = model.train_step(x.unsqueeze(0), y, uniforms=u)
    g1 = [p.grad.clone() for p in model.parameters()]
    l2, o2 = model.train_step_batch([x], y, uniforms=u.unsqueeze(0))
    assert torch.equal(l1, l2[0])
    assert torch.equal(o1["A_out"], o2["A_out"])
    assert torch.equal(o1["topk_idx"], o2["topk_idx"][0]) and torch.equal(o1["masked_idx"], o2["masked_idx"][0])
    for a, p in zip(g1, model.parameters()):
        assert torch.equal(a, p.grad)


def test_group_step_with_the_optimizer_inside_equals_step_then_optimizer():
    """acmil_ga_train_step_group(adamw): the closing launch applies AdamW on the group's mean gradient; same parameters, moments and
    packed buffer as the group step followed by FlatAdamW.step(), over several steps with a changing learning rate."""
    from oracle import ga_oracle as O
    from acmil_amd.optim import FlatAdamW
    D, Di, K, C = 512, 256, 5, 7
    rows = [600, 1500, 300, 2100]
    sd = O.default_state_dict(D, Di, C, K)
    bags = [b.cuda() for b in _bags(rows, D, 40)]
    labels = torch.tensor([0, 3, 6, 2]).cuda()
    gen = torch.Generator().manual_seed(2)
    us = [torch.rand(len(rows), K, 10, generator=gen).cuda() for _ in range(4)]

    def run(in_step):
        model = _ga(sd, K, C, D, Di, n_masked_patch=10, mask_drop=0.6).train()
        opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-2, on_step=model.invalidate_packed)
        ids = []
        for it in range(4):
            opt.param_groups[0]["lr"] = 1e-3 * (1 + it)
            losses, out = model.train_step_batch(bags, labels, uniforms=us[it], guard_flag=opt.guard_flag, optimizer=opt, track_flag=True,
                                                 in_step=in_step)
            if out["opt_step_id"] is None:
                opt.step(track_flag=True)
            else:
                ids.append(out["opt_step_id"])
            opt.poll_skipped(2)
        opt.poll_skipped(0)
        torch.cuda.synchronize()
        return model, opt, losses, ids

    m1, o1, l1, ids1 = run(True)
    m2, o2, l2, ids2 = run(False)
    assert len(ids1) == 4 and len(ids2) == 0          # the in-step path really ran
    assert torch.equal(l1, l2)
    assert torch.equal(o1.flat, o2.flat) and torch.equal(o1.exp_avg, o2.exp_avg) and torch.equal(o1.exp_avg_sq, o2.exp_avg_sq)
    # the step's private packed buffer holds the updated weights: a fifth step needs no re-pack and gives identical losses
    la, _ = m1.train_step_batch(bags, labels, uniforms=us[0], guard_flag=o1.guard_flag, optimizer=o1, track_flag=True)
    lb, _ = m2.train_step_batch(bags, labels, uniforms=us[0], guard_flag=o2.guard_flag, optimizer=o2, track_flag=True, in_step=False)
    assert torch.equal(la, lb)


def test_group_with_a_bag_smaller_than_n_masked_patch_runs_bag_by_bag():
    """N < n_masked_patch clamps k per bag (transformer.py:313): such a group takes the serial route, same mean gradient."""
    from oracle import ga_oracle as O
    D, Di, K, C = 512, 256, 5, 2
    rows = [7, 400]
    sd = O.default_state_dict(D, Di, C, K)
    bags = _bags(rows, D, 77, sd=sd)
    labels = torch.tensor([1, 0])
    us = [torch.rand(K, 7, generator=torch.Generator().manual_seed(4)), torch.rand(K, 10, generator=torch.Generator().manual_seed(5))]
    gsum = None
    for b in range(2):
        _, _, g = _oracle_step(sd, bags[b].float(), us[b], labels[b:b + 1], K, dtype=torch.float64)
        gsum = g if gsum is None else {n: gsum[n] + g[n] for n in g}
    model = _ga(sd, K, C, D, Di, n_masked_patch=10, mask_drop=0.6).train()
    # (the serial route takes the per-bag draws as a list: k differs per bag)
    losses, out = model.train_step_batch([b.cuda() for b in bags], labels.cuda(), uniforms=[u.cuda() for u in us])
    assert losses.shape == (2, 4)
    for name, p in model.named_parameters():
        want = gsum[name] / 2
        scale = want.abs().max().item()
        if scale < 1e-9:
            continue
        assert (p.grad.cpu().double() - want).abs().max().item() / scale <= 2e-3, name      # (vs fp64; the fp32 oracle's own distance is of this size)
    # and the public entry routes there by itself (device draw: finite losses, gradients present)
    l2, _ = model.train_step_batch([b.cuda() for b in bags], labels.cuda())
    assert l2.shape == (2, 4) and torch.isfinite(l2).all()


def test_flagged_group_is_repeated_in_fp32_bag_by_bag():
    """A bag value outside the f16 range flags the whole group; the repeat runs the exact-fp32 step bag by bag and averages: equals the
    fp32-precision model's group step, finite everywhere."""
    from oracle import ga_oracle as O
    D, Di, K, C = 512, 256, 5, 2
    rows = [300, 500]
    sd = O.default_state_dict(D, Di, C, K)
    bags = [b.float() for b in _bags(rows, D, 11)]
    bags[1][17, 5] = 1e5
    labels = torch.tensor([1, 0]).cuda()
    us = torch.rand(2, K, 10, generator=torch.Generator().manual_seed(8)).cuda()
    m16 = _ga(sd, K, C, D, Di, n_masked_patch=10, mask_drop=0.6).train()
    m32 = _ga(sd, K, C, D, Di, precision="fp32", n_masked_patch=10, mask_drop=0.6).train()
    l16, o16 = m16.train_step_batch([b.cuda() for b in bags], labels, uniforms=us)
    l32, _ = m32.train_step_batch([b.cuda() for b in bags], labels, uniforms=us)
    assert o16.get("range_fallback") and m16.range_fallbacks == 1
    assert torch.isfinite(l16).all() and torch.equal(l16, l32)
    for (n, p), q in zip(m16.named_parameters(), m32.parameters()):
        assert torch.isfinite(p.grad).all() and torch.equal(p.grad, q.grad), n
