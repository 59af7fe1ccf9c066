"""The one-wave-per-SIMD fused forward (csrc/ga_forward_kernel_v3.h) against the CPU oracle: the wide families it was built for
(CLIP-L 768 -> 384, UNI 1024 -> 512; Step3_WSI_classification_ACMIL.py:78-87) at tile-boundary bag sizes, all three storage
formats, ragged batches, the score pass of a training step, and the 64-patch wave tile of the D_inner = 256 family (A/B build)."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ab_environ

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(d, di, k, c, seed=0, **kw):
    from acmil_amd import synthetic as S
    from acmil_amd.architecture.transformer import ACMIL_GA

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    m = ACMIL_GA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6, **kw)
    sd = S.ga_state_dict(d, di, c, k, seed=seed)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


@pytest.mark.parametrize("d,di", [(768, 384), (1024, 512)])
@pytest.mark.parametrize("k,c", [(5, 2), (1, 7), (3, 3)])
def test_wide_fused_forward_matches_oracle(d, di, k, c):
    """N = 1, a partial wave block, one row past a 128-patch tile, several tiles, more tiles than CUs (dynamic draws)."""
    from oracle import ga_oracle as O
    model, sd = _model(d, di, k, c, seed=d + k)
    assert model._is_wide_fused()
    for i, n in enumerate([1, 31, 129, 640, 4097, 40000]):
        x = O.synthetic_bag(n, d, slide_idx=70 + i)
        with torch.no_grad():
            sub, slide, a = model(x.cuda())
        ref = O.acmil_ga_forward(x, sd, n_token=k)
        assert (a.cpu() - ref["A_out"]).abs().max().item() < TOL, n
        assert (sub.cpu() - ref["sub_preds"]).abs().max().item() < TOL, n
        assert (slide[0].cpu() - ref["slide_pred"]).abs().max().item() < TOL, n
        kk = min(10, n)
        assert torch.equal(torch.topk(a.cpu(), kk, dim=-1).indices, torch.topk(ref["A_out"], kk, dim=-1).indices), n
    assert model.range_fallbacks == 0


@pytest.mark.parametrize("d,di", [(768, 384), (1024, 512)])
def test_wide_fused_16bit_bags_and_ragged_batch(d, di):
    """fp16 / bf16 bags are converted in registers (no lo plane): bit-identical to the fp32 launch of the same values; a ragged
    batch in one launch equals the per-bag launches bit for bit (same 128-patch tiles, fixed-order merge)."""
    from oracle import ga_oracle as O
    model, sd = _model(d, di, 5, 2, seed=3)
    ns = [300, 1, 5000, 129, 17000]
    bags = [O.synthetic_bag(n, d, slide_idx=90 + i)[0] for i, n in enumerate(ns)]
    for cast in (torch.float16, torch.bfloat16):
        lo = [b.to(cast).cuda() for b in bags]
        with torch.no_grad():
            single = [model(b.unsqueeze(0)) for b in lo]
            widened = [model(b.float().unsqueeze(0)) for b in lo]
            batch = model.forward_batch(lo)
        for (s0, b0, a0), (s1, b1, a1), (s2, b2, a2) in zip(single, widened, batch):
            assert torch.equal(a0, a1) and torch.equal(s0, s1) and torch.equal(b0, b1)
            assert torch.equal(a0, a2) and torch.equal(s0, s2) and torch.equal(b0, b2)
    ref = O.acmil_ga_forward(bags[2].half().float().unsqueeze(0), sd, n_token=5)
    with torch.no_grad():
        sub, slide, a = model(bags[2].half().cuda().unsqueeze(0))
    assert (a.cpu() - ref["A_out"]).abs().max().item() < TOL and (sub.cpu() - ref["sub_preds"]).abs().max().item() < TOL


def test_wide_fused_range_guard_falls_back_to_fp32():
    """A bag value outside the f16 range: the fused launch flags it, the module redoes the bag op by op in fp32 arithmetic."""
    from oracle import ga_oracle as O
    model, sd = _model(1024, 512, 5, 2, seed=1)
    x = O.synthetic_bag(700, 1024, slide_idx=5)
    x[0, 123, 7] = 1.0e5
    with torch.no_grad():
        sub, slide, a = model(x.cuda())
    ref = O.acmil_ga_forward(x, sd, n_token=5)
    assert model.range_fallbacks == 1 and torch.isfinite(a).all()
    assert (a.cpu() - ref["A_out"]).abs().max().item() < 2e-2 * ref["A_out"].abs().max().item() + TOL      # fp32 arithmetic on 1e5-sized operands
    assert (sub.cpu() - ref["sub_preds"]).abs().max().item() < 1e-3


@pytest.mark.parametrize("d,di", [(768, 384), (1024, 512)])
def test_wide_score_pass_saves_h_and_trains(d, di):
    """Score pass of a training step at the wide widths through the fused kernel: A and h (fp32 rows, relu applied) against the
    oracle's projection, then one op-by-op training step against the oracle's autograd."""
    from acmil_amd import ops
    from oracle import ga_oracle as O
    model, sd = _model(d, di, 5, 3, seed=11)
    x = O.synthetic_bag(3000, d, slide_idx=3)
    packed, dims = model._packed()
    A, h = ops.ga_scores(x[0].cuda(), packed, dims, "f16x3")
    h_ref = torch.relu(x[0].double() @ sd["dimreduction.fc1.weight"].double().T).float()
    assert (h.cpu() - h_ref).abs().max().item() < 2e-5 * max(1.0, h_ref.abs().max().item())
    ref = O.acmil_ga_forward(x, sd, n_token=5)
    assert (A.cpu() - ref["A_out"][0]).abs().max().item() < TOL
    model.train()
    u = torch.rand(5, 10, generator=torch.Generator().manual_seed(4)).cuda()
    y = torch.tensor([2], device="cuda")
    losses, out = model.train_step(x.cuda(), y, uniforms=u)
    # the same step through the oracle's autograd (transformer.py:305-330 + Step3_WSI_classification_ACMIL.py:201-216)
    sdg = {n: v.clone().requires_grad_(True) for n, v in sd.items()}
    r = O.acmil_ga_forward(x, sdg, n_token=5, n_masked_patch=10, mask_drop=0.6, training=True, uniforms=u.cpu())
    l0, l1, dl = O.acmil_losses(r["sub_preds"], r["slide_pred"], r["A_out"], y.cpu(), 5)
    (dl + l0 + l1).backward()
    assert torch.equal(out["masked_idx"].cpu().sort(1).values, r["masked_idx"].sort(1).values)
    assert abs(losses[3].item() - (dl + l0 + l1).item()) < 1e-4
    for (n, p) in model.named_parameters():
        g = sdg[n].grad
        assert (p.grad.cpu() - g).abs().max().item() <= 2e-3 * max(1e-4, g.abs().max().item()), n


def test_64_patch_wave_tile_of_the_256_family_matches_the_default_kernel(tmp_path):
    """ACMIL_GA3=1 (A/B build): D_inner = 256 on the one-wave-per-SIMD kernel with TWO 32-patch blocks per wave.  Per-patch scores
    are bit-identical to the default kernel (same products, same accumulation order per patch); pooled outputs agree to rounding
    (256- instead of 128-patch tiles = another summation order); both within 1e-4 of the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from acmil_amd import ops; from oracle import ga_oracle as O; from acmil_amd import synthetic as S\n"
        "res = []\n"
        "for k, c in ((5, 2), (1, 7)):\n"
        "    sd = {n: v.cuda() for n, v in S.ga_state_dict(512, 256, c, k, seed=k).items()}\n"
        "    packed, dims = ops.ga_pack_weights(sd['dimreduction.fc1.weight'], sd['attention.attention_V.0.weight'], sd['attention.attention_V.0.bias'],\n"
        "        sd['attention.attention_U.0.weight'], sd['attention.attention_U.0.bias'], sd['attention.attention_weights.weight'],\n"
        "        sd['attention.attention_weights.bias'], [sd['classifier.%%d.fc.weight' %% i] for i in range(k)],\n"
        "        [sd['classifier.%%d.fc.bias' %% i] for i in range(k)], sd['Slide_classifier.fc.weight'], sd['Slide_classifier.fc.bias'], 'f16x3')\n"
        "    xs = [O.synthetic_bag(n, 512, 300 + i)[0].cuda() for i, n in enumerate([1, 255, 257, 3000, 70000])]\n"
        "    xs += [xs[3].half(), xs[3].bfloat16()]\n"
        "    outs = [ops.ga_forward(x, packed, dims, 'f16x3') for x in xs]\n"
        "    b = ops.ga_forward_batch(xs[:5], packed, dims, 'f16x3')\n"
        "    res.append([(o['A_out'].cpu(), o['sub_preds'].cpu(), o['slide_pred'].cpu()) for o in outs]\n"
        "               + [(b['A_out'][i].cpu(), b['sub_preds'][i].cpu(), b['slide_pred'][i].cpu()) for i in range(5)])\n"
        "torch.save(res, sys.argv[1])\n" % root)
    got = {}
    for tag, env in (("v2", {}), ("v3", {"ACMIL_GA3": "1"})):
        path = str(tmp_path / (tag + ".pt"))
        r = subprocess.run([sys.executable, "-c", code, path], env=ab_environ(**env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        got[tag] = torch.load(path)
    for fam2, fam3 in zip(got["v2"], got["v3"]):
        for (a2, s2, b2), (a3, s3, b3) in zip(fam2, fam3):
            assert torch.equal(a2, a3)
            assert (s2 - s3).abs().max().item() < 2e-6 and (b2 - b3).abs().max().item() < 2e-6
        # v3: batched launch == single launches, bit for bit
        for i in range(5):
            assert all(torch.equal(p, q) for p, q in zip(fam3[i], fam3[7 + i]))


def test_wide_abmil_and_batched_evaluate():
    """ABMIL (one branch, no bag head) at the UNI width runs the fused kernel too; train.evaluate drives the wide module through
    forward_batch with the lagged range check: per-slide probabilities equal the one-slide-per-call loop, also with one bag outside the
    f16 range (its batch is repeated op by op in fp32 arithmetic)."""
    from acmil_amd import synthetic as S
    from acmil_amd import train as T
    from acmil_amd.architecture.transformer import ABMIL
    from oracle import ga_oracle as O

    class Conf:
        D_feat, D_inner, n_class, n_token = 1024, 512, 3, 1

    sd = S.ga_state_dict(1024, 512, 3, 1, seed=5, abmil=True)
    ab = ABMIL(Conf)
    ab.load_state_dict(sd)
    ab = ab.cuda().eval()
    assert ab._is_wide_fused()
    x = O.synthetic_bag(2500, 1024, slide_idx=12)
    with torch.no_grad():
        logits = ab(x.cuda())
    assert (logits.cpu() - O.abmil_forward(x, sd)).abs().max().item() < TOL

    conf = T.Struct(train_epoch=1, warmup_epoch=0, wd=1e-2, lr=1e-3, min_lr=0, n_class=3, n_token=5, n_masked_patch=0, mask_drop=0.0,
                    arch="ga", precision="f16x3", seed=1, D_feat=1024, D_inner=512)
    dev = torch.device("cuda", 0)
    T.set_seed(2)
    model = T.build_model(conf).to(dev)
    g = torch.Generator().manual_seed(4)
    bags = [(torch.randn(200 + 37 * i, 1024, generator=g).half(), i % 3) for i in range(T.EVAL_BATCH + 5)]
    bad = bags[7][0].float().clone(); bad[3, 5] = 2.0e5
    bags[7] = (bad, bags[7][1])

    class _Bags:
        def __len__(self): return len(bags)
        def __getitem__(self, i): return {"input": bags[i][0], "label": bags[i][1]}

    d_b, d_s = {}, {}
    model.range_fallbacks = 0
    res_b = T.evaluate(model, _Bags(), dev, conf, "Val", batched=True, detail=d_b)
    assert model.range_fallbacks == 1
    res_s = T.evaluate(model, _Bags(), dev, conf, "Val", batched=False, detail=d_s)
    assert torch.isfinite(d_b["prob"]).all()
    assert (d_b["prob"] - d_s["prob"]).abs().max().item() < 1e-5 and (d_b["loss"] - d_s["loss"]).abs().max().item() < 1e-5
    for a, b in zip(res_b, res_s):
        assert abs(a - b) < 1e-5


def _wide_cases(seed, count):
    import random
    rng = random.Random(seed)
    edge = [1, 2, 31, 32, 33, 127, 128, 129, 255, 256, 257, 383, 385, 511, 513, 1025]
    out = []
    for _ in range(count):
        n = rng.choice(edge) if rng.random() < 0.4 else rng.randint(1, 7000)
        d, di = rng.choice([(768, 384), (1024, 512), (512, 384), (1536, 512), (384, 512)])
        out.append((n, d, di, rng.randint(1, 5), rng.randint(2, 7), rng.choice(["float32", "float32", "float16", "bfloat16"])))
    return out


@pytest.mark.parametrize("case", _wide_cases(77, 40), ids=lambda c: "n%d_d%d_di%d_k%d_c%d_%s" % c)
def test_wide_fused_forward_random_shapes(case):
    """Seeded sweep of ga_fwd3_kernel<12 | 16, 1>: ragged N incl. tile edges, any D (also D < D_inner), n_token 1..5, classes, bag dtypes."""
    from acmil_amd import ops, synthetic as S
    from oracle import ga_oracle as O
    n, d, di, k, c, xdt = case
    sd = S.ga_state_dict(d, di, c, k, seed=n + d + k)
    x = S.synthetic_bag(n, d, slide_idx=n)[0].to(getattr(torch, xdt))
    ref = O.acmil_ga_forward(x.float().unsqueeze(0), sd, n_token=k)
    dev = {kk: v.cuda() for kk, v in sd.items()}
    packed, dims = ops.ga_pack_weights(
        dev["dimreduction.fc1.weight"], dev["attention.attention_V.0.weight"], dev["attention.attention_V.0.bias"],
        dev["attention.attention_U.0.weight"], dev["attention.attention_U.0.bias"], dev["attention.attention_weights.weight"],
        dev["attention.attention_weights.bias"], [dev["classifier.%d.fc.weight" % i] for i in range(k)],
        [dev["classifier.%d.fc.bias" % i] for i in range(k)], dev["Slide_classifier.fc.weight"], dev["Slide_classifier.fc.bias"], "f16x3")
    out = ops.ga_forward(x.cuda(), packed, dims, "f16x3", want_afeat=True, want_bag_feat=True)
    assert int(out["range_status"]) == 0
    assert (out["A_out"].cpu() - ref["A_out"][0]).abs().max() < TOL
    assert (out["sub_preds"].cpu() - ref["sub_preds"]).abs().max() < TOL
    assert (out["slide_pred"].cpu() - ref["slide_pred"][0]).abs().max() < TOL
    assert (out["afeat"].cpu() - ref["afeat"]).abs().max() < TOL
    assert (out["bag_feat"].cpu() - ref["bag_feat"][0]).abs().max() < TOL
    kk = min(10, n)
    assert torch.equal(torch.topk(out["A_out"].cpu(), kk, dim=-1).indices, torch.topk(ref["A_out"][0], kk, dim=-1).indices)
    # the score pass of a training step on the same bag: same scores bit for bit, h = relu(x W1^T)
    A, h = ops.ga_scores(x.cuda(), packed, dims, "f16x3")
    assert torch.equal(A, out["A_out"])
    assert (h.cpu() - ref["h"]).abs().max() < 2e-5 * max(1.0, ref["h"].abs().max().item())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wide_batched_launch_random_ragged_bags(seed):
    import random
    from acmil_amd import ops, synthetic as S
    from oracle import ga_oracle as O
    rng = random.Random(seed)
    d, di, k, c = rng.choice([(1024, 512, 5, 2), (768, 384, 3, 4), (768, 512, 1, 7)])
    sd = S.ga_state_dict(d, di, c, k, seed=seed)
    dev = {kk: v.cuda() for kk, v in sd.items()}
    packed, dims = ops.ga_pack_weights(
        dev["dimreduction.fc1.weight"], dev["attention.attention_V.0.weight"], dev["attention.attention_V.0.bias"],
        dev["attention.attention_U.0.weight"], dev["attention.attention_U.0.bias"], dev["attention.attention_weights.weight"],
        dev["attention.attention_weights.bias"], [dev["classifier.%d.fc.weight" % i] for i in range(k)],
        [dev["classifier.%d.fc.bias" % i] for i in range(k)], dev["Slide_classifier.fc.weight"], dev["Slide_classifier.fc.bias"], "f16x3")
    ns = [rng.choice([1, 33, 128, 129, 700, 2500, 4097]) if rng.random() < 0.5 else rng.randint(1, 5000) for _ in range(rng.randint(2, 24))]
    bags = [S.synthetic_bag(n, d, slide_idx=100 * seed + i)[0] for i, n in enumerate(ns)]
    out = ops.ga_forward_batch([b.cuda() for b in bags], packed, dims, "f16x3")
    for i, b in enumerate(bags):
        ref = O.acmil_ga_forward(b.unsqueeze(0), sd, n_token=k)
        assert (out["A_out"][i].cpu() - ref["A_out"][0]).abs().max() < TOL, (i, ns[i])
        assert (out["sub_preds"][i].cpu() - ref["sub_preds"]).abs().max() < TOL, (i, ns[i])
        assert (out["slide_pred"][i].cpu() - ref["slide_pred"][0]).abs().max() < TOL, (i, ns[i])
        single = ops.ga_forward(b.cuda(), packed, dims, "f16x3")
        assert torch.equal(single["A_out"], out["A_out"][i]) and torch.equal(single["sub_preds"], out["sub_preds"][i])


@pytest.mark.parametrize("d,di,k", [(768, 384, 5), (1024, 512, 1), (1024, 512, 3)])
@pytest.mark.parametrize("xdt", ["float32", "bfloat16"])
def test_wide_device_side_guard_repeats_a_flagged_bag_in_fp32(d, di, k, xdt):
    """acmil_ga_forward_guarded_wide: no host read-back -- the op-by-op exact-fp32 repeat is enqueued behind the fused launch with every
    kernel predicated on its status word.  An in-range bag: the fused result, counter untouched; a bag with a value outside the f16
    range (fp32 or bf16 storage): the fp32 result (finite, equal to the oracle to fp32 round-off on 1e5-sized operands), counted once;
    forward_feature (no scores requested: the repeat keeps them in its scratch) likewise; the next in-range bag is unaffected."""
    from oracle import ga_oracle as O
    model, sd = _model(d, di, k, 2, seed=k + 1)
    good = O.synthetic_bag(1500, d, slide_idx=3).to(getattr(torch, xdt))
    bad = good.clone()
    bad[0, 700, 11] = 1.0e5
    with torch.no_grad():
        s0, b0, a0 = model(good.cuda())
        assert model.range_fallbacks == 0
        s1, b1, a1 = model(bad.cuda())
        assert model.range_fallbacks == 1
        f1 = model.forward_feature(bad.cuda())
        assert model.range_fallbacks == 2
        s2, b2, a2 = model(good.cuda())
        assert model.range_fallbacks == 2
    assert torch.equal(s0, s2) and torch.equal(a0, a2) and torch.equal(b0, b2)
    ref = O.acmil_ga_forward(bad.float(), sd, n_token=k)
    assert torch.isfinite(a1).all() and torch.isfinite(s1).all()
    scale = ref["A_out"].abs().max().item()
    assert (a1.cpu() - ref["A_out"]).abs().max().item() < 1e-5 * scale + TOL
    assert (s1.cpu() - ref["sub_preds"]).abs().max().item() < 1e-3
    assert (b1.cpu() - ref["slide_pred"]).abs().max().item() < 1e-3
    assert (f1.cpu() - ref["bag_feat"]).abs().max().item() < 1e-3 * max(1.0, ref["bag_feat"].abs().max().item())
    refg = O.acmil_ga_forward(good.float(), sd, n_token=k)
    assert (a0.cpu() - refg["A_out"]).abs().max().item() < TOL


@pytest.mark.parametrize("d,di,k", [(1536, 768, 5), (512, 256, 8), (1024, 512, 10)])
@pytest.mark.parametrize("xdt", ["float32", "bfloat16"])
def test_composed_path_device_side_guard(d, di, k, xdt):
    """The composed eval path (GigaPath width; n_token > 5 at any width) resolves its range guard on the device too
    (acmil_ga_rescore_fp32_cond: h and A are overwritten with their exact-fp32 values iff the projection's status word is set, every
    launch predicated): an in-range bag is untouched (same bits as with the guard switched off), a flagged one gets the fp32 result
    and is counted once; no host synchronisation in either case."""
    from oracle import ga_oracle as O
    model, sd = _model(d, di, k, 2, seed=k)
    assert not model._is_fused() and not model._is_wide_fused()
    good = O.synthetic_bag(1300, d, slide_idx=8).to(getattr(torch, xdt))
    bad = good.clone()
    bad[0, 77, 3] = 1.0e5
    with torch.no_grad():
        s0, b0, a0 = model(good.cuda())
        assert model.range_fallbacks == 0
        s1, b1, a1 = model(bad.cuda())
        assert model.range_fallbacks == 1
        s2, b2, a2 = model(good.cuda())
        model.range_guard = False
        s3, b3, a3 = model(good.cuda())
        model.range_guard = True
    assert torch.equal(s0, s2) and torch.equal(a0, a2) and torch.equal(s0, s3) and torch.equal(a0, a3)
    ref = O.acmil_ga_forward(bad.float(), sd, n_token=k)
    assert torch.isfinite(a1).all() and torch.isfinite(s1).all()
    assert (a1.cpu() - ref["A_out"]).abs().max().item() < 1e-5 * ref["A_out"].abs().max().item() + TOL
    assert (s1.cpu() - ref["sub_preds"]).abs().max().item() < 1e-3 and (b1.cpu() - ref["slide_pred"]).abs().max().item() < 1e-3
    refg = O.acmil_ga_forward(good.float(), sd, n_token=k)
    assert (a0.cpu() - refg["A_out"]).abs().max().item() < TOL and (s0.cpu() - refg["sub_preds"]).abs().max().item() < TOL


@pytest.mark.parametrize("d,di", [(768, 384), (1024, 512)])
def test_wide_fused_skips_lo_products_of_fp16_valued_rows_only(d, di):
    """fp32 bags of f16-exact values (the reference's loader: fp16 features up-cast, Step3_WSI_classification_ACMIL.py:193): the W_hi x_lo
    MFMAs are branched over per wave and K step inside ga_fwd3_kernel.  A bag with a few genuine fp32 rows keeps them for those waves:
    every patch's scores equal the fp16-valued launch's bit for bit where the row is unchanged (the skipped products are zeros), and the
    perturbed rows match the oracle; also through the training score pass (h saved)."""
    from oracle import ga_oracle as O
    from acmil_amd import ops
    model, sd = _model(d, di, 5, 2, seed=9)
    n = 5000
    x16 = O.synthetic_bag(n, d, slide_idx=123)[0].half()
    xe = x16.float()
    g = torch.Generator().manual_seed(4)
    rows = torch.randint(0, n, (40,), generator=g)
    xm = xe.clone()
    xm[rows] += torch.randn(40, d, generator=g) * 1e-4
    with torch.no_grad():
        s_e, l_e, a_e = model(xe.cuda().unsqueeze(0))
        s_h, l_h, a_h = model(x16.cuda().unsqueeze(0))
        s_m, l_m, a_m = model(xm.cuda().unsqueeze(0))
    assert torch.equal(a_e, a_h) and torch.equal(s_e, s_h) and torch.equal(l_e, l_h)
    keep = torch.ones(n, dtype=torch.bool); keep[rows] = False
    assert torch.equal(a_m[0].cpu()[:, keep], a_e[0].cpu()[:, keep])
    ref = O.acmil_ga_forward(xm.unsqueeze(0), sd, n_token=5)
    assert (a_m.cpu() - ref["A_out"]).abs().max().item() < TOL
    assert (s_m.cpu() - ref["sub_preds"]).abs().max().item() < TOL
    packed, dims = model._packed()
    A1, h1 = ops.ga_scores(xe.cuda(), packed, dims, "f16x3")
    A2, h2 = ops.ga_scores(x16.cuda(), packed, dims, "f16x3")
    assert torch.equal(A1, A2) and torch.equal(h1, h2)
