"""Seeded random-shape sweep of the fused GA forward (eval, batched, training score/pool path) against the oracle:
ragged N (1 .. 6000, incl. tile edges), D / D_inner / n_token / n_class combinations, bag dtypes, both parity modes."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cases(seed, count):
    rng = random.Random(seed)
    edge = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1025]
    out = []
    for _ in range(count):
        n = rng.choice(edge) if rng.random() < 0.4 else rng.randint(1, 6000)
        out.append((n, rng.choice([384, 512, 768, 1024]), rng.choice([128, 256]), rng.randint(1, 5), rng.randint(2, 7),
                    rng.choice(["float32", "float32", "float16", "bfloat16"]), rng.choice(["f16x3", "f16x3", "fp32"])))
    return out


@pytest.mark.parametrize("case", _cases(1234, 60), ids=lambda c: "n%d_d%d_di%d_k%d_c%d_%s_%s" % c)
def test_eval_forward_random_shapes(case):
    from acmil_amd import ops, synthetic as S
    from oracle import ga_oracle as O
    n, d, di, k, c, xdt, prec = case
    sd = S.ga_state_dict(d, di, c, k, seed=n + d + k)
    x = S.synthetic_bag(n, d, slide_idx=n)[0].to(getattr(torch, xdt))
    ref = O.acmil_ga_forward(x.float().unsqueeze(0), sd, n_token=k)
    dev = {kk: v.cuda() for kk, v in sd.items()}
    packed, dims = ops.ga_pack_weights(
        dev["dimreduction.fc1.weight"], dev["attention.attention_V.0.weight"], dev["attention.attention_V.0.bias"],
        dev["attention.attention_U.0.weight"], dev["attention.attention_U.0.bias"], dev["attention.attention_weights.weight"],
        dev["attention.attention_weights.bias"], [dev["classifier.%d.fc.weight" % i] for i in range(k)],
        [dev["classifier.%d.fc.bias" % i] for i in range(k)], dev["Slide_classifier.fc.weight"], dev["Slide_classifier.fc.bias"], prec)
    out = ops.ga_forward(x.cuda(), packed, dims, prec, want_afeat=True, want_bag_feat=True)
    assert (out["A_out"].cpu() - ref["A_out"][0]).abs().max() < 1e-4
    assert (out["sub_preds"].cpu() - ref["sub_preds"]).abs().max() < 1e-4
    assert (out["slide_pred"].cpu() - ref["slide_pred"][0]).abs().max() < 1e-4
    assert (out["afeat"].cpu() - ref["afeat"]).abs().max() < 1e-4
    kk = min(10, n)
    assert torch.equal(torch.topk(out["A_out"].cpu(), kk, dim=-1).indices, torch.topk(ref["A_out"][0], kk, dim=-1).indices)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_batched_launch_random_ragged_bags(seed):
    from acmil_amd import ops, synthetic as S
    from oracle import ga_oracle as O
    rng = random.Random(seed)
    d, di, k, c = rng.choice([(512, 256, 5, 2), (384, 128, 3, 4), (768, 256, 1, 7)])
    sd = S.ga_state_dict(d, di, c, k, seed=seed)
    dev = {kk: v.cuda() for kk, v in sd.items()}
    packed, dims = ops.ga_pack_weights(
        dev["dimreduction.fc1.weight"], dev["attention.attention_V.0.weight"], dev["attention.attention_V.0.bias"],
        dev["attention.attention_U.0.weight"], dev["attention.attention_U.0.bias"], dev["attention.attention_weights.weight"],
        dev["attention.attention_weights.bias"], [dev["classifier.%d.fc.weight" % i] for i in range(k)],
        [dev["classifier.%d.fc.bias" % i] for i in range(k)], dev["Slide_classifier.fc.weight"], dev["Slide_classifier.fc.bias"], "f16x3")
    ns = [rng.choice([1, 33, 128, 129, 700, 2500, 4097]) if rng.random() < 0.5 else rng.randint(1, 5000) for _ in range(rng.randint(2, 16))]
    bags = [S.synthetic_bag(n, d, slide_idx=100 * seed + i)[0] for i, n in enumerate(ns)]
    out = ops.ga_forward_batch([b.cuda() for b in bags], packed, dims, "f16x3")
    for i, b in enumerate(bags):
        ref = O.acmil_ga_forward(b.unsqueeze(0), sd, n_token=k)
        assert (out["A_out"][i].cpu() - ref["A_out"][0]).abs().max() < 1e-4, (i, ns[i])
        assert (out["sub_preds"][i].cpu() - ref["sub_preds"]).abs().max() < 1e-4, (i, ns[i])
        assert (out["slide_pred"][i].cpu() - ref["slide_pred"][0]).abs().max() < 1e-4, (i, ns[i])


def _train_cases(seed, count):
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        n = rng.choice([7, 33, 128, 129, 300, 1000, 2049]) if rng.random() < 0.5 else rng.randint(12, 3000)
        out.append((n, rng.choice([384, 512]), rng.choice([128, 256]), rng.choice([1, 3, 5]), rng.randint(2, 7), rng.choice([0, 10]),
                    rng.choice(["f16x3", "fp32"])))
    return out


@pytest.mark.parametrize("case", _train_cases(99, 14), ids=lambda c: "n%d_d%d_di%d_k%d_c%d_mask%d_%s" % c)
def test_fused_train_step_random_shapes_vs_oracle_autograd(case):
    """model.train_step (score pass, STKIM, masked pooling, fused loss, HIP backward) against torch autograd over the oracle
    with the same uniforms: losses and every parameter gradient."""
    from acmil_amd import synthetic as S
    from acmil_amd.architecture.transformer import ACMIL_GA
    from oracle import ga_oracle as O
    n, d, di, k, c, nm, prec = case
    sd = S.ga_state_dict(d, di, c, k, seed=n + k)
    x = S.synthetic_bag(n, d, slide_idx=n + 1)
    label = torch.tensor([n % c])
    kk = min(nm, n)
    uni = torch.rand(k, kk, generator=torch.Generator().manual_seed(n)) if kk > 0 else None
    sdr = {kq: v.clone().requires_grad_(True) for kq, v in sd.items()}
    ref = O.acmil_ga_forward(x, sdr, n_token=k, n_masked_patch=nm, mask_drop=0.6, training=True, uniforms=uni)
    l0, l1, dl = O.acmil_losses(ref["sub_preds"], ref["slide_pred"], ref["A_out"], label, k)
    (l0 + l1 + dl).backward()

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k
    model = ACMIL_GA(Conf, n_token=k, n_masked_patch=nm, mask_drop=0.6, precision=prec)
    model.load_state_dict(sd)
    model = model.cuda().train()
    losses, out = model.train_step(x.cuda(), label.cuda(), uniforms=None if uni is None else uni.cuda())
    got = losses.cpu()
    assert abs(got[0].item() - float(l0.detach())) < 5e-5 and abs(got[1].item() - float(l1.detach())) < 5e-5 and abs(got[2].item() - float(dl.detach())) < 5e-5
    if kk > 0:
        assert torch.equal(out["topk_idx"].cpu(), ref["topk_idx"]) and torch.equal(out["masked_idx"].cpu(), ref["masked_idx"])
    for name_p, p in model.named_parameters():
        r = sdr[name_p].grad
        r = torch.zeros_like(p.grad.cpu()) if r is None else r
        err = (p.grad.cpu() - r).abs().max().item()
        assert err <= 5e-4 * r.abs().max().item() + 2e-7, (name_p, err, r.abs().max().item())
