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
