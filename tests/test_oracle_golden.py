"""Pin the CPU oracle (oracle/ga_oracle.py) against outputs of the reference itself.

The fixtures were produced by tests/golden/make_golden.py running /root/reference; the reference
holds no tests of its own for this path.  Forward: bit-exact in fp32 (the oracle
issues the same ATen ops in the same order).  Train step: losses and every parameter gradient."""
import numpy as np
import pytest
import torch

from conftest import EVAL_CASES, TRAIN_CASES, case_dims, load_golden, trajectory_bags, trajectory_stable
from oracle import ga_oracle as O


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)  # the fixtures were generated single-threaded (reduction order)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("name", EVAL_CASES)
def test_eval_forward_bit_exact(name):
    case, sd = load_golden(name)
    d, di, k, c = case_dims(sd)
    x = torch.from_numpy(case["x"]).float()
    out = O.acmil_ga_forward(x, sd, n_token=k)
    assert np.array_equal(out["A_out"].numpy(), case["A_out"])
    assert np.array_equal(out["sub_preds"].numpy(), case["sub_preds"])
    assert np.array_equal(out["slide_pred"].numpy(), case["slide_pred"])
    feat = O.acmil_ga_forward_feature(x, sd)
    assert np.array_equal(feat.numpy(), case["bag_feat"])
    assert out["A_out"].shape == (1, k, x.shape[1]) and out["sub_preds"].shape == (k, c)


@pytest.mark.parametrize("tag", ["d512_k8_c2", "d384_k10_c7", "d1024_k16_c2"])
def test_eval_forward_bit_exact_other_branch_counts(tag):
    """The oracle at n_token = 8 / 10 / 16 (the reference takes any: Step3_WSI_classification_ACMIL.py:39) against fixtures of the
    real reference (tests/golden/make_golden_ntoken.py)."""
    case, sd = load_golden("ga_eval_n300_" + tag)
    d, di, k, c = case_dims(sd)
    x = torch.from_numpy(case["x"]).float()
    out = O.acmil_ga_forward(x, sd, n_token=k)
    assert np.array_equal(out["A_out"].numpy(), case["A_out"]) and np.array_equal(out["sub_preds"].numpy(), case["sub_preds"])
    assert np.array_equal(out["slide_pred"].numpy(), case["slide_pred"])
    assert out["A_out"].shape == (1, k, 300) and out["sub_preds"].shape == (k, c)


def test_abmil_bit_exact():
    case, sd = load_golden("abmil_eval_n1000_d512_c2")
    logits = O.abmil_forward(torch.from_numpy(case["x"]).float(), sd)
    assert np.array_equal(logits.numpy(), case["logits"])


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_forward_and_step(name):
    case, sd = load_golden(name)
    d, di, k, c = case_dims(sd)
    x = torch.from_numpy(case["x"]).float()
    n = x.shape[1]
    sdg = {kk: v.clone().requires_grad_(True) for kk, v in sd.items()}
    out = O.acmil_ga_forward(x, sdg, n_token=k, n_masked_patch=10, mask_drop=0.6, training=True,
                             uniforms=torch.from_numpy(case["uniforms"]))
    kk = min(10, n)
    assert out["topk_idx"].shape == (k, kk)
    assert np.array_equal(out["topk_idx"].numpy(), case["topk_idx"])
    assert np.array_equal(np.sort(out["masked_idx"].numpy(), axis=1), case["masked_idx"])
    assert case["masked_idx"].shape[1] == int(kk * 0.6)
    assert np.array_equal(out["A_out"].detach().numpy(), case["A_out"])
    assert np.array_equal(out["sub_preds"].detach().numpy(), case["sub_preds"])
    assert np.array_equal(out["slide_pred"].detach().numpy(), case["slide_pred"])
    label = torch.from_numpy(case["label"])
    loss0, loss1, diff = O.acmil_losses(out["sub_preds"], out["slide_pred"], out["A_out"], label, k)
    assert float(loss0.detach()) == pytest.approx(float(case["loss0"]), abs=1e-6)
    assert float(loss1.detach()) == pytest.approx(float(case["loss1"]), abs=1e-6)
    (diff + loss0 + loss1).backward()
    for key, p in sdg.items():
        g_ref = case["grad." + key]
        np.testing.assert_allclose(p.grad.numpy(), g_ref, rtol=1e-4, atol=1e-7, err_msg=key)
    # the reference's AdamW step (lr as set by its cosine schedule at epoch 0)
    lr = O.adjust_learning_rate(0.0, 1e-4, 0.0, 0.0, 50.0)
    assert lr == pytest.approx(float(case["lr"]))
    params = [v.detach().clone().requires_grad_(True) for v in sd.values()]
    for p, kname in zip(params, sd.keys()):
        p.grad = torch.from_numpy(case["grad." + kname]).clone()
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=float(case["wd"]))
    opt.step()
    for p, kname in zip(params, sd.keys()):
        np.testing.assert_allclose(p.detach().numpy(), case["after." + kname], rtol=0, atol=1e-7, err_msg=kname)


def test_lr_schedule_values():
    # utils/utils.py:250-262 : warm-up then half cosine
    assert O.adjust_learning_rate(0.5, 1e-3, 1e-5, 1.0, 10.0) == pytest.approx(5e-4)
    assert O.adjust_learning_rate(1.0, 1e-3, 1e-5, 1.0, 10.0) == pytest.approx(1e-3)
    assert O.adjust_learning_rate(10.0, 1e-3, 1e-5, 1.0, 10.0) == pytest.approx(1e-5)
    assert O.adjust_learning_rate(5.5, 1e-3, 0.0, 1.0, 10.0) == pytest.approx(5e-4)


def test_fp64_ground_truth_close_to_fp32():
    case, sd = load_golden("ga_eval_n257_d512_k5_c2")
    x = torch.from_numpy(case["x"])
    o32 = O.acmil_ga_forward(x, sd, n_token=5)
    o64 = O.acmil_ga_forward(x.double(), {k: v.double() for k, v in sd.items()}, n_token=5)
    assert (o32["A_out"].double() - o64["A_out"]).abs().max() < 2e-6
    assert (o32["sub_preds"].double() - o64["sub_preds"]).abs().max() < 2e-6


def test_ten_step_trajectory_of_the_reference_loop():
    """The oracle's forward / losses / schedule under torch.optim.AdamW reproduce the TEN-step trajectory the reference's own
    train_one_epoch produced (two epochs over five slides: warm-up from lr 0, cosine, moments and bias corrections evolving):
    per-step learning rates and losses, and the final parameters."""
    case, sd = load_golden("ga_trajectory_d512_k5_c7")
    bags = trajectory_bags(case)
    params = {n: v.clone().requires_grad_(True) for n, v in sd.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=0.001, weight_decay=float(case["wd"]))
    step = 0
    for epoch, order in enumerate(case["orders"].tolist()):
        for it, i in enumerate(order):
            lr = O.adjust_learning_rate(epoch + it / len(order), float(case["lr"]), 0.0, float(case["warmup_epoch"]), float(case["train_epoch"]))
            assert lr == pytest.approx(float(case["lrs"][step]), abs=1e-12)
            for gq in opt.param_groups:
                gq["lr"] = lr
            out = O.acmil_ga_forward(bags[i].float(), params, n_token=5, n_masked_patch=10, mask_drop=0.6, training=True,
                                     uniforms=torch.from_numpy(case["uniforms"][step]))
            l0, l1, dl = O.acmil_losses(out["sub_preds"], out["slide_pred"], out["A_out"], torch.tensor([int(case["labels"][i])]), 5)
            assert float(l0.detach()) == pytest.approx(float(case["losses"][step, 0]), abs=2e-6)
            assert float(l1.detach()) == pytest.approx(float(case["losses"][step, 1]), abs=2e-6)
            opt.zero_grad()
            (dl + l0 + l1).backward()
            opt.step()
            step += 1
    for n, p in params.items():
        ok = trajectory_stable(case, n)
        assert ok.mean() > 0.9 or n == "attention.attention_weights.bias", (n, ok.mean())
        d = np.abs(p.detach().numpy() - case["final." + n])[ok]
        assert d.max(initial=0.0) <= 1e-6, (n, d.max())
