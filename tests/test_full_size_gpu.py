"""BASELINE.json configs at their FULL sizes in the driver-run GPU suite (parity against the oracle on the same values):
cfg2 (N=10 000 masked training step), cfg3 (N=50 000, D=384, D_inner=128, bf16 bag), cfg4 (TransMIL N=100 000, D=768).
The oracle runs once per test on the host (seconds)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ga(sd, k, c, d, di, precision, **kw):
    from acmil_amd.architecture.transformer import ACMIL_GA

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    m = ACMIL_GA(Conf, n_token=k, precision=precision, **kw)
    m.load_state_dict(sd)
    return m.cuda()


def _oracle_train_step(sd, x, u, label, k, dtype):
    from oracle import ga_oracle as O
    sdg = {n: v.clone().to(dtype).requires_grad_(True) for n, v in sd.items()}
    ref = O.acmil_ga_forward(x.to(dtype).unsqueeze(0), sdg, n_token=k, n_masked_patch=10, mask_drop=0.6, uniforms=u.to(dtype), training=True)
    l0, l1, dl = O.acmil_losses(ref["sub_preds"], ref["slide_pred"], ref["A_out"], label, k)
    (l0 + l1 + dl).backward()
    return ref, (float(l0.detach()), float(l1.detach()), float(dl.detach())), {n: v.grad.double() for n, v in sdg.items()}


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_cfg2_masked_train_step_n10000_matches_oracle_autograd(precision):
    """configs[1]: ACMIL n_token=5, n_masked_patch=10, mask_drop=0.6, N=10 000, D=512: top-k / masked indices bit-exact, outputs
    and losses within 1e-4, parameter gradients against the oracle's torch-CPU autograd.
    At this size fp32 autograd itself is ~2e-3 (relative to the largest entry) away from the same computation in fp64 for
    dW1 -- the ReLU mask of pre-activations near zero depends on the last bit of h -- so the bound is two-sided: exact-fp32
    mode reproduces the fp32 oracle to 2e-4, and every mode is at least as close to the fp64 oracle as the fp32 oracle is."""
    from oracle import ga_oracle as O
    N, D, Di, K, C = 10000, 512, 256, 5, 2
    sd = O.default_state_dict(D, Di, C, K)
    x = O.synthetic_bag(N, D, 21)[0]
    u = torch.rand(K, 10, generator=torch.Generator().manual_seed(5))
    label = torch.tensor([1])
    ref, loss32, g32 = _oracle_train_step(sd, x, u, label, K, torch.float32)
    _, _, g64 = _oracle_train_step(sd, x, u, label, K, torch.float64)
    model = _ga(sd, K, C, D, Di, precision, n_masked_patch=10, mask_drop=0.6).train()
    losses, out = model.train_step(x.cuda().unsqueeze(0), label.cuda(), uniforms=u.cuda())
    assert np.array_equal(out["topk_idx"].cpu().numpy(), ref["topk_idx"].numpy())
    assert np.array_equal(np.sort(out["masked_idx"].cpu().numpy(), 1), np.sort(ref["masked_idx"].numpy(), 1))
    a_ref = ref["A_out"].detach().reshape(K, N)
    assert (out["A_out"].cpu() - a_ref).abs().max().item() < TOL
    assert int((out["A_out"] == -1e9).sum()) == K * 6
    assert (out["sub_preds"].cpu() - ref["sub_preds"].detach()).abs().max().item() < TOL
    for got, want in zip(losses[:3].tolist(), loss32):
        assert got == pytest.approx(want, abs=2e-5)
    for name, p in model.named_parameters():
        scale = g64[name].abs().max().item()
        if scale < 1e-9:          # attention_weights.bias: analytically zero under the softmax
            continue
        g = p.grad.cpu().double()
        e64 = (g - g64[name]).abs().max().item() / scale
        e32 = (g - g32[name]).abs().max().item() / scale
        ref_e = (g32[name] - g64[name]).abs().max().item() / scale      # the reference arithmetic's own distance from fp64
        assert e64 <= max(2e-4, 1.5 * ref_e), (name, e64, ref_e)
        if precision == "fp32":
            assert e32 <= 2e-4, (name, e32)


def test_train_n50k_bench_shape_matches_oracle_autograd():
    """The `train_n50k` bench shape itself -- N = 50 000, D = 512, C = 7 (BRACS, configs[4]), fp16 bag, masked step on the 64-row
    backward tiles (`ga_bwd_tile_kernel<5, 256, 64>`) and the 128 x 256 weight-gradient tiles: top-k / masked indices bit-exact,
    losses within 2e-5, every parameter gradient against the oracle's torch-CPU autograd (two-sided bound of the cfg2 test).
    Patches with a pre-activation within 1e-5 of zero are replaced by fresh draws (d relu / d pre is discontinuous there: the ReLU
    mask of such an element is not a function of the data at fp32 precision, see tests/test_train_group_gpu.py::_bags)."""
    from oracle import ga_oracle as O
    N, D, Di, K, C = 50000, 512, 256, 5, 7
    sd = O.default_state_dict(D, Di, C, K)
    x = O.synthetic_bag(N, D, 0)[0].half()
    w1 = sd["dimreduction.fc1.weight"].double()
    for attempt in range(20):
        bad = ((x.double() @ w1.T).abs() < 1e-5).any(dim=1)
        if not bool(bad.any()):
            break
        x[bad] = torch.randn(int(bad.sum()), D, generator=torch.Generator().manual_seed(900 + attempt)).half()
    u = torch.rand(K, 10, generator=torch.Generator().manual_seed(6))
    label = torch.tensor([4])
    ref, loss32, g32 = _oracle_train_step(sd, x.float(), u, label, K, torch.float32)
    _, _, g64 = _oracle_train_step(sd, x.float(), u, label, K, torch.float64)
    model = _ga(sd, K, C, D, Di, "f16x3", n_masked_patch=10, mask_drop=0.6).train()
    losses, out = model.train_step(x.cuda().unsqueeze(0), label.cuda(), uniforms=u.cuda())
    assert np.array_equal(out["topk_idx"].cpu().numpy(), ref["topk_idx"].numpy())
    assert np.array_equal(np.sort(out["masked_idx"].cpu().numpy(), 1), np.sort(ref["masked_idx"].numpy(), 1))
    assert (out["A_out"].cpu() - ref["A_out"].detach().reshape(K, N)).abs().max().item() < TOL
    assert (out["sub_preds"].cpu() - ref["sub_preds"].detach()).abs().max().item() < TOL
    for got, want in zip(losses[:3].tolist(), loss32):
        assert got == pytest.approx(want, abs=2e-5)
    for name, p in model.named_parameters():
        scale = g64[name].abs().max().item()
        if scale < 1e-9:
            continue
        g = p.grad.cpu().double()
        e64 = (g - g64[name]).abs().max().item() / scale
        ref_e = (g32[name] - g64[name]).abs().max().item() / scale
        assert e64 <= max(2e-4, 1.5 * ref_e), (name, e64, ref_e)
    # the same slide as one member of a GROUP of two (the group kernels at this size): its losses are unchanged
    x2 = O.synthetic_bag(30000, D, 1)[0].half()
    l2, o2 = model.train_step_batch([x.cuda(), x2.cuda()], torch.tensor([4, 1]).cuda(),
                                    uniforms=torch.stack([u, torch.rand(K, 10, generator=torch.Generator().manual_seed(7))]).cuda())
    assert (l2[0] - losses).abs().max().item() < 1e-6
    assert torch.equal(o2["topk_idx"][0], out["topk_idx"])


def test_cfg3_camelyon_shape_bf16_bag_n50000_matches_oracle():
    """configs[2]: ACMIL n_token=5, N=50 000, D=384 (SSL ViT-S), D_inner=128, bf16 bag: scores / logits within 1e-4 of the oracle on
    the same (bf16-rounded) values, top-10 per branch identical, batched launch equal to the single one."""
    from oracle import ga_oracle as O
    N, D, Di, K, C = 50000, 384, 128, 5, 2
    sd = O.default_state_dict(D, Di, C, K)
    xb = O.synthetic_bag(N, D, 33)[0].to(torch.bfloat16)
    ref = O.acmil_ga_forward(xb.float().unsqueeze(0), sd, n_token=K)
    model = _ga(sd, K, C, D, Di, "f16x3").eval()
    with torch.no_grad():
        sub, slide, a = model(xb.cuda().unsqueeze(0))
        outs = model.forward_batch([xb.cuda(), xb.cuda()[:777]])
    a_ref = ref["A_out"].reshape(K, N)
    assert (a[0].cpu() - a_ref).abs().max().item() < TOL
    assert (sub.cpu() - ref["sub_preds"]).abs().max().item() < TOL and (slide.cpu() - ref["slide_pred"]).abs().max().item() < TOL
    assert torch.equal(torch.topk(a[0].cpu(), 10, dim=-1).indices, torch.topk(a_ref, 10, dim=-1).indices)
    assert torch.equal(outs[0][2], a) and torch.equal(outs[0][0], sub)
    assert int(model._last["range_status"]) == 0


def test_cfg4_transmil_n100000_d768_matches_oracle():
    """configs[3]: TransMIL / Nystrom attention, N=100 000, D=768, D_inner=384: logits within 1e-4 of the oracle (one ~5 s CPU run)."""
    from acmil_amd import synthetic as S
    from acmil_amd.architecture.transMIL import TransMIL
    from oracle import transmil_oracle as TO
    N, D, Di, C = 100000, 768, 384, 2

    class Conf:
        D_feat, D_inner, n_class = D, Di, C

    sd = S.transmil_state_dict(D, Di, C, seed=1)
    model = TransMIL(Conf)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    x = torch.randn(1, N, D, generator=torch.Generator().manual_seed(1000))
    with torch.no_grad():
        got = model(x.cuda()).cpu()
    ref = TO.transmil_forward(x, sd)["logits"]
    assert got.shape == (1, C) and torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < TOL


def _ref_fixture():
    return np.load("tests/golden/ga_fullsize_reference.npz")


@pytest.mark.parametrize("key", ["eval_n50000_d512", "eval_n50000_d384_bf16"])
def test_full_size_eval_matches_the_real_reference(key):
    """North star (N = 50 000, D = 512) and configs[2] (D = 384, D_inner = 128, bf16 bag) against outputs of the REAL reference
    module at full size (tests/golden/make_golden_fullsize.py; weights and bag regenerated from the generator's seeds):
    logits within 1e-4, the top-10 patches of every branch in the reference's order, scores within 1e-4."""
    from acmil_amd import synthetic as S
    z = _ref_fixture()
    n, d, di, k, c, slide, bf16 = [int(v) for v in z[key + ".meta"]]
    model = _ga(S.ga_state_dict(d, di, c, k), k, c, d, di, "f16x3").eval()
    x = S.synthetic_bag(n, d, slide_idx=slide)[0]
    xg = x.bfloat16().cuda() if bf16 else x.cuda()
    with torch.no_grad():
        sub, slide_pred, a = model(xg.unsqueeze(0))
    assert np.abs(sub.cpu().numpy() - z[key + ".sub_preds"]).max() < TOL and np.abs(slide_pred.cpu().numpy() - z[key + ".slide_pred"]).max() < TOL
    assert np.array_equal(torch.topk(a[0].cpu(), 10, dim=-1).indices.numpy(), z[key + ".topk"])
    assert np.abs(a[0].cpu()[:, ::997].numpy() - z[key + ".A_sample"]).max() < TOL
    ac = a[0].cpu().double()
    got = np.stack([[float(ac[i].mean()), float(ac[i].abs().mean()), float(ac[i].max()), float(ac[i].min())] for i in range(k)])
    assert np.abs(got - z[key + ".A_stats"]).max() < TOL


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_cfg2_training_step_matches_the_real_reference(precision):
    """configs[1] at full size (N = 10 000, n_masked_patch 10, mask_drop 0.6) against ONE iteration of the reference's own
    train_one_epoch: masked indices exact, logits and both cross-entropies within 1e-4, every parameter gradient by norm
    (1e-3 relative) and by a 1-in-997 sample (2e-3 of the parameter's largest gradient; fp32 autograd itself is that far from
    fp64 for dW1 at this size, see the oracle test above)."""
    from acmil_amd import synthetic as S
    z = _ref_fixture()
    key = "train_n10000_d512"
    n, d, di, k, c, slide, label, _ = [int(v) for v in z[key + ".meta"]]
    model = _ga(S.ga_state_dict(d, di, c, k), k, c, d, di, precision, n_masked_patch=10, mask_drop=0.6).train()
    x = S.synthetic_bag(n, d, slide_idx=slide)[0].cuda()
    losses, out = model.train_step(x.unsqueeze(0), torch.tensor([label]).cuda(), uniforms=torch.from_numpy(z[key + ".uniforms"]).cuda())
    assert np.array_equal(np.sort(out["masked_idx"].cpu().numpy(), axis=1), z[key + ".masked_idx"])
    assert np.abs(out["sub_preds"].cpu().numpy() - z[key + ".sub_preds"]).max() < TOL
    assert np.abs(out["slide_pred"].cpu().numpy().reshape(-1) - z[key + ".slide_pred"].reshape(-1)).max() < TOL
    l = losses.cpu().numpy()
    assert abs(l[0] - float(z[key + ".loss0"])) < TOL and abs(l[1] - float(z[key + ".loss1"])) < TOL
    for name, p in model.named_parameters():
        g = p.grad.detach().double().reshape(-1).cpu()
        norm_ref, max_ref = z[key + ".gnorm." + name]
        # (+ 2e-7 absolute: the gradient of attention_weights.bias is a sum of softmax gradients = 0 up to rounding, ~1e-8)
        assert abs(float(g.norm()) - norm_ref) <= 1e-3 * norm_ref + 2e-7, name
        assert np.abs(g[::997].float().numpy() - z[key + ".gsample." + name]).max() <= 2e-3 * max_ref + 2e-7, name
