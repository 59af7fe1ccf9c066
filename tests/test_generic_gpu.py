"""GPU parity tests of the constructor arguments the fused kernels do not cover and the aggregators serve on their op-by-op path
(ACMIL_GA / ABMIL with an attention width other than 128, classifier dropout in training mode, DimReduction residual blocks) and of the
`MHA` module -- against fixtures from the real reference (tests/golden/make_golden_generic.py)."""
import numpy as np
import pytest
import torch

from conftest import case_dims, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
FAMILIES = ["d512_a64_k5_c2", "d384_a256_k3_c7"]


def _model(sd, precision="f16x3", **kw):
    from acmil_amd.architecture.transformer import ACMIL_GA
    d, di, k, c = case_dims(sd)
    da = sd["attention.attention_V.0.weight"].shape[0]

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    m = ACMIL_GA(Conf, D=da, n_token=k, n_masked_patch=10, mask_drop=0.6, precision=precision, **kw)
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("tag", FAMILIES)
def test_other_attention_width_eval_matches_reference(tag, precision):
    """ACMIL_GA(conf, D=64 / 256) (transformer.py:240,292): scores, logits, bag feature within 1e-4, top-10 order identical."""
    case, sd = load_golden("ga_eval_n300_" + tag)
    d, di, k, c = case_dims(sd)
    model = _model(sd, precision).eval()
    assert model._generic()
    x = torch.from_numpy(case["x"]).cuda()
    with torch.no_grad():
        sub, slide, a = model(x.float())
        sub16, _, a16 = model(x)                      # 16-bit bag fed directly
        feat = model.forward_feature(x.float())
        triples = model.forward_batch([x[0], x[0].float()])
    assert a.shape == (1, k, 300) and sub.shape == (k, c) and slide.shape == (1, c) and feat.shape == (1, di)
    np.testing.assert_allclose(a.cpu().numpy(), case["A_out"], rtol=0, atol=TOL)
    np.testing.assert_allclose(sub.cpu().numpy(), case["sub_preds"], rtol=0, atol=TOL)
    np.testing.assert_allclose(slide.cpu().numpy(), case["slide_pred"], rtol=0, atol=TOL)
    np.testing.assert_allclose(feat.cpu().numpy(), case["bag_feat"], rtol=0, atol=TOL)
    assert torch.equal(a, a16) and torch.equal(sub, sub16)
    assert np.array_equal(np.argsort(-case["A_out"][0], axis=-1, kind="stable")[:, :10], torch.topk(a[0], 10, dim=-1).indices.cpu().numpy())
    assert torch.equal(triples[0][0], sub) and torch.equal(triples[1][2], a)


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("tag", FAMILIES)
def test_other_attention_width_train_step_matches_reference(tag, precision):
    """One iteration of the reference's own train_one_epoch at D = 64 / 256: STKIM indices bit-exact, losses, every parameter
    gradient -- through torch.autograd over the module's forward AND through ACMIL_GA.train_step."""
    from test_train_gpu import _losses
    case, sd = load_golden("ga_train_n200_" + tag)
    d, di, k, c = case_dims(sd)
    stride = int(case["w1_row_stride"])
    x = torch.from_numpy(case["x"]).cuda()
    label = torch.from_numpy(case["label"]).cuda()
    u = torch.from_numpy(case["uniforms"]).cuda()

    def check_grads(model):
        for name, p in model.named_parameters():
            ref = case["grad." + name]
            got = p.grad.cpu().numpy()
            if name == "dimreduction.fc1.weight":
                got = got[::stride]
            scale = max(np.abs(ref).max(), 1e-30)
            if scale < 1e-7:          # attention_weights.bias: analytically zero under the softmax, rounding noise in any implementation
                continue
            assert np.abs(got - ref).max() <= 3e-4 * scale + 1e-9, (name, np.abs(got - ref).max() / scale)

    model = _model(sd, precision).train()
    sub, slide, attn = model(x.float(), uniforms=u)
    assert np.array_equal(model._last["topk_idx"].cpu().numpy(), case["topk_idx"])
    assert np.array_equal(np.sort(model._last["masked_idx"].cpu().numpy(), axis=1), np.sort(case["masked_idx"], axis=1))
    np.testing.assert_allclose(attn.detach().cpu().numpy(), case["A_out"], rtol=0, atol=TOL)
    loss0, loss1, diff = _losses(sub, slide, attn, label, k)
    assert float(loss0.detach()) == pytest.approx(float(case["loss0"]), abs=2e-5)
    assert float(loss1.detach()) == pytest.approx(float(case["loss1"]), abs=2e-5)
    (diff + loss0 + loss1).backward()
    check_grads(model)
    m2 = _model(sd, precision).train()
    losses, out = m2.train_step(x, label, uniforms=u)
    assert losses[0].item() == pytest.approx(float(case["loss0"]), abs=2e-5) and losses[1].item() == pytest.approx(float(case["loss1"]), abs=2e-5)
    assert losses[2].item() == pytest.approx(float(diff.detach()), abs=2e-5)
    check_grads(m2)
    # a group step of such a model takes the serial route: mean of two slides' gradients
    g1 = [p.grad.clone() for p in m2.parameters()]
    l2, _ = m2.train_step_batch([x[0], x[0]], torch.cat([label, label]), uniforms=[u, u])
    assert l2.shape == (2, 4) and (l2[0] - losses).abs().max().item() < 1e-6
    for a, p in zip(g1, m2.parameters()):
        assert (a - p.grad).abs().max().item() <= 1e-6 * max(1e-6, a.abs().max().item())


def test_abmil_other_attention_width_matches_reference():
    from acmil_amd.architecture.transformer import ABMIL
    case, sd = load_golden("abmil_eval_n400_a64_d512_c2")

    class Conf:
        D_feat, D_inner, n_class, n_token = 512, 256, 2, 1

    model = ABMIL(Conf, D=64)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    with torch.no_grad():
        logits = model(torch.from_numpy(case["x"]).cuda())
    np.testing.assert_allclose(logits.cpu().numpy(), case["logits"], rtol=0, atol=TOL)


def test_dimreduction_residual_blocks_match_reference():
    """DimReduction(numLayer_Res=2) (network.py:22-34,44-56): forward and the gradients of mean-square output."""
    from acmil_amd.architecture.network import DimReduction
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "dimreduction_res2_n500_d384.npz"))
    dr = DimReduction(384, 128, numLayer_Res=2)
    dr.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")})
    dr = dr.cuda()
    out = dr(torch.from_numpy(z["x"]).cuda().float())
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=0, atol=TOL)
    (out.square().sum() / out.shape[0]).backward()
    for name, p in dr.named_parameters():
        ref = z["grad." + name]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-4 * np.abs(ref).max(), name


def test_residual_dimreduction_inside_the_aggregator_runs_the_generic_path():
    """An ACMIL_GA whose DimReduction carries residual blocks (not constructible through the reference's ACMIL_GA ctor, but a valid
    composition of its modules): the module's own op-by-op forward equals the oracle restatement of the same composition."""
    from acmil_amd.architecture.network import DimReduction
    from oracle import ga_oracle as O
    sd = O.default_state_dict(384, 128, 2, 5)
    model = _model(sd)
    torch.manual_seed(3)
    model.dimreduction = DimReduction(384, 128, numLayer_Res=1).cuda()
    model.dimreduction.fc1.weight.data.copy_(sd["dimreduction.fc1.weight"])
    assert model._generic()
    x = O.synthetic_bag(500, 384, 4)[0]
    with torch.no_grad():
        sub, slide, a = model.eval()(x.cuda().unsqueeze(0))
    h = torch.relu(x @ sd["dimreduction.fc1.weight"].T)
    blk = model.dimreduction.resBlocks[0].block
    h = h + torch.relu(torch.relu(h @ blk[0].weight.detach().cpu().T) @ blk[2].weight.detach().cpu().T)
    A = O.attention_gated(h, sd)
    ref_af = torch.softmax(A, 1) @ h
    ref_sub = torch.stack([ref_af[i] @ sd["classifier.%d.fc.weight" % i].T + sd["classifier.%d.fc.bias" % i] for i in range(5)])
    assert (a[0].cpu() - A).abs().max().item() < TOL and (sub.cpu() - ref_sub).abs().max().item() < TOL


def test_training_with_classifier_dropout_applies_the_mask_before_the_heads():
    """ACMIL_GA(droprate=0.25).train() (network.py:10-19 through transformer.py:292-299): torch's generator draws one mask per head on
    the [1, D_inner] feature; with the same generator state the logits are the oracle's heads on the masked features, the gradients
    flow through the mask; in eval mode the model is the fused path's, bit for bit."""
    from oracle import ga_oracle as O
    import torch.nn.functional as F
    D, Di, K, C, p = 512, 256, 5, 2, 0.25
    sd = O.default_state_dict(D, Di, C, K)
    x = O.synthetic_bag(700, D, 9)[0]
    model = _model(sd, droprate=p)
    plain = _model(sd)
    with torch.no_grad():
        a, b = model.eval()(x.cuda().unsqueeze(0)), plain.eval()(x.cuda().unsqueeze(0))
    assert not model._generic() and all(torch.equal(u, v) for u, v in zip(a, b))          # eval: dropout is the identity, fused path
    model.train()
    assert model._generic()
    u = torch.rand(K, 10, generator=torch.Generator().manual_seed(2))
    torch.manual_seed(123)
    sub, slide, attn = model(x.cuda().unsqueeze(0), uniforms=u.cuda())
    torch.manual_seed(123)
    masks = [F.dropout(torch.ones(1, Di, device="cuda"), p, True) for _ in range(K + 1)]
    ref = O.acmil_ga_forward(x.unsqueeze(0), sd, n_token=K, n_masked_patch=10, mask_drop=0.6, uniforms=u, training=True)
    h = torch.relu(x @ sd["dimreduction.fc1.weight"].T)
    af = torch.softmax(ref["A_out"].reshape(K, -1), 1) @ h
    for i in range(K):
        want = (af[i] * masks[i][0].cpu()) @ sd["classifier.%d.fc.weight" % i].T + sd["classifier.%d.fc.bias" % i]
        assert (sub[i].detach().cpu() - want).abs().max().item() < TOL
    want = (af.mean(0) * masks[K][0].cpu()) @ sd["Slide_classifier.fc.weight"].T + sd["Slide_classifier.fc.bias"]
    assert (slide[0].detach().cpu() - want).abs().max().item() < TOL
    assert any((m == 0).any() for m in masks)
    (sub.sum() + slide.sum()).backward()
    gw = model.classifier[0].fc.weight.grad                      # d sum(logits) / d W[c, :] = masked feature
    assert (gw[0].cpu() - af[0] * masks[0][0].cpu()).abs().max().item() < TOL
    losses, _ = model.train_step(x.cuda().unsqueeze(0), torch.tensor([1]).cuda(), uniforms=u.cuda())
    assert torch.isfinite(losses).all() and all(torch.isfinite(q.grad).all() for q in model.parameters())


def test_mha_module_matches_reference():
    """`MHA(conf)` (transformer.py:86-104): logits at the constructor's query and at a trained-size query, gradients of the
    cross-entropy with Dropout(0.1) off (eval mode, gradients enabled)."""
    from acmil_amd.architecture.transformer import MHA
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "mha_single_n350_d384_c3.npz"))

    class Conf:
        D_feat, D_inner, n_class, n_token = 384, 128, 3, 1

    model = MHA(Conf)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    assert set(sd) == set(model.state_dict())
    q1 = sd["q"].clone()
    model.load_state_dict(sd)
    model = model.cuda().eval()
    x = torch.from_numpy(z["x"]).cuda()
    logits = model(x)
    assert logits.shape == (1, 3)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z["logits"], rtol=0, atol=TOL)
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(z["label"]).cuda()).backward()
    for name, p in model.named_parameters():
        ref = z["grad." + name]
        scale = np.abs(ref).max()
        if scale < 1e-7:              # k_proj.bias: a constant shift of every score, analytically zero under the softmax
            continue
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-3 * scale, (name, np.abs(p.grad.cpu().numpy() - ref).max() / scale)
    # the constructor's query (std 1e-6: the scores are equal to ~1e-6, the attention is uniform): logits of the fixture's first capture
    sd0 = dict(sd)
    sd0["q"] = torch.zeros_like(q1)
    m0 = MHA(Conf)
    m0.load_state_dict(sd0)
    with torch.no_grad():
        l0 = m0.cuda().eval()(x)
    assert np.abs(l0.cpu().numpy() - z["logits_init"]).max() < TOL


def test_attention_layers_standalone_other_head_count_and_downsampling():
    """`MutiHeadAttention(128, 4, downsample_rate=2)` and `MutiHeadAttention_modify` (transformer.py:107-236) called directly, as the
    reference's classes allow: output, raw attention logits and every gradient of sum(out^2) against the real reference (eval mode)."""
    from acmil_amd.architecture.transformer import MutiHeadAttention, MutiHeadAttention_modify
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "mha_layer_h4_ds2_n500_e128.npz"))
    att = MutiHeadAttention(128, 4, downsample_rate=2)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    assert set(sd) == set(att.state_dict()) and att.q_proj.weight.shape == (64, 128) and att.out_proj.weight.shape == (128, 64)
    att.load_state_dict(sd)
    att = att.cuda().eval()
    q = torch.from_numpy(z["q"]).cuda().requires_grad_(True)
    kv = torch.from_numpy(z["kv"]).cuda()
    out, attn = att(q, kv, kv)
    assert out.shape == (3, 128) and attn.shape == (4, 3, 500)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=0, atol=TOL)
    np.testing.assert_allclose(attn.detach().cpu().numpy(), z["attn"], rtol=0, atol=TOL)
    out.square().sum().backward()
    assert np.abs(q.grad.cpu().numpy() - z["grad_q"]).max() <= 3e-3 * np.abs(z["grad_q"]).max()
    for name, p in att.named_parameters():
        ref = z["grad." + name]
        scale = np.abs(ref).max()
        if scale < 1e-7:
            continue
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-3 * scale, (name, np.abs(p.grad.cpu().numpy() - ref).max() / scale)
    mod = MutiHeadAttention_modify(128, 4, downsample_rate=2)
    mod.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("wm.")})
    mod = mod.cuda().eval()
    with torch.no_grad():
        om = mod(kv, torch.from_numpy(z["pa"]).cuda())
    np.testing.assert_allclose(om.cpu().numpy(), z["out_modify"], rtol=0, atol=TOL)
