"""GPU parity of the GA path at the reference's wider feature families (Step3_WSI_classification_ACMIL.py:78-87:
path-clip-L-336 768/384, UNI 1024/512, GigaPath 1536/768) against fixtures captured from the real reference
(tests/golden/make_golden_wide.py), plus forward_feature(use_attention_mask=True) (reference transformer.py:338-347)."""
import os

import numpy as np
import pytest
import torch

from conftest import case_dims, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
WIDE = ["d768_k5_c2", "d1024_k5_c7", "d1536_k5_c2"]
# other branch counts (`--n_token` is free in the reference: Step3_WSI_classification_ACMIL.py:39, transformer.py:292-301), from the
# real reference (tests/golden/make_golden_ntoken.py): K = 8 and 10 at the fused widths, K = 16 at the UNI width
NTOK = ["d512_k8_c2", "d384_k10_c7", "d1024_k16_c2"]


def _model(sd, precision, **kw):
    from acmil_amd.architecture.transformer import ACMIL_GA
    d, di, k, c = case_dims(sd)

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    m = ACMIL_GA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6, precision=precision, **kw)
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("tag", WIDE + NTOK)
def test_wide_eval_forward_matches_reference(tag, precision):
    case, sd = load_golden("ga_eval_n300_" + tag)
    d, di, k, c = case_dims(sd)
    model = _model(sd, precision).eval()
    x = torch.from_numpy(case["x"]).cuda()            # fp16 bag, as stored on disk
    with torch.no_grad():
        sub, slide, a = model(x.float().unsqueeze(0) if x.dim() == 2 else x.float())
        sub16, slide16, a16 = model(x.unsqueeze(0) if x.dim() == 2 else x)      # 16-bit bag fed directly
        feat = model.forward_feature(x.float())
    assert a.shape == (1, k, 300) and sub.shape == (k, c) and slide.shape == (1, c) and feat.shape == (1, di)
    np.testing.assert_allclose(a.cpu().numpy(), case["A_out"], rtol=0, atol=TOL)
    np.testing.assert_allclose(sub.cpu().numpy(), case["sub_preds"], rtol=0, atol=TOL)
    np.testing.assert_allclose(slide.cpu().numpy(), case["slide_pred"], rtol=0, atol=TOL)
    np.testing.assert_allclose(feat.cpu().numpy(), case["bag_feat"], rtol=0, atol=TOL)
    assert torch.equal(a, a16) and torch.equal(sub, sub16)
    # top-k of every branch identical to the reference's
    ref_top = np.argsort(-case["A_out"][0], axis=-1, kind="stable")[:, :10]
    got_top = torch.topk(a[0], 10, dim=-1).indices.cpu().numpy()
    assert np.array_equal(ref_top, got_top)
    # batched entry: same per-bag triples
    outs = model.forward_batch([x.float()[0] if x.dim() == 3 else x.float()] * 2)
    assert torch.equal(outs[1][0], sub) and torch.equal(outs[0][2], a)


@pytest.mark.parametrize("tag", WIDE + NTOK + ["d512_k5_c2m", "fused_d512"])
def test_forward_feature_with_attention_mask_matches_reference(tag):
    """forward_feature(x, use_attention_mask=True): G1-G3 with the STKIM mask, then bag_feat (transformer.py:338-347)."""
    if tag == "fused_d512":
        # no reference capture at this width: the masked path must equal the same module's own masked forward internals
        case, sd = load_golden("ga_train_n640_d512_k5_c2")
        model = _model(sd, "f16x3").train()
        x = torch.from_numpy(case["x"]).cuda().float()
        u = torch.from_numpy(case["uniforms"]).cuda()
        with torch.no_grad():
            f1 = model.forward_feature(x, use_attention_mask=True, uniforms=u)
            packed, dims = model._packed()
            out = model._masked_forward(x[0], packed, dims, u, want_bag_feat=True)
            f0 = model.forward_feature(x)
        assert torch.equal(f1[0], out["bag_feat"]) and not torch.equal(f1, f0)
        # masked positions = the reference's (fixture), so the masked bag feature is the reference's too: check via afeat mean
        assert np.array_equal(np.sort(out["masked_idx"].cpu().numpy(), axis=1), np.sort(case["masked_idx"], axis=1))
        return
    case, sd = load_golden("ga_eval_n300_" + tag)      # "d512_k5_c2m": the reference's masked feature at the FUSED width 512 / 256
    model = _model(sd, "f16x3").train()
    x = torch.from_numpy(case["x"]).cuda().float()
    with torch.no_grad():
        f = model.forward_feature(x, use_attention_mask=True, uniforms=torch.from_numpy(case["bag_feat_masked_uniforms"]).cuda())
    np.testing.assert_allclose(f.cpu().numpy(), case["bag_feat_masked"], rtol=0, atol=TOL)
    assert np.abs(case["bag_feat_masked"] - case["bag_feat"]).max() > 1e-6      # the mask does change the feature


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("tag", WIDE + NTOK)
def test_wide_train_step_matches_reference(tag, precision):
    """One training step (losses, every parameter gradient, AdamW update) against the reference's own train_one_epoch capture,
    through BOTH entries: torch.autograd over the HIP Function, and the autograd-free fused ACMIL_GA.train_step."""
    from test_train_gpu import _losses
    case, sd = load_golden("ga_train_n200_" + tag)
    d, di, k, c = case_dims(sd)
    stride = int(case["w1_row_stride"])
    model = _model(sd, precision).train()
    x = torch.from_numpy(case["x"]).cuda()
    label = torch.from_numpy(case["label"]).cuda()
    u = torch.from_numpy(case["uniforms"]).cuda()
    sub, slide, attn = model(x.float(), uniforms=u)
    assert np.array_equal(np.sort(model._last["masked_idx"].cpu().numpy(), axis=1), np.sort(case["masked_idx"], axis=1))
    assert np.array_equal(model._last["topk_idx"].cpu().numpy(), case["topk_idx"])
    loss0, loss1, diff = _losses(sub, slide, attn, label, k)
    assert float(loss0.detach()) == pytest.approx(float(case["loss0"]), abs=2e-5)
    assert float(loss1.detach()) == pytest.approx(float(case["loss1"]), abs=2e-5)
    (diff + loss0 + loss1).backward()

    def ref_of(prefix, name_p):
        return case[prefix + name_p]

    def thin(name_p, arr):
        return arr[::stride] if name_p == "dimreduction.fc1.weight" else arr

    auto = {}
    for name_p, p in model.named_parameters():
        ref = ref_of("grad.", name_p)
        got = thin(name_p, p.grad.cpu().numpy())
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(got - ref).max() <= 2e-4 * scale + 1e-7, (name_p, np.abs(got - ref).max(), scale)
        auto[name_p] = p.grad.clone()
    # fused step writes the same gradients (no autograd)
    fused = _model(sd, precision).train()
    losses, _ = fused.train_step(x.unsqueeze(0) if x.dim() == 2 else x, label, uniforms=u)
    assert float(losses[0]) == pytest.approx(float(case["loss0"]), abs=2e-5)
    assert float(losses[1]) == pytest.approx(float(case["loss1"]), abs=2e-5)
    for name_p, p in fused.named_parameters():
        a = auto[name_p]
        assert (p.grad - a).abs().max().item() <= 2e-4 * a.abs().max().item() + 1e-7, name_p
    opt = torch.optim.AdamW(model.parameters(), lr=float(case["lr"]), weight_decay=float(case["wd"]))
    opt.step()
    for name_p, p in model.named_parameters():
        ok = np.abs(ref_of("grad.", name_p)) >= 1e-6
        got, ref = thin(name_p, p.detach().cpu().numpy()), ref_of("after.", name_p)
        assert np.abs(got - ref)[ok].max(initial=0.0) <= 2e-6, name_p


def test_trainer_main_with_wide_family(tmp_path):
    """`--arch ga --pretrain UNI` (1024/512) trains end to end (Step3_WSI_classification_ACMIL.py:82-83, :126)."""
    from acmil_amd import train as T
    out = str(tmp_path / "uni")
    T.main(["--arch", "ga", "--pretrain", "UNI", "--synthetic_slides", "12", "--synthetic_patches", "400", "--train_epoch", "2",
            "--out_dir", out, "--n_token", "5", "--n_masked_patch", "10", "--mask_drop", "0.6"])
    ck = torch.load(os.path.join(out, "checkpoint-last.pth"), weights_only=False)
    assert ck["model"]["dimreduction.fc1.weight"].shape == (512, 1024) and ck["epoch"] == 1


def test_trainer_main_with_eight_branches(tmp_path):
    """`--arch ga --n_token 8` trains end to end (Step3_WSI_classification_ACMIL.py:39: any branch count; K > 5 runs the composed
    kernels at the fused width 512 / 256 as well)."""
    from acmil_amd import train as T
    out = str(tmp_path / "k8")
    T.main(["--arch", "ga", "--pretrain", "plip", "--synthetic_slides", "12", "--synthetic_patches", "500", "--train_epoch", "2",
            "--out_dir", out, "--n_token", "8", "--n_masked_patch", "10", "--mask_drop", "0.6"])
    ck = torch.load(os.path.join(out, "checkpoint-last.pth"), weights_only=False)
    assert ck["model"]["attention.attention_weights.weight"].shape == (8, 128) and len([k for k in ck["model"] if k.startswith("classifier.")]) == 16


def test_range_guard_on_the_composed_path_uses_the_projection_kernels_status_word():
    """D_inner 512 (no fused family): a bag value outside the f16 range is flagged by the range word the projection kernel leaves
    (csrc/linear_kernel.h) -- no extra pass over h -- and the bag is redone in fp32 arithmetic: outputs equal the fp32-mode module's."""
    case, sd = load_golden("ga_eval_n300_d1024_k5_c7")
    guarded, exact = _model(sd, "f16x3").eval(), _model(sd, "fp32").eval()
    x = torch.from_numpy(case["x"]).float().clone()
    x[0, 17, 5] = 2.0e5
    with torch.no_grad():
        before = guarded.range_fallbacks
        sub_g, slide_g, a_g = guarded(x.cuda())
        sub_e, slide_e, a_e = exact(x.cuda())
        assert guarded.range_fallbacks == before + 1
        assert torch.isfinite(a_g).all() and torch.equal(a_g, a_e) and torch.equal(sub_g, sub_e) and torch.equal(slide_g, slide_e)
        ok = torch.from_numpy(case["x"]).float().cuda()
        guarded(ok)
        assert guarded.range_fallbacks == before + 1          # an in-range bag stays on the split-f16 path


# ------------------------------------------------------------------------------------------------ grouped eval of the composed families
@pytest.mark.parametrize("tag", ["d1536_k5_c2", "d512_k8_c2", "d1024_k16_c2"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_forward_group_equals_per_slide_forward_and_reference(tag, dtype):
    """ACMIL_GA.forward_group (rows of several slides back to back: ONE projection launch, ONE gated-score launch, per-bag pooling
    tiles, batched merge + heads) == `model(x)` per slide BIT FOR BIT (every output element is the same fixed-order sum), and the
    fixture bag inside the group matches the reference (transformer.py:305-330)."""
    case, sd = load_golden("ga_eval_n300_" + tag)
    d, di, k, c = case_dims(sd)
    model = _model(sd, "f16x3").eval()
    assert model._is_composed_groupable("f16x3")
    x0 = torch.from_numpy(case["x"]).cuda()
    x0 = (x0[0] if x0.dim() == 3 else x0).to(dtype)
    g = torch.Generator().manual_seed(7)
    rows = [129, 300, 1, 1000, 57, 128, 257]                       # ragged, tile-boundary cases of the 128-row pooling tiles and the 256-row projection tiles
    bags = [(torch.randn(n, d, generator=g) * 0.5).cuda().to(dtype) if i != 1 else x0 for i, n in enumerate(rows)]
    xcat = torch.cat(bags, 0)
    with torch.no_grad():
        triples = model.forward_group(xcat, rows)
        singles = [model(b.unsqueeze(0)) for b in bags]
    assert len(triples) == len(rows)
    for (sub, slide, a), (sub1, slide1, a1), n in zip(triples, singles, rows):
        assert a.shape == (1, k, n) and sub.shape == (k, c) and slide.shape == (1, c)
        assert torch.equal(a, a1) and torch.equal(sub, sub1) and torch.equal(slide, slide1)
    sub, slide, a = triples[1]
    np.testing.assert_allclose(a.cpu().numpy(), case["A_out"], rtol=0, atol=TOL)
    np.testing.assert_allclose(sub.cpu().numpy(), case["sub_preds"], rtol=0, atol=TOL)
    np.testing.assert_allclose(slide.cpu().numpy(), case["slide_pred"], rtol=0, atol=TOL)
    assert model.range_fallbacks == 0


def test_forward_group_more_than_sixteen_bags_guard_and_other_families():
    """> 16 bags go out as several groups; a group with a value outside the f16 range is repeated in fp32 (== the per-slide guarded
    forward within the parity bound); a deferred ticket reports it instead; a fused family routes to forward_batch."""
    case, sd = load_golden("ga_eval_n300_d1536_k5_c2")
    d, di, k, c = case_dims(sd)
    model = _model(sd, "f16x3").eval()
    g = torch.Generator().manual_seed(11)
    rows = [40 + 13 * i for i in range(19)]
    bags = [(torch.randn(n, d, generator=g) * 0.5).cuda() for n in rows]
    with torch.no_grad():
        triples = model.forward_group(torch.cat(bags, 0), rows)
        for (sub, slide, a), b in zip(triples, bags):
            s1, l1, a1 = model(b.unsqueeze(0))
            assert torch.equal(a, a1) and torch.equal(sub, s1) and torch.equal(slide, l1)
        bad = [b.clone() for b in bags[:5]]
        bad[2][7, 3] = 1.0e5                                     # |x| >= 65504: the split-f16 arithmetic cannot represent it
        rows5 = rows[:5]
        before = model.range_fallbacks
        t_bad = model.forward_group(torch.cat(bad, 0), rows5)
        assert model.range_fallbacks > before
        m32 = _model(sd, "fp32").eval()
        for (sub, slide, a), b in zip(t_bad, bad):
            s1, l1, a1 = m32(b.unsqueeze(0))
            assert torch.isfinite(a).all()
            np.testing.assert_allclose(a.cpu().numpy(), a1.cpu().numpy(), rtol=0, atol=TOL)
            np.testing.assert_allclose(sub.cpu().numpy(), s1.cpu().numpy(), rtol=0, atol=TOL)
        _, ticket = model.forward_group(torch.cat(bad, 0), rows5, defer_guard=True)
        assert ticket is not None and int(ticket) != 0
        _, ticket = model.forward_group(torch.cat(bags[:5], 0), rows5, defer_guard=True)
        assert ticket is not None and int(ticket) == 0
    # a fused family: the same call, served by forward_batch on the row views
    case2, sd2 = load_golden("ga_eval_n257_d512_k5_c2")
    m2 = _model(sd2, "f16x3").eval()
    assert not m2._is_composed_groupable("f16x3")
    xb = torch.randn(300, 512, generator=g).cuda()
    with torch.no_grad():
        t = m2.forward_group(torch.cat([xb, xb[:77]], 0), [300, 77])
        s1, l1, a1 = m2(xb.unsqueeze(0))
    np.testing.assert_allclose(t[0][2].cpu().numpy(), a1.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(t[0][0].cpu().numpy(), s1.cpu().numpy(), rtol=0, atol=2e-6)


def test_evaluate_groups_the_composed_family(tmp_path):
    """train.evaluate on a GigaPath-shaped model: staged row-concatenated groups through forward_group == the per-slide loop."""
    from acmil_amd import train as T
    case, sd = load_golden("ga_eval_n300_d1536_k5_c2")
    d, di, k, c = case_dims(sd)
    model = _model(sd, "f16x3").eval()
    data = T.SyntheticBags(21, (60, 400), d, c, seed=3)
    conf = T.Struct(n_class=c, arch="ga")
    dev = torch.device("cuda", torch.cuda.current_device())
    det_g, det_s = {}, {}
    with torch.no_grad():
        rg = T.evaluate(model, data, dev, conf, detail=det_g)
        rs = T.evaluate(model, data, dev, conf, batched=False, detail=det_s)
    assert torch.equal(det_g["prob"], det_s["prob"])
    np.testing.assert_allclose(det_g["loss"].numpy(), det_s["loss"].numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(det_g["div"].numpy(), det_s["div"].numpy(), rtol=0, atol=1e-6)
    assert rg[:3] == rs[:3]
