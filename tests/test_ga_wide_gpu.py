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
