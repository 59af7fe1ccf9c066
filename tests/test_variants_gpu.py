"""GPU parity of the N4 modules (DTFD attention blocks, IBMIL, CLAM_SB) against the reference fixtures and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_dtfd_attention_with_classifier(precision):
    from acmil_amd.architecture.Attention import Attention_with_Classifier
    case, sd = load_golden("variants_dtfd_n700_l256_k3_c4")
    m = Attention_with_Classifier(L=256, D=128, K=3, num_cls=4, precision=precision)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd); m = m.cuda().eval()
    x = torch.from_numpy(case["x"]).cuda()
    with torch.no_grad():
        np.testing.assert_allclose(m(x).cpu().numpy(), case["pred"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(m.attention(x).cpu().numpy(), case["A_norm"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(m.attention(x, isNorm=False).cpu().numpy(), case["A_raw"], rtol=0, atol=1e-5)
    out = m(x)        # gradients enabled: the differentiable op-by-op path gives the same prediction
    assert out.requires_grad
    np.testing.assert_allclose(out.detach().cpu().numpy(), case["pred"], rtol=0, atol=1e-4)


def test_ibmil():
    from acmil_amd.architecture.ibmil import IBMIL
    case, sd = load_golden("variants_ibmil_n900_d384_c3")

    class Conf:
        D_feat, D_inner, n_class, c_path = 384, 128, 3, None
    m = IBMIL(Conf)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd); m = m.cuda().eval()
    with torch.no_grad():                                    # fused forward kernel
        y, mm, a = m(torch.from_numpy(case["x"]).cuda())
    y2, mm2, a2 = m(torch.from_numpy(case["x"]).cuda())        # gradients enabled: differentiable op-by-op path, same values
    assert y2.requires_grad and (y2.detach() - y).abs().max() < 1e-4 and (a2.detach() - a).abs().max() < 1e-6
    assert y.shape == (1, 3) and mm.shape == (1, 128) and a.shape == (1, 900)
    np.testing.assert_allclose(y.cpu().numpy(), case["Y_prob"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(mm.cpu().numpy(), case["M"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(a.cpu().numpy(), case["A"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,size_arg,d,di", [("variants_clam_small_n600_d384", "small", 384, 128), ("variants_clam_big_n600_d256", "big", 256, 128)])
def test_clam_sb(name, size_arg, d, di):
    from acmil_amd.architecture.clam import CLAM_SB
    case, sd = load_golden(name)

    class Conf:
        D_feat, D_inner, n_class = d, di, 2
    m = CLAM_SB(Conf, size_arg=size_arg)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd); m = m.cuda().eval()
    x = torch.from_numpy(case["x"]).cuda()
    with torch.no_grad():                                     # fused eval ops
        np.testing.assert_allclose(m(x).cpu().numpy(), case["logits"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(m(x, attention_only=True).cpu().numpy(), case["A_raw"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(m(x).detach().cpu().numpy(), case["logits"], rtol=0, atol=1e-4)      # op-by-op differentiable path
    with torch.no_grad(), pytest.raises(NotImplementedError):
        m(x, label=torch.tensor([1]), instance_eval=True)      # a training loss: needs gradients enabled


def test_generic_blocks_large_bag_vs_oracle():
    """N = 50 000, L = 512, Da = 256 (a width the fused GA kernel does not have), K = 5."""
    from acmil_amd import ops
    from oracle import attn_variants_oracle as VO
    g = torch.Generator().manual_seed(9)
    n, l, da, k = 50000, 512, 256, 5
    h = torch.randn(n, l, generator=g).relu()
    wv, wu = torch.randn(da, l, generator=g) * l ** -0.5, torch.randn(da, l, generator=g) * l ** -0.5
    bv, bu = torch.randn(da, generator=g) * 0.1, torch.randn(da, generator=g) * 0.1
    ww, bw = torch.randn(k, da, generator=g) * da ** -0.5, torch.randn(k, generator=g) * 0.1
    ref = VO.gated_scores(h, wv, bv, wu, bu, ww, bw)
    c = lambda t: t.cuda()
    A = ops.gated_scores(c(h), c(wv), c(bv), c(wu), c(bu), c(ww), c(bw))
    assert (A.cpu() - ref).abs().max() < 1e-5
    af = ops.attn_pool(c(h), A)
    assert (af.cpu() - torch.softmax(ref, 1) @ h).abs().max() < 1e-5
    assert (ops.softmax_rows(A).cpu() - torch.softmax(ref, 1)).abs().max() < 1e-7


def test_attention_map_consumers_entropy_and_heatmap():
    """div_loss of evaluate() (Step3_WSI_classification_ACMIL.py:259) and the heat-map scores of
    Step4_visualize_heatmap_camelyon.py:117-119 from the HIP row-statistics pass vs the reference's torch expressions."""
    import torch.nn.functional as F
    from acmil_amd import heatmap, ops
    g = torch.Generator().manual_seed(3)
    for shape in [(1, 5, 50000), (1, 1, 777), (8, 5, 4096)]:
        attn = (torch.randn(shape, generator=g) * 3.0).cuda()
        attn[..., 7] = -1e9                                    # masked entries of a training-mode map
        ref = torch.sum(F.softmax(attn.double(), dim=-1) * F.log_softmax(attn.double(), dim=-1)) / attn.shape[1]
        got = ops.attn_entropy_loss(attn)
        assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-6
        if shape[0] == 1:
            probs = torch.softmax(attn.double(), dim=-1)[0].mean(0)
            want = probs * probs.numel() * 2.5 * 100
            got_map = ops.attn_heatmap(attn, zoom_factor=2.5) * 100.0
            assert got_map.shape == (shape[2],)
            assert (got_map.double().cpu() - want.cpu()).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-6
    # through a model
    from oracle import ga_oracle as O
    from acmil_amd.architecture.transformer import ACMIL_GA
    sd = O.default_state_dict(512, 256, 2, 5)
    m = ACMIL_GA(type("C", (), dict(D_feat=512, D_inner=256, n_class=2, n_token=5)), n_token=5)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = O.synthetic_bag(3000, 512, 4)[0].cuda().unsqueeze(0)
    scores = heatmap.heatmap_scores(m, x, zoom_factor=1.0)
    with torch.no_grad():
        a = m(x)[2]
    want = torch.softmax(a, dim=-1)[0].mean(0) * 3000 * 100
    assert (scores - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    b0 = heatmap.branch_heatmap_scores(a, 2)
    assert (b0 - torch.softmax(a, dim=-1)[0, 2] * 3000 * 100).abs().max().item() <= 1e-4 * float(b0.abs().max())
