"""GPU parity of ACMIL_MHA (csrc/mha.hip, single-query folding) against the reference fixtures and the oracle."""
import numpy as np
import pytest
import torch

from test_oracle_mha import CASES, load_mha

pytestmark = pytest.mark.gpu


def _model(sd, d, di, k, c, precision):
    from acmil_amd.architecture.transformer import ACMIL_MHA

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k
    m = ACMIL_MHA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6, precision=precision)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.cuda().eval()


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("name", CASES)
def test_matches_reference_golden(name, precision):
    case, sd = load_mha(name)
    di, d = sd["dimreduction.fc1.weight"].shape
    k, c = int(case["n_token"]), sd["Slide_classifier.fc.weight"].shape[0]
    model = _model(sd, d, di, k, c, precision)
    with torch.no_grad():
        sub, slide, attns = model(torch.from_numpy(case["x"]).cuda())
    assert sub.shape == (k, c) and slide.shape == (1, c) and attns.shape == case["attns"].shape
    np.testing.assert_allclose(attns.cpu().numpy(), case["attns"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(sub.cpu().numpy(), case["sub_preds"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(slide.cpu().numpy(), case["slide_pred"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("n,d,di,k,c", [(20000, 512, 256, 5, 2), (777, 1024, 512, 3, 4), (1, 384, 128, 2, 2)])
def test_matches_oracle_other_shapes(n, d, di, k, c):
    from oracle import mha_oracle as MO
    sd = MO.default_state_dict(d, di, c, k, seed=5)
    x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(n))
    ref = MO.acmil_mha_forward(x, sd, k)
    model = _model(sd, d, di, k, c, "f16x3")
    with torch.no_grad():
        sub, slide, attns = model(x.cuda())
    assert (attns.cpu() - ref["attns"]).abs().max() < 1e-5
    assert (sub.cpu() - ref["sub_preds"]).abs().max() < 1e-4
    assert (slide.cpu() - ref["slide_pred"]).abs().max() < 1e-4


def test_train_mode_without_gradients_raises():
    """train mode draws dropout / mask-drop randomness: that path needs the differentiable forward (gradients enabled)"""
    from oracle import mha_oracle as MO
    sd = MO.default_state_dict(384, 128, 2, 2, seed=1)
    model = _model(sd, 384, 128, 2, 2, "f16x3").train()
    with pytest.raises(NotImplementedError), torch.no_grad():
        model(torch.randn(1, 100, 384, device="cuda"))
