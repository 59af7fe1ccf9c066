"""Pin the TransMIL oracle against the reference run (tests/golden/make_golden_transmil.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import transmil_oracle as TO

CASES = ["transmil_eval_n1_d384_c2", "transmil_eval_n50_d384_c2", "transmil_eval_n129_d384_c2", "transmil_eval_n1000_d384_c2"]


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads(); torch.set_num_threads(1); yield; torch.set_num_threads(n)


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference(name):
    case, sd = load_golden(name)
    out = TO.transmil_forward(torch.from_numpy(case["x"]), sd)
    for key in ("h1", "hp", "h2", "logits"):
        np.testing.assert_allclose(out[key].numpy(), case[key], rtol=0, atol=2e-6, err_msg=key)


def test_batch_forward_matches_reference():
    """B = 2: the pinv initialisation's maxima run over batch and heads (nystrom_attention.py:16-18); the oracle keeps that coupling."""
    case, sd = load_golden("transmil_eval_b2_n200_d384_c2")
    out = TO.transmil_forward(torch.from_numpy(case["x"]), sd)
    for key in ("h2", "logits"):
        np.testing.assert_allclose(out[key].numpy(), case[key], rtol=0, atol=2e-6, err_msg=key)


def test_pinv_matches_reference():
    z = np.load("tests/golden/pinv_h8_m64.npz")
    out = TO.moore_penrose_iter_pinv(torch.from_numpy(z["x"]), 6)
    np.testing.assert_allclose(out.numpy(), z["z"], rtol=0, atol=1e-5)


def test_state_dict_helper_has_reference_keys():
    _, sd = load_golden(CASES[0])
    mine = TO.default_state_dict(384, 128, 2)
    assert set(mine) == set(sd) and all(mine[k].shape == sd[k].shape for k in sd)
