"""The ACMIL_MHA oracle (oracle/mha_oracle.py) against the vectors captured from the real reference module."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import mha_oracle as MO

CASES = ["mha_eval_n1000_d384_k5_c2_init", "mha_eval_n1000_d384_k5_c2_q05", "mha_eval_n257_d512_k1_c7_init", "mha_eval_n257_d512_k1_c7_q05"]


def load_mha(name):
    case, sd = load_golden(name)
    sd = dict(sd)
    sd["q"] = torch.from_numpy(case["q"])       # the q05 cases override the stored init query
    return case, sd


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    case, sd = load_mha(name)
    out = MO.acmil_mha_forward(torch.from_numpy(case["x"]), sd, int(case["n_token"]))
    np.testing.assert_allclose(out["attns"].numpy(), case["attns"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["sub_preds"].numpy(), case["sub_preds"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["slide_pred"].numpy(), case["slide_pred"], rtol=0, atol=2e-6)


def test_default_state_dict_has_reference_keys():
    case, sd = load_mha(CASES[0])
    mine = MO.default_state_dict(384, 128, 2, 5, seed=1)
    assert set(mine) == set(sd) and all(mine[k].shape == sd[k].shape for k in sd)
