"""oracle/attn_variants_oracle.py against vectors captured from the real reference modules (DTFD attention, IBMIL, CLAM_SB)."""
import numpy as np
import torch

from conftest import load_golden
from oracle import attn_variants_oracle as VO


def test_dtfd_attention_with_classifier():
    case, sd = load_golden("variants_dtfd_n700_l256_k3_c4")
    x = torch.from_numpy(case["x"])
    np.testing.assert_allclose(VO.attention_with_classifier(x, sd).numpy(), case["pred"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(VO.attention_gated(x, sd, "attention.").numpy(), case["A_norm"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(VO.attention_gated(x, sd, "attention.", is_norm=False).numpy(), case["A_raw"], rtol=0, atol=1e-6)


def test_ibmil():
    case, sd = load_golden("variants_ibmil_n900_d384_c3")
    y, m, a = VO.ibmil_forward(torch.from_numpy(case["x"]), sd)
    np.testing.assert_allclose(y.numpy(), case["Y_prob"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(m.numpy(), case["M"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(a.numpy(), case["A"], rtol=0, atol=1e-7)


def test_clam_sb():
    for name in ("variants_clam_small_n600_d384", "variants_clam_big_n600_d256"):
        case, sd = load_golden(name)
        logits, a, _ = VO.clam_sb_forward(torch.from_numpy(case["x"]), sd)
        np.testing.assert_allclose(logits.numpy(), case["logits"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(a.numpy(), case["A_raw"], rtol=0, atol=2e-6)
