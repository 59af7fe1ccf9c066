"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/acmil_hip.h declares; argument validation works without a GPU; host modules mirror the reference."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from acmil_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from acmil_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "acmil_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(acmil_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.acmil_version().startswith(b"acmil_hip")


def test_product_library_reads_no_environment_variable(lib):
    """Measurement switches (csrc/ab_knobs.h) are compiled out of the product build: no ACMIL_* name is even present in
    libacmil_hip.so, so a stray variable in a user's environment cannot change kernels or summation order; the A/B twin built
    beside it (tests / tools load it through ACMIL_HIP_LIB) carries them, and both export the same C ABI."""
    import subprocess
    from acmil_amd import _lib
    from conftest import AB_LIB
    prod = os.path.join(ROOT, "acmil_amd", "libacmil_hip.so")

    def names(path):
        return set(re.findall(rb"ACMIL_[A-Z0-9_]+", open(path, "rb").read()))

    assert names(prod) == set(), names(prod)
    ab = names(AB_LIB)
    assert {b"ACMIL_GA2_WAVES", b"ACMIL_TM_SIDE_STREAM", b"ACMIL_GA_BWD_TILE", b"ACMIL_LIN64"} <= ab
    # the only environment variables of the product live in Python and choose WHICH library / reduction runs, not how it computes
    py = set()
    for root, _, files in os.walk(os.path.join(ROOT, "acmil_amd")):
        for f in files:
            if f.endswith(".py"):
                py |= set(re.findall(r"environ[^\n]*?[\"'](ACMIL_[A-Z0-9_]+)[\"']", open(os.path.join(root, f)).read()))
    assert py == {"ACMIL_HIP_LIB", "ACMIL_DP_REDUCE", "ACMIL_PEER_TIMEOUT_S"}, py
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("acmil_")}
    assert exported(prod) == exported(AB_LIB) == set(_lib.SIGNATURES)


def test_size_queries_and_argument_validation(lib):
    from acmil_amd import _lib
    # sizes: D=512, Di=256, K=5, C=2
    for mode in (_lib.MODE_F32, _lib.MODE_F16X3, _lib.MODE_F16):
        n = lib.acmil_ga_packed_bytes(512, 256, 128, 5, 2, mode)
        stream = 768 * 1024 if mode != _lib.MODE_F16 else 384 * 1024
        assert n >= stream and n % 256 == 0
    assert lib.acmil_ga_packed_bytes(500, 256, 128, 5, 2, 0) == 0          # D not a multiple of 64
    assert lib.acmil_ga_packed_bytes(512, 256, 64, 5, 2, 0) == 0           # Da must be 128
    assert lib.acmil_ga_workspace_bytes(50000, 512, 256, 5, 2, 1) >= 391 * 5 * 258 * 4
    # error codes instead of crashes (no kernel is launched for invalid arguments)
    assert lib.acmil_ga_forward(None, 0, 10, None, 512, 256, 128, 5, 2, 0, None, None, None, None, None, None, 1, None, None) == -3
    assert lib.acmil_ga_forward(None, 0, 0, None, 512, 256, 128, 5, 2, 0, None, None, None, None, None, None, 1, None, None) == -1
    # K above ACMIL_MAX_TOKENS (16) is refused up front; K = 6..16 passes the shape check (the fused families are K <= 5: their
    # dispatch answers -2 once the pointers are there; the composed entry points take those K)
    assert lib.acmil_ga_forward(None, 0, 10, None, 512, 256, 128, 17, 2, 0, None, None, None, None, None, None, 1, None, None) == -2
    assert lib.acmil_ga_forward(None, 0, 10, None, 512, 256, 128, 9, 2, 0, None, None, None, None, None, None, 1, None, None) == -3
    assert lib.acmil_ga_packed_bytes(512, 256, 128, 16, 2, 1) > 0 and lib.acmil_ga_packed_bytes(512, 256, 128, 17, 2, 1) == 0
    assert lib.acmil_ga_train_step_workspace_bytes(1000, 512, 256, 8, 2, 10) == 0          # the one-call step: K <= 5
    assert lib.acmil_ga_loss_workspace_bytes(1000, 16) > lib.acmil_ga_loss_workspace_bytes(1000, 5) > 0
    assert lib.acmil_stkim_select(None, 100, 5, 200, 0, None, None, None, None, None) == -1   # k > N
    assert lib.acmil_stkim_workspace_bytes(50000, 5, 10) >= 5 * 13 * 10 * 8
    # the one-call training step: workspace covers the saved h, the scores' partials and the backward scratch; bad arguments
    # come back as codes
    ws = lib.acmil_ga_train_step_workspace_bytes(10000, 512, 256, 5, 7, 10)
    assert ws >= 10000 * 256 * 4 * 2 + 10000 * 256 * 4 and ws % 256 == 0
    assert lib.acmil_ga_train_step_workspace_bytes(0, 512, 256, 5, 7, 10) == 0
    args = [None, 0, 100, None, 1] + [None] * 7 + [None, None, None, None] + [None] * 7 + [None, None, None, None]
    assert lib.acmil_ga_train_step(*args, 512, 256, 128, 5, 7, 1, None, None, 10, 6, None, None, None, None, None, None, None, None, None) == -3
    assert lib.acmil_ga_train_step(*args, 512, 256, 128, 5, 7, 1, None, None, 200, 6, None, None, None, None, None, None, None, None, None) == -1   # k_top > N
    assert lib.acmil_adamw_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, None, None) == -3
    assert lib.acmil_adamw_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None, None) == -1          # step ordinals start at 1
    assert lib.acmil_linear_packed_bytes(256, 1024) > 0 and lib.acmil_linear_packed_bytes(100, 1024) == 0                   # n_out % 128
    # pooling + heads of a row-concatenated group (the composed families' batched eval): partials of sum ceil(rows / 128) <= tiles(N) + bags
    import ctypes
    w16 = lib.acmil_ga_pool_group_workspace_bytes(800000, 16, 768, 5)
    assert w16 >= 256 + (6250 + 16) * 5 * 770 * 4 + 16 * 5 * 768 * 4 and lib.acmil_ga_pool_group_workspace_bytes(800000, 17, 768, 5) == 0
    rows = (ctypes.c_int * 2)(60, 40)
    assert lib.acmil_ga_pool_group(None, None, 100, 2, rows, None, 1536, 768, 128, 5, 2, 1, None, None, None, None, 1, None, None) == -3
    assert lib.acmil_ga_pool_group(None, None, 100, 17, rows, None, 1536, 768, 128, 5, 2, 1, None, None, None, None, 1, None, None) == -1


def test_modules_mirror_reference_surface():
    from acmil_amd.architecture.network import Classifier_1fc, DimReduction
    from acmil_amd.architecture.transformer import ABMIL, ACMIL_GA, Attention_Gated

    class Conf:
        D_feat, D_inner, n_class, n_token = 384, 128, 7, 5

    m = ACMIL_GA(Conf, n_token=5, n_masked_patch=10, mask_drop=0.6)
    keys = list(m.state_dict().keys())
    assert keys[:7] == ["dimreduction.fc1.weight", "attention.attention_V.0.weight", "attention.attention_V.0.bias",
                        "attention.attention_U.0.weight", "attention.attention_U.0.bias",
                        "attention.attention_weights.weight", "attention.attention_weights.bias"]
    assert "classifier.4.fc.weight" in keys and "Slide_classifier.fc.bias" in keys and len(keys) == 19
    assert m.state_dict()["dimreduction.fc1.weight"].shape == (128, 384)
    assert m.state_dict()["attention.attention_weights.weight"].shape == (5, 128)
    assert m.n_token == 5 and m.n_masked_patch == 10 and m.mask_drop == 0.6
    ab = ABMIL(Conf)
    assert "classifier.fc.weight" in ab.state_dict() and len(ab.state_dict()) == 9
    assert isinstance(m.dimreduction, DimReduction) and isinstance(m.Slide_classifier, Classifier_1fc)
    assert isinstance(m.attention, Attention_Gated)
    # no CPU fallback: a CPU bag is rejected loudly
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 8, 384))


def test_fixtures_load_into_modules():
    from conftest import case_dims, load_golden
    from acmil_amd.architecture.transformer import ACMIL_GA
    case, sd = load_golden("ga_eval_n257_d512_k5_c2")
    d, di, k, c = case_dims(sd)

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    m = ACMIL_GA(Conf, n_token=k)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def test_classifier_dropout_and_other_attention_widths_construct_and_route():
    """Classifier_1fc(droprate != 0) (reference network.py:10-16) and Attention_Gated(D != 128) (transformer.py:240): same state_dict
    keys as the reference's modules, the aggregators construct, and the routing is as documented -- eval mode with dropout stays on the
    fused path (dropout is the identity there), training mode with dropout / another attention width / residual blocks take the
    op-by-op path; none of them has a CPU fallback (a CPU tensor is refused by the library wrappers, not silently computed)."""
    from acmil_amd.architecture.network import DimReduction
    from acmil_amd.architecture.transformer import ABMIL, ACMIL_GA, MHA

    class Conf:
        D_feat, D_inner, n_class, n_token = 384, 128, 3, 5

    plain = ACMIL_GA(Conf, n_token=5)
    drop = ACMIL_GA(Conf, droprate=0.25, n_token=5)
    assert list(plain.state_dict()) == list(drop.state_dict())
    assert drop.train()._generic() and not drop.eval()._generic() and not plain.train()._generic()
    wide = ACMIL_GA(Conf, D=64, n_token=5)
    assert wide._generic() and wide.state_dict()["attention.attention_V.0.weight"].shape == (64, 128)
    assert wide.state_dict()["attention.attention_weights.weight"].shape == (5, 64)
    assert ABMIL(Conf, D=256)._generic() and ABMIL(Conf, droprate=0.1).train()._generic()
    assert list(DimReduction(384, 128, numLayer_Res=2).state_dict()) == [
        "fc1.weight", "resBlocks.0.block.0.weight", "resBlocks.0.block.2.weight", "resBlocks.1.block.0.weight", "resBlocks.1.block.2.weight"]
    assert sorted(MHA(Conf).state_dict())[:3] == ["attention.k_proj.bias", "attention.k_proj.weight", "attention.layer_norm.bias"]
    for m in (drop.train(), wide, drop.eval()):
        with pytest.raises(RuntimeError):          # reaches the library wrappers: a CPU tensor is refused there
            with torch.no_grad():
                m(torch.zeros(1, 4, 384))
