import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The A/B twin of the library (csrc/ab_knobs.h): the only build that reads the ACMIL_* measurement switches.  Tests that compare
# kernel variants run their subprocesses on it (ab_environ); everything else runs on the product library, which reads none.
AB_LIB = os.path.join(ROOT, "acmil_amd", "libacmil_hip_ab.so")


def ab_environ(**knobs):
    """Environment of a subprocess that runs on the A/B build with the given ACMIL_* switches set (all others cleared)."""
    e = {k: v for k, v in os.environ.items() if not k.startswith("ACMIL_")}
    e["ACMIL_HIP_LIB"] = AB_LIB
    e.update(knobs)
    return e


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a fixture written by tests/golden/make_golden.py -> (dict of np arrays, state_dict of torch tensors)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = {k: z[k] for k in z.files}
    sd = None
    if "weights" in case:
        w = np.load(os.path.join(GOLDEN, str(case["weights"]) + ".npz"))
        sd = {k: torch.from_numpy(w[k]) for k in w.files}
    if "x_from" in case:
        case["x"] = np.load(os.path.join(GOLDEN, str(case["x_from"]) + ".npz"))["x"]
    return case, sd


def case_dims(sd):
    """(D_feat, D_inner, K, C) from a GA state_dict."""
    di, d = sd["dimreduction.fc1.weight"].shape
    k = sd["attention.attention_weights.weight"].shape[0]
    ckey = "Slide_classifier.fc.weight" if "Slide_classifier.fc.weight" in sd else "classifier.fc.weight"
    return d, di, k, sd[ckey].shape[0]


EVAL_CASES = ["ga_eval_n257_d512_k5_c2", "ga_eval_n1_d512_k5_c2", "ga_eval_n33_d512_k5_c2",
              "ga_eval_n1000_d512_k1_c2", "ga_eval_n1000_d384_k5_c7"]
TRAIN_CASES = ["ga_train_n7_d512_k5_c2", "ga_train_n640_d512_k5_c2", "ga_train_n2048_d512_k5_c7"]


@pytest.fixture(scope="session")
def have_gpu():
    return torch.cuda.is_available()


def trajectory_bags(case):
    """The five fp16 bags of ga_trajectory_d512_k5_c7 (tests/golden/make_golden_trajectory.py), regenerated from their generator seeds
    and checked against the fixture's checksums (a torch build whose CPU randn stream differs would fail here, not silently)."""
    bags = []
    for (n, seed), chk in zip(case["bag_shapes"].tolist(), case["bag_checks"]):
        b = torch.randn(1, n, 512, generator=torch.Generator().manual_seed(seed)).half()
        got = np.array([float(b.float().sum()), float(b.float().abs().sum()), float(b[0, -1, -1])])
        assert np.allclose(got, chk, rtol=0, atol=1e-6 * max(1.0, np.abs(chk).max())), "synthetic bag stream differs from the fixture's"
        bags.append(b)
    return bags


def trajectory_stable(case, name, floor=1e-7, rel=0.0):
    """Elements of parameter `name` whose Adam trajectory is a stable function of the gradient: every step's |g| >= floor and
    >= rel x the tensor's largest such value.  (The update is ~ lr * g / (|g| + eps): a relative error e of g moves the step by ~ e * lr,
    and on an element whose gradient is rounding noise the step is lr * sign(noise) -- attention_weights.bias, whose gradient is
    analytically zero under the softmax, walks differently in ANY two fp32 implementations.)"""
    m = case["mingrad." + name]
    return (m >= floor) & (m >= rel * m.max())
