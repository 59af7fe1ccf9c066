"""Training paths of ACMIL_MHA, the DTFD attention block and IBMIL: gradients against fixtures from the real reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _check_grads(model, case, rel=3e-3):
    for name, p in model.named_parameters():
        ref = case["grad." + name]
        assert p.grad is not None, name
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= rel * max(1e-3, np.abs(ref).max()), "%s: %.3e vs max %.3e" % (name, err, np.abs(ref).max())


def test_mha_gradients_match_reference():
    from acmil_amd.architecture.transformer import ACMIL_MHA
    case, sd = load_golden("train_mha_n400_d384_k3_c2")

    class Conf:
        D_feat, D_inner, n_class, n_token = 384, 128, 2, 3
    m = ACMIL_MHA(Conf, n_token=3, n_masked_patch=0, mask_drop=0.0)
    m.load_state_dict(sd); m = m.cuda().eval()          # eval + gradients: dropout inactive, as in the fixture (p = 0)
    sub, slide, attns = m(torch.from_numpy(case["x"]).cuda())
    label = torch.tensor([1]).cuda()
    loss = F.cross_entropy(sub, label.repeat(3)) + F.cross_entropy(slide, label) + 0.5 * attns.pow(2).mean()
    loss.backward()
    np.testing.assert_allclose(sub.detach().cpu().numpy(), case["sub_preds"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(attns.detach().cpu().numpy(), case["attns"], rtol=0, atol=1e-5)
    assert abs(loss.item() - float(case["loss"])) < 1e-4
    _check_grads(m, case)


def test_mha_train_mode_masks_and_dropout_run():
    from acmil_amd.architecture.transformer import ACMIL_MHA
    from oracle import mha_oracle as MO
    sd = MO.default_state_dict(384, 128, 2, 2, seed=1)

    class Conf:
        D_feat, D_inner, n_class, n_token = 384, 128, 2, 2
    m = ACMIL_MHA(Conf, n_token=2, n_masked_patch=10, mask_drop=0.6)
    m.load_state_dict(sd); m = m.cuda().train()
    sub, slide, attns = m(torch.randn(1, 300, 384, device="cuda"))
    assert attns.shape == (8, 2, 300) and int((attns == -1e9).sum()) == 8 * 2 * 6      # 6 of the top-10 masked per row
    (sub.sum() + slide.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_dtfd_block_gradients_match_reference():
    from acmil_amd.architecture.Attention import Attention_with_Classifier
    case, sd = load_golden("train_dtfd_n500_l256_k3_c4")
    m = Attention_with_Classifier(L=256, D=128, K=3, num_cls=4)
    m.load_state_dict(sd); m = m.cuda().train()
    pred = m(torch.from_numpy(case["x"]).cuda())
    loss = F.cross_entropy(pred, torch.tensor([0, 3, 1]).cuda())
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), case["pred"], rtol=0, atol=1e-4)
    _check_grads(m, case)


def test_ibmil_gradients_match_reference():
    from acmil_amd.architecture.ibmil import IBMIL
    case, sd = load_golden("train_ibmil_n600_d384_c2")

    class Conf:
        D_feat, D_inner, n_class, c_path = 384, 128, 2, None
    m = IBMIL(Conf)
    m.load_state_dict(sd); m = m.cuda().train()
    y, mm, a = m(torch.from_numpy(case["x"]).cuda())
    loss = F.cross_entropy(y, torch.tensor([1]).cuda()) + 0.01 * mm.sum() + 10.0 * (a * a).sum()
    loss.backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), case["Y_prob"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(a.detach().cpu().numpy(), case["A"], rtol=0, atol=1e-6)
    assert abs(loss.item() - float(case["loss"])) < 1e-4
    _check_grads(m, case)


@pytest.mark.parametrize("name,ncls,d,di,label", [("train_clam_bin_n600_d384_c2", 2, 384, 128, 1), ("train_clam_sub_n500_d512_c3", 3, 512, 256, 2)])
def test_clam_sb_training_step_matches_reference(name, ncls, d, di, label):
    """CLAM_SB training forward with the instance-level clustering loss (clam.py:159-197, :130-157) + backward, against the
    reference's own loss and gradients (dropout=False fixture; bag_weight 0.7 as in the CLAM trainer)."""
    import torch.nn as nn
    from acmil_amd.architecture.clam import CLAM_SB
    case, sd = load_golden(name)

    class Conf:
        D_feat, D_inner, n_class = d, di, ncls
    m = CLAM_SB(Conf, size_arg="small", k_sample=8, dropout=False, instance_loss_fn=nn.CrossEntropyLoss())
    m.load_state_dict(sd); m = m.cuda().train()
    y = torch.tensor([label]).cuda()
    logits, inst_loss = m(torch.from_numpy(case["x"]).cuda(), label=y, instance_eval=True)
    loss = 0.7 * F.cross_entropy(logits, y) + 0.3 * inst_loss
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), case["logits"], rtol=0, atol=1e-4)
    assert abs(float(inst_loss.detach()) - float(case["inst_loss"])) < 1e-4 and abs(float(loss.detach()) - float(case["loss"])) < 1e-4
    for pname, p in m.named_parameters():
        ref = case["grad." + pname]
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()       # unused instance classifiers: no gradient
        err = np.abs(got - ref).max()
        assert err <= 3e-3 * max(1e-3, np.abs(ref).max()), "%s: %.3e vs max %.3e" % (pname, err, np.abs(ref).max())


def test_clam_sb_dropout_configuration_trains():
    """dropout=True (the reference default): masks are drawn, every used parameter receives a finite gradient, eval() is unaffected."""
    from acmil_amd.architecture.clam import CLAM_SB

    class Conf:
        D_feat, D_inner, n_class = 384, 128, 2
    torch.manual_seed(0)
    m = CLAM_SB(Conf, size_arg="small", k_sample=8, dropout=True).cuda().train()
    x = torch.randn(1, 300, 384, device="cuda")
    y = torch.tensor([0]).cuda()
    logits, inst = m(x, label=y, instance_eval=True)
    (F.cross_entropy(logits, y) + inst).backward()
    used = [p for n, p in m.named_parameters() if not n.startswith("instance_classifiers.1")]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in used)
    m.eval()
    with torch.no_grad():
        l1, l2 = m(x), m(x)
    assert torch.equal(l1, l2)


@pytest.mark.parametrize("name,merge,learn", [("train_ibmil_conf_cat_learn_n700_d384_c3", "cat", True),
                                              ("train_ibmil_conf_sub_fixed_n700_d384_c3", "sub", False)])
def test_ibmil_confounder_branch_matches_reference(name, merge, learn, tmp_path):
    """IBMIL with the deconfounding stage (ibmil.py:45-67, :93-107): eval outputs and one forward + backward against the
    reference's fixture; the confounder dictionary is a learnable parameter (c_learn) or a buffer."""
    from acmil_amd.architecture.ibmil import IBMIL
    case, sd = load_golden(name)
    cpath = str(tmp_path / "conf.npy")
    np.save(cpath, sd["confounder_feat"].numpy())

    class Conf:
        D_feat, D_inner, n_class, c_path, c_learn = 384, 128, 3, [cpath], learn
    m = IBMIL(Conf, confounder_merge=merge)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd); m = m.cuda()
    x = torch.from_numpy(case["x"]).cuda()
    m.eval()
    with torch.no_grad():
        y, mm, da = m(x)
    np.testing.assert_allclose(y.cpu().numpy(), case["Y_prob"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(mm.cpu().numpy(), case["M"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(da.cpu().numpy(), case["deconf_A"], rtol=0, atol=1e-6)
    m.train()
    y, mm, da = m(x)
    loss = F.cross_entropy(y, torch.tensor([2]).cuda()) + 0.01 * mm.sum() + 3.0 * (da * da).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(case["loss"])) < 1e-4
    assert ("confounder_feat" in dict(m.named_parameters())) == learn
    _check_grads(m, case)


def test_dtfd_double_tier_step_matches_reference():
    """One slide through acmil_amd.dtfd.train_step against the REAL reference's train_one_epoch
    (Step3_WSI_classification_DTFD.py:61-160): same patch permutation, both losses, every module's gradient as backward left
    it (captured ahead of the reference's clip_grad_norm_), and the parameters after the two Adam steps."""
    import torch.nn as nn
    from acmil_amd import dtfd, train as T
    z = np.load("tests/golden/train_dtfd_step_n900_d384_c3.npz")
    conf = T.Struct(D_feat=384, D_inner=128, n_class=3, numGroup=4, total_instance=8, grad_clipping=5.0, lr=1e-3, wd=1e-5)
    mods = dict(zip(("classifier", "attention", "dimReduction", "attCls"), dtfd.build_dtfd(conf)))
    for name, m in mods.items():
        m.load_state_dict({k[len("before." + name) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("before." + name + ".")})
        m.cuda().train()
    grads = {}
    real_clip = nn.utils.clip_grad_norm_

    def spy(params, max_norm, *a, **k):
        params = list(params)
        for name, m in mods.items():
            own = list(m.parameters())
            if len(own) == len(params) and all(p is q for p, q in zip(own, params)):
                for (pn, _), p in zip(m.named_parameters(), params):
                    grads["grad.%s.%s" % (name, pn)] = p.grad.detach().cpu().numpy().copy()
        return real_clip(params, max_norm, *a, **k)
    nn.utils.clip_grad_norm_ = spy
    try:
        opt0, opt1 = dtfd.make_optimizers(mods["classifier"], mods["attention"], mods["dimReduction"], mods["attCls"], conf)
        l0, l1 = dtfd.train_step(mods["classifier"], mods["attention"], mods["dimReduction"], mods["attCls"],
                                 torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["label"]).cuda(), opt0, opt1, conf,
                                 perm=torch.from_numpy(z["perm"]).cuda())
    finally:
        nn.utils.clip_grad_norm_ = real_clip
    assert abs(float(l0) - float(z["loss0"])) < 1e-4 and abs(float(l1) - float(z["loss1"])) < 1e-4
    assert len(grads) == 17
    for k, g in grads.items():
        ref = z[k]
        assert np.abs(g - ref).max() <= 3e-3 * max(1e-3, np.abs(ref).max()), k
    # Adam's first step moves an element by lr * g / (|g| + 1e-8) ~ lr * sign(g): compare where |g| is well above that eps
    for name, m in mods.items():
        for pn, p in m.named_parameters():
            ref, g = z["after.%s.%s" % (name, pn)], z["grad.%s.%s" % (name, pn)]
            sure = np.abs(g) > 1e-6
            if sure.any():
                assert np.abs(p.detach().cpu().numpy() - ref)[sure].max() < 5e-5, (name, pn)


def test_dtfd_trainer_main_runs_and_predicts():
    from acmil_amd import dtfd
    best = dtfd.main(["--synthetic_slides", "8", "--synthetic_patches", "400", "--train_epoch", "1", "--n_class", "2"])
    assert set(best) == {"epoch", "val_auc", "val_f1"} and 0.0 <= best["val_auc"] <= 1.0
