"""The numpy student oracle vs golden vectors produced by the reference itself
(tests/golden/make_student_golden.py; reference train_and_eval.py:59-86,108-136, models.py:7-53)."""
import numpy as np
import pytest

from golden_inputs import CASES, Golden
from oracle import student_oracle as so
from parity_rules import check_eval_out, check_final_state, eval_loss_tol, eval_mean_tol, eval_tol, is_gauge, moment_tols

TOL = 1e-4   # north-star bar: 1e-4 abs, fp32


@pytest.mark.parametrize("name", CASES)
def test_single_step_grads(name):
    g = Golden(name)
    L = len(g.dims) - 1
    feats_l, labels_l = g.feats[g.idx_l], g.labels[g.idx_l]
    for kind, x, y in (("nll", feats_l, labels_l), ("kl", g.feats, g.out_t)):
        st = so.MLPState(g.sd0, L, g.norm, dropout_ratio=g.dropout)
        bsz = min(g.B, x.shape[0])
        lam = float(g.z[f"step_{kind}_lamb"])
        logits, cache = so.mlp_forward(st, x[:bsz], training=True, masks=g.masks(1, bsz))
        loss, dlogits = so.loss_and_dlogits(logits, y[:bsz], kind, lam)
        assert abs(float(loss) - float(g.z[f"step_{kind}_loss"])) < TOL
        np.testing.assert_allclose(g.view(logits), g.z[f"step_{kind}_logits"], atol=TOL, rtol=0)
        np.testing.assert_allclose(g.view(dlogits), g.z[f"step_{kind}_dlogits"], atol=1e-6, rtol=1e-4)
        grads = so.mlp_backward(st, cache, dlogits)
        for pname, gr in zip(g.param_names, grads):
            ref = g.z[f"step_{kind}_grad.{pname}"]
            np.testing.assert_allclose(g.view(gr), ref, atol=TOL, rtol=1e-4, err_msg=f"{kind} {pname}")
            nref = float(g.z[f"step_{kind}_gradnorm.{pname}"])
            assert abs(np.linalg.norm(gr.astype(np.float64)) - nref) <= 1e-4 * max(1.0, nref)


@pytest.mark.parametrize("name", CASES)
def test_distill_passes_and_eval(name):
    g = Golden(name)
    L = len(g.dims) - 1
    st = so.MLPState(g.sd0, L, g.norm, dropout_ratio=g.dropout)
    feats_l, labels_l = g.feats[g.idx_l], g.labels[g.idx_l]
    losses, means, pi = [], [], 0
    masks_fn = (lambda i, rows: g.masks(st.t + 1, rows)) if g.dropout > 0 else None     # st.t = optimiser steps taken so far
    for _ in range(g.epochs):          # reference train_and_eval.py:559-566: hard pass then soft pass
        m, ls = so.train_mini_batch(st, feats_l, labels_l, g.B, "nll", g.lamb, g.perms[pi], g.lr, g.wd, masks_fn)
        means.append(m); losses += ls; pi += 1
        m, ls = so.train_mini_batch(st, g.feats, g.out_t, g.B, "kl", 1 - g.lamb, g.perms[pi], g.lr, g.wd, masks_fn)
        means.append(m); losses += ls; pi += 1
    np.testing.assert_allclose(losses, g.z["step_losses"], atol=TOL, rtol=0)
    np.testing.assert_allclose(means, g.z["pass_means"], atol=TOL, rtol=0)
    assert st.t == int(g.z["adam.step"])
    check_final_state(g, st.state_dict())
    for pname, m, v in zip(g.param_names, st.m, st.v):
        if is_gauge(g, pname):
            continue
        am, av = moment_tols(g, pname)
        np.testing.assert_allclose(g.view(m), g.z[f"adam.exp_avg.{pname}"], atol=am, rtol=1e-3)
        np.testing.assert_allclose(g.view(v), g.z[f"adam.exp_avg_sq.{pname}"], atol=av, rtol=1e-3)
    out = so.evaluate_mini_batch(st, g.feats, g.B)
    check_eval_out(g, out)                     # |impl - ref_fp64| <= 2 x |ref_fp32 - ref_fp64| (tests/parity_rules.py)
    np.testing.assert_allclose(g.view(out), g.z["eval_out"], atol=eval_tol(g), rtol=0)
    assert np.abs(g.view(out) - g.z["eval_out"]).mean() <= eval_mean_tol(g)
    assert abs(so.nll_loss(out, g.labels) - float(g.z["eval_loss"])) < eval_loss_tol(g)
    assert abs(so.accuracy(out, g.labels) - float(g.z["eval_score"])) < (1e-6 if eval_tol(g) == TOL else 5e-3)
