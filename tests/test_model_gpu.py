"""GPU parity of the preserved Python surface (Model / train_mini_batch / evaluate_mini_batch /
SAGE.inference / GCN.forward / feature_prop) against the golden vectors produced by the reference
and against the CPU oracle.  Bar: 1e-4 abs fp32 (tests/parity_rules.py explains the Adam gauge cases)."""
import os

import numpy as np
import pytest
import torch

from golden_inputs import CASES, Golden
from graphgen import random_graph
from oracle import student_oracle as so
from oracle import teacher_oracle as to
from parity_rules import check_eval_out, check_final_state, eval_loss_tol, eval_mean_tol, eval_tol, has_gauge

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _student(g, dropout=0.0):
    from glnn_amd.models import Model
    L = len(g.dims) - 1
    conf = dict(model_name="MLP", num_layers=L, feat_dim=g.dims[0], hidden_dim=g.dims[1], label_dim=g.dims[-1],
                dropout_ratio=dropout, norm_type=g.norm, device=DEV)
    model = Model(conf)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in g.sd0.items()})
    opt = torch.optim.Adam(model.parameters(), lr=g.lr, weight_decay=g.wd)      # train_student.py:275-277
    return model, opt


def test_model_state_dict_keys_match_reference():
    from glnn_amd.models import Model
    g = Golden("bn_small")
    model, _ = _student(g)
    assert list(model.state_dict().keys()) == list(g.sd0.keys())
    sage = Model(dict(model_name="SAGE", num_layers=3, feat_dim=8, hidden_dim=16, label_dim=4, dropout_ratio=0.0,
                      norm_type="batch", device=DEV))
    keys = list(sage.state_dict().keys())
    assert "encoder.layers.0.fc_neigh.weight" in keys and "encoder.layers.2.fc_neigh.bias" in keys
    assert tuple(sage.state_dict()["encoder.layers.0.fc_neigh.weight"].shape) == (16, 8)
    gcn = Model(dict(model_name="GCN", num_layers=2, feat_dim=8, hidden_dim=16, label_dim=4, dropout_ratio=0.0,
                     norm_type="none", device=DEV))
    assert tuple(gcn.state_dict()["encoder.layers.0.weight"].shape) == (8, 16)     # dgl GraphConv: [in, out]
    with pytest.raises(NotImplementedError):
        Model(dict(model_name="GAT", num_layers=2, feat_dim=8, hidden_dim=16, label_dim=4, dropout_ratio=0.0,
                   norm_type="none", device=DEV, attn_dropout_ratio=0.1))


@pytest.mark.parametrize("name", CASES)
def test_single_step_gradients_vs_reference_golden(name):
    from glnn_amd import ops
    from glnn_amd.student import StudentEngine
    g = Golden(name)
    feats_l, labels_l = g.feats[g.idx_l], g.labels[g.idx_l]
    for kind, x, y in (("nll", feats_l, labels_l), ("kl", g.feats, g.out_t)):
        model, opt = _student(g, dropout=g.dropout)
        model.train()
        bsz = min(g.B, x.shape[0])
        eng = StudentEngine(model, opt, bsz)
        eng.base_seed = g.drop_base_seed          # dropout case: the masks the golden run fed to the reference's MLP
        lam = float(g.z[f"step_{kind}_lamb"])
        xd = ops.as_feat(torch.from_numpy(x).to(DEV))
        yd = torch.from_numpy(y).to(DEV)
        idx = torch.arange(bsz, device=DEV)
        eng.step(xd, idx, ops.LOSS_NLL if kind == "nll" else ops.LOSS_KL, yd if kind == "nll" else ops.as_feat(yd), lam)
        assert abs(eng.loss_out.item() - float(g.z[f"step_{kind}_loss"])) < TOL
        np.testing.assert_allclose(g.view(eng.logits[:bsz].cpu().numpy()), g.z[f"step_{kind}_logits"], atol=TOL, rtol=0)
        np.testing.assert_allclose(g.view(eng.dlogits[:bsz].cpu().numpy()), g.z[f"step_{kind}_dlogits"], atol=1e-6, rtol=1e-4)
        for pname, p in model.named_parameters():
            np.testing.assert_allclose(g.view(p.grad.cpu().numpy()), g.z[f"step_{kind}_grad.{pname}"], atol=TOL, rtol=1e-4,
                                       err_msg=f"{kind} {pname}")


@pytest.mark.parametrize("name", CASES)
def test_distill_passes_vs_reference_golden(name, monkeypatch):
    """The reference's epoch body (train_and_eval.py:559-566) through glnn_amd.train_and_eval with the
    reference's own criterion / optimizer objects; permutations replayed from the golden run."""
    from glnn_amd import train_and_eval as te
    g = Golden(name)
    torch.manual_seed(g.drop_base_seed)            # the engine seeds its dropout stream from torch.initial_seed()
    model, opt = _student(g, dropout=g.dropout)
    criterion_l = torch.nn.NLLLoss()
    criterion_t = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    perms = iter(g.perms)
    monkeypatch.setattr(torch, "randperm", lambda n, *a, **k: torch.from_numpy(next(perms)))
    feats, labels, out_t = (torch.from_numpy(a).to(DEV) for a in (g.feats, g.labels, g.out_t))
    idx_l = torch.from_numpy(g.idx_l).to(DEV)
    feats_l, labels_l = feats[idx_l], labels[idx_l]
    means = []
    for _ in range(g.epochs):
        means.append(te.train_mini_batch(model, feats_l, labels_l, g.B, criterion_l, opt, g.lamb))
        means.append(te.train_mini_batch(model, feats, out_t, g.B, criterion_t, opt, 1 - g.lamb))
    np.testing.assert_allclose(means, g.z["pass_means"], atol=TOL, rtol=0)
    assert int(opt.state[next(model.parameters())]["step"]) == int(g.z["adam.step"])
    check_final_state(g, {k: v.cpu().numpy() for k, v in model.state_dict().items()})
    evaluator = lambda o, y: o.argmax(1).eq(y).float().mean().item()
    out, loss_e, score_e = te.evaluate_mini_batch(model, feats, labels, criterion_l, g.B, evaluator)
    check_eval_out(g, out.cpu().numpy())                     # |impl - ref_fp64| <= 2 x |ref_fp32 - ref_fp64| (tests/parity_rules.py)
    np.testing.assert_allclose(g.view(out.cpu().numpy()), g.z["eval_out"], atol=eval_tol(g), rtol=0)
    assert np.abs(g.view(out.cpu().numpy()) - g.z["eval_out"]).mean() <= eval_mean_tol(g)
    assert abs(loss_e - float(g.z["eval_loss"])) < eval_loss_tol(g)
    assert abs(score_e - float(g.z["eval_score"])) < (1e-6 if not has_gauge(g) else 5e-3)


def test_eval_forward_at_identical_state_vs_oracle():
    """Eval-mode forward (BN running stats) at a FIXED state: 1e-4 even for the gauge configs."""
    from glnn_amd import train_and_eval as te
    g = Golden("arxiv_dims")
    model, _ = _student(g, dropout=0.5)           # dropout must be a no-op in eval mode
    st = so.MLPState(g.sd0, len(g.dims) - 1, g.norm)
    want = so.evaluate_mini_batch(st, g.feats, g.B)
    out, _, _ = te.evaluate_mini_batch(model, torch.from_numpy(g.feats).to(DEV), torch.from_numpy(g.labels).to(DEV),
                                       torch.nn.NLLLoss(), g.B, lambda o, y: 0.0)
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=TOL, rtol=0)
    model.eval()
    with torch.no_grad():
        h_list, h = model.forward_fitnet(None, torch.from_numpy(g.feats).to(DEV))     # (h_list, logits), models.py:414-423
    _, cache = so.mlp_forward(st, g.feats, training=False)
    assert len(h_list) == len(g.dims) - 2
    for got_z, want_z in zip(h_list, cache["z"]):
        np.testing.assert_allclose(got_z.cpu().numpy(), want_z, atol=TOL, rtol=0)


@pytest.mark.parametrize("norm", ["batch", "none"])
def test_eval_chain_logits_only_form_equals_the_form_that_keeps_hidden_outputs(norm):
    """Model.forward in eval mode folds BN(eval)+ReLU into the producing GEMM's epilogue (next GEMM reads a plain operand);
    MLP.forward / forward_fitnet keep the raw hidden Linear outputs and fold them into the next operand load: same logits."""
    from glnn_amd import ops
    from glnn_amd.models import Model
    torch.manual_seed(2)
    model = Model(dict(model_name="MLP", num_layers=3, feat_dim=100, hidden_dim=256, label_dim=47, dropout_ratio=0.3, norm_type=norm, device=DEV))
    with torch.no_grad():
        for bn in model.encoder.norms:
            bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    model.eval()
    x = ops.as_feat(torch.randn(3001, 100, device=DEV))
    a = model(None, x)
    h_list, b = model.forward_fitnet(None, x)
    assert len(h_list) == 2 and a.shape == b.shape == (3001, 47)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5, rtol=1e-5)
    # and against plain torch ops on the same parameters
    enc = model.encoder
    h = x[:, :100]
    for l, layer in enumerate(enc.layers):
        h = torch.nn.functional.linear(h, layer.weight, layer.bias)
        if l < 2:
            if norm == "batch":
                h = torch.nn.functional.batch_norm(h, enc.norms[l].running_mean, enc.norms[l].running_var, enc.norms[l].weight, enc.norms[l].bias, False, 0.1, enc.norms[l].eps)
            h = torch.relu(h)
    np.testing.assert_allclose(a.cpu().numpy(), h.detach().cpu().numpy(), atol=TOL, rtol=1e-4)


@pytest.mark.parametrize("rows,h,p,seed", [(32, 48, 0.4, 0x00C0FFEE), (4096, 2048, 0.2, 12345), (513, 257, 0.5, 0xFFFFFFFF), (1, 1, 0.9, 7)])
def test_dropout_mask_kernel_equals_numpy_restatement(rows, h, p, seed):
    """glnn_dropout_mask_u8 (what every fused kernel evaluates on the fly) == oracle/dropout_mask.py, bit for bit: the
    restatement is what fed the reference's MLP when the dropout golden was generated."""
    from glnn_amd import ops
    from oracle.dropout_mask import keep_mask
    got = ops.dropout_mask(rows, h, p, seed, DEV).cpu().numpy()
    assert np.array_equal(got, keep_mask(rows, h, p, seed))


@pytest.mark.parametrize("materialize", ["0", "1"])
def test_training_step_with_dropout_matches_oracle_given_the_mask(materialize, monkeypatch):
    """Both forms of a hidden layer's tail: recomputed inside the operand loads ("0") and written once by glnn_act_fwd_f32 ("1")."""
    from glnn_amd import ops
    from glnn_amd.student import StudentEngine
    monkeypatch.setenv("GLNN_STUDENT_MATERIALIZE_ACT", materialize)
    monkeypatch.setenv("GLNN_STUDENT_PREGATHER", materialize)      # likewise feats[idx]: gathered in the operand loads / copied once
    g = Golden("bn_small")
    p = 0.4
    model, opt = _student(g, dropout=p)
    model.train()
    L = len(g.dims) - 1
    bsz = 64
    eng = StudentEngine(model, opt, bsz)
    x = ops.as_feat(torch.from_numpy(g.feats).to(DEV))
    idx = torch.arange(100, 100 + bsz, device=DEV)
    # the engine derives its dropout seeds from (base_seed, step_count); recompute those of step 1
    eng.step_count = 1
    seeds = [eng._seed(l) for l in range(L - 1)]
    eng.step_count = 0
    eng.step(x, idx, ops.LOSS_KL, ops.as_feat(torch.from_numpy(g.out_t).to(DEV)), 0.7)
    masks = [ops.dropout_mask(bsz, g.dims[l + 1], p, seeds[l], DEV).cpu().numpy().astype(np.float32) for l in range(L - 1)]
    for mk in masks:
        assert abs(mk.mean() - (1 - p)) < 0.05
    st = so.MLPState(g.sd0, L, g.norm, dropout_ratio=p)
    rows = idx.cpu().numpy()
    logits, cache = so.mlp_forward(st, g.feats[rows], training=True, masks=masks)
    loss, dlogits = so.loss_and_dlogits(logits, g.out_t[rows], "kl", 0.7)
    grads = so.mlp_backward(st, cache, dlogits)
    assert abs(eng.loss_out.item() - float(loss)) < TOL
    np.testing.assert_allclose(eng.logits[:bsz].cpu().numpy(), logits, atol=TOL, rtol=0)
    for (pname, prm), gr in zip(model.named_parameters(), grads):
        np.testing.assert_allclose(prm.grad.cpu().numpy(), gr, atol=TOL, rtol=1e-4, err_msg=pname)


@pytest.mark.parametrize("dims,bsz,norm,p", [([100, 512, 512, 47], 4096, "batch", 0.2), ([24, 64, 64, 5], 300, "none", 0.5),
                                             ([128, 1024, 1024, 40], 512, "batch", 0.5)])
def test_materialised_activation_steps_equal_recomputed_ones_bit_for_bit(dims, bsz, norm, p, monkeypatch):
    """glnn_mlp_step_desc.act: the stored tail and the one re-evaluated in the GEMM operand loads are the same fp32 expression
    on the same counter-based mask, so three optimiser steps end in identical parameters, moments and running statistics --
    as long as both run the same GEMM kernels: the pipelined kernels (plain operands only) are switched off here, because the
    weight-gradient launcher gives them other reduction splits than the operand-transform kernels get (another summation order);
    likewise the classifier's weight gradient out of the BatchNorm backward's first pass (bn_bwd_partial_wg_sk), which only the
    re-evaluating form takes (it has act(z) on chip; a stored tail goes through gemm_tn)."""
    monkeypatch.setenv("GLNN_GEMM_PIPE", "0")
    monkeypatch.setenv("GLNN_STUDENT_NARROW_WGRAD", "0")
    import copy
    from glnn_amd import ops
    from glnn_amd.models import Model
    from glnn_amd.student import StudentEngine
    torch.manual_seed(5)
    base = Model(dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                      dropout_ratio=p, norm_type=norm, device=DEV))
    x = ops.as_feat(torch.randn(2 * bsz, dims[0], device=DEV))
    tgt = ops.as_feat(torch.log_softmax(torch.randn(2 * bsz, dims[-1], device=DEV), 1))
    states = []
    for mode in ("0", "1"):
        monkeypatch.setenv("GLNN_STUDENT_MATERIALIZE_ACT", mode)
        model = copy.deepcopy(base)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
        eng = StudentEngine(model, opt, bsz)
        assert all((a is not None) == (mode == "1") for a in eng.act)
        for i in range(3):
            eng.step(x, torch.arange(i * 7, i * 7 + bsz, device=DEV), ops.LOSS_KL, tgt, 1.0)
        torch.cuda.synchronize()
        states.append(([t.detach().clone() for t in model.state_dict().values()],
                       [opt.state[q]["exp_avg"].clone() for q in model.parameters()], eng.loss_out.clone()))
    for a, b in zip(states[0][0] + states[0][1] + [states[0][2]], states[1][0] + states[1][1] + [states[1][2]]):
        assert torch.equal(a, b)


def _variant_run(base, dims, bsz, x, tgt, kind, steps, lr=0.01):
    """`steps` optimiser steps of a fresh copy of `base` under the current environment -> (state, moments, grads, loss, logits)."""
    import copy
    from glnn_amd import ops
    from glnn_amd.student import StudentEngine
    model = copy.deepcopy(base)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=5e-4)
    eng = StudentEngine(model, opt, bsz)
    for i in range(steps):
        eng.step(x, torch.arange(i * 7, i * 7 + bsz, device=DEV), kind, tgt, 0.7)
    torch.cuda.synchronize()
    if eng.sync_counters is not None:
        assert int(eng.sync_counters.abs().sum()) == 0            # every counter protocol returned its counters to zero
    return ([t.detach().clone() for t in model.state_dict().values()], [opt.state[q]["exp_avg"].clone() for q in model.parameters()],
            [g.clone() for g in eng.grads], eng.loss_out.clone(), eng.logits[:bsz].clone())


def _variant_inputs(dims, bsz, norm, p, kind, seed):
    from glnn_amd import ops
    from glnn_amd.models import Model
    torch.manual_seed(seed)
    base = Model(dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                      dropout_ratio=p, norm_type=norm, device=DEV))
    x = ops.as_feat(torch.randn(2 * bsz, dims[0], device=DEV))
    if kind == "nll":
        return base, x, torch.randint(0, dims[-1], (2 * bsz,), device=DEV), ops.LOSS_NLL
    return base, x, ops.as_feat(torch.log_softmax(torch.randn(2 * bsz, dims[-1], device=DEV), 1)), ops.LOSS_KL


SMALL_STEP_CASES = [([128, 256, 256, 40], 512, "batch", 0.2, "kl"), ([100, 72, 72, 47], 300, "batch", 0.5, "nll"),
                    ([24, 64, 64, 5], 77, "none", 0.0, "kl"), ([128, 1024, 1024, 40], 512, "batch", 0.5, "kl"),
                    ([1433, 128, 7], 140, "none", 0.6, "kl"),        # cora: W0's rows are not float4-addressable (gemm_lat_kernel<BU>)
                    ([100, 256, 256, 47], 4096, "batch", 0.5, "kl")]


@pytest.mark.parametrize("dims,bsz,norm,p,kind", SMALL_STEP_CASES)
def test_one_call_train_step_equals_fwd_bwd_plus_adam_bit_for_bit(dims, bsz, norm, p, kind, monkeypatch):
    """glnn_mlp_train_step_f32 (ABI 6) leaves the last sums of the gradient partials -- split-K slabs of the batched weight gradients,
    per-chunk column sums of the BatchNorm backward, the loss kernel's per-workgroup loss / bias-gradient partials -- to the Adam launch.
    Same partials, same order: three steps end in the same parameters, moments, running statistics, gradients and loss as
    glnn_mlp_fwd_bwd_f32 + glnn_adam_step_f32, bit for bit (and with GLNN_STUDENT_ADAM_FOLDS=0, the one-call form without the folds)."""
    base, x, tgt, k = _variant_inputs(dims, bsz, norm, p, kind, 21)
    runs = []
    # the one-call form also splits the latency weight-gradient kernel's reduction over workgroups (Adam folds the slabs): another
    # summation order, compared to fp32 rounding in test_small_batch_step_forms_agree; switched off here
    monkeypatch.setenv("GLNN_GEMM_TN_LAT_SPLITS", "1")
    monkeypatch.setenv("GLNN_STUDENT_FUSE_APPLY", "0")        # (the first hidden layer's apply pass inside the weight-gradient launch: one-call form only)
    for one_call, folds in (("0", "1"), ("1", "1"), ("1", "0")):
        monkeypatch.setenv("GLNN_STUDENT_ONE_CALL", one_call)
        monkeypatch.setenv("GLNN_STUDENT_ADAM_FOLDS", folds)
        runs.append(_variant_run(base, dims, bsz, x, tgt, k, 3))
    for other in runs[1:]:
        for a, b in zip(runs[0][0] + runs[0][1] + runs[0][2] + [runs[0][3]], other[0] + other[1] + other[2] + [other[3]]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("seed", [0, 1])
def test_student_step_vs_float64_torch_on_random_shapes(seed):
    """scripts/fuzz_vs_torch.py: 40 random students per seed (1-3 layers, widths 7 ... 1433, 2 ... 4096 rows, BatchNorm / none, NLL / KL, lamb,
    dropout through the engine's own masks) -- loss, logits and every gradient of one step against torch autograd in float64 on the
    same module, 1e-4 of the largest gradient (observed <= 4e-5).  (It found the one-layer student with more than 64 classes getting no
    bias gradient from the batched weight-gradient form.)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_vs_torch
    bad = [(d, e) for d, e in fuzz_vs_torch.run(seed, 40, verbose=False) if not e < 1e-4]
    assert not bad, bad


@pytest.mark.parametrize("seed", [0, 1])
def test_small_step_forms_agree_on_random_shapes(seed):
    """scripts/fuzz_small_step.py: 40 random (layers, widths, batch, norm, dropout, loss, weight decay) per seed -- odd widths (7, 50, 130,
    257, 1433: rows that are not float4-addressable), partial tiles (1 ... 1100 rows), 1-3 layers -- default forms (latency kernels, one-call
    step, Adam folds) against the tiled two-call forms: logits, loss, every gradient.  (It found the one-layer student with an unaligned
    weight registering Adam folds for a launch that then did not happen.)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_small_step
    bad = [(d, e) for d, e, ok in fuzz_small_step.run(seed, 40, verbose=False) if not ok or e >= 2e-3]
    assert not bad, bad


@pytest.mark.parametrize("knob,modes", [("GLNN_GEMM_LAT", "01"), ("GLNN_STUDENT_DEFER_STATS", "01"), ("GLNN_STUDENT_SLAB_CONSUMERS", "01"),
                                        ("GLNN_GEMM_TN_LAT", "01"), ("GLNN_GEMM_TN_LAT_SPLITS", "18"), ("GLNN_STUDENT_ONE_CALL", "01"),
                                        ("GLNN_STUDENT_LAT_BN_BWD", "01"), ("GLNN_STUDENT_FUSE_APPLY", "01")])
@pytest.mark.parametrize("dims,bsz,norm,p,kind", SMALL_STEP_CASES[:5])
def test_small_batch_step_forms_agree(dims, bsz, norm, p, kind, knob, modes, monkeypatch):
    """The latency forms of the B <= 1024 step against the forms they replace, one optimiser step from the same state:
      GLNN_GEMM_LAT=0              tiled GEMMs + separate statistics / loss launches instead of mlp_lat.hip (K split over the four waves
                                   of a workgroup, statistics / loss as epilogue),
      GLNN_STUDENT_DEFER_STATS=0   statistics finished by the last workgroup of the producing launch instead of in the consumer's prologue,
      GLNN_STUDENT_SLAB_CONSUMERS=0  split-K partials folded by a launch instead of by the statistics / BatchNorm-backward kernel,
      GLNN_GEMM_TN_LAT=0           the batched 64 x 64 weight-gradient kernel + fold instead of gemm_tn_lat_kernel,
      GLNN_GEMM_TN_LAT_SPLITS=1|8  its reduction kept inside one workgroup or split over up to 8 (the one-call form, Adam folds),
      GLNN_STUDENT_ONE_CALL=0      glnn_mlp_fwd_bwd_f32 + glnn_adam_step_f32 instead of glnn_mlp_train_step_f32,
      GLNN_STUDENT_FUSE_APPLY=0    the first hidden layer's BatchNorm apply as a launch of its own instead of in the operand loads of the
                                   weight-gradient launch (TnProblem::bn_z),
      GLNN_STUDENT_LAT_BN_BWD=0    input-gradient GEMM + one-launch BatchNorm backward (workgroups wait for each other) instead of the
                                   GEMM whose epilogue leaves the column partial sums + bn_apply_tiles_kernel (glnn::lat_dgrad_bn_bwd).
    They differ by summation order only: logits, loss and every gradient agree to fp32 rounding (partial tiles: 300 / 77 rows, 72 and 5
    columns, K = 100 and 24)."""
    base, x, tgt, k = _variant_inputs(dims, bsz, norm, p, kind, 33)
    runs = []
    for mode in modes:
        monkeypatch.setenv(knob, mode)
        runs.append(_variant_run(base, dims, bsz, x, tgt, k, 1))
    (_, _, g0, l0, z0), (_, _, g1, l1, z1) = runs
    assert abs(float(l0) - float(l1)) <= 2e-6 * max(1.0, abs(float(l0)))
    torch.testing.assert_close(z1, z0, atol=2e-5, rtol=1e-5)
    for a, b in zip(g0, g1):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (tuple(a.shape), float((a - b).abs().max()), scale)


@pytest.mark.parametrize("dims,bsz,p,kind", [([100, 256, 256, 47], 4096, 0.5, "kl"), ([100, 2048, 2048, 47], 4096, 0.2, "kl"),
                                             ([50, 72, 7], 1100, 0.0, "nll"), ([24, 260, 260, 64], 1500, 0.3, "kl"),
                                             ([130, 128, 2], 2049, 0.5, "nll"), ([64, 512, 40], 1301, 0.2, "kl"),
                                             ([100, 256, 64, 47], 2048, 0.2, "kl")])
@pytest.mark.parametrize("wgrad", ["1", "0"])
def test_classifier_input_gradient_recomputed_in_the_batchnorm_backward(dims, bsz, p, kind, wgrad, monkeypatch):
    """Large batches in front of a NARROW last layer: the input gradient da = dlogits . W is never written -- both passes of the BatchNorm
    backward recompute their tile of it on the matrix cores (student.hip bn_bwd_partial_sk / bn_bwd_apply_sk) -- against the form that
    writes it with a GEMM (GLNN_STUDENT_NARROW_BWD=0).  Different summation orders only: loss identical (the forward is untouched),
    every gradient to fp32 rounding; ragged shapes (rows % 128, hidden % 64, 2 ... 64 classes), with and without dropout.
    wgrad = 1: the first pass also leaves the classifier's own weight / bias gradient as row-chunk partials (bn_bwd_partial_wg_sk)
    instead of a gemm_tn launch.  [100, 256, 64, 47]: a narrow HIDDEN layer as well -- its dz lives in the buffer the pass writes, so
    only the last layer may take the recomputing form."""
    monkeypatch.setenv("GLNN_STUDENT_NARROW_BWD_MIN", "1")
    monkeypatch.setenv("GLNN_STUDENT_NARROW_WGRAD", wgrad)
    base, x, tgt, k = _variant_inputs(dims, bsz, "batch", p, kind, 35)
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("GLNN_STUDENT_NARROW_BWD", mode)
        runs.append(_variant_run(base, dims, bsz, x, tgt, k, 1))
    (_, _, g0, l0, z0), (_, _, g1, l1, z1) = runs
    assert float(l0) == float(l1) and torch.equal(z0, z1)
    gs = max(float(a.abs().max()) for a in g0)
    assert any(not torch.equal(a, b) for a, b in zip(g0, g1)), "both runs took the same path"
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 2e-6 * gs, (tuple(a.shape), float((a - b).abs().max()), gs)


@pytest.mark.parametrize("dims,bsz,p,kind", [([100, 2048, 2048, 47], 4096, 0.2, "kl"), ([100, 256, 256, 47], 4096, 0.5, "kl"),
                                             ([72, 544, 544, 40], 2500, 0.3, "kl"), ([100, 608, 608, 7], 2049, 0.0, "nll"),
                                             ([100, 512, 512, 512, 47], 4096, 0.2, "kl")])      # (feature width > 64: the weight gradient's pipelined kernel)
def test_first_hidden_layer_batchnorm_backward_out_of_the_input_gradient_product(dims, bsz, p, kind, monkeypatch):
    """Large batches (round 6): the first hidden layer's BatchNorm backward has no passes of its own -- the input-gradient product's epilogue
    stores dy and the tile column sums (gemm.hip pipe_tile_bn_dy), one launch makes the constants, the first layer's weight gradient
    applies dz = alpha dy + beta z + gamma in its operand loads -- against partial + apply behind a plain product
    (GLNN_STUDENT_BN0_IN_GEMM=0).  Different summation orders only: forward identical, every gradient to fp32 rounding (the bias
    gradient in front of the BatchNorm, mathematically 0, is exactly 0 in the new form and rounding noise in the old one); ragged rows,
    hidden widths off the 128-column tile grid, with / without dropout, a four-layer student."""
    base, x, tgt, k = _variant_inputs(dims, bsz, "batch", p, kind, 37)
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("GLNN_STUDENT_BN0_IN_GEMM", mode)
        runs.append(_variant_run(base, dims, bsz, x, tgt, k, 1))
    (_, _, g0, l0, z0), (_, _, g1, l1, z1) = runs
    assert float(l0) == float(l1) and torch.equal(z0, z1)
    gs = max(float(a.abs().max()) for a in g0)
    assert any(not torch.equal(a, b) for a, b in zip(g0, g1)), "both runs took the same path"
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 2e-6 * gs, (tuple(a.shape), float((a - b).abs().max()), gs)


@pytest.mark.parametrize("dims,bsz,p", [([100, 2048, 2048, 47], 4096, 0.2), ([72, 544, 544, 40], 2500, 0.3)])
def test_deferred_batchnorm_constants_made_in_the_weight_gradient_prologue_are_the_same_bits(dims, bsz, p, monkeypatch):
    """... and the constants of that deferred apply (alpha, beta, gamma per column from the folded tile sums) are made in the prologue of the
    weight-gradient product itself (pipe_mainloop<.., AX>, PipeAx::p1) instead of by a launch in front of it
    (GLNN_STUDENT_BN0_CONSTS_IN_GEMM=0: bn_bwd_parts_consts_kernel): the same sums in the same order, the same expressions -- every
    gradient, the loss and the logits bit for bit."""
    base, x, tgt, k = _variant_inputs(dims, bsz, "batch", p, "kl", 39)
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("GLNN_STUDENT_BN0_CONSTS_IN_GEMM", mode)
        runs.append(_variant_run(base, dims, bsz, x, tgt, k, 2))
    (s0, m0, g0, l0, z0), (s1, m1, g1, l1, z1) = runs
    for a, b in zip(s0 + m0 + g0 + [l0, z0], s1 + m1 + g1 + [l1, z1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dims,bsz,norm,p", [([1433, 128, 7], 140, "none", 0.6), ([3703, 128, 6], 512, "none", 0.6), ([4814, 64, 64, 2], 300, "batch", 0.2),
                                             ([1433, 256, 256, 40], 4096, "batch", 0.5)])
def test_wide_unaligned_first_layer_through_a_padded_shadow_of_its_weight(dims, bsz, norm, p, monkeypatch):
    """Feature rows that are not float4-addressable and >= 1024 wide (cora 1433, citeseer 3703, penn94 4814): the step copies W_0 into a
    padded shadow (tail of ws_gemm, one small launch) so that the first layer's product runs on the tiled split-K kernels instead of the
    unaligned-W latency kernel (GLNN_STUDENT_PAD_W0=0).  Other summation order: logits, loss and gradients to fp32 rounding."""
    base, x, tgt, k = _variant_inputs(dims, bsz, norm, p, "kl", 43)
    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("GLNN_STUDENT_PAD_W0", mode)
        runs.append(_variant_run(base, dims, bsz, x, tgt, k, 1))
    (_, _, g0, l0, z0), (_, _, g1, l1, z1) = runs
    assert not torch.equal(z0, z1), "both runs took the same path"
    assert abs(float(l0) - float(l1)) <= 2e-6 * max(1.0, abs(float(l0)))
    torch.testing.assert_close(z1, z0, atol=3e-5, rtol=1e-5)
    gs = max(float(a.abs().max()) for a in g0)
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 2e-5 * gs + 1e-7, (tuple(a.shape), float((a - b).abs().max()), gs)


@pytest.mark.parametrize("dims,bsz,norm", [([100, 512, 512, 7], 6754, "batch"), ([100, 512, 512, 70], 2580, "none"), ([4814, 256, 256, 2], 512, "none")])
def test_large_step_with_a_cramped_weight_gradient_workspace(dims, bsz, norm):
    """In the one-call step every layer's split-reduction slabs wait in ws_tn for the Adam launch to fold them.  The Python mirror sizes
    ws_tn for all of them; a host that gives less must not end with a later layer running UNSPLIT (vk_class: four workgroups for 60 us):
    a product that finds ws_tn cramped takes the idle ws_gemm and folds at once.  Same arithmetic, other split plans: gradients to fp32
    rounding of the largest one.  (The shapes are the reference's vk_class / house_class-sized / penn94 students.)"""
    import copy
    from glnn_amd import ops
    from glnn_amd.student import StudentEngine
    base, x, tgt, k = _variant_inputs(dims, bsz, norm, 0.0, "kl", 41)
    grads = []
    for shrink in (False, True):
        model = copy.deepcopy(base)
        model.train()
        eng = StudentEngine(model, torch.optim.Adam(model.parameters(), lr=0.01), bsz)
        if shrink:
            full = int(eng.desc.ws_tn_floats)
            eng.desc.ws_tn_floats = min(full, 64 * max(dims) + 5 * dims[1] * dims[2] // 4 + dims[1] * dims[0] // 2)   # room for ~1 slab of the middle layer
            assert int(eng.desc.ws_tn_floats) < full
        eng.step(x, torch.arange(bsz, device=DEV), k, tgt, 1.0)
        torch.cuda.synchronize()
        grads.append([g.clone() for g in eng.grads] + [eng.loss_out.clone()])
    gs = max(float(a.abs().max()) for a in grads[0][:-1])
    assert float(grads[0][-1]) == float(grads[1][-1])
    for a, b in zip(grads[0][:-1], grads[1][:-1]):
        assert float((a - b).abs().max()) <= 2e-6 * gs, (tuple(a.shape), float((a - b).abs().max()), gs)


@pytest.mark.parametrize("dims,bsz,p", [([128, 256, 256, 40], 512, 0.2), ([100, 256, 256, 47], 4096, 0.5), ([24, 64, 64, 5], 300, 0.0),
                                        ([128, 1024, 1024, 40], 512, 0.5)])
def test_one_launch_batchnorm_backward_equals_the_two_launch_form_bit_for_bit(dims, bsz, p, monkeypatch):
    """bn_bwd_fused (partial sums -> wait for the column block's other row chunks -> apply, one launch for co-resident grids)
    performs the sums and the per-element arithmetic of bn_bwd_partial + bn_bwd_apply in the same order: three optimiser steps
    end in identical parameters, moments, running statistics and loss.  Run twice in a row: the counters must come back to 0."""
    import copy
    from glnn_amd import ops
    from glnn_amd.models import Model
    from glnn_amd.student import StudentEngine
    torch.manual_seed(9)
    base = Model(dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                      dropout_ratio=p, norm_type="batch", device=DEV))
    x = ops.as_feat(torch.randn(2 * bsz, dims[0], device=DEV))
    tgt = ops.as_feat(torch.log_softmax(torch.randn(2 * bsz, dims[-1], device=DEV), 1))
    states = []
    for mode in ("0", "1"):
        monkeypatch.setenv("GLNN_BN_BWD_ONE_LAUNCH", mode)
        model = copy.deepcopy(base)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
        eng = StudentEngine(model, opt, bsz)
        assert eng.sync_counters is not None
        for i in range(3):
            eng.step(x, torch.arange(i * 7, i * 7 + bsz, device=DEV), ops.LOSS_KL, tgt, 1.0)
        torch.cuda.synchronize()
        assert int(eng.sync_counters.abs().sum()) == 0
        states.append(([t.detach().clone() for t in model.state_dict().values()],
                       [opt.state[q]["exp_avg"].clone() for q in model.parameters()], eng.loss_out.clone()))
    for a, b in zip(states[0][0] + states[0][1] + [states[0][2]], states[1][0] + states[1][1] + [states[1][2]]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------- teacher
def _sage_model(dims, norm, seed):
    from glnn_amd.models import Model
    L = len(dims) - 1
    torch.manual_seed(seed)
    model = Model(dict(model_name="SAGE", num_layers=L, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                       dropout_ratio=0.5, norm_type=norm, device=DEV))
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for i, bn in enumerate(model.encoder.norms):
            h = bn.weight.shape[0]
            bn.weight.copy_(torch.from_numpy(rs.uniform(.5, 1.5, h).astype(np.float32)))
            bn.bias.copy_(torch.from_numpy(rs.uniform(-.2, .2, h).astype(np.float32)))
            bn.running_mean.copy_(torch.from_numpy(rs.uniform(-.3, .3, h).astype(np.float32)))
            bn.running_var.copy_(torch.from_numpy(rs.uniform(.5, 1.5, h).astype(np.float32)))
        for lay in model.encoder.layers:
            lay.fc_neigh.bias.copy_(torch.from_numpy((rs.standard_normal(lay.fc_neigh.bias.shape[0]) * .1).astype(np.float32)))
    model.eval()
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    layers = [dict(weight=sd[f"encoder.layers.{i}.fc_neigh.weight"], bias=sd[f"encoder.layers.{i}.fc_neigh.bias"]) for i in range(L)]
    norms = [dict(weight=sd[f"encoder.norms.{i}.weight"], bias=sd[f"encoder.norms.{i}.bias"],
                  running_mean=sd[f"encoder.norms.{i}.running_mean"], running_var=sd[f"encoder.norms.{i}.running_var"])
             for i in range(L - 1)] if norm == "batch" else None
    return model, layers, norms


@pytest.mark.parametrize("dims,norm", [([128, 256, 256, 40], "batch"), ([100, 256, 256, 47], "batch"), ([20, 32, 6], "none")])
def test_sage_inference_vs_oracle(dims, norm):
    """SAGE.inference (reference models.py:121-148) through Model.inference: whole-graph fast path AND the
    reference's chunked sweep, both against the CPU oracle."""
    from glnn_amd.graph import CSRGraph, FullNeighborLoader
    n = 4000
    indptr, indices = random_graph(n, 14, seed=dims[0], power=0.6, isolated=9, hub=1500)
    x = np.random.RandomState(0).standard_normal((n, dims[0])).astype(np.float32)
    model, layers, norms = _sage_model(dims, norm, seed=1)
    want = to.sage_inference(indptr, indices, x, layers, norms)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    loader = FullNeighborLoader(g, 512)
    got = model.inference(loader, torch.from_numpy(x).to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)
    got_chunked = model.encoder.inference(loader, torch.from_numpy(x).to(DEV), whole_graph=False)
    np.testing.assert_allclose(got_chunked.cpu().numpy(), want, atol=TOL, rtol=0)
    assert loader.global_blocks is False       # (the engine mode of the sweep -- global-id row-range blocks, no gather / scatter -- is switched back off)

    class Plain:      # a loader WITHOUT the engine mode: the literal loop of models.py:133-145 (block build, feats[input_nodes], conv, y[output_nodes] = h)
        def __init__(self, inner):
            self.inner = inner

        def __iter__(self):
            return iter(self.inner)
    got_literal = model.encoder.inference(Plain(loader), torch.from_numpy(x).to(DEV), whole_graph=False)
    np.testing.assert_allclose(got_literal.cpu().numpy(), want, atol=TOL, rtol=0)
    assert float((got_literal - got_chunked).abs().max()) <= 2e-5


@pytest.mark.parametrize("dims", [[100, 256, 256, 47], [64, 48, 48, 12]])
def test_sage_inference_with_placed_buffers_is_the_same_forward(dims, monkeypatch):
    """Round 5: the matrices the whole-graph launches gather from -- the input features, a hidden layer's rows, the chained projection --
    live in PLACED allocations (ops.placed_for_gather: several candidate allocations timed with a gather over the graph, the fastest
    kept; on the full-size graph the same launch takes 18.1 or 19.4 ms depending on the allocation).  Placement chooses memory, nothing
    else: the forward equals the unplaced one bit for bit, the buffers are reused by the next call, a modified feature tensor is
    re-read, what inference returns is a fresh tensor every time, and the oracle agrees."""
    from glnn_amd import ops
    from glnn_amd.graph import CSRGraph, FullNeighborLoader
    n = 6000
    indptr, indices = random_graph(n, 10, seed=dims[0], power=0.6, isolated=4, hub=1200)
    x = np.random.RandomState(1).standard_normal((n, dims[0])).astype(np.float32)
    model, layers, norms = _sage_model(dims, "batch", seed=2)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    loader = FullNeighborLoader(g, 512)
    feats = torch.from_numpy(x).to(DEV)
    monkeypatch.setattr(ops, "PLACEMENT_CANDIDATES", 1)                    # off: the plain allocations
    plain = model.inference(loader, feats).clone()
    monkeypatch.setattr(ops, "PLACEMENT_CANDIDATES", 4)
    monkeypatch.setattr(ops, "PLACEMENT_MIN_BYTES", 0)                       # (the real threshold is 256 MB: matrices that miss the caches)
    del ops.PLACEMENT_LOG[:]
    a = model.inference(loader, feats)
    tuned = [r["what"] for r in ops.PLACEMENT_LOG]
    assert len(tuned) >= 2 and any("features" in w for w in tuned) and any("y0" in w for w in tuned)
    assert all(len(r["ms"]) == 4 and 0 <= r["chosen"] < 4 for r in ops.PLACEMENT_LOG)
    b = model.inference(loader, feats)
    assert len(ops.PLACEMENT_LOG) == len(tuned)                              # second call: every buffer reused, nothing tuned again
    assert a.data_ptr() != b.data_ptr() and torch.equal(a, b) and torch.equal(a, plain)
    np.testing.assert_allclose(a.cpu().numpy(), to.sage_inference(indptr, indices, x, layers, norms), atol=TOL, rtol=0)
    feats.mul_(0.5)                                                          # in place: the remembered copy of the features is stale
    c = model.inference(loader, feats)
    np.testing.assert_allclose(c.cpu().numpy(), to.sage_inference(indptr, indices, 0.5 * x, layers, norms), atol=TOL, rtol=0)
    assert torch.equal(a, plain)                                             # earlier results are untouched by later calls


def test_placement_search_on_a_full_device_uses_the_candidates_that_fit(monkeypatch):
    """The placement search is an optimisation: when the device runs out of memory while candidates are being allocated (other tenants,
    two ranks on one GPU) it chooses among the ones that did fit instead of failing the forward."""
    from glnn_amd import ops
    n, d = 3000, 64
    indptr, indices = random_graph(n, 6, seed=3)
    ip, ix = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
    monkeypatch.setattr(ops, "PLACEMENT_CANDIDATES", 6)
    monkeypatch.setattr(ops, "PLACEMENT_MIN_BYTES", 0)
    real, calls = ops.feat_empty, []

    def short_of_memory(rows, dd, device, zero=False):
        calls.append(rows)
        if len(calls) > 2:
            raise torch.cuda.OutOfMemoryError("no memory for candidate %d" % len(calls))
        return real(rows, dd, device, zero=zero)
    monkeypatch.setattr(ops, "feat_empty", short_of_memory)
    del ops.PLACEMENT_LOG[:]
    buf = ops.placed_for_gather(n, d, torch.device(DEV), ip, ix, n, what="test", probe=lambda c: c.zero_())
    assert buf.shape == (n, d) and len(ops.PLACEMENT_LOG) == 1 and len(ops.PLACEMENT_LOG[0]["ms"]) == 2


def test_remembered_packed_weights_and_folded_tails_follow_every_kind_of_parameter_update():
    """SAGE.inference remembers the packed weights and the folded eval-mode BatchNorm tails across calls (round 5: eight small launches
    of a ~0.9 ms arxiv forward).  They must follow the parameters however those change: an in-place torch update, load_state_dict, and the
    writes THROUGH RAW POINTERS of this library's own training step (running statistics, fused Adam -- invisible to torch's version
    counters: ops.PARAM_EPOCH).  After each, inference equals the oracle evaluated at the model's current state."""
    from glnn_amd.graph import CSRGraph, FullNeighborLoader, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.teacher import TeacherEngine
    dims, n = [24, 32, 32, 5], 3000
    indptr, indices = random_graph(n, 8, seed=5, power=0.5, isolated=3)
    x = np.random.RandomState(3).standard_normal((n, dims[0])).astype(np.float32)
    labels = torch.from_numpy(np.random.RandomState(4).randint(0, dims[-1], n)).to(DEV)
    model, _, _ = _sage_model(dims, "batch", seed=4)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    loader, feats = FullNeighborLoader(g, 512), torch.from_numpy(x).to(DEV)

    def oracle_now():
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        L = len(dims) - 1
        layers = [dict(weight=sd[f"encoder.layers.{i}.fc_neigh.weight"], bias=sd[f"encoder.layers.{i}.fc_neigh.bias"]) for i in range(L)]
        norms = [dict(weight=sd[f"encoder.norms.{i}.weight"], bias=sd[f"encoder.norms.{i}.bias"], running_mean=sd[f"encoder.norms.{i}.running_mean"],
                      running_var=sd[f"encoder.norms.{i}.running_var"]) for i in range(L - 1)]
        return to.sage_inference(indptr, indices, x, layers, norms)

    def check(what):
        model.eval()
        got = model.inference(loader, feats).cpu().numpy()
        want = oracle_now()
        np.testing.assert_allclose(got, want, atol=TOL * max(1.0, float(np.abs(want).max())), rtol=0, err_msg=what)
        return got

    a = check("fresh model")
    check("second call (everything remembered)")
    with torch.no_grad():                                              # in-place torch updates: version counters
        model.encoder.layers[0].fc_neigh.weight.mul_(1.5)
        model.encoder.norms[1].running_var.add_(0.7)
        model.encoder.layers[1].fc_neigh.bias.add_(0.3)
    b = check("after in-place torch updates")
    assert np.abs(a - b).max() > 1e-3
    sd = {k: (v * 0.9 if v.dtype.is_floating_point else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    check("after load_state_dict")
    opt = torch.optim.Adam(model.parameters(), lr=0.05)                # this library's training step: raw-pointer writes
    model.train()
    eng = TeacherEngine(model, opt)
    nl = NodeDataLoader(g, torch.arange(1024), MultiLayerNeighborSampler([4, 4, 4]), batch_size=512, shuffle=False, seed=1)
    for input_nodes, output_nodes, blocks in nl:
        eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
    c = check("after two TeacherEngine steps")
    assert np.abs(b * 0 + c - a).max() > 1e-3


def test_evaluate_sage_log_probs_and_score():
    from glnn_amd import train_and_eval as te
    from glnn_amd.graph import CSRGraph, FullNeighborLoader
    n, dims = 1500, [16, 32, 5]
    indptr, indices = random_graph(n, 8, seed=4, power=0.5)
    x = np.random.RandomState(3).standard_normal((n, dims[0])).astype(np.float32)
    y = np.random.RandomState(4).randint(0, 5, n).astype(np.int64)
    model, layers, norms = _sage_model(dims, "batch", seed=2)
    want = to.log_softmax_(to.sage_inference(indptr, indices, x, layers, norms))
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    out, loss, score = te.evaluate(model, FullNeighborLoader(g, 256), torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV),
                                   torch.nn.NLLLoss(), lambda o, l: o.argmax(1).eq(l).float().mean().item())
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=TOL, rtol=0)
    assert abs(loss - so.nll_loss(want, y)) < TOL and abs(score - so.accuracy(want, y)) < 1e-6


def test_gcn_forward_cora_shape_vs_oracle():
    """config 0: cora-shaped GCN teacher (1433 -> 64 -> 7, reference train.conf.yaml:12-15)."""
    from glnn_amd import data
    from glnn_amd.models import Model
    g = data.make_graph("cora", seed=0, device="cpu")
    n = g.n_dst
    x = np.random.RandomState(0).standard_normal((n, 1433)).astype(np.float32) * 0.1
    torch.manual_seed(0)
    model = Model(dict(model_name="GCN", num_layers=2, feat_dim=1433, hidden_dim=64, label_dim=7, dropout_ratio=0.8,
                       norm_type="none", device=DEV))
    with torch.no_grad():
        for lay in model.encoder.layers:
            lay.bias.copy_(torch.randn_like(lay.bias) * 0.1)
    model.eval()
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    layers = [dict(weight=sd[f"encoder.layers.{i}.weight"], bias=sd[f"encoder.layers.{i}.bias"]) for i in range(2)]
    want = to.gcn_forward(g.indptr.numpy(), g.indices.numpy(), x, layers)
    with torch.no_grad():
        got = model.inference(g.to(DEV), torch.from_numpy(x).to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)


def test_teacher_composition_vs_reference_golden():
    """tests/golden/teacher_composition.npz holds what the reference's OWN models.py produced (SAGE.inference over a
    block dataloader; GCN.forward_fitnet) with dgl's two layers stubbed by torch.sparse stand-ins.  The reference's
    state_dict must load into this package's Model (strict) and reproduce those logits on the HIP path, through the
    whole-graph fast path, the reference-style chunked sweep, and the sampled-block forward."""
    from golden_inputs import teacher_composition
    from glnn_amd.graph import CSRGraph, FullNeighborLoader
    from glnn_amd.models import Model
    gold = teacher_composition()
    s = gold["sage"]
    dims = [int(v) for v in s["dims"]]
    model = Model(dict(model_name="SAGE", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                       dropout_ratio=0.5, norm_type="batch", device=DEV))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in s["sd"].items()}, strict=True)
    model.eval()
    n = len(s["indptr"]) - 1
    g = CSRGraph(torch.from_numpy(s["indptr"]).to(DEV), torch.from_numpy(s["indices"]).to(DEV), n)
    loader = FullNeighborLoader(g, int(s["batch_size"]))
    x = torch.from_numpy(s["feats"]).to(DEV)
    np.testing.assert_allclose(model.inference(loader, x).cpu().numpy(), s["logits"], atol=TOL, rtol=0)
    np.testing.assert_allclose(model.encoder.inference(loader, x, whole_graph=False).cpu().numpy(), s["logits"], atol=TOL, rtol=0)
    with torch.no_grad():      # SAGE.forward over [g, g, g] as full-neighbour "blocks" is the same function in eval mode
        np.testing.assert_allclose(model([g] * (len(dims) - 1), x).cpu().numpy(), s["logits"], atol=TOL, rtol=0)

    c = gold["gcn"]
    dims = [int(v) for v in c["dims"]]
    gcn = Model(dict(model_name="GCN", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                     dropout_ratio=0.8, norm_type="none", device=DEV))
    gcn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in c["sd"].items()}, strict=True)
    gcn.eval()
    n = len(c["indptr"]) - 1
    g = CSRGraph(torch.from_numpy(c["indptr"]).to(DEV), torch.from_numpy(c["indices"]).to(DEV), n)
    with torch.no_grad():
        h_list, logits = gcn.forward_fitnet(g, torch.from_numpy(c["feats"]).to(DEV))
    np.testing.assert_allclose(logits.cpu().numpy(), c["logits"], atol=TOL, rtol=0)
    np.testing.assert_allclose(h_list[0].cpu().numpy(), c["h0"], atol=TOL, rtol=0)


def test_feature_prop_vs_oracle():
    from glnn_amd import utils
    from glnn_amd.graph import CSRGraph
    n = 2000
    indptr, indices = random_graph(n, 7, seed=12, power=0.5, symmetric=True, self_loops=True)
    x = np.random.RandomState(5).standard_normal((n, 128)).astype(np.float32)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    got = utils.feature_prop(torch.from_numpy(x).to(DEV), g, 2)
    np.testing.assert_allclose(got.cpu().numpy(), to.feature_prop(indptr, indices, x, 2), atol=TOL, rtol=0)


def test_feature_prop_and_min_cut_vs_reference_golden():
    """tests/golden/host_logic.npz: what the reference's utils.feature_prop (utils.py:171-189) and its dense
    compute_min_cut_loss (utils.py:159-168) returned on a 220-node multigraph (isolated rows, a hub, multi-edges);
    here both run on the aggregation kernel (no dense adjacency)."""
    import os
    from glnn_amd import utils
    from glnn_amd.graph import CSRGraph
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic.npz"))
    ip, ix = z["mincut.indptr"], z["mincut.indices"]
    g = CSRGraph(torch.from_numpy(ip).to(DEV), torch.from_numpy(ix).to(DEV), len(ip) - 1)
    for k in (1, 3):
        got = utils.feature_prop(torch.from_numpy(z["fprop.feats"]).to(DEV), g, k)
        np.testing.assert_allclose(got.cpu().numpy(), z[f"fprop.k{k}"], atol=TOL, rtol=0)
    got = utils.compute_min_cut_loss(g, torch.from_numpy(z["mincut.logp"]))       # the reference passes a CPU `out` too (:160)
    assert abs(got - float(z["mincut.value"])) < TOL
    assert abs(got - to.min_cut_loss(ip, ix, z["mincut.logp"])) < 1e-5


def test_teacher_autograd_matches_torch_dense():
    """Aggregation + projection backward (the teacher-training direction) vs dense torch autograd."""
    from glnn_amd import ops
    from glnn_amd.autograd import SpmmFn, linear_fn
    from glnn_amd.graph import CSRGraph
    n, d, o = 300, 24, 10
    indptr, indices = random_graph(n, 5, seed=2, power=0.5, isolated=4)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    a = torch.zeros(n, n, dtype=torch.float64)
    for v in range(n):
        for e in range(indptr[v], indptr[v + 1]):
            a[v, indices[e]] += 1
    deg = a.sum(1, keepdim=True)
    x0 = torch.randn(n, d, dtype=torch.float64)
    w0 = torch.randn(o, d, dtype=torch.float64) / 5
    b0 = torch.randn(o, dtype=torch.float64)
    xr, wr, br = x0.clone().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
    yr = ((a @ xr + xr) / (deg + 1)) @ wr.t() + br
    (yr.pow(2).sum()).backward()
    x = x0.float().to(DEV).requires_grad_(); w = w0.float().to(DEV).requires_grad_(); b = b0.float().to(DEV).requires_grad_()
    y = linear_fn(SpmmFn.apply(g, x, ops.AGG_SAGE_GCN), w, b)
    (y.pow(2).sum()).backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(w.grad.cpu().numpy(), wr.grad.numpy(), atol=1e-2, rtol=1e-4)
    np.testing.assert_allclose(b.grad.cpu().numpy(), br.grad.numpy(), atol=1e-2, rtol=1e-4)


# ------------------------------------------------------------------------------------------- full size
def test_full_size_products_properties():
    """ogbn-products-shaped aggregation at FULL size through size-independent properties:
    (1) conservation: sum_v (deg_v+1) * out[v] == sum_u (outdeg_u+1) * x[u]  (fp64 check);
    (2) row-range sharding: out[a:b] from the row shard == the unsharded rows, bit for bit;
    (3) linearity: agg(2x + y) == 2 agg(x) + agg(y) to rounding."""
    from glnn_amd import data, ops
    g = data.make_graph("ogbn-products", seed=0, device=DEV)
    n, d = g.n_dst, 100
    assert n == 2449029 and g.num_edges() == 123718280
    x = torch.randn(n, d, device=DEV)
    out = ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN)
    deg = g.in_degrees().double()
    outdeg = g.out_degrees().double()
    lhs = ((deg + 1).unsqueeze(1) * out.double()).sum(0)
    rhs = ((outdeg + 1).unsqueeze(1) * x.double()).sum(0)
    assert float((lhs - rhs).abs().max() / rhs.abs().max().clamp(min=1)) < 1e-4
    lo, hi = n // 3, n // 3 + 300000
    shard = g.row_range(lo, hi)
    part = ops.spmm(shard.indptr, shard.indices, x, hi - lo, ops.AGG_SUM)
    full_sum = ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SUM)
    assert torch.equal(part, full_sum[lo:hi])
    y = torch.randn(n, d, device=DEV)
    lin = ops.spmm(g.indptr, g.indices, ops.as_feat(2 * x + y), n, ops.AGG_SAGE_GCN)
    outy = ops.spmm(g.indptr, g.indices, y, n, ops.AGG_SAGE_GCN)
    assert float((lin - (2 * out + outy)).abs().max()) < 1e-4


# ------------------------------------------------------------------------------------------- sampler
@pytest.mark.parametrize("norm", ["batch", "none", "layer"])
def test_training_mode_forward_under_autograd_matches_torch_ops(norm):
    """MLP.forward in TRAINING mode under torch autograd (callers that differentiate Model.forward themselves): Linear,
    norm -> ReLU -> dropout run as differentiable HIP ops (glnn_amd.autograd).  With dropout 0 outputs, gradients and the
    BatchNorm running statistics must equal the same chain written with torch ops IN THIS TEST (F.linear / F.batch_norm /
    relu -- the ops the reference's modules issue, models.py:42-53).  (The SAGE / GCN training direction is pinned by
    tests/test_teacher_gpu.py against the reference-generated golden.)"""
    import copy
    import torch.nn.functional as F
    from glnn_amd.models import Model
    n, dims = 1500, [24, 48, 48, 7]
    torch.manual_seed(0)
    x = torch.randn(n, dims[0], device=DEV)
    y = torch.randint(0, dims[-1], (n,), device=DEV)
    fused = Model(dict(model_name="MLP", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                       dropout_ratio=0.0, norm_type=norm, device=DEV))
    plain = copy.deepcopy(fused)
    fused.train(); plain.train()
    out_f = fused(None, x)
    F.nll_loss(out_f.log_softmax(1), y).backward()
    h = x
    enc = plain.encoder
    for l, layer in enumerate(enc.layers):
        h = F.linear(h, layer.weight, layer.bias)
        if l != len(enc.layers) - 1:
            if norm == "batch":
                bn = enc.norms[l]
                h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
                bn.num_batches_tracked += 1
            elif norm == "layer":
                ln = enc.norms[l]
                h = F.layer_norm(h, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
            h = F.relu(h)
    F.nll_loss(h.log_softmax(1), y).backward()
    np.testing.assert_allclose(out_f.detach().cpu().numpy(), h.detach().cpu().numpy(), atol=TOL, rtol=0)
    for (k, pf), (_, pp) in zip(fused.named_parameters(), plain.named_parameters()):
        if norm == "batch" and ".bias" in k and ".norms." not in k and not k.startswith(f"encoder.layers.{len(dims) - 2}"):
            continue      # bias in front of a BatchNorm: zero true gradient (tests/parity_rules.py)
        np.testing.assert_allclose(pf.grad.cpu().numpy(), pp.grad.cpu().numpy(), atol=2e-5, rtol=1e-3, err_msg=k)
    for (k, bf), (_, bp) in zip(fused.named_buffers(), plain.named_buffers()):
        np.testing.assert_allclose(bf.cpu().numpy(), bp.cpu().numpy(), atol=1e-5, rtol=1e-5, err_msg=k)


@pytest.mark.parametrize("norm", ["batch", "layer", "none"])
def test_gcn_with_norm_layers_training_step_and_eval_vs_dense_torch(norm):
    """GCN.forward with a norm behind the GraphConv (reference models.py:189-199: conv(relu inside) -> norms[l] -> dropout, NO ReLU
    behind the norm; train.conf.yaml's pokec / penn94 GCN sections use norm_type batch): the reference's full-graph `train` step
    (train_and_eval.py:12-29) on TeacherEngine.step_gcn and the eval forward, against the same model written with dense torch
    ops in this test (D^-1/2 A D^-1/2 as a dense matrix, F.batch_norm / F.layer_norm, torch.optim.Adam)."""
    import copy
    import torch.nn.functional as F
    from glnn_amd import train_and_eval as te
    from glnn_amd.graph import CSRGraph
    from glnn_amd.models import Model
    n, dims = 700, [20, 32, 32, 5]
    indptr, indices = random_graph(n, 6, seed=11, power=0.5, symmetric=True, self_loops=True)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    a = torch.zeros(n, n, dtype=torch.float64)
    for v in range(n):
        for e in range(indptr[v], indptr[v + 1]):
            a[v, indices[e]] += 1
    dinv_in, dinv_out = a.sum(1).clamp(min=1).pow(-0.5), a.sum(0).clamp(min=1).pow(-0.5)
    a_hat = (dinv_in[:, None] * a * dinv_out[None, :]).float().to(DEV)
    torch.manual_seed(3)
    x = torch.randn(n, dims[0], device=DEV)
    y = torch.randint(0, dims[-1], (n,), device=DEV)
    idx_train = torch.arange(0, n, 3, device=DEV)
    model = Model(dict(model_name="GCN", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                       norm_type=norm, device=DEV))
    with torch.no_grad():
        for nm in model.encoder.norms:
            nm.weight.uniform_(.5, 1.5); nm.bias.uniform_(-.2, .2)
    plain = copy.deepcopy(model)

    def dense_forward(m, training):
        h = x
        enc = m.encoder
        for l, layer in enumerate(enc.layers):
            h = a_hat @ (h @ layer.weight) + layer.bias
            if l != len(enc.layers) - 1:
                h = F.relu(h)
                if norm == "batch":
                    bn = enc.norms[l]
                    h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
                    if training:
                        bn.num_batches_tracked += 1
                elif norm == "layer":
                    ln = enc.norms[l]
                    h = F.layer_norm(h, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
        return h

    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    opt_p = torch.optim.Adam(plain.parameters(), lr=0.01, weight_decay=5e-4)
    crit = torch.nn.NLLLoss()
    for step in range(3):
        loss = te.train(model, g, x, y, crit, opt, idx_train)
        plain.train()
        out = dense_forward(plain, True).log_softmax(1)
        lp = crit(out[idx_train], y[idx_train])
        opt_p.zero_grad(); lp.backward(); opt_p.step()
        assert abs(loss - lp.item()) < TOL, (step, loss, lp.item())
    for (k, pf), (_, pp) in zip(model.named_parameters(), plain.named_parameters()):
        if norm == "batch" and k.endswith(".bias") and ".norms." not in k and not k.startswith("encoder.layers.2"):
            continue      # a conv bias in front of ReLU -> BatchNorm is NOT a pure gauge (the ReLU sits between), but keep the rule uniform
        np.testing.assert_allclose(pf.detach().cpu().numpy(), pp.detach().cpu().numpy(), atol=2e-4, rtol=0, err_msg=k)
    for (k, bf), (_, bp) in zip(model.named_buffers(), plain.named_buffers()):
        np.testing.assert_allclose(bf.cpu().numpy(), bp.cpu().numpy(), atol=1e-5, rtol=1e-5, err_msg=k)
    model.eval(); plain.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(g, x).cpu().numpy(), dense_forward(plain, False).cpu().numpy(), atol=2e-4, rtol=0)
    # the autograd surface (callers that differentiate GCN.forward themselves) agrees with the engine's first-step gradients
    m2 = copy.deepcopy(plain); m3 = copy.deepcopy(plain)
    m2.train(); m3.train()
    crit(m2(g, x).log_softmax(1)[idx_train], y[idx_train]).backward()
    crit(dense_forward(m3, True).log_softmax(1)[idx_train], y[idx_train]).backward()
    for (k, pf), (_, pp) in zip(m2.named_parameters(), m3.named_parameters()):
        np.testing.assert_allclose(pf.grad.cpu().numpy(), pp.grad.cpu().numpy(), atol=2e-5, rtol=1e-3, err_msg=k)


@pytest.mark.parametrize("dims", [[16, 64, 64, 5], [40, 300, 7]])
def test_sage_with_layernorm_inference_and_block_forward_vs_torch(dims):
    """SAGE with norm_type 'layer' (reference models.py:87-90): layer-wise inference (whole graph and chunked) and the
    training-mode block forward under autograd vs the same model with the aggregation as a dense matrix and F.layer_norm."""
    import copy
    import torch.nn.functional as F
    from glnn_amd.graph import CSRGraph, FullNeighborLoader
    from glnn_amd.models import Model
    n = 900
    indptr, indices = random_graph(n, 7, seed=5, power=0.5, isolated=4)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    a = torch.zeros(n, n)
    for v in range(n):
        for e in range(indptr[v], indptr[v + 1]):
            a[v, indices[e]] += 1
    a = ((a + torch.eye(n)) / (a.sum(1, keepdim=True) + 1)).to(DEV)          # SAGE-"gcn" mean incl. the self row
    torch.manual_seed(1)
    L = len(dims) - 1
    model = Model(dict(model_name="SAGE", num_layers=L, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                       norm_type="layer", device=DEV))
    with torch.no_grad():
        for nm in model.encoder.norms:
            nm.weight.uniform_(.5, 1.5); nm.bias.uniform_(-.2, .2)
    x = torch.randn(n, dims[0], device=DEV)

    def dense(m):
        h = x
        for l, layer in enumerate(m.encoder.layers):
            h = F.linear(a @ h, layer.fc_neigh.weight, layer.fc_neigh.bias)
            if l != L - 1:
                ln = m.encoder.norms[l]
                h = F.relu(F.layer_norm(h, ln.normalized_shape, ln.weight, ln.bias, ln.eps))
        return h

    model.eval()
    with torch.no_grad():
        want = dense(model).cpu().numpy()
    loader = FullNeighborLoader(g, 256)
    np.testing.assert_allclose(model.inference(loader, x).cpu().numpy(), want, atol=TOL, rtol=0)
    np.testing.assert_allclose(model.encoder.inference(loader, x, whole_graph=False).cpu().numpy(), want, atol=TOL, rtol=0)
    m2, m3 = copy.deepcopy(model), copy.deepcopy(model)
    m2.train(); m3.train()
    out2 = m2([g] * L, x)
    out2.pow(2).sum().backward()
    out3 = dense(m3)
    out3.pow(2).sum().backward()
    np.testing.assert_allclose(out2.detach().cpu().numpy(), out3.detach().cpu().numpy(), atol=TOL, rtol=0)
    for (k, pf), (_, pp) in zip(m2.named_parameters(), m3.named_parameters()):
        np.testing.assert_allclose(pf.grad.cpu().numpy(), pp.grad.cpu().numpy(), atol=2e-3, rtol=1e-3, err_msg=k)


def test_neighbor_sampler_and_blocks():
    from glnn_amd import ops
    from glnn_amd.graph import CSRGraph, MultiLayerNeighborSampler, NodeDataLoader
    n = 3000
    indptr, indices = random_graph(n, 20, seed=3, power=0.7, isolated=10, hub=400)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    seeds = torch.arange(0, n, 3, device=DEV)
    src, cnt = ops.sample_neighbors(g.indptr, g.indices, seeds, 7, 1234)
    deg = (g.indptr[seeds + 1] - g.indptr[seeds]).cpu().numpy()
    np.testing.assert_array_equal(cnt.cpu().numpy(), np.minimum(deg, 7))
    srcn, cntn = src.cpu().numpy(), cnt.cpu().numpy()
    for i, v in enumerate(seeds.cpu().numpy()):
        nb = indices[indptr[v]:indptr[v + 1]]
        got = srcn[i, :cntn[i]]
        # a sample without replacement of edge POSITIONS: multiplicities never exceed the row's
        u, c = np.unique(got, return_counts=True)
        un, cn = np.unique(nb, return_counts=True)
        mult = dict(zip(un, cn))
        assert all(mult.get(a, 0) >= b for a, b in zip(u, c))
    src2, _ = ops.sample_neighbors(g.indptr, g.indices, seeds, 7, 99)
    assert not torch.equal(src, src2)                                    # different rng seed -> different sample
    loader = NodeDataLoader(g, torch.arange(100, 900), MultiLayerNeighborSampler([5, 10]), batch_size=256, shuffle=True)
    assert len(loader) == 4
    seen = []
    for input_nodes, output_nodes, blocks in loader:
        assert len(blocks) == 2 and blocks[1].num_dst_nodes() == len(output_nodes)
        assert blocks[0].num_dst_nodes() == blocks[1].num_src_nodes() and blocks[0].num_src_nodes() == len(input_nodes)
        assert torch.equal(input_nodes[: len(output_nodes)], output_nodes)
        assert int(blocks[1].in_degrees().max()) <= 10 and int(blocks[0].in_degrees().max()) <= 5
        seen.append(output_nodes)
    assert sorted(torch.cat(seen).cpu().tolist()) == list(range(100, 900))


def test_train_sage_on_sampled_blocks_learns():
    """reference train_sage (train_and_eval.py:32-56) on GPU-sampled blocks: the loss must go down."""
    from glnn_amd import train_and_eval as te
    from glnn_amd.graph import CSRGraph, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    torch.manual_seed(0)
    n, f, c = 2000, 32, 4
    indptr, indices = random_graph(n, 10, seed=8, power=0.5, symmetric=True, self_loops=True)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    from glnn_amd import ops
    x = torch.randn(n, f, device=DEV)
    agg = ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN)          # labels depend on the NEIGHBOURHOOD mean
    y = (agg @ torch.randn(f, c, device=DEV)).argmax(1)
    model = Model(dict(model_name="SAGE", num_layers=2, feat_dim=f, hidden_dim=64, label_dim=c, dropout_ratio=0.0,
                       norm_type="batch", device=DEV))
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    loader = NodeDataLoader(g, torch.arange(n), MultiLayerNeighborSampler([5, 5]), batch_size=500, shuffle=True)
    losses = [te.train_sage(model, loader, x, y, torch.nn.NLLLoss(), opt) for _ in range(12)]
    assert losses[-1] < 0.85 * losses[0] and losses[-1] < min(losses[:3]), losses   # 5-of-~21 neighbour sampling is noisy


def test_full_size_arxiv_teacher_forward_vs_oracle():
    """BASELINE configs[1] at FULL size: ogbn-arxiv-shaped graph (169,343 nodes, 2,501,829 in-edges incl. reverse
    edges, multi-edges and self-loops), SAGE 128-256-256-40 with BatchNorm, layer-wise full-neighbour inference on
    the HIP path vs the CPU oracle (OpenMP) on identical inputs: max |diff| <= 1e-4."""
    from glnn_amd import data
    from glnn_amd.graph import FullNeighborLoader
    g = data.make_graph("ogbn-arxiv", seed=0, device="cpu")
    n = g.n_dst
    assert n == 169343 and g.num_edges() == 2501829
    x = np.random.RandomState(0).standard_normal((n, 128)).astype(np.float32)
    model, layers, norms = _sage_model([128, 256, 256, 40], "batch", seed=3)
    want = to.sage_inference(g.indptr.numpy(), g.indices.numpy(), x, layers, norms, threads=max(1, min(16, to.max_threads())))
    got = model.inference(FullNeighborLoader(g.to(DEV), 512), torch.from_numpy(x).to(DEV))
    assert got.shape == (n, 40)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)


def test_full_size_products_teacher_forward_vs_oracle():
    """The HEADLINE kernel at the HEADLINE size (BASELINE configs[3], what bench.py times): the ogbn-products-shaped graph at
    scale 1.0 (2,449,029 nodes, 123,718,280 in-edges, power-law with ~17k-degree hubs), SAGE 100-256-256-47 + BatchNorm,
    `Model.inference` through exactly the launches of the bench -- sage_fused_kernel<32> (100 -> 256), sage_fused_kernel<64> with
    the chained 256 -> 47 projection (76,532 tiles, > 128-degree rows through the 8-wave deferred path, the LDS ticket under real
    contention), spmm_csr_kernel<16> (47 wide) -- vs the CPU oracle (OpenMP) on identical inputs: max |diff| <= 1e-4.
    (reference models.py:121-148; the oracle aggregates first in every layer as dgl 0.6.1 does, the HIP path projects the
    last layer first.)  Then, at the same size: fused == unfused (stand-alone aggregation + GEMM) for both fused layers, and
    the fp64 conservation identity of the 256-wide aggregation."""
    from glnn_amd import data, ops
    from glnn_amd.graph import FullNeighborLoader
    g = data.make_graph("ogbn-products", seed=0, device=DEV)
    n = g.n_dst
    assert n == 2449029 and g.num_edges() == 123718280
    assert int(g.in_degrees().max()) > 4096            # the hub rows exist
    x = np.random.RandomState(0).standard_normal((n, 100)).astype(np.float32)
    model, layers, norms = _sage_model([100, 256, 256, 47], "batch", seed=5)
    xd = torch.from_numpy(x).to(DEV)
    loader = FullNeighborLoader(g, 4096)
    got = model.inference(loader, xd)
    assert got.shape == (n, 47)
    gc = g.to("cpu")
    want = to.sage_inference(gc.indptr.numpy(), gc.indices.numpy(), x, layers, norms, threads=to.max_threads())
    got_h = got.cpu().numpy()
    err = np.abs(got_h - want)
    assert np.isfinite(got_h).all()
    assert float(err.max()) <= TOL, (float(err.max()), int(np.argmax(err.max(1))))
    # the same forward without the chained projection and without the fused kernel: same numbers to rounding
    enc = model.encoder
    l0, l1, l2 = enc.layers
    s0, h0, _ = enc._tail(0)
    s1, h1, _ = enc._tail(1)
    xf = ops.as_feat(xd)
    y0_f = ops.sage_fused(g.indptr, g.indices, xf, n, l0.fc_neigh.weight, ep_scale=s0, ep_shift=h0, relu=True)
    agg0 = ops.spmm(g.indptr, g.indices, xf, n, ops.AGG_SAGE_GCN)
    y0_u = ops.gemm(agg0, l0.fc_neigh.weight, ep_scale=s0, ep_shift=h0, relu=True)
    assert float((y0_f - y0_u).abs().max()) <= 2e-5
    del agg0, y0_u
    y1_f, p2_f = ops.sage_fused(g.indptr, g.indices, y0_f, n, l1.fc_neigh.weight, ep_scale=s1, ep_shift=h1, relu=True,
                                w_next=l2.fc_neigh.weight)
    agg1 = ops.spmm(g.indptr, g.indices, y0_f, n, ops.AGG_SAGE_GCN)
    # conservation on the 256-wide aggregation (fp64): sum_v (deg_v + 1) * mean_v == sum_u (outdeg_u + 1) * y0[u]
    deg, outdeg = g.in_degrees().double(), g.out_degrees().double()
    lhs = ((deg + 1).unsqueeze(1) * agg1[:, :256].double()).sum(0)
    rhs = ((outdeg + 1).unsqueeze(1) * y0_f[:, :256].double()).sum(0)
    assert float((lhs - rhs).abs().max() / rhs.abs().max().clamp(min=1)) < 1e-5
    y1_u = ops.gemm(agg1, l1.fc_neigh.weight, ep_scale=s1, ep_shift=h1, relu=True)
    assert float((y1_f - y1_u).abs().max()) <= 2e-5
    p2_u = ops.gemm(y1_u, l2.fc_neigh.weight)
    assert float((p2_f[:, :47] - p2_u[:, :47]).abs().max()) <= 5e-5
    # chained-only form (hidden rows never written: what SAGE.inference launches) == the form that also writes them, bit for bit
    _, p2_c = ops.sage_fused(g.indptr, g.indices, y0_f, n, l1.fc_neigh.weight, ep_scale=s1, ep_shift=h1, relu=True,
                             w_next=l2.fc_neigh.weight, want_out=False)
    assert torch.equal(p2_c[:, :47], p2_f[:, :47])


def test_full_size_xl_teacher_forward_properties():
    """BASELINE configs[4] at FULL size on one GPU, as a TEACHER FORWARD (reference models.py:121-148): rank 4 of the 8-rank run --
    12.5 M destination rows with 250 M in-edges whose sources are drawn over ALL 100 M nodes -- through all three layers of
    glnn_amd.dist.ShardedTeacher (128-256-256-47, BN eval) with the peers emulated (dist.EmulatedPeers: every all-gather a local
    fill of the same bytes, so the layer-2 kernel gathers from a 102 GB hidden buffer and layer 3 from the 19 GB projected one, as
    a rank of the real run does).  `bench.py --workload xl` IS that run; its self-check holds, for every launch of a verification
    forward, size-independent properties: a sample of rows recomputed independently in torch fp64 from the launch's own inputs
    (<= 1e-4), stand-alone aggregations re-launched as a row range (bit-equal) and held to the fp64 conservation identity
    sum_v (deg_v + 1) mean_v == sum_u edges_out(u) x_u + sum_v x_self_v over ALL their rows; a repeated forward is bit-identical."""
    import json, os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "xl", "--steps", "3", "--warmup", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("DETAIL {")][0][len("DETAIL "):])
    cfg = out["config"]
    assert cfg["rows_per_gpu"] == 12_500_000 and cfg["nnz_per_gpu"] == 250_000_000 and cfg["nodes_total"] == 100_000_000 and cfg["rank_timed"] == 4
    assert out["verified"] is True and out["verify"]["repeat_forward_bit_equal"] and out["verify"]["finite"]
    launches = out["verify"]["launches"]
    assert len(launches) == 7                        # ONE aggregation launch over the 4 chunks (signals) + 4 replicated projections + ONE fused + chained launch over the chunks + ONE layer-3 aggregate
    for l in launches:
        assert l["max_abs_diff_vs_fp64"] <= 1e-4, l
        if l["launch"].startswith("spmm"):
            assert l["row_range_relaunch_bit_equal"] and l["conservation_rel_err_fp64_all_rows"] < 1e-5, l
    assert [l["layer"][0] for l in out["layers"]] == ["1", "1", "1", "2", "3", "3"]
    assert abs(out["per_forward"]["GB_received_per_rank"] - 4e-9 * 8 * 4 * 3_125_120 * (128 + 48)) < 1e-3      # (chunks of 3,125,000 rows padded to 128-row workgroups)


# ------------------------------------------------------------------------------------------- driver loops
@pytest.mark.parametrize("name", ["tran_nonorm", "tran_bn_wd", "ind_nonorm", "plain_mlp_tran"])
def test_driver_loops_vs_reference_golden(name):
    """The reference's own distill_run_transductive / distill_run_inductive / run_transductive (MLP branch) were run on small
    gauge-free configs (tests/golden/make_driver_golden.py): epoch order, per-epoch evaluation rows, best-validation
    snapshot + patience early stop, final evaluation of the restored state.  The mirror must stop at the same epoch, log the
    same "Best valid model" line and return the same log-probs; it draws its permutations with torch.randperm after
    set_seed(seed) exactly like the reference, so nothing is replayed."""
    import os
    from glnn_amd import train_and_eval as te
    from glnn_amd import utils
    from glnn_amd.models import Model
    from golden_inputs import make_inputs, make_state
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "driver_loops.npz"))
    cfg = lambda k: z[f"{name}.cfg.{k}"]
    dims, norm, seed, n = [int(d) for d in cfg("dims")], str(cfg("norm")), int(cfg("seed")), int(cfg("n"))
    feats, _, out_t, _ = make_inputs(seed, n, dims[0], dims[-1], 10)
    w = np.random.RandomState(seed).standard_normal((dims[0], dims[-1])).astype(np.float32)
    labels = (feats @ w).argmax(1).astype(np.int64)
    perm = np.random.RandomState(seed + 1).permutation(n)
    idx_train, idx_val, idx_test = (torch.from_numpy(perm[a:b].astype(np.int64)) for a, b in ((0, 120), (120, 200), (200, n)))
    conf = dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                norm_type=norm, device=DEV, seed=seed, batch_size=int(cfg("B")), lamb=float(cfg("lamb")), max_epoch=int(cfg("max_epoch")),
                patience=int(cfg("patience")), eval_interval=1)
    model = Model(conf)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state(seed, dims, norm).items()})
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=float(cfg("wd")))
    crit_l, crit_t = torch.nn.NLLLoss(), torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    evaluator = utils.get_evaluator("cora")

    class Log:
        lines = []

        def debug(self, m):
            self.lines.append(m)

        info = debug

    logger, las = Log(), []
    logger.lines = []
    tf, tl, tt = torch.from_numpy(feats), torch.from_numpy(labels), torch.from_numpy(out_t)
    kind = str(cfg("kind"))
    if kind == "plain":
        res = te.run_transductive(conf, model, None, tf, tl, (idx_train, idx_val, idx_test), crit_l, evaluator, opt, logger, las)
    elif kind == "ind":
        obs_tr, obs_va, obs_te, idx_obs, idx_ti = utils.graph_split(idx_train, idx_val, idx_test, 0.25, seed)
        res = te.distill_run_inductive(conf, model, tf, tl, tt, (obs_tr, torch.cat([obs_tr, obs_va, obs_te]), obs_va, obs_te, idx_obs, idx_ti),
                                       crit_l, crit_t, evaluator, opt, logger, las)
    else:
        res = te.distill_run_transductive(conf, model, tf, tl, tt, (idx_train, torch.cat([idx_train, idx_val, idx_test]), idx_val, idx_test),
                                          crit_l, crit_t, evaluator, opt, logger, las)
    want_las = z[f"{name}.loss_and_score"]
    assert len(las) == len(want_las), (len(las), len(want_las))                      # stopped at the same epoch
    np.testing.assert_allclose(np.asarray(las, np.float64), want_las, atol=2e-4, rtol=0)
    assert logger.lines[-1] == str(z[f"{name}.last_log"])
    np.testing.assert_allclose(np.asarray(res[1:], np.float64), z[f"{name}.scores"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(res[0].cpu().numpy(), z[f"{name}.out"], atol=2e-4, rtol=0)
    for k, v in model.state_dict().items():
        if v.ndim:
            np.testing.assert_allclose(v.cpu().numpy(), z[f"{name}.final.{k}"], atol=2e-4, rtol=0, err_msg=k)


def test_teacher_forward_and_student_steps_are_bit_reproducible_across_processes():
    """scripts/forward_repro_probe.py in two processes: the same synthetic graphs, the same SAGE.inference output (arxiv shape and a
    0.25-scale products shape: fused launches, hub rows, chained projection) and the same student parameters after five steps (MLP3w4 at
    B = 512: latency kernels, Adam folds; MLP3w8 at B = 4096: pipelined GEMMs, split-K slabs), sha256 for sha256.  Every reduction in the
    library has a fixed order; nothing depends on addresses or timing."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "forward_repro_probe.py")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith(("teacher forward", "student"))])
    assert len(outs[0]) == 4 and outs[0] == outs[1], outs
