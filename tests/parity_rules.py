"""How a trained student state is compared with the reference's.

A Linear bias (and the BatchNorm running_mean that tracks it) directly in front of a BatchNorm has a
mathematically ZERO gradient: the computed gradient is pure fp32 rounding noise, and Adam turns noise of
any size into +-lr steps.  With weight_decay = 0 (the ogbn-arxiv / ogbn-products configs, reference
train.conf.yaml:142-194) those entries are a gauge freedom the network output does not depend on, and no
two implementations (or two BLAS builds under the reference itself) agree on them.  The same mechanism
makes a few weight entries whose true gradient is ~0 wander by a fraction of lr.  So:
  * outputs (losses per step, eval log-probs) are held to the 1e-4 bar -- they are gauge-invariant;
  * well-conditioned state (last layer, BN affine, running_var) is held to 1e-4;
  * hidden-layer weights: mean |diff| <= max(1e-4, lr/20) and max |diff| <= 2 lr (entries whose first Adam steps are
    +-lr with a noise-decided sign -- two such steps for a handful of the 1-4 M entries of the wide students: HIP vs
    reference max 1.6e-2 on MLP3w4, reference vs itself 1.3e-2 on MLP3w8; measured on MLP3w4 after 4 steps, where the weights have moved by 1.3e-2 on average:
    numpy oracle vs reference mean 7.7e-5, HIP vs reference 1.4e-4, reference vs itself under a one-ulp perturbation of
    the initial weights max 7.8e-4 -- all ~1 % of the movement); at the
    full widths the reference runs (MLP3w4 1024, MLP3w8 2048) the flipped entries of W_l perturb the next steps'
    gradients of everything upstream, so the BatchNorm affine parameters of those configs get the same rule
    (measured numpy-oracle vs reference, MLP3w4 after 4 steps: gamma_0 mean 6e-5, max 5e-4 = lr/20) -- the
    per-step losses, which see all of it, still agree to 1e-4;
  * gauge entries are skipped when weight_decay == 0, and held to 1e-4 otherwise;
  * EVAL-mode outputs after training (round 3): anchored in the reference's FLOAT64 run -- check_eval_out() below; the
    paragraph that follows is the round-2 rule (4 x self-noise), kept for the `noise.*` keys it explains, which still bound
    the trained STATE entries.
  * (round 2) EVAL-mode outputs after training: the bar comes from the REFERENCE ITSELF.  Every fixture stores `noise.*`:
    what two runs of the reference's own train_mini_batch / evaluate_mini_batch disagree by when the initial
    weights of one are moved by a single fp32 ulp (tests/golden/make_student_golden.py).  Without the gauge
    freedom that is ~3e-6 (bn_small, nonorm_fullbatch, dropout case) and eval_tol() is the 1e-4 bar.  In the
    BatchNorm + weight_decay=0 configs the reference disagrees with ITSELF by 2.3e-3 (arxiv dims), 3.1e-3
    (products dims), 4.1e-3 (MLP3w4), 3.5e-2 (MLP3w8, 3 steps) in the eval log-probs -- the bias noise enters
    through (b_final - EMA_t(b_t)) / sqrt(running_var) and Adam's first steps amplify it -- so no implementation,
    the reference included, reproduces those outputs to 1e-4; eval_tol() = 4 x that measured self-noise (max and
    mean both checked).  A gauge-fixed comparison (our trained state with the reference's final bias /
    running_mean substituted) was tried and does not help: 2.0e-3 -> 1.5e-3 on arxiv dims, because the
    perturbed hidden weights, not the gauge entries themselves, carry most of the difference.
    Eval-mode forward at IDENTICAL state is pinned to 1e-4 separately (test_eval_forward_at_identical_state).
"""
import re

import numpy as np

TOL = 1e-4
NOISE_FACTOR = 4.0        # tolerances derived from the reference's own self-noise (fixture keys noise.*) use this multiple


def is_gauge(g, key):
    if g.norm != "batch" or g.wd != 0:
        return False
    L = len(g.dims) - 1
    m = re.match(r"encoder\.layers\.(\d+)\.bias", key)
    if m and int(m.group(1)) < L - 1:
        return True
    return bool(re.match(r"encoder\.norms\.\d+\.running_mean", key))


def check_final_state(g, sd, tol=TOL):
    L = len(g.dims) - 1
    for k, v in sd.items():
        ref = g.z[f"final.{k}"]
        v = np.asarray(v)
        if v.ndim == 0:
            assert int(v) == int(ref), k
            continue
        if is_gauge(g, k):
            continue
        d = np.abs(g.view(v).astype(np.float64) - ref)
        m = re.match(r"encoder\.layers\.(\d+)\.weight", k)
        affine = re.match(r"encoder\.norms\.\d+\.(weight|bias)", k)
        if ((m and int(m.group(1)) < L - 1) or affine) and g.norm == "batch" and g.wd == 0:
            assert d.mean() <= max(tol, 0.05 * g.lr) and d.max() <= 2 * g.lr, (k, d.mean(), d.max())
        else:   # 1e-4, or 4x what the reference itself moves by under a one-ulp perturbation (MLP3w8 last layer: 1.5e-4)
            assert d.max() <= max(tol, NOISE_FACTOR * float(g.z[f"noise.final.{k}"][0])), (k, d.max())


def has_gauge(g):
    return g.norm == "batch" and g.wd == 0


# post-training outputs: |impl - ref_fp64| <= 2 x the reference's own fp32-to-fp64 distance, taken over its TWO fp32 draws
# (3 x until round 4: the pipelined GEMMs now accumulate in blocks of 256 k, see csrc/gemm.hip pipe_mainloop).
ANCHOR_FACTOR = 2.0


def _anchor(g, which):
    """max (which=0) / mean (which=1) distance of the reference's fp32 eval log-probs from its fp64 ones: the larger of the
    plain fp32 run and the fp32 run whose initial weights were moved by one ulp (both in every fixture)."""
    return max(float(g.z["f64.dist_eval_out"][which]), float(g.z["f64.dist_eval_out_perturbed"][which]))


def check_eval_out(g, out):
    """Eval-mode log-probs AFTER training (round 3): anchored in the reference's own FLOAT64 run of the same passes (fixture
    keys f64.*, tests/golden/make_student_golden.py).  The fixtures hold TWO draws of the reference's fp32 rounding noise
    around that fp64 result -- the plain fp32 run and the run from initial weights moved by one ulp; on MLP3w8 they sit 9.2e-3 /
    3.4e-2 (max) and 9.5e-4 / 3.4e-3 (mean) away from it: one ulp decides the sign of Adam's first steps on zero-gradient
    entries.  An implementation must stay within 2 x the larger draw, max and mean, never tighter than the 1e-4 bar.
    Measured in round 4 (scripts/anchor_ratios.py, ratio to the larger draw): HIP 0.25 - 0.94 x on the BatchNorm students (MLP3w8:
    max 0.48 x, mean 0.88 x -- 3.0e-3 against the perturbed reference's 3.4e-3; against the plain fp32 run alone that is 3.2 x, which
    says how far apart the reference's own two draws are, not how noisy the GEMMs are: it did not move when the pipelined GEMMs'
    rounding noise dropped 2.8 - 3.9 x to below numpy's, scripts/gemm_noise.py), 1.1 - 1.8 x on the four fixtures whose distances are
    1e-7 .. 3e-6, i.e. decided by the 1e-4 floor.  Per-step losses, step-1 gradients and eval at identical state all hold 1e-4.
    Round 2 used 4 x the fp32 self-noise against the reference's FP32 outputs (0.14 max on MLP3w8), round 3 a factor of 3 here."""
    d = np.abs(g.view(np.asarray(out)).astype(np.float64) - np.asarray(g.z["f64.eval_out"], np.float64))
    tol_max, tol_mean = max(TOL, ANCHOR_FACTOR * _anchor(g, 0)), max(TOL / 5, ANCHOR_FACTOR * _anchor(g, 1))
    assert d.max() <= tol_max, ("eval max", d.max(), tol_max)
    assert d.mean() <= tol_mean, ("eval mean", d.mean(), tol_mean)


def eval_tol(g):
    """max-abs distance allowed between an implementation's eval log-probs and the reference's FP32 ones after training: both
    lie within their anchors of the fp64 result, so (1 + ANCHOR_FACTOR) x |ref_fp32 - ref_fp64|, never below 1e-4."""
    return max(TOL, (1.0 + ANCHOR_FACTOR) * _anchor(g, 0))


def eval_mean_tol(g):
    return max(TOL / 5, (1.0 + ANCHOR_FACTOR) * _anchor(g, 1))


def eval_loss_tol(g):
    return max(TOL, ANCHOR_FACTOR * max(float(g.z["f64.dist_eval_loss"]), float(g.z["f64.dist_eval_loss_perturbed"])), eval_mean_tol(g))


def moment_tols(g, pname):
    """(atol exp_avg, atol exp_avg_sq): Adam moments are gradient-scale quantities.  Held to the 1e-4 bar in the gauge
    configs (their gradients from step 2 on are taken at the perturbed weights described above; measured: a few entries
    at 1.6e-5 on MLP3w4) and 10x tighter elsewhere."""
    return (1e-4, 1e-6) if has_gauge(g) else (1e-5, 1e-7)
