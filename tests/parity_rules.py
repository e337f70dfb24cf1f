"""How a trained student state is compared with the reference's.

A Linear bias (and the BatchNorm running_mean that tracks it) directly in front of a BatchNorm has a
mathematically ZERO gradient: the computed gradient is pure fp32 rounding noise, and Adam turns noise of
any size into +-lr steps.  With weight_decay = 0 (the ogbn-arxiv / ogbn-products configs, reference
train.conf.yaml:142-194) those entries are a gauge freedom the network output does not depend on, and no
two implementations (or two BLAS builds under the reference itself) agree on them.  The same mechanism
makes a few weight entries whose true gradient is ~0 wander by a fraction of lr.  So:
  * outputs (losses per step, eval log-probs) are held to the 1e-4 bar -- they are gauge-invariant;
  * well-conditioned state (last layer, BN affine, running_var) is held to 1e-4;
  * hidden-layer weights: mean |diff| <= 1e-4 and max |diff| <= lr (a handful of sign-flipped entries);
  * gauge entries are skipped when weight_decay == 0, and held to 1e-4 otherwise;
  * EVAL-mode outputs after training in a gauge case: the bias noise enters through
    (b_final - EMA_t(b_t)) / sqrt(running_var) and shows up as ~1e-3 in the log-probs after a few steps
    (measured: 2e-3 between the numpy oracle and torch on the same CPU) -> eval_tol() = 1e-2 there,
    1e-4 everywhere else.  Eval-mode forward at IDENTICAL state is pinned to 1e-4 separately.
"""
import re

import numpy as np

TOL = 1e-4


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
        if m and int(m.group(1)) < L - 1 and g.norm == "batch" and g.wd == 0:
            assert d.mean() <= tol and d.max() <= g.lr, (k, d.mean(), d.max())
        else:
            assert d.max() <= tol, (k, d.max())


def has_gauge(g):
    return g.norm == "batch" and g.wd == 0


def eval_tol(g):
    return 1e-2 if has_gauge(g) else TOL
