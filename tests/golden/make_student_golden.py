"""Generate golden vectors for the student path FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the produced .npz files are committed
and are what travels.  The reference's Python is imported, never copied: `dgl`/`ogb` (absent here,
needed only by top-level imports at reference models.py:4, train_and_eval.py:4, utils.py:10-11) are
replaced by empty import-only stubs.

What is exercised (all reference code): models.Model / models.MLP, train_and_eval.train_mini_batch,
train_and_eval.evaluate_mini_batch, with torch.optim.Adam / NLLLoss / KLDivLoss(batchmean,
log_target=True) constructed exactly as train_student.py:274-279 does.

Inputs and initial weights come from numpy's legacy RandomState (bit-stable across platforms), so
big cases store only the seed; expected outputs are stored as arrays (full for small cases,
strided samples + norms for the arxiv-dims case).

    python tests/golden/make_student_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    dgl = types.ModuleType("dgl")
    dgl_nn = types.ModuleType("dgl.nn")
    for n in ("GraphConv", "SAGEConv", "APPNPConv", "GATConv"):
        setattr(dgl_nn, n, type(n, (), {}))
    dgl_fn = types.ModuleType("dgl.function")
    dgl.nn, dgl.function = dgl_nn, dgl_fn
    ogb = types.ModuleType("ogb")
    ogb_npp = types.ModuleType("ogb.nodeproppred")
    ogb_npp.Evaluator = type("Evaluator", (), {})
    ogb_npp.DglNodePropPredDataset = type("DglNodePropPredDataset", (), {})
    ogb.nodeproppred = ogb_npp
    sys.modules.update({"dgl": dgl, "dgl.nn": dgl_nn, "dgl.function": dgl_fn, "ogb": ogb,
                        "ogb.nodeproppred": ogb_npp})


def make_inputs(seed, n, f, c, n_l):
    """Shared with tests/ (tests/golden_inputs.py re-implements the same three lines)."""
    rs = np.random.RandomState(seed)
    feats = rs.standard_normal((n, f)).astype(np.float32)
    labels = rs.randint(0, c, size=n).astype(np.int64)
    t = rs.standard_normal((n, c)).astype(np.float32)
    t = t - t.max(1, keepdims=True)
    out_t = (t - np.log(np.exp(t.astype(np.float64)).sum(1, keepdims=True))).astype(np.float32)
    idx_l = rs.permutation(n)[:n_l].astype(np.int64)
    return feats, labels, out_t, idx_l


def make_state(seed, dims, norm):
    """dims = [f, h, ..., c]; uniform(-1/sqrt(in), 1/sqrt(in)) like nn.Linear's default range."""
    rs = np.random.RandomState(seed + 1000)
    sd = {}
    L = len(dims) - 1
    for i in range(L):
        bound = 1.0 / np.sqrt(dims[i])
        sd[f"encoder.layers.{i}.weight"] = rs.uniform(-bound, bound, (dims[i + 1], dims[i])).astype(np.float32)
        sd[f"encoder.layers.{i}.bias"] = rs.uniform(-bound, bound, (dims[i + 1],)).astype(np.float32)
    if norm == "layer":
        for i in range(L - 1):
            h = dims[i + 1]
            sd[f"encoder.norms.{i}.weight"] = rs.uniform(0.5, 1.5, (h,)).astype(np.float32)
            sd[f"encoder.norms.{i}.bias"] = rs.uniform(-0.2, 0.2, (h,)).astype(np.float32)
    if norm == "batch":
        for i in range(L - 1):
            h = dims[i + 1]
            sd[f"encoder.norms.{i}.weight"] = rs.uniform(0.5, 1.5, (h,)).astype(np.float32)
            sd[f"encoder.norms.{i}.bias"] = rs.uniform(-0.2, 0.2, (h,)).astype(np.float32)
            sd[f"encoder.norms.{i}.running_mean"] = rs.uniform(-0.1, 0.1, (h,)).astype(np.float32)
            sd[f"encoder.norms.{i}.running_var"] = rs.uniform(0.8, 1.2, (h,)).astype(np.float32)
            sd[f"encoder.norms.{i}.num_batches_tracked"] = np.int64(0)
    return sd


CASES = {
    # name: dims, norm, dropout (train-mode dropout is 0 in parity cases), B, N, N_l, lamb, lr, wd, epochs, store_full
    "bn_small": dict(dims=[24, 48, 48, 10], norm="batch", B=32, n=200, n_l=70, lamb=0.3, lr=0.01, wd=5e-4, epochs=2, full=True, seed=1),
    "nonorm_fullbatch": dict(dims=[60, 32, 7], norm="none", B=512, n=300, n_l=140, lamb=0.0, lr=0.01, wd=5e-3, epochs=2, full=True, seed=2),
    "arxiv_dims": dict(dims=[128, 256, 256, 40], norm="batch", B=512, n=1536, n_l=600, lamb=0.0, lr=0.01, wd=0.0, epochs=1, full=False, seed=3),
    "products_dims_narrow": dict(dims=[100, 256, 256, 47], norm="batch", B=4096, n=8192, n_l=4100, lamb=0.5, lr=0.01, wd=0.0, epochs=1, full=False, seed=4),
    # the students the reference's experiments actually run, at FULL width (experiments/glnn_arxiv.sh:7, glnn_products.sh:7;
    # train.conf.yaml:149-154, 187-194); lamb = 0 is the CLI default (train_student.py:154-159): the hard pass still steps
    "mlp3w4": dict(dims=[128, 1024, 1024, 40], norm="batch", B=512, n=1536, n_l=600, lamb=0.0, lr=0.01, wd=0.0, epochs=1, full=False, seed=5),
    "mlp3w8": dict(dims=[100, 2048, 2048, 47], norm="batch", B=4096, n=8192, n_l=4100, lamb=0.0, lr=0.01, wd=0.0, epochs=1, full=False, seed=6, stride=211),
    # dropout > 0 with the keep-mask applied OUTSIDE: the reference's nn.Dropout is swapped for a multiply by the
    # counter-based mask of libglnn_hip.so (restated in numpy: oracle/dropout_mask.py), seeds as StudentEngine derives them
    # nn.LayerNorm tails (reference models.py:30-31; train.conf.yaml:257-264 house_class MLP: 3 x 512, norm_type layer, dropout 0,
    # weight_decay 0): a small full-detail case with weight decay and the house_class shape (its batch is the whole training set)
    "ln_small": dict(dims=[24, 48, 48, 10], norm="layer", B=32, n=200, n_l=70, lamb=0.3, lr=0.01, wd=5e-4, epochs=2, full=True, seed=8),
    "ln_house_dims": dict(dims=[16, 512, 512, 5], norm="layer", B=512, n=1536, n_l=600, lamb=0.0, lr=0.01, wd=0.0, epochs=1, full=False, seed=9),
    "bn_small_dropout": dict(dims=[24, 48, 48, 10], norm="batch", B=32, n=200, n_l=70, lamb=0.3, lr=0.01, wd=5e-4, epochs=2, full=True, seed=7,
                             dropout=0.4, drop_base_seed=0x00C0FFEE),
}

SAMPLE_STRIDE = 53


_stride = [SAMPLE_STRIDE]        # per case (cfg["stride"]): the 4.5 M-parameter MLP3w8 is sampled more sparsely


def sample(a):
    a = np.asarray(a, dtype=np.float32).ravel()
    return a[::_stride[0]].copy()


class MaskDropout(torch.nn.Module):
    """Stands in for the reference MLP's `self.dropout` (models.py:18,52) in the dropout case: multiplies by the keep-mask
    of (optimiser step, hidden layer) and 1/(1-p); identity in eval mode like nn.Dropout.  The reference calls it once per
    hidden layer per forward, in layer order, so a call counter identifies (step, layer)."""

    def __init__(self, p, base_seed, hidden_layers):
        super().__init__()
        self.p, self.base_seed, self.hidden_layers = p, base_seed, hidden_layers
        self.calls = 0

    def forward(self, h):
        if not self.training:
            return h
        from oracle.dropout_mask import engine_seed, keep_mask
        step, layer = self.calls // self.hidden_layers + 1, self.calls % self.hidden_layers
        self.calls += 1
        keep = keep_mask(h.shape[0], h.shape[1], self.p, engine_seed(self.base_seed, step, layer)).astype(np.float32)
        return h * torch.from_numpy(keep) * np.float32(1.0 / (1.0 - self.p))


def run_case(name, cfg):
    import models as ref_models            # noqa: reference module
    import train_and_eval as ref_te        # noqa: reference module

    dims, norm = cfg["dims"], cfg["norm"]
    L = len(dims) - 1
    _stride[0] = int(cfg.get("stride", SAMPLE_STRIDE))
    feats, labels, out_t, idx_l = make_inputs(cfg["seed"], cfg["n"], dims[0], dims[-1], cfg["n_l"])
    sd0 = make_state(cfg["seed"], dims, norm)

    conf = dict(model_name="MLP", num_layers=L, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                dropout_ratio=0.0, norm_type=norm, device="cpu")
    torch.manual_seed(cfg["seed"])
    model = ref_models.Model(conf)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd0.items()})
    p_drop = float(cfg.get("dropout", 0.0))
    if p_drop > 0:
        model.encoder.dropout = MaskDropout(p_drop, cfg["drop_base_seed"], L - 1)
    # exactly train_student.py:274-279
    optimizer = torch.optim.Adam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
    criterion_l = torch.nn.NLLLoss()
    criterion_t = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)

    tf, tl, tt = torch.from_numpy(feats), torch.from_numpy(labels), torch.from_numpy(out_t)
    til = torch.from_numpy(idx_l)
    feats_l, labels_l = tf[til], tl[til]            # train_and_eval.py:553
    feats_t, out_tt = tf, tt                        # idx_t = all nodes here

    out = {}
    # ---- single-step gradients (loop body of train_and_eval.py:74-85, before the optimizer step)
    for kind, (x, y, crit, lam) in {"nll": (feats_l, labels_l, criterion_l, cfg["lamb"] if cfg["lamb"] else 1.0),
                                    "kl": (feats_t, out_tt, criterion_t, 1 - cfg["lamb"])}.items():
        bsz = min(cfg["B"], x.shape[0])
        model.train()
        sd_before = {k: v.clone() for k, v in model.state_dict().items()}
        logits = model(None, x[:bsz])
        o = logits.log_softmax(dim=1)
        loss = crit(o, y[:bsz])
        lv = loss.item()
        loss = loss * lam
        optimizer.zero_grad()
        logits.retain_grad()
        loss.backward()
        out[f"step_{kind}_lamb"] = np.float32(lam)
        out[f"step_{kind}_loss"] = np.float32(lv)
        g_logits = logits.grad.numpy()
        out[f"step_{kind}_logits"] = logits.detach().numpy() if cfg["full"] else sample(logits.detach().numpy())
        out[f"step_{kind}_dlogits"] = g_logits if cfg["full"] else sample(g_logits)
        for pname, p in model.named_parameters():
            g = p.grad.numpy()
            out[f"step_{kind}_grad.{pname}"] = g.copy() if cfg["full"] else sample(g)
            out[f"step_{kind}_gradnorm.{pname}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
        optimizer.zero_grad()
        model.load_state_dict(sd_before)            # undo the BN running-stat update
        if p_drop > 0:
            model.encoder.dropout.calls = 0         # every single-step probe and the passes below start at optimiser step 1

    # ---- full passes through the reference's train_mini_batch, recording perms and step losses
    perms, step_losses = [], []
    real_randperm = torch.randperm

    def rec_randperm(*a, **k):
        p = real_randperm(*a, **k)
        perms.append(p.numpy().copy())
        return p

    class Rec:
        def __init__(self, crit):
            self.crit = crit

        def __call__(self, o, y):
            l = self.crit(o, y)
            step_losses.append(float(l.item()))
            return l

    torch.randperm = rec_randperm
    pass_means = []
    try:
        torch.manual_seed(cfg["seed"])
        for _ in range(cfg["epochs"]):                # train_and_eval.py:559-566
            pass_means.append(ref_te.train_mini_batch(model, feats_l, labels_l, cfg["B"], Rec(criterion_l), optimizer, cfg["lamb"]))
            pass_means.append(ref_te.train_mini_batch(model, feats_t, out_tt, cfg["B"], Rec(criterion_t), optimizer, 1 - cfg["lamb"]))
    finally:
        torch.randperm = real_randperm

    out["pass_means"] = np.asarray(pass_means, np.float64)
    out["step_losses"] = np.asarray(step_losses, np.float64)
    for i, p in enumerate(perms):
        out[f"perm_{i}"] = p.astype(np.int32)
    out["num_perms"] = np.int64(len(perms))
    for k, v in model.state_dict().items():
        a = v.numpy()
        out[f"final.{k}"] = a.copy() if (cfg["full"] or a.ndim == 0) else sample(a)
        if a.ndim > 0:
            out[f"finalnorm.{k}"] = np.float64(np.linalg.norm(a.astype(np.float64)))
    st = optimizer.state_dict()["state"]
    for i, (pname, _) in enumerate(model.named_parameters()):
        for key in ("exp_avg", "exp_avg_sq"):
            a = st[i][key].numpy()
            out[f"adam.{key}.{pname}"] = a.copy() if cfg["full"] else sample(a)
    out["adam.step"] = np.int64(int(st[0]["step"]))

    # ---- evaluate_mini_batch (train_and_eval.py:108-136) on all nodes
    evaluator = lambda o, y: o.argmax(1).eq(y).float().mean().item()     # utils.py:151-156
    o_all, loss_e, score_e = ref_te.evaluate_mini_batch(model, tf, tl, criterion_l, cfg["B"], evaluator)
    out["eval_out"] = o_all.numpy() if cfg["full"] else sample(o_all.numpy())
    out["eval_loss"] = np.float64(loss_e)
    out["eval_score"] = np.float64(score_e)

    # ---- the reference's OWN sensitivity to rounding: the same passes (same permutations) from initial weights moved by one
    # fp32 ulp (x * (1 +- 2^-23), seeded signs).  Whatever two runs of the reference disagree by under a perturbation the
    # size of a single rounding error is the floor below which a trained state / its eval outputs are not determined;
    # tests/parity_rules.py derives its post-training tolerances from these numbers instead of from a hand-set constant.
    rs_n = np.random.RandomState(cfg["seed"] + 77)
    sd1 = {}
    for k, v in sd0.items():
        a = np.asarray(v)
        sd1[k] = (a * (1.0 + rs_n.choice([-1.0, 1.0], size=a.shape) * 2.0 ** -23)).astype(np.float32) if a.dtype == np.float32 else a
    model2 = ref_models.Model(conf)
    model2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd1.items()})
    if p_drop > 0:
        model2.encoder.dropout = MaskDropout(p_drop, cfg["drop_base_seed"], L - 1)
    opt2 = torch.optim.Adam(model2.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
    replay = iter(perms)
    losses2 = []

    class Rec2(Rec):
        def __call__(self, o, y):
            l = self.crit(o, y)
            losses2.append(float(l.item()))
            return l

    torch.randperm = lambda *a, **k: torch.from_numpy(next(replay).astype(np.int64))
    try:
        for _ in range(cfg["epochs"]):
            ref_te.train_mini_batch(model2, feats_l, labels_l, cfg["B"], Rec2(criterion_l), opt2, cfg["lamb"])
            ref_te.train_mini_batch(model2, feats_t, out_tt, cfg["B"], Rec2(criterion_t), opt2, 1 - cfg["lamb"])
    finally:
        torch.randperm = real_randperm
    out["noise.step_losses"] = np.float64(np.abs(np.asarray(losses2) - np.asarray(step_losses)).max())
    sda, sdb = model.state_dict(), model2.state_dict()
    for k in sda:
        if sda[k].ndim > 0:
            d = (sda[k] - sdb[k]).abs().double()
            out[f"noise.final.{k}"] = np.asarray([d.max().item(), d.mean().item()], np.float64)
    st2 = opt2.state_dict()["state"]
    for i, (pname, _) in enumerate(model.named_parameters()):
        for key in ("exp_avg", "exp_avg_sq"):
            out[f"noise.adam.{key}.{pname}"] = np.float64((st[i][key] - st2[i][key]).abs().max().item())
    o2, loss_e2, _ = ref_te.evaluate_mini_batch(model2, tf, tl, criterion_l, cfg["B"], evaluator)
    d = (o_all - o2).abs().double()
    out["noise.eval_out"] = np.asarray([d.max().item(), d.mean().item()], np.float64)
    out["noise.eval_loss"] = np.float64(abs(loss_e - loss_e2))

    # ---- an ABSOLUTE anchor for the post-training outputs: the same passes (same permutations, same initial state, same masks)
    # through the reference in float64.  |ref_fp32 - ref_fp64| is how far fp32 rounding alone moves the reference's own result;
    # tests/parity_rules.py holds an implementation to |impl - ref_fp64| <= 2 x that distance (max and mean), which -- unlike a
    # multiple of the fp32 self-noise -- cannot be met by a subtly biased implementation whose bias exceeds the rounding scale.
    model64 = ref_models.Model(conf).double()
    model64.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd0.items()})
    if p_drop > 0:
        model64.encoder.dropout = MaskDropout(p_drop, cfg["drop_base_seed"], L - 1)
    opt64 = torch.optim.Adam(model64.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
    replay = iter(perms)
    losses64 = []

    class Rec64(Rec):
        def __call__(self, o, y):
            l = self.crit(o, y)
            losses64.append(float(l.item()))
            return l

    tf64, tt64 = tf.double(), tt.double()
    torch.randperm = lambda *a, **k: torch.from_numpy(next(replay).astype(np.int64))
    try:
        for _ in range(cfg["epochs"]):
            ref_te.train_mini_batch(model64, tf64[til], labels_l, cfg["B"], Rec64(criterion_l), opt64, cfg["lamb"])
            ref_te.train_mini_batch(model64, tf64, tt64, cfg["B"], Rec64(criterion_t), opt64, 1 - cfg["lamb"])
    finally:
        torch.randperm = real_randperm
    o64, loss_e64, score_e64 = ref_te.evaluate_mini_batch(model64, tf64, tl, criterion_l, cfg["B"], evaluator)
    o64n = o64.numpy()
    out["f64.eval_out"] = o64n.copy() if cfg["full"] else np.asarray(o64n, np.float64).ravel()[::_stride[0]].copy()
    out["f64.eval_loss"] = np.float64(loss_e64)
    out["f64.eval_score"] = np.float64(score_e64)
    out["f64.step_losses"] = np.asarray(losses64, np.float64)
    d64 = (o_all.double() - o64).abs()
    out["f64.dist_eval_out"] = np.asarray([d64.max().item(), d64.mean().item()], np.float64)      # |ref_fp32 - ref_fp64|: max, mean
    out["f64.dist_eval_loss"] = np.float64(abs(loss_e - loss_e64))
    # a SECOND draw of the reference's fp32 rounding noise around the same fp64 result: the one-ulp-perturbed fp32 run above
    # (its distance to fp64 is several times the unperturbed run's on the wide students: 1 ulp decides Adam's first signs)
    d64p = (o2.double() - o64).abs()
    out["f64.dist_eval_out_perturbed"] = np.asarray([d64p.max().item(), d64p.mean().item()], np.float64)
    out["f64.dist_eval_loss_perturbed"] = np.float64(abs(loss_e2 - loss_e64))
    out["f64.dist_step_losses"] = np.float64(np.abs(np.asarray(losses64) - np.asarray(step_losses)).max())

    # ---- config + (small cases) inputs
    for k in ("B", "n", "n_l", "lamb", "lr", "wd", "epochs", "seed"):
        out[f"cfg.{k}"] = np.float64(cfg[k])
    out["cfg.dims"] = np.asarray(dims, np.int64)
    out["cfg.norm"] = np.asarray(norm)
    out["cfg.dropout"] = np.float64(p_drop)
    out["cfg.drop_base_seed"] = np.int64(cfg.get("drop_base_seed", 0))
    out["cfg.full"] = np.int64(1 if cfg["full"] else 0)
    out["cfg.sample_stride"] = np.int64(_stride[0])
    if cfg["full"]:
        out["in.feats"], out["in.labels"], out["in.out_t"], out["in.idx_l"] = feats, labels, out_t, idx_l
        for k, v in sd0.items():
            out[f"init.{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, f"student_{name}.npz"), **out)
    print(name, "steps", len(step_losses), "pass_means", pass_means, "size",
          os.path.getsize(os.path.join(HERE, f"student_{name}.npz")))


if __name__ == "__main__":
    _stub_modules()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repo root: oracle.dropout_mask
    torch.set_num_threads(1)
    only = sys.argv[1:]
    for name, cfg in CASES.items():
        if not only or name in only:
            run_case(name, cfg)
