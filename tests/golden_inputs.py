"""Deterministic regeneration of the golden cases' inputs (numpy legacy RandomState is bit-stable).
Mirrors make_inputs/make_state of tests/golden/make_student_golden.py, which produced the fixtures."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["bn_small", "nonorm_fullbatch", "arxiv_dims", "products_dims_narrow", "mlp3w4", "mlp3w8", "bn_small_dropout", "ln_small",
         "ln_house_dims"]


def make_inputs(seed, n, f, c, n_l):
    rs = np.random.RandomState(seed)
    feats = rs.standard_normal((n, f)).astype(np.float32)
    labels = rs.randint(0, c, size=n).astype(np.int64)
    t = rs.standard_normal((n, c)).astype(np.float32)
    t = t - t.max(1, keepdims=True)
    out_t = (t - np.log(np.exp(t.astype(np.float64)).sum(1, keepdims=True))).astype(np.float32)
    idx_l = rs.permutation(n)[:n_l].astype(np.int64)
    return feats, labels, out_t, idx_l


def make_state(seed, dims, norm):
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


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, f"student_{name}.npz"))
        z = self.z
        self.dims = [int(d) for d in z["cfg.dims"]]
        self.norm = str(z["cfg.norm"])
        self.B, self.n, self.n_l = int(z["cfg.B"]), int(z["cfg.n"]), int(z["cfg.n_l"])
        self.lamb, self.lr, self.wd = float(z["cfg.lamb"]), float(z["cfg.lr"]), float(z["cfg.wd"])
        self.epochs, self.seed = int(z["cfg.epochs"]), int(z["cfg.seed"])
        self.full = bool(int(z["cfg.full"]))
        self.dropout = float(z["cfg.dropout"]) if "cfg.dropout" in z.files else 0.0
        self.drop_base_seed = int(z["cfg.drop_base_seed"]) if "cfg.drop_base_seed" in z.files else 0
        self.stride = int(z["cfg.sample_stride"])
        self.feats, self.labels, self.out_t, self.idx_l = make_inputs(self.seed, self.n, self.dims[0], self.dims[-1], self.n_l)
        self.sd0 = make_state(self.seed, self.dims, self.norm)
        if self.full:   # the stored copies must equal the regenerated ones
            assert np.array_equal(z["in.feats"], self.feats) and np.array_equal(z["in.idx_l"], self.idx_l)
            for k, v in self.sd0.items():
                assert np.array_equal(z[f"init.{k}"], v)
        self.perms = [z[f"perm_{i}"].astype(np.int64) for i in range(int(z["num_perms"]))]
        self.param_names = [f"encoder.layers.{i}.{s}" for i in range(len(self.dims) - 1) for s in ("weight", "bias")]
        if self.norm in ("batch", "layer"):
            self.param_names += [f"encoder.norms.{i}.{s}" for i in range(len(self.dims) - 2) for s in ("weight", "bias")]

    def masks(self, step, rows):
        """Keep-masks of optimiser step `step` (1-based) for every hidden layer (None without dropout): the ones the golden
        run fed to the reference's MLP (oracle/dropout_mask.py restates the library's counter-based mask)."""
        if self.dropout == 0.0:
            return None
        from oracle.dropout_mask import engine_seed, keep_mask
        return [keep_mask(rows, self.dims[l + 1], self.dropout, engine_seed(self.drop_base_seed, step, l)).astype(np.float32)
                for l in range(len(self.dims) - 2)]

    def view(self, a):
        """How an array is stored in this fixture (full, or strided sample)."""
        a = np.asarray(a)
        if self.full or a.ndim == 0:
            return a
        return a.astype(np.float32).ravel()[:: self.stride]


def teacher_composition():
    """tests/golden/teacher_composition.npz: outputs of the reference's own SAGE.inference / GCN.forward_fitnet
    (dgl layers stubbed by torch.sparse stand-ins; see make_teacher_golden.py).  Returns {prefix: dict}."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teacher_composition.npz"))
    out = {"sage": {"sd": {}}, "gcn": {"sd": {}}}
    for k in z.files:
        fam, rest = k.split(".", 1)
        if rest.startswith("sd."):
            out[fam]["sd"][rest[3:]] = z[k]
        else:
            out[fam][rest] = z[k]
    return out


def sage_layers_from_sd(sd, num_layers, batch_norm=True):
    layers = [dict(weight=sd[f"encoder.layers.{i}.fc_neigh.weight"], bias=sd[f"encoder.layers.{i}.fc_neigh.bias"])
              for i in range(num_layers)]
    norms = [dict(weight=sd[f"encoder.norms.{i}.weight"], bias=sd[f"encoder.norms.{i}.bias"],
                  running_mean=sd[f"encoder.norms.{i}.running_mean"], running_var=sd[f"encoder.norms.{i}.running_var"])
             for i in range(num_layers - 1)] if batch_norm else None
    return layers, norms


def teacher_training():
    """tests/golden/teacher_training.npz (make_teacher_train_golden.py): the reference's train_sage over fixed blocks (two
    norm variants) and its full-graph GCN `train`.  Returns the raw npz plus helpers to rebuild the batches."""
    z = np.load(os.path.join(GOLDEN_DIR, "teacher_training.npz"))
    batches = []
    for b in range(3):
        blocks = [(z[f"sage.b{b}.l{l}.indptr"], z[f"sage.b{b}.l{l}.indices"], int(z[f"sage.b{b}.l{l}.n_src"])) for l in range(3)]
        batches.append((z[f"sage.b{b}.input_nodes"], z[f"sage.b{b}.output_nodes"], blocks))
    return z, batches


def sub_dict(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}
