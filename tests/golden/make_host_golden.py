"""Golden vectors for the HOST-SIDE pieces either side of the hot path, produced by the reference's own Python
(SURVEY.md 8f rows 3 and 4): index algebra must be bit-exact, floats within 1e-4.

What runs here is reference code, imported from /root/reference (never copied):
  * utils.idx_split / utils.graph_split                  reference utils.py:88-127   (inductive split, torch.randperm)
  * dataloader.load_cpf_data and everything under it     reference dataloader.py:82-111, 518-527, 534-590, 593-700;
    (load_npz_to_sparse_graph, SparseGraph.standardize,  data_preprocess.py:32-41, 44-50, 53-80, 138-170
     largest_connected_components, binarize_labels,
     sample_per_class, get_train_val_test_split, normalize_adj)
  * utils.compute_min_cut_loss                           reference utils.py:159-168  (dense tr(S'AS)/tr(S'DS))
  * utils.feature_prop                                   reference utils.py:171-189  (its normalisation is reference code;
                                                         only update_all(copy_u, sum) is dgl -> a scipy stand-in)
`dgl` is absent here: `dgl.graph((row, col))` is replaced by a recorder that keeps the coordinates it was given, and the
graph object handed to compute_min_cut_loss / feature_prop is a small stand-in exposing adj(), in_degrees(), num_nodes(),
ndata and update_all(copy_u, sum) with dgl's published meaning (message = source feature, reduce = sum at the destination).
category_encoders / google_drive_downloader / ogb (top-level imports of dataloader.py) are empty import-only stubs.

    python tests/golden/make_host_golden.py          (build container only; writes host_logic.npz and cpf/tiny_cpf.npz)
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from graphgen import random_graph  # noqa: E402


class RecordedGraph:
    """What `dgl.graph((row, col))` returns here: the coordinates, plus the few members the reference touches."""

    def __init__(self, edges, num_nodes=None):
        self.row, self.col = (np.asarray(e).astype(np.int64) for e in edges)        # edge row[i] -> col[i]
        self.n = int(max(self.row.max(), self.col.max()) + 1) if num_nodes is None else int(num_nodes)
        self.ndata = {}

    def num_nodes(self):
        return self.n

    def in_degrees(self):
        return torch.from_numpy(np.bincount(self.col, minlength=self.n))

    def adj(self):
        idx = torch.from_numpy(np.stack([self.row, self.col]))
        return torch.sparse_coo_tensor(idx, torch.ones(len(self.row)), (self.n, self.n))

    def update_all(self, message, reduce):
        assert message == ("copy_u", "h", "m") and reduce == ("sum", "m", "h")
        a = sp.csr_matrix((np.ones(len(self.row), np.float64), (self.col, self.row)), shape=(self.n, self.n))   # dst x src
        self.ndata["h"] = torch.from_numpy((a @ self.ndata["h"].double().numpy()).astype(np.float32))


def _stub_modules():
    dgl = types.ModuleType("dgl")
    dgl.graph = lambda edges, **kw: RecordedGraph(edges)
    dgl_nn = types.ModuleType("dgl.nn")
    for n in ("GraphConv", "SAGEConv", "APPNPConv", "GATConv"):
        setattr(dgl_nn, n, type(n, (), {}))
    dgl_fn = types.ModuleType("dgl.function")
    dgl_fn.copy_u = lambda a, b: ("copy_u", a, b)
    dgl_fn.sum = lambda a, b: ("sum", a, b)
    dgl_data = types.ModuleType("dgl.data")
    dgl_data_utils = types.ModuleType("dgl.data.utils")
    dgl_data_utils.load_graphs = None
    dgl.nn, dgl.function, dgl.data = dgl_nn, dgl_fn, dgl_data
    dgl_data.utils = dgl_data_utils
    ogb = types.ModuleType("ogb")
    ogb_npp = types.ModuleType("ogb.nodeproppred")
    ogb_npp.Evaluator = type("Evaluator", (), {})
    ogb_npp.DglNodePropPredDataset = type("DglNodePropPredDataset", (), {})
    ogb.nodeproppred = ogb_npp
    ce = types.ModuleType("category_encoders")
    ce.CatBoostEncoder = type("CatBoostEncoder", (), {})
    gdd = types.ModuleType("google_drive_downloader")
    gdd.GoogleDriveDownloader = type("GoogleDriveDownloader", (), {})
    sys.modules.update({"dgl": dgl, "dgl.nn": dgl_nn, "dgl.function": dgl_fn, "dgl.data": dgl_data, "dgl.data.utils": dgl_data_utils,
                        "ogb": ogb, "ogb.nodeproppred": ogb_npp, "category_encoders": ce, "google_drive_downloader": gdd})


def make_tiny_cpf(path):
    """A CPF-format .npz (the layout load_npz_to_sparse_graph reads, reference dataloader.py:534-590) with everything the
    standardisation has to deal with: directed + weighted edges, self-loops, a duplicate-direction pair, three small
    components next to the big one, sparse bag-of-words attributes, 5 classes with string-free integer labels."""
    rs = np.random.RandomState(11)
    n, f, c = 360, 50, 5
    comp = np.zeros(n, np.int64)
    comp[300:330], comp[330:350], comp[350:] = 1, 2, 3          # node blocks that stay disconnected from the big one
    src, dst = [], []
    for k, (lo, hi, m) in enumerate([(0, 300, 900), (300, 330, 60), (330, 350, 30), (350, 360, 12)]):
        src.append(rs.randint(lo, hi, m)); dst.append(rs.randint(lo, hi, m))
    src, dst = np.concatenate(src), np.concatenate(dst)
    loops = rs.choice(n, 25, replace=False)
    src, dst = np.concatenate([src, loops]), np.concatenate([dst, loops])
    w = rs.uniform(0.5, 3.0, len(src))
    adj = sp.coo_matrix((w, (src, dst)), shape=(n, n)).tocsr()       # duplicates summed -> weights != 1
    attr = sp.random(n, f, density=0.15, random_state=rs, format="csr", dtype=np.float32)
    attr.data[:] = 1.0
    labels = rs.randint(0, c, n).astype(np.int64)
    np.savez(path, adj_data=adj.data, adj_indices=adj.indices, adj_indptr=adj.indptr, adj_shape=np.asarray(adj.shape),
             attr_data=attr.data, attr_indices=attr.indices, attr_indptr=attr.indptr, attr_shape=np.asarray(attr.shape),
             labels=labels)


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import dataloader as ref_dl          # noqa: reference modules, unmodified
    import utils as ref_utils            # noqa
    torch.set_num_threads(1)
    out = {}

    # ---- idx_split / graph_split (utils.py:88-127) --------------------------------------------------------------------
    rs = np.random.RandomState(5)
    perm = rs.permutation(1000)
    idx_train, idx_val, idx_test = (torch.from_numpy(perm[a:b].astype(np.int64)) for a, b in ((0, 80), (80, 200), (200, 1000)))
    for rate, seed in ((0.2, 0), (0.5, 3)):
        res = ref_utils.graph_split(idx_train, idx_val, idx_test, rate, seed)
        for name, t in zip(("obs_idx_train", "obs_idx_val", "obs_idx_test", "idx_obs", "idx_test_ind"), res):
            out[f"graph_split.r{rate}_s{seed}.{name}"] = t.numpy()
    a, b = ref_utils.idx_split(idx_test, 0.37, seed=9)
    out["idx_split.a"], out["idx_split.b"] = a.numpy(), b.numpy()
    out["split.idx_train"], out["split.idx_val"], out["split.idx_test"] = idx_train.numpy(), idx_val.numpy(), idx_test.numpy()

    # ---- the whole CPF ingestion (dataloader.py:82-111) on a synthetic CPF file ---------------------------------------
    os.makedirs(os.path.join(HERE, "cpf"), exist_ok=True)
    cpf_path = os.path.join(HERE, "cpf", "tiny_cpf.npz")
    make_tiny_cpf(cpf_path)
    cwd = os.getcwd()
    os.chdir(HERE)                       # load_cpf_data resolves Path.cwd()/dataset_path/<name>.npz
    try:
        for seed, ltr, lva in ((0, 6, 9), (4, 3, 5)):
            g, labels, i_tr, i_va, i_te = ref_dl.load_cpf_data("tiny_cpf", "cpf", seed, ltr, lva)
            tag = f"cpf.s{seed}_{ltr}_{lva}"
            out[f"{tag}.row"], out[f"{tag}.col"] = g.row.astype(np.int32), g.col.astype(np.int32)
            out[f"{tag}.num_nodes"] = np.int64(g.num_nodes())
            out[f"{tag}.feat"] = g.ndata["feat"].numpy()
            out[f"{tag}.labels"] = labels.numpy()
            out[f"{tag}.idx_train"], out[f"{tag}.idx_val"], out[f"{tag}.idx_test"] = i_tr.numpy(), i_va.numpy(), i_te.numpy()
    finally:
        os.chdir(cwd)

    # ---- compute_min_cut_loss (utils.py:159-168) and feature_prop (utils.py:171-189) ----------------------------------
    n, c = 220, 7
    indptr, indices = random_graph(n, 6, seed=31, power=0.5, isolated=4, hub=90)       # CSR over destinations, multi-edges
    dst = np.repeat(np.arange(n), np.diff(indptr))
    g = RecordedGraph((indices.astype(np.int64), dst), num_nodes=n)
    rs = np.random.RandomState(31)
    logp = torch.log_softmax(torch.from_numpy(rs.standard_normal((n, c)).astype(np.float32)), dim=1)
    out["mincut.indptr"], out["mincut.indices"], out["mincut.logp"] = indptr, indices, logp.numpy()
    out["mincut.value"] = np.float64(ref_utils.compute_min_cut_loss(g, logp))
    feats = torch.from_numpy(rs.standard_normal((n, 19)).astype(np.float32))
    out["fprop.feats"] = feats.numpy()
    for k in (1, 3):
        out[f"fprop.k{k}"] = ref_utils.feature_prop(feats.clone(), g, k).numpy()

    path = os.path.join(HERE, "host_logic.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; cpf file", os.path.getsize(cpf_path), "bytes")


if __name__ == "__main__":
    main()
