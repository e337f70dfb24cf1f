"""`load_data` with the reference's contract (reference dataloader.py:42-58): returns
(g, labels, idx_train, idx_val, idx_test) with node features in g.ndata["feat"] -- but `g` is a
glnn_amd CSRGraph (CSR over destination rows), never a DGL graph.

Datasets
  * "synthetic-<name>[@scale]" with <name> in {cora, ogbn-arxiv, ogbn-products}: seeded stand-ins of the
    public shapes (glnn_amd/data.py) -- neither the datasets nor a network exist in this environment.
  * CPF citation/co-purchase graphs ("cora", "citeseer", "pubmed", "a-computer", "a-photo") from
    <data_path>/<name>.npz when the file is present: the same pipeline as reference dataloader.py:82-111 /
    data_preprocess.py: unweighted + undirected + no self-loops -> largest connected component ->
    per-class label-rate split -> graph = sparsity pattern of (A + I)  (the row-normalised weights are
    discarded by the reference too: only coordinates reach dgl.graph, dataloader.py:103-105).
  * `load_out_t` (reference dataloader.py:169-170): teacher log-probs from out.npz["arr_0"]."""
import os
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import torch

from . import data as synth
from .graph import CSRGraph

CPF_data = ["cora", "citeseer", "pubmed", "a-computer", "a-photo"]
OGB_data = ["ogbn-arxiv", "ogbn-products"]


def load_data(dataset, dataset_path, **kwargs):
    if dataset.startswith("synthetic-"):
        return load_synthetic_data(dataset, seed=kwargs.get("seed", 0))
    if dataset in CPF_data:
        return load_cpf_data(dataset, dataset_path, kwargs["seed"], kwargs["labelrate_train"], kwargs["labelrate_val"])
    if dataset in OGB_data:
        return load_ogb_data(dataset, dataset_path)
    raise ValueError(f"Unknown dataset: {dataset} (this build ingests CPF .npz files, ogbn-arxiv / ogbn-products through the "
                     "`ogb` package and synthetic-* shapes; the NonHom / BGNN loaders are outside the hot-path scope)")


def load_ogb_data(dataset, dataset_path):
    """ogbn-arxiv / ogbn-products (reference dataloader.py:61-79) WITHOUT dgl: `ogb.nodeproppred.NodePropPredDataset` is the
    library-agnostic form of the DglNodePropPredDataset the reference uses (same files, same split).  arxiv is made
    undirected exactly like the reference does: every edge gets its reverse appended (multi-edges kept, :74-76), then all
    self-loops are removed and one self-loop per node is added (:77); products is used as stored."""
    try:
        from ogb.nodeproppred import NodePropPredDataset
    except ImportError as e:
        raise ImportError(f"{dataset} needs the `ogb` package and the dataset files under {dataset_path} (neither ships with this "
                          f"image and there is no network here); use synthetic-{dataset} for a stand-in of the same shape") from e
    data = NodePropPredDataset(dataset, dataset_path)
    split = data.get_idx_split()
    graph, labels = data[0]
    n = int(graph["num_nodes"])
    src = torch.as_tensor(np.asarray(graph["edge_index"][0]), dtype=torch.int64)
    dst = torch.as_tensor(np.asarray(graph["edge_index"][1]), dtype=torch.int64)
    if dataset == "ogbn-arxiv":
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
        keep = src != dst
        loops = torch.arange(n, dtype=torch.int64)
        src, dst = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
    g = CSRGraph.from_edges(src, dst, n)
    g.ndata["feat"] = torch.as_tensor(np.asarray(graph["node_feat"]), dtype=torch.float32)
    labels = torch.as_tensor(np.asarray(labels)).squeeze().long()
    as_idx = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int64)
    return g, labels, as_idx(split["train"]), as_idx(split["valid"]), as_idx(split["test"])


def load_synthetic_data(dataset, seed=0):
    name = dataset[len("synthetic-"):]
    scale = 1.0
    if "@" in name:
        name, s = name.split("@")
        scale = float(s)
    g = synth.make_graph(name, seed=seed, device="cpu", scale=scale)
    feats, labels, _, (idx_train, idx_val, idx_test) = synth.make_node_data(name, seed=seed, device="cpu", n=g.n_dst)
    # give the synthetic labels SOME dependence on the features so that training is not pure noise
    w = torch.randn(feats.shape[1], int(labels.max()) + 1, generator=torch.Generator().manual_seed(seed + 7))
    labels = (feats @ w).argmax(1)
    g.ndata["feat"] = feats
    return g, labels, idx_train, idx_val, idx_test


# ---------------------------------------------------------------------------------------------- CPF
def _load_npz(path):
    with np.load(path, allow_pickle=True) as z:
        z = dict(z)
    adj = sp.csr_matrix((z["adj_data"], z["adj_indices"], z["adj_indptr"]), shape=z["adj_shape"])
    if "attr_data" in z:
        attr = sp.csr_matrix((z["attr_data"], z["attr_indices"], z["attr_indptr"]), shape=z["attr_shape"])
    else:
        attr = z["attr_matrix"]
    labels = z["labels"]
    return adj, attr, labels


def _standardize(adj, attr, labels):
    """unweighted, undirected, self-loop-free, largest connected component (dataloader.py:518-527)."""
    adj = adj.tocsr().copy()
    adj.data[:] = 1.0
    adj = adj + adj.T
    adj.data[:] = 1.0
    adj = adj.tolil()
    adj.setdiag(0)
    adj = adj.tocsr()
    adj.eliminate_zeros()
    _, comp = sp.csgraph.connected_components(adj)
    keep = np.flatnonzero(comp == np.argmax(np.bincount(comp)))
    adj = adj[keep][:, keep]
    return adj, attr[keep], labels[keep]


def _split_per_class(rs, onehot, per_class, forbidden=None):
    forb = set() if forbidden is None else set(int(i) for i in forbidden)
    parts = []
    for c in range(onehot.shape[1]):
        cand = [i for i in np.flatnonzero(onehot[:, c] > 0) if i not in forb]
        parts.append(rs.choice(cand, per_class, replace=False))
    return np.concatenate(parts)


def load_cpf_data(dataset, dataset_path, seed, labelrate_train, labelrate_val):
    path = Path.cwd().joinpath(dataset_path, f"{dataset}.npz")
    if not os.path.isfile(path):
        raise ValueError(f"{path} doesn't exist.")
    adj, attr, labels = _standardize(*_load_npz(path))
    classes = np.unique(labels)
    onehot = (labels[:, None] == classes[None, :]).astype(np.float32)
    rs = np.random.RandomState(seed)
    idx_train = _split_per_class(rs, onehot, labelrate_train)
    idx_val = _split_per_class(rs, onehot, labelrate_val, forbidden=idx_train)
    idx_test = np.setdiff1d(np.arange(len(labels)), np.concatenate([idx_train, idx_val]))
    feats = torch.from_numpy(np.asarray(attr.todense() if sp.issparse(attr) else attr, dtype=np.float32))
    y = torch.from_numpy(onehot.argmax(1).astype(np.int64))
    pat = (adj + sp.eye(adj.shape[0])).tocoo()            # sparsity pattern of normalize_adj(adj) = A + I
    g = CSRGraph.from_edges(torch.from_numpy(pat.row.astype(np.int64)), torch.from_numpy(pat.col.astype(np.int64)),
                            adj.shape[0])
    g.ndata["feat"] = feats
    return g, y, torch.from_numpy(idx_train).long(), torch.from_numpy(idx_val).long(), torch.from_numpy(idx_test).long()


def load_out_t(out_t_dir):
    return torch.from_numpy(np.load(Path(out_t_dir).joinpath("out.npz"))["arr_0"])
