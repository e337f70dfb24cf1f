"""CPU tests of the host-side logic that surrounds the hot path: config merge, index splits, the CSR graph
container and its DGL-like surface, the full-neighbour chunk loader, the CPF .npz ingestion, CLI flags."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_config_merge_yaml_overrides_cli():
    from glnn_amd import cli, utils
    conf = utils.get_training_config(os.path.join(ROOT, "train.conf.yaml"), "MLP3w8", "ogbn-products")
    assert conf == dict(num_layers=3, hidden_dim=2048, dropout_ratio=0.2, learning_rate=0.01, weight_decay=0,
                        norm_type="batch", batch_size=4096, model_name="MLP3w8")
    conf = utils.get_training_config(os.path.join(ROOT, "train.conf.yaml"), "GCN", "cora")
    assert conf["hidden_dim"] == 64 and conf["num_layers"] == 2            # dataset section over `global`
    args = cli.get_student_args(["--num_layers", "5", "--hidden_dim", "77", "--dataset", "ogbn-arxiv", "--student", "MLP3w4"])
    merged = dict(args.__dict__, **utils.get_training_config(args.model_config_path, args.student, args.dataset))
    assert merged["num_layers"] == 3 and merged["hidden_dim"] == 1024      # YAML beats the CLI flags (train_student.py:264-270)
    assert merged["lamb"] == 0 and merged["batch_size"] == 512


def test_cli_flags_match_reference_surface():
    from glnn_amd import cli
    t = cli.get_teacher_args([])
    s = cli.get_student_args([])
    for name, default in dict(device=-1, seed=0, log_level=20, output_path="outputs", num_exp=1, exp_setting="tran",
                              eval_interval=1, dataset="cora", data_path="./data", labelrate_train=20, labelrate_val=30,
                              split_idx=0, teacher="SAGE", num_layers=2, hidden_dim=128, dropout_ratio=0, norm_type="none",
                              batch_size=512, fan_out="5,5", num_workers=0, learning_rate=0.01, weight_decay=0.0005,
                              max_epoch=500, patience=50, feature_noise=0, split_rate=0.2, feature_aug_k=0).items():
        assert getattr(t, name) == default and getattr(s, name) == default, name
    assert s.student == "MLP" and s.lamb == 0 and s.out_t_path == "outputs"
    assert not hasattr(t, "student") and t.console_log is False and t.save_results is False and t.compute_min_cut is False


def test_graph_split_index_algebra():
    from glnn_amd import utils
    idx_train, idx_val, idx_test = torch.arange(0, 10), torch.arange(10, 15), torch.arange(15, 115)
    o_tr, o_va, o_te, idx_obs, idx_ind = utils.graph_split(idx_train, idx_val, idx_test, 0.2, seed=3)
    assert len(idx_ind) == 20 and len(idx_obs) == 10 + 5 + 80
    assert torch.equal(idx_obs[o_tr], idx_train) and torch.equal(idx_obs[o_va], idx_val)
    assert set(idx_obs[o_te].tolist()) | set(idx_ind.tolist()) == set(idx_test.tolist())
    assert not (set(idx_obs.tolist()) & set(idx_ind.tolist()))
    again = utils.graph_split(idx_train, idx_val, idx_test, 0.2, seed=3)
    assert torch.equal(again[4], idx_ind)                                  # seeded: reproducible


def test_csr_graph_surface():
    from glnn_amd.graph import CSRGraph, FullNeighborLoader
    src = torch.tensor([0, 2, 2, 3, 1, 4, 4]); dst = torch.tensor([1, 1, 1, 3, 0, 2, 0])
    g = CSRGraph.from_edges(src, dst, 5)
    assert g.num_nodes() == g.number_of_nodes() == 5 and g.number_of_edges() == 7 and g.num_dst_nodes() == 5
    assert g.in_degrees().tolist() == [2, 3, 1, 1, 0] and g.out_degrees().tolist() == [1, 1, 2, 1, 2]
    assert g.int() is g and g.to("cpu") is g and g.create_formats_() is None
    rev = g.reverse()
    assert rev.in_degrees().tolist() == g.out_degrees().tolist()
    assert len(FullNeighborLoader(g, 2)) == 3
    with pytest.raises(RuntimeError):
        next(iter(FullNeighborLoader(g, 2)))                                # blocks are built on the GPU (tests/test_teacher_gpu.py)
    sub = g.subgraph(torch.tensor([1, 2, 0]))                               # relabel: 1->0, 2->1, 0->2
    dense = torch.zeros(3, 3)
    for v in range(3):
        for u in sub.indices[sub.indptr[v]:sub.indptr[v + 1]].tolist():
            dense[v, u] += 1
    assert dense.tolist() == [[0, 2, 1], [0, 0, 0], [1, 0, 0]]
    shard = g.row_range(1, 4)
    assert shard.n_dst == 3 and shard.n_src == 5 and shard.num_edges() == 5


def test_synthetic_shapes_are_the_public_statistics():
    from glnn_amd import data
    g = data.make_graph("cora", seed=0)
    assert g.n_dst == 2485 and g.num_edges() == 12623
    g = data.make_graph("ogbn-arxiv", seed=0)
    assert g.n_dst == 169343 and g.num_edges() == 2501829
    g2 = data.make_graph("ogbn-arxiv", seed=0)
    assert torch.equal(g.indices, g2.indices)                               # seeded
    small = data.make_graph("ogbn-products", seed=0, scale=0.002)
    assert small.num_edges() == 2 * int(61859140 * 0.002)
    feats, labels, out_t, (tr, va, te) = data.make_node_data("ogbn-arxiv", n=1000)
    assert feats.shape == (1000, 128) and out_t.shape == (1000, 40) and len(tr) + len(va) + len(te) == 1000
    assert torch.allclose(out_t.exp().sum(1), torch.ones(1000), atol=1e-5)


def test_cpf_npz_ingestion(tmp_path, monkeypatch):
    """A tiny CPF-format file through load_data: LCC, self-loop removal, symmetrisation, A+I pattern, split."""
    from glnn_amd.dataloader import load_data
    rs = np.random.RandomState(0)
    n = 60
    a = sp.random(n, n, density=0.08, random_state=rs, format="csr")
    a = a + sp.eye(n, format="csr")                                         # self-loops must be dropped
    a[50:, :] = 0; a[:, 50:] = 0                                            # nodes 50.. isolated -> outside the LCC
    a = sp.csr_matrix(a); a.eliminate_zeros()
    x = sp.random(n, 12, density=0.3, random_state=rs, format="csr")
    y = rs.randint(0, 3, n)
    os.makedirs(tmp_path / "data")
    np.savez(tmp_path / "data" / "cora.npz", adj_data=a.data, adj_indices=a.indices, adj_indptr=a.indptr, adj_shape=a.shape,
             attr_data=x.data, attr_indices=x.indices, attr_indptr=x.indptr, attr_shape=x.shape, labels=y)
    monkeypatch.chdir(tmp_path)
    g, labels, tr, va, te = load_data("cora", "./data", seed=1, labelrate_train=3, labelrate_val=2, split_idx=0)
    n_lcc = g.num_nodes()
    assert n_lcc <= 50 and g.ndata["feat"].shape == (n_lcc, 12) and len(labels) == n_lcc
    dense = torch.zeros(n_lcc, n_lcc)
    for v in range(n_lcc):
        for u in g.indices[g.indptr[v]:g.indptr[v + 1]].tolist():
            dense[v, u] += 1
    assert torch.equal(dense, dense.t()) and float(dense.diag().min()) == 1 and float(dense.max()) == 1
    assert len(tr) == 9 and len(va) == 6 and len(set(tr.tolist()) & set(va.tolist())) == 0
    assert len(tr) + len(va) + len(te) == n_lcc
    with pytest.raises(ValueError):
        load_data("pokec", "./data", seed=0, labelrate_train=1, labelrate_val=1)


def test_evaluator_is_plain_argmax_accuracy():
    from glnn_amd import utils
    ev = utils.get_evaluator("ogbn-arxiv")
    out = torch.tensor([[0.1, 0.9], [0.8, 0.2], [0.3, 0.7]])
    assert abs(ev(out, torch.tensor([1, 0, 0])) - 2 / 3) < 1e-7


# ---------------------------------------------------------------------------------- reference-generated goldens
def _host_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "host_logic.npz"))


def test_graph_split_and_idx_split_equal_reference_bit_for_bit():
    """tests/golden/host_logic.npz holds what the reference's utils.graph_split / idx_split returned (utils.py:88-127)."""
    from glnn_amd import utils
    z = _host_golden()
    idx_train, idx_val, idx_test = (torch.from_numpy(z[f"split.{k}"]) for k in ("idx_train", "idx_val", "idx_test"))
    for rate, seed in ((0.2, 0), (0.5, 3)):
        got = utils.graph_split(idx_train, idx_val, idx_test, rate, seed)
        for name, t in zip(("obs_idx_train", "obs_idx_val", "obs_idx_test", "idx_obs", "idx_test_ind"), got):
            want = z[f"graph_split.r{rate}_s{seed}.{name}"]
            assert t.dtype == torch.int64 and np.array_equal(t.numpy(), want), (rate, seed, name)
    a, b = utils.idx_split(idx_test, 0.37, seed=9)
    assert np.array_equal(a.numpy(), z["idx_split.a"]) and np.array_equal(b.numpy(), z["idx_split.b"])


@pytest.mark.parametrize("seed,ltr,lva", [(0, 6, 9), (4, 3, 5)])
def test_cpf_ingestion_equals_reference_bit_for_bit(seed, ltr, lva, monkeypatch):
    """The reference's own load_cpf_data (dataloader.py:82-111: load_npz_to_sparse_graph -> standardize -> LCC ->
    binarize_labels -> get_train_val_test_split / sample_per_class -> normalize_adj pattern -> dgl.graph) was run on
    tests/golden/cpf/tiny_cpf.npz; this loader must return the same graph, features, labels and splits exactly."""
    from glnn_amd.dataloader import load_cpf_data
    from glnn_amd.graph import CSRGraph
    z = _host_golden()
    tag = f"cpf.s{seed}_{ltr}_{lva}"
    monkeypatch.chdir(os.path.join(ROOT, "tests", "golden"))
    g, labels, tr, va, te = load_cpf_data("tiny_cpf", "cpf", seed, ltr, lva)
    n = int(z[f"{tag}.num_nodes"])
    assert g.num_nodes() == n
    want = CSRGraph.from_edges(torch.from_numpy(z[f"{tag}.row"].astype(np.int64)), torch.from_numpy(z[f"{tag}.col"].astype(np.int64)), n)
    assert torch.equal(g.indptr, want.indptr) and torch.equal(g.indices, want.indices)     # dgl.graph((row, col)): row -> col
    assert np.array_equal(g.ndata["feat"].numpy(), z[f"{tag}.feat"]) and g.ndata["feat"].dtype == torch.float32
    assert np.array_equal(labels.numpy(), z[f"{tag}.labels"]) and labels.dtype == torch.int64
    for got, name in ((tr, "idx_train"), (va, "idx_val"), (te, "idx_test")):
        assert got.dtype == torch.int64 and np.array_equal(got.numpy(), z[f"{tag}.{name}"]), name


def test_oracle_feature_prop_and_min_cut_vs_reference_golden():
    """utils.feature_prop (utils.py:171-189; its normalisation is reference code, copy_u/sum a scipy stand-in) and the dense
    compute_min_cut_loss (utils.py:159-168) as the reference computed them, vs the CPU oracle's restatement."""
    from oracle import teacher_oracle as to
    z = _host_golden()
    ip, ix = z["mincut.indptr"], z["mincut.indices"]
    for k in (1, 3):
        np.testing.assert_allclose(to.feature_prop(ip, ix, z["fprop.feats"], k), z[f"fprop.k{k}"], atol=1e-5, rtol=1e-5)
    assert abs(to.min_cut_loss(ip, ix, z["mincut.logp"]) - float(z["mincut.value"])) < 1e-5


def test_ogb_loader_arxiv_symmetrise_and_self_loops(monkeypatch):
    """load_ogb_data against a stand-in `ogb` package (the real one and its datasets are absent here): ogbn-arxiv gets
    reverse edges (multi-edges kept) and exactly one self-loop per node (reference dataloader.py:74-77); ogbn-products is
    used as stored; a missing `ogb` raises ImportError, not ValueError."""
    import sys
    import types
    from glnn_amd.dataloader import load_data
    edge_index = np.array([[0, 0, 1, 2, 3, 3], [1, 1, 2, 2, 0, 3]])           # a duplicate edge 0->1, two self-loops (2, 3)

    class FakeDataset:
        def __init__(self, name, root):
            self.name = name

        def get_idx_split(self):
            return {"train": np.array([0, 1]), "valid": np.array([2]), "test": np.array([3])}

        def __getitem__(self, i):
            return {"edge_index": edge_index, "node_feat": np.arange(8, dtype=np.float32).reshape(4, 2), "num_nodes": 4}, np.array([[0], [1], [0], [1]])

    ogb = types.ModuleType("ogb")
    npp = types.ModuleType("ogb.nodeproppred")
    npp.NodePropPredDataset = FakeDataset
    ogb.nodeproppred = npp
    monkeypatch.setitem(sys.modules, "ogb", ogb)
    monkeypatch.setitem(sys.modules, "ogb.nodeproppred", npp)
    g, labels, tr, va, te = load_data("ogbn-arxiv", "./data")
    dense = torch.zeros(4, 4)
    for v in range(4):
        for u in g.indices[g.indptr[v]:g.indptr[v + 1]].tolist():
            dense[v, u] += 1
    #                 in-edges of:   0            1            2            3        (row v, column u: edge u -> v)
    assert dense.tolist() == [[1, 2, 0, 1], [2, 1, 1, 0], [0, 1, 1, 0], [1, 0, 0, 1]]
    assert g.number_of_edges() == 2 * 4 + 4 and labels.tolist() == [0, 1, 0, 1] and g.ndata["feat"].shape == (4, 2)
    assert tr.tolist() == [0, 1] and va.tolist() == [2] and te.tolist() == [3]
    g2, *_ = load_data("ogbn-products", "./data")
    assert g2.number_of_edges() == 6                                           # untouched
    monkeypatch.setitem(sys.modules, "ogb.nodeproppred", None)
    with pytest.raises(ImportError):
        load_data("ogbn-arxiv", "./data")


def test_randperm_cpu_is_torch_randperm_on_one_thread():
    """ops.randperm_cpu: the reference's mini-batch permutation (train_and_eval.py:66) drawn with one intra-op thread -- the same
    numbers from the same global generator state, and the thread count is put back."""
    import torch
    from glnn_amd import ops
    k = torch.get_num_threads()
    torch.manual_seed(123)
    want = [torch.randperm(n) for n in (1, 7, 90941)]
    torch.manual_seed(123)
    got = [ops.randperm_cpu(n) for n in (1, 7, 90941)]
    assert torch.get_num_threads() == k
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_as_feat_remembers_the_padded_copy_of_an_unchanged_tensor():
    """ops.as_feat pads rows that are not float4-addressable (cora's 1433 features, penn94's 4814).  The copy of a large matrix is reused
    while the SAME tensor object comes back unmodified (identity + in-place version counter), so the reference's per-epoch train / evaluate
    calls do not re-pad the feature matrix; any in-place change, or another tensor at the same address, gets a fresh copy."""
    import torch
    from glnn_amd import ops
    t = torch.randn(1500, 1433)
    a = ops.as_feat(t)
    assert a.stride(0) % 4 == 0 and torch.equal(a[:, :1433], t) and ops.as_feat(t) is a
    t.mul_(2.0)
    b = ops.as_feat(t)
    assert b is not a and torch.equal(b[:, :1433], t)
    u = torch.randn(1500, 1433)
    c = ops.as_feat(u)
    assert ops.as_feat(t) is b and ops.as_feat(u) is c                    # two entries
    assert ops.as_feat(t.detach()) is not b                               # another tensor object: never trusted
    small = torch.randn(10, 7)
    assert ops.as_feat(small) is not ops.as_feat(small)                   # small matrices are not kept
    al = torch.randn(64, 128)
    assert ops.as_feat(al) is al


def test_default_nllloss_is_evaluated_by_gather_and_mean_and_everything_else_by_the_callable():
    """evaluate / evaluate_mini_batch keep the reference's signature (the criterion is the caller's callable, train_and_eval.py:89-136); a
    plain default torch.nn.NLLLoss over >= 65536 rows is computed as -mean(out[i, y_i]) (torch's CUDA kernel reduces in one workgroup),
    anything else -- weights, another reduction, ignore_index hits, few rows, other callables -- is passed through untouched."""
    import torch
    from glnn_amd import train_and_eval as te
    out = torch.log_softmax(torch.randn(70000, 5), 1)
    y = torch.randint(0, 5, (70000,))
    assert abs(float(te._apply_criterion(torch.nn.NLLLoss(), out, y)) - float(torch.nn.NLLLoss()(out, y))) < 1e-5
    y2 = y.clone(); y2[3] = -100
    assert float(te._apply_criterion(torch.nn.NLLLoss(), out, y2)) == float(torch.nn.NLLLoss()(out, y2))
    for crit in (torch.nn.NLLLoss(reduction="sum"), torch.nn.NLLLoss(weight=torch.rand(5)), lambda o, l: o.sum() * 0 + 7.0):
        assert float(te._apply_criterion(crit, out, y)) == float(crit(out, y))
    assert float(te._apply_criterion(torch.nn.NLLLoss(), out[:100], y[:100])) == float(torch.nn.NLLLoss()(out[:100], y[:100]))
