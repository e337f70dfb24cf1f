"""GPU parity of the teacher TRAINING path: the device-side block builder and CSR transpose
(integer work: bit-exact against numpy restatements), the neighbour sampler's statistics, and TeacherEngine's
forward + loss + backward + Adam against the golden vectors the reference's own train_sage / train produced
(tests/golden/teacher_training.npz) and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from golden_inputs import sub_dict, teacher_training
from graphgen import random_graph
from oracle import teacher_train_oracle as tt

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _graph(indptr, indices, n_src=None):
    from glnn_amd.graph import CSRGraph
    n = len(indptr) - 1
    return CSRGraph(torch.from_numpy(np.asarray(indptr, np.int64)).to(DEV), torch.from_numpy(np.asarray(indices, np.int32)).to(DEV), n,
                    n if n_src is None else n_src)


# ------------------------------------------------------------------------------------------------ integer kernels
@pytest.mark.parametrize("n,deg,kw", [(1000, 7, dict(power=0.6, hub=700, isolated=9)), (257, 3, dict(symmetric=True, self_loops=True)),
                                      (5000, 20, dict(power=0.8, hub=4000)), (64, 1, {}), (3, 0.4, {})])
@pytest.mark.parametrize("add_self", [False, True])
def test_csr_transpose_equals_numpy_bit_for_bit(n, deg, kw, add_self):
    """glnn_csr_transpose: rows = original sources, entries = destinations, sorted within a row (multi-edges kept);
    add_self appends u <- u for every destination.  Hub rows (hundreds to thousands of entries) take the long-row sort."""
    from glnn_amd import ops
    indptr, indices = random_graph(n, deg, seed=n, **kw)
    nnz = int(indptr[-1])
    t_ip, t_ix = ops.csr_transpose(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n, n, nnz, add_self)
    dst = np.repeat(np.arange(n), np.diff(indptr))
    src = indices.astype(np.int64)
    if add_self:
        src, dst = np.concatenate([src, np.arange(n)]), np.concatenate([dst, np.arange(n)])
    order = np.lexsort((dst, src))
    want_ix = dst[order].astype(np.int32)
    want_ip = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=want_ip[1:])
    assert np.array_equal(t_ip.cpu().numpy(), want_ip)
    assert np.array_equal(t_ix.cpu().numpy(), want_ix)
    again = ops.csr_transpose(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n, n, nnz, add_self)
    assert torch.equal(again[1], t_ix)                       # deterministic run to run


def test_csr_transpose_of_a_block_and_empty_inputs():
    from glnn_amd import ops
    # a block: 3 destinations, 6 sources
    indptr = np.array([0, 2, 2, 5], np.int64)
    indices = np.array([4, 1, 5, 5, 0], np.int32)
    t_ip, t_ix = ops.csr_transpose(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), 3, 6, 5, True)
    assert t_ip.tolist() == [0, 2, 4, 5, 5, 6, 8]
    assert t_ix.tolist() == [0, 2, 0, 1, 2, 0, 2, 2]
    t_ip, t_ix = ops.csr_transpose(torch.zeros(5, dtype=torch.int64, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV), 4, 4, 0, False)
    assert t_ip.tolist() == [0] * 5 and t_ix.numel() == 0


def _first_appearance_block(seeds, rows):
    """numpy restatement of glnn_block_build: rows[i] = global source ids of destination i, in order."""
    pos = {int(v): i for i, v in enumerate(seeds)}
    input_nodes = [int(v) for v in seeds]
    indptr, local = [0], []
    for r in rows:
        for u in r:
            u = int(u)
            if u not in pos:
                pos[u] = len(input_nodes)
                input_nodes.append(u)
            local.append(pos[u])
        indptr.append(len(local))
    return np.asarray(indptr, np.int64), np.asarray(local, np.int32), np.asarray(input_nodes, np.int64)


@pytest.mark.parametrize("known_ids", [False, True])
@pytest.mark.parametrize("ns", [1, 37, 1000, 5000])
def test_block_build_full_neighbourhood_equals_numpy(ns, known_ids):
    """known_ids: the builder is told the size of the id universe (glnn_block_build_ids) and indexes its tables by the node id itself
    whenever that universe is no larger than the frontier's hash table would be (here: ns >= 1000) -- identical blocks either way."""
    from glnn_amd import ops
    n = 6000
    indptr, indices = random_graph(n, 8, seed=5, power=0.7, hub=3000, isolated=20)
    rs = np.random.RandomState(ns)
    seeds = rs.permutation(n)[:ns].astype(np.int64)
    rows = [indices[indptr[v]:indptr[v + 1]] for v in seeds]
    w_ip, w_ix, w_in = _first_appearance_block(seeds, rows)
    g_ip, g_ix = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
    ip, ix, gix, inp, nnz, n_src = ops.block_build(torch.from_numpy(seeds).to(DEV), g_ip, g_ix, nnz_cap=int(w_ip[-1]) + 11, want_global=True,
                                                   n_nodes=n if known_ids else 0)
    assert nnz == w_ip[-1] and n_src == len(w_in)
    assert np.array_equal(ip.cpu().numpy(), w_ip) and np.array_equal(ix.cpu().numpy(), w_ix) and np.array_equal(inp.cpu().numpy(), w_in)
    assert np.array_equal(gix.cpu().numpy(), np.concatenate(rows) if nnz else np.zeros(0, np.int32))
    assert np.array_equal(w_in[ix.cpu().numpy()], gix.cpu().numpy())          # local ids name the same nodes


@pytest.mark.parametrize("known_ids", [False, True])
@pytest.mark.parametrize("ns,fanout", [(512, 15), (7000, 10), (3, 1), (20000, 5)])
def test_block_build_from_sampled_neighbours(ns, fanout, known_ids):
    from glnn_amd import ops
    n = 30000
    indptr, indices = random_graph(n, 12, seed=9, power=0.6, hub=9000, isolated=50)
    g_ip, g_ix = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
    seeds = torch.from_numpy(np.random.RandomState(ns).permutation(n)[:ns].astype(np.int64)).to(DEV)
    smp, cnt = ops.sample_neighbors(g_ip, g_ix, seeds, fanout, 1234)
    ip, ix, gix, inp, nnz, n_src = ops.block_build(seeds, smp_src=smp, smp_cnt=cnt, want_global=True, n_nodes=n if known_ids else 0)
    smp_h, cnt_h = smp.cpu().numpy(), cnt.cpu().numpy()
    rows = [smp_h[i, :cnt_h[i]] for i in range(ns)]
    w_ip, w_ix, w_in = _first_appearance_block(seeds.cpu().numpy(), rows)
    assert nnz == w_ip[-1] and n_src == len(w_in)
    assert np.array_equal(ip.cpu().numpy(), w_ip) and np.array_equal(ix.cpu().numpy(), w_ix) and np.array_equal(inp.cpu().numpy(), w_in)
    deg = np.diff(indptr)[seeds.cpu().numpy()]
    assert np.array_equal(cnt_h, np.minimum(deg, fanout))                      # dgl: all in-edges when in_deg <= fanout
    for i in (0, ns // 2, ns - 1):                                             # sampled edges are edges of the graph, without replacement
        v = int(seeds[i])
        pool = indices[indptr[v]:indptr[v + 1]]
        got = np.sort(rows[i])
        assert len(got) == min(len(pool), fanout)
        avail = np.sort(pool).tolist()
        for u in got:
            avail.remove(int(u))                                               # multi-edge multiplicity is respected


@pytest.mark.parametrize("ns,fanout", [(512, 15), (20000, 5), (3, 1), (700, None), (1, None)])
def test_global_id_block_equals_the_full_build(ns, fanout):
    """glnn_block_build_ids with indices = input_nodes = NULL (round 5; ops.block_build(global_only=True)): indptr, nnz and the edges' global
    source ids of the block -- no frontier table, no relabelling -- are exactly those of the full build, in the sampled and the
    full-neighbour mode; a loader asked for it (NodeDataLoader.global_first_block, what train_sage sets) yields the same inner blocks, an
    outermost block whose column ids are those global ids over the whole graph's nodes, and input_nodes = None."""
    from glnn_amd import ops
    n = 30000
    indptr, indices = random_graph(n, 12, seed=9, power=0.6, hub=9000, isolated=50)
    g_ip, g_ix = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
    seeds = torch.from_numpy(np.random.RandomState(ns).permutation(n)[:ns].astype(np.int64)).to(DEV)
    if fanout is None:
        cap = int((g_ip[seeds + 1] - g_ip[seeds]).sum()) + 5
        full = ops.block_build(seeds, g_ip, g_ix, nnz_cap=cap, want_global=True, n_nodes=n)
        glob = ops.block_build(seeds, g_ip, g_ix, nnz_cap=cap, n_nodes=n, global_only=True)
    else:
        smp, cnt = ops.sample_neighbors(g_ip, g_ix, seeds, fanout, 99)
        full = ops.block_build(seeds, smp_src=smp, smp_cnt=cnt, want_global=True, n_nodes=n)
        glob = ops.block_build(seeds, smp_src=smp, smp_cnt=cnt, n_nodes=n, global_only=True)
    assert glob[1] is None and glob[3] is None and glob[5] is None and glob[4] == full[4]
    assert torch.equal(glob[0], full[0]) and torch.equal(glob[2], full[2])


def test_loader_with_global_first_block_yields_the_same_batches():
    from glnn_amd.graph import CSRGraph, MultiLayerNeighborSampler, NodeDataLoader
    n = 30000
    indptr, indices = random_graph(n, 12, seed=9, power=0.6, hub=9000, isolated=50)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    nids = torch.from_numpy(np.random.RandomState(1).permutation(n)[:3000].astype(np.int64)).to(DEV)
    runs = []
    for glob in (False, True):
        ld = NodeDataLoader(g, nids, MultiLayerNeighborSampler([5, 10, 15]), batch_size=1024, shuffle=False, seed=7)
        ld.global_first_block = glob
        runs.append([(i, o, b) for i, o, b in ld])
    assert len(runs[0]) == len(runs[1]) == 3
    for (i0, o0, b0), (i1, o1, b1) in zip(*runs):
        assert i1 is None and i0 is not None and torch.equal(o0, o1)
        for x, y in zip(b0[1:], b1[1:]):
            assert torch.equal(x.indptr, y.indptr) and torch.equal(x.indices, y.indices) and x.n_src == y.n_src
            # engine mode: the inner blocks arrive with what the step's backward needs of them (built beside the previous step)
            from glnn_amd import ops
            assert x.t_indptr is None and y.t_indptr is not None
            ti, tx = ops.csr_transpose(x.indptr, x.indices, x.n_dst, x.n_src, x.num_edges(), add_self=True)
            assert torch.equal(y.t_indptr, ti) and torch.equal(y.t_indices, tx)
            deg = (x.indptr[1:] - x.indptr[:-1]).float()
            torch.testing.assert_close(y.inv_deg, 1.0 / (deg + 1.0), rtol=1e-6, atol=0)
        assert torch.equal(b0[0].indptr, b1[0].indptr) and torch.equal(b0[0].gindices, b1[0].gindices) and torch.equal(b0[0].dst_nodes, b1[0].dst_nodes)
        assert b1[0].indices is b1[0].gindices or torch.equal(b1[0].indices, b1[0].gindices)
        assert b1[0].n_src == n and b1[0].num_edges() == b0[0].num_edges()
        assert torch.equal(i0[b0[0].indices.long()], b0[0].gindices.long())          # the full build's local ids name those global ids


def test_loader_worker_thread_stops_on_early_exit_and_hands_over_exceptions():
    """The batches of a device-resident graph are built by a worker thread (NodeDataLoader.threaded): a consumer that leaves the loop early
    does not leave the thread behind, and an exception raised while a batch is built surfaces in the consumer's loop."""
    import gc, threading
    from glnn_amd.graph import CSRGraph, MultiLayerNeighborSampler, NodeDataLoader
    n = 20000
    indptr, indices = random_graph(n, 10, seed=3, power=0.5)
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    ld = NodeDataLoader(g, torch.arange(8192, device=DEV), MultiLayerNeighborSampler([5, 10]), batch_size=512, seed=1)
    assert ld.threaded
    alive = lambda: [t for t in threading.enumerate() if t.name == "glnn-node-loader"]
    for k, batch in enumerate(ld):
        if k == 1:
            break
    gc.collect()
    assert not alive()
    it = iter(ld)
    next(it)
    it.close()
    assert not alive()
    real, calls = ld._batch, []

    def failing(b, idx, fanouts, *rest):
        calls.append(b)
        if b == 2:
            raise RuntimeError("boom in batch 2")
        return real(b, idx, fanouts, *rest)

    ld._batch = failing
    seen = 0
    with pytest.raises(RuntimeError, match="boom in batch 2"):
        for batch in ld:
            seen += 1
    assert seen == 2 and not alive()


def test_products_scale_sampled_blocks_properties():
    """The block builder at the products training configuration (B = 4096, fan-out 15 / 10 / 5 from the output layer inwards, 2.45 M nodes:
    the two wide blocks index their position table by the node id, the first one probes a hash table), checked through properties that
    need no CPU restatement: local ids name the sampled global ids, destinations come first and in order, sources are unique, new
    sources appear in order of first occurrence, and the builder that is NOT told the id universe (hash tables throughout) returns
    the same block bit for bit."""
    from glnn_amd import data, ops
    g = data.make_graph("ogbn-products", seed=0, device=DEV)
    n = g.n_dst
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(7))[:4096].to(DEV)
    for level, fanout in enumerate((15, 10, 5)):
        smp, cnt = ops.sample_neighbors(g.indptr, g.indices, seeds, fanout, 99 + level)
        blocks = [ops.block_build(seeds, smp_src=smp, smp_cnt=cnt, want_global=True, n_nodes=nn) for nn in (n, 0)]
        ip, ix, gix, inp, nnz, n_src = blocks[0]
        for a, b in zip(blocks[0][:4], blocks[1][:4]):
            assert torch.equal(a, b)
        assert blocks[0][4:] == blocks[1][4:]
        ns = seeds.numel()
        assert nnz == int(cnt.sum()) and torch.equal(ip[1:] - ip[:-1], cnt.long())
        assert torch.equal(inp[:ns], seeds)                                        # destinations first, in order
        assert torch.equal(inp[ix.long()], gix.long())                             # local ids name the sampled nodes
        assert torch.unique(inp).numel() == n_src                                  # every source once
        first = torch.full((n,), nnz, dtype=torch.int64, device=DEV)
        first.scatter_reduce_(0, gix.long(), torch.arange(nnz, device=DEV), "amin")
        new = inp[ns:]
        assert bool((first[new][1:] > first[new][:-1]).all())                      # new sources in order of first appearance
        is_dst = torch.zeros(n, dtype=torch.bool, device=DEV)
        is_dst[seeds] = True
        assert not bool(is_dst[new].any())
        seeds = inp                                                                # the next (wider) block's destinations


def test_full_neighbour_loader_chunks_map_back_to_the_graph():
    """The reference's dataloader_eval (train_and_eval.py:193-202) as glnn_amd.graph.FullNeighborLoader: chunks in node-id
    order, the chunk's destinations first among the block's sources, block edges = the graph's edges."""
    from glnn_amd.graph import FullNeighborLoader
    n = 777
    indptr, indices = random_graph(n, 6, seed=2, power=0.5, hub=300, isolated=7)
    g = _graph(indptr, indices)
    seen = 0
    loader = FullNeighborLoader(g, 100)
    assert len(loader) == 8
    for input_nodes, output_nodes, blocks in loader:
        b = blocks[0]
        inp = input_nodes.cpu().numpy()
        assert output_nodes.tolist() == list(range(seen, min(n, seen + 100)))
        assert np.array_equal(inp[: len(output_nodes)], output_nodes.cpu().numpy()) and b.num_dst_nodes() == len(output_nodes)
        assert len(np.unique(inp)) == len(inp) == b.num_src_nodes()
        bi, bx = b.indptr.cpu().numpy(), b.indices.cpu().numpy()
        for i, v in enumerate(output_nodes.tolist()):
            assert np.array_equal(inp[bx[bi[i]:bi[i + 1]]], indices[indptr[v]:indptr[v + 1]])
        seen += len(output_nodes)
    assert seen == n


def test_neighbour_sampler_is_uniform_without_replacement():
    """chi-square on the selection frequency of each in-edge of a degree-40 row over 10^4 independent draws, fan-out 10
    (expected 2500 per edge): glnn_sample_neighbors draws uniformly WITHOUT replacement; a row with duplicate (multi-)edges
    returns each copy independently."""
    from glnn_amd import ops
    deg, fanout, draws = 40, 10, 10000
    nbr = np.arange(100, 100 + deg).astype(np.int32)
    nbr[7] = nbr[3]                                           # a multi-edge: node 103 appears twice in the row
    indptr = torch.tensor([0, deg], dtype=torch.int64, device=DEV)
    indices = torch.from_numpy(nbr).to(DEV)
    seeds = torch.zeros(draws, dtype=torch.int64, device=DEV)                  # the same row `draws` times: the RNG is keyed by the seed index
    smp, cnt = ops.sample_neighbors(indptr, indices, seeds, fanout, 20240917)
    assert int(cnt.min()) == fanout and int(cnt.max()) == fanout
    s = smp.cpu().numpy()
    counts = np.bincount(s.ravel() - 100, minlength=deg).astype(np.float64)
    counts[3] += counts[7]                                    # node 103 is drawn through either copy
    per_slot = draws * fanout / deg
    expected = np.full(deg, per_slot)
    expected[3], expected[7] = 2 * per_slot, 0
    keep = expected > 0
    chi2 = (((counts - expected) ** 2)[keep] / expected[keep]).sum()
    assert chi2 < 80.0, chi2                                  # 38 degrees of freedom: P(chi2 > 80) < 1e-4
    # without replacement: per draw, node 103 at most twice (its two copies), every other node at most once
    for row in s[:200]:
        vals, c = np.unique(row, return_counts=True)
        assert all(cc <= (2 if v == 103 else 1) for v, cc in zip(vals, c))
    # a second RNG seed gives different draws; the same seed the same draws
    smp2, _ = ops.sample_neighbors(indptr, indices, seeds, fanout, 20240917)
    smp3, _ = ops.sample_neighbors(indptr, indices, seeds, fanout, 99)
    assert torch.equal(smp, smp2) and not torch.equal(smp, smp3)


# ------------------------------------------------------------------------------------------------ TeacherEngine
def _teacher(kind, dims, norm, sd, wd):
    from glnn_amd.models import Model
    conf = dict(model_name=kind, num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                norm_type=norm, device=DEV)
    model = Model(conf)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=wd)       # train_teacher.py:234-236
    return model, opt


def _is_gauge(norm, k):
    return norm == "batch" and ((k.endswith("fc_neigh.bias") and not k.startswith("encoder.layers.2")) or k.endswith("running_mean"))


@pytest.mark.parametrize("norm", ["batch", "none"])
def test_train_sage_vs_reference_golden(norm):
    """glnn_amd.train_and_eval.train_sage (TeacherEngine) over the fixture's blocks vs the reference's own train_sage:
    first-step gradients, per-epoch mean losses, final parameters / BatchNorm buffers / Adam step count."""
    from glnn_amd import ops
    from glnn_amd import train_and_eval as te
    from glnn_amd.teacher import TeacherEngine
    z, batches = teacher_training()
    tag = f"sage.{norm}"
    dims = [int(d) for d in z["sage.dims"]]
    feats, labels = torch.from_numpy(z["sage.feats"]).to(DEV), torch.from_numpy(z["sage.labels"]).to(DEV)
    dev_batches = [(torch.from_numpy(i).to(DEV), torch.from_numpy(o).to(DEV), [_graph(ip, ix, ns) for ip, ix, ns in blks])
                   for i, o, blks in batches]
    model, opt = _teacher("SAGE", dims, norm, sub_dict(z, f"{tag}.init."), float(z[f"{tag}.wd"]))
    model.train()
    eng = TeacherEngine(model, opt)
    inp, outn, blks = dev_batches[0]
    snap = {k: v.clone() for k, v in model.state_dict().items()}
    eng.step_sage(blks, ops.as_feat(feats), labels, outn, 1.0, input_nodes=inp)
    for pname, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), z[f"{tag}.grad0.{pname}"], atol=1e-5, rtol=1e-4, err_msg=pname)
    assert abs(eng.loss_out.item() - float(z[f"{tag}.step_losses"][0])) < TOL
    # the full two epochs through the preserved train_sage surface, from a fresh model / optimizer
    model, opt = _teacher("SAGE", dims, norm, {k: v.cpu().numpy() for k, v in snap.items()}, float(z[f"{tag}.wd"]))
    crit = torch.nn.NLLLoss()
    means = [te.train_sage(model, dev_batches, feats, labels, crit, opt) for _ in range(2)]
    np.testing.assert_allclose(means, z[f"{tag}.epoch_losses"], atol=TOL, rtol=0)
    assert int(opt.state[next(model.parameters())]["step"]) == 6
    for k, v in model.state_dict().items():
        want = z[f"{tag}.final.{k}"]
        if v.ndim == 0:
            assert int(v) == int(want), k
        elif not _is_gauge(norm, k):
            np.testing.assert_allclose(v.cpu().numpy(), want, atol=2e-4 if norm == "batch" else TOL, rtol=0, err_msg=k)


def test_train_gcn_vs_reference_golden():
    from glnn_amd import train_and_eval as te
    z, _ = teacher_training()
    dims = [int(d) for d in z["gcn.dims"]]
    model, opt = _teacher("GCN", dims, "none", sub_dict(z, "gcn.init."), 1e-3)
    g = _graph(z["gcn.indptr"], z["gcn.indices"])
    feats, labels = torch.from_numpy(z["gcn.feats"]).to(DEV), torch.from_numpy(z["gcn.labels"]).to(DEV)
    idx_train = torch.from_numpy(z["gcn.idx_train"]).to(DEV)
    losses = [te.train(model, g, feats, labels, torch.nn.NLLLoss(), opt, idx_train) for _ in range(5)]
    np.testing.assert_allclose(losses, z["gcn.losses"], atol=TOL, rtol=0)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), z[f"gcn.final.{k}"], atol=TOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("tag,norm", [("gcnbn", "batch"), ("gcnln", "layer")])
def test_train_gcn_with_norm_layer_vs_reference_golden(tag, norm):
    """GCN with a norm layer (round 3): five steps of the reference's own `train` (train_and_eval.py:12-29) over `GCN.forward`
    (models.py:189-199: GraphConv(relu) -> norms[l] -> dropout, no ReLU behind the norm) with nn.BatchNorm1d / nn.LayerNorm, Adam
    with weight_decay 0.001 (the pokec / penn94 GCN sections of train.conf.yaml) -- per-step losses, final parameters and
    BatchNorm buffers, and the eval-mode logits of the trained model, against TeacherEngine.step_gcn / GCN.forward."""
    from glnn_amd import train_and_eval as te
    z, _ = teacher_training()
    dims = [int(d) for d in z[f"{tag}.dims"]]
    model, opt = _teacher("GCN", dims, norm, sub_dict(z, f"{tag}.init."), 1e-3)
    g = _graph(z["gcn.indptr"], z["gcn.indices"])
    feats, labels = torch.from_numpy(z["gcn.feats"]).to(DEV), torch.from_numpy(z["gcn.labels"]).to(DEV)
    idx_train = torch.from_numpy(z["gcn.idx_train"]).to(DEV)
    losses = [te.train(model, g, feats, labels, torch.nn.NLLLoss(), opt, idx_train) for _ in range(5)]
    np.testing.assert_allclose(losses, z[f"{tag}.losses"], atol=TOL, rtol=0)
    for k, v in model.state_dict().items():
        want = z[f"{tag}.final.{k}"]
        if v.dim() == 0:
            assert int(v) == int(want), k
        else:
            np.testing.assert_allclose(v.cpu().numpy(), want, atol=2e-4, rtol=0, err_msg=k)
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(g, feats).cpu().numpy(), z[f"{tag}.eval_logits"], atol=5e-4, rtol=0)


def test_gcn_step_both_weight_orders_and_dropout_vs_oracle():
    """GraphConv aggregates first when in <= out and multiplies first when in > out (dgl): a 12->20->20->5 GCN takes both
    branches in one step; then the same with dropout 0.5, the oracle fed with the kernels' own keep-masks."""
    from glnn_amd import ops
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    from oracle.dropout_mask import keep_mask
    n, dims = 500, [12, 20, 20, 5]
    indptr, indices = random_graph(n, 4, seed=3, symmetric=True, self_loops=True)
    rs = np.random.RandomState(3)
    feats = rs.standard_normal((n, dims[0])).astype(np.float32)
    labels = rs.randint(0, dims[-1], n).astype(np.int64)
    idx_train = np.sort(rs.permutation(n)[:120]).astype(np.int64)
    g = _graph(indptr, indices)
    for p in (0.0, 0.5):
        torch.manual_seed(11)
        model = Model(dict(model_name="GCN", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type="none", device=DEV))
        with torch.no_grad():
            for lay in model.encoder.layers:
                lay.bias.copy_(torch.randn_like(lay.bias) * 0.1)
        sd0 = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3)
        model.train()
        eng = TeacherEngine(model, opt)
        eng.step_gcn(g, torch.from_numpy(feats).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(idx_train).to(DEV), 1.0)
        st = tt.TeacherState(sd0, "gcn", 3, "none")
        masks = None
        if p > 0:
            eng.step_count = 1
            masks = [keep_mask(n, dims[l + 1], p, eng._seed(l)).astype(np.float32) for l in range(2)]
        logits, cache = tt.gcn_forward(st, indptr, indices, feats, masks=masks, p=p)
        from oracle import student_oracle as so
        loss, dl = so.loss_and_dlogits(logits[idx_train], labels[idx_train], "nll", 1.0)
        dlogits = np.zeros_like(logits)
        dlogits[idx_train] = dl
        grads = tt.gcn_backward(st, cache, dlogits, p=p)
        assert abs(eng.loss_out.item() - float(loss)) < TOL
        for (pname, prm), gr in zip(model.named_parameters(), grads):
            np.testing.assert_allclose(prm.grad.cpu().numpy(), gr, atol=1e-5, rtol=1e-4, err_msg=f"p={p} {pname}")


def test_train_sage_on_device_built_blocks_vs_oracle_and_global_gather():
    """Blocks sampled and relabelled on the device (NodeDataLoader): (i) the engine's step equals the numpy oracle on the
    same blocks; (ii) aggregating the outermost block straight out of `feats` through its global ids (no feats[input_nodes])
    gives bit-identical gradients to the gathered form."""
    from glnn_amd import ops
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    from oracle import student_oracle as so
    n, dims = 20000, [40, 64, 64, 9]
    indptr, indices = random_graph(n, 10, seed=8, power=0.6, hub=5000, isolated=30)
    rs = np.random.RandomState(8)
    feats = rs.standard_normal((n, dims[0])).astype(np.float32)
    labels = rs.randint(0, dims[-1], n).astype(np.int64)
    g = _graph(indptr, indices)
    loader = NodeDataLoader(g, torch.arange(700), MultiLayerNeighborSampler([5, 10, 15]), batch_size=512, shuffle=False, seed=5)
    assert len(loader) == 2
    batch = list(loader)[1]                                   # the short last batch (drop_last=False): 188 seeds
    input_nodes, output_nodes, blocks = batch
    assert output_nodes.tolist() == list(range(512, 700)) and blocks[0].gindices is not None
    assert torch.equal(input_nodes[blocks[0].indices.long()], blocks[0].gindices.long())
    for b_outer, b_inner in zip(blocks[:-1], blocks[1:]):
        assert b_outer.num_dst_nodes() == b_inner.num_src_nodes()              # the inner block's sources are the outer's destinations
    fd, ld = ops.as_feat(torch.from_numpy(feats).to(DEV)), torch.from_numpy(labels).to(DEV)
    grads = {}
    for mode in ("global", "gathered"):
        torch.manual_seed(2)
        model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                           norm_type="batch", device=DEV))
        sd0 = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
        opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=0.0)
        model.train()
        eng = TeacherEngine(model, opt)
        saved = blocks[0].gindices
        if mode == "gathered":
            blocks[0].gindices = None
        eng.step_sage(blocks, fd, ld, output_nodes, 1.0, input_nodes=input_nodes)
        blocks[0].gindices = saved
        grads[mode] = [p.grad.clone() for p in model.parameters()]
        loss_gpu = eng.loss_out.item()
    for a, b in zip(grads["global"], grads["gathered"]):
        assert torch.equal(a, b)
    st = tt.TeacherState(sd0, "sage", 3, "batch")
    nb = [(b.indptr.cpu().numpy(), b.indices.cpu().numpy(), b.num_src_nodes()) for b in blocks]
    inp = input_nodes.cpu().numpy()
    logits, cache = tt.sage_forward(st, nb, feats[inp])
    loss, dl = so.loss_and_dlogits(logits, labels[output_nodes.cpu().numpy()], "nll", 1.0)
    assert abs(loss_gpu - float(loss)) < TOL
    for g_gpu, g_ref, (pname, _) in zip(grads["global"], tt.sage_backward(st, cache, dl), model.named_parameters()):
        np.testing.assert_allclose(g_gpu.cpu().numpy(), g_ref, atol=1e-5, rtol=1e-4, err_msg=pname)


def test_autograd_surface_matches_the_engine():
    """Callers that differentiate Model.forward themselves (loss.backward() as in the reference's loops) get the same
    gradients as the engine: both run the same HIP kernels."""
    from glnn_amd import ops
    from glnn_amd.teacher import TeacherEngine
    z, batches = teacher_training()
    dims = [int(d) for d in z["sage.dims"]]
    feats, labels = torch.from_numpy(z["sage.feats"]).to(DEV), torch.from_numpy(z["sage.labels"]).to(DEV)
    inp, outn, blks = batches[0]
    inp, outn = torch.from_numpy(inp).to(DEV), torch.from_numpy(outn).to(DEV)
    blocks = [_graph(ip, ix, ns) for ip, ix, ns in blks]
    model, opt = _teacher("SAGE", dims, "batch", sub_dict(z, "sage.batch.init."), 0.0)
    model.train()
    logits = model(blocks, ops.gather_rows(ops.as_feat(feats), inp))
    loss = torch.nn.NLLLoss()(logits.log_softmax(dim=1), labels[outn])       # the caller's own torch code
    loss.backward()
    auto = [p.grad.clone() for p in model.parameters()]
    model2, opt2 = _teacher("SAGE", dims, "batch", sub_dict(z, "sage.batch.init."), 0.0)
    model2.train()
    eng = TeacherEngine(model2, opt2)
    eng.step_sage(blocks, ops.as_feat(feats), labels, outn, 1.0, input_nodes=inp)
    for a, p, (pname, _) in zip(auto, model2.parameters(), model2.named_parameters()):
        np.testing.assert_allclose(a.cpu().numpy(), p.grad.cpu().numpy(), atol=1e-6, rtol=1e-5, err_msg=pname)
    # GCN through autograd as well
    g = _graph(z["gcn.indptr"], z["gcn.indices"])
    gdims = [int(d) for d in z["gcn.dims"]]
    gcn, _ = _teacher("GCN", gdims, "none", sub_dict(z, "gcn.init."), 1e-3)
    gcn.train()
    gf, gl = torch.from_numpy(z["gcn.feats"]).to(DEV), torch.from_numpy(z["gcn.labels"]).to(DEV)
    it = torch.from_numpy(z["gcn.idx_train"]).to(DEV)
    out = gcn(g, gf).log_softmax(dim=1)
    torch.nn.NLLLoss()(out[it], gl[it]).backward()
    gcn2, gopt2 = _teacher("GCN", gdims, "none", sub_dict(z, "gcn.init."), 1e-3)
    gcn2.train()
    TeacherEngine(gcn2, gopt2).step_gcn(g, gf, gl, it, 1.0)
    for p, q, (pname, _) in zip(gcn.parameters(), gcn2.parameters(), gcn2.named_parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), atol=1e-6, rtol=1e-5, err_msg=pname)


def test_unsupported_teacher_configurations_raise():
    from glnn_amd import train_and_eval as te
    from glnn_amd.models import Model
    z, _ = teacher_training()
    dims = [int(d) for d in z["gcn.dims"]]
    model, opt = _teacher("GCN", dims, "none", sub_dict(z, "gcn.init."), 1e-3)
    g = _graph(z["gcn.indptr"], z["gcn.indices"])
    feats, labels = torch.from_numpy(z["gcn.feats"]).to(DEV), torch.from_numpy(z["gcn.labels"]).to(DEV)
    it = torch.from_numpy(z["gcn.idx_train"]).to(DEV)
    with pytest.raises(NotImplementedError):
        te.train(model, g, feats, labels, torch.nn.CrossEntropyLoss(), opt, it)
    with pytest.raises(NotImplementedError):
        te.train(model, g, feats, labels, torch.nn.NLLLoss(), torch.optim.SGD(model.parameters(), lr=0.1), it)
    cpu_model = Model(dict(model_name="MLP", num_layers=2, feat_dim=8, hidden_dim=8, label_dim=3, dropout_ratio=0.0, norm_type="none",
                           device="cpu"))
    with pytest.raises(RuntimeError):
        te.train_mini_batch(cpu_model, torch.randn(16, 8), torch.zeros(16, dtype=torch.int64), 8, torch.nn.NLLLoss(),
                            torch.optim.Adam(cpu_model.parameters()))
    with pytest.raises(Exception):
        cpu_model(None, torch.randn(4, 8))                    # no CPU forward either


def test_train_sage_step_with_dropout_matches_oracle_given_the_masks():
    """Dropout 0.5 in the sampled-block step: the engine's counter-based masks (per hidden layer, keyed by the activation's
    row / column) restated in numpy and fed to the oracle -> same loss and gradients."""
    from glnn_amd import ops
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    from oracle import student_oracle as so
    from oracle.dropout_mask import keep_mask
    z, batches = teacher_training()
    dims = [int(d) for d in z["sage.dims"]]
    feats, labels = z["sage.feats"], z["sage.labels"]
    inp, outn, blks = batches[1]
    p = 0.5
    for norm in ("batch", "none"):
        torch.manual_seed(4)
        model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type=norm, device=DEV))
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sub_dict(z, f"sage.{norm}.init.").items()})
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)
        model.train()
        eng = TeacherEngine(model, opt)
        blocks = [_graph(ip, ix, ns) for ip, ix, ns in blks]
        eng.step_sage(blocks, ops.as_feat(torch.from_numpy(feats).to(DEV)), torch.from_numpy(labels).to(DEV), torch.from_numpy(outn).to(DEV), 1.0,
                      input_nodes=torch.from_numpy(inp).to(DEV))
        masks = [keep_mask(len(blks[l][0]) - 1, dims[l + 1], p, eng._seed(l)).astype(np.float32) for l in range(2)]   # step_count == 1
        assert all(abs(m.mean() - (1 - p)) < 0.1 for m in masks)
        st = tt.TeacherState(sub_dict(z, f"sage.{norm}.init."), "sage", 3, norm)
        logits, cache = tt.sage_forward(st, blks, feats[inp], masks=masks, p=p)
        loss, dl = so.loss_and_dlogits(logits, labels[outn], "nll", 1.0)
        assert abs(eng.loss_out.item() - float(loss)) < TOL
        for (pname, prm), gr in zip(model.named_parameters(), tt.sage_backward(st, cache, dl, p=p)):
            if norm == "batch" and pname.endswith("fc_neigh.bias") and not pname.startswith("encoder.layers.2"):
                continue            # zero true gradient in front of a BatchNorm: both sides hold rounding noise
            np.testing.assert_allclose(prm.grad.cpu().numpy(), gr, atol=2e-5, rtol=1e-4, err_msg=f"{norm} {pname}")


def test_full_size_products_training_epoch_properties():
    """The reference's products teacher config at FULL size (reference train.conf.yaml:196-204: fan-out 5,10,15, B=4096,
    dropout 0.5, BatchNorm, lr 0.003) on the 2.45 M-node / 124 M-edge products-shaped graph: block invariants on the biggest
    blocks the path sees (hundreds of thousands of destinations), finite losses, BatchNorm / Adam counters in step."""
    from glnn_amd import data
    from glnn_amd import train_and_eval as te
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    torch.manual_seed(0)
    g = data.make_graph("ogbn-products", seed=0, device=DEV)
    n = g.n_dst
    feats, labels, _, _ = data.make_node_data("ogbn-products", seed=0, device=DEV, n=n)
    idx_train = torch.randperm(n)[:40960].to(DEV)
    loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=4096, shuffle=True, drop_last=False)
    input_nodes, output_nodes, blocks = next(iter(loader))
    assert len(output_nodes) == 4096 and blocks[2].num_dst_nodes() == 4096 and int(blocks[2].in_degrees().max()) <= 15
    assert int(blocks[0].in_degrees().max()) <= 5 and blocks[0].num_src_nodes() == len(input_nodes) > 500_000
    assert torch.equal(input_nodes[:4096], output_nodes)
    assert int(torch.unique(input_nodes).numel()) == len(input_nodes)                       # a node appears once among the sources
    assert torch.equal(input_nodes[blocks[0].indices.long()], blocks[0].gindices.long())
    e = torch.randint(0, blocks[0].num_edges(), (2000,), device=DEV)                          # sampled edges are edges of the graph
    dst_of = torch.searchsorted(blocks[0].indptr, e, right=True) - 1
    v, u = input_nodes[dst_of], blocks[0].gindices[e].long()
    lo, hi = g.indptr[v], g.indptr[v + 1]
    hit = torch.zeros(2000, dtype=torch.bool, device=DEV)
    for k in range(int((hi - lo).max())):
        pos = (lo + k).clamp(max=g.num_edges() - 1)
        hit |= (k < (hi - lo)) & (g.indices[pos].long() == u)
    assert bool(hit.all())
    model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=100, hidden_dim=256, label_dim=47, dropout_ratio=0.5, norm_type="batch",
                       device=DEV))
    opt = torch.optim.Adam(model.parameters(), lr=0.003)
    l1 = te.train_sage(model, loader, feats, labels, torch.nn.NLLLoss(), opt)
    l2 = te.train_sage(model, loader, feats, labels, torch.nn.NLLLoss(), opt)
    assert np.isfinite(l1) and np.isfinite(l2) and 3.0 < l2 < 5.0            # ln(47) = 3.85: random labels cannot be learnt
    assert int(model.encoder.norms[0].num_batches_tracked) == 20 and int(opt.state[next(model.parameters())]["step"]) == 20
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_block_builder_and_loader_edge_cases():
    """Empty and ragged inputs: no seeds, seeds without in-edges, a single seed, duplicate-free relabelling when every sampled
    source is itself a seed, drop_last / short last batch, prefetch on and off giving the same batches."""
    from glnn_amd import ops
    from glnn_amd.graph import MultiLayerFullNeighborSampler, MultiLayerNeighborSampler, NodeDataLoader
    n = 500
    indptr, indices = random_graph(n, 4, seed=1, isolated=40)
    g = _graph(indptr, indices)
    iso = torch.from_numpy(np.flatnonzero(np.diff(indptr) == 0)[:7].astype(np.int64)).to(DEV)
    # seeds without any in-edge: an empty block whose sources are the seeds themselves
    ip, ix, _, inp, nnz, n_src = ops.block_build(iso, g.indptr, g.indices, nnz_cap=0)
    assert nnz == 0 and n_src == 7 and ip.tolist() == [0] * 8 and ix.numel() == 0 and torch.equal(inp, iso)
    smp, cnt = ops.sample_neighbors(g.indptr, g.indices, iso, 5, 3)
    ip, ix, _, inp, nnz, n_src = ops.block_build(iso, smp_src=smp, smp_cnt=cnt)
    assert nnz == 0 and n_src == 7 and int(cnt.max()) == 0
    # no seeds at all
    none = torch.zeros(0, dtype=torch.int64, device=DEV)
    ip, ix, _, inp, nnz, n_src = ops.block_build(none, g.indptr, g.indices, nnz_cap=0)
    assert nnz == 0 and n_src == 0 and ip.tolist() == [0]
    # a complete graph on 6 nodes as seeds: every source is a seed -> no extra input nodes
    src = np.repeat(np.arange(6), 6); dst = np.tile(np.arange(6), 6)
    from graphgen import csr_from_edges
    kip, kix = csr_from_edges(src, dst, 6)
    kg = _graph(kip, kix)
    seeds = torch.tensor([3, 1, 5, 0, 2, 4], device=DEV)
    ip, ix, _, inp, nnz, n_src = ops.block_build(seeds, kg.indptr, kg.indices, nnz_cap=36)
    assert n_src == 6 and nnz == 36 and torch.equal(inp, seeds) and torch.equal(seeds[ix.long()], kg.indices.long()[
        torch.cat([torch.arange(int(kg.indptr[v]), int(kg.indptr[v + 1])) for v in seeds.tolist()]).to(DEV)])
    # an id outside the universe the caller states (id-indexed tables): refused through counts, never used as an index
    from glnn_amd import GlnnError
    with pytest.raises(GlnnError):
        ops.block_build(seeds, kg.indptr, kg.indices, nnz_cap=36, n_nodes=4)
    ip2, ix2, _, inp2, nnz2, n_src2 = ops.block_build(seeds, kg.indptr, kg.indices, nnz_cap=36, n_nodes=6)
    assert nnz2 == 36 and n_src2 == 6 and torch.equal(ix2, ix)
    # loaders: short last batch / drop_last; the side-stream prefetch does not change what is produced
    nids = torch.arange(10, 110)
    for sampler in (MultiLayerNeighborSampler([3, 4]), MultiLayerFullNeighborSampler(2)):
        batches = {}
        for prefetch in (True, False):
            loader = NodeDataLoader(g, nids, sampler, batch_size=32, shuffle=False, drop_last=False, seed=9)
            loader.prefetch = prefetch
            assert len(loader) == 4
            batches[prefetch] = [(i.clone(), o.clone(), [b.indices.clone() for b in bl]) for i, o, bl in loader]
            assert [len(o) for _, o, _ in batches[prefetch]] == [32, 32, 32, 4]
        for (ia, oa, ba), (ib, ob, bb) in zip(batches[True], batches[False]):
            assert torch.equal(ia, ib) and torch.equal(oa, ob) and all(torch.equal(x, y) for x, y in zip(ba, bb))
        assert len(list(NodeDataLoader(g, nids, sampler, batch_size=32, drop_last=True))) == 3
    # transposing an empty block
    t = _graph(np.zeros(4, np.int64), np.zeros(0, np.int32), 9).transposed(add_self=True)
    assert t.indptr.tolist() == [0, 1, 2, 3, 3, 3, 3, 3, 3, 3] and t.indices.tolist() == [0, 1, 2]


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("GLNN_FUZZ_CASES", "12")))) + ["wide"])
def test_sage_and_gcn_teachers_on_random_shapes_vs_oracle(seed):
    """Randomised shapes (feature / hidden / class widths that are not multiples of 4, 1-3 layers, short last batches, hubs and
    isolated rows): SAGE inference through both sweeps, one sampled-block training step, and one full-graph GCN step, each
    against the numpy oracle.  The fixed cases above pin the reference's configurations; this one walks the dispatch
    (unaligned operands -> latency GEMM, narrow tails, one-layer models) that no configuration names."""
    from glnn_amd import ops
    from glnn_amd.graph import FullNeighborLoader, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    from oracle import student_oracle as so
    from oracle import teacher_oracle as to
    wide = seed == "wide"      # hidden_dim 512 (> 256: the tail-in-gather kernel does not take it -- h is materialised)
    seed = 77 if wide else seed
    rs = np.random.RandomState(1000 + seed)
    pick = lambda xs: xs[rs.randint(len(xs))]
    L = pick([1, 2, 2, 3])
    f, h, c = pick([5, 7, 33, 50, 100, 130, 257]), pick([8, 17, 33, 64, 100, 256]), pick([2, 3, 7, 40, 47, 70])
    norm = pick(["batch", "none"])
    if wide:
        L, h, norm = 3, 512, "batch"
    n = int(pick([300, 1111, 4000, 9000]))
    dims = [f] + [h] * (L - 1) + [c]
    indptr, indices = random_graph(n, pick([2, 6, 14]), seed=seed, power=pick([0.0, 0.6]), isolated=pick([0, 7]), hub=pick([0, n // 3]))
    feats = rs.standard_normal((n, f)).astype(np.float32)
    labels = rs.randint(0, c, n).astype(np.int64)
    g = _graph(indptr, indices)
    fd, ld = ops.as_feat(torch.from_numpy(feats).to(DEV)), torch.from_numpy(labels).to(DEV)
    tag = f"seed={seed} dims={dims} norm={norm} n={n}"

    # ---- SAGE: inference (eval) through the whole-graph path and the chunked sweep
    torch.manual_seed(seed)
    model = Model(dict(model_name="SAGE", num_layers=L, feat_dim=f, hidden_dim=h, label_dim=c, dropout_ratio=0.0, norm_type=norm, device=DEV))
    with torch.no_grad():
        for bn in model.encoder.norms:
            bn.weight.copy_(torch.from_numpy(rs.uniform(.5, 1.5, bn.weight.shape[0]).astype(np.float32)))
            bn.bias.copy_(torch.from_numpy(rs.uniform(-.2, .2, bn.weight.shape[0]).astype(np.float32)))
            bn.running_mean.copy_(torch.from_numpy(rs.uniform(-.3, .3, bn.weight.shape[0]).astype(np.float32)))
            bn.running_var.copy_(torch.from_numpy(rs.uniform(.5, 1.5, bn.weight.shape[0]).astype(np.float32)))
    sd0 = {k: v.cpu().numpy().copy() for k, v in model.state_dict().items()}
    layers = [dict(weight=sd0[f"encoder.layers.{i}.fc_neigh.weight"], bias=sd0[f"encoder.layers.{i}.fc_neigh.bias"]) for i in range(L)]
    norms = [dict(weight=sd0[f"encoder.norms.{i}.weight"], bias=sd0[f"encoder.norms.{i}.bias"], running_mean=sd0[f"encoder.norms.{i}.running_mean"],
                  running_var=sd0[f"encoder.norms.{i}.running_var"]) for i in range(L - 1)] if norm == "batch" else None
    want = to.sage_inference(indptr, indices, feats, layers, norms)
    model.eval()
    loader = FullNeighborLoader(g, int(pick([100, 512, 3000])))
    with torch.no_grad():
        got = model.inference(loader, fd)
        got_chunked = model.encoder.inference(loader, fd, whole_graph=False)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL * scale, rtol=0, err_msg=tag)
    np.testing.assert_allclose(got_chunked.cpu().numpy(), want, atol=TOL * scale, rtol=0, err_msg=tag)

    # ---- SAGE: one sampled-block training step on the short last batch
    bs = int(pick([64, 200, 512]))
    n_seed = min(n, bs + int(pick([1, 37, bs - 1])))
    nl = NodeDataLoader(g, torch.arange(n_seed), MultiLayerNeighborSampler([int(pick([3, 5, 10]))] * L), batch_size=bs, shuffle=False, seed=seed)
    input_nodes, output_nodes, blocks = list(nl)[-1]
    opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=0.0)
    model.train()
    eng = TeacherEngine(model, opt)
    eng.step_sage(blocks, fd, ld, output_nodes, 1.0, input_nodes=input_nodes)
    st = tt.TeacherState(sd0, "sage", L, norm)
    nb = [(b.indptr.cpu().numpy(), b.indices.cpu().numpy(), b.num_src_nodes()) for b in blocks]
    logits, cache = tt.sage_forward(st, nb, feats[input_nodes.cpu().numpy()])
    loss, dl = so.loss_and_dlogits(logits, labels[output_nodes.cpu().numpy()], "nll", 1.0)
    assert abs(eng.loss_out.item() - float(loss)) < TOL * max(1.0, abs(float(loss))), tag
    grads = tt.sage_backward(st, cache, dl)
    gmax = max(float(np.abs(gr).max()) for gr in grads)
    edge = min([float(np.abs(t["y"]).min() / np.abs(t["y"]).max()) for t in cache["tails"]] + [1.0])
    for (pname, prm), gr in zip(model.named_parameters(), grads):
        if edge < 3e-7:
            break          # a pre-activation within two fp32 ulps of 0: which side of the ReLU it falls on is rounding, not arithmetic
        np.testing.assert_allclose(prm.grad.cpu().numpy(), gr, atol=2e-5 * max(1.0, gmax), rtol=1e-4, err_msg=f"{tag} {pname} edge={edge:.2e}")

    # ---- GCN: one full-graph step (symmetric graph with self loops, both weight orders decided by the widths)
    ip2, ix2 = random_graph(n, 4, seed=seed + 50, symmetric=True, self_loops=True, isolated=0)
    g2 = _graph(ip2, ix2)
    torch.manual_seed(seed)
    gcn = Model(dict(model_name="GCN", num_layers=max(L, 2), feat_dim=f, hidden_dim=h, label_dim=c, dropout_ratio=0.0, norm_type="none", device=DEV))
    sd1 = {k: v.cpu().numpy().copy() for k, v in gcn.state_dict().items()}
    opt2 = torch.optim.Adam(gcn.parameters(), lr=0.01, weight_decay=0.0)
    gcn.train()
    eng2 = TeacherEngine(gcn, opt2)
    idx = np.sort(rs.choice(n, size=max(1, n // 3), replace=False))
    eng2.step_gcn(g2, fd, ld, torch.from_numpy(idx).to(DEV), 1.0)
    st2 = tt.TeacherState(sd1, "gcn", max(L, 2), "none")
    logits2, cache2 = tt.gcn_forward(st2, ip2, ix2, feats)
    loss2, dl2 = so.loss_and_dlogits(logits2[idx], labels[idx], "nll", 1.0)
    dfull = np.zeros_like(logits2)
    dfull[idx] = dl2
    assert abs(eng2.loss_out.item() - float(loss2)) < TOL * max(1.0, abs(float(loss2))), tag
    grads2 = tt.gcn_backward(st2, cache2, dfull)
    gmax2 = max(float(np.abs(gr).max()) for gr in grads2)
    edge2 = min(float(np.abs(t["pre"]).min() / np.abs(t["pre"]).max()) for t in cache2["tails"])
    for (pname, prm), gr in zip(gcn.named_parameters(), grads2):
        if edge2 < 3e-7:
            break
        np.testing.assert_allclose(prm.grad.cpu().numpy(), gr, atol=2e-5 * max(1.0, gmax2), rtol=1e-4, err_msg=f"{tag} gcn {pname} edge={edge2:.2e}")


@pytest.mark.parametrize("norm,p,full", [("batch", 0.3, False), ("none", 0.5, False), ("batch", 0.0, False), ("batch", 0.4, True)])
def test_sage_step_tail_in_the_gather_is_bit_identical_to_the_materialised_tail(norm, p, full, monkeypatch):
    """glnn_sage_layer.h == NULL (round 4, the default of TeacherEngine): a hidden layer's h = dropout(relu(norm(z))) is not written, the
    next layer's aggregation applies that tail to every z row it gathers (glnn::spmm_csr_tail -- hub rows through the long-row role,
    self rows, rows gathered many times) with act_fwd's arithmetic element for element: three optimiser steps end in the same
    parameters, BatchNorm buffers and loss bit for bit as with GLNN_TEACHER_GATHER_TAIL=0 (the act_fwd launches)."""
    from glnn_amd import ops
    from glnn_amd.graph import MultiLayerFullNeighborSampler, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    n, dims = 20000, [40, 64, 64, 9]
    indptr, indices = random_graph(n, 10, seed=9, power=0.6, hub=5000, isolated=30)
    rs = np.random.RandomState(9)
    fd = ops.as_feat(torch.from_numpy(rs.standard_normal((n, dims[0])).astype(np.float32)).to(DEV))
    ld = torch.from_numpy(rs.randint(0, dims[-1], n).astype(np.int64)).to(DEV)
    g = _graph(indptr, indices)
    if full:      # full neighbourhoods: the 5000-edge hub row is a destination of the inner blocks -> the long-row role applies the tail too
        batches = list(NodeDataLoader(g, torch.arange(768), MultiLayerFullNeighborSampler(3), batch_size=256, shuffle=False, seed=5))
        assert max(int(b.in_degrees().max()) for b in batches[0][2][1:]) > 128
    else:
        batches = list(NodeDataLoader(g, torch.arange(1536), MultiLayerNeighborSampler([5, 10, 15]), batch_size=512, shuffle=False, seed=5))
    states = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GLNN_TEACHER_GATHER_TAIL", mode)
        torch.manual_seed(2)
        model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type=norm, device=DEV))
        opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=0.0)
        model.train()
        eng = TeacherEngine(model, opt)
        assert eng.gather_tail == (mode == "1")
        for input_nodes, output_nodes, blocks in batches:
            eng.step_sage(blocks, fd, ld, output_nodes, 1.0, input_nodes=input_nodes)
        torch.cuda.synchronize()
        states.append([t.detach().clone() for t in model.state_dict().values()] + [eng.loss_out.clone()])
    diffs = [float((a.double() - b.double()).abs().max()) for a, b in zip(*states)]
    assert all(torch.equal(a, b) for a, b in zip(*states)), diffs
    assert bool(torch.isfinite(states[0][-1]).all())


@pytest.mark.parametrize("norm,p,wd", [("batch", 0.3, 0.0), ("none", 0.5, 5e-4), ("batch", 0.0, 5e-4)])
def test_one_call_sage_train_step_equals_fwd_bwd_plus_adam_bit_for_bit(norm, p, wd, monkeypatch):
    """glnn_sage_train_step_f32 (round 6): forward + NLL + backward + Adam of the sampled-block teacher in ONE call, the backward's last
    partial sums -- the split slabs of every weight gradient, the column partials behind the last layer's bias gradient, the per-workgroup
    losses -- folded by the Adam launch instead of by six launches of their own.  Same partials, same order: three steps end in the same
    parameters, moments, BatchNorm buffers, gradients and loss as glnn_sage_fwd_bwd_f32 + glnn_adam_step_f32 (GLNN_TEACHER_ONE_CALL=0), bit for
    bit (and with GLNN_STUDENT_ADAM_FOLDS=0: the one-call form without the folds)."""
    from glnn_amd import ops
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    n, dims = 60000, [100, 128, 128, 9]
    indptr, indices = random_graph(n, 10, seed=11, power=0.6, hub=3000, isolated=30)
    rs = np.random.RandomState(11)
    fd = ops.as_feat(torch.from_numpy(rs.standard_normal((n, dims[0])).astype(np.float32)).to(DEV))
    ld = torch.from_numpy(rs.randint(0, dims[-1], n).astype(np.int64)).to(DEV)
    g = _graph(indptr, indices)
    batches = list(NodeDataLoader(g, torch.arange(3072), MultiLayerNeighborSampler([5, 10, 15]), batch_size=1024, shuffle=False, seed=5))
    states = []
    for one_call, folds in (("0", "1"), ("1", "1"), ("1", "0")):
        monkeypatch.setenv("GLNN_TEACHER_ONE_CALL", one_call)
        monkeypatch.setenv("GLNN_STUDENT_ADAM_FOLDS", folds)
        torch.manual_seed(2)
        model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type=norm, device=DEV))
        opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=wd)
        model.train()
        eng = TeacherEngine(model, opt)
        assert eng._one_call == (one_call == "1")
        for input_nodes, output_nodes, blocks in batches:
            eng.step_sage(blocks, fd, ld, output_nodes, 1.0, input_nodes=input_nodes)
        torch.cuda.synchronize()
        states.append([t.detach().clone() for t in model.state_dict().values()] + [opt.state[q]["exp_avg"].clone() for q in model.parameters()]
                      + [eng.grad(q).clone() for q in model.parameters()] + [eng.loss_out.clone()])
    for other in states[1:]:
        diffs = [float((a.double() - b.double()).abs().max()) for a, b in zip(states[0], other)]
        assert all(torch.equal(a, b) for a, b in zip(states[0], other)), diffs
    assert bool(torch.isfinite(states[0][-1]).all())


@pytest.mark.parametrize("p", [0.5, 0.0])
def test_sage_step_bn_backward_apply_inside_the_weight_gradient_gemm(p, monkeypatch):
    """Round 5: the outermost block's dz has one consumer, dW_0.  Layer 0's BatchNorm backward is then spread over its neighbours:
    (dy) the transposed aggregation that produces da stores dy -- da behind the dropout / ReLU masks -- instead, and leaves the column sums of
    the rows it finishes (spmm_bn_dy_kernel: da is never written; GLNN_SAGE_FUSE_BN_DY=0: a pass of its own, bn_bwd_partial's bits);
    (apply) dz = alpha dy + beta z + gamma is evaluated on the staged operand pieces of the pipelined weight-gradient kernel
    (glnn::gemm_tn(..., bn)): dz_0 is never written (GLNN_SAGE_FUSE_BN_APPLY=0: the plain three-pass form).
    Against the plain form: same loss and same gradients behind layer 0 bit for bit; layer 0's BatchNorm parameter gradients bit for bit with
    the separate dy pass, to summation order with the epilogue; dW_0 to rounding (an affine map instead of bn_dz's expression); the bias in
    front of the BatchNorm -- true gradient 0, rounding noise in the plain form -- exactly 0.  The block is big enough (> 64 row chunks of
    layer-0 destinations, hub rows among them) for the deferred forms to engage."""
    from glnn_amd import ops
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    n, dims = 120000, [100, 256, 256, 47]
    # symmetric: the 3000-edge hub is also a SOURCE of 3000 rows, so the transposed block of layer 1 has a long row (the long-row role of
    # the aggregation whose epilogue is the dy pass)
    indptr, indices = random_graph(n, 6, seed=21, power=0.6, hub=3000, isolated=50, symmetric=True)
    rs = np.random.RandomState(21)
    fd = ops.as_feat(torch.from_numpy(rs.standard_normal((n, dims[0])).astype(np.float32)).to(DEV))
    ld = torch.from_numpy(rs.randint(0, dims[-1], n).astype(np.int64)).to(DEV)
    g = _graph(indptr, indices)
    (input_nodes, output_nodes, blocks), = list(NodeDataLoader(g, torch.arange(2048), MultiLayerNeighborSampler([5, 10, 15]), batch_size=2048,
                                                                shuffle=False, seed=5))
    assert blocks[0].num_dst_nodes() > 64 * 128
    assert int(torch.bincount(blocks[1].indices.long()).max()) > 128, "no long row in the transposed block of layer 1"
    grads, losses = {}, {}
    for mode in ("11", "10", "00", "11 again"):
        monkeypatch.setenv("GLNN_SAGE_FUSE_BN_APPLY", mode[0])
        monkeypatch.setenv("GLNN_SAGE_FUSE_BN_DY", mode[1])
        torch.manual_seed(2)
        model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type="batch", device=DEV))
        opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=0.0)
        model.train()
        eng = TeacherEngine(model, opt)
        eng.step_sage(blocks, fd, ld, output_nodes, 1.0, input_nodes=input_nodes)
        torch.cuda.synchronize()
        grads[mode] = {k: prm.grad.detach().clone() for k, prm in model.named_parameters()}
        losses[mode] = eng.loss_out.clone()
    assert torch.equal(losses["11"], losses["00"]) and torch.equal(losses["10"], losses["00"])
    assert all(torch.equal(grads["11"][k], grads["11 again"][k]) for k in grads["11"])       # the epilogue's column sums are deterministic
    for mode in ("11", "10"):
        for k in grads["00"]:
            a, b = grads[mode][k], grads["00"][k]
            scale = max(1.0, float(b.abs().max()))
            if k.startswith("encoder.layers.0") and k.endswith("fc_neigh.bias"):
                assert float(a.abs().max()) == 0.0 and float(b.abs().max()) < 1e-4, (mode, k)
            elif k.startswith("encoder.layers.0"):
                assert float((a - b).abs().max()) <= 2e-5 * scale, (mode, k, float((a - b).abs().max()), scale)
            elif k.startswith("encoder.norms.0") and mode == "11":
                assert float((a - b).abs().max()) <= 2e-5 * scale, (mode, k, float((a - b).abs().max()), scale)     # (other summation order)
            else:
                assert torch.equal(a, b), (mode, k)        # behind layer 0: the same launches; "10": the norm's sums are bn_bwd_partial's bits
    assert not torch.equal(grads["11"]["encoder.layers.0.fc_neigh.weight"], grads["00"]["encoder.layers.0.fc_neigh.weight"])      # (the switches did switch:
    assert not torch.equal(grads["11"]["encoder.norms.0.weight"], grads["10"]["encoder.norms.0.weight"])    #  the epilogue's column sums have their own order)


@pytest.mark.parametrize("norm,p,full,gather_tail,hidden", [("batch", 0.3, False, "1", 256), ("none", 0.5, False, "1", 136), ("batch", 0.4, True, "1", 256),
                                                            ("batch", 0.2, False, "0", 72), ("batch", 0.3, "sparse", "0", 256),
                                                            ("none", 0.0, "sparse", "1", 100)])
def test_sage_step_short_row_aggregation_is_bit_identical_to_one_row_per_wave(norm, p, full, gather_tail, hidden, monkeypatch):
    """Round 5: the outermost block of a sampled batch (<= 6 in-edges per row on average) is aggregated by spmm_csr_short_kernel (four rows
    per wave in flight: one request for their indptr entries, one for their indices, then the first load units of all four rows together;
    longer rows continue alone with their sums carried on; rows above the long-row threshold stay with the shared long-row role).  A lane
    group sees a row's edges in the same order as in spmm_csr_kernel: three optimiser steps end in the same parameters, BatchNorm buffers
    and loss bit for bit as with GLNN_SPMM_SHORT=0 -- sampled fan-outs (5 for the outermost block: the short kernel; with materialised
    activations also for an inner block when its average degree allows) and full neighbourhoods (average degree above the bound: both
    modes run the row kernel, the switch must then change nothing)."""
    from glnn_amd import ops
    from glnn_amd.graph import MultiLayerFullNeighborSampler, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.teacher import TeacherEngine
    n, dims = 20000, [100, hidden, hidden, 9]               # 100 / 72 and 136 / 256 floats per row: both lane layouts the short kernel is built for
    if full == "sparse":
        # full neighbourhoods of a SPARSE graph (3 in-edges per row on average) with a power-law tail and a 600-edge hub among the seeds: the
        # short kernel runs on every block, and every path of it is taken -- batch, carried tail (rows of 4 .. 128 edges), lone row, long row
        indptr, indices = random_graph(n, 3, seed=9, power=0.8, hub=600, isolated=30)
        hub = int(np.argmax(np.diff(indptr)))
        seeds = torch.from_numpy(np.concatenate([[hub], np.setdiff1d(np.arange(600), [hub])[:511]]).astype(np.int64))
    else:
        indptr, indices = random_graph(n, 10, seed=9, power=0.6, hub=5000, isolated=30)
    rs = np.random.RandomState(9)
    fd = ops.as_feat(torch.from_numpy(rs.standard_normal((n, dims[0])).astype(np.float32)).to(DEV))
    ld = torch.from_numpy(rs.randint(0, dims[-1], n).astype(np.int64)).to(DEV)
    g = _graph(indptr, indices)
    if full == "sparse":
        batches = list(NodeDataLoader(g, seeds, MultiLayerFullNeighborSampler(3), batch_size=256, shuffle=False, seed=5))
        b0 = batches[0][2][0]
        deg0 = b0.in_degrees()
        assert int(deg0.max()) > 128 and int(((deg0 > 3) & (deg0 <= 128)).sum()) > 10 and float(deg0.float().mean()) <= 6.0
    elif full:
        batches = list(NodeDataLoader(g, torch.arange(768), MultiLayerFullNeighborSampler(3), batch_size=256, shuffle=False, seed=5))
    else:
        batches = list(NodeDataLoader(g, torch.arange(1536), MultiLayerNeighborSampler([5, 10, 15]), batch_size=512, shuffle=False, seed=5))
    monkeypatch.setenv("GLNN_TEACHER_GATHER_TAIL", gather_tail)
    states = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GLNN_SPMM_SHORT", mode)
        torch.manual_seed(2)
        model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type=norm, device=DEV))
        opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=0.0)
        model.train()
        eng = TeacherEngine(model, opt)
        for input_nodes, output_nodes, blocks in batches:
            eng.step_sage(blocks, fd, ld, output_nodes, 1.0, input_nodes=input_nodes)
        torch.cuda.synchronize()
        states.append([t.detach().clone() for t in model.state_dict().values()] + [eng.loss_out.clone()])
    diffs = [float((a.double() - b.double()).abs().max()) for a, b in zip(*states)]
    assert all(torch.equal(a, b) for a, b in zip(*states)), diffs
    assert bool(torch.isfinite(states[0][-1]).all())


def test_loaders_of_one_device_share_their_side_stream():
    """Batches are built one ahead on a side stream; the caching allocator keeps a pool per stream, so every loader of a device uses the
    same one (a stream per loader grew the reserved memory by ~2 GB per loader on the products configuration)."""
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    indptr, indices = random_graph(3000, 6, seed=4)
    g = _graph(indptr, indices)
    sides = []
    for k in range(3):
        loader = NodeDataLoader(g, torch.arange(512), MultiLayerNeighborSampler([3, 3]), batch_size=128, shuffle=True, seed=k)
        assert len(list(loader)) == 4
        sides.append(loader._side)
    assert sides[0] is not None and all(s is sides[0] for s in sides)


def test_sampled_training_epoch_is_reproducible_within_and_across_processes():
    """scripts/train_determinism_probe.py on the arxiv configuration: graph generation -> neighbour sampling -> block building -> step -> Adam.
    In one process, with the batches built in line, one ahead on the side stream, and by the loader's worker thread: the same blocks, losses
    and parameters bit for bit; so with the outermost block built as a global-id block only (what train_sage asks its loader for).  In
    two processes: the same parameters (sha256) -- round 5 found the "seeded" synthetic graph in two variants from process to process (a
    device cumsum in float64 under its power-law endpoints: a look-back scan whose grouping follows timing; now summed on the host)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_determinism_probe.py"), "ogbn-arxiv", "8"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.splitlines()
        same = [ln for ln in lines if "blocks equal" in ln]
        assert len(same) == 5 and all("blocks equal True, losses equal True" in ln and "parameters equal True" in ln for ln in same), same
        outs.append([ln for ln in lines if ln.startswith("sha256")])
    assert outs[0] and outs[0] == outs[1], outs


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_train_sage_with_the_engine_mode_loader_equals_the_default_blocks_on_random_graphs(seed):
    """train_sage asks its loader for global-id outermost blocks and prebuilt transposes (round 5); stepping TeacherEngine by hand over the
    loader's default (fully relabelled) blocks is the same training bit for bit -- on random graphs with isolated nodes, hubs, fan-outs above
    and below the degrees, 1-3 layers, hidden widths on both sides of the dy-in-aggregation form's limits, a ragged last batch, dropout."""
    import copy
    from glnn_amd import teacher, train_and_eval as te
    from glnn_amd.graph import CSRGraph, MultiLayerFullNeighborSampler, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    rs = np.random.RandomState(100 + seed)
    n = int(rs.choice([300, 2000, 9000]))
    indptr, indices = random_graph(n, float(rs.choice([2, 6, 20])), seed=seed, power=float(rs.choice([0.0, 0.6])), hub=int(rs.choice([0, n // 3])),
                                   isolated=int(rs.choice([0, 11])))
    g = CSRGraph(torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV), n)
    L = int(rs.choice([1, 2, 3]))
    f, h, c = int(rs.choice([16, 100])), int(rs.choice([64, 128, 256])), int(rs.choice([5, 47]))
    p = float(rs.choice([0.0, 0.5]))
    sampler = MultiLayerNeighborSampler([int(v) for v in rs.choice([2, 5, 15], L)]) if rs.rand() < 0.8 else MultiLayerFullNeighborSampler(L)
    torch.manual_seed(seed)
    base = Model(dict(model_name="SAGE", num_layers=L, feat_dim=f, hidden_dim=h, label_dim=c, dropout_ratio=p, norm_type=str(rs.choice(["batch", "none"])),
                      device=DEV))
    feats = torch.randn(n, f, device=DEV)
    labels = torch.randint(0, c, (n,), device=DEV)
    nids = torch.from_numpy(rs.permutation(n)[: int(0.7 * n)].astype(np.int64)).to(DEV)
    bs = int(rs.choice([64, 257]))
    outs = []
    for engine_mode in (False, True):
        model = copy.deepcopy(base)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
        loader = NodeDataLoader(g, nids, sampler, batch_size=bs, shuffle=False, drop_last=False, seed=77)
        if engine_mode:
            mean = te.train_sage(model, loader, feats, labels, torch.nn.NLLLoss(), opt)
            assert loader.global_first_block is False                      # restored
        else:
            eng = teacher.get_engine(model, opt)
            eng.loss_accum.zero_()
            k = 0
            for input_nodes, output_nodes, blocks in loader:
                assert input_nodes is not None and blocks[-1].t_indptr is None
                eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
                k += 1
            eng.sync_optimizer_state()
            mean = eng.loss_accum.item() / k
        outs.append((mean, [v.detach().clone() for v in model.state_dict().values()]))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
