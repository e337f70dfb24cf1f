"""N > 1 path on CPU: world_size-2 `gloo` processes run glnn_amd.dist.ShardedTeacher with an ORACLE-backed
compute stand-in (same signatures as glnn_amd.ops), and the DP gradient all-reduce.  What is verified is the
sharding / slot / all-gather orchestration: the gathered result must equal the unsharded oracle forward."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """CPU stand-in for glnn_amd.ops built on the oracle (tests only)."""
    AGG_SUM, AGG_SAGE_GCN = 0, 1

    def __init__(self):
        from oracle import teacher_oracle as to
        self.to = to

    @staticmethod
    def feat_empty(n, d, device, zero=False):
        # same layout contract as glnn_amd.ops.feat_empty: [n, d] view of a [n, round4(d)] buffer
        return torch.full((n, (d + 3) // 4 * 4), float("nan"))[:, :d]

    @staticmethod
    def as_feat(t):
        return t

    def gemm(self, a, w, ep_scale=None, ep_shift=None, relu=False, out=None, **kw):
        y = torch.from_numpy(self.to.linear(a.contiguous().numpy(), w.detach().contiguous().numpy(), None))
        if ep_scale is not None:
            y = y * ep_scale
        if ep_shift is not None:
            y = y + ep_shift.detach()
        if relu:
            y = y.clamp(min=0)
        if out is None:
            return y
        out.copy_(y)
        return out

    def spmm(self, indptr, indices, x, n_dst, mode, ep_scale=None, ep_shift=None, relu=False, out=None, x_self=None, row_scale=None, **kw):
        s = self.to.spmm_sum(indptr.numpy(), indices.numpy(), x.contiguous().numpy(), n_dst=n_dst)
        if mode == self.AGG_SAGE_GCN:
            xs = x if x_self is None else x_self
            deg = (indptr[1:] - indptr[:-1]).float().unsqueeze(1)
            y = (torch.from_numpy(s) + xs[:n_dst]) / (deg + 1)
        else:
            y = torch.from_numpy(s)
            if row_scale is not None:
                y = y * row_scale.unsqueeze(1)
        if ep_scale is not None:
            y = y * ep_scale
        if ep_shift is not None:
            y = y + ep_shift.detach()
        if relu:
            y = y.clamp(min=0)
        if out is None:
            return y
        out.copy_(y)
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, dims, seed, chunks, balanced, widening, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd.dist import RowShards, ShardedTeacher, make_grad_sync
        from glnn_amd.graph import CSRGraph
        from glnn_amd.models import SAGE
        from graphgen import random_graph
        import torch.nn.functional as F
        indptr, indices = random_graph(n, 9, seed=seed, power=0.5, isolated=3, hub=200)
        g = CSRGraph(torch.from_numpy(indptr), torch.from_numpy(indices), n)
        torch.manual_seed(seed)
        enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
        with torch.no_grad():
            for bn in enc.norms:
                bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
        enc.eval()
        x = torch.from_numpy(np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32))
        bounds = RowShards.balanced_bounds(g.indptr, world) if balanced else None      # cut by work: uneven row ranges
        sh = RowShards(n, world, rank, chunks=chunks, bounds=bounds)
        be = OracleBackend()
        from glnn_amd import dist as gdist
        gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
        with torch.no_grad():
            y_own = ShardedTeacher(enc, g.row_range(sh.lo, sh.hi), sh, be, widening_exchange=widening.split("-")[0],
                                   mixed_quantum=8 if widening.endswith("cut") else 0).forward(x)
        stats = dict(gdist.EXCHANGE_STATS, n_pad=sh.n_pad)
        # DP gradient exchange
        flat = torch.full((10,), float(rank + 1))
        make_grad_sync(flat, world, average=True)()
        q.put((rank, sh.lo, sh.hi, y_own.numpy().copy(), flat.numpy().copy(), stats))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,dims,chunks,balanced,widening", [
    (2, 1001, [12, 16, 16, 5], 1, False, "narrow"), (2, 640, [8, 24, 6], 1, False, "narrow"), (2, 1003, [8, 24, 24, 6], 3, False, "narrow"),
    (2, 90, [6, 16, 5], 4, False, "narrow"), (2, 1001, [12, 16, 16, 5], 1, True, "narrow"), (4, 1003, [8, 24, 24, 6], 3, True, "narrow"),
    (4, 777, [8, 24, 6], 1, True, "narrow"), (4, 640, [12, 16, 16, 5], 2, False, "narrow"), (8, 1001, [8, 24, 24, 6], 2, True, "narrow"),
    (8, 203, [6, 16, 5], 1, False, "narrow"),
    (2, 1003, [8, 24, 24, 6], 3, False, "wide"), (2, 1003, [8, 24, 24, 6], 1, True, "wide"), (4, 1003, [8, 24, 24, 6], 4, True, "wide"),
    (2, 90, [6, 16, 5], 4, False, "wide"),
    (2, 1003, [8, 24, 24, 6], 3, False, "mixed"), (4, 1003, [8, 24, 24, 6], 2, True, "mixed"), (2, 90, [6, 16, 5], 4, False, "mixed"),
    (2, 1003, [8, 24, 24, 6], 3, False, "mixed-cut"), (4, 1003, [8, 24, 24, 6], 2, True, "mixed-cut")])
def test_sharded_teacher_gloo_equals_unsharded(world, n, dims, chunks, balanced, widening):
    """world 2 / 4 / 8, equal and work-balanced (uneven) row ranges, plain and chunked-overlapped exchange; a widening layer
    exchanging its narrow aggregate (replicated projection), its wide output, or (round 6, "mixed") half of every chunk's rows each way --
    unequal chunks in the chunk-major layout (bench.py --layer1-exchange)."""
    sys.path.insert(0, ROOT)
    from oracle import teacher_oracle as to
    from graphgen import random_graph
    import torch.nn.functional as F
    from glnn_amd.models import SAGE
    seed = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, dims, seed, chunks, balanced, widening, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded reference on the same seeded inputs
    indptr, indices = random_graph(n, 9, seed=seed, power=0.5, isolated=3, hub=200)
    torch.manual_seed(seed)
    enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
    with torch.no_grad():
        for bn in enc.norms:
            bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
    sd = {k: v.numpy() for k, v in enc.state_dict().items()}
    L = len(dims) - 1
    layers = [dict(weight=sd[f"layers.{i}.fc_neigh.weight"], bias=sd[f"layers.{i}.fc_neigh.bias"]) for i in range(L)]
    norms = [dict(weight=sd[f"norms.{i}.weight"], bias=sd[f"norms.{i}.bias"], running_mean=sd[f"norms.{i}.running_mean"],
                  running_var=sd[f"norms.{i}.running_var"]) for i in range(L - 1)]
    x = np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32)
    want = to.sage_inference(indptr, indices, x, layers, norms)
    covered = np.zeros(n, bool)
    # what a forward may move: per node, the NARROW side of every layer boundary that needs remote rows -- the aggregate of
    # a widening layer (2*in <= out), the projected rows of a narrowing layer (in > out), the output of any other layer
    # unless the next layer is narrowing (it projects its own rows only)
    r4 = lambda d: (d + 3) // 4 * 4
    per_node = 0
    for l in range(L):
        d_in, d_out = dims[l], dims[l + 1]
        if d_in > d_out:
            per_node += r4(d_out)
        elif l < L - 1 and 2 * d_in <= d_out and widening in ("narrow", "mixed", "mixed-cut"):
            per_node += r4(d_in)
        elif l < L - 1 and not dims[l + 1] > dims[l + 2]:
            per_node += r4(d_out)
    for rank, lo, hi, y, flat, stats in res:
        np.testing.assert_allclose(y, want[lo:hi], atol=1e-4, rtol=0)
        covered[lo:hi] = True
        if not widening.startswith("mixed"):      # (mixed: W chunks move d_out floats per row, N chunks d_in: counted in test_emulated_rank_equals_unsharded_rows)
            assert stats["floats_received"] == stats["n_pad"] * per_node, (stats, per_node)
        np.testing.assert_allclose(flat, np.full(10, (world + 1) / 2))      # mean of 1..world
    assert covered.all()
    if balanced:        # the work-balanced cut really is uneven on this power-law graph, and evens out the edge counts
        rows = sorted(hi - lo for _, lo, hi, *_ in res)
        assert rows[0] < rows[-1]
        nnz = [int(indptr[hi] - indptr[lo]) for _, lo, hi, *_ in res]
        assert max(nnz) - min(nnz) <= 0.5 * max(nnz) + 250


def _halo_worker(rank, world, port, n, dims, seed, shuffle, overlap, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd import data
        from glnn_amd import dist as gdist
        from glnn_amd.models import SAGE
        import torch.nn.functional as F
        g = data.make_clustered_graph(n, 10, communities=16, p_in=0.95, seed=seed, shuffle_ids=shuffle)
        torch.manual_seed(seed)
        enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
        with torch.no_grad():
            for bn in enc.norms:
                bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
        enc.eval()
        x = torch.from_numpy(np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32))
        sh = gdist.RowShards(n, world, rank, bounds=gdist.RowShards.balanced_bounds(g.indptr, world))
        t = gdist.HaloShardedTeacher(enc, g.row_range(sh.lo, sh.hi), sh, OracleBackend(), overlap=overlap)
        gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
        with torch.no_grad():
            y_own = t.forward(x)
        q.put((rank, sh.lo, sh.hi, y_own.numpy().copy(), dict(gdist.EXCHANGE_STATS), t.plan.n_halo, sum(t.plan.send_counts)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,shuffle,overlap", [
    (2, [12, 16, 16, 5], False, False), (4, [8, 24, 24, 6], False, False), (4, [8, 24, 6], True, False), (4, [16, 16, 16, 4], False, False),
    (2, [12, 16, 16, 5], False, True), (4, [8, 24, 24, 6], False, True), (4, [8, 24, 6], True, True), (4, [16, 16, 16, 4], False, True),
    (4, [8, 16, 16, 16], False, True)])
def test_halo_sharded_teacher_gloo_equals_unsharded(world, dims, shuffle, overlap):
    """The halo exchange (only the referenced remote rows, all-to-all over index lists built at partition time), synchronous
    and OVERLAPPED (asynchronous all-to-all hidden behind the own-row projection of a widening layer / the local-source pass of
    a two-pass aggregation over the split CSR): result == the unsharded oracle forward, same bytes on the wire; on the locality-ordered graph the bytes moved are a fraction of the all-gather's, on the
    id-shuffled copy of the same graph (no locality) they approach it."""
    sys.path.insert(0, ROOT)
    from oracle import teacher_oracle as to
    from glnn_amd import data
    import torch.nn.functional as F
    from glnn_amd.models import SAGE
    n, seed = 1600, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, n, dims, seed, shuffle, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = data.make_clustered_graph(n, 10, communities=16, p_in=0.95, seed=seed, shuffle_ids=shuffle)
    torch.manual_seed(seed)
    enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
    with torch.no_grad():
        for bn in enc.norms:
            bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
    sd = {k: v.numpy() for k, v in enc.state_dict().items()}
    L = len(dims) - 1
    layers = [dict(weight=sd[f"layers.{i}.fc_neigh.weight"], bias=sd[f"layers.{i}.fc_neigh.bias"]) for i in range(L)]
    norms = [dict(weight=sd[f"norms.{i}.weight"], bias=sd[f"norms.{i}.bias"], running_mean=sd[f"norms.{i}.running_mean"],
                  running_var=sd[f"norms.{i}.running_var"]) for i in range(L - 1)]
    x = np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32)
    want = to.sage_inference(g.indptr.numpy(), g.indices.numpy(), x, layers, norms)
    covered = np.zeros(n, bool)
    r4 = lambda d: (d + 3) // 4 * 4
    per_node = 0        # floats per halo row and forward: the narrow side of every boundary that is exchanged
    for l in range(L):
        d_in, d_out = dims[l], dims[l + 1]
        if d_in > d_out:
            per_node += r4(d_out)
        elif l < L - 1 and 2 * d_in <= d_out:
            per_node += r4(d_in)
        elif l < L - 1 and not dims[l + 1] > dims[l + 2]:
            per_node += r4(d_out)
    for rank, lo, hi, y, stats, n_halo, n_send in res:
        np.testing.assert_allclose(y, want[lo:hi], atol=1e-4, rtol=0)
        covered[lo:hi] = True
        assert stats["floats_received"] == n_halo * per_node, (stats, n_halo, per_node)
        remote_rows = n - (hi - lo)
        if shuffle:
            assert n_halo > 0.8 * remote_rows            # no locality: nearly every remote row is referenced
        else:
            assert n_halo < 0.55 * remote_rows, (n_halo, remote_rows)
    assert covered.all()
    assert sum(r[5] for r in res) == sum(r[6] for r in res)            # every requested row is sent by exactly one owner


def _partition_worker(rank, world, port, n, dims, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd import data
        from glnn_amd import dist as gdist
        from glnn_amd.models import SAGE
        import torch.nn.functional as F
        g0 = data.make_clustered_graph(n, 10, communities=16, p_in=0.95, seed=seed, shuffle_ids=True)     # communities exist, ids say nothing
        perm = data.locality_order(g0, seed=1)              # deterministic: every rank derives the same order
        g = data.relabel(g0, perm)
        torch.manual_seed(seed)
        enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
        with torch.no_grad():
            for bn in enc.norms:
                bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
        enc.eval()
        x0 = torch.from_numpy(np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32))
        x = x0[perm]
        bounds = gdist.RowShards.balanced_bounds(g.indptr, world)
        sh = gdist.RowShards(n, world, rank, bounds=bounds)
        halo_before = gdist.HaloPlan(g0.row_range(*[gdist.RowShards.balanced_bounds(g0.indptr, world)[rank + i] for i in (0, 1)]),
                                     gdist.RowShards(n, world, rank, bounds=gdist.RowShards.balanced_bounds(g0.indptr, world))).n_halo
        t = gdist.HaloShardedTeacher(enc, g.row_range(sh.lo, sh.hi), sh, OracleBackend(), overlap=True)
        gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
        with torch.no_grad():
            y_own = t.forward(x)
        q.put((rank, sh.lo, sh.hi, y_own.numpy().copy(), perm.numpy().copy(), dict(gdist.EXCHANGE_STATS), t.plan.n_halo, halo_before))
    finally:
        dist.destroy_process_group()


def test_locality_partitioner_makes_the_halo_exchange_pay():
    """A clustered graph whose node ids were shuffled (communities exist, the id order hides them): data.locality_order (label
    propagation over the CSR) + data.relabel give node ranges with locality again -- per rank the halo shrinks to < 0.6 x the
    remote rows (an all-gather moves all of them; before the reorder the halo is > 0.8 x), the bytes received follow, and the
    overlapped halo forward on the relabelled graph equals the unsharded forward on the ORIGINAL graph, row for row."""
    sys.path.insert(0, ROOT)
    from oracle import teacher_oracle as to
    from glnn_amd import data
    import torch.nn.functional as F
    from glnn_amd.models import SAGE
    world, n, seed, dims = 4, 1600, 7, [8, 24, 24, 6]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partition_worker, args=(r, world, port, n, dims, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0 = data.make_clustered_graph(n, 10, communities=16, p_in=0.95, seed=seed, shuffle_ids=True)
    torch.manual_seed(seed)
    enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
    with torch.no_grad():
        for bn in enc.norms:
            bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
    sd = {k: v.numpy() for k, v in enc.state_dict().items()}
    L = len(dims) - 1
    layers = [dict(weight=sd[f"layers.{i}.fc_neigh.weight"], bias=sd[f"layers.{i}.fc_neigh.bias"]) for i in range(L)]
    norms = [dict(weight=sd[f"norms.{i}.weight"], bias=sd[f"norms.{i}.bias"], running_mean=sd[f"norms.{i}.running_mean"],
                  running_var=sd[f"norms.{i}.running_var"]) for i in range(L - 1)]
    x0 = np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32)
    want = to.sage_inference(g0.indptr.numpy(), g0.indices.numpy(), x0, layers, norms)       # original ids
    per_node = 8 + 8            # r4(8) aggregate of the widening layer + r4(6) projected rows of the narrowing one
    seen = np.zeros(n, bool)
    for rank, lo, hi, y, perm, stats, n_halo, halo_before in res:
        np.testing.assert_allclose(y, want[perm[lo:hi]], atol=1e-4, rtol=0)       # new row r is old node perm[r]
        seen[perm[lo:hi]] = True
        remote = n - (hi - lo)
        assert n_halo < 0.6 * remote, (n_halo, remote)
        assert halo_before > 0.8 * remote, (halo_before, remote)
        assert stats["floats_received"] == n_halo * per_node                        # vs remote * per_node for an all-gather
    assert seen.all()


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd.dist import StatExchange
        h = 5
        ex = StatExchange(world, rank, h, "cpu")
        # what the library does per BatchNorm layer: fill `send`, call the hook with the descriptor's pointers
        rs = np.random.RandomState(10 + rank)
        rows = 7 + 4 * rank
        zloc = rs.standard_normal((rows, h)).astype(np.float32)
        ex.send[:h] = rows
        ex.send[h:2 * h] = torch.from_numpy(zloc.mean(0))
        ex.send[2 * h:3 * h] = torch.from_numpy(((zloc - zloc.mean(0)) ** 2).sum(0))
        rc = ex.callback(None, ex.send.data_ptr(), ex.recv.data_ptr(), 3 * h, None)
        bad = ex.callback(None, ex.send.data_ptr() + 4, ex.recv.data_ptr(), 3 * h, None)     # foreign buffer -> refused
        q.put((rank, rc, bad, repr(ex.error), ex.calls, ex.recv[:world * 3 * h].numpy().copy(), zloc))
    finally:
        dist.destroy_process_group()


def test_stat_exchange_hook_world2_gloo():
    """The glnn_exchange_fn host hook (glnn_amd.dist.StatExchange): rank-ordered all-gather of the (count, mean, M2)
    triples; combining them in rank order (Chan) reproduces the whole-batch statistics on every rank."""
    world, h = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = np.concatenate([r[6] for r in res])
    for rank, rc, bad, err, calls, recv, _ in res:
        assert rc == 0 and bad == 1 and "does not own" in err and calls == 1
        np.testing.assert_array_equal(recv, res[0][5])                 # identical gathered buffer on every rank
        n = mean = m2 = 0.0
        for r in range(world):
            nb, mb, qb = (recv[r * 3 * h + j * h:r * 3 * h + (j + 1) * h].astype(np.float64) for j in range(3))
            delta, nn = mb - mean, n + nb
            mean = mean + delta * nb / nn
            m2 = m2 + qb + delta * delta * n * nb / nn
            n = nn
        np.testing.assert_allclose(mean, z.mean(0), atol=1e-6)
        np.testing.assert_allclose(m2 / n, z.var(0), atol=1e-6)


def _probe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd.dist import probe_link
        q.put((rank, probe_link(world, rank, 50_000, "cpu", reps=2)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_link_probe_measures_an_in_place_all_gather_and_returns_the_same_record_on_every_rank(world):
    """dist.probe_link (round 5: what bench.py --gpus N runs before it picks its exchange form): an in-place all-gather of one slab per
    rank on the job's own transport, checked for content; seconds = the maximum over ranks, so every rank derives the same rates."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    first = res[0][1]
    assert first["correct"] and first["backend"] == "gloo" and first["reps"] == 2 and abs(first["slab_MB"] - 0.2) < 1e-9
    assert first["seconds"] > 0 and abs(first["received_GBps"] - (world - 1) * first["per_link_GBps"]) < 1e-9 * first["received_GBps"] + 1e-12
    assert abs(first["per_link_GBps"] - 4.0 * 50_000 / first["seconds"] / 1e9) < 1e-12
    for _, rec in res[1:]:
        assert rec == first


def _emu_setup(n, dims, seed, clustered=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from glnn_amd import data
    from glnn_amd.graph import CSRGraph
    from glnn_amd.models import SAGE
    from graphgen import random_graph
    import torch.nn.functional as F
    if clustered:
        g = data.make_clustered_graph(n, 10, communities=16, p_in=0.95, seed=seed)
    else:
        indptr, indices = random_graph(n, 9, seed=seed, power=0.5, isolated=3, hub=200)
        g = CSRGraph(torch.from_numpy(indptr), torch.from_numpy(indices), n)
    torch.manual_seed(seed)
    enc = SAGE(len(dims) - 1, dims[0], dims[1], dims[-1], 0.5, F.relu, "batch")
    with torch.no_grad():
        for bn in enc.norms:
            bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
    enc.eval()
    x = torch.from_numpy(np.random.RandomState(seed).standard_normal((n, dims[0])).astype(np.float32))
    return g, enc, x


@pytest.mark.parametrize("world,n,dims,chunks,balanced,widening", [
    (2, 1001, [12, 16, 16, 5], 1, False, "narrow"), (4, 1003, [8, 24, 24, 6], 3, True, "narrow"), (8, 1001, [8, 24, 24, 6], 2, True, "narrow"),
    (4, 1003, [8, 24, 24, 6], 4, True, "wide"), (2, 90, [6, 16, 5], 4, False, "wide"), (4, 640, [12, 16, 16, 5], 2, False, "narrow"),
    (4, 1003, [8, 24, 24, 6], 2, True, "mixed"), (8, 1001, [8, 24, 24, 6], 4, False, "mixed"), (4, 1003, [8, 24, 24, 6], 2, True, "mixed-cut")])
def test_emulated_rank_equals_unsharded_rows(world, n, dims, chunks, balanced, widening):
    """dist.EmulatedPeers (bench.py --emulate N / --workload xl): ONE process plays rank r of an N-rank job, every collective is a
    local fill of the same bytes.  With the truth of an unsharded forward as the peers' data, every emulated rank's output equals
    the unsharded rows and the bytes "received" are exactly the real exchange's; without truth (peers' slots = copies of the own
    slab, the synthetic-XL mode) the same launches run over the same volume."""
    from glnn_amd import dist as gdist
    g, enc, x = _emu_setup(n, dims, 5)
    be = OracleBackend()
    with torch.no_grad():
        truth, want = gdist.record_truth(enc, g, x, be)
    bounds = gdist.RowShards.balanced_bounds(g.indptr, world) if balanced else None
    r4 = lambda d: (d + 3) // 4 * 4
    L = len(dims) - 1
    per_node = 0
    for l in range(L):
        d_in, d_out = dims[l], dims[l + 1]
        if d_in > d_out:
            per_node += r4(d_out)
        elif l < L - 1 and 2 * d_in <= d_out and widening in ("narrow", "mixed", "mixed-cut"):
            per_node += r4(d_in)
        elif l < L - 1 and not dims[l + 1] > dims[l + 2]:
            per_node += r4(d_out)
    covered = np.zeros(n, bool)
    for rank in range(world):
        sh = gdist.RowShards(n, world, rank, chunks=chunks, bounds=bounds)
        for use_truth in (True, False):
            peers = gdist.EmulatedPeers(world, rank, truth=truth if use_truth else None)
            gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
            with torch.no_grad():
                st = gdist.ShardedTeacher(enc, g.row_range(sh.lo, sh.hi), sh, be, group=peers, widening_exchange=widening.split("-")[0],
                                          mixed_quantum=8 if widening.endswith("cut") else 0)
                y = st.forward(x)
            if widening.startswith("mixed"):      # layer 1: "W" chunks move the wide output, "N" chunks the narrow aggregate; the other layers as in the narrow form
                m = st.sh
                assert set(m.kinds) == {"W", "N"} and m.rpr == sh.rpr
                l1 = sum(world * c * (r4(dims[1]) if k == "W" else r4(dims[0])) for c, k in zip(m.csize, m.kinds))
                assert gdist.EXCHANGE_STATS["floats_received"] == l1 + m.n_pad * (per_node - r4(dims[0]))
            else:
                assert gdist.EXCHANGE_STATS["floats_received"] == sh.n_pad * per_node
            assert tuple(y.shape) == (sh.rows, dims[-1]) and bool(torch.isfinite(y).all())
            if use_truth:
                np.testing.assert_allclose(y.numpy(), want[sh.lo:sh.hi].numpy(), atol=1e-4, rtol=0)
                covered[sh.lo:sh.hi] = True
    assert covered.all()


@pytest.mark.parametrize("world,dims,overlap", [(2, [12, 16, 16, 5], False), (4, [8, 24, 24, 6], True), (4, [16, 16, 16, 4], True),
                                                (8, [8, 24, 6], False)])
def test_emulated_rank_halo_equals_unsharded_rows(world, dims, overlap):
    """The halo form under dist.EmulatedPeers: the peers' requests are derived from the full graph (what their HaloPlan would send),
    the halo rows come from the truth of the unsharded forward -> equal rows, and the plan's send lists equal what a real peer asks
    for (checked against HaloPlans built for the peers themselves)."""
    from glnn_amd import dist as gdist
    n = 1600
    g, enc, x = _emu_setup(n, dims, 7, clustered=True)
    be = OracleBackend()
    with torch.no_grad():
        truth, want = gdist.record_truth(enc, g, x, be)
    bounds = gdist.RowShards.balanced_bounds(g.indptr, world)
    plans = []
    for rank in range(world):
        sh = gdist.RowShards(n, world, rank, bounds=bounds)
        peers = gdist.EmulatedPeers(world, rank, truth=truth, full_graph=g)
        t = gdist.HaloShardedTeacher(enc, g.row_range(sh.lo, sh.hi), sh, be, group=peers, overlap=overlap)
        with torch.no_grad():
            y = t.forward(x)
        np.testing.assert_allclose(y.numpy(), want[sh.lo:sh.hi].numpy(), atol=1e-4, rtol=0)
        plans.append((sh, t.plan))
    for sh, pl in plans:            # rows rank r sends to p == rows p receives from r
        for p, (shp, plp) in enumerate(plans):
            assert pl.send_counts[p] == plp.recv_counts[sh.rank]
        assert pl.n_send == sum(plp.recv_counts[sh.rank] for _, plp in plans)
