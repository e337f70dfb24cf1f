"""Synthetic stand-ins for the reference's datasets (reference dataloader.py:42-111 loads the real
ones through dgl/ogb; neither the datasets nor a network exist here -- SURVEY.md section 8d).

Shapes are the public statistics of the datasets BASELINE.json names; graphs are seeded random
multigraphs with a power-law degree profile in RANDOM node order (worst case for locality), stored as
CSR over destination rows (int64 indptr, int32 indices) -- the layout libglnn_hip.so consumes.
Everything is built with torch so that the big ones are generated directly in HBM."""
import math

import torch

from .graph import CSRGraph

SHAPES = {
    # name: nodes, feature dim, classes, (train, val, test)
    "cora": dict(n=2485, f=1433, c=7, split=(140, 210, 2135)),
    "ogbn-arxiv": dict(n=169343, f=128, c=40, split=(90941, 29799, 48603)),
    "ogbn-products": dict(n=2449029, f=100, c=47, split=(196615, 39323, 2213091)),
}


def _powerlaw_endpoints(n, m, alpha, offset, gen, device):
    """m node ids drawn with P(i) ~ (rank_i + offset)^-alpha, ranks randomly permuted over node ids."""
    # The cumulative weights are summed on the HOST: a device cumsum in float64 is a decoupled look-back scan whose grouping of the partial
    # sums follows the timing of its workgroups, so its last bits -- and with them a few of the 124 M endpoints -- changed from process to
    # process (scripts/train_determinism_probe.py: two variants of the "seeded" products graph, sampled batches differing by one node).
    w = (torch.arange(n, dtype=torch.float64) + offset).pow(-alpha)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    cdf = cdf.to(device)
    perm = torch.randperm(n, generator=gen, device=device)
    out = torch.empty(m, dtype=torch.int64, device=device)
    step = 1 << 24
    for s in range(0, m, step):
        u = torch.rand(min(step, m - s), generator=gen, device=device, dtype=torch.float64)
        out[s:s + step] = perm[torch.searchsorted(cdf, u).clamp_(max=n - 1)]
    return out


def csr_from_edges(src, dst, n_dst, n_src=None):
    """CSR over destinations; duplicates kept (multi-edges count multiply, like dgl)."""
    n_src = n_dst if n_src is None else n_src
    order = torch.argsort(dst, stable=True)
    indices = src[order].to(torch.int32)
    counts = torch.bincount(dst, minlength=n_dst)
    indptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=dst.device)
    torch.cumsum(counts, 0, out=indptr[1:])
    return CSRGraph(indptr, indices, n_dst, n_src)


def make_graph(name, seed=0, device="cpu", scale=1.0):
    """cora / ogbn-arxiv / ogbn-products shaped graph. `scale` < 1 shrinks node and edge counts (tests)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if name == "cora":
        # 5,069 undirected edges, symmetrised, + self-loops -> 12,623 nnz   (dataloader.py:103-105)
        n = max(8, int(2485 * scale))
        m = max(8, int(5069 * scale))
        a = torch.randint(0, n, (m,), generator=gen, device=device)
        b = torch.randint(0, n, (m,), generator=gen, device=device)
        loops = torch.arange(n, device=device)
        return csr_from_edges(torch.cat([a, b, loops]), torch.cat([b, a, loops]), n)
    if name == "ogbn-arxiv":
        # 1,166,243 directed edges, + reverse (no dedup) + self-loops -> 2,501,829 nnz (dataloader.py:74-77)
        n = max(8, int(169343 * scale))
        m = max(8, int(1166243 * scale))
        src = torch.randint(0, n, (m,), generator=gen, device=device)
        dst = _powerlaw_endpoints(n, m, 0.6, 10.0, gen, device)
        loops = torch.arange(n, device=device)
        return csr_from_edges(torch.cat([src, dst, loops]), torch.cat([dst, src, loops]), n)
    if name == "ogbn-products":
        # 61,859,140 undirected edges stored in both directions -> 123,718,280 nnz, no self-loops
        n = max(8, int(2449029 * scale))
        m = max(8, int(61859140 * scale))
        # alpha 0.5, offset 5: expected max degree ~17.7k at full scale (real graph: 17,481), min ~10
        ab = _powerlaw_endpoints(n, 2 * m, 0.5, 5.0, gen, device)     # ONE rank->node permutation for both ends
        a, b = ab[:m], ab[m:]
        return csr_from_edges(torch.cat([a, b]), torch.cat([b, a]), n)
    raise ValueError(f"Unknown dataset shape: {name}")


def make_clustered_graph(n, avg_deg, communities=64, p_in=0.9, seed=0, device="cpu", shuffle_ids=False):
    """A graph WITH locality (the prescribed products-shaped generator has none): node v belongs to community v // (n / communities)
    and an edge keeps its second endpoint inside the first endpoint's community with probability p_in (uniform elsewhere
    otherwise); both directions stored.  With shuffle_ids the node ids are randomly permuted afterwards (the "random node
    order" of the same graph).  Used to exercise the halo exchange of glnn_amd.dist.HaloShardedTeacher."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    m = int(n * avg_deg / 2)
    size = max(1, n // communities)
    a = torch.randint(0, n, (m,), generator=gen, device=device)
    inside = torch.rand(m, generator=gen, device=device) < p_in
    b_in = (a // size) * size + torch.randint(0, size, (m,), generator=gen, device=device)
    b_out = torch.randint(0, n, (m,), generator=gen, device=device)
    b = torch.where(inside, b_in, b_out).clamp_(max=n - 1)
    if shuffle_ids:
        perm = torch.randperm(n, generator=gen, device=device)
        a, b = perm[a], perm[b]
    return csr_from_edges(torch.cat([a, b]), torch.cat([b, a]), n)


def reorder_by_degree(g):
    """Renumber the nodes of a square graph by descending in-degree (stable): new id r <- old id perm[r].  Returns
    (graph with relabelled rows and columns, perm); x_new = x[perm].  One-time host-side style preparation (torch
    index ops on the graph's device), the locality-ordered variant SURVEY.md 8(d) allows beside the random node order."""
    if g.n_dst != g.n_src:
        raise ValueError("reorder_by_degree: square graphs only")
    deg = g.in_degrees()
    perm = torch.argsort(deg, descending=True, stable=True)
    rank = torch.empty_like(perm)
    rank[perm] = torch.arange(perm.numel(), device=perm.device)
    dst_old = torch.repeat_interleave(torch.arange(g.n_dst, device=g.device), deg)
    return csr_from_edges(rank[g.indices.long()], rank[dst_old], g.n_dst), perm


def relabel(g, perm):
    """The square graph g with its nodes renumbered: new id r <- old id perm[r] (rows and columns); x_new = x[perm]."""
    if g.n_dst != g.n_src:
        raise ValueError("relabel: square graphs only")
    rank = torch.empty_like(perm)
    rank[perm] = torch.arange(perm.numel(), device=perm.device)
    dst_old = torch.repeat_interleave(torch.arange(g.n_dst, device=g.device), g.in_degrees().long())
    return csr_from_edges(rank[g.indices.long()], rank[dst_old], g.n_dst)


def locality_order(g, iters=12, seed=0):
    """A node order with LOCALITY for node-range sharding (what a METIS-style partitioner is for; SURVEY.md 8e "halo all-to-all
    of only the referenced remote rows"): label propagation over the CSR -- every node repeatedly adopts the most frequent label
    among its in-neighbours (ties: the smallest label), half of the nodes per sweep (semi-synchronous: a synchronous sweep
    oscillates on near-bipartite neighbourhoods) -- then nodes are sorted by label (stable), so a community becomes a contiguous id
    range.  Index arithmetic only (sort / unique / scatter-reduce on the graph's device; one-time preparation, O(iters * nnz log
    nnz)); the aggregation arithmetic is untouched: the forward on the relabelled graph is the same sums in the same order.
    Returns perm (new id r <- old id perm[r]); use data.relabel(g, perm) and x[perm]."""
    n, dev = g.n_dst, g.device
    if g.n_dst != g.n_src:
        raise ValueError("locality_order: square graphs only")
    deg = g.in_degrees().long()
    dst = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    src = g.indices.long()
    labels = torch.arange(n, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    for it in range(iters):
        key = dst * n + labels[src]                              # (destination, neighbour label) pairs
        uk, cnt = torch.unique(key, return_counts=True)           # sorted by destination, then label
        ud, ul = uk // n, uk % n
        best = torch.zeros(n, dtype=cnt.dtype, device=dev).scatter_reduce_(0, ud, cnt, reduce="amax", include_self=True)
        is_best = cnt == best[ud]
        cand = torch.full((n,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, ud[is_best], ul[is_best], reduce="amin", include_self=True)
        cand = torch.where(cand < n, cand, labels)                # isolated nodes keep their label
        move = torch.rand(n, generator=gen, device=dev) < 0.5 if it + 1 < iters else torch.ones(n, dtype=torch.bool, device=dev)
        new = torch.where(move, cand, labels)
        changed = int((new != labels).sum())
        labels = new
        if changed == 0:
            break
    return torch.argsort(labels, stable=True)


def make_uniform_graph(n, avg_deg, seed=0, device="cpu"):
    """Uniform random directed multigraph (the synthetic-XL config's per-shard generator)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    m = int(n * avg_deg)
    src = torch.randint(0, n, (m,), generator=gen, device=device)
    dst = torch.randint(0, n, (m,), generator=gen, device=device)
    return csr_from_edges(src, dst, n)


def make_xl_shard(rows, avg_deg, n_total, seed=0, device="cpu"):
    """One rank's destination rows of the synthetic-XL graph (BASELINE configs[4]: 100 M nodes / 2 B edges over 8 ranks = 12.5 M
    rows and 250 M in-edges per rank): `rows` destinations, rows * avg_deg in-edges whose sources are uniform over ALL n_total
    nodes (global ids), generated on `device` from a seeded counter RNG -- the graph never crosses PCIe."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    nnz = int(rows * avg_deg)
    dst = torch.randint(0, rows, (nnz,), generator=gen, device=device)
    src = torch.randint(0, n_total, (nnz,), generator=gen, device=device)
    return csr_from_edges(src, dst, rows, n_total)


def make_node_data(name, seed=0, device="cpu", scale=1.0, n=None):
    """feats ~ N(0,1) fp32, labels uniform int64, teacher out_t = log_softmax(N(0,1)), index split."""
    sh = SHAPES[name]
    n = n if n is not None else max(8, int(sh["n"] * scale))
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 12345)
    feats = torch.randn((n, sh["f"]), generator=gen, device=device, dtype=torch.float32)
    labels = torch.randint(0, sh["c"], (n,), generator=gen, device=device, dtype=torch.int64)
    out_t = torch.log_softmax(torch.randn((n, sh["c"]), generator=gen, device=device, dtype=torch.float32), dim=1)
    perm = torch.randperm(n, generator=gen, device=device)
    tr, va, te = sh["split"]
    tot = tr + va + te
    n_tr, n_va = max(1, math.floor(n * tr / tot)), max(1, math.floor(n * va / tot))
    idx_train, idx_val, idx_test = perm[:n_tr], perm[n_tr:n_tr + n_va], perm[n_tr + n_va:]
    return feats, labels, out_t, (idx_train, idx_val, idx_test)
