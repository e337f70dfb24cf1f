"""Small seeded CSR graphs for the tests (numpy; CSR over destination rows, int64 indptr, int32 indices)."""
import numpy as np


def csr_from_edges(src, dst, n_dst):
    order = np.argsort(dst, kind="stable")
    indices = np.asarray(src)[order].astype(np.int32)
    counts = np.bincount(dst, minlength=n_dst)
    indptr = np.zeros(n_dst + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, indices


def random_graph(n, avg_deg, seed, power=0.0, self_loops=False, symmetric=False, isolated=0, hub=0):
    """Random multigraph (duplicates kept on purpose: ogbn-arxiv keeps multi-edges, reference
    dataloader.py:75-76).  `isolated` rows get zero in-degree, `hub` adds one very-high-degree row."""
    rs = np.random.RandomState(seed)
    m = int(n * avg_deg)
    if power > 0:
        w = (np.arange(n) + 3.0) ** (-power)
        w = rs.permutation(w / w.sum())
        dst = rs.choice(n, size=m, p=w)
    else:
        dst = rs.randint(0, n, size=m)
    src = rs.randint(0, n, size=m)
    if hub:
        h = rs.randint(0, n)
        src = np.concatenate([src, rs.randint(0, n, size=hub)])
        dst = np.concatenate([dst, np.full(hub, h)])
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    if isolated:
        iso = rs.choice(n, size=isolated, replace=False)
        keep = ~np.isin(dst, iso)
        src, dst = src[keep], dst[keep]
    if self_loops:
        src = np.concatenate([src, np.arange(n)])
        dst = np.concatenate([dst, np.arange(n)])
    return csr_from_edges(src, dst, n)
