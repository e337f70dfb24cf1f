"""Would cache-blocking the gather by SOURCE stripes pay?  (propagation blocking: aggregate only the edges whose source lies in
one 1/S slice of the node range, so the gathered rows of a pass fit the 256 MB Infinity Cache; S passes accumulate.)
Probe with the existing kernel: the sub-CSR of one stripe vs 1/S of the full launch, products shape, D = 256 / 100."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.graph import CSRGraph

dev = "cuda:0"
g = data.make_graph("ogbn-products", seed=0, device=dev)
n, nnz = g.n_dst, g.num_edges()
dst = torch.repeat_interleave(torch.arange(n, device=dev), g.in_degrees())
src = g.indices.long()


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for d in (256, 100):
    x = torch.randn(n, d, device=dev)
    out = ops.feat_empty(n, d, dev)
    full = timeit(lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SUM, out=out))
    print(f"D={d}: full launch {full:.2f} ms ({nnz / full / 1e6:.2f} G edges/s)")
    for S in (4, 8, 16, 32):
        lo, hi = (S // 2) * (n // S), (S // 2 + 1) * (n // S)
        keep = (src >= lo) & (src < hi)
        sub = data.csr_from_edges(src[keep], dst[keep], n)
        t = timeit(lambda: ops.spmm(sub.indptr, sub.indices, x, n, ops.AGG_SUM, out=out))
        e = int(keep.sum())
        print(f"   S={S:2d}: stripe of {4e-6 * (hi - lo) * d:6.0f} MB, {e / 1e6:5.1f} M edges: {t:.2f} ms ({e / t / 1e6:.2f} G edges/s) -> {S} passes ~ {S * t:.1f} ms "
              f"(incl. {S}x the per-row read/write of the output)")
