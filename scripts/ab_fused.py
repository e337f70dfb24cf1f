"""Interleaved A/B on one box: fused K1F layer vs aggregation + GEMM, products shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops

dev = "cuda:0"
g = data.make_graph("ogbn-products", seed=0, device=dev)
n = g.n_dst


def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def agg47():
    x = torch.randn(n, 47, device=dev); x = ops.as_feat(x); out = ops.feat_empty(n, 47, dev)
    return t(lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=out))


print("spmm d=47:", " ".join(f"{agg47():.2f}" for _ in range(3)), flush=True)
for d_in, d_out in [(100, 256), (256, 256)]:
    x = torch.randn(n, d_in, device=dev)
    w = torch.randn(d_out, d_in, device=dev) / d_in ** 0.5
    sc = torch.rand(d_out, device=dev) + 0.5; sh = torch.randn(d_out, device=dev)
    wp = ops.pack_weight(w)
    out = ops.feat_empty(n, d_out, dev); agg = ops.feat_empty(n, d_in, dev)
    def fused():
        ops.sage_fused(g.indptr, g.indices, x, n, w, ep_scale=sc, ep_shift=sh, relu=True, out=out, w_packed=wp)
    def unfused():
        ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=agg)
        ops.gemm(agg, w, ep_scale=sc, ep_shift=sh, relu=True, out=out)
    def spmm_only():
        ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=agg)
    res = []
    for rep in range(3):
        res.append((t(fused), t(unfused), t(spmm_only)))
    print(d_in, d_out, " | ".join(f"fused {a:6.2f} unfused {b:6.2f} (spmm {c:6.2f})" for a, b, c in res), flush=True)
