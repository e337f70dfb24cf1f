"""In some processes EVERY fresh allocation is slow for the gather (19.4 ms; scripts/ab_placement.sh rounds 2-3).  Does pushing the candidates into
another region of HBM help?  Groups of 5 candidates separated by 50 GB ballast allocations that stay alive."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
dev = "cuda:0"
g = data.make_graph("ogbn-products", seed=0, device=dev, scale=1.0)
n = g.n_dst
w2 = torch.randn(256, 256, device=dev) / 16
w3 = torch.randn(47, 256, device=dev) / 16
order = g.fused_tile_order()
o47 = ops.feat_empty(n, 47, dev)


def timed(x, reps=3):
    f = lambda: ops.sage_fused(g.indptr, g.indices, x, n, w2, relu=True, x_self=x, w_next=w3, out_next=o47, want_out=False, tile_order=order)
    f()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


keep, ballast = [], []
for grp in range(4):
    bufs = [torch.zeros(n, 256, device=dev) for _ in range(5)]
    print(f"group {grp} (ballast below: {50 * grp} GB) " + " ".join(f"{timed(x):6.2f}" for x in bufs), flush=True)
    keep.append(bufs)
    ballast.append(torch.empty(50 << 30, dtype=torch.uint8, device=dev))
