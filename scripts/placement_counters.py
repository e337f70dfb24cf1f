"""VERDICT r05 item 3, the counter evidence: ONE process allocates eight 2.5 GB source matrices, times the fused D=256 launch on each, then
launches it three times on the FASTEST and three times on the SLOWEST buffer (the last six sage_fused dispatches of the process, in that
order).  Run under `rocprofv3 --kernel-trace --pmc <group>` (scripts/placement_counters.sh) the counter CSV then holds the same kernel over
the same graph with the two kinds of backing."""
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
src = torch.randn(n, 256, device=dev).relu_()
f = lambda x: ops.sage_fused(g.indptr, g.indices, x, n, w2, relu=True, x_self=x, w_next=w3, out_next=o47, want_out=False, tile_order=order)


def timed(x):
    f(x)
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(x); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


spacers, bufs = [], []
for i in range(8):
    spacers.append(torch.empty(64 << 20, dtype=torch.uint8, device=dev))      # (a fragmented pool makes slow backings likely: r05_placement_cause.txt)
    b = torch.empty(n, 256, device=dev)
    b.copy_(src)
    bufs.append(b)
ms = [timed(b) for b in bufs]
fast, slow = min(range(8), key=lambda i: ms[i]), max(range(8), key=lambda i: ms[i])
print("ms per buffer:", " ".join(f"{t:.2f}" for t in ms), f" fast = #{fast} ({ms[fast]:.2f})  slow = #{slow} ({ms[slow]:.2f})", flush=True)
torch.cuda.synchronize()
for b in (bufs[fast], bufs[slow]):
    for _ in range(3):
        f(b)
    torch.cuda.synchronize()
print("MARK: the last six sage_fused dispatches = 3 x fast, then 3 x slow", flush=True)
