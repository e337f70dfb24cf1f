"""What makes an allocation fast or slow for the gather (scripts/bimodal_probe.py found 18.1 ... 19.4 ms per allocation)?  One process:
(a) ONE 26 GB allocation carved into eight 2.5 GB slices 3.25 GB apart, (b) eight separate allocations made back to back, (c) eight separate
allocations with 64 MB spacers allocated in between (a fragmented pool), (d) again (b) after freeing everything (reuse of the pool)."""
import os, sys, statistics
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


def show(tag, bufs):
    ms = []
    for x in bufs:
        x.normal_().relu_()
        ms.append(timed(x))
    print(f"{tag:34s} " + " ".join(f"{t:6.2f}" for t in ms) + "   ptrs " + " ".join(f"{x.data_ptr() >> 20:#x}" for x in bufs), flush=True)


rows_gap = (3328 << 20) // 1024            # 3.25 GB in rows of 1 KB
big = torch.empty(8 * rows_gap, 256, device=dev)
show("(a) slices of ONE 26 GB allocation", [big[i * rows_gap: i * rows_gap + n] for i in range(8)])
del big
torch.cuda.empty_cache()
sep = [torch.empty(n, 256, device=dev) for _ in range(8)]
show("(b) 8 separate allocations", sep)
del sep
torch.cuda.empty_cache()
frag, spacers = [], []
for i in range(8):
    spacers.append(torch.empty(64 << 20, dtype=torch.uint8, device=dev))
    frag.append(torch.empty(n, 256, device=dev))
show("(c) separate, 64 MB spacers between", frag)
del frag, spacers
sep = [torch.empty(n, 256, device=dev) for _ in range(8)]
show("(d) separate again, pool reused", sep)
del sep
torch.cuda.empty_cache()
sep = [torch.empty(n, 256, device=dev) for _ in range(8)]
show("(e) separate again, pool emptied", sep)
