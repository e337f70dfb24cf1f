"""The dominant launch (fused D=256 gather over the products graph) runs at 18.0 OR 19.4 ms from one bench process to the next, whatever the
library build (scripts/ab_multi.sh, round 5).  What decides it?  One process: the same launch with its 2.5 GB source matrix placed at different
offsets inside one big allocation, with a fresh allocation per trial, and with the index array re-allocated; prints ms and the addresses."""
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
src = torch.randn(n, 256, device=dev).relu_()
o47 = ops.feat_empty(n, 47, dev)


def timed(x, reps=5):
    f = lambda: ops.sage_fused(g.indptr, g.indices, x, n, w2, relu=True, x_self=x, w_next=w3, out_next=o47, want_out=False, tile_order=order)
    f(); f()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts), min(ts), max(ts)


print(f"indices ptr {g.indices.data_ptr():#x} indptr {g.indptr.data_ptr():#x}", flush=True)
big = torch.empty(n * 256 + (64 << 20), device=dev)            # + 256 MB of slack
for off_bytes in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 32 << 20, 64 << 20, 100 << 20, 128 << 20, 200 << 20, 255 << 20):
    x = big[off_bytes // 4: off_bytes // 4 + n * 256].view(n, 256)
    x.copy_(src)
    med, lo, hi = timed(x)
    print(f"offset {off_bytes:>10d} ptr {x.data_ptr():#x}  {med:7.3f} ms (min {lo:.3f} max {hi:.3f})", flush=True)
del big
torch.cuda.empty_cache()
for trial in range(6):                                           # fresh allocations of different sizes in between: different addresses
    pad = torch.empty((trial * 37 + 1) << 20, device=dev)
    x = src.clone()
    med, lo, hi = timed(x)
    print(f"fresh alloc {trial} ptr {x.data_ptr():#x}  {med:7.3f} ms (min {lo:.3f} max {hi:.3f})", flush=True)
    del x
for trial in range(3):                                           # the same matrix, the index array moved
    ix = g.indices.clone()
    pad2 = torch.empty((trial * 53 + 7) << 20, device=dev)
    f = lambda: ops.sage_fused(g.indptr, ix, src, n, w2, relu=True, x_self=src, w_next=w3, out_next=o47, want_out=False, tile_order=order)
    f(); f()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    print(f"indices moved {trial} ptr {ix.data_ptr():#x} x ptr {src.data_ptr():#x}  {statistics.median(ts):7.3f} ms", flush=True)
# several 2.5 GB source matrices alive at once (distinct physical memory), each timed; then the OUTPUT / index buffers likewise
held = [src.clone() for _ in range(10)]
for i, x in enumerate(held):
    med, lo, hi = timed(x)
    print(f"simultaneous alloc {i} ptr {x.data_ptr():#x}  {med:7.3f} ms (min {lo:.3f} max {hi:.3f})", flush=True)
