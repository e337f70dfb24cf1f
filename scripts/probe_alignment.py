"""Does the relative placement of the gathered matrix and the output matrix matter?  (bimodal 19.5 / 22.3 ms seen
across processes for the same kernel).  Carve x and out from one pool at controlled byte offsets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
dev = "cuda:0"
g = data.make_graph("ogbn-products", seed=0, device=dev)
n, d = g.n_dst, 256
def t(fn, iters=4):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
pool = torch.empty(3 * n * d + (1 << 26), device=dev)     # floats
base = pool.data_ptr()
w = torch.randn(d, d, device=dev) / 16
wp = ops.pack_weight(w)
print("pool base % 2MiB =", base % (2 << 20), "indices ptr % 2MiB =", g.indices.data_ptr() % (2 << 20))
x = pool[: n * d].view(n, d); x.normal_()
for off_bytes in [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 20, (8 << 20) + 512, 1 << 25]:
    start = n * d + off_bytes // 4
    out = pool[start: start + n * d].view(n, d)
    f = t(lambda: ops.sage_fused(g.indptr, g.indices, x, n, w, out=out, w_packed=wp))
    s_ = t(lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=out))
    print(f"out offset +{off_bytes:>9d} B: fused {f:6.2f} ms  spmm {s_:6.2f} ms", flush=True)
