"""Where does the aggregation kernel lose time on uniform random graphs?  t(rows, degree, D) for the stand-alone kernel:
fit  t = a * rows + b * edges  per width (a = per-row overhead: indptr -> indices -> gather chain, self row, store;
b = per-edge cost).  usage: python scripts/probe_gather.py [rows_millions] [src_millions]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops  # noqa: E402

dev = "cuda:0"
rows = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 4_000_000
nsrc = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 32_000_000


def timeit(fn, warmup=2, iters=7):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2]


gen = torch.Generator(device=dev); gen.manual_seed(3)
for d in (64, 128, 256):
    x = torch.empty(nsrc, d, device=dev)
    for s0 in range(0, nsrc, 1 << 23):
        x[s0:s0 + (1 << 23)].normal_(generator=gen)
    out = ops.feat_empty(rows, d, dev)
    res = []
    for deg in (5, 10, 20, 50, 100):
        nnz = rows * deg
        indptr = torch.arange(0, nnz + 1, deg, dtype=torch.int64, device=dev)
        indices = torch.randint(0, nsrc, (nnz,), generator=gen, device=dev, dtype=torch.int32)
        t = timeit(lambda: ops.spmm(indptr, indices, x, rows, ops.AGG_SAGE_GCN, out=out, x_self=x[:rows]))
        b = nnz * (4 * d + 4) + rows * (8 * d + 8)
        res.append((deg, t))
        print(f"D={d:4d} deg={deg:4d} rows={rows} : {t:8.3f} ms  {b / t / 1e6:7.1f} GB/s alg ({b / t / 8e7:5.1f} % of 8 TB/s)  "
              f"{t * 1e6 / rows:7.2f} ns/row {t * 1e6 / nnz:6.3f} ns/edge", flush=True)
        del indptr, indices
    # two-point fit on deg 10 and 100
    (d1, t1), (d2, t2) = res[1], res[4]
    bb = (t2 - t1) / ((d2 - d1) * rows)
    aa = t1 / rows - bb * d1
    print(f"   fit D={d}: per-row {aa * 1e6:.2f} ns, per-edge {bb * 1e6:.3f} ns  -> edge-only rate {(4 * d + 4) / bb / 1e6 / 1e3:.2f} TB/s", flush=True)
    del x, out
    torch.cuda.empty_cache()
