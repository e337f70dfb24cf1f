"""Interleaved A/B of the wave-per-row and the group-per-row aggregation kernels inside ONE process (GLNN_SPMM_GPR is read per
call): products-shaped graph at D = 47 / 100 / 128 / 256, then the XL-style uniform shard (D = 128, degree 20).
Checks the two kernels against each other (<= 2e-5) first.   usage: python scripts/ab_spmm.py [scale] [xl_rows_millions]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops  # noqa: E402

dev = "cuda:0"
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
xl_rows = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 12_500_000


def run(variant, fn):
    os.environ["GLNN_SPMM_GPR"] = variant
    return fn()


def timed(fn, iters=5):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2]


def ab(name, fn, bytes_alg, rounds=3):
    a = run("0", fn).clone()
    b = run("1", fn).clone()
    err = float((a - b).abs().max() / a.abs().max().clamp(min=1))
    res = {"0": [], "1": []}
    for _ in range(rounds):
        for v in ("0", "1"):
            os.environ["GLNN_SPMM_GPR"] = v
            fn(); fn()
            res[v].append(timed(fn))
    t0, t1 = min(res["0"]), min(res["1"])
    print(f"{name:28s} wave/row {t0:8.3f} ms ({bytes_alg / t0 / 8e7:5.1f} %)   group/row {t1:8.3f} ms ({bytes_alg / t1 / 8e7:5.1f} %)   "
          f"x{t0 / t1:5.3f}   max|diff| {err:.2e}", flush=True)
    assert err <= 2e-5, err


g = data.make_graph("ogbn-products", seed=0, device=dev, scale=scale)
n, nnz = g.n_dst, g.num_edges()
print(f"products-shaped: n={n} nnz={nnz}", flush=True)
for d in (47, 64, 100, 128, 256):
    x = ops.as_feat(torch.randn(n, d, device=dev))
    out = ops.feat_empty(n, d, dev)
    ab(f"products D={d}", lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=out), nnz * (4 * d + 4) + n * (8 * d + 8))
    if d == 47:
        bias = torch.randn(d, device=dev)
        ab(f"products D={d} +bias", lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=out, ep_shift=bias), nnz * (4 * d + 4) + n * (8 * d + 8))
        ab(f"products D={d} SUM", lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SUM, out=out), nnz * (4 * d + 4) + n * (8 * d + 8))
    del x, out
del g
torch.cuda.empty_cache()
if xl_rows > 0:
    rows, deg, d = xl_rows, 20, 128
    n_total = rows * 8 if rows >= 12_000_000 else rows * 4
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    dst = torch.randint(0, rows, (rows * deg,), generator=gen, device=dev)
    src = torch.randint(0, n_total, (rows * deg,), generator=gen, device=dev)
    order = torch.argsort(dst)
    indices = src[order].to(torch.int32)
    indptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(dst, minlength=rows), 0, out=indptr[1:])
    del dst, src, order
    x = torch.empty(n_total, d, device=dev)
    for s0 in range(0, n_total, 1 << 23):
        x[s0:s0 + (1 << 23)].normal_(generator=gen)
    out = ops.feat_empty(rows, d, dev)
    lo = 3 * rows
    ab(f"XL shard D=128 rows={rows}", lambda: ops.spmm(indptr, indices, x, rows, ops.AGG_SAGE_GCN, out=out, x_self=x[lo:lo + rows]),
       rows * deg * (4 * d + 4) + rows * (8 * d + 8))
