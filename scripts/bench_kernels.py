"""Kernel micro-benchmarks on one MI355X (development aid; bench.py is the contract benchmark).
usage: python scripts/bench_kernels.py [--scale 1.0] [--what spmm,gemm]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops  # noqa: E402


def timeit(fn, warmup=3, iters=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--what", default="spmm,gemm")
    ap.add_argument("--graph", default="ogbn-products")
    a = ap.parse_args()
    dev = "cuda:0"
    what = a.what.split(",")
    if "spmm" in what:
        t0 = time.time()
        g = data.make_graph(a.graph, seed=0, device=dev, scale=a.scale)
        torch.cuda.synchronize()
        n, nnz = g.n_dst, g.num_edges()
        deg = g.in_degrees()
        print(f"graph {a.graph} scale {a.scale}: n={n} nnz={nnz} max_deg={int(deg.max())} min_deg={int(deg.min())} "
              f"rows>512: {int((deg > 512).sum())} gen {time.time() - t0:.1f}s", flush=True)
        for d, ld in [(100, 100), (100, 128), (128, 128), (256, 256), (47, 48), (47, 64), (64, 64)]:
            buf = torch.randn((n, ld), device=dev)
            x = buf[:, :d]
            out = ops.feat_empty(n, d, dev) if ld == ops.round4(d) else torch.empty((n, ld), device=dev)[:, :d]
            med, best = timeit(lambda: ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN, out=out))
            balg = nnz * (4 * d + 4) + n * (8 * d + 8)
            print(f"spmm sage_gcn d={d:4d} ld={ld:4d}: {med:8.3f} ms (best {best:.3f})  {nnz / med / 1e6:7.2f} Gedges/s  "
                  f"alg {balg / med / 1e6:7.1f} GB/s = {balg / med / 1e6 / 8000 * 100:5.1f}% of 8 TB/s", flush=True)
            del buf, x, out
    if "gemm" in what:
        for m, k, n in [(2449029, 100, 256), (2449029, 256, 256), (2449029, 256, 47), (4096, 100, 2048), (4096, 2048, 2048),
                        (4096, 2048, 47), (4096, 4096, 4096), (512, 128, 256), (512, 256, 256)]:
            m = int(m * a.scale) if m > 100000 else m
            x = torch.randn((m, k), device=dev)
            w = torch.randn((n, k), device=dev) / k ** 0.5
            out = ops.feat_empty(m, n, dev)
            wsg = torch.empty(1 << 24, device=dev)
            med, best = timeit(lambda: ops.gemm(x, w, out=out, workspace=wsg))
            fl = 2.0 * m * n * k
            medt, _ = timeit(lambda: torch.matmul(x, w.t()))
            print(f"gemm m={m} k={k} n={n}: {med:8.3f} ms  {fl / med / 1e9:8.1f} TF/s (torch/rocBLAS {medt:8.3f} ms {fl / medt / 1e9:8.1f} TF/s)  "
                  f"bytes {(m * k + m * n) * 4 / med / 1e6:7.1f} GB/s", flush=True)
        for m, ka, nb in [(4096, 2048, 2048), (4096, 2048, 100), (4096, 47, 2048)]:
            dz = torch.randn((m, ka), device=dev)
            act = torch.randn((m, nb), device=dev)
            outw = torch.empty((ka, nb), device=dev)
            ws = torch.empty(64 * ka + max(64 * ka * nb, 2 * ka * nb + (1 << 22)), device=dev)
            med, _ = timeit(lambda: ops.gemm_tn(dz, act, out=outw, workspace=ws))
            medt, _ = timeit(lambda: torch.matmul(dz.t(), act))
            fl = 2.0 * m * ka * nb
            print(f"gemm_tn m={m} ka={ka} nb={nb}: {med:8.3f} ms {fl / med / 1e9:8.1f} TF/s (torch {medt:8.3f} ms {fl / medt / 1e9:8.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
