"""Sustained time of a list of GEMM shapes, pipelined kernels on / off (GLNN_GEMM_PIPE): python scripts/gemm_shapes.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
dev = "cuda:0"
def sustained(fn, secs=0.5):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); it += 50
    return (time.perf_counter() - t0) / it
SHAPES = [("tn", 4096, 2048, 100), ("tn", 4096, 2048, 2048), ("tn", 60000, 256, 128), ("tn", 500000, 256, 100), ("tn", 18000, 256, 256),
          ("nt", 60000, 256, 256), ("kn", 60000, 256, 256), ("nt", 4096, 128, 2048), ("tn", 512, 256, 256), ("tn", 1024, 1024, 1024)]
for form, m, k, n in SHAPES:
    if form == "tn":
        a = ops.as_feat(torch.randn(m, k, device=dev)); b = ops.as_feat(torch.randn(m, n, device=dev))
        out = torch.empty(k, n, device=dev); ws = torch.empty(64 * k + 2 * k * n + (1 << 24), device=dev)
        fn = lambda: ops.gemm_tn(a, b, out=out, workspace=ws)
    else:
        a = ops.as_feat(torch.randn(m, k, device=dev))
        w = ops.as_feat(torch.randn(k, n, device=dev)) if form == "kn" else torch.randn(n, k, device=dev)
        out = ops.feat_empty(m, n, dev); ws = torch.empty(1 << 24, device=dev)
        fn = lambda: ops.gemm(a, w, w_is_kn=(form == "kn"), out=out, workspace=ws)
    res = []
    for mode in ("1", "0"):
        os.environ["GLNN_GEMM_PIPE"] = mode
        res.append(sustained(fn))
    fl = 2.0 * m * k * n
    print(f"{form} m={m:7d} k={k:5d} n={n:5d}: pipe {res[0] * 1e6:8.1f} us ({fl / res[0] / 1e12:6.1f} TF) | off {res[1] * 1e6:8.1f} us ({fl / res[1] / 1e12:6.1f} TF)", flush=True)
