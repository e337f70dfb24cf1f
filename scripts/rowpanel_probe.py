"""K3r (csrc/gemm_rowpanel.hip) on its three shapes, timed with HIP events against the tiled kernels (GLNN_GEMM_ROWPANEL=0); also the
target of the SQ counter pass in scripts/rowpanel_pmc.sh."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import _lib, ops
dev = "cuda:0"
shapes = [("products replicated projection", 2449029, 100, 256), ("teacher-training layer 0", 500000, 100, 256), ("student first layer", 4096, 100, 2048),
          ("xl chunk (1/10)", 2500000, 128, 256)]
for what, m, k, n in shapes:
    a = ops.as_feat(torch.randn(m, k, device=dev))
    w = torch.randn(n, k, device=dev) / k ** 0.5
    es, eh = torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev)
    out = ops.feat_empty(m, n, dev)
    res = {}
    for mode in ("1", "0"):
        os.environ["GLNN_GEMM_ROWPANEL"] = mode
        _lib.lib().glnn_reload_options()
        for _ in range(3):
            ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=True, out=out)
        ts = []
        for _ in range(15):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=True, out=out); e.record(); e.synchronize()
            ts.append(s.elapsed_time(e))
        res[mode] = statistics.median(ts)
    fl = 2.0 * m * k * n
    print(f"{what:34s} m={m:8d} k={k:3d} n={n:4d}  rowpanel {res['1'] * 1e3:9.1f} us = {fl / res['1'] / 1e9:6.1f} TF   "
          f"tiled {res['0'] * 1e3:9.1f} us = {fl / res['0'] / 1e9:6.1f} TF", flush=True)
os.environ["GLNN_GEMM_ROWPANEL"] = "1"
_lib.lib().glnn_reload_options()
