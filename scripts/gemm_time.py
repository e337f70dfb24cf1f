"""Median time of our NT / KN / TN GEMM at the MLP3w8 shape with whichever lib GLNN_LIB_PATH selects.
python scripts/gemm_time.py [tag]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("GLNN_LIB_PATH", "default"))
dev = "cuda:0"
m, k, n = 4096, 2048, 2048
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
out = ops.feat_empty(m, n, dev); ws = torch.empty(1 << 24, device=dev)
dz = torch.randn(m, n, device=dev); outw = torch.empty(n, k, device=dev)
wst = torch.empty(64 * n + 2 * n * k + (1 << 22), device=dev)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return statistics.median(ts)
fl = 2.0 * m * k * n
res = []
for name, fn in (("NT", lambda: ops.gemm(x, w, out=out, workspace=ws)),
                 ("KN", lambda: ops.gemm(dz, w, w_is_kn=True, out=out, workspace=ws)),
                 ("TN", lambda: ops.gemm_tn(dz, x, out=outw, workspace=wst)),
                 ("torchNT", lambda: torch.matmul(x, w.t()))):
    t = timeit(fn)
    res.append(f"{name} {t * 1e3:6.1f}us {fl / t / 1e9:6.1f}TF")
print(f"{tag:24s} " + " | ".join(res), flush=True)
