"""Back-to-back (no idle gaps, clocks stay up) time of every GEMM form the MLP3w8 student step launches, and of the whole step:
python scripts/gemm_sustained.py [tag]      (GLNN_LIB_PATH selects a variant library)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("GLNN_LIB_PATH", "default"))
dev = "cuda:0"
m, k, n = 4096, 2048, 2048
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
sc = torch.rand(k, device=dev) + 0.5; sh = torch.randn(k, device=dev) * 0.1
out = ops.feat_empty(m, n, dev); ws = torch.empty(1 << 24, device=dev)
dz = torch.randn(m, n, device=dev); outw = torch.empty(n, k, device=dev)
wst = torch.empty(64 * n + 2 * n * k + (1 << 22), device=dev)
def sustained(fn, secs=1.0):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(100): fn()
        torch.cuda.synchronize(); it += 100
    return (time.perf_counter() - t0) / it
fl = 2.0 * m * k * n
forms = (("NT", lambda: ops.gemm(x, w, out=out, workspace=ws)),
         ("NT+bn", lambda: ops.gemm(x, w, a_scale=sc, a_shift=sh, out=out, workspace=ws)),
         ("NT+bn+drop", lambda: ops.gemm(x, w, a_scale=sc, a_shift=sh, drop_p=0.2, drop_seed=7, out=out, workspace=ws)),
         ("KN", lambda: ops.gemm(dz, w, w_is_kn=True, out=out, workspace=ws)),
         ("TN", lambda: ops.gemm_tn(dz, x, out=outw, workspace=wst)),
         ("TN+bn+drop", lambda: ops.gemm_tn(dz, x, b_scale=sc, b_shift=sh, drop_p=0.2, drop_seed=7, out=outw, workspace=wst)),
         ("torchNT", lambda: torch.matmul(x, w.t())),
         ("torchNN", lambda: torch.matmul(dz, w)),
         ("torchTN", lambda: torch.matmul(dz.t(), x)))
res = []
for name, fn in forms:
    t = sustained(fn)
    res.append(f"{name} {t * 1e6:6.1f}us {fl / t / 1e12:6.1f}TF")
print(f"{tag:20s} " + " | ".join(res), flush=True)
