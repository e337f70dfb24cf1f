"""Our GEMMs and torch.matmul (rocBLAS/hipBLASLt) side by side on the student shapes, for rocprofv3 passes:
python scripts/gemm_ab.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
m, k, n = 4096, 2048, 2048
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
out = ops.feat_empty(m, n, dev); ws = torch.empty(1 << 24, device=dev)
dz = torch.randn(m, n, device=dev); outw = torch.empty(n, k, device=dev)
wst = torch.empty(64 * n + 2 * n * k + (1 << 22), device=dev)
for _ in range(it):
    ops.gemm(x, w, out=out, workspace=ws)               # NT
    torch.matmul(x, w.t())
    ops.gemm(dz, w, w_is_kn=True, out=out[:, :k] if k <= n else None, workspace=ws)   # KN (dgrad)
    torch.matmul(dz, w)
    ops.gemm_tn(dz, x, out=outw, workspace=wst)         # TN (wgrad)
    torch.matmul(dz.t(), x)
torch.cuda.synchronize()
