"""GEMM-only probe for PMC passes: python scripts/gemm_probe.py M K N [tn]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
m, k, n = (int(a) for a in sys.argv[1:4])
tn = len(sys.argv) > 4
dev = "cuda:0"
if tn:
    a = torch.randn(m, k, device=dev); b = torch.randn(m, n, device=dev)
    ws = torch.empty(64 * k + 4 * k * n + (1 << 22), device=dev)
    out = torch.empty(k, n, device=dev)
    for _ in range(5):
        ops.gemm_tn(a, b, out=out, workspace=ws)
else:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev)
    out = ops.feat_empty(m, n, dev)
    for _ in range(5):
        ops.gemm(x, w, out=out)
torch.cuda.synchronize()
