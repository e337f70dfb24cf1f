import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
dev = "cuda:0"
os.environ["GLNN_SPMM_GPR_MIN_ROWS"] = "0"
torch.manual_seed(0)
for n, d, maxdeg in [(64, 64, 6), (64, 64, 40), (1000, 47, 30), (5000, 128, 60), (5000, 256, 200)]:
    deg = torch.randint(0, maxdeg + 1, (n,))
    indptr = torch.zeros(n + 1, dtype=torch.int64); indptr[1:] = deg.cumsum(0)
    nnz = int(indptr[-1])
    indices = torch.randint(0, n, (nnz,), dtype=torch.int32)
    x = torch.randn(n, d)
    ip, ix, xd = indptr.to(dev), indices.to(dev), ops.as_feat(x.to(dev))
    os.environ["GLNN_SPMM_GPR"] = "0"; a = ops.spmm(ip, ix, xd, n, ops.AGG_SAGE_GCN).clone()
    os.environ["GLNN_SPMM_GPR"] = "1"; b = ops.spmm(ip, ix, xd, n, ops.AGG_SAGE_GCN).clone()
    err = (a - b).abs().max(1).values.cpu()
    bad = (err > 1e-5).nonzero().flatten()
    print(f"n={n} d={d} maxdeg={maxdeg}: bad rows {bad.numel()} / {n}; first {bad[:20].tolist()}; their deg {deg[bad[:20]].tolist()}", flush=True)
