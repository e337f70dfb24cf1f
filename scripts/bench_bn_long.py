"""BatchNorm / activation passes on the LONG activations of a sampled teacher batch ([0.5 M x 256] = 512 MB per array, products
config): time and effective bandwidth per pass (development aid).  usage: python scripts/bench_bn_long.py [rows] [h] [p]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2]


z = ops.as_feat(torch.randn(rows, h, device=dev))
da = ops.as_feat(torch.randn(rows, h, device=dev))
gamma, beta = torch.rand(h, device=dev) + 0.5, torch.zeros(h, device=dev)
rm, rv, nbt = torch.zeros(h, device=dev), torch.ones(h, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
mb = rows * h * 4 / 1e6
t = timeit(lambda: ops.bn_stats(z, gamma, beta, rm, rv, nbt))
print(f"bn_stats            {t * 1e3:8.1f} us  {1 * mb / t / 1e3:6.2f} TB/s (1 array read)")
mean, rstd, sc, sh = ops.bn_stats(z, gamma, beta, rm, rv, nbt)
y = ops.feat_empty(rows, h, dev)
t = timeit(lambda: ops.act_fwd(z, sc, sh, drop_p=p, drop_seed=1, out=y))
print(f"act_fwd             {t * 1e3:8.1f} us  {2 * mb / t / 1e3:6.2f} TB/s (1 read + 1 write)")
dz = ops.feat_empty(rows, h, dev)
dzsum = torch.empty(h, device=dev)
t = timeit(lambda: ops.bn_relu_bwd(da, z, gamma, mean, rstd, sc, sh, dz=dz, drop_p=p, drop_seed=1, dz_col_sum=dzsum))
print(f"bn_relu_bwd (BN)    {t * 1e3:8.1f} us  {5 * mb / t / 1e3:6.2f} TB/s (4 reads + 1 write over two launches + folds)")
t = timeit(lambda: ops.bn_relu_bwd(da, z, dz=dz, drop_p=p, drop_seed=1, dz_col_sum=dzsum))
print(f"relu/dropout bwd    {t * 1e3:8.1f} us  {3 * mb / t / 1e3:6.2f} TB/s (2 reads + 1 write)")
t = timeit(lambda: dz.copy_(da))
print(f"torch copy          {t * 1e3:8.1f} us  {2 * mb / t / 1e3:6.2f} TB/s")
