"""Rate of the BatchNorm statistics / backward kernels with and without the dropout hash (development aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
dev = "cuda:0"
def sustained(fn, secs=0.3):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); it += 50
    return (time.perf_counter() - t0) / it
for rows, h in ((4096, 2048), (500000, 256)):
    z = ops.feat_empty(rows, h, dev); z.normal_()
    da = ops.feat_empty(rows, h, dev); da.normal_()
    dz = ops.feat_empty(rows, h, dev)
    g = torch.rand(h, device=dev) + 0.5; mean = z[:, :h].mean(0); rstd = 1 / z[:, :h].std(0)
    sc = g * rstd; sh = -mean * sc
    dg = torch.empty(h, device=dev); db = torch.empty(h, device=dev); cs = torch.empty(h, device=dev)
    ws = torch.empty((3 * ((rows + 127) // 128) + 2) * h + (1 << 20), device=dev)
    mb = 5 * rows * h * 4 / 1e6          # partial: da, z; apply: da, z, dz
    for name, p in (("no dropout", 0.0), ("dropout 0.2", 0.2)):
        t = sustained(lambda: ops.bn_relu_bwd(da, z, g, mean, rstd, sc, sh, dz=dz, dgamma=dg, dbeta=db, workspace=ws, drop_p=p, drop_seed=5, dz_col_sum=cs))
        print(f"[{rows}x{h}] bn_relu_bwd {name:12s} {t * 1e6:8.1f} us  {mb / t / 1e6:6.2f} TB/s (all launches of the call)", flush=True)
