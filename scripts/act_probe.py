"""Rate of glnn_act_fwd_f32 (plain ReLU / BN+ReLU / +dropout) next to torch.relu and a copy (development aid)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
    out = ops.feat_empty(rows, h, dev)
    sc = torch.rand(h, device=dev) + 0.5; sh = torch.randn(h, device=dev) * 0.1
    mb = 2 * rows * h * 4 / 1e6
    for name, fn in (("relu only", lambda: ops.act_fwd(z, out=out)), ("bn+relu", lambda: ops.act_fwd(z, sc, sh, out=out)),
                     ("bn+relu+drop", lambda: ops.act_fwd(z, sc, sh, 0.2, 7, out=out)), ("torch relu", lambda: torch.relu(z, )),
                     ("torch copy_", lambda: out.copy_(z))):
        t = sustained(fn)
        print(f"[{rows}x{h}] {name:14s} {t * 1e6:8.1f} us  {mb / t / 1e6:6.2f} TB/s", flush=True)
