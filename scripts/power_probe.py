"""Socket power and engine clock under sustained launches of one kernel at a time (rocm-smi sampled from a side thread), to tell a
power-limited kernel from an instruction-bound one.  usage: python scripts/power_probe.py [seconds per kernel]
GLNN_LIB_PATH selects the library build; GLNN_POWER_KERNELS = comma list of {pipe,walk,panel,tiled,fused,spmm47,adam-ish}."""
import os, re, subprocess, sys, threading, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import _lib, data, ops
dev = "cuda:0"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
which = os.environ.get("GLNN_POWER_KERNELS", "idle,pipe,walk,panel,tiled,fused256,spmm47").split(",")
label = os.environ.get("GLNN_POWER_LABEL", "default")


def smi():
    """(watts, sclk MHz, mclk MHz) from one rocm-smi call; None where the field is absent."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None
    w = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([0-9.]+)", out) or re.search(r"Power \(W\):\s*([0-9.]+)", out)
    s = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    m = re.search(r"mclk clock level:.*?\((\d+)Mhz\)", out)
    return (float(w.group(1)) if w else None, int(s.group(1)) if s else None, int(m.group(1)) if m else None, out if not (w or s) else None)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            r = smi()
            if r:
                self.rows.append(r)
            time.sleep(0.05)


def setmode(mode):
    os.environ["GLNN_GEMM_ROWPANEL"] = mode
    _lib.lib().glnn_reload_options()


def sustained(name, fn, flops=None, gbytes=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        k += 20
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.stop = True; s.join()
    rows = s.rows[1:] if len(s.rows) > 2 else s.rows          # the first sample straddles the start
    w = [r[0] for r in rows if r[0] is not None]; c = [r[1] for r in rows if r[1] is not None]; mc = [r[2] for r in rows if r[2] is not None]
    per = dt / max(k, 1)
    line = f"[{label}] {name:30s} {per * 1e6:9.1f} us/launch"
    if flops:
        line += f" {flops / per / 1e12:6.1f} TF"
    if gbytes:
        line += f" {gbytes / per / 1e3:6.2f} TB/s(alg)"
    line += f"  power W mean {statistics.mean(w):7.1f} max {max(w):7.1f}" if w else "  power n/a"
    line += f"  sclk MHz mean {statistics.mean(c):6.0f} min {min(c)} max {max(c)}" if c else "  sclk n/a"
    line += f"  mclk {max(mc)}" if mc else ""
    line += f"  samples {len(rows)}"
    print(line, flush=True)
    if rows and rows[0][3]:
        print(rows[0][3][:1500])


if "idle" in which:
    sustained("idle (sleep)", lambda: time.sleep(0.01))
m, k, n = 2449029, 100, 256
a = ops.as_feat(torch.randn(m, k, device=dev)); w = torch.randn(n, k, device=dev) / 10; out = ops.feat_empty(m, n, dev)
es, eh = torch.rand(n, device=dev) + .5, torch.randn(n, device=dev)
for nm, mode in (("walk", "1"), ("panel", "2"), ("tiled", "0")):
    if nm in which:
        setmode(mode)
        sustained(f"K3 {nm} 2.45M x100x256", lambda: ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=True, out=out), 2.0 * m * k * n, (m * (400 + 1024)) / 1e9)
setmode("1")
del a, out
if "pipe" in which:
    A = torch.randn(4096, 2048, device=dev); W = torch.randn(2048, 2048, device=dev) / 45; C = torch.empty(4096, 2048, device=dev)
    sustained("K3 pipe 4096x2048x2048", lambda: ops.gemm(A, W, out=C), 2.0 * 4096 * 2048 * 2048)
    A = torch.randn(32768, 2048, device=dev); C = torch.empty(32768, 2048, device=dev)
    sustained("K3 pipe 32768x2048x2048", lambda: ops.gemm(A, W, out=C), 2.0 * 32768 * 2048 * 2048)
    del A, W, C
if "fused256" in which or "spmm47" in which:
    g = data.make_graph("ogbn-products", seed=0, device=dev, scale=1.0)
    nn_, nnz = g.n_dst, g.num_edges()
    if "spmm47" in which:
        x = ops.as_feat(torch.randn(nn_, 47, device=dev))
        sustained("K1 spmm d=47 products", lambda: ops.spmm(g.indptr, g.indices, x, nn_, ops.AGG_SAGE_GCN), None, (nnz * (4 * 47 + 4) + nn_ * (8 * 47 + 8)) / 1e9)
    if "fused256" in which:
        x = ops.as_feat(torch.randn(nn_, 256, device=dev)); wf = torch.randn(256, 256, device=dev) / 16
        sustained("K1F fused d=256->256 products", lambda: ops.sage_fused(g.indptr, g.indices, x, nn_, wf), None, (nnz * (4 * 256 + 4) + nn_ * (8 * 256 + 8)) / 1e9)
