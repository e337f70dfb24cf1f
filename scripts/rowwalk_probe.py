"""K3w / K3r (csrc/gemm_rowpanel.hip) on their shapes, one library build per process: HIP-event medians, bit-equality against the tiled
kernels, and (with the statistics epilogue) the layer-0 / student shapes.  usage: python scripts/rowwalk_probe.py [label] [modes]
(GLNN_LIB_PATH selects the build; modes = comma list of GLNN_GEMM_ROWPANEL values to time, default "1,2")"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import _lib, ops
dev = "cuda:0"
label = sys.argv[1] if len(sys.argv) > 1 else "default"
modes = (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")
shapes = [("products replicated projection", 2449029, 100, 256), ("teacher-training layer 0", 500000, 100, 256), ("student first layer", 4096, 100, 2048),
          ("xl chunk (1/10)", 2500000, 128, 256)]


def setmode(mode):
    os.environ["GLNN_GEMM_ROWPANEL"] = mode
    _lib.lib().glnn_reload_options()


def timed(fn, reps=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


for what, m, k, n in shapes:
    a = ops.as_feat(torch.randn(m, k, device=dev))
    w = torch.randn(n, k, device=dev) / k ** 0.5
    es, eh = torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev)
    out = ops.feat_empty(m, n, dev)
    setmode("0")
    ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=True, out=out)
    ref = out.clone()
    line = f"[{label}] {what:32s} m={m:8d} k={k:3d} n={n:4d}"
    fl = 2.0 * m * k * n
    for mode in modes:
        setmode(mode)
        out.zero_()
        ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=True, out=out)
        same = bool(torch.equal(out, ref))
        t = timed(lambda: ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=True, out=out))
        line += f"  mode{mode} {t * 1e3:8.1f} us = {fl / t / 1e9:6.1f} TF {'==' if same else '!= TILED'}"
    print(line, flush=True)
    del a, out, ref

# statistics epilogue: z bit-identical to the two-call form, statistics close; timed
for what, m, k, n in [("teacher-training layer 0 + stats", 500000, 100, 256), ("student first layer + stats", 4096, 100, 2048)]:
    a = ops.as_feat(torch.randn(m, k, device=dev))
    w = torch.randn(n, k, device=dev) / k ** 0.5
    bias = torch.randn(n, device=dev) * 5
    gamma, beta = torch.rand(n, device=dev) + .5, torch.randn(n, device=dev) * .2
    line = f"[{label}] {what:32s} m={m:8d} k={k:3d} n={n:4d}"
    res = {}
    for mode in modes:
        setmode(mode)
        rm, rv, nbt = torch.zeros(n, device=dev), torch.ones(n, device=dev), torch.tensor([0], device=dev)
        z, mean, rstd, a_sc, a_sh = ops.linear_bn_stats(a, w, bias, gamma, beta, rm, rv, nbt)
        zd = z[:, :n].double()
        mu, var = zd.mean(0), zd.var(0, unbiased=False)
        e_mean = float((mean.double() - mu).abs().max())
        e_rstd = float((rstd.double() * torch.sqrt(var + 1e-5) - 1).abs().max())
        t = timed(lambda: ops.linear_bn_stats(a, w, bias, gamma, beta, rm, rv, nbt))
        res[mode] = z[:, :n].clone()
        line += f"  mode{mode} {t * 1e3:8.1f} us  |mean err| {e_mean:.1e} rstd rel {e_rstd:.1e}"
    ms = list(res)
    if len(ms) > 1:
        line += "  z " + ("==" if torch.equal(res[ms[0]], res[ms[1]]) else "!=")
    print(line, flush=True)
setmode("1")
