"""Rounding noise of the fp32 GEMMs on the wide students' shapes: rms / max error against an fp64 product of the same operands,
for the HIP kernels, numpy (the oracle's BLAS) and torch fp32 on the device.  usage: python scripts/gemm_noise.py
(is the sequential MFMA accumulation chain measurably noisier than a blocked CPU sgemm?)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glnn_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)


def err(c, ref):
    d = (c.double() - ref).abs()
    s = ref.pow(2).mean().sqrt()
    return float(d.pow(2).mean().sqrt() / s), float(d.max() / s)


def report(tag, rows):
    print(f"{tag:44s} " + " | ".join(f"{n} rms {r:.2e} max {m:.2e}" for n, (r, m) in rows), flush=True)


for kind in ("normal", "relu"):
    for m, k, n in ((4096, 2048, 2048), (4096, 4096, 2048), (4096, 1024, 1024), (4096, 128, 2048)):
        a = torch.randn(m, k, device=DEV)
        if kind == "relu":
            a = a.relu()                       # the hidden activations: a positive mean, partial sums that grow along K
        w = torch.randn(n, k, device=DEV) / k ** 0.5
        ref = a.double() @ w.double().t()
        c_hip = ops.gemm(a, w)
        c_t = a @ w.t()
        c_np = torch.from_numpy(a.cpu().numpy() @ w.cpu().numpy().T).to(DEV)
        report(f"{kind} A[{m},{k}] . W[{n},{k}]^T", (("hip", err(c_hip, ref)), ("numpy", err(c_np, ref)), ("torch", err(c_t, ref))))
        wk = w.t().contiguous()
        c_hip = ops.gemm(a, wk, w_is_kn=True)
        report(f"{kind} A[{m},{k}] . W[{k},{n}]", (("hip", err(c_hip, ref)),))
    # weight gradient: reduction over the batch rows
    for mrows, ka, nb in ((4096, 2048, 2048), (4096, 1024, 1024)):
        a = torch.randn(mrows, ka, device=DEV)
        b = torch.randn(mrows, nb, device=DEV)
        if kind == "relu":
            b = b.relu()
        ref = a.double().t() @ b.double()
        c_hip = ops.gemm_tn(a, b)
        c_t = a.t() @ b
        c_np = torch.from_numpy(a.cpu().numpy().T @ b.cpu().numpy()).to(DEV)
        report(f"{kind} A[{mrows},{ka}]^T . B[{mrows},{nb}]", (("hip", err(c_hip, ref)), ("numpy", err(c_np, ref)), ("torch", err(c_t, ref))))
