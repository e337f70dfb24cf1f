"""A/B of the large-batch student step with the first hidden layer's BatchNorm backward as partial + apply passes and a plain weight gradient
(GLNN_STUDENT_FUSE_APPLY_BIG=0) vs the partial pass only and the rest applied in the operand loads of the first layer's weight-gradient GEMM
(default; glnn::gemm_tn_bn: dz of that layer is never written), interleaved in one process; first compares loss and gradients of one step
(different summation orders: agreement to fp32 rounding, not bit for bit).
usage: python scripts/ab_student_fuse_big.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine

CONFIGS = {
    "products-MLP3w8": dict(dims=[100, 2048, 2048, 47], B=4096, p=0.2, n=400000, norm="batch"),
    "products-MLP": dict(dims=[100, 256, 256, 47], B=4096, p=0.5, n=400000, norm="batch"),
    "B2048-w1024-c40": dict(dims=[128, 1024, 1024, 40], B=2048, p=0.5, n=169343, norm="batch"),
    "B8192-w512-c7": dict(dims=[64, 512, 7], B=8192, p=0.0, n=100000, norm="batch"),
    "B4096-w512-c70": dict(dims=[100, 512, 512, 70], B=4096, p=0.3, n=100000, norm="batch"),
}
dev = "cuda:0"
for name, c in CONFIGS.items():
    d = c["dims"]
    feats = ops.as_feat(torch.randn(c["n"], d[0], device=dev))
    out_t = ops.as_feat(torch.log_softmax(torch.randn(c["n"], d[-1], device=dev), 1))
    nb = c["n"] // c["B"]
    perm = torch.randperm(c["n"])[: nb * c["B"]].view(nb, -1).to(dev)
    engs = {}
    for mode in ("0", "1"):
        torch.manual_seed(0)
        model = Model(dict(model_name="MLP", num_layers=len(d) - 1, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=c["p"],
                           norm_type=c["norm"], device=dev))
        model.train()
        engs[mode] = (StudentEngine(model, torch.optim.Adam(model.parameters(), lr=0.01), c["B"]), model)
    def step(mode, i):
        os.environ["GLNN_STUDENT_FUSE_APPLY_BIG"] = mode            # read by every C call
        engs[mode][0].step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
    for mode in ("0", "1"):
        step(mode, 0)
    torch.cuda.synchronize()
    gs = max(float(g.abs().max()) for g in engs["0"][0].grads)
    err = max(float((a - b).abs().max()) for a, b in zip(engs["0"][0].grads, engs["1"][0].grads)) / gs
    lerr = abs(float(engs["0"][0].loss_out) - float(engs["1"][0].loss_out))
    res = {"0": [], "1": []}
    for rnd in range(3):
        for mode in ("0", "1"):
            for i in range(30):
                step(mode, i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(300):
                step(mode, i)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 300 * 1e3)
    t0, t1 = min(res["0"]), min(res["1"])
    print(f"{name:18s} apply pass {t0:.4f} ms   in the operand loads {t1:.4f} ms   x{t0 / t1:.3f}   max gradient difference / largest gradient {err:.1e}, loss {lerr:.1e}",
          flush=True)
