"""A/B of the small student steps with and without the batched (deferred) weight gradients (GLNN_STUDENT_BATCHED_WGRAD, read per
step), interleaved in one process; first checks bit-identical parameters after 5 steps.  usage: python scripts/ab_student_batched_wgrad.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine

CONFIGS = {
    "arxiv-MLP": dict(dims=[128, 256, 256, 40], B=512, p=0.2, n=169343, norm="batch"),
    "arxiv-MLP3w4": dict(dims=[128, 1024, 1024, 40], B=512, p=0.5, n=169343, norm="batch"),
    "cora-MLP": dict(dims=[1433, 128, 7], B=140, p=0.6, n=2485, norm="none"),
    "products-MLP": dict(dims=[100, 256, 256, 47], B=4096, p=0.5, n=400000, norm="batch"),
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
    for i in range(5):
        for mode in ("0", "1"):
            os.environ["GLNN_STUDENT_BATCHED_WGRAD"] = mode
            engs[mode][0].step(feats, perm[i], ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(engs["0"][1].state_dict().values(), engs["1"][1].state_dict().values()))
    res = {"0": [], "1": []}
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["GLNN_STUDENT_BATCHED_WGRAD"] = mode
            e = engs[mode][0]
            for i in range(50):
                e.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(1000):
                e.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 1000 * 1e3)
    t0, t1 = min(res["0"]), min(res["1"])
    print(f"{name:14s} per-layer wgrads {t0:.4f} ms   batched {t1:.4f} ms   x{t0 / t1:.3f}   bit-identical after 5 steps: {same}", flush=True)
