"""A/B of the small student steps as two C calls (glnn_mlp_fwd_bwd_f32 + glnn_adam_step_f32) vs ONE (glnn_mlp_train_step_f32: Adam
folds the backward's gradient partials), interleaved in one process; first checks parameters, gradients and loss after 5 steps bit for bit.
usage: python scripts/ab_student_one_call.py"""
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
    "products-MLP3w8": dict(dims=[100, 2048, 2048, 47], B=4096, p=0.2, n=400000, norm="batch"),
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
        os.environ["GLNN_STUDENT_ONE_CALL"] = mode              # read when the engine is built
        engs[mode] = (StudentEngine(model, torch.optim.Adam(model.parameters(), lr=0.01), c["B"]), model)
    for i in range(5):
        for mode in ("0", "1"):
            engs[mode][0].step(feats, perm[i], ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(engs["0"][1].state_dict().values(), engs["1"][1].state_dict().values()))
    same = same and all(torch.equal(a, b) for a, b in zip(engs["0"][0].grads, engs["1"][0].grads))
    same = same and torch.equal(engs["0"][0].loss_out, engs["1"][0].loss_out)
    res = {"0": [], "1": []}
    for rnd in range(3):
        for mode in ("0", "1"):
            e = engs[mode][0]
            for i in range(50):
                e.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(1000):
                e.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 1000 * 1e3)
    t0, t1 = min(res["0"]), min(res["1"])
    print(f"{name:14s} two calls {t0:.4f} ms   one call {t1:.4f} ms   x{t0 / t1:.3f}   bit-identical after 5 steps: {same}", flush=True)
