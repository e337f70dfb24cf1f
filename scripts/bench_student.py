"""Student-step micro-benchmark across the reference's student configs (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine

CONFIGS = {  # reference train.conf.yaml
    "arxiv-MLP": dict(dims=[128, 256, 256, 40], B=512, p=0.2, n=169343),
    "arxiv-MLP3w4": dict(dims=[128, 1024, 1024, 40], B=512, p=0.5, n=169343),
    "products-MLP": dict(dims=[100, 256, 256, 47], B=4096, p=0.5, n=2449029),
    "products-MLP3w8": dict(dims=[100, 2048, 2048, 47], B=4096, p=0.2, n=2449029),
    "cora-MLP": dict(dims=[1433, 128, 7], B=140, p=0.6, n=2485, norm="none"),
}
dev = "cuda:0"
which = sys.argv[1:] or list(CONFIGS)
for name in which:
    c = CONFIGS[name]
    d = c["dims"]
    torch.manual_seed(0)
    model = Model(dict(model_name="MLP", num_layers=len(d) - 1, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1],
                       dropout_ratio=c["p"], norm_type=c.get("norm", "batch"), device=dev))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    n = c["n"]
    feats = ops.as_feat(torch.randn(n, d[0], device=dev))
    out_t = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1))
    eng = StudentEngine(model, opt, c["B"])
    nb = max(1, n // c["B"])
    perm = torch.randperm(n)[: nb * c["B"]].view(nb, -1).to(dev)
    steps = 300
    for i in range(20):
        eng.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the reference's loop body with stock torch ops on the same GPU, for scale (not the product path)
    import torch.nn as tnn
    layers = []
    for i in range(len(d) - 1):
        layers.append(tnn.Linear(d[i], d[i + 1]))
        if i < len(d) - 2:
            if c.get("norm", "batch") == "batch":
                layers.append(tnn.BatchNorm1d(d[i + 1]))
            layers += [tnn.ReLU(), tnn.Dropout(c["p"])]
    ref = tnn.Sequential(*layers).to(dev).train()
    ropt = torch.optim.Adam(ref.parameters(), lr=0.01)
    crit = tnn.KLDivLoss(reduction="batchmean", log_target=True)
    def ref_step(i):
        idx = perm[i % nb]
        out = ref(feats[idx]).log_softmax(1)
        loss = crit(out, out_t[idx]); loss.item()
        ropt.zero_grad(); loss.backward(); ropt.step()
    for i in range(10): ref_step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): ref_step(i)
    torch.cuda.synchronize(); rt = (time.perf_counter() - t0) / 100
    print(f"{name:18s} fused {dt*1e3:7.3f} ms/step {1/dt:8.1f} steps/s | torch-eager loop {rt*1e3:7.3f} ms/step {1/rt:8.1f} steps/s | x{rt/dt:.2f}", flush=True)
