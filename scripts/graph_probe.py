"""What would a HIP graph of the student step buy?  Captures ONE StudentEngine.step (fixed batch rows and dropout seeds: timing
only) into a torch.cuda.CUDAGraph and compares replay time with the eager launch sequence (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine

CONFIGS = {"arxiv-MLP": ([128, 256, 256, 40], 512, 0.2), "arxiv-MLP3w4": ([128, 1024, 1024, 40], 512, 0.5),
           "products-MLP": ([100, 256, 256, 47], 4096, 0.5), "products-MLP3w8": ([100, 2048, 2048, 47], 4096, 0.2)}
dev = "cuda:0"
for name, (d, B, p) in CONFIGS.items():
    torch.manual_seed(0)
    model = Model(dict(model_name="MLP", num_layers=3, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=p, norm_type="batch", device=dev))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    n = 200000
    feats = ops.as_feat(torch.randn(n, d[0], device=dev))
    out_t = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1))
    eng = StudentEngine(model, opt, B)
    idx = torch.randperm(n)[:B].to(dev)
    for _ in range(20):
        eng.step(feats, idx, ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        eng.step(feats, idx, ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 300
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.step(feats, idx, ops.LOSS_KL, out_t, 1.0)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 300
    print(f"{name:18s} eager {eager * 1e3:7.3f} ms/step   graph replay {graph * 1e3:7.3f} ms/step   x{eager / graph:.2f}", flush=True)
