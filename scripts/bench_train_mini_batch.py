"""train_mini_batch (the reference's loop, train_and_eval.py:59-86) end to end on the arxiv students: seconds per pass and steps/s as a
caller of the preserved surface sees them (randperm + H2D copy of the batch table + the per-step Python around StudentEngine.step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops, train_and_eval as te
from glnn_amd.models import Model
dev = "cuda:0"
n = 90941                                                      # ogbn-arxiv training rows
for name, dims, p in (("MLP", [128, 256, 256, 40], 0.2), ("MLP3w4", [128, 1024, 1024, 40], 0.5)):
    torch.manual_seed(0)
    model = Model(dict(model_name="MLP", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                       norm_type="batch", device=dev))
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    feats = ops.as_feat(torch.randn(n, dims[0], device=dev))
    out_t = ops.as_feat(torch.log_softmax(torch.randn(n, dims[-1], device=dev), 1))
    crit = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    for _ in range(2):
        te.train_mini_batch(model, feats, out_t, 512, crit, opt, 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        loss = te.train_mini_batch(model, feats, out_t, 512, crit, opt, 1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    steps = n // 512
    print(f"arxiv {name}: {steps} steps per pass in {dt * 1e3:.2f} ms = {steps / dt:.0f} steps/s ({dt / steps * 1e6:.1f} us per step), loss {loss:.4f}", flush=True)
