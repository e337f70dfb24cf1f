"""Eval-mode student pass (what evaluate_mini_batch / serving runs) over ogbn-products-sized inputs: rows/s.
python scripts/bench_eval.py [rows]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
for name, dims in (("MLP3w8", [100, 2048, 2048, 47]), ("MLP", [100, 256, 256, 47]), ("arxiv MLP3w4", [128, 1024, 1024, 40])):
    torch.manual_seed(0)
    model = Model(dict(model_name="MLP", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.2,
                       norm_type="batch", device=dev))
    model.eval()
    x = ops.as_feat(torch.randn(n, dims[0], device=dev))
    fl = 2.0 * n * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    res = []
    for label, fn in (("logits only", lambda: model(None, x)), ("with hidden outputs", lambda: model.forward_fitnet(None, x))):
        for _ in range(3): y = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): y = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        res.append(f"{label}: {dt * 1e3:7.2f} ms = {n / dt / 1e6:6.1f} M rows/s, {fl / dt / 1e12:6.1f} TF")
    a = model(None, x); b = model.forward_fitnet(None, x)[1]
    print(f"{name:14s} {n} rows | " + " | ".join(res) + f" | max |diff| between the two forms {float((a - b).abs().max()):.2e}", flush=True)
