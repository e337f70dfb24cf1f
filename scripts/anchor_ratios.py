"""How far the HIP student lands from the reference's float64 run after training, relative to the reference's own fp32 run
(tests/parity_rules.py: check_eval_out).  usage: python scripts/anchor_ratios.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_inputs import CASES, Golden
from glnn_amd import train_and_eval as te
from glnn_amd.models import Model
DEV = "cuda:0"
for name in CASES:
    g = Golden(name)
    if g.dropout > 0:
        continue
    L = len(g.dims) - 1
    model = Model(dict(model_name="MLP", num_layers=L, feat_dim=g.dims[0], hidden_dim=g.dims[1], label_dim=g.dims[-1],
                       dropout_ratio=0.0, norm_type=g.norm, device=DEV))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in g.sd0.items()}, strict=True)
    opt = torch.optim.Adam(model.parameters(), lr=g.lr, weight_decay=g.wd)
    cl, ct = torch.nn.NLLLoss(), torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    feats, labels, out_t = (torch.from_numpy(a).to(DEV) for a in (g.feats, g.labels, g.out_t))
    idx_l = torch.from_numpy(g.idx_l).to(DEV)
    perms = iter(g.perms)
    real = torch.randperm
    torch.randperm = lambda *a, **k: torch.from_numpy(next(perms))
    try:
        for _ in range(g.epochs):
            te.train_mini_batch(model, feats[idx_l], labels[idx_l], g.B, cl, opt, g.lamb)
            te.train_mini_batch(model, feats, out_t, g.B, ct, opt, 1 - g.lamb)
    finally:
        torch.randperm = real
    out, _, _ = te.evaluate_mini_batch(model, feats, labels, cl, g.B, lambda o, y: 0.0)
    o = g.view(out.cpu().numpy()).astype(np.float64)
    d = np.abs(o - g.z["f64.eval_out"]); dist = g.z["f64.dist_eval_out"]; d32 = np.abs(o - g.z["eval_out"])
    print(f"{name:24s} hip-f64 max {d.max():.3e} mean {d.mean():.3e} | ref32-f64 max {dist[0]:.3e} mean {dist[1]:.3e} | ratio max "
          f"{d.max() / dist[0]:.2f} mean {d.mean() / dist[1]:.2f} | hip-ref32 max {d32.max():.3e} mean {d32.mean():.3e}", flush=True)
