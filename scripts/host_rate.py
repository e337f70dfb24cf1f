"""Host issue time vs wall time of one small-batch student step (is the loop host- or GPU-bound?) -- development aid."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"
for name, d, B, p in (("arxiv-MLP", [128, 256, 256, 40], 512, 0.2), ("arxiv-MLP3w4", [128, 1024, 1024, 40], 512, 0.5)):
    torch.manual_seed(0)
    model = Model(dict(model_name="MLP", num_layers=3, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=p, norm_type="batch", device=dev)); model.train()
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    n = 169343
    feats = ops.as_feat(torch.randn(n, d[0], device=dev)); out_t = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1))
    eng = StudentEngine(model, opt, B)
    perm = torch.randperm(n)[: (n // B) * B].view(-1, B).to(dev)
    for i in range(50): eng.step(feats, perm[i], ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize()
    # host issue rate: tiny GPU work (B=1 rows) so the queue never fills
    t0 = time.perf_counter()
    for i in range(300): eng.step(feats, perm[i % perm.shape[0]], ops.LOSS_KL, out_t, 1.0)
    t_issue = (time.perf_counter() - t0) / 300
    torch.cuda.synchronize(); t_total = (time.perf_counter() - t0) / 300
    print(f"{name}: host issue {t_issue * 1e6:.1f} us/step, wall {t_total * 1e6:.1f} us/step", flush=True)
