"""200 steps of an MLP student of any shape for a rocprofv3 --kernel-trace timeline:
python scripts/trace_student_any.py 65-256-256-2 512 none 0.2 [kl|nll]"""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"
d = [int(v) for v in sys.argv[1].split("-")]
B, norm, p = int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
kind = sys.argv[5] if len(sys.argv) > 5 else "kl"
n = max(8 * B, 50000)
torch.manual_seed(0)
model = Model(dict(model_name="MLP", num_layers=len(d) - 1, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=p, norm_type=norm, device=dev))
model.train()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
feats = ops.as_feat(torch.randn(n, d[0], device=dev))
tgt = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1)) if kind == "kl" else torch.randint(0, d[-1], (n,), device=dev)
eng = StudentEngine(model, opt, B)
perm = torch.randperm(n)[: (n // B) * B].view(-1, B).to(dev)
import time
for i in range(50):
    eng.step(feats, perm[i % perm.shape[0]], ops.LOSS_KL if kind == "kl" else ops.LOSS_NLL, tgt, 1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200):
    eng.step(feats, perm[i % perm.shape[0]], ops.LOSS_KL if kind == "kl" else ops.LOSS_NLL, tgt, 1.0)
torch.cuda.synchronize()
print(f"{sys.argv[1]} B={B} {norm} p={p} {kind}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step", flush=True)
