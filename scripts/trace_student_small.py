"""200 steps of a B = 512 student (arxiv MLP 128-256-256-40 by default, or "w4": MLP3w4) for a rocprofv3 --kernel-trace timeline."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"
w4 = len(sys.argv) > 1 and sys.argv[1] == "w4"
d, B, n, p = ([128, 1024, 1024, 40] if w4 else [128, 256, 256, 40]), 512, 169343, (0.5 if w4 else 0.2)
torch.manual_seed(0)
model = Model(dict(model_name="MLP", num_layers=3, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=p, norm_type="batch", device=dev))
model.train()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
feats = ops.as_feat(torch.randn(n, d[0], device=dev))
out_t = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1))
eng = StudentEngine(model, opt, B)
perm = torch.randperm(n)[: (n // B) * B].view(-1, B).to(dev)
for i in range(200):
    eng.step(feats, perm[i % perm.shape[0]], ops.LOSS_KL, out_t, 1.0)
torch.cuda.synchronize()
