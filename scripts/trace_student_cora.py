"""200 steps of the cora student (MLP 1433-128-7, full batch of 140 rows, no norm, p = 0.6) for a rocprofv3 --kernel-trace timeline."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"
d, B, n, p = [1433, 128, 7], 140, 2485, 0.6
torch.manual_seed(0)
model = Model(dict(model_name="MLP", num_layers=2, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=p, norm_type="none", device=dev))
model.train()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
feats = ops.as_feat(torch.randn(n, d[0], device=dev))
out_t = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1))
eng = StudentEngine(model, opt, B)
perm = torch.randperm(n)[: (n // B) * B].view(-1, B).to(dev)
for i in range(200):
    eng.step(feats, perm[i % perm.shape[0]], ops.LOSS_KL, out_t, 1.0)
torch.cuda.synchronize()
