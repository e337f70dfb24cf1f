"""60 steps of the products MLP3w8 student step (for a rocprofv3 --kernel-trace timeline); GLNN_STUDENT_TWO_STREAMS selects the mode."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"
d, B, n = [100, 2048, 2048, 47], 4096, 200000
torch.manual_seed(0)
model = Model(dict(model_name="MLP", num_layers=3, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=0.2, norm_type="batch", device=dev))
model.train()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
feats = ops.as_feat(torch.randn(n, d[0], device=dev))
out_t = ops.as_feat(torch.log_softmax(torch.randn(n, d[-1], device=dev), 1))
eng = StudentEngine(model, opt, B)
perm = torch.randperm(n)[: (n // B) * B].view(-1, B).to(dev)
for i in range(60):
    eng.step(feats, perm[i % perm.shape[0]], ops.LOSS_KL, out_t, 1.0)
torch.cuda.synchronize()
