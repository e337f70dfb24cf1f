import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from glnn_amd import ops, train_and_eval as te
from glnn_amd.models import Model
dev = "cuda:0"; n = 90941; dims=[128,256,256,40]
torch.manual_seed(0)
model = Model(dict(model_name="MLP", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.2, norm_type="batch", device=dev))
opt = torch.optim.Adam(model.parameters(), lr=0.01)
feats = ops.as_feat(torch.randn(n, dims[0], device=dev)); out_t = ops.as_feat(torch.log_softmax(torch.randn(n, dims[-1], device=dev), 1))
crit = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
for _ in range(2): te.train_mini_batch(model, feats, out_t, 512, crit, opt, 1.0)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): te.train_mini_batch(model, feats, out_t, 512, crit, opt, 1.0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
