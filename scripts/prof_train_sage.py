"""cProfile of one train_sage epoch (arxiv config) on the GPU host: where the HOST time of the sampled-block loop goes."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, train_and_eval as te
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
from glnn_amd.models import Model
dev = "cuda:0"
torch.manual_seed(0)
g = data.make_graph("ogbn-arxiv", seed=0, device=dev)
n = g.n_dst
feats, labels, _, _ = data.make_node_data("ogbn-arxiv", seed=0, device=dev, n=n)
model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=128, hidden_dim=256, label_dim=40, dropout_ratio=0.2, norm_type="batch", device=dev))
opt = torch.optim.Adam(model.parameters(), lr=0.01)
idx_train = torch.randperm(n)[:90941].to(dev)
loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=512, shuffle=True, drop_last=False)
crit = torch.nn.NLLLoss()
for _ in range(2):
    te.train_sage(model, loader, feats, labels, crit, opt)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for _ in range(3):
    te.train_sage(model, loader, feats, labels, crit, opt)
pr.disable(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"epoch {dt * 1e3:.1f} ms = {len(loader) / dt:.0f} steps/s (under cProfile)")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
