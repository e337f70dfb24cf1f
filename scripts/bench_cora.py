"""Full-graph GCN teacher training step and MLP student step on the synthetic cora-shaped graph (the reference's default
dataset): ms per step (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, teacher, train_and_eval as te
from glnn_amd.models import Model
dev = "cuda:0"
torch.manual_seed(0)
g = data.make_graph("cora", seed=0, device=dev)
n = g.n_dst
feats, labels, _, _ = data.make_node_data("cora", seed=0, device=dev, n=n)
idx_train = torch.randperm(n)[:140].to(dev)
for name in ("GCN", "SAGE"):
    model = Model(dict(model_name=name, num_layers=2, feat_dim=feats.shape[1], hidden_dim=64 if name == "GCN" else 128, label_dim=7,
                       dropout_ratio=0.8 if name == "GCN" else 0.0, norm_type="none", device=dev))
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    crit = torch.nn.NLLLoss()
    if name == "GCN":
        step = lambda: te.train(model, g, feats, labels, crit, opt, idx_train)
    else:
        from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
        loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 5]), batch_size=512, shuffle=True)
        step = lambda: te.train_sage(model, loader, feats, labels, crit, opt)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"cora {name} teacher epoch (one step): {dt * 1e3:.3f} ms", flush=True)
