"""Does re-creating the NodeDataLoader (one per epoch loop, as a caller might) cost time or memory?  Six loaders in one process on the products
training configuration: ms per step and the allocator's reserved memory after each (development aid, round 5: a side stream PER LOADER grew the
reserved memory by 2 GB per loader -- the allocator pools per stream; the loaders now share one side stream per device)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, train_and_eval as te
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
from glnn_amd.models import Model
dev = "cuda:0"
torch.manual_seed(0)
g = data.make_graph("ogbn-products", seed=0, device=dev)
n = g.n_dst
feats, labels, _, _ = data.make_node_data("ogbn-products", seed=0, device=dev, n=n)
model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=100, hidden_dim=256, label_dim=47, dropout_ratio=0.5, norm_type="batch", device=dev))
opt = torch.optim.Adam(model.parameters(), lr=0.003)
idx_train = torch.randperm(n)[:196615].to(dev)
crit = torch.nn.NLLLoss()
shared = None
for k in range(6):
    loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=4096, shuffle=True, drop_last=False)
    te.train_sage(model, loader, feats, labels, crit, opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        te.train_sage(model, loader, feats, labels, crit, opt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (3 * len(loader)) * 1e3
    print(f"loader {k} : {ms:.3f} ms per step; reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB", flush=True)
