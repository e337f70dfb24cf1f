"""Full-graph GCN (or full-neighbour-free sampled SAGE) teacher training steps on a synthetic graph of any shape, for a rocprofv3
--kernel-trace timeline:  python scripts/trace_teacher_any.py GCN n avg_deg feat hidden classes norm [layers]
(pokec: GCN 1632803 19 65 32 2 batch; penn94: GCN 41554 33 4814 64 2 batch)"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glnn_amd import data, train_and_eval as te
from glnn_amd.models import Model
dev = "cuda:0"
name, n, deg, f, h, c, norm = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
L = int(sys.argv[8]) if len(sys.argv) > 8 else 2
torch.manual_seed(0)
g = data.make_uniform_graph(n, deg, seed=0, device=dev)
if name == "GCN":
    from glnn_amd.graph import CSRGraph
    # symmetric + self loops, as the reference prepares its GCN graphs
    src = torch.repeat_interleave(torch.arange(n, device=dev), (g.indptr[1:] - g.indptr[:-1]))
    dst = g.indices.long()
    e_s = torch.cat([src, dst, torch.arange(n, device=dev)]); e_d = torch.cat([dst, src, torch.arange(n, device=dev)])
    key = torch.unique(e_d * n + e_s)
    g = data.csr_from_edges(key % n, key // n, n)
feats = torch.randn(n, f, device=dev)
labels = torch.randint(0, c, (n,), device=dev)
idx_train = torch.randperm(n)[: max(n // 2, 1)].to(dev)
model = Model(dict(model_name=name, num_layers=L, feat_dim=f, hidden_dim=h, label_dim=c, dropout_ratio=0.5, norm_type=norm, device=dev))
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)
crit = torch.nn.NLLLoss()
if name == "GCN":
    step = lambda: te.train(model, g, feats, labels, crit, opt, idx_train)
else:
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    loader = NodeDataLoader(g, idx_train[:20480], MultiLayerNeighborSampler([5] * L), batch_size=4096, shuffle=True)
    step = lambda: te.train_sage(model, loader, feats, labels, crit, opt)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(f"{' '.join(sys.argv[1:])}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per epoch call", flush=True)
