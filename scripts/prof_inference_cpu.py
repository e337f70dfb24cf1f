"""Host-side cost of SAGE.inference on the arxiv-shaped graph (the forward is ~0.9 ms of kernels: three launches -- the Python around them must stay below that): cProfile of 300 forwards."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.graph import FullNeighborLoader
from glnn_amd.models import Model
dev = "cuda:0"
g = data.make_graph("ogbn-arxiv", seed=0, device=dev, scale=1.0)
feats, labels, out_t, _ = data.make_node_data("ogbn-arxiv", seed=0, device=dev, n=g.n_dst)
feats = ops.as_feat(feats)
m = Model(dict(model_name="SAGE", num_layers=3, feat_dim=128, hidden_dim=256, label_dim=40, dropout_ratio=0.5, norm_type="batch", device=dev))
m.eval()
loader = FullNeighborLoader(g, 4096)
for _ in range(10):
    m.inference(loader, feats)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    m.inference(loader, feats)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per forward {1e3 * (t1 - t0) / 300:.3f} ms; with the final sync {1e3 * (t2 - t0) / 300:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    m.inference(loader, feats)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
