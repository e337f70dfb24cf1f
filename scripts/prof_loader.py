"""cProfile of the host side of the sampled-block loader alone (arxiv config) -- development aid."""
import os, sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from glnn_amd import data
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
dev = "cuda:0"
g = data.make_graph("ogbn-arxiv", seed=0, device=dev)
idx_train = torch.randperm(g.n_dst)[:90941].to(dev)
loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=512, shuffle=True)
loader.prefetch = False
for b in loader: pass
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for b in loader: n += 1
torch.cuda.synchronize(); print("loader alone:", (time.perf_counter() - t0) / n * 1e3, "ms/batch")
pr = cProfile.Profile(); pr.enable()
for b in loader: pass
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
