"""Cross-process reproducibility of the sampled blocks (products configuration): checksums of every sampler / block-builder output of the first
batches.  Run it in several processes and diff the lines.  python scripts/block_repro_probe.py [batches]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
dev = "cuda:0"
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
g = data.make_graph("ogbn-products", seed=0, device=dev)
n = g.n_dst
idx_train = torch.randperm(n)[: 4096 * nb].to(dev)
s0, b0 = ops.sample_neighbors, ops.block_build


def cs(t):
    t = t.to(torch.int64).reshape(-1)
    return f"{int(t.sum())}/{int((t * (torch.arange(t.numel(), device=t.device) % 1009 + 1)).sum())}"


def sample(indptr, indices, seeds, fanout, rng):
    smp, cnt = s0(indptr, indices, seeds, fanout, rng)
    valid = torch.arange(fanout, device=dev)[None, :] < cnt[:, None]
    print(f"  sample fanout {fanout} seeds {seeds.numel()} ({cs(seeds)}) cnt {cs(cnt)} src {cs(torch.where(valid, smp.view(-1, fanout), 0))}", flush=True)
    return smp, cnt


def build(seeds, *a, **k):
    out = b0(seeds, *a, **k)
    indptr, indices, gidx, input_nodes, nnz, n_src = out
    print(f"  block ns {seeds.numel()} nnz {nnz} n_src {n_src} indptr {cs(indptr)} indices {cs(indices[:nnz])} input_nodes {cs(input_nodes[:n_src])}"
          f" distinct {torch.unique(input_nodes[:n_src]).numel()}", flush=True)
    return out


ops.sample_neighbors, ops.block_build = sample, build
loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=4096, shuffle=False, drop_last=False, seed=1234)
loader.prefetch = os.environ.get("PREFETCH", "1") == "1"
for b, (input_nodes, output_nodes, blocks) in enumerate(loader):
    print(f"batch {b}: sources {input_nodes.numel()}", flush=True)
