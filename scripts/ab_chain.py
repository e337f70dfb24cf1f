"""Interleaved A/B of the chained next-layer projection in SAGE.inference (products shape): ms per forward, and max |diff|."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.graph import FullNeighborLoader
from glnn_amd.models import SAGE, Model
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "ogbn-products"
dims = [100, 256, 256, 47] if name == "ogbn-products" else [128, 256, 256, 40]
g = data.make_graph(name, seed=0, device=dev)
feats = ops.as_feat(torch.randn(g.n_dst, dims[0], device=dev))
model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=256, label_dim=dims[-1], dropout_ratio=0.5, norm_type="batch", device=dev))
model.eval()
loader = FullNeighborLoader(g, 4096)
outs = {}
for rep in range(3):
    for chain in (False, True):
        SAGE.CHAIN_NEXT_PROJECTION = chain
        outs[chain] = model.inference(loader, feats)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            model.inference(loader, feats)
        torch.cuda.synchronize()
        print(f"chain={chain}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms / forward", flush=True)
print("max |diff|", float((outs[True] - outs[False]).abs().max()))
