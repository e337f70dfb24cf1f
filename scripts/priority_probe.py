"""Does a stream priority change what the sampler (side stream) costs the sampled-block training step (products configuration)?
modes: default | main_high (the consumer's stream at the highest priority) | side_high (the sampler's stream at the highest priority) |
nothread (GLNN_LOADER_THREAD=0: batches built by the consumer's thread, one ahead).
python scripts/priority_probe.py MODE  ->  ms per step over two epochs of 48 steps"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, graph, train_and_eval as te
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
from glnn_amd.models import Model
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
torch.manual_seed(0)
g = data.make_graph("ogbn-products", seed=0, device=dev)
n = g.n_dst
feats, labels, _, _ = data.make_node_data("ogbn-products", seed=0, device=dev, n=n)
model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=100, hidden_dim=256, label_dim=47, dropout_ratio=0.5, norm_type="batch", device=dev))
opt = torch.optim.Adam(model.parameters(), lr=0.003)
idx_train = torch.randperm(n)[:196608].to(dev)
loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=4096, shuffle=True, drop_last=False)
if mode == "nothread":
    loader.threaded = False
crit = torch.nn.NLLLoss()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
if mode == "side_high":
    graph._SIDE_STREAMS[torch.cuda.current_device()] = torch.cuda.Stream(dev, priority=-1)
main = torch.cuda.Stream(dev, priority=-1) if mode == "main_high" else torch.cuda.current_stream()
with torch.cuda.stream(main):
    for ep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = te.train_sage(model, loader, feats, labels, crit, opt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{mode} epoch {ep}: {1e3 * dt / len(loader):.3f} ms per step, loss {loss:.4f}", flush=True)
