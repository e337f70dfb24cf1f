"""One full products-configuration batch of sampled-block SAGE training (B = 4096, fan-out 5,10,15) stepped through the TeacherEngine alone
(no sampling beside it), for a rocprofv3 --kernel-trace timeline (scripts/train_sage_timeline.sh)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, teacher
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
loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=4096, shuffle=True, drop_last=False)
input_nodes, output_nodes, blocks = next(iter(loader))
torch.cuda.synchronize()
model.train()
eng = teacher.get_engine(model, opt)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
torch.cuda.synchronize()
print(f"products teacher-training step (engine alone): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms; block rows "
      f"{[b.num_dst_nodes() for b in blocks]}, sources {input_nodes.numel()}", flush=True)
