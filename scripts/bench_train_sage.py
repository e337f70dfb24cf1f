"""Sampled-block SAGE teacher training (reference train_sage, train_and_eval.py:28-56; fan-out 5,10,15, B=512) on the
synthetic ogbn-arxiv graph: steps/s of sampling + gather + forward + backward + Adam (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, train_and_eval as te
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
from glnn_amd.models import Model
dev = "cuda:0"
torch.manual_seed(0)
# reference train.conf.yaml:170-177 (arxiv: B=512, dropout 0.2, lr 0.01) / :196-204 (products: B=4096, dropout 0.5, lr 0.003)
CFG = {"ogbn-arxiv": dict(f=128, c=40, B=512, p=0.2, lr=0.01, n_train=90941), "ogbn-products": dict(f=100, c=47, B=4096, p=0.5, lr=0.003, n_train=196615)}
name = sys.argv[1] if len(sys.argv) > 1 else "ogbn-arxiv"
c = CFG[name]
g = data.make_graph(name, seed=0, device=dev)
n = g.n_dst
feats, labels, _, _ = data.make_node_data(name, seed=0, device=dev, n=n)
model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=c["f"], hidden_dim=256, label_dim=c["c"], dropout_ratio=c["p"], norm_type="batch", device=dev))
opt = torch.optim.Adam(model.parameters(), lr=c["lr"])
idx_train = torch.randperm(n)[:c["n_train"]].to(dev)
loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=c["B"], shuffle=True, drop_last=False)
crit = torch.nn.NLLLoss()
for ep in range(int(os.environ.get("GLNN_BENCH_EPOCHS", "3"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = te.train_sage(model, loader, feats, labels, crit, opt)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epoch {ep}: {len(loader)} steps in {dt:.2f} s = {len(loader) / dt:.1f} steps/s, loss {loss:.4f}", flush=True)

# where the step goes: (i) sampling + block building alone, (ii) the engine alone, on ONE batch repeated (no allocator churn)
torch.cuda.synchronize(); t0 = time.perf_counter()
n_b = 0
for b in loader:
    n_b += 1
    if n_b == 1:
        last = b            # a FULL batch (the epoch's last one is the remainder)
    if n_b == min(50, len(loader)):
        break
torch.cuda.synchronize(); t_s = (time.perf_counter() - t0) / n_b
from glnn_amd import teacher
model.train()
eng = teacher.get_engine(model, opt)
input_nodes, output_nodes, blocks = last
for _ in range(3):
    eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
t_issue = (time.perf_counter() - t0) / 50            # host time to ISSUE a step (the queue is far from full after 50 steps)
torch.cuda.synchronize(); t_c = (time.perf_counter() - t0) / 50
print(f"per step: sampling + blocks {1e3 * t_s:.2f} ms (side stream, overlapped in the epochs above), fwd + loss + bwd + Adam "
      f"(TeacherEngine) {1e3 * t_c:.2f} ms of which host issue {1e3 * t_issue:.2f} ms; sources of that batch {input_nodes.numel()}", flush=True)
