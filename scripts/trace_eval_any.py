import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from glnn_amd import data, train_and_eval as te
from glnn_amd.models import Model
from glnn_amd.graph import FullNeighborLoader
dev = "cuda:0"
name, n, deg, f, h, c, norm = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
torch.manual_seed(0)
g = data.make_uniform_graph(n, deg, seed=0, device=dev)
feats = torch.randn(n, f, device=dev); labels = torch.randint(0, c, (n,), device=dev)
model = Model(dict(model_name=name, num_layers=2 if name != "MLP" else 3, feat_dim=f, hidden_dim=h, label_dim=c, dropout_ratio=0.5, norm_type=norm, device=dev))
model.eval()
crit = torch.nn.NLLLoss(); ev = lambda o, l: o.argmax(1).eq(l).float().mean().item()
if name == "SAGE":
    data_arg = FullNeighborLoader(g, 4096); fn = lambda: te.evaluate(model, data_arg, feats, labels, crit, ev)
elif name == "GCN":
    fn = lambda: te.evaluate(model, g, feats, labels, crit, ev)
else:
    fn = lambda: te.evaluate_mini_batch(model, feats, labels, crit, 4096, ev)
for _ in range(2): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fn()
torch.cuda.synchronize(); print(f"evaluate {' '.join(sys.argv[1:])}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
