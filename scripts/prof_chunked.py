"""Host-side profile of the chunked sweep (SAGE.inference(whole_graph=False), engine mode): where the ~0.1 ms per chunk goes.
  python scripts/prof_chunked.py [scale]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.graph import FullNeighborLoader
from glnn_amd.models import Model
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
dev = torch.device("cuda", 0)
g = data.make_graph("ogbn-products", seed=0, device=dev, scale=scale)
x = ops.as_feat(torch.randn(g.n_dst, 100, device=dev))
m = Model(dict(model_name="SAGE", num_layers=3, feat_dim=100, hidden_dim=256, label_dim=47, dropout_ratio=0.5, norm_type="batch", device=dev)).eval()
ld = FullNeighborLoader(g, 4096)
for _ in range(2):
    m.encoder.inference(ld, x, whole_graph=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    m.encoder.inference(ld, x, whole_graph=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"chunked sweep: {1e3 * dt:.1f} ms for {3 * len(ld)} chunk launches = {1e6 * dt / (3 * len(ld)):.1f} us per chunk")
pr = cProfile.Profile()
pr.enable()
m.encoder.inference(ld, x, whole_graph=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
