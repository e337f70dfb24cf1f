"""Target of the L2 / fabric PMC passes (scripts/pmc_l2.sh): a calibration copy of known size (4 GiB read + 4 GiB written
with 16-byte-per-lane accesses by this library's move_rows_kernel), then two products-shaped teacher forwards (the three aggregation launches of bench.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops  # noqa: E402
from glnn_amd.graph import FullNeighborLoader  # noqa: E402
from glnn_amd.models import Model  # noqa: E402

dev = torch.device("cuda", 0)
# calibration launch of known size: move_rows_kernel copying a [4 Mi, 256] fp32 matrix row by row in natural order
# (4 GiB read + 32 MiB of row ids, 4 GiB written, 16 bytes per lane) -- 16x the Infinity Cache
src = torch.randn(1 << 22, 256, device=dev)
rows = torch.arange(1 << 22, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    ops.gather_rows(src, rows, out=dst)
torch.cuda.synchronize()
del src, dst, rows
torch.manual_seed(0)
g = data.make_graph("ogbn-products", seed=0, device=dev)
feats, _, _, _ = data.make_node_data("ogbn-products", seed=0, device=dev, n=g.n_dst)
feats = ops.as_feat(feats)
teacher = Model(dict(model_name="SAGE", num_layers=3, feat_dim=100, hidden_dim=256, label_dim=47, dropout_ratio=0.5, norm_type="batch", device=dev))
teacher.eval()
loader = FullNeighborLoader(g, 4096)
for _ in range(2):
    teacher.inference(loader, feats)
torch.cuda.synchronize()
print("done")
