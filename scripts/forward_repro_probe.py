"""Cross-process reproducibility of the teacher forward and the student step: sha256 of SAGE.inference's output on the arxiv-shaped graph and a
0.25-scale products-shaped one, and of an MLP3w4 / MLP3w8 student's parameters after five steps.  Run in several processes and compare."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.graph import FullNeighborLoader
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"


def sha(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


for name, scale, dims in (("ogbn-arxiv", 1.0, (128, 40)), ("ogbn-products", 0.25, (100, 47))):
    torch.manual_seed(0)
    g = data.make_graph(name, seed=0, device=dev, scale=scale)
    feats, labels, _, _ = data.make_node_data(name, seed=0, device=dev, n=g.n_dst)
    m = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=256, label_dim=dims[1], dropout_ratio=0.5, norm_type="batch", device=dev))
    m.eval()
    out = m.inference(FullNeighborLoader(g, 4096), ops.as_feat(feats))
    print(f"teacher forward {name} x{scale}: graph {sha([g.indptr, g.indices])} feats {sha([feats])} out {sha([out])}", flush=True)
for d, B, p in (([128, 1024, 1024, 40], 512, 0.5), ([100, 2048, 2048, 47], 4096, 0.2)):
    torch.manual_seed(1)
    s = Model(dict(model_name="MLP", num_layers=3, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1], dropout_ratio=p, norm_type="batch", device=dev))
    s.train()
    opt = torch.optim.Adam(s.parameters(), lr=0.01)
    x = ops.as_feat(torch.randn(8 * B, d[0], device=dev))
    t = ops.as_feat(torch.log_softmax(torch.randn(8 * B, d[-1], device=dev), 1))
    eng = StudentEngine(s, opt, B)
    perm = torch.randperm(8 * B).view(-1, B).to(dev)
    for i in range(5):
        eng.step(x, perm[i], ops.LOSS_KL, t, 1.0)
    torch.cuda.synchronize()
    print(f"student {'-'.join(map(str, d))} B={B}: params {sha(list(s.state_dict().values()))} loss {float(eng.loss_out):.8f}", flush=True)
