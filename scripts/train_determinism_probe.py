"""Is a sampled-block training epoch reproducible?  Two runs from the same initial model and the same loader seed, in ONE process:
per-step losses and final parameters compared bit for bit, with the batches built one ahead on the side stream (prefetch), by the loader's
worker thread, and in line.
python scripts/train_determinism_probe.py [ogbn-products|ogbn-arxiv] [steps]"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, teacher
from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
from glnn_amd.models import Model
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "ogbn-products"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = {"ogbn-arxiv": (128, 40, 512, 0.2), "ogbn-products": (100, 47, 4096, 0.5)}[name]
torch.manual_seed(0)
g = data.make_graph(name, seed=0, device=dev)
n = g.n_dst
feats, labels, _, _ = data.make_node_data(name, seed=0, device=dev, n=n)
base = Model(dict(model_name="SAGE", num_layers=3, feat_dim=cfg[0], hidden_dim=256, label_dim=cfg[1], dropout_ratio=cfg[3], norm_type="batch", device=dev))
idx_train = torch.randperm(n)[: cfg[2] * steps].to(dev)


def run(prefetch, threaded=False, global_first=False):
    model = copy.deepcopy(base)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=0.003)
    loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=cfg[2], shuffle=False, drop_last=False, seed=1234)
    loader.prefetch, loader.threaded, loader.global_first_block = prefetch, threaded, global_first
    eng = teacher.get_engine(model, opt)
    losses, nsrc = [], []
    for input_nodes, output_nodes, blocks in loader:
        eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
        losses.append(eng.loss_out.clone())
        nsrc.append(None if input_nodes is None else (input_nodes.numel(), int(input_nodes.sum()), [int(b.indices.to(torch.int64).sum()) for b in blocks]))
    eng.sync_optimizer_state()
    torch.cuda.synchronize()
    return torch.stack(losses).flatten().cpu(), [v.detach().clone() for v in model.state_dict().values()], nsrc


ref = run(False)
for tag, pf, th, gf in (("no prefetch again", False, False, False), ("prefetch", True, False, False), ("worker thread", True, True, False),
                        ("worker thread again", True, True, False), ("global-id outermost block", True, True, True)):
    got = run(pf, th, gf)
    same_blocks = True if gf else got[2] == ref[2]
    same_loss = torch.equal(got[0], ref[0])
    same_par = all(torch.equal(a, b) for a, b in zip(got[1], ref[1]))
    first = next((i for i in range(len(ref[0])) if got[0][i] != ref[0][i]), None)
    print(f"{tag}: blocks equal {same_blocks}, losses equal {same_loss} (first differing step {first}), parameters equal {same_par}", flush=True)
    if not same_loss:
        print("   ", [f"{float(a):.7f}/{float(b):.7f}" for a, b in zip(ref[0][:6], got[0][:6])])
import hashlib
h = hashlib.sha256()
for t in ref[1]:
    h.update(t.cpu().numpy().tobytes())
print("sha256 of the final parameters:", h.hexdigest()[:16], " losses:", [f"{float(v):.7f}" for v in ref[0][:4]], " blocks:", [b[:2] for b in ref[2][:4]], " idx_train", int(idx_train.sum()), int(idx_train[:4096].sum()), flush=True)
