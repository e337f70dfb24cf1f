"""Round 6 fuzzer of the chunked ONE-LAUNCH forms of the sharded teacher forward (GPU box): random graph sizes, layer widths, world sizes, chunk
counts, exchange forms and ranks; every case = one emulated rank (dist.EmulatedPeers with the rows of an unsharded forward) whose output must equal the
unsharded rows (<= 1e-5) and must be the SAME BITS with dist.ONE_LAUNCH on and off.
  python scripts/fuzz_sharded.py SEED [CASES]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd import dist as gdist
from glnn_amd.models import Model

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rs = np.random.RandomState(seed)
dev = torch.device("cuda", 0)
worst, bad = 0.0, []
for case in range(cases):
    scale = float(rs.choice([0.004, 0.01, 0.03, 0.08]))
    dims = [int(rs.choice([100, 128, 64, 47])), int(rs.choice([256, 128, 200])), None, int(rs.choice([47, 40, 7]))]
    dims[2] = dims[1]
    world = int(rs.choice([2, 3, 4, 8]))
    chunks = int(rs.choice([1, 2, 3, 4, 8]))
    form = str(rs.choice(["narrow", "wide", "mixed"]))
    frac = float(rs.choice([0.25, 0.5, 0.75]))
    rank = int(rs.randint(world))
    balanced = bool(rs.randint(2))
    torch.manual_seed(seed * 1000 + case)
    g = data.make_graph("ogbn-products", seed=case, device=dev, scale=scale)
    n = g.n_dst
    model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[3], dropout_ratio=0.5, norm_type="batch", device=dev))
    with torch.no_grad():
        for bn in model.encoder.norms:
            bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
    model.eval()
    x = ops.as_feat(torch.randn(n, dims[0], device=dev))
    outs = []
    with torch.no_grad():
        truth, want = gdist.record_truth(model.encoder, g, x, ops)
        bounds = gdist.RowShards.balanced_bounds(g.indptr, world) if balanced else None
        sh = gdist.RowShards(n, world, rank, chunks=chunks, bounds=bounds)
        for one in (True, False):
            gdist.ONE_LAUNCH = one
            t = gdist.ShardedTeacher(model.encoder, g.row_range(sh.lo, sh.hi), sh, ops, group=gdist.EmulatedPeers(world, rank, truth=truth),
                                     widening_exchange=form, mixed_fraction=frac)
            y = t.forward(x)
            y2 = t.forward(x)                      # (a second forward: the next epoch of the signals, counters back at zero)
            torch.cuda.synchronize()
            outs.append((y.clone(), bool(torch.equal(y, y2))))
            del t
        gdist.ONE_LAUNCH = True
    c = want.shape[1]
    err = float((outs[0][0][:, :c] - want[sh.lo:sh.hi, :c]).abs().max()) if sh.rows else 0.0
    same = bool(torch.equal(outs[0][0], outs[1][0])) and outs[0][1] and outs[1][1]
    worst = max(worst, err)
    if err > 1e-5 or not same:
        bad.append((case, n, dims, world, rank, chunks, form, frac, balanced, err, same))
    del g, x, truth, want, outs, model
    torch.cuda.empty_cache()
print(f"fuzz_sharded seed {seed}: {cases} cases, worst {worst:.2e}; BAD: {bad}" if bad else f"fuzz_sharded seed {seed}: {cases} cases, worst {worst:.2e}; not ok: []")
sys.exit(1 if bad else 0)
