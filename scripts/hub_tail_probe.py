"""What the hub rows cost a SHARD's short launches (before building a cross-workgroup split): the chunk launches of one rank
of the 8-rank products forward -- fused D=256 -> 256 -> 47, fused D=100 -> 256, stand-alone D=47 and D=100 -- on the real shard and on the same
shard with every row's degree capped at H (its first H in-edges kept).  The capped graph does the same work minus the hubs' edges: the time it
saves beyond its share of the edges is the tail the hubs cause.  usage: python scripts/hub_tail_probe.py [ranks=5,6,7]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
from glnn_amd.dist import RowShards
from glnn_amd.graph import CSRGraph
dev = "cuda:0"
ranks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5,6,7").split(",")]
g = data.make_graph("ogbn-products", seed=0, device=dev, scale=1.0)
n = g.n_dst
N = 8
bounds = RowShards.balanced_bounds(g.indptr, N)
deg_all = g.indptr[1:] - g.indptr[:-1]
print(f"graph: n {n} nnz {g.num_edges()} max degree {int(deg_all.max())}; rows with degree > 128 / 512 / 2048 / 8192: "
      f"{int((deg_all > 128).sum())} / {int((deg_all > 512).sum())} / {int((deg_all > 2048).sum())} / {int((deg_all > 8192).sum())}; "
      f"their edges: {int(deg_all[deg_all > 128].sum())} / {int(deg_all[deg_all > 512].sum())} / {int(deg_all[deg_all > 2048].sum())} / {int(deg_all[deg_all > 8192].sum())}", flush=True)
x256 = ops.as_feat(torch.randn(n, 256, device=dev).relu_())
x100 = ops.as_feat(torch.randn(n, 100, device=dev))
x47 = ops.as_feat(torch.randn(n, 47, device=dev))
w2 = torch.randn(256, 256, device=dev) / 16
w3 = torch.randn(47, 256, device=dev) / 16
w1 = torch.randn(256, 100, device=dev) / 10


def capped(shard, H):
    deg = shard.indptr[1:] - shard.indptr[:-1]
    keep_deg = deg.clamp(max=H)
    ip = torch.zeros(shard.n_dst + 1, dtype=torch.int64, device=dev)
    ip[1:] = keep_deg.cumsum(0)
    pos = torch.arange(int(shard.indptr[-1]), device=dev) - torch.repeat_interleave(shard.indptr[:-1], deg)
    idx = shard.indices[pos < torch.repeat_interleave(keep_deg, deg)].contiguous()
    return CSRGraph(ip, idx, shard.n_dst, shard.n_src)


def timed(fn, reps=9):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


for r in ranks:
    sh = RowShards(n, N, r, chunks=4, bounds=bounds)
    shard = g.row_range(sh.lo, sh.hi)
    for H in (None, 2048, 512):
        gg = shard if H is None else capped(shard, H)
        tot = {"fused256": 0.0, "fused100": 0.0, "spmm47": 0.0, "spmm100": 0.0}
        edges = 0
        for c in range(sh.chunks):
            off, nr = sh.chunk_rows(c)
            if nr == 0:
                continue
            ip = gg.indptr[off:off + nr + 1]
            edges += int(ip[-1] - ip[0])
            order = ops.fused_tile_order(ip, nr)
            xs256, xs100, xs47 = x256[sh.lo + off:sh.lo + off + nr], x100[sh.lo + off:sh.lo + off + nr], x47[sh.lo + off:sh.lo + off + nr]
            o47 = ops.feat_empty(nr, 47, dev); o256 = ops.feat_empty(nr, 256, dev); o100 = ops.feat_empty(nr, 100, dev)
            tot["fused256"] += timed(lambda: ops.sage_fused(ip, gg.indices, x256, nr, w2, relu=True, x_self=xs256, w_next=w3, out_next=o47, want_out=False, tile_order=order))
            tot["fused100"] += timed(lambda: ops.sage_fused(ip, gg.indices, x100, nr, w1, relu=True, x_self=xs100, out=o256, tile_order=order))
            tot["spmm47"] += timed(lambda: ops.spmm(ip, gg.indices, x47, nr, ops.AGG_SAGE_GCN, x_self=xs47, out=o47))
            tot["spmm100"] += timed(lambda: ops.spmm(ip, gg.indices, x100, nr, ops.AGG_SAGE_GCN, x_self=xs100, out=o100))
        print(f"rank {r} rows {sh.rows} cap {H}: edges {edges}  " + "  ".join(f"{k} {v:.3f} ms ({edges / v / 1e6:.2f} Ge/s)" for k, v in tot.items()), flush=True)
