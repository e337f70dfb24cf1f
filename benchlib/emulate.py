"""--emulate 2,4,8: the compute half of the multi-GPU scaling model, every rank of an N-rank job played on ONE GPU."""
import time

import numpy as np
import torch

from . import common as C
from .line import emit



def run_emulated(args, dev):
    """bench.py --emulate 2,4,8: the COMPUTE half of the multi-GPU scaling model, measured on one GPU.  For every world size N and
    every rank r the sharded forward of rank r runs here with dist.EmulatedPeers: the row ranges are RowShards.balanced_bounds',
    every collective is a local fill of the same bytes with the rows an unsharded forward produced (so rank r's output must equal
    the unsharded rows: checked), and ops' per-launch events give the rank's kernel time.  Three forms: all-gather exchange with
    the narrow / the wide layer 1 on the prescribed products-shaped graph, and the overlapped halo exchange on a clustered graph
    whose ids were shuffled and restored by the label-propagation partitioner."""
    from glnn_amd import data, ops
    from glnn_amd import dist as gdist
    from glnn_amd.graph import FullNeighborLoader
    from glnn_amd.models import Model
    worlds = [int(v) for v in args.emulate.split(",") if v]
    n_full = int(data.SHAPES[C.GRAPH]["n"] * args.scale)
    torch.manual_seed(0)
    teacher = Model(dict(model_name="SAGE", num_layers=3, feat_dim=C.SAGE_DIMS[0], hidden_dim=C.SAGE_DIMS[1], label_dim=C.SAGE_DIMS[-1], dropout_ratio=0.5,
                         norm_type="batch", device=dev))
    teacher.eval()
    enc = teacher.encoder
    steps, out = max(1, args.steps), {}
    forms = [("allgather-narrow", "products", dict(exchange="allgather", l1="narrow")), ("allgather-wide", "products", dict(exchange="allgather", l1="wide")),
             (f"allgather-mixed{args.mixed_fraction:g}", "products", dict(exchange="allgather", l1="mixed")), ("halo-lp", "clustered", dict(exchange="halo"))]
    if args.emulate_forms:
        forms = [f for f in forms if any(w in f[0] for w in args.emulate_forms.split(","))]
    graphs = {}
    for form, gkind, cfg in forms:
        if gkind not in graphs:
            graphs.clear()
            torch.cuda.empty_cache()
            if gkind == "products":
                g = data.make_graph(C.GRAPH, seed=0, device=dev, scale=args.scale)
                prep = None
            else:
                g0 = data.make_clustered_graph(n_full, 50.5, communities=64, p_in=0.95, seed=0, device=dev, shuffle_ids=True)
                t0 = time.perf_counter()
                perm = data.locality_order(g0, seed=0)
                g = data.relabel(g0, perm)
                torch.cuda.synchronize()
                prep = time.perf_counter() - t0
                del g0, perm
            feats = ops.as_feat(torch.randn(g.n_dst, C.SAGE_DIMS[0], device=dev))
            with torch.no_grad():
                truth, want = gdist.record_truth(enc, g, feats, ops)
                loader = FullNeighborLoader(g, 4096)
                for _ in range(2):
                    teacher.inference(loader, feats)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    teacher.inference(loader, feats)
                torch.cuda.synchronize()
                one_gpu_ms = 1e3 * (time.perf_counter() - t0) / steps
            graphs[gkind] = (g, feats, truth, want, one_gpu_ms, prep)
        g, feats, truth, want, one_gpu_ms, prep = graphs[gkind]
        n, nnz = g.n_dst, g.num_edges()
        res = {"graph": ("products-shaped power-law multigraph, random node order" if gkind == "products" else
                         "community-structured graph (64 communities, 0.95 of the edges inside), node ids shuffled, then renumbered by data.locality_order"),
               "nodes": n, "nnz": nnz, "one_gpu_forward_ms": one_gpu_ms, "partition_seconds": prep, "worlds": {}}
        for N in worlds:
            bounds = gdist.RowShards.balanced_bounds(g.indptr, N)
            ranks = []
            for r in range(N):
                sh = gdist.RowShards(n, N, r, chunks=args.chunks or 4, bounds=bounds)
                peers = gdist.EmulatedPeers(N, r, truth=truth, full_graph=g if cfg["exchange"] == "halo" else None)
                shard = g.row_range(sh.lo, sh.hi)
                if cfg["exchange"] == "halo":
                    t = gdist.HaloShardedTeacher(enc, shard, sh, ops, group=peers, overlap=True)
                else:
                    t = gdist.ShardedTeacher(enc, shard, sh, ops, group=peers, widening_exchange=cfg["l1"], mixed_fraction=args.mixed_fraction)
                with torch.no_grad():
                    for _ in range(2):
                        t.forward(feats)                   # warm-up (buffers, relabelled columns, packed weights, hub plans, first launches)
                    timing, peers.events = [], []
                    if hasattr(t, "chunk_ready"):
                        t.chunk_ready = []                 # (one-launch layers: when every chunk's completion signal fired, relative to the launch)
                    gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
                    torch.cuda.synchronize()
                    ops.set_timing(timing)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        y = t.forward(feats)
                    torch.cuda.synchronize()
                    wall = 1e3 * (time.perf_counter() - t0) / steps
                    ops.set_timing(None)
                kms = C.kernel_breakdown_median(timing, steps)
                fill = sum(s_.elapsed_time(e_) for _, _, s_, e_ in peers.events) / steps
                c = want.shape[1]
                diff = float((y[:, :c] - want[sh.lo:sh.hi, :c]).abs().max()) if sh.rows else 0.0
                # per one-launch layer in forward order: the ms after the launch began at which chunk 0, 1, .. became sendable, then the launch's end
                launches, cur_ = [], []
                for k_, s_, e_ in (getattr(t, "chunk_ready", None) or []):
                    cur_.append(s_.elapsed_time(e_))
                    if k_ == -1:
                        launches.append(cur_)
                        cur_ = []
                n_ol = len(launches) // max(1, steps)
                signals = [[round(float(np.median([launches[s_ * n_ol + i][j] for s_ in range(steps)])), 3) for j in range(len(launches[i]))]
                           for i in range(n_ol)] if n_ol else None
                ranks.append({"rank": r, "rows": sh.rows, "chunk_signal_ms": signals, "nnz": int(shard.num_edges()), "wall_ms": wall, "kernel_ms": sum(kms.values()), "fill_ms": fill,
                              "GB_received": 4e-9 * gdist.EXCHANGE_STATS["floats_received"] / steps, "kernels": kms,
                              "halo_rows": getattr(getattr(t, "plan", None), "n_halo", None), "max_abs_diff_vs_unsharded": diff})
                del t, peers, shard, y
                torch.cuda.empty_cache()
            kmax = max(x_["kernel_ms"] for x_ in ranks)
            gb = max(x_["GB_received"] for x_ in ranks)
            # per link: an all-gather's count includes the own slab (N slabs, N-1 of them arrive, one per link); a halo exchange's does not
            per_link = gb / N if cfg["exchange"] == "allgather" else gb / max(1, N - 1)
            link = [1e3 * per_link / rt for rt in C.XGMI_LINK_GBS]                 # the N-1 peers send over N-1 links in parallel
            res["worlds"][str(N)] = {
                "max_kernel_ms": kmax, "mean_kernel_ms": float(np.mean([x_["kernel_ms"] for x_ in ranks])), "max_GB_received_per_rank": gb,
                "modelled_link_ms": link, "forward_ms_exchange_hidden": max(kmax, link[1]), "forward_ms_exchange_exposed": kmax + link[0],
                "speedup_vs_one_gpu": [one_gpu_ms / (kmax + link[0]), one_gpu_ms / max(kmax, link[1])],
                "verified": all(x_["max_abs_diff_vs_unsharded"] <= 1e-4 for x_ in ranks), "ranks": ranks}
        out[form] = res
    result = {"metric": "per-rank kernel time of the N-rank sharded teacher forward, every rank emulated on ONE GPU (compute half of the scaling model)",
              "value": None, "unit": "ms", "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": None, "higher_is_better": False, "scaling": "strong",
              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"{C.GRAPH}-shaped SAGE teacher forward, ranks of N = {worlds} emulated (dist.EmulatedPeers, truth fills)", "scale": args.scale,
                         "link_GBps_assumed": list(C.XGMI_LINK_GBS)},
              "verified": all(w["verified"] for f in out.values() for w in f["worlds"].values()),
              "scale_model": out}
    emit(result, args)
