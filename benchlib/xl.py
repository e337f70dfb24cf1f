"""--workload xl: BASELINE.json configs[4], one rank's 3-layer forward of the synthetic 100M-node / 2B-edge graph (peers emulated at N = 1)."""
import json
import os
import time

import torch

from . import common as C
from .checker import CheckedBackend
from .line import emit



def run_xl(args, rank, world, dev, barrier):
    """BASELINE.json configs[4] as a TEACHER FORWARD (reference models.py:121-148: L layers of aggregate -> project -> BatchNorm(eval)
    -> ReLU): synthetic 100M-node / 2B-edge graph over 8 GPUs = 12.5M destination rows and 250M in-edges per GPU (generated on the
    device, never crossing PCIe), 128-d fp32 features replicated (static layer-1 input), 3-layer SAGE 128-256-256-47 through
    glnn_amd.dist.ShardedTeacher: layer 1 aggregates at D=128 and exchanges the narrow aggregate (every rank projects all rows
    itself) or its 256-wide output (--layer1-exchange wide); layer 2 is the fused aggregate + project kernel at D=256 chained
    into layer 3's 256->47 projection; layer 3 exchanges the 47-wide rows and aggregates them.  Weak scaling: per-GPU work is
    fixed, value = total edges/s.  N = 1 plays ONE rank of the --xl-shards-way run (dist.EmulatedPeers): every collective is a
    local fill of the same bytes into the same slots -- the peers' rows are copies of this rank's own slab (the unsharded forward
    needs 8 GPUs), so every kernel gathers over the full 100M-row buffers with the real run's (absent) locality; the fills stand
    where the xGMI transfers would and are reported separately."""
    import torch.distributed as dist
    from glnn_amd import data, ops
    from glnn_amd import dist as gdist
    from glnn_amd.models import Model
    rows, deg, dims = int(12_500_000 * args.scale), 20, C.XL_DIMS
    shards_n = world if world > 1 else max(1, args.xl_shards)      # N = 1: ONE rank's shard of the xl_shards-way run
    me = rank if world > 1 else (args.emulate_rank if args.emulate_rank is not None else shards_n // 2)
    n_total = rows * shards_n
    g = data.make_xl_shard(rows, deg, n_total, seed=1000 + me, device=dev)
    nnz = g.num_edges()
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242)                                          # the replicated input: identical on every rank
    x = torch.empty(n_total, dims[0], device=dev)                  # 51.2 GB at 8 x 12.5M rows, filled in slabs
    for s0 in range(0, n_total, 1 << 23):
        x[s0:s0 + (1 << 23)].normal_(generator=gen)
    torch.manual_seed(0)
    teacher = Model(dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.5,
                         norm_type="batch", device=dev))
    teacher.eval()
    sh = gdist.RowShards(n_total, shards_n, me, chunks=args.xl_chunks)
    peers = gdist.EmulatedPeers(shards_n, me) if world == 1 and shards_n > 1 else None
    sharded = gdist.ShardedTeacher(teacher.encoder, g, sh, ops, group=peers, widening_exchange=args.layer1_exchange)

    def step():
        with torch.no_grad():
            return sharded.forward(x)

    for _ in range(max(1, args.warmup)):
        step()
    timing = []
    barrier()
    if rank == 0:
        ops.set_timing(timing)
        if peers is not None:
            peers.events = []
    gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_timing(None)
    exch_gb = 4e-9 * gdist.EXCHANGE_STATS["floats_received"] / args.steps
    n_coll = gdist.EXCHANGE_STATS["collectives"] / args.steps
    fills = peers.events if peers is not None else []
    if peers is not None:
        peers.events = None
    verify = None
    if not args.no_verify:                   # one more forward through the checking proxy (outside the timed region)
        chk = CheckedBackend(ops, conservation=True)
        sharded.be = chk
        out_c = step()
        sharded.be = ops
        same = bool(torch.equal(out_c, out))
        finite = bool(torch.isfinite(out).all())
        t = torch.tensor([0.0 if (chk.ok and same and finite) else 1.0], device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        verify = {"ok": float(t.item()) == 0.0, "tolerance": chk.tol, "rows_sampled_per_launch": chk.sample, "launches": chk.report,
                  "repeat_forward_bit_equal": same, "finite": finite,
                  "what": "every launch of one extra forward: a sample of its rows recomputed in torch fp64 from the launch's own inputs, stand-alone "
                          "aggregations also re-launched as a row range (bit-equal); max over ranks"}
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        dist.destroy_process_group()
        return
    torch.cuda.synchronize()
    ms = C.kernel_breakdown(timing, args.steps)
    fill_ms = {}
    for tag, nbytes, s_, e_ in fills:
        k = f"{tag[0]}{tag[1]}"
        fill_ms[k] = fill_ms.get(k, 0.0) + s_.elapsed_time(e_) / args.steps
    d0, d1, d2, c = dims
    wide = args.layer1_exchange == "wide"
    layers = []

    def xl_traffic(pmc_key):
        """HBM bytes per forward of one layer's launches: a constant from the committed PMC passes of this command (profiles/pmc_traffic_xl.json,
        scripts/pmc_xl.sh: per-launch mean x the chunk launches), valid for the full-size shard with 4 chunks only; else None."""
        if args.scale != 1.0 or sh.chunks != 4 or shards_n != 8:
            return None
        try:
            with open(os.path.join(C.ROOT, "profiles", "pmc_traffic_xl.json")) as f:
                return 4 * json.load(f)["per_launch_bytes"][pmc_key]["total"]
        except Exception:
            return None

    def agg_layer(name, key, d, d_written, what, pmc_key=None):
        t = ms.get(key)
        if t is None:
            return
        b = C.alg_bytes(nnz, rows, d, d_written)
        layers.append({"layer": name, "kernel": what, "bound": "hbm", "ms": t, "alg_GB": b / 1e9, "achieved": b / t / 1e6, "peak": C.HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": b / t / 1e6 / C.HBM_PEAK_GBS, "Gedges_per_s": nnz / t / 1e6,
                       "traffic": xl_traffic(pmc_key) if pmc_key else None})

    def gemm_layer(name, key, m, k, n):
        t = ms.get(key)
        if t is None:
            return
        fl = 2.0 * m * k * n
        layers.append({"layer": name, "kernel": key, "bound": "mfma", "ms": t, "GFLOP": fl / 1e9, "achieved": fl / t / 1e9, "peak": 157.3,
                       "unit": "TFLOP/s", "frac": fl / t / 1e9 / 157.3})

    def exchange_layer(name, key, width):
        gb = 4e-9 * sh.n_pad * ((width + 3) // 4 * 4) * (shards_n - 1) / shards_n
        link = [1e3 * gb / (shards_n - 1) / r for r in C.XGMI_LINK_GBS] if shards_n > 1 else [0.0, 0.0]
        layers.append({"layer": name, "kernel": "all-gather over xGMI" + (" (EMULATED: local fill of the same bytes)" if peers is not None else ""),
                       "GB_received_per_rank": gb, "emulated_fill_ms": fill_ms.get(key), "modelled_link_ms": link,
                       "model": f"{shards_n - 1} peers on {shards_n - 1} links in parallel at {C.XGMI_LINK_GBS[0]:.0f}-{C.XGMI_LINK_GBS[1]:.0f} GB/s each; chunked x{sh.chunks}: "
                                "all but the first chunk can hide under the producing kernel"})

    if wide:
        agg_layer("1 aggregate+project (own rows)", f"sage_fused d={d0}->{d1}", d0, d1, f"sage_fused_kernel<LPR={C.lanes_per_row(d0)}>")
        exchange_layer("1 exchange (256-wide output)", "y0", d1)
    else:
        agg_layer("1 aggregate (own rows)", f"spmm d={d0}", d0, None, f"spmm_csr_kernel<LPR={C.lanes_per_row(d0)},U={C.SPMM_U},SAGE_GCN>",
                  f"spmm_csr_kernel<LPR={C.lanes_per_row(d0)},U={C.SPMM_U},SAGE_GCN>")
        exchange_layer("1 exchange (128-wide aggregate)", "agg0", d0)
        gemm_layer("1 projection (replicated: ALL rows on every rank)", f"gemm k={d0} n={d1}", sh.n_pad, d0, d1)
    agg_layer("2 aggregate+project+chained 256->47 (own rows)", f"sage_fused d={d1}->{d2}->{c}", d1, c,
              f"sage_fused_kernel<LPR=64> (only the 47 chained floats per row are written)", f"sage_fused_kernel<LPR=64,U={C.SPMM_U}>")
    exchange_layer("3 exchange (47-wide projected rows)", "hw2", c)
    agg_layer("3 aggregate (own rows)", f"spmm d={c}", c, None, f"spmm_csr_kernel<LPR={C.lanes_per_row(c)},U={C.SPMM_U},SAGE_GCN>",
              f"spmm_csr_kernel<LPR={C.lanes_per_row(c)},U={C.SPMM_U},SAGE_GCN>")
    kernel_ms = sum(ms.values())
    fill_total = sum(fill_ms.values())
    dom = max((l for l in layers if l.get("bound") == "hbm"), key=lambda l: l["ms"])
    result = {
        "metric": "aggregated edges/sec, 3-layer SAGE teacher forward, synthetic 100M-node / 2B-edge graph (BASELINE configs[4])",
        "value": world * 3 * nnz * args.steps / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "verified": None if verify is None else verify["ok"],
        "config": {"workload": f"synthetic-XL teacher forward: SAGE {'-'.join(map(str, dims))} (BN eval), one rank's shard of a uniform random multigraph "
                               "whose sources are drawn over ALL nodes_total rows", "rows_per_gpu": rows, "nnz_per_gpu": nnz, "nodes_total": n_total,
                   "shards": shards_n, "rank_timed": me, "chunks": sh.chunks, "layer1_exchange": args.layer1_exchange,
                   "input_GB": 4e-9 * n_total * d0, "hidden_GB": 4e-9 * sh.n_pad * d1,
                   "parallelism": f"row shards x{shards_n}, ShardedTeacher" + (f"; N = 1: rank {me} with EMULATED peers (collectives = local fills, peers' rows = "
                                                                              "copies of the own slab)" if peers is not None else ", RCCL all-gathers")},
        "per_forward": {"wall_ms": 1e3 * dt / args.steps, "kernel_ms": kernel_ms, "emulated_fill_ms": fill_total if peers is not None else None,
                        "GB_received_per_rank": exch_gb, "collectives": n_coll, "kernels": ms},
        "layers": layers,
        "roofline": {"bound": "hbm", "kernel": f"{dom['kernel']} (layer {dom['layer']})", "achieved": dom["achieved"], "peak": C.HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["frac"], "traffic": dom.get("traffic"), "algorithmic_bytes_per_launch_sum": dom["alg_GB"] * 1e9, "avg_ms_per_forward": dom["ms"],
                     "traffic_source": None if dom.get("traffic") is None else "static: profiles/pmc_traffic_xl.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"},
        "verify": verify,
    }
    emit(result, args)
    if world > 1:
        dist.destroy_process_group()
