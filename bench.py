#!/usr/bin/env python
"""Contract benchmark: BASELINE.json's metric -- aggregated edges/sec of the GLNN teacher forward
(SAGE layer-wise full-neighbour inference, reference models.py:121-148) plus student distillation
steps/sec (loop body of reference train_and_eval.py:74-85) -- on ogbn-products-SHAPED synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full teacher forward over the products-shaped graph (3 layers, every edge aggregated once
per layer = 371,154,840 edge aggregations); `value` = aggregated edges / s over the whole job, inputs
resident in HBM.  N > 1: strong scaling, the graph's destination rows are range-sharded over the ranks and
one all-gather per layer exchanges activations (glnn_amd/dist.py).  The student leg times the fused
distillation step of the products student MLP3w8 (100-2048-2048-47, B=4096, KL soft-label pass) under the
same protocol and is reported in the "student" object.

Extra objects on the JSON line (rank 0, N = 1): "roofline" for the dominant kernel (the aggregation), from
per-launch HIP events recorded inside the timed region on the launch stream; "roofline_reordered": the same kernels on
the same graph with its nodes renumbered by descending in-degree (SURVEY 8d allows a locality-ordered figure beside the
random-order one; `value` stays the random-order number); and "cpu_baseline" (SURVEY 8d / BASELINE.md section 3): the
teacher forward on the host cores -- this repo's OpenMP C restatement ("port": the reference's own dgl CPU path cannot
run, dgl is not installed) with torch.sparse_csr @ X beside it as a second opinion, on the full-size graph of the metric -- and
the student step as the SAME SEQUENCE OF
PyTorch CPU OPS the reference issues (nn.Linear / BatchNorm1d / relu / dropout / log_softmax / KLDivLoss / Adam,
reference train_and_eval.py:74-85), thread count stated."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X spec HBM3E bandwidth (MI355X_MICROARCH.md)
SAGE_DIMS = [100, 256, 256, 47]  # reference train.conf.yaml:196-204 (ogbn-products SAGE, hidden 256, BN)
STUDENT = dict(name="MLP3w8", dims=[100, 2048, 2048, 47], batch=4096, dropout=0.2, lr=0.01, wd=0.0)   # :187-194
GRAPH = "ogbn-products"
CPU_SAMPLE_SCALE = 1.0            # the CPU baseline runs the metric's own configuration (full-size graph): ~15 s per forward on 128 threads
SPMM_U = 8                        # in-flight gathers per lane group: GLNN_SPMM_U / GLNN_FUSED_U of csrc/spmm.hip
# --workload arxiv = BASELINE.json configs[1] + [2]: ogbn-arxiv-shaped SAGE teacher forward (train.conf.yaml:170-177) and the
# MLP3w4 student the reference's experiments/glnn_arxiv.sh uses (:149-154).  Features (87 MB) fit the 256 MB Infinity Cache,
# so the HBM fraction of its roofline object is not meaningful -- edges/s is the number.
ARXIV = dict(graph="ogbn-arxiv", sage_dims=[128, 256, 256, 40], cpu_sample_scale=1.0,
             student=dict(name="MLP3w4", dims=[128, 1024, 1024, 40], batch=512, dropout=0.5, lr=0.01, wd=0.0))


def agg_width(d_in, d_out):
    return d_out if d_in > d_out else d_in      # project-first when the layer narrows


def lanes_per_row(d):
    dv = (d + 3) // 4
    return 4 if dv <= 4 else 8 if dv <= 8 else 16 if dv <= 16 else 32 if dv <= 32 else 64


PMC_FILE = os.path.join("profiles", "pmc_traffic.json")


def pmc_traffic(kernel):
    """HBM bytes per launch of the named kernel instantiation.  NOT measured by this run: a constant read from the
    committed rocprofv3 PMC passes of the same command (profiles/pmc_traffic.json: FETCH_SIZE x2 (gfx950 correction) +
    WRITE_SIZE, separate --pmc passes); None if absent.  The line says so in roofline.traffic_source."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)["per_launch_bytes"][kernel]["total"]
    except Exception:
        return None


def alg_bytes(nnz, n_dst, d, d_out=None):
    """per edge one gathered fp32 source row + one int32 index; per dst row one self-row read,
    one output-row write (d_out wide for the fused aggregate+project kernel), one int64 indptr entry."""
    d_out = d if d_out is None else d_out
    return nnz * (4 * d + 4) + n_dst * (4 * d + 4 * d_out + 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed teacher forwards (default 100: a >= 3 s timed region at ~33 ms each; "
                    "--workload xl: 10 rank-forwards of ~170 ms; --emulate: 3 per emulated rank)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "halo"],
                    help="N > 1: per-layer all-gather of every rank's rows (default) or halo exchange of only the referenced remote rows")
    ap.add_argument("--locality", type=float, default=0.0,
                    help="0 = the prescribed products-shaped generator (no locality); p in (0,1] = a community-structured graph of the same "
                         "size (64 communities, a fraction p of the edges inside them): what the halo exchange is for")
    ap.add_argument("--reorder", default="degree", choices=["degree", "none"],
                    help="N = 1: also time the forward on the graph renumbered by descending in-degree -> roofline_reordered")
    ap.add_argument("--reorder-steps", type=int, default=10)
    ap.add_argument("--xl-shards", type=int, default=8, help="--workload xl at N = 1: the number of shards of the full graph (the single "
                    "rank holds ONE shard's rows and gathers from all xl-shards x 12.5M source rows, as a rank of the real run does)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; 1.0 = the metric's config)")
    ap.add_argument("--student-steps-per-step", type=int, default=30, help="student steps timed per --steps unit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the sampled-block teacher-training object (N = 1)")
    ap.add_argument("--no-grad-overlap", action="store_true",
                    help="N > 1: one gradient all-reduce after the whole backward instead of starting the big layers' all-reduce "
                         "from inside it (glnn_amd.dist.OverlappedGradSync)")
    ap.add_argument("--student-global-bn", action="store_true",
                    help="N > 1: take the student's BatchNorm batch statistics over the global (N x B rows) batch through the "
                         "exchange hook, i.e. exactly the single-GPU step on that batch (default: per-rank statistics)")
    ap.add_argument("--no-halo-overlap", action="store_true",
                    help="--exchange halo: the synchronous form (default: every all-to-all is asynchronous and hidden behind own-row work)")
    ap.add_argument("--shuffle-ids", action="store_true", help="--locality p: randomly permute the node ids of the clustered graph")
    ap.add_argument("--partition", default="none", choices=["none", "lp"],
                    help="N > 1: renumber the nodes with glnn_amd.data.locality_order (label propagation) before the row ranges are cut")
    ap.add_argument("--layer1-exchange", default="auto", choices=["auto", "narrow", "wide"],
                    help="N > 1, all-gather exchange: what the widening first layer (100 -> 256) puts on the wire -- its 100-wide aggregate "
                         "(every rank projects all rows itself) or its 256-wide fused output (no replicated work, 2.56x the bytes).  auto (default): "
                         "both forms are timed on this job's transport before the timed region (3 forwards each) and the faster one runs; "
                         "--workload xl and N = 1 treat auto as narrow")
    ap.add_argument("--chunks", type=int, default=0, help="N > 1 / --emulate: chunks of the overlapped all-gather exchange (0 = 4, or chosen by the "
                    "layer-1 autotune among 2, 4, 8)")
    ap.add_argument("--no-verify", action="store_true", help="skip the self-check of the timed output ('verified' on the JSON line)")
    ap.add_argument("--no-clustered-leg", action="store_true", help="N = 1: skip roofline_clustered (the forward on a graph with communities)")
    ap.add_argument("--no-small-students", action="store_true", help="N = 1: skip students_small (the B = 512 arxiv students' step times)")
    ap.add_argument("--workload", default="products", choices=["products", "arxiv", "xl"],
                    help="products = the metric's config (default); xl = BASELINE.json configs[4]: 12.5M-node / 250M-edge shard per GPU "
                         "of a 100M-node / 2B-edge synthetic graph, 128-d features, the 3-layer SAGE teacher forward 128-256-256-47 through "
                         "glnn_amd.dist.ShardedTeacher (weak scaling); N = 1 times ONE rank's shard of the 8-rank run with the peers emulated")
    ap.add_argument("--xl-chunks", type=int, default=4, help="--workload xl: chunks of the overlapped exchange (RowShards.chunks)")
    ap.add_argument("--emulate-rank", type=int, default=None, help="--workload xl at N = 1: which rank of the --xl-shards-way run to play (default: the middle one)")
    ap.add_argument("--emulate", type=str, default=None,
                    help="N = 1, products: comma-separated world sizes (e.g. 2,4,8).  Times EVERY rank's shard of the N-rank sharded forward on this "
                         "one GPU (glnn_amd.dist.EmulatedPeers: each collective replaced by a local fill of the same bytes with the rows an unsharded "
                         "forward produced), for the all-gather exchange (narrow and wide layer 1) and the halo exchange on a clustered graph with "
                         "shuffled ids re-partitioned by label propagation: the compute half of DESIGN.md section 6's scaling model, measured")
    ap.add_argument("--no-xl-leg", action="store_true", help="N = 1, products at scale 1.0: skip the 'xl' object (BASELINE configs[4]: one rank-forward "
                    "of the synthetic 100M-node / 2B-edge graph, run as a child process of this one after the products legs)")
    ap.add_argument("--no-arxiv-leg", action="store_true", help="N = 1, products at scale 1.0: skip the 'arxiv' object (BASELINE configs[1] + [2])")
    ap.add_argument("--detail-file", default=None, help="also write the long detail object to this file (default: gpurun_out/bench_detail.json when that directory exists)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 3 if args.emulate else (10 if args.workload == "xl" else 100)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)          # `python bench.py --gpus N` by itself: spawn the N ranks (one per GPU) and relay their line
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    from glnn_amd import data, ops
    from glnn_amd.dist import HaloShardedTeacher, OverlappedGradSync, RowShards, ShardedTeacher, make_grad_sync
    from glnn_amd.graph import FullNeighborLoader
    from glnn_amd.models import Model
    from glnn_amd.student import StudentEngine

    # test-only knobs: GLNN_SINGLE_DEVICE=1 maps every rank to cuda:0 and GLNN_DIST_BACKEND=gloo swaps the transport, so
    # that the N > 1 code path can be smoke-tested on a 1-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("GLNN_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GLNN_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "xl":
        if args.layer1_exchange == "auto":      # (link-bound at this size by every estimate: the narrow aggregate)
            args.layer1_exchange = "narrow"
        return run_xl(args, rank, world, dev, barrier)
    if args.emulate:
        if world != 1:
            raise SystemExit("--emulate runs on ONE GPU (it plays the ranks of an N-rank job one after the other)")
        return run_emulated(args, dev)
    global SAGE_DIMS, STUDENT, GRAPH, CPU_SAMPLE_SCALE
    if args.workload == "arxiv":
        SAGE_DIMS, STUDENT, GRAPH, CPU_SAMPLE_SCALE = ARXIV["sage_dims"], ARXIV["student"], ARXIV["graph"], ARXIV["cpu_sample_scale"]

    # ---- synthetic ogbn-products-shaped inputs, generated in HBM (seed 0, identical on every rank) --------
    torch.manual_seed(0)
    if args.locality > 0:
        n_full = int(data.SHAPES[GRAPH]["n"] * args.scale)
        g = data.make_clustered_graph(n_full, 50.5 if GRAPH == "ogbn-products" else 14.8, communities=64, p_in=args.locality, seed=0, device=dev,
                                      shuffle_ids=args.shuffle_ids)
    else:
        g = data.make_graph(GRAPH, seed=0, device=dev, scale=args.scale)
    n, nnz = g.n_dst, g.num_edges()
    feats, labels, out_t, _ = data.make_node_data(GRAPH, seed=0, device=dev, n=n)
    partition_s = None
    if world > 1 and args.partition == "lp":      # one-time preparation, outside every timed region (identical on every rank)
        t0 = time.perf_counter()
        perm = data.locality_order(g, seed=0)
        g = data.relabel(g, perm)
        feats, labels, out_t = feats[perm], labels[perm], out_t[perm]
        torch.cuda.synchronize()
        partition_s = time.perf_counter() - t0
    feats = ops.as_feat(feats)

    teacher = Model(dict(model_name="SAGE", num_layers=3, feat_dim=SAGE_DIMS[0], hidden_dim=SAGE_DIMS[1],
                         label_dim=SAGE_DIMS[-1], dropout_ratio=0.5, norm_type="batch", device=dev))
    teacher.eval()
    # N > 1: destination-row ranges cut by WORK (in-edges + 2 per row), not by row count
    shards = RowShards(n, world, rank, chunks=(args.chunks or 4) if world > 1 else 1, bounds=RowShards.balanced_bounds(g.indptr, world) if world > 1 else None)
    ref_own, link_probe, autotune = None, None, None
    if world == 1 and args.layer1_exchange == "auto":
        args.layer1_exchange = "narrow"
    if world > 1:
        from glnn_amd.dist import probe_link as gdist_probe
        if not args.no_verify:      # the unsharded forward of this rank's rows, before the full graph is dropped: the sharded result
            with torch.no_grad():   # (whatever the transport did) must reproduce it
                ref_own = teacher.inference(FullNeighborLoader(g, 4096), feats)[shards.lo:shards.hi].clone()
        shard_graph = g.row_range(shards.lo, shards.hi)
        # what the transport delivers for the layer-1 payloads (narrow / wide slab of one rank): recorded, and the model's link rate
        link_probe = {w_: gdist_probe(world, rank, shards.rpr * ((d_ + 3) // 4 * 4), dev) for w_, d_ in (("narrow", SAGE_DIMS[0]), ("wide", SAGE_DIMS[1]))}
        if args.exchange == "halo":
            sharded = HaloShardedTeacher(teacher.encoder, shard_graph, shards, ops, overlap=not args.no_halo_overlap)
        else:
            if args.layer1_exchange == "auto":
                # self-tuning: the driver passes no flags and the right form depends on a link rate nobody has measured -- so measure the
                # forms themselves (what layer 1 puts on the wire x how many chunks the overlapped exchange is cut into), on this
                # transport, outside the timed region (max over ranks; identical decision on every rank)
                autotune, best = {}, None
                for form in ("narrow", "wide"):
                    for ch in ([args.chunks] if args.chunks else [2, 4, 8]):
                        sh_c = RowShards(n, world, rank, chunks=ch, bounds=shards.bounds)
                        cand = ShardedTeacher(teacher.encoder, shard_graph, sh_c, ops, widening_exchange=form)
                        with torch.no_grad():
                            cand.forward(feats)
                            barrier()
                            t0 = time.perf_counter()
                            for _ in range(3):
                                cand.forward(feats)
                            barrier()
                        tt = torch.tensor([(time.perf_counter() - t0) / 3], device=dev, dtype=torch.float64)
                        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        autotune[f"{form}/{ch}"] = 1e3 * float(tt.item())
                        if best is None or autotune[f"{form}/{ch}"] < autotune[best]:
                            best = f"{form}/{ch}"
                        del cand, sh_c
                        torch.cuda.empty_cache()
                args.layer1_exchange, ch = best.split("/")
                shards = RowShards(n, world, rank, chunks=int(ch), bounds=shards.bounds)
            sharded = ShardedTeacher(teacher.encoder, shard_graph, shards, ops, widening_exchange=args.layer1_exchange)
        shard_rows, shard_nnz = shards.rows, int(shard_graph.num_edges())
        del g
        torch.cuda.empty_cache()

        def teacher_forward():
            with torch.no_grad():
                return sharded.forward(feats)
    else:
        loader = FullNeighborLoader(g, 4096)

        def teacher_forward():
            return teacher.inference(loader, feats)

    edges_per_forward = 3 * nnz

    # ---- teacher: W warm-up forwards, then exactly K timed forwards ---------------------------------------
    for _ in range(args.warmup):
        teacher_forward()
    timing = []
    barrier()
    ops.set_timing(timing)          # per-launch HIP events on every rank (N > 1: kernel time vs wall time = the exposed exchange)
    from glnn_amd import dist as gdist
    gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    out_timed = None
    for _ in range(args.steps):
        out_timed = teacher_forward()
    ev1.record()                    # this rank's own end on its compute stream, before it waits for the others
    barrier()
    t_teacher = time.perf_counter() - t0
    ops.set_timing(None)
    rank_diag = None
    if world > 1:
        kms = kernel_breakdown(timing, args.steps)
        own_ms = ev0.elapsed_time(ev1) / args.steps
        mine = {"rank": rank, "rows": shard_rows, "nnz": shard_nnz, "wall_ms": own_ms, "kernel_ms": sum(kms.values()),
                "exchange_exposed_ms": own_ms - sum(kms.values()), "kernels": kms}
        rank_diag = [None] * world
        dist.all_gather_object(rank_diag, mine)
    verify = None
    if not args.no_verify:
        verify = verify_single(g, feats, teacher, out_timed, ops) if world == 1 else verify_sharded(out_timed, ref_own, dev, dist)
    del out_timed, ref_own
    placement = None
    if world > 1:
        mine = {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev)}
        placement = [None] * world
        dist.all_gather_object(placement, mine)
    if world > 1:
        tt = torch.tensor([t_teacher], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_teacher = float(tt.item())
    edges_per_s = edges_per_forward * args.steps / t_teacher

    # ---- student: fused distillation steps (soft-label pass: KL against teacher log-probs) ----------------
    sd = STUDENT
    student = Model(dict(model_name=sd["name"], num_layers=3, feat_dim=sd["dims"][0], hidden_dim=sd["dims"][1],
                         label_dim=sd["dims"][-1], dropout_ratio=sd["dropout"], norm_type="batch", device=dev))
    student.train()
    opt = torch.optim.Adam(student.parameters(), lr=sd["lr"], weight_decay=sd["wd"])
    eng = StudentEngine(student, opt, sd["batch"])
    if world > 1 and args.student_global_bn:     # one N*B-row batch split over ranks, global BN statistics, summed gradients
        eng.enable_batch_split(world, rank)
    elif world > 1:   # data parallel: every rank runs its own B-row batches, gradients averaged over ranks
        if args.no_grad_overlap:
            eng.grad_sync = make_grad_sync(eng.flat_grads, world, average=True)
        else:                                   # big weight gradients are all-reduced from inside the backward (grad_ready hook)
            eng.overlap = OverlappedGradSync(eng, world, average=True)
    out_t = ops.as_feat(out_t)
    k_student = args.steps * args.student_steps_per_step
    w_student = max(args.warmup, 3)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234 + rank)
    nb = max(1, n // sd["batch"])
    perm = torch.randperm(n, generator=gen)[: nb * sd["batch"]].view(nb, -1).to(dev)     # train_and_eval.py:65-71
    for i in range(w_student):
        eng.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
    barrier()
    t0 = time.perf_counter()
    for i in range(k_student):
        eng.step(feats, perm[(w_student + i) % nb], ops.LOSS_KL, out_t, 1.0)
    barrier()
    t_student = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([t_student], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_student = float(tt.item())
    student_steps_per_s = world * k_student / t_student      # B-row batches processed per second, whole job
    student_local_ms = None
    if world > 1 and not args.student_global_bn:
        # diagnostic, outside the timed region: the same step WITHOUT the gradient exchange (a second engine on a copy of the model: one
        # C call per step, Adam fused) -- the difference to the timed step is what data parallelism costs per step (exposed all-reduce
        # + the two-call form of the step), max over ranks
        s2 = Model(dict(model_name=sd["name"], num_layers=3, feat_dim=sd["dims"][0], hidden_dim=sd["dims"][1], label_dim=sd["dims"][-1],
                        dropout_ratio=sd["dropout"], norm_type="batch", device=dev))
        s2.train()
        e2 = StudentEngine(s2, torch.optim.Adam(s2.parameters(), lr=sd["lr"], weight_decay=sd["wd"]), sd["batch"])
        k2 = min(k_student, 200)
        for i in range(3):
            e2.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
        barrier()
        t0 = time.perf_counter()
        for i in range(k2):
            e2.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / k2], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        student_local_ms = 1e3 * float(tt.item())
        del s2, e2

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    result = {
        "metric": "aggregated edges/sec (teacher fwd) + student distill steps/sec, ogbn-products 1/2/4/8 GPU" if GRAPH == "ogbn-products"
                  else f"aggregated edges/sec (teacher fwd) + student distill steps/sec, {GRAPH} (BASELINE configs[1]+[2])",
        "value": edges_per_s, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_teacher / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "verified": None if verify is None else verify["ok"], "verify": verify,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
        "devices": placement,
        "config": {"workload": f"{GRAPH}-shaped SAGE teacher forward (3 layers {'-'.join(map(str, SAGE_DIMS))}, BN, layer-wise "
                               f"full-neighbour inference, reference models.py:121-148) + {STUDENT['name']} student KL distillation step",
                   "nodes": n, "nnz": nnz, "edges_aggregated_per_step": edges_per_forward,
                   "graph": "seeded power-law multigraph, random node order" if args.locality == 0 else
                            f"community-structured random graph (64 communities, {args.locality:.2f} of the edges inside), "
                            + ("node ids shuffled" if args.shuffle_ids else "community node order"),
                   "scale": args.scale, "exchange": args.exchange if world > 1 else None,
                   "layer1_exchange": args.layer1_exchange if (world > 1 and args.exchange == "allgather") else None,
                   "halo_overlap": (not args.no_halo_overlap) if (world > 1 and args.exchange == "halo") else None,
                   "partition": args.partition if world > 1 else None, "partition_seconds": partition_s,
                   "parallelism": "1 GPU" if world == 1 else f"node-range row shards x{world}, all-gather per layer; student dp{world}"},
        "exchange": None if world == 1 else {
            "GB_received_per_rank_per_forward": 4e-9 * gdist.EXCHANGE_STATS["floats_received"] / args.steps,
            "collectives_per_forward": gdist.EXCHANGE_STATS["collectives"] / args.steps,
            "link_probe": link_probe, "link_GBps_measured": link_probe["narrow"]["per_link_GBps"] if link_probe else None,
            "layer1_autotune_ms": autotune, "layer1_chosen": args.layer1_exchange if args.exchange == "allgather" else None,
            "chunks": shards.chunks,
            "ranks": rank_diag,
            "kernel_ms_max": max(r_["kernel_ms"] for r_ in rank_diag), "kernel_ms_mean": float(np.mean([r_["kernel_ms"] for r_ in rank_diag])),
            "wall_ms_max": max(r_["wall_ms"] for r_ in rank_diag),
            "exchange_exposed_ms_max": max(r_["exchange_exposed_ms"] for r_ in rank_diag),
            "exchange_exposed_ms_mean": float(np.mean([r_["exchange_exposed_ms"] for r_ in rank_diag])),
            "what": (("all-gathers of the narrow side of each layer boundary: 100-wide aggregate of layer 1 (chunked, overlapped "
                      "with the aggregation; the projection is replicated and consumes chunks in arrival order)" if args.layer1_exchange == "narrow" else
                      "all-gathers: 256-wide fused output of layer 1 (chunked, overlapped with the aggregation; no replicated projection)")
                     + ", 47-wide projection of layer 3 (chunked, overlapped with layer 2); layer 2 needs none")
                    if args.exchange == "allgather" else
                    "halo all-to-all of the narrow side of each layer boundary, only the remote rows this rank's edges reference"},
        "student": {"metric": f"student distill steps/s ({sd['name']} {'-'.join(map(str, sd['dims']))}, B={sd['batch']} per rank, dropout "
                              f"{sd['dropout']}, BN, KL soft-label step incl. gather, fwd, loss, bwd, Adam)",
                    "value": student_steps_per_s, "unit": "steps/s", "steps": k_student, "warmup": w_student,
                    "ms_per_step": 1e3 * t_student / k_student, "global_batch": world * sd["batch"],
                    "batchnorm": "global batch statistics (exchange hook)" if (world > 1 and args.student_global_bn)
                                 else "per-rank batch statistics",
                    "scaling": "weak",
                    "gradient_exchange": None if world == 1 else ("one all-reduce after the backward" if args.no_grad_overlap and not args.student_global_bn
                                                                  else "weight gradients >= 1 MB all-reduced from inside the backward (grad_ready hook), the rest after it"),
                    "local_step_ms": student_local_ms,
                    "dp_overhead_ms": None if student_local_ms is None else 1e3 * t_student / k_student - student_local_ms,
                    "gflop_per_step": 3 * 2 * sd["batch"] * sum(a * b for a, b in zip(sd["dims"][:-1], sd["dims"][1:])) / 1e9},
    }
    result["student"]["tflops"] = result["student"]["gflop_per_step"] * k_student / t_student / 1e3      # per GPU
    result["student"]["frac_of_fp32_mfma_peak"] = result["student"]["tflops"] / 157.3      # v_mfma_f32_32x32x2_f32 dense peak

    # ---- roofline of the dominant kernel (N = 1): per-launch HIP events from the timed region -------------
    if world == 1 and timing:
        full = GRAPH == "ogbn-products" and args.scale == 1.0
        result["roofline"] = roofline_object(timing, nnz, n, with_traffic=full)
        result["roofline"].update(hbm_estimate(result["roofline"], g))
        if args.reorder != "none":
            result["roofline_reordered"] = reordered_leg(args, g, feats, teacher, FullNeighborLoader, ops, data)

        if GRAPH == "ogbn-products" and not args.no_clustered_leg and args.locality == 0:
            result["roofline_clustered"] = clustered_leg(args, n, teacher, FullNeighborLoader, ops, data, dev)

    # ---- the B = 512 students of BASELINE configs[2] (latency-bound: steps/s, not an MFMA fraction) -- an extra object ----
    if world == 1 and not args.no_small_students:
        result["students_small"] = small_student_leg(dev, Model, StudentEngine, ops)

    # ---- sampled-block teacher TRAINING (SURVEY 8f rows 1+2; reference train_sage, train_and_eval.py:32-56) -- an extra object ----
    if world == 1 and not args.no_train_leg:
        result["teacher_training"] = teacher_training_leg(g, feats, labels, dev, data)

    # ---- CPU baseline on the host cores (oracle = 'port'; bounded sample) ---------------------------------
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sd, dev, min(CPU_SAMPLE_SCALE, args.scale), 1.0 if args.scale >= 1.0 else 0.1)

    # ---- BASELINE configs[1]+[2] and configs[4] on the same clock (N = 1, full-size products run): child processes of this one, after
    #      this process has released its device memory (the XL rank-forward keeps 228 GB resident).  A failed leg is reported, not fatal.
    if world == 1 and GRAPH == "ogbn-products" and args.scale == 1.0 and args.locality == 0:
        del g, feats, labels, out_t, teacher, student, eng, opt, perm, loader, shards
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if not args.no_arxiv_leg:
            result["arxiv"] = child_leg("arxiv", ["--workload", "arxiv", "--steps", "200", "--warmup", "5", "--student-steps-per-step", "10", "--no-cpu-baseline",
                                                   "--no-train-leg", "--no-small-students", "--reorder", "none"], 600)
        if not args.no_xl_leg:
            result["xl"] = child_leg("xl", ["--workload", "xl", "--steps", "5", "--warmup", "1"], 900)

    emit(result, args)
    if world > 1:
        dist.destroy_process_group()


def child_leg(name, argv, timeout_s):
    """Run `python bench.py <argv>` as a child process and return its DETAIL object (+ wall seconds); {"error": ...} if it failed."""
    import subprocess
    detail = os.path.join(ROOT, "gpurun_out", f"bench_detail_{name}.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.devnull
    t0 = time.perf_counter()
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--detail-file", detail], capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in p.stdout.splitlines() if l.startswith("DETAIL {")]
        if p.returncode != 0 or not lines:
            return {"error": f"rc {p.returncode}: " + (p.stderr.strip().splitlines() or ["no output"])[-1][:300], "wall_s": time.perf_counter() - t0}
        r = json.loads(lines[-1][len("DETAIL "):])
        r["wall_s"] = time.perf_counter() - t0
        return r
    except Exception as e:      # timeout, unparsable output: the products line is still printed
        return {"error": f"{type(e).__name__}: {e}"[:300], "wall_s": time.perf_counter() - t0}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run on
    127.0.0.1, RCCL), so that the collectives really see N ranks whatever command line the caller used."""
    import socket
    import subprocess
    if os.environ.get("GLNN_SINGLE_DEVICE") != "1" and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def verify_single(g, feats, teacher, out_timed, ops):
    """Self-check of the TIMED output (N = 1): the same forward recomputed WITHOUT the fused kernel, the chained projection and
    project-first -- every layer as stand-alone aggregation (spmm_csr_kernel) + GEMM, aggregate first as dgl does -- must agree
    within 1e-4; and the layer-1 aggregation satisfies the conservation identity
    sum_v (deg_v + 1) * mean_v == sum_u (outdeg_u + 1) * x_u in fp64."""
    enc = teacher.encoder
    n = g.n_dst
    with torch.no_grad():
        x = feats
        cons = None
        for l, layer in enumerate(enc.layers):
            es, eh, relu = enc._tail(l)
            agg = ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN)
            if l == 0:
                deg, outdeg = g.in_degrees().double(), g.out_degrees().double()
                d = x.shape[1]
                lhs = ((deg + 1).unsqueeze(1) * agg[:, :d].double()).sum(0)
                rhs = ((outdeg + 1).unsqueeze(1) * x[:, :d].double()).sum(0)
                cons = float((lhs - rhs).abs().max() / rhs.abs().max().clamp(min=1))
            x = ops.gemm(agg, layer.fc_neigh.weight, ep_scale=es, ep_shift=eh, relu=relu)
            del agg
        c = enc.layers[-1].fc_neigh.weight.shape[0]
        diff = float((x[:, :c] - out_timed[:, :c]).abs().max())
        finite = bool(torch.isfinite(out_timed[:, :c]).all())
    return {"ok": bool(finite and diff <= 1e-4 and cons < 1e-5), "max_abs_diff_vs_unfused_aggregate_first": diff, "tolerance": 1e-4,
            "layer1_conservation_rel_err_fp64": cons, "finite": finite, "rows_checked": n,
            "what": "timed output vs stand-alone aggregation + GEMM per layer (no fused kernel, no chained projection, aggregate-first)"}


def verify_sharded(out_own, ref_own, dev, dist):
    """Self-check of the TIMED output (N > 1): every rank's rows of the sharded forward vs the unsharded forward of the same
    rows computed on that rank before the graph was sharded; max over ranks."""
    c = ref_own.shape[1]
    d = (out_own[:, :c] - ref_own).abs().max() if out_own.numel() else torch.zeros((), device=dev)
    bad = (~torch.isfinite(out_own[:, :c])).any().float() if out_own.numel() else torch.zeros((), device=dev)
    t = torch.stack([d.double(), bad.double()])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    diff, nonfinite = float(t[0]), bool(t[1] > 0)
    return {"ok": bool(diff <= 1e-4 and not nonfinite), "max_abs_diff_vs_unsharded": diff, "tolerance": 1e-4, "finite": not nonfinite,
            "what": "each rank's rows of the sharded forward vs the unsharded forward of those rows (max over ranks)"}


def teacher_training_leg(g, feats, labels, dev, data):
    """Epochs of the reference's train_sage on the bench graph with the reference's config for it (train.conf.yaml:170-177 /
    196-204: fan-out 5,10,15; B=512 dropout 0.2 lr 0.01 on arxiv, B=4096 dropout 0.5 lr 0.003 on products): neighbour sampling
    and block building on the device (one batch ahead on a side stream), forward + NLL + backward + Adam on TeacherEngine."""
    from glnn_amd import train_and_eval as te
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    prod = GRAPH == "ogbn-products"
    bsz, p, lr = (4096, 0.5, 0.003) if prod else (512, 0.2, 0.01)
    n = g.n_dst
    n_train = max(bsz, int(n * (196615 / 2449029 if prod else 90941 / 169343)))
    torch.manual_seed(0)
    model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=SAGE_DIMS[0], hidden_dim=SAGE_DIMS[1], label_dim=SAGE_DIMS[-1],
                       dropout_ratio=p, norm_type="batch", device=dev))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    idx_train = torch.randperm(n)[:n_train].to(dev)
    loader = NodeDataLoader(g, idx_train, MultiLayerNeighborSampler([5, 10, 15]), batch_size=bsz, shuffle=True, drop_last=False)
    crit = torch.nn.NLLLoss()
    losses = [te.train_sage(model, loader, feats, labels, crit, opt)]            # warm-up epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    epochs = 3
    for _ in range(epochs):
        losses.append(te.train_sage(model, loader, feats, labels, crit, opt))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * len(loader)
    return {"metric": f"sampled-block GraphSAGE training steps/s ({GRAPH}-shaped graph, fan-out 5,10,15, B={bsz}, dropout {p}, BN; "
                      "sampling + block building + forward + NLL + backward + Adam, all on the device)",
            "value": steps / dt, "unit": "steps/s", "steps": steps, "ms_per_step": 1e3 * dt / steps, "epoch_s": dt / epochs,
            "train_nodes": n_train, "loss_first_last": [losses[0], losses[-1]]}


def roofline_object(timing, nnz, n, with_traffic):
    """The "roofline" object from (kernel, info, start_event, end_event) records of the aggregation launches."""
    torch.cuda.synchronize()
    per = {}
    for name, info, s, e in timing:
        key = (name, info.get("d", info.get("n")), info.get("d_out", info.get("k")), info.get("d_chain", 0), info.get("d_written"))
        per.setdefault(key, []).append(s.elapsed_time(e))
    layers, tot_b, tot_ms = [], 0.0, 0.0
    for (name, d, d_out, d_chain, d_written), ms in per.items():
        if name not in ("spmm", "sage_fused"):
            continue
        fused = name == "sage_fused"
        b = alg_bytes(nnz, n, d, (d_written if d_written is not None else d_out) if fused else None)     # bytes actually written per row
        avg = float(np.mean(ms))
        layers.append({"kernel": (f"sage_fused_kernel<LPR={lanes_per_row(((d + 7) // 8) * 8)},U={SPMM_U}> (aggregate {d} wide + project to {d_out} on MFMA"
                                  + (f", then to {d_chain} for the next layer: only those {d_written} floats per row are written)" if d_chain else ")")
                                  if fused else f"spmm_csr_kernel<LPR={lanes_per_row(d)},U={SPMM_U},SAGE_GCN>"),
                       "d": d, "avg_ms": avg, "alg_GB": b / 1e9, "GBps": b / avg / 1e6, "launches": len(ms),
                       "Gedges_per_s": nnz / avg / 1e6})
        tot_b += b
        tot_ms += avg
    gemm_ms = sum(float(np.mean(ms)) for (name, *_), ms in per.items() if name == "gemm")
    dom = max(layers, key=lambda r: r["avg_ms"])
    traffic = pmc_traffic(dom["kernel"].split(">")[0] + ">") if with_traffic else None
    return {
        "bound": "hbm", "kernel": f"{dom['kernel']} (D={dom['d']})",
        "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": None if traffic is None else f"{PMC_FILE} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                                       "committed; not measured by this run)",
        "algorithmic_bytes_per_launch": dom["alg_GB"] * 1e9, "avg_launch_ms": dom["avg_ms"],
        "all_aggregation_launches": sorted(layers, key=lambda r: r["d"]),
        "aggregation_total": {"alg_GB": tot_b / 1e9, "ms": tot_ms, "GBps": tot_b / tot_ms / 1e6, "frac": tot_b / tot_ms / 1e6 / HBM_PEAK_GBS},
        "dense_projection_ms_per_forward": gemm_ms,
        "note": "achieved = SURVEY 8(d) algorithmic bytes nnz*(4D+4)+n*(8D+8) / mean HIP-event launch time over the timed region"
                + ("" if GRAPH == "ogbn-products" else "; the feature matrix fits the 256 MB Infinity Cache at this shape, so the "
                   "rate is a cache rate and the HBM fraction is not meaningful (SURVEY 8d)"),
    }


def hbm_estimate(rf, g):
    """How much of the dominant launch's fabric traffic can the 256 MiB Infinity Cache have served?  rocprofv3 exposes no MALL
    hit counter on gfx950 (scripts/pmc_l2.sh lists what exists: the L2's fabric-side request counters count hits and misses of
    the memory-side cache alike), so the HBM bytes are BRACKETED: upper = every fabric request came from HBM (= `traffic`, or
    the algorithmic bytes when no PMC constant applies); lower = an ideal cache that pins the hottest source rows -- as many
    rows by out-degree as fit 256 MiB -- and serves every gather of them."""
    d = int(rf["kernel"].rsplit("D=", 1)[1].rstrip(")"))
    row_bytes = -(-4 * d // 128) * 128                      # a gathered row in whole 128-byte lines
    n, nnz = g.n_dst, g.num_edges()
    k = min(n, (256 << 20) // row_bytes)
    outdeg = g.out_degrees()
    hot = float(torch.topk(outdeg, k).values.double().sum() / max(1, nnz))
    upper = rf["traffic"] if rf.get("traffic") else rf["algorithmic_bytes_per_launch"]
    lower = upper - hot * nnz * row_bytes
    sec = rf["avg_launch_ms"] / 1e3
    return {"hbm_bytes_estimated": {"upper": upper, "lower": lower, "infinity_cache_hit_bound": hot, "hot_rows": k,
                                    "frac_of_peak_upper": upper / sec / (HBM_PEAK_GBS * 1e9), "frac_of_peak_lower": lower / sec / (HBM_PEAK_GBS * 1e9),
                                    "how": "no MALL hit counter on gfx950: upper = all fabric traffic from HBM; lower = the hottest source rows "
                                           "that fit 256 MiB (by out-degree) pinned in the Infinity Cache, every gather of them a hit"}}


def reordered_leg(args, g, feats, teacher, FullNeighborLoader, ops, data):
    """The same teacher forward on the SAME graph with its nodes renumbered by descending in-degree (hub rows -- the ones most
    edges gather -- become neighbours in memory, so they share cache lines and stay resident): SURVEY 8(d) allows this
    locality-ordered figure beside the random-order one.  Same algorithmic bytes, same kernels."""
    g2, perm = data.reorder_by_degree(g)
    feats2 = ops.as_feat(feats[perm])
    loader = FullNeighborLoader(g2, 4096)
    for _ in range(2):
        teacher.inference(loader, feats2)
    timing = []
    torch.cuda.synchronize()
    ops.set_timing(timing)
    t0 = time.perf_counter()
    for _ in range(args.reorder_steps):
        teacher.inference(loader, feats2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_timing(None)
    obj = roofline_object(timing, g2.num_edges(), g2.n_dst, with_traffic=False)
    obj.update({"order": "nodes renumbered by descending in-degree (stable)", "steps": args.reorder_steps,
                "edges_per_s": 3 * g2.num_edges() * args.reorder_steps / dt, "ms_per_step": 1e3 * dt / args.reorder_steps})
    return obj


def small_student_leg(dev, Model, StudentEngine, ops, steps=3000, warmup=100):
    """The reference's ogbn-arxiv students (train.conf.yaml:142-154: MLP 128-256-256-40 p=0.2 and MLP3w4 128-1024-1024-40 p=0.5,
    B = 512, BatchNorm, Adam lr 0.01) on arxiv-shaped synthetic rows: the whole KL distillation step (gather, forward, loss, backward,
    Adam) as StudentEngine.step issues it -- ONE C call, glnn_mlp_train_step_f32.  Latency-bound: a step is 11-13 dependent launches
    of 5-28 us (profiles/r03_student_arxiv_mlp_timeline.txt, ..._mlp3w4_timeline.txt), so the figure of merit is ms per step."""
    out = []
    # + two more sections of the reference's train.conf.yaml, one per other regime of the step: the products MLP (B = 4096 between the
    # latency and the streaming kernels, :179-185) and the cora MLP (BASELINE configs[0]: 1433 unaligned features, no norm, :17-21)
    for name, dims, p, B, norm, n in (("MLP", [128, 256, 256, 40], 0.2, 512, "batch", 169343), ("MLP3w4", [128, 1024, 1024, 40], 0.5, 512, "batch", 169343),
                                      ("products-MLP", [100, 256, 256, 47], 0.5, 4096, "batch", 400000), ("cora-MLP", [1433, 128, 7], 0.6, 140, "none", 2485)):
        torch.manual_seed(0)
        model = Model(dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=p,
                           norm_type=norm, device=dev))
        model.train()
        eng = StudentEngine(model, torch.optim.Adam(model.parameters(), lr=0.01), B)
        feats = ops.as_feat(torch.randn(n, dims[0], device=dev))
        out_t = ops.as_feat(torch.log_softmax(torch.randn(n, dims[-1], device=dev), 1))
        nb = n // B
        perm = torch.randperm(n)[: nb * B].view(nb, -1).to(dev)
        for i in range(warmup):
            eng.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out.append({"student": name, "dims": dims, "batch": B, "norm": norm, "dropout": p, "ms_per_step": 1e3 * dt, "steps_per_s": 1.0 / dt,
                    "steps": steps, "loss_finite": bool(torch.isfinite(eng.loss_out).all())})
        del eng, model, feats, out_t
    return out


def clustered_leg(args, n, teacher, FullNeighborLoader, ops, data, dev):
    """The same teacher forward on a graph WITH communities (the prescribed generator has none, so its gathers are uniformly
    random): data.make_clustered_graph, same node count and mean degree, 64 communities in id order, 95 % of the edges inside
    them -- a community's feature rows (38 k nodes x 1 KB = 39 MB at D=256) fit the 256 MB Infinity Cache, which is how a
    partition-ordered real co-purchase graph behaves.  Same kernels, same algorithmic byte model."""
    g2 = data.make_clustered_graph(n, 50.5, communities=64, p_in=0.95, seed=0, device=dev)
    feats2 = ops.as_feat(torch.randn(n, SAGE_DIMS[0], device=dev))
    loader = FullNeighborLoader(g2, 4096)
    for _ in range(2):
        teacher.inference(loader, feats2)
    timing = []
    torch.cuda.synchronize()
    ops.set_timing(timing)
    t0 = time.perf_counter()
    for _ in range(args.reorder_steps):
        teacher.inference(loader, feats2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_timing(None)
    obj = roofline_object(timing, g2.num_edges(), g2.n_dst, with_traffic=False)
    obj.update({"graph": "community-structured random graph: 64 communities in id order, 0.95 of the edges inside them, "
                         f"n={g2.n_dst}, nnz={g2.num_edges()}", "steps": args.reorder_steps,
                "edges_per_s": 3 * g2.num_edges() * args.reorder_steps / dt, "ms_per_step": 1e3 * dt / args.reorder_steps,
                "note": "algorithmic (no-reuse) bytes / launch time: with locality most gathers hit the Infinity Cache, so 'achieved' "
                        "may exceed what HBM alone could deliver -- it is a rate of the algorithm, not an HBM fraction"})
    return obj


class CheckedBackend:
    """A proxy of glnn_amd.ops for ONE verification forward of a sharded teacher (never inside a timed region): every aggregation /
    fused / GEMM launch runs as usual, then `sample` of its rows are (a) recomputed independently with torch index arithmetic in
    fp64 from the launch's own inputs -- mean = (sum_{e in row} x[src_e] + x_self) / (deg + 1), projections as fp64 matmuls, the
    epilogue per column -- and compared within `tol`, and (b) for stand-alone aggregations re-launched as a row range of their own,
    which must reproduce the rows bit for bit.  Works whatever filled the input buffers (real collectives, truth fills, synthetic
    fills): each launch is checked against ITS inputs."""

    def __init__(self, be, sample=4096, tol=1e-4, conservation=False):
        """conservation: stand-alone SAGE aggregations (no ReLU) are also held to the identity over ALL their rows, in fp64:
        sum_v (deg_v + 1) * mean_v == sum_u (edges of the launch out of u) * x_u + sum_v x_self_v  (a full pass over x per launch)."""
        self.be, self.sample, self.tol, self.report, self.ok, self.conservation = be, int(sample), tol, [], True, conservation

    def _conservation(self, indptr, indices, x, n_dst, xs, out, ep_scale, ep_shift):
        d, dev = x.shape[1], x.device
        e0, e1 = int(indptr[0]), int(indptr[n_dst])
        cnt = torch.bincount(indices[e0:e1].long(), minlength=x.shape[0]).double()
        deg1 = (indptr[1:n_dst + 1] - indptr[:n_dst]).double() + 1
        lhs = torch.zeros(d, dtype=torch.float64, device=dev)
        rhs = torch.zeros(d, dtype=torch.float64, device=dev)
        step = 1 << 20                                         # fp64 reductions in slabs (bounded temporaries)
        for s0 in range(0, n_dst, step):
            sl = slice(s0, min(n_dst, s0 + step))
            y = out[sl, :d].double()
            if ep_shift is not None:
                y = y - ep_shift.double()
            if ep_scale is not None:
                y = y / ep_scale.double()
            lhs += (deg1[sl].unsqueeze(1) * y).sum(0)
            rhs += xs[sl, :d].double().sum(0)
        for s0 in range(0, x.shape[0], step):
            sl = slice(s0, min(x.shape[0], s0 + step))
            rhs += (cnt[sl].unsqueeze(1) * x[sl, :d].double()).sum(0)
        return float((lhs - rhs).abs().max() / rhs.abs().max().clamp(min=1))

    def __getattr__(self, name):
        return getattr(self.be, name)

    def _range(self, n):
        k = min(self.sample, n)
        r0 = (n - k) // 2
        return r0, k

    def _agg_ref(self, indptr, indices, x, r0, k, mode, x_self, row_scale=None, col_scale=None):
        d = x.shape[1]
        e0, e1 = int(indptr[r0]), int(indptr[r0 + k])
        idx = indices[e0:e1].long()
        deg = indptr[r0 + 1:r0 + k + 1] - indptr[r0:r0 + k]
        dst = torch.repeat_interleave(torch.arange(k, device=x.device), deg)
        rows = x[idx][:, :d].double()
        if col_scale is not None:
            rows = rows * col_scale[idx].double().unsqueeze(1)
        acc = torch.zeros(k, d, dtype=torch.float64, device=x.device).index_add_(0, dst, rows)
        if mode == self.be.AGG_SAGE_GCN:
            return (acc + x_self[r0:r0 + k, :d].double()) / (deg.double() + 1).unsqueeze(1)
        return acc * row_scale[r0:r0 + k].double().unsqueeze(1) if row_scale is not None else acc

    @staticmethod
    def _epi(y, ep_scale, ep_shift, relu):
        if ep_scale is not None:
            y = y * ep_scale.double()
        if ep_shift is not None:
            y = y + ep_shift.double()
        return y.clamp(min=0) if relu else y

    def _note(self, what, diff, exact=None):
        good = bool(diff <= self.tol) and (exact is None or exact)
        self.ok = self.ok and good
        self.report.append({"launch": what, "max_abs_diff_vs_fp64": diff, **({} if exact is None else {"row_range_relaunch_bit_equal": exact})})

    def spmm(self, indptr, indices, x, n_dst, mode, row_scale=None, col_scale=None, ep_scale=None, ep_shift=None, relu=False, out=None,
             x_self=None, self_rows=None, **kw):
        # (kw: the hub plan of the HIP backend.  The row-range relaunch below runs WITHOUT one: plan and no plan must agree bit for bit)
        out = self.be.spmm(indptr, indices, x, n_dst, mode, row_scale=row_scale, col_scale=col_scale, ep_scale=ep_scale, ep_shift=ep_shift,
                           relu=relu, out=out, x_self=x_self, self_rows=self_rows, **kw)
        if n_dst and self_rows is None:
            xs = x if x_self is None else x_self
            r0, k = self._range(n_dst)
            ref = self._epi(self._agg_ref(indptr, indices, x, r0, k, mode, xs, row_scale, col_scale), ep_scale, ep_shift, relu)
            diff = float((out[r0:r0 + k].double() - ref).abs().max())
            again = self.be.spmm(indptr[r0:r0 + k + 1], indices, x, k, mode, row_scale=None if row_scale is None else row_scale[r0:r0 + k],
                                 col_scale=col_scale, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, x_self=xs[r0:r0 + k])
            self._note(f"spmm d={x.shape[1]} rows={n_dst}", diff, bool(torch.equal(again, out[r0:r0 + k])))
            if self.conservation and mode == self.be.AGG_SAGE_GCN and not relu:
                err = self._conservation(indptr, indices, x, n_dst, xs, out, ep_scale, ep_shift)
                self.report[-1]["conservation_rel_err_fp64_all_rows"] = err
                self.ok = self.ok and err < 1e-5
        return out

    def gemm(self, a, w, ep_scale=None, ep_shift=None, relu=False, out=None, **kw):
        out = self.be.gemm(a, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out, **kw)
        if not kw and a.shape[0]:
            r0, k = self._range(a.shape[0])
            ref = self._epi(a[r0:r0 + k].double() @ w.detach().double().t(), ep_scale, ep_shift, relu)
            self._note(f"gemm m={a.shape[0]} k={a.shape[1]} n={w.shape[0]}", float((out[r0:r0 + k].double() - ref).abs().max()))
        return out

    def sage_fused(self, indptr, indices, x, n_dst, w, ep_scale=None, ep_shift=None, relu=False, out=None, x_self=None, w_packed=None,
                   w_next=None, out_next=None, want_out=True, tile_order=None, **kw):
        res = self.be.sage_fused(indptr, indices, x, n_dst, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out, x_self=x_self,
                                 w_packed=w_packed, w_next=w_next, out_next=out_next, want_out=want_out, tile_order=tile_order, **kw)
        if n_dst:
            xs = x if x_self is None else x_self
            r0, k = self._range(n_dst)
            h = self._epi(self._agg_ref(indptr, indices, x, r0, k, self.be.AGG_SAGE_GCN, xs) @ w.detach().double().t(), ep_scale, ep_shift, relu)
            o, o2 = (res, None) if w_next is None else res
            diff = 0.0
            if o is not None:
                diff = float((o[r0:r0 + k].double() - h).abs().max())
            if w_next is not None:
                diff = max(diff, float((o2[r0:r0 + k].double() - h @ w_next.detach().double().t()).abs().max()))
            self._note(f"sage_fused d={x.shape[1]}->{w.shape[0]}" + (f"->{w_next.shape[0]}" if w_next is not None else "") + f" rows={n_dst}", diff)
        return res


XL_DIMS = [128, 256, 256, 47]      # 128-d features (BASELINE configs[4]); hidden 256 / 47 classes / BatchNorm as the products teacher (train.conf.yaml:196-204)
XGMI_LINK_GBS = (64.0, 77.0)       # effective one-direction rate of ONE xGMI link; 7 links per GPU, one per peer


def kernel_breakdown(timing, steps):
    """{launch family: ms per forward} from ops' (name, info, start, end) records (call after a synchronize)."""
    per = {}
    for name, info, s, e in timing:
        if name == "gemm":
            key = f"gemm k={info['k']} n={info['n']}"
        elif name == "sage_fused":
            key = f"sage_fused d={info['d']}->{info['d_out']}" + (f"->{info['d_chain']}" if info.get("d_chain") else "")
        else:
            key = f"spmm d={info['d']}"
        per[key] = per.get(key, 0.0) + s.elapsed_time(e)
    return {k: v / steps for k, v in per.items()}


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
    rows, deg, dims = int(12_500_000 * args.scale), 20, XL_DIMS
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
    ms = kernel_breakdown(timing, args.steps)
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
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_xl.json")) as f:
                return 4 * json.load(f)["per_launch_bytes"][pmc_key]["total"]
        except Exception:
            return None

    def agg_layer(name, key, d, d_written, what, pmc_key=None):
        t = ms.get(key)
        if t is None:
            return
        b = alg_bytes(nnz, rows, d, d_written)
        layers.append({"layer": name, "kernel": what, "bound": "hbm", "ms": t, "alg_GB": b / 1e9, "achieved": b / t / 1e6, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": b / t / 1e6 / HBM_PEAK_GBS, "Gedges_per_s": nnz / t / 1e6,
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
        link = [1e3 * gb / (shards_n - 1) / r for r in XGMI_LINK_GBS] if shards_n > 1 else [0.0, 0.0]
        layers.append({"layer": name, "kernel": "all-gather over xGMI" + (" (EMULATED: local fill of the same bytes)" if peers is not None else ""),
                       "GB_received_per_rank": gb, "emulated_fill_ms": fill_ms.get(key), "modelled_link_ms": link,
                       "model": f"{shards_n - 1} peers on {shards_n - 1} links in parallel at {XGMI_LINK_GBS[0]:.0f}-{XGMI_LINK_GBS[1]:.0f} GB/s each; chunked x{sh.chunks}: "
                                "all but the first chunk can hide under the producing kernel"})

    if wide:
        agg_layer("1 aggregate+project (own rows)", f"sage_fused d={d0}->{d1}", d0, d1, f"sage_fused_kernel<LPR={lanes_per_row(d0)}>")
        exchange_layer("1 exchange (256-wide output)", "y0", d1)
    else:
        agg_layer("1 aggregate (own rows)", f"spmm d={d0}", d0, None, f"spmm_csr_kernel<LPR={lanes_per_row(d0)},U={SPMM_U},SAGE_GCN>",
                  f"spmm_csr_kernel<LPR={lanes_per_row(d0)},U={SPMM_U},SAGE_GCN>")
        exchange_layer("1 exchange (128-wide aggregate)", "agg0", d0)
        gemm_layer("1 projection (replicated: ALL rows on every rank)", f"gemm k={d0} n={d1}", sh.n_pad, d0, d1)
    agg_layer("2 aggregate+project+chained 256->47 (own rows)", f"sage_fused d={d1}->{d2}->{c}", d1, c,
              f"sage_fused_kernel<LPR=64> (only the 47 chained floats per row are written)", f"sage_fused_kernel<LPR=64,U={SPMM_U}>")
    exchange_layer("3 exchange (47-wide projected rows)", "hw2", c)
    agg_layer("3 aggregate (own rows)", f"spmm d={c}", c, None, f"spmm_csr_kernel<LPR={lanes_per_row(c)},U={SPMM_U},SAGE_GCN>",
              f"spmm_csr_kernel<LPR={lanes_per_row(c)},U={SPMM_U},SAGE_GCN>")
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
        "roofline": {"bound": "hbm", "kernel": f"{dom['kernel']} (layer {dom['layer']})", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["frac"], "traffic": dom.get("traffic"), "algorithmic_bytes_per_launch_sum": dom["alg_GB"] * 1e9, "avg_ms_per_forward": dom["ms"],
                     "traffic_source": None if dom.get("traffic") is None else "static: profiles/pmc_traffic_xl.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"},
        "verify": verify,
    }
    emit(result, args)
    if world > 1:
        dist.destroy_process_group()


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
    n_full = int(data.SHAPES[GRAPH]["n"] * args.scale)
    torch.manual_seed(0)
    teacher = Model(dict(model_name="SAGE", num_layers=3, feat_dim=SAGE_DIMS[0], hidden_dim=SAGE_DIMS[1], label_dim=SAGE_DIMS[-1], dropout_ratio=0.5,
                         norm_type="batch", device=dev))
    teacher.eval()
    enc = teacher.encoder
    steps, out = max(1, args.steps), {}
    forms = [("allgather-narrow", "products", dict(exchange="allgather", l1="narrow")), ("allgather-wide", "products", dict(exchange="allgather", l1="wide")),
             ("halo-lp", "clustered", dict(exchange="halo"))]
    graphs = {}
    for form, gkind, cfg in forms:
        if gkind not in graphs:
            graphs.clear()
            torch.cuda.empty_cache()
            if gkind == "products":
                g = data.make_graph(GRAPH, seed=0, device=dev, scale=args.scale)
                prep = None
            else:
                g0 = data.make_clustered_graph(n_full, 50.5, communities=64, p_in=0.95, seed=0, device=dev, shuffle_ids=True)
                t0 = time.perf_counter()
                perm = data.locality_order(g0, seed=0)
                g = data.relabel(g0, perm)
                torch.cuda.synchronize()
                prep = time.perf_counter() - t0
                del g0, perm
            feats = ops.as_feat(torch.randn(g.n_dst, SAGE_DIMS[0], device=dev))
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
                    t = gdist.ShardedTeacher(enc, shard, sh, ops, group=peers, widening_exchange=cfg["l1"])
                with torch.no_grad():
                    for _ in range(2):
                        t.forward(feats)                   # warm-up (buffers, relabelled columns, packed weights, hub plans, first launches)
                    timing, peers.events = [], []
                    gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
                    torch.cuda.synchronize()
                    ops.set_timing(timing)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        y = t.forward(feats)
                    torch.cuda.synchronize()
                    wall = 1e3 * (time.perf_counter() - t0) / steps
                    ops.set_timing(None)
                kms = kernel_breakdown(timing, steps)
                fill = sum(s_.elapsed_time(e_) for _, _, s_, e_ in peers.events) / steps
                c = want.shape[1]
                diff = float((y[:, :c] - want[sh.lo:sh.hi, :c]).abs().max()) if sh.rows else 0.0
                ranks.append({"rank": r, "rows": sh.rows, "nnz": int(shard.num_edges()), "wall_ms": wall, "kernel_ms": sum(kms.values()), "fill_ms": fill,
                              "GB_received": 4e-9 * gdist.EXCHANGE_STATS["floats_received"] / steps, "kernels": kms,
                              "halo_rows": getattr(getattr(t, "plan", None), "n_halo", None), "max_abs_diff_vs_unsharded": diff})
                del t, peers, shard, y
                torch.cuda.empty_cache()
            kmax = max(x_["kernel_ms"] for x_ in ranks)
            gb = max(x_["GB_received"] for x_ in ranks)
            # per link: an all-gather's count includes the own slab (N slabs, N-1 of them arrive, one per link); a halo exchange's does not
            per_link = gb / N if cfg["exchange"] == "allgather" else gb / max(1, N - 1)
            link = [1e3 * per_link / rt for rt in XGMI_LINK_GBS]                 # the N-1 peers send over N-1 links in parallel
            res["worlds"][str(N)] = {
                "max_kernel_ms": kmax, "mean_kernel_ms": float(np.mean([x_["kernel_ms"] for x_ in ranks])), "max_GB_received_per_rank": gb,
                "modelled_link_ms": link, "forward_ms_exchange_hidden": max(kmax, link[1]), "forward_ms_exchange_exposed": kmax + link[0],
                "speedup_vs_one_gpu": [one_gpu_ms / (kmax + link[0]), one_gpu_ms / max(kmax, link[1])],
                "verified": all(x_["max_abs_diff_vs_unsharded"] <= 1e-4 for x_ in ranks), "ranks": ranks}
        out[form] = res
    result = {"metric": "per-rank kernel time of the N-rank sharded teacher forward, every rank emulated on ONE GPU (compute half of the scaling model)",
              "value": None, "unit": "ms", "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": None, "higher_is_better": False, "scaling": "strong",
              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"{GRAPH}-shaped SAGE teacher forward, ranks of N = {worlds} emulated (dist.EmulatedPeers, truth fills)", "scale": args.scale,
                         "link_GBps_assumed": list(XGMI_LINK_GBS)},
              "verified": all(w["verified"] for f in out.values() for w in f["worlds"].values()),
              "scale_model": out}
    emit(result, args)


def emit(result, args):
    """Print the long object first (one line, prefixed so that it is not mistaken for THE line), write it to the detail file, then
    ONE compact JSON line (<= 4 KB) last: the driver's record keeps only the tail of stdout."""
    path = args.detail_file
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        path = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    if path:
        try:
            with open(path, "w") as f:
                json.dump(result, f)
        except OSError:
            path = None
    print("DETAIL " + json.dumps(result), flush=True)
    print(json.dumps(compact(result, path)), flush=True)


def _short(v, n=160):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def compact(r, detail_path):
    """The contract keys + the headline numbers of every object; long prose, per-launch lists and sweeps stay in the DETAIL line."""
    c = {k: r.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data", "verified", "rccl_ranks", "backend")}
    cfg = r.get("config", {})
    c["config"] = {k: _short(v, 200) for k, v in cfg.items() if k in ("workload", "nodes", "nnz", "edges_aggregated_per_step", "scale", "exchange",
                                                                       "layer1_exchange", "partition", "parallelism", "rows_per_gpu", "nnz_per_gpu",
                                                                       "nodes_total", "shards", "rank_timed", "link_GBps_assumed")}
    v = r.get("verify")
    if v:
        c["verify"] = {k: v[k] for k in ("max_abs_diff_vs_unfused_aggregate_first", "max_abs_diff_vs_unsharded", "layer1_conservation_rel_err_fp64",
                                         "tolerance", "repeat_forward_bit_equal") if k in v}
        if "launches" in v:
            c["verify"]["max_abs_diff_vs_fp64"] = max([l["max_abs_diff_vs_fp64"] for l in v["launches"]] or [0.0])
    rf = r.get("roofline")
    if rf:
        c["roofline"] = {k: _short(rf.get(k), 120) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                              "avg_launch_ms", "algorithmic_bytes_per_launch_sum", "avg_ms_per_forward") if k in rf}
        if rf.get("traffic") is not None:
            c["roofline"]["traffic_source"] = "static: " + PMC_FILE
        if "hbm_bytes_estimated" in rf:
            h = rf["hbm_bytes_estimated"]
            c["roofline"]["hbm_frac_bracket"] = [h["frac_of_peak_lower"], h["frac_of_peak_upper"]]
        if "all_aggregation_launches" in rf:
            c["roofline"]["launches"] = [{"d": l["d"], "ms": l["avg_ms"], "GBps": l["GBps"]} for l in rf["all_aggregation_launches"]]
            c["roofline"]["dense_ms"] = rf.get("dense_projection_ms_per_forward")
    for key in ("roofline_reordered", "roofline_clustered"):
        if key in r:
            c[key] = {"edges_per_s": r[key]["edges_per_s"], "ms_per_step": r[key]["ms_per_step"], "frac": r[key]["frac"]}
    if "layers" in r:
        c["layers"] = [{k: _short(l.get(k), 60) for k in ("layer", "bound", "ms", "achieved", "unit", "frac", "GB_received_per_rank", "emulated_fill_ms",
                                                           "modelled_link_ms") if l.get(k) is not None} for l in r["layers"]]
        c["per_forward"] = {k: r["per_forward"][k] for k in ("wall_ms", "kernel_ms", "emulated_fill_ms", "GB_received_per_rank")}
    st = r.get("student")
    if st:
        c["student"] = {"metric": _short(st["metric"], 90), "value": st["value"], "unit": st["unit"], "ms_per_step": st["ms_per_step"], "steps": st["steps"],
                        "tflops": st["tflops"], "frac_of_fp32_mfma_peak": st["frac_of_fp32_mfma_peak"]}
        if st.get("local_step_ms") is not None:
            c["student"].update(local_step_ms=st["local_step_ms"], dp_overhead_ms=st["dp_overhead_ms"])
    if "students_small" in r:
        c["students_small"] = {s_["student"]: round(s_["ms_per_step"], 4) for s_ in r["students_small"]}
    if "teacher_training" in r:
        c["teacher_training"] = {"steps_per_s": r["teacher_training"]["value"], "ms_per_step": r["teacher_training"]["ms_per_step"]}
    cb = r.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb["sample"], 200),
                             "student_steps_per_s": cb["student_steps_per_s"], "student_threads": cb["student_threads_best"]}
    if r.get("exchange"):
        ex = r["exchange"]
        c["exchange"] = {k: ex.get(k) for k in ("GB_received_per_rank_per_forward", "collectives_per_forward", "link_GBps_measured", "layer1_autotune_ms",
                                                "layer1_chosen", "chunks", "kernel_ms_max", "kernel_ms_mean", "wall_ms_max", "exchange_exposed_ms_max",
                                                "exchange_exposed_ms_mean")}
        if ex.get("link_probe"):
            c["exchange"]["link_probe_GBps"] = {k: [round(v["per_link_GBps"], 2), round(v["received_GBps"], 2)] for k, v in ex["link_probe"].items()}
        if ex.get("ranks"):       # per rank: [kernel ms, wall ms, exposed exchange ms]; per launch family: the slowest rank's ms
            c["exchange"]["ranks"] = [[round(q["kernel_ms"], 3), round(q["wall_ms"], 3), round(q["exchange_exposed_ms"], 3)] for q in ex["ranks"]]
            fam = {}
            for q in ex["ranks"]:
                for k, v in q["kernels"].items():
                    fam[k] = max(fam.get(k, 0.0), v)
            c["exchange"]["family_ms_max"] = {k: round(v, 3) for k, v in fam.items()}
    x = r.get("xl")
    if x:                      # BASELINE configs[4] on the same clock: one rank-forward of the synthetic 100M-node / 2B-edge graph (child process)
        c["xl"] = {"error": _short(x["error"], 120)} if "error" in x else {
            "ms": x["ms_per_step"], "Gedges_per_s_per_gpu": x["value"] / 1e9, "verified": x.get("verified"), "kernel_ms": x["per_forward"]["kernel_ms"],
            "rows_per_gpu": x["config"]["rows_per_gpu"], "nnz_per_gpu": x["config"]["nnz_per_gpu"], "shards": x["config"]["shards"],
            "layers": [{"ms": round(l["ms"], 3), "bound": l["bound"], "frac": round(l["frac"], 4)} for l in x["layers"] if "frac" in l], "wall_s": round(x["wall_s"], 1)}
    x = r.get("arxiv")
    if x:                      # BASELINE configs[1] + [2]: arxiv-shaped teacher forward + the MLP3w4 student step (child process)
        c["arxiv"] = {"error": _short(x["error"], 120)} if "error" in x else {
            "ms": x["ms_per_step"], "Gedges_per_s": x["value"] / 1e9, "verified": x.get("verified"), "student": x["student"]["metric"].split("(", 1)[-1].split(" ", 1)[0],
            "student_ms": x["student"]["ms_per_step"], "nodes": x["config"]["nodes"], "nnz": x["config"]["nnz"], "wall_s": round(x["wall_s"], 1)}
    if "scale_model" in r:
        c["scale_model"] = {f: {"one_gpu_ms": o["one_gpu_forward_ms"],
                                **{N: {"max_kernel_ms": w["max_kernel_ms"], "GB": w["max_GB_received_per_rank"], "link_ms": w["modelled_link_ms"],
                                       "speedup": w["speedup_vs_one_gpu"]} for N, w in o["worlds"].items()}} for f, o in r["scale_model"].items()}
    c["detail"] = "the preceding stdout line (prefix 'DETAIL ')" + (f" and {os.path.relpath(detail_path, ROOT)}" if detail_path else "")
    return c


def cpu_baseline(sd, dev, scale, budget):
    """SURVEY 8(d) / BASELINE.md section 3 CPU baseline on this host's cores (bounded: ~30-60 s).
    (i) teacher: the 3-layer SAGE forward of the metric on the metric's own full-size products-shaped graph (scale 1.0: n 2,449,029 /
        nnz 123,718,280; same generator as the GPU run), oracle/glnn_oracle.c with OpenMP on all host threads = kind "port" (the reference's own
        dgl CPU path cannot be timed: dgl is not installed); beside it torch.sparse_csr @ X for the layer-1 aggregation.
    (ii) student: the reference's loop body (train_and_eval.py:74-85) as the SAME PyTorch CPU ops it issues -- nn.Linear,
        nn.BatchNorm1d, relu, nn.Dropout, log_softmax, nn.KLDivLoss(batchmean, log_target), loss.backward(),
        optim.Adam.step() -- written out here with torch.nn modules (not an import of the reference, which is absent
        on the GPU box), torch.get_num_threads() threads."""
    from oracle import teacher_oracle as to
    from glnn_amd import data
    threads = to.max_threads()
    g = data.make_graph(GRAPH, seed=0, device=dev, scale=scale).to("cpu")          # generated in HBM, copied to the host once
    n, nnz = g.n_dst, g.num_edges()
    rs = np.random.RandomState(0)
    x = rs.standard_normal((n, SAGE_DIMS[0])).astype(np.float32)
    layers, norms = [], []
    for i in range(3):
        layers.append(dict(weight=(rs.standard_normal((SAGE_DIMS[i + 1], SAGE_DIMS[i])) / np.sqrt(SAGE_DIMS[i])).astype(np.float32),
                           bias=np.zeros(SAGE_DIMS[i + 1], np.float32)))
        if i < 2:
            h = SAGE_DIMS[i + 1]
            norms.append(dict(weight=np.ones(h, np.float32), bias=np.zeros(h, np.float32),
                              running_mean=np.zeros(h, np.float32), running_var=np.ones(h, np.float32)))
    ip, ix = g.indptr.numpy(), g.indices.numpy()
    to.sage_gcn_agg(ip, ix, x, threads=threads)                                    # page in / warm up
    rep_s = []
    while len(rep_s) < (2 if scale >= 1.0 else 3):                                 # full forwards (two at full size), the best one is reported
        t0 = time.perf_counter()
        to.sage_inference(ip, ix, x, layers, norms, threads=threads)
        rep_s.append(time.perf_counter() - t0)
    reps, t_teacher = len(rep_s), min(rep_s)
    t1 = time.perf_counter()
    areps = 0
    while areps < 1 or time.perf_counter() - t1 < 2.0 * budget:
        to.sage_gcn_agg(ip, ix, x, threads=threads)
        areps += 1
    t_agg = (time.perf_counter() - t1) / areps
    # second opinion for the aggregation: torch.sparse_csr @ X (what a torch-only CPU port would call)
    torch_threads = torch.get_num_threads()
    a = torch.sparse_csr_tensor(g.indptr, g.indices.long(), torch.ones(nnz), size=(n, n))
    xt = torch.from_numpy(x)
    a @ xt
    t2 = time.perf_counter()
    sreps = 0
    while sreps < 1 or time.perf_counter() - t2 < 2.0 * budget:
        a @ xt
        sreps += 1
    t_sparse = (time.perf_counter() - t2) / sreps
    del a
    # student: the reference's step as PyTorch CPU ops (train_and_eval.py:74-85; modules as models.py:7-53 builds them)
    dims, B = sd["dims"], sd["batch"]
    nn = torch.nn
    lin = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(3)])
    bns = nn.ModuleList([nn.BatchNorm1d(dims[i + 1]) for i in range(2)])
    drop = nn.Dropout(sd["dropout"])
    params = list(lin.parameters()) + list(bns.parameters())
    opt = torch.optim.Adam(params, lr=sd["lr"], weight_decay=sd["wd"])
    crit = nn.KLDivLoss(reduction="batchmean", log_target=True)
    feats = torch.randn(4 * B, dims[0])
    out_t = torch.log_softmax(torch.randn(4 * B, dims[-1]), dim=1)

    def step(i):
        idx = torch.arange((i % 4) * B, (i % 4 + 1) * B)
        h = feats[idx]
        for l in range(3):
            h = lin[l](h)
            if l != 2:
                h = drop(torch.relu(bns[l](h)))
        loss = crit(h.log_softmax(dim=1), out_t[idx])
        loss.item()
        loss = loss * 1.0
        opt.zero_grad()
        loss.backward()
        opt.step()

    # thread sweep: 128 torch threads on a 128-core host ran these GEMMs SLOWER than fewer (oversubscribed fork/join per op);
    # the best setting is the baseline, every point of the sweep is reported
    cores = os.cpu_count() or torch_threads
    sweep, steps = [], 0
    for nt in sorted({t for t in (8, 16, 32, 64, 128) if t <= cores}):      # (256 SMT threads: 0.09 steps/s, 23 s for two steps -- not swept)
        torch.set_num_threads(nt)
        step(0)
        t3 = time.perf_counter()
        k = 0
        while k < 2 or time.perf_counter() - t3 < 3.0 * budget:
            step(k)
            k += 1
        sweep.append({"threads": nt, "steps_per_s": k / (time.perf_counter() - t3), "steps": k})
        steps += k
    torch.set_num_threads(torch_threads)
    best = max(sweep, key=lambda r: r["steps_per_s"])
    t_step = 1.0 / best["steps_per_s"]
    return {"value": 3 * nnz / t_teacher, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"3-layer SAGE forward ({'-'.join(map(str, SAGE_DIMS))}, BN eval) on a {scale}-scale {GRAPH}-shaped graph "
                      f"(n={n}, nnz={nnz}; feature matrices {4e-9 * n * SAGE_DIMS[0]:.2f} / {4e-9 * n * SAGE_DIMS[1]:.2f} GB > LLC), "
                      f"oracle/glnn_oracle.c with OpenMP on {threads} threads, best of {reps} forwards ({', '.join(f'{t:.2f}' for t in rep_s)} s); "
                      "the reference's own dgl CPU path cannot be timed (dgl not installed)",
            "aggregation_only_edges_per_s": nnz / t_agg,
            "aggregation_torch_sparse_csr_edges_per_s": nnz / t_sparse,
            "aggregation_torch_sparse_csr_threads": torch_threads,
            "teacher_reps": reps,
            "student_steps_per_s": 1.0 / t_step, "student_threads_best": best["threads"], "student_thread_sweep": sweep,
            "student_kind": "reference-equivalent PyTorch CPU ops (nn.Linear / BatchNorm1d / relu / Dropout / log_softmax / KLDivLoss / "
                            "loss.backward / Adam.step in the order of reference train_and_eval.py:74-85)",
            "student_sample": f"{sd['name']} dims, B={B}, dropout {sd['dropout']}, {steps} steps over a thread sweep "
                              f"({', '.join(str(r['threads']) for r in sweep)} torch threads; best = {best['threads']}), host cores = {cores}"}


if __name__ == "__main__":
    main()
