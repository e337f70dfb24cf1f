"""The products / arxiv workloads of bench.py: teacher forward (1 GPU or row-sharded over N ranks) + student distillation steps, and the
extra objects of the N = 1 line."""
import os
import time

import numpy as np
import torch

from . import common as C
from .checker import verify_sharded, verify_single
from .cpu_baseline import cpu_baseline
from .legs import child_leg, chunked_leg, clustered_leg, hbm_estimate, reordered_leg, roofline_object, small_student_leg, teacher_training_leg
from .line import emit


class Ladder:
    """The fail-safe ladder of an N > 1 run.  No multi-rank RCCL collective of this repository has ever executed
    on hardware, so every stage that touches the transport -- link probe, every autotune candidate, the timed regions -- runs under
    attempt(): an exception on ANY rank is recorded and agreed on over a gloo side group (not over the communicator that may just
    have failed), and the run steps down to the next, plainer form instead of ending without a line:
        teacher  autotuned chunked / overlapped all-gathers -> synchronous un-chunked in-place all-gather -> synchronous out-of-place
                 (list form) all-gather -> a line with value null and the errors
        student  all-reduce from inside the backward -> one all-reduce after the backward -> student object with value null
    A collective that HANGS cannot be caught; the deadline watchdog of bench.py prints the null line for that case."""

    def __init__(self, world, side):
        self.world, self.side, self.errors, self.stage = world, side, {}, "start"

    def agree(self, ok):
        if self.world == 1 or self.side is None:
            return ok
        import torch.distributed as dist
        t = torch.tensor([0 if ok else 1], dtype=torch.int32)          # a CPU tensor: the gloo side group
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.side)
        return int(t.item()) == 0

    def attempt(self, key, fn):
        """fn() on every rank -> (ok on ALL ranks, its result here).  Exactly one agree() per call on every rank."""
        self.stage = key
        err, res = None, None
        try:
            res = fn()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001  (whatever the transport raises)
            err = f"{type(e).__name__}: {e}".replace("\n", " ")[:300]
        ok = self.agree(err is None)
        if not ok:
            self.errors[key] = err or "failed on another rank"
        return ok, (res if ok else None)


def emit_failure(args, rank, world, lad, why):
    """The line of a run that could not produce its number: the contract keys with value null, and what failed where."""
    if rank == 0:
        emit({"metric": "aggregated edges/sec (teacher fwd) + student distill steps/sec, ogbn-products 1/2/4/8 GPU", "value": None, "unit": "edges/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
              "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": None, "rccl_ranks": world, "backend": None,
              "config": {"workload": f"{C.GRAPH}-shaped SAGE teacher forward + {C.STUDENT['name']} student step", "scale": args.scale},
              "error": why, "exchange": {"ladder": {"teacher_rung": None, "student_rung": None, "errors": (lad.errors if lad else None) or None,
                                                    "stage": lad.stage if lad else None}}}, args)
    return None


def run_products(args, rank, world, dev, barrier):
    import torch.distributed as dist
    from glnn_amd import data, ops
    from glnn_amd.dist import HaloShardedTeacher, OverlappedGradSync, RowShards, ShardedTeacher, make_grad_sync
    from glnn_amd.graph import FullNeighborLoader
    from glnn_amd.models import Model
    from glnn_amd.student import StudentEngine
    C.use_workload(args.workload)

    # ---- synthetic ogbn-products-shaped inputs, generated in HBM (seed 0, identical on every rank) --------
    torch.manual_seed(0)
    if args.locality > 0:
        n_full = int(data.SHAPES[C.GRAPH]["n"] * args.scale)
        g = data.make_clustered_graph(n_full, 50.5 if C.GRAPH == "ogbn-products" else 14.8, communities=64, p_in=args.locality, seed=0, device=dev,
                                      shuffle_ids=args.shuffle_ids)
    else:
        g = data.make_graph(C.GRAPH, seed=0, device=dev, scale=args.scale)
    n, nnz = g.n_dst, g.num_edges()
    # which graph this is (the seeded generator changed once, in round 5: numbers of different rounds are comparable only under one fingerprint)
    fingerprint = f"{n}-{nnz}-{int(g.indices[::max(1, nnz // 65536)].long().sum().item()):x}-{int(g.indptr[::max(1, n // 65536)].sum().item()):x}"
    feats, labels, out_t, _ = data.make_node_data(C.GRAPH, seed=0, device=dev, n=n)
    partition_s = None
    if world > 1 and args.partition == "lp":      # one-time preparation, outside every timed region (identical on every rank)
        t0 = time.perf_counter()
        perm = data.locality_order(g, seed=0)
        g = data.relabel(g, perm)
        feats, labels, out_t = feats[perm], labels[perm], out_t[perm]
        torch.cuda.synchronize()
        partition_s = time.perf_counter() - t0
    feats = ops.as_feat(feats)

    teacher = Model(dict(model_name="SAGE", num_layers=3, feat_dim=C.SAGE_DIMS[0], hidden_dim=C.SAGE_DIMS[1],
                         label_dim=C.SAGE_DIMS[-1], dropout_ratio=0.5, norm_type="batch", device=dev))
    teacher.eval()
    # N > 1: destination-row ranges cut by WORK (in-edges + 2 per row), not by row count
    shards = RowShards(n, world, rank, chunks=(args.chunks or 4) if world > 1 else 1, bounds=RowShards.balanced_bounds(g.indptr, world) if world > 1 else None)
    ref_own, link_probe, autotune = None, None, None
    lad = Ladder(world, getattr(args, "side_group", None))
    args.ladder = lad
    rung = None
    if world == 1 and args.layer1_exchange == "auto":
        args.layer1_exchange = "narrow"
    if world > 1:
        from glnn_amd import dist as gdist
        from glnn_amd.dist import probe_link as gdist_probe
        for spec in filter(None, os.environ.get("GLNN_BENCH_INJECT", "").split(",")):      # tests: "async:1000000,probe:1"
            kind, cnt = spec.split(":")
            gdist.INJECT_FAIL[kind] = int(cnt)
        if not args.no_verify:      # the unsharded forward of this rank's rows, before the full graph is dropped: the sharded result
            with torch.no_grad():   # (whatever the transport did) must reproduce it
                ref_own = teacher.inference(FullNeighborLoader(g, 4096), feats)[shards.lo:shards.hi].clone()
        shard_graph = g.row_range(shards.lo, shards.hi)
        # what the transport delivers for the layer-1 payloads (narrow / wide slab of one rank): recorded, and the model's link rate
        _, link_probe = lad.attempt("link probe", lambda: {w_: gdist_probe(world, rank, shards.rpr * ((d_ + 3) // 4 * 4), dev)
                                                           for w_, d_ in (("narrow", C.SAGE_DIMS[0]), ("wide", C.SAGE_DIMS[1]))})
        base_bounds = shards.bounds

        def time_candidate(form, ch, frac=None):
            sh_c = RowShards(n, world, rank, chunks=ch, bounds=base_bounds)
            cand = ShardedTeacher(teacher.encoder, shard_graph, sh_c, ops, widening_exchange=form, mixed_fraction=frac or args.mixed_fraction)
            with torch.no_grad():
                cand.forward(feats)
                barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    cand.forward(feats)
                barrier()
            tt = torch.tensor([(time.perf_counter() - t0) / 3], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt.item())

        def build_first():
            """rung 0: what the flags ask for -- the halo exchange, a given all-gather form, or (default) the autotuned one"""
            nonlocal autotune, shards
            if args.exchange == "halo":
                return HaloShardedTeacher(teacher.encoder, shard_graph, shards, ops, overlap=not args.no_halo_overlap)
            if args.layer1_exchange == "auto":
                # self-tuning: the driver passes no flags and the right form depends on a link rate nobody has measured -- so measure the
                # forms themselves (what layer 1 puts on the wire x how many chunks the overlapped exchange is cut into), on this
                # transport, outside the timed region (max over ranks; identical decision on every rank).  A candidate that fails on any
                # rank is recorded as an error string and skipped on every rank.
                autotune, best, fracs = {}, None, {}
                # ("mixed0.75": three quarters of the chunks travel wide -- a stop of the dial that exists from 4 chunks on; with the fused layers
                #  as ONE launch over the chunks the chunk count costs them nothing, and the emulated model has it fastest: 4.66 ms per rank at N = 8)
                for form, frac in (("narrow", None), ("wide", None), ("mixed", None), ("mixed0.75", 0.75)):
                    for ch in ([args.chunks] if args.chunks else [2, 4, 8]):
                        if frac is not None and ch < 4:
                            continue
                        key = f"{form}/{ch}"
                        fracs[key] = frac
                        ok, ms = lad.attempt(f"autotune {key}", lambda: time_candidate("mixed" if frac else form, ch, frac))
                        autotune[key] = ms if ok else "error: " + lad.errors[f"autotune {key}"]
                        if ok and (best is None or ms < autotune[best]):
                            best = key
                        torch.cuda.empty_cache()
                if best is None:
                    raise RuntimeError("every overlapped all-gather form failed")
                form, ch = best.split("/")
                if fracs[best]:
                    form, args.mixed_fraction = "mixed", fracs[best]
                args.layer1_exchange = form
                shards = RowShards(n, world, rank, chunks=int(ch), bounds=base_bounds)
            return ShardedTeacher(teacher.encoder, shard_graph, shards, ops, widening_exchange=args.layer1_exchange, mixed_fraction=args.mixed_fraction)

        def build_sync(list_form):
            nonlocal shards
            gdist.SAFE_LIST_FORM = list_form
            args.exchange, args.layer1_exchange = "allgather", "narrow"
            shards = RowShards(n, world, rank, chunks=1, bounds=base_bounds)
            return ShardedTeacher(teacher.encoder, shard_graph, shards, ops, widening_exchange="narrow")

        as_asked = (args.layer1_exchange, args.exchange, args.mixed_fraction)

        def build_per_chunk():
            """rung 1: the same forms with ONE LAUNCH PER CHUNK (no completion signals, no hipStreamWaitValue32): what rounds 4-5 shipped"""
            gdist.ONE_LAUNCH = False
            args.layer1_exchange, args.exchange, args.mixed_fraction = as_asked
            return build_first()

        rungs = [("as configured", build_first),
                 ("as configured, one launch per chunk (no completion signals)", build_per_chunk),
                 ("synchronous un-chunked in-place all-gather", lambda: build_sync(False)),
                 ("synchronous out-of-place (list form) all-gather", lambda: build_sync(True))]
        shard_rows, shard_nnz = shards.rows, int(shard_graph.num_edges())
        del g
        torch.cuda.empty_cache()
        sharded = None

        def teacher_forward():
            with torch.no_grad():
                return sharded.forward(feats)
    else:
        loader = FullNeighborLoader(g, 4096)
        rungs = [("1 GPU", None)]

        def teacher_forward():
            return teacher.inference(loader, feats)

    edges_per_forward = 3 * nnz

    def timed_teacher():
        nonlocal out_timed
        # ---- teacher: W warm-up forwards, then exactly K timed forwards ---------------------------------------
        for _ in range(args.warmup):
            out_timed = teacher_forward()                   # (held like the timed loop holds it: the allocator then has its second output buffer)
        if world == 1 and C.GRAPH != "ogbn-products":      # sub-millisecond forwards: --warmup of them is a few ms of load, and the part's clocks
            t_w = time.perf_counter()                      # take a few hundred ms to come up (the first 0.17 s region of a fresh process measured
            while time.perf_counter() - t_w < 0.6:         # 0.96-1.21 ms per forward, every later one 0.83): warm up for 0.6 s
                for _ in range(20):
                    out_timed = teacher_forward()
                torch.cuda.synchronize()
        timing = []
        # A sub-millisecond forward (the arxiv-shaped graph: three launches, 0.85 ms) is timed WITHOUT the per-launch events and the events
        # are taken over a second pass of the same K forwards: two timing events per launch inside the timed region cost it 0.25-0.4 ms per
        # forward (profiles/r05_arxiv_event_overhead.txt: 1.10-1.26 ms with them, 0.84 without).  The products forward (33 ms) keeps them
        # inside the timed region, as the contract asks.
        events_apart = world == 1 and C.GRAPH != "ogbn-products"
        barrier()
        if not events_apart:
            ops.set_timing(timing)      # per-launch HIP events on every rank (N > 1: kernel time vs wall time = the exposed exchange)
        gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            out_timed = teacher_forward()
        ev1.record()                    # this rank's own end on its compute stream, before it waits for the others
        barrier()
        t_teacher = time.perf_counter() - t0
        regions_ms = None
        if events_apart:
            # ... and the K forwards are timed three times, the median region is the figure (a 0.17 s region right after process start
            # measured 0.84 / 0.97 / 1.21 ms per forward from process to process around a kernel sum of 0.84; all three regions are recorded)
            regs = [t_teacher]
            for _ in range(2):
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    out_timed = teacher_forward()
                barrier()
                regs.append(time.perf_counter() - t0)
            regions_ms = [1e3 * r_ / args.steps for r_ in regs]
            t_teacher = sorted(regs)[1]
            ops.set_timing(timing)
            for _ in range(args.steps):
                teacher_forward()
            barrier()
        ops.set_timing(None)
        return dict(timing=timing, ev0=ev0, ev1=ev1, t_teacher=t_teacher, regions_ms=regions_ms)

    out_timed, tr = None, None
    from glnn_amd import dist as gdist
    for rung, build in rungs:
        if build is not None:
            ok, sharded = lad.attempt(f"set-up [{rung}]", build)
            if not ok:
                continue
        ok, tr = lad.attempt(f"timed region [{rung}]", timed_teacher)
        ops.set_timing(None)
        if ok:
            break
        sharded, out_timed = None, None
        torch.cuda.empty_cache()
    if tr is None:
        return emit_failure(args, rank, world, lad, "teacher forward: every rung of the ladder failed")
    timing, ev0, ev1, t_teacher, regions_ms = tr["timing"], tr["ev0"], tr["ev1"], tr["t_teacher"], tr["regions_ms"]
    rank_diag = None
    if world > 1:
        kms = C.kernel_breakdown(timing, args.steps)
        own_ms = ev0.elapsed_time(ev1) / args.steps
        mine = {"rank": rank, "rows": shard_rows, "nnz": shard_nnz, "wall_ms": own_ms, "kernel_ms": sum(kms.values()),
                "exchange_exposed_ms": own_ms - sum(kms.values()), "kernels": kms}
        rank_diag = [None] * world
        dist.all_gather_object(rank_diag, mine)
    verify = None
    if not args.no_verify:
        verify = verify_single(g, feats, teacher, out_timed, ops) if world == 1 else verify_sharded(out_timed, ref_own, dev, dist)
    chunked = None
    if world == 1 and not args.no_chunked_leg:
        chunked = chunked_leg(g, feats, teacher, FullNeighborLoader, ops, out_timed)
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
    sd = C.STUDENT
    student = Model(dict(model_name=sd["name"], num_layers=3, feat_dim=sd["dims"][0], hidden_dim=sd["dims"][1],
                         label_dim=sd["dims"][-1], dropout_ratio=sd["dropout"], norm_type="batch", device=dev))
    student.train()
    opt = torch.optim.Adam(student.parameters(), lr=sd["lr"], weight_decay=sd["wd"])
    eng = StudentEngine(student, opt, sd["batch"])
    out_t = ops.as_feat(out_t)
    k_student = args.steps * args.student_steps_per_step
    w_student = max(args.warmup, 3)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234 + rank)
    nb = max(1, n // sd["batch"])
    perm = torch.randperm(n, generator=gen)[: nb * sd["batch"]].view(nb, -1).to(dev)     # train_and_eval.py:65-71

    def student_overlapped():
        eng.overlap = OverlappedGradSync(eng, world, average=True)      # big weight gradients are all-reduced from inside the backward (grad_ready hook)

    def student_after_backward():
        if getattr(eng, "overlap", None) is not None:
            eng.overlap.detach()
            eng.overlap = None
        eng.grad_sync = make_grad_sync(eng.flat_grads, world, average=True)

    def timed_student():
        for i in range(w_student):
            eng.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
        barrier()
        t0 = time.perf_counter()
        for i in range(k_student):
            eng.step(feats, perm[(w_student + i) % nb], ops.LOSS_KL, out_t, 1.0)
        barrier()
        return time.perf_counter() - t0

    if world == 1:
        s_rungs = [("1 GPU", None)]
    elif args.student_global_bn:     # one N*B-row batch split over ranks, global BN statistics, summed gradients
        s_rungs = [("batch split, global BatchNorm statistics", lambda: eng.enable_batch_split(world, rank))]
    elif args.no_grad_overlap:       # data parallel: every rank runs its own B-row batches, gradients averaged over ranks
        s_rungs = [("one all-reduce after the backward", student_after_backward)]
    else:
        s_rungs = [("all-reduce from inside the backward", student_overlapped), ("one all-reduce after the backward", student_after_backward)]
    t_student, s_rung = None, None
    for s_rung, setup in s_rungs:
        if setup is not None:
            ok, _ = lad.attempt(f"student set-up [{s_rung}]", setup)
            if not ok:
                continue
        ok, t_student = lad.attempt(f"student timed region [{s_rung}]", timed_student)
        if ok:
            break
    if t_student is None:            # no gradient exchange works: the student figure is null, the line still stands
        t_student = float("nan")
    if world > 1 and t_student == t_student:
        tt = torch.tensor([t_student], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_student = float(tt.item())
    student_steps_per_s = world * k_student / t_student      # B-row batches processed per second, whole job
    student_local_ms = None
    if world > 1 and not args.student_global_bn and t_student == t_student:
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
        "metric": "aggregated edges/sec (teacher fwd) + student distill steps/sec, ogbn-products 1/2/4/8 GPU" if C.GRAPH == "ogbn-products"
                  else f"aggregated edges/sec (teacher fwd) + student distill steps/sec, {C.GRAPH} (BASELINE configs[1]+[2])",
        "value": edges_per_s, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_teacher / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timed_regions_ms_per_step": regions_ms,
        "verified": None if verify is None else verify["ok"], "verify": verify,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
        "devices": placement,
        "config": {"workload": f"{C.GRAPH}-shaped SAGE teacher forward (3 layers {'-'.join(map(str, C.SAGE_DIMS))}, BN, layer-wise "
                               f"full-neighbour inference, reference models.py:121-148) + {C.STUDENT['name']} student KL distillation step",
                   "nodes": n, "nnz": nnz, "edges_aggregated_per_step": edges_per_forward, "graph_fingerprint": fingerprint,
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
            "ladder": {"teacher_rung": rung, "student_rung": s_rung, "errors": lad.errors or None,
                       "side_channel": "gloo group" if lad.side is not None else None},
            "chunks": shards.chunks,
            "ranks": rank_diag,
            "kernel_ms_max": max(r_["kernel_ms"] for r_ in rank_diag), "kernel_ms_mean": float(np.mean([r_["kernel_ms"] for r_ in rank_diag])),
            "wall_ms_max": max(r_["wall_ms"] for r_ in rank_diag),
            "exchange_exposed_ms_max": max(r_["exchange_exposed_ms"] for r_ in rank_diag),
            "exchange_exposed_ms_mean": float(np.mean([r_["exchange_exposed_ms"] for r_ in rank_diag])),
            "what": (("all-gathers of the narrow side of each layer boundary: 100-wide aggregate of layer 1 (chunked, overlapped "
                      "with the aggregation; the projection is replicated and consumes chunks in arrival order)" if args.layer1_exchange == "narrow" else
                      "all-gathers: 256-wide fused output of layer 1 (chunked, overlapped with the aggregation; no replicated projection)"
                      if args.layer1_exchange == "wide" else
                      f"all-gathers: layer 1 mixed -- {args.mixed_fraction:g} of a rank's chunks as 256-wide fused output, the rest as 100-wide "
                      "aggregate with a replicated projection (chunked, overlapped)")
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
                    "gradient_exchange": None if world == 1 else ("one all-reduce after the backward" if "after" in (s_rung or "") and not args.student_global_bn
                                                                  else "weight gradients >= 1 MB all-reduced from inside the backward (grad_ready hook), the rest after it"),
                    "local_step_ms": student_local_ms,
                    "dp_overhead_ms": None if student_local_ms is None else 1e3 * t_student / k_student - student_local_ms,
                    "gflop_per_step": 3 * 2 * sd["batch"] * sum(a * b for a, b in zip(sd["dims"][:-1], sd["dims"][1:])) / 1e9},
    }
    result["student"]["tflops"] = result["student"]["gflop_per_step"] * k_student / t_student / 1e3      # per GPU
    result["student"]["frac_of_fp32_mfma_peak"] = result["student"]["tflops"] / 157.3      # v_mfma_f32_32x32x2_f32 dense peak
    if t_student != t_student:      # every gradient-exchange rung failed: null figures, the errors are under exchange.ladder
        for k_ in ("value", "ms_per_step", "tflops", "frac_of_fp32_mfma_peak"):
            result["student"][k_] = None

    if chunked is not None:
        result["chunked"] = chunked
    # ---- roofline of the dominant kernel (N = 1): per-launch HIP events from the timed region -------------
    if world == 1 and timing:
        full = C.GRAPH == "ogbn-products" and args.scale == 1.0
        result["roofline"] = roofline_object(timing, nnz, n, with_traffic=full)
        result["roofline"].update(hbm_estimate(result["roofline"], g))
        if args.reorder != "none":
            result["roofline_reordered"] = reordered_leg(args, g, feats, teacher, FullNeighborLoader, ops, data)

        if C.GRAPH == "ogbn-products" and not args.no_clustered_leg and args.locality == 0:
            result["roofline_clustered"] = clustered_leg(args, n, teacher, FullNeighborLoader, ops, data, dev)

    # ---- the B = 512 students of BASELINE configs[2] (latency-bound: steps/s, not an MFMA fraction) -- an extra object ----
    if world == 1 and not args.no_small_students:
        result["students_small"] = small_student_leg(dev, Model, StudentEngine, ops)

    # ---- sampled-block teacher TRAINING (SURVEY 8f rows 1+2; reference train_sage, train_and_eval.py:32-56) -- an extra object ----
    if world == 1 and not args.no_train_leg:
        result["teacher_training"] = teacher_training_leg(g, feats, labels, dev, data)

    # ---- CPU baseline on the host cores (oracle = 'port'; bounded sample) ---------------------------------
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sd, dev, min(C.CPU_SAMPLE_SCALE, args.scale), 1.0 if args.scale >= 1.0 else 0.1)

    # ---- BASELINE configs[1]+[2] and configs[4] on the same clock (N = 1, full-size products run): child processes of this one, after
    #      this process has released its device memory (the XL rank-forward keeps 228 GB resident).  A failed leg is reported, not fatal.
    if world == 1 and C.GRAPH == "ogbn-products" and args.scale == 1.0 and args.locality == 0:
        del g, feats, labels, out_t, teacher, student, eng, opt, perm, loader, shards
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if not args.no_arxiv_leg:
            result["arxiv"] = child_leg("arxiv", ["--workload", "arxiv", "--steps", "200", "--warmup", "5", "--student-steps-per-step", "10", "--no-cpu-baseline",
                                                   "--no-train-leg", "--no-small-students", "--reorder", "none"], 600)
        if not args.no_xl_leg:
            result["xl"] = child_leg("xl", ["--workload", "xl", "--steps", "5", "--warmup", "1"], 900)

    if ops.PLACEMENT_LOG:       # which allocations were tried for the gathered matrices, and the probe ms of each (set-up, outside every timed region)
        result["placement"] = list(ops.PLACEMENT_LOG)
    emit(result, args)
    if world > 1:
        dist.destroy_process_group()
