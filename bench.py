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
reference train_and_eval.py:74-85), thread count stated.

This file is the entry point the driver calls: flags, process-group set-up, dispatch.  The parts live in benchlib/: products.py (the
products / arxiv workloads), xl.py (configs[4]), emulate.py (--emulate), legs.py (extra objects), checker.py (self-checks), cpu_baseline.py,
line.py (the DETAIL line + the compact final line), common.py (constants)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.checker import CheckedBackend  # noqa: E402,F401  (tests import these through `bench`)
from benchlib.emulate import run_emulated  # noqa: E402
from benchlib.line import compact, emit  # noqa: E402,F401
from benchlib.products import run_products  # noqa: E402
from benchlib.xl import run_xl  # noqa: E402


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
    ap.add_argument("--mixed-fraction", type=float, default=0.5, help="--layer1-exchange mixed: the fraction of every chunk's rows that travels wide")
    ap.add_argument("--emulate-forms", default=None, help="--emulate: comma-separated subset of narrow,wide,mixed,halo-lp (default: all)")
    ap.add_argument("--layer1-exchange", default="auto", choices=["auto", "narrow", "wide", "mixed"],
                    help="N > 1, all-gather exchange: what the widening first layer (100 -> 256) puts on the wire -- its 100-wide aggregate "
                         "(every rank projects all rows itself) or its 256-wide fused output (no replicated work, 2.56x the bytes).  auto (default): "
                         "the forms (narrow, wide, mixed at --mixed-fraction) are timed on this job's transport before the timed region (3 forwards each) and the fastest one runs; "
                         "--workload xl and N = 1 treat auto as narrow")
    ap.add_argument("--chunks", type=int, default=0, help="N > 1 / --emulate: chunks of the overlapped all-gather exchange (0 = 4, or chosen by the "
                    "layer-1 autotune among 2, 4, 8)")
    ap.add_argument("--no-verify", action="store_true", help="skip the self-check of the timed output ('verified' on the JSON line)")
    ap.add_argument("--no-clustered-leg", action="store_true", help="N = 1: skip roofline_clustered (the forward on a graph with communities)")
    ap.add_argument("--no-chunked-leg", action="store_true", help="N = 1: skip `chunked` (the reference's chunk-by-chunk sweep, SAGE.inference(whole_graph=False), timed beside the whole-graph form)")
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
    ap.add_argument("--deadline-s", type=float, default=1500.0, help="N > 1: rank 0 prints a line with value null and exits when the run has not finished after this many "
                    "seconds (a collective that hangs cannot be caught as an exception; the RCCL timeout is set behind it)")
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

    # test-only knobs: GLNN_SINGLE_DEVICE=1 maps every rank to cuda:0 and GLNN_DIST_BACKEND=gloo swaps the transport, so
    # that the N > 1 code path can be smoke-tested on a 1-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("GLNN_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GLNN_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    args.side_group, args.ladder = None, None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=args.deadline_s + 120)      # behind the deadline: rank 0's own null line comes first
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        # the side channel of the fail-safe ladder (benchlib.products.Ladder): success / failure of every transport stage is agreed on over
        # gloo, not over the communicator that may just have failed
        try:
            args.side_group = dist.new_group(backend="gloo", timeout=tmo)
        except Exception as e:      # noqa: BLE001
            print(f"bench.py: no gloo side group ({type(e).__name__}: {e}); failures will not be agreed on across ranks", file=sys.stderr, flush=True)
        if rank == 0 and args.workload == "products" and not args.emulate:
            import threading
            from benchlib.products import emit_failure

            def watchdog():
                emit_failure(args, 0, world, args.ladder, f"deadline of {args.deadline_s:.0f} s exceeded (a collective that never returns?)")
                os._exit(3)
            wd = threading.Timer(args.deadline_s, watchdog)
            wd.daemon = True
            wd.start()

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
    if world == 1:
        return run_products(args, rank, world, dev, barrier)
    try:
        return run_products(args, rank, world, dev, barrier)
    except Exception as e:      # noqa: BLE001  (N > 1: whatever still escapes the ladder ends in a line, not in a traceback only)
        import traceback
        from benchlib.products import emit_failure
        traceback.print_exc()
        emit_failure(args, rank, world, args.ladder, f"{type(e).__name__}: {e}"[:300])
        sys.exit(0 if rank == 0 else 1)


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


if __name__ == "__main__":
    main()
