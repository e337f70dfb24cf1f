"""Extra objects of the N = 1 line: roofline of the dominant kernel, reordered / clustered graphs, small students, teacher training, and the child-process legs (xl, arxiv)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

from . import common as C



def child_leg(name, argv, timeout_s):
    """Run `python bench.py <argv>` as a child process and return its DETAIL object (+ wall seconds); {"error": ...} if it failed."""
    detail = os.path.join(C.ROOT, "gpurun_out", f"bench_detail_{name}.json") if os.path.isdir(os.path.join(C.ROOT, "gpurun_out")) else os.devnull
    t0 = time.perf_counter()
    try:
        p = subprocess.run([sys.executable, C.BENCH_PY] + argv + ["--detail-file", detail], capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in p.stdout.splitlines() if l.startswith("DETAIL {")]
        if p.returncode != 0 or not lines:
            return {"error": f"rc {p.returncode}: " + (p.stderr.strip().splitlines() or ["no output"])[-1][:300], "wall_s": time.perf_counter() - t0}
        r = json.loads(lines[-1][len("DETAIL "):])
        r["wall_s"] = time.perf_counter() - t0
        return r
    except Exception as e:      # timeout, unparsable output: the products line is still printed
        return {"error": f"{type(e).__name__}: {e}"[:300], "wall_s": time.perf_counter() - t0}


def chunked_leg(g, feats, teacher, FullNeighborLoader, ops, whole_out, forwards=3):
    """The DROP-IN form of the metric's forward: SAGE.inference(loader, feats, whole_graph=False) is the literal
    loop of reference models.py:133-145 -- per layer, per 4096-row chunk of the loader: block of the chunk's in-edges, gather the input
    rows, SAGEConv on the block, BN / ReLU epilogue, scatter the output rows -- and what every loader that is not an arange sweep gets.
    Same edges, same result (max |diff| against the whole-graph output of the timed region is reported); `launches` = library calls per
    forward (one batch of chunk blocks is built per sweep, the rest is per chunk)."""
    loader = FullNeighborLoader(g, 4096)
    enc = teacher.encoder
    out = enc.inference(loader, feats, whole_graph=False)       # warm-up: allocator, packed weights
    torch.cuda.synchronize()
    calls = []
    ops.set_timing(calls)
    enc.inference(loader, feats, whole_graph=False)
    torch.cuda.synchronize()
    ops.set_timing(None)
    t0 = time.perf_counter()
    for _ in range(forwards):
        out = enc.inference(loader, feats, whole_graph=False)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / forwards
    nnz = g.num_edges()
    diff = float((out[:whole_out.shape[0]] - whole_out).abs().max()) if whole_out is not None else None
    return {"what": "SAGE.inference(loader, feats, whole_graph=False): the reference's chunked sweep (models.py:133-145), "
                    f"{-(-g.n_dst // 4096)} chunks x {teacher.encoder.num_layers} layers",
            "ms": ms, "Gedges_per_s": 3 * nnz / ms / 1e6, "launches": len(calls), "forwards": forwards,
            "max_abs_diff_vs_whole_graph": diff}


def teacher_training_leg(g, feats, labels, dev, data):
    """Epochs of the reference's train_sage on the bench graph with the reference's config for it (train.conf.yaml:170-177 /
    196-204: fan-out 5,10,15; B=512 dropout 0.2 lr 0.01 on arxiv, B=4096 dropout 0.5 lr 0.003 on products): neighbour sampling
    and block building on the device (one batch ahead on a side stream), forward + NLL + backward + Adam on TeacherEngine."""
    from glnn_amd import train_and_eval as te
    from glnn_amd.graph import MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    prod = C.GRAPH == "ogbn-products"
    bsz, p, lr = (4096, 0.5, 0.003) if prod else (512, 0.2, 0.01)
    n = g.n_dst
    n_train = max(bsz, int(n * (196615 / 2449029 if prod else 90941 / 169343)))
    torch.manual_seed(0)
    model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=C.SAGE_DIMS[0], hidden_dim=C.SAGE_DIMS[1], label_dim=C.SAGE_DIMS[-1],
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
    # the same step WITHOUT the sampler beside it: one full batch stepped through the engine alone (what the kernels of a step cost;
    # the difference to ms_per_step is the sampler's kernels sharing the GPU from their side stream)
    from glnn_amd import teacher
    input_nodes, output_nodes, blocks = next(iter(loader))
    model.train()
    eng = teacher.get_engine(model, opt)
    for _ in range(3):
        eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(30):
        eng.step_sage(blocks, feats, labels, output_nodes, 1.0, input_nodes=input_nodes)
    torch.cuda.synchronize()
    engine_ms = 1e3 * (time.perf_counter() - t1) / 30
    return {"metric": f"sampled-block GraphSAGE training steps/s ({C.GRAPH}-shaped graph, fan-out 5,10,15, B={bsz}, dropout {p}, BN; "
                      "sampling + block building + forward + NLL + backward + Adam, all on the device)",
            "value": steps / dt, "unit": "steps/s", "steps": steps, "ms_per_step": 1e3 * dt / steps, "epoch_s": dt / epochs,
            "engine_alone_ms_per_step": engine_ms, "train_nodes": n_train, "loss_first_last": [losses[0], losses[-1]]}


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
        b = C.alg_bytes(nnz, n, d, (d_written if d_written is not None else d_out) if fused else None)     # bytes actually written per row
        avg = float(np.mean(ms))
        layers.append({"kernel": (f"sage_fused_kernel<LPR={C.lanes_per_row(((d + 7) // 8) * 8)},U={C.SPMM_U}> (aggregate {d} wide + project to {d_out} on MFMA"
                                  + (f", then to {d_chain} for the next layer: only those {d_written} floats per row are written)" if d_chain else ")")
                                  if fused else f"spmm_csr_kernel<LPR={C.lanes_per_row(d)},U={C.SPMM_U},SAGE_GCN>"),
                       "d": d, "avg_ms": avg, "alg_GB": b / 1e9, "GBps": b / avg / 1e6, "launches": len(ms),
                       "Gedges_per_s": nnz / avg / 1e6})
        tot_b += b
        tot_ms += avg
    gemm_ms = sum(float(np.mean(ms)) for (name, *_), ms in per.items() if name == "gemm")
    dom = max(layers, key=lambda r: r["avg_ms"])
    traffic = C.pmc_traffic(dom["kernel"].split(">")[0] + ">") if with_traffic else None
    return {
        "bound": "hbm", "kernel": f"{dom['kernel']} (D={dom['d']})",
        "achieved": dom["GBps"], "peak": C.HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / C.HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": None if traffic is None else f"{C.PMC_FILE} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                                       "committed; not measured by this run)",
        "algorithmic_bytes_per_launch": dom["alg_GB"] * 1e9, "avg_launch_ms": dom["avg_ms"],
        "all_aggregation_launches": sorted(layers, key=lambda r: r["d"]),
        "aggregation_total": {"alg_GB": tot_b / 1e9, "ms": tot_ms, "GBps": tot_b / tot_ms / 1e6, "frac": tot_b / tot_ms / 1e6 / C.HBM_PEAK_GBS},
        "dense_projection_ms_per_forward": gemm_ms,
        "note": "achieved = SURVEY 8(d) algorithmic bytes nnz*(4D+4)+n*(8D+8) / mean HIP-event launch time over the timed region"
                + ("" if C.GRAPH == "ogbn-products" else "; the feature matrix fits the 256 MB Infinity Cache at this shape, so the "
                   "rate is a cache rate and the HBM fraction is not meaningful (SURVEY 8d); the events are taken over a second pass of "
                   "the same forwards (inside the timed region they cost a sub-millisecond forward a third of its time), and a launch's "
                   "bracket includes what the event records themselves put in front of it"),
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
                                    "frac_of_peak_upper": upper / sec / (C.HBM_PEAK_GBS * 1e9), "frac_of_peak_lower": lower / sec / (C.HBM_PEAK_GBS * 1e9),
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
    feats2 = ops.as_feat(torch.randn(n, C.SAGE_DIMS[0], device=dev))
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
