"""bench.py contract (one JSON line, required keys) at a shrunken graph, for N=1 and for the N=2 launch line the driver
uses (`python -m torch.distributed.run ...`) with both ranks on cuda:0 over gloo (test-only knobs in bench.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]                  # THE line is the last one ...
    assert len(lines[0]) <= 4096, len(lines[0])                            # ... and fits the tail the driver's record keeps
    compact = json.loads(lines[0])
    detail = [ln for ln in r.stdout.splitlines() if ln.startswith("DETAIL {")]
    assert len(detail) == 1
    out = json.loads(detail[0][len("DETAIL "):])
    for k in REQUIRED:                                                     # the compact line carries the contract keys with the same values
        if k != "config":
            assert compact[k] == out[k], k
    assert compact["config"]["workload"] and compact.get("verified") == out.get("verified")
    out["_compact"] = compact
    return out


def test_bench_single_gpu_line():
    out = _run([sys.executable, "bench.py", "--scale", "0.02", "--steps", "2", "--warmup", "1"])
    assert REQUIRED <= out.keys()
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["unit"] == "edges/s" and out["value"] > 0
    assert "workload" in out["config"] and out["dtype"] == "f32" and out["vs_baseline"] is None
    rf, cb = out["roofline"], out["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1.2 and rf["unit"] == "GB/s"
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["aggregation_torch_sparse_csr_edges_per_s"] > 0 and cb["student_steps_per_s"] > 0 and "PyTorch CPU ops" in cb["student_kind"]
    assert out["student"]["value"] > 0
    tt = out["teacher_training"]
    assert tt["value"] > 0 and tt["unit"] == "steps/s" and all(np.isfinite(tt["loss_first_last"]))
    assert 0 < tt["engine_alone_ms_per_step"] <= 1.5 * tt["ms_per_step"]           # the engine's kernels without the sampler beside them
    rr = out["roofline_reordered"]
    assert rr["order"].startswith("nodes renumbered") and rr["edges_per_s"] > 0 and rr["traffic"] is None
    # the timed output checks itself: fused + chained + project-first forward == stand-alone aggregation + GEMM, aggregate-first
    assert out["verified"] is True and out["verify"]["max_abs_diff_vs_unfused_aggregate_first"] <= 1e-4
    assert out["verify"]["layer1_conservation_rel_err_fp64"] < 1e-5 and out["rccl_ranks"] == 1
    assert cb["student_threads_best"] >= 1 and len(cb["student_thread_sweep"]) >= 1 and cb["teacher_reps"] >= 3
    rc = out["roofline_clustered"]
    assert rc["edges_per_s"] > 0 and "communities" in rc["graph"]
    c = out["_compact"]      # what the driver's record keeps: the verdict of the self-check, the roofline, both baselines, the student figure
    assert c["verified"] is True and c["verify"]["max_abs_diff_vs_unfused_aggregate_first"] <= 1e-4
    assert c["roofline"]["frac"] == rf["frac"] and c["roofline"]["peak"] == 8000.0 and len(c["roofline"]["hbm_frac_bracket"]) == 2
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] == cb["value"] and c["cpu_baseline"]["cores"] == cb["cores"]
    assert c["student"]["frac_of_fp32_mfma_peak"] == out["student"]["frac_of_fp32_mfma_peak"] and c["student"]["ms_per_step"] > 0
    assert c["teacher_training"]["engine_alone_ms"] > 0
    assert c["teacher_training"]["steps_per_s"] == tt["value"] and set(c["students_small"]) == {"MLP", "MLP3w4", "products-MLP", "cora-MLP"}


@pytest.mark.parametrize("extra", [[], ["--student-global-bn"]])
def test_bench_two_ranks_driver_launch_line(extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1"] + extra,
               env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo"})
    assert REQUIRED <= out.keys()
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["student"]["global_batch"] == 2 * 4096
    ex = out["exchange"]       # per forward: n_pad * (100 + 48) floats = the narrow sides only (layer 2 exchanges nothing); wide: 256 + 48
    n_pad = -(-out["config"]["nodes"] // 8) * 8        # work-balanced (uneven) row ranges pad every slot to the longest range
    # round 5: the driver passes no flags -- the layer-1 exchange form is chosen by timing both forms on this transport, and the line says so
    l1 = out["config"]["layer1_exchange"]
    assert l1 in ("narrow", "wide", "mixed") and ex["layer1_chosen"] == l1 and ex["chunks"] in (2, 4, 8)
    assert set(ex["layer1_autotune_ms"]) == {f"{f}/{c}" for f in ("narrow", "wide", "mixed") for c in (2, 4, 8)} | {"mixed0.75/4", "mixed0.75/8"}   # form x chunks of the overlapped exchange
    best = min(ex["layer1_autotune_ms"], key=ex["layer1_autotune_ms"].get)
    form, ch = best.split("/")
    assert int(ch) == ex["chunks"] and l1 == ("mixed" if form.startswith("mixed") else form)
    wide_share = {"narrow": 0.0, "wide": 1.0, "mixed": round(0.5 * int(ch)) / int(ch), "mixed0.75": round(0.75 * int(ch)) / int(ch)}[form]
    per_node = wide_share * 256 + (1 - wide_share) * 100 + 48
    assert -0.02 * 4e-9 * n_pad * per_node <= ex["GB_received_per_rank_per_forward"] - 4e-9 * n_pad * per_node < 0.15 * 4e-9 * n_pad * per_node, ex
    assert ex["collectives_per_forward"] == 2 * ex["chunks"]            # layer 1's and layer 3's payload, chunk by chunk
    assert ("global" in out["student"]["batchnorm"]) == bool(extra)
    # ... and the record can diagnose itself: the measured link rate, and per rank kernel time vs wall time = the exposed exchange
    assert ex["link_GBps_measured"] > 0 and ex["link_probe"]["narrow"]["correct"] and ex["link_probe"]["wide"]["correct"]
    assert ex["link_probe"]["wide"]["slab_MB"] > 2.4 * ex["link_probe"]["narrow"]["slab_MB"]
    ranks = ex["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and sum(r["rows"] for r in ranks) == out["config"]["nodes"] and sum(r["nnz"] for r in ranks) == out["config"]["nnz"]
    for r in ranks:
        assert r["kernel_ms"] > 0 and r["wall_ms"] >= r["kernel_ms"] * 0.99 and abs(r["exchange_exposed_ms"] - (r["wall_ms"] - r["kernel_ms"])) < 1e-9
        assert abs(sum(r["kernels"].values()) - r["kernel_ms"]) < 1e-6 and any(k.startswith("sage_fused") for k in r["kernels"])
    assert ex["kernel_ms_max"] == max(r["kernel_ms"] for r in ranks) and ex["exchange_exposed_ms_max"] == max(r["exchange_exposed_ms"] for r in ranks)
    c = out["_compact"]["exchange"]
    assert c["layer1_chosen"] == l1 and len(c["ranks"]) == 2 and len(c["ranks"][0]) == 3 and c["link_GBps_measured"] == ex["link_GBps_measured"]
    assert set(c["family_ms_max"]) == set().union(*[r["kernels"].keys() for r in ranks])
    st = out["student"]
    if extra:
        assert st["local_step_ms"] is None and st["dp_overhead_ms"] is None
    else:
        assert st["local_step_ms"] > 0 and abs(st["dp_overhead_ms"] - (st["ms_per_step"] - st["local_step_ms"])) < 1e-9
        assert out["_compact"]["student"]["dp_overhead_ms"] == st["dp_overhead_ms"]
    assert len(json.dumps(out["_compact"])) <= 4096


def test_bench_two_ranks_survive_failing_collectives():
    """the first hardware `--gpus N` run must not be able to end without a line.  With every chunked / overlapped
    all-gather, the link probe and the in-backward gradient all-reduce made to raise (glnn_amd.dist.INJECT_FAIL through GLNN_BENCH_INJECT),
    the driver's own launch line still ends in ONE valid line: every autotune candidate is an error string, the teacher ran on the
    synchronous un-chunked all-gather (and verifies against the unsharded forward), the student on one all-reduce after the backward,
    and the line says which rungs ran and what failed above them."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1"],
               env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo", "GLNN_BENCH_INJECT": "async:1000000,grad_overlap:1000000,probe:1"})
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["verified"] is True and out["verify"]["max_abs_diff_vs_unsharded"] <= 1e-4
    ex = out["exchange"]
    assert all(isinstance(v, str) and v.startswith("error: ") and "injected" in v for v in ex["layer1_autotune_ms"].values()) and len(ex["layer1_autotune_ms"]) == 11
    lad = ex["ladder"]
    assert lad["teacher_rung"] == "synchronous un-chunked in-place all-gather" and lad["student_rung"] == "one all-reduce after the backward"
    assert "link probe" in lad["errors"] and "set-up [as configured]" in lad["errors"] and any(k.startswith("student") for k in lad["errors"])
    assert ex["link_probe"] is None and ex["link_GBps_measured"] is None and ex["chunks"] == 1 and out["config"]["layer1_exchange"] == "narrow"
    assert out["student"]["value"] > 0 and "one all-reduce after" in out["student"]["gradient_exchange"]
    c = out["_compact"]["exchange"]["ladder"]
    assert c["teacher_rung"] == lad["teacher_rung"] and c["student_rung"] == lad["student_rung"] and c["errors"]


def test_bench_two_ranks_fall_back_to_one_launch_per_chunk_when_the_signals_fail():
    """Round 6: the chunked layers are ONE launch with a completion signal per chunk (hipStreamWaitValue32 on the exchange stream).  If that
    fails on a node (injected: every wait raises), the ladder's next rung runs the same autotuned forms with one launch per chunk."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1"],
               env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo", "GLNN_BENCH_INJECT": "signal:1000000"})
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["verified"] is True
    lad = out["exchange"]["ladder"]
    assert lad["teacher_rung"] == "as configured, one launch per chunk (no completion signals)" and any("signal" in e for e in lad["errors"].values())
    assert all(isinstance(v, float) for v in out["exchange"]["layer1_autotune_ms"].values())      # (the second autotune, without signals: every form timed)


def test_bench_two_ranks_print_a_null_line_when_no_all_gather_works():
    """... and when every rung fails (every all-gather of the forward raises), the line carries value null and the errors instead of a number."""
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1"],
               env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo", "GLNN_BENCH_INJECT": "async:1000000,sync:1000000"})
    assert out["value"] is None and out["n_gpus"] == 2 and "every rung" in out["error"] and out["_compact"]["error"] == out["error"]
    errs = out["exchange"]["ladder"]["errors"]
    assert "timed region [synchronous un-chunked in-place all-gather]" in errs and "timed region [synchronous out-of-place (list form) all-gather]" in errs


@pytest.mark.parametrize("l1", ["narrow", "wide"])
def test_bench_self_launches_its_ranks(l1):
    """`python bench.py --gpus 2` with NO launcher: bench.py spawns the two ranks itself (it used to run one rank and
    print n_gpus 1); both layer-1 exchange variants; the sharded output verifies against the unsharded forward."""
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1", "--layer1-exchange", l1],
               env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert [d["rank"] for d in out["devices"]] == [0, 1]
    assert out["verified"] is True and out["verify"]["max_abs_diff_vs_unsharded"] <= 1e-4
    assert out["config"]["layer1_exchange"] == l1
    n_pad = -(-out["config"]["nodes"] // 8) * 8
    per_node = 148 if l1 == "narrow" else 256 + 48
    ex = out["exchange"]
    assert 0 <= ex["GB_received_per_rank_per_forward"] - 4e-9 * n_pad * per_node < 0.15 * 4e-9 * n_pad * per_node, ex


def test_bench_halo_exchange_with_partitioner_two_ranks():
    """bench.py --exchange halo on a clustered graph whose ids were shuffled, re-partitioned by label propagation (--partition lp),
    overlapped exchange: the sharded output verifies against the unsharded forward on the relabelled graph, and the halo moves
    fewer bytes than an all-gather of the same rows would (n_pad * 148 floats per rank and forward)."""
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--scale", "0.02", "--steps", "2", "--warmup", "1", "--exchange", "halo",
                "--locality", "0.95", "--shuffle-ids", "--partition", "lp"], env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["verified"] is True and out["config"]["partition"] == "lp" and out["config"]["halo_overlap"] is True
    assert out["config"]["partition_seconds"] > 0
    full = 4e-9 * out["config"]["nodes"] * 148 / 2          # what one rank would receive from the other under the all-gather
    assert out["exchange"]["GB_received_per_rank_per_forward"] < 0.8 * full, (out["exchange"], full)


@pytest.mark.parametrize("l1", ["narrow", "wide"])
def test_bench_xl_is_a_three_layer_forward_with_emulated_peers(l1):
    """--workload xl (BASELINE configs[4]) at a shrunken shard: one rank of the 8-way run, all three layers through ShardedTeacher with
    dist.EmulatedPeers; one object per layer; every launch of a verification forward checked against torch fp64 on a row sample."""
    out = _run([sys.executable, "bench.py", "--workload", "xl", "--scale", "0.002", "--steps", "2", "--warmup", "1", "--layer1-exchange", l1])
    assert out["verified"] is True and out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    cfg = out["config"]
    assert cfg["shards"] == 8 and cfg["rank_timed"] == 4 and cfg["nodes_total"] == 8 * cfg["rows_per_gpu"] and cfg["nnz_per_gpu"] == 20 * cfg["rows_per_gpu"]
    names = [l["layer"][0] for l in out["layers"]]
    assert names == (["1", "1", "2", "3", "3"] if l1 == "wide" else ["1", "1", "1", "2", "3", "3"])
    hbm = [l for l in out["layers"] if l.get("bound") == "hbm"]
    assert len(hbm) == 3 and all(l["ms"] > 0 and l["achieved"] > 0 for l in hbm)
    v = out["verify"]
    # (narrow: one aggregation launch over the 4 chunks + 4 replicated projections + one fused launch over the chunks + layer 3; wide: one launch per layer)
    assert v["repeat_forward_bit_equal"] and all(l["max_abs_diff_vs_fp64"] <= 1e-4 for l in v["launches"]) and len(v["launches"]) == (3 if l1 == "wide" else 7)
    assert all(l.get("row_range_relaunch_bit_equal", True) for l in v["launches"])
    pf = out["per_forward"]
    r4 = lambda d: (d + 3) // 4 * 4
    n_pad = 8 * 4 * (-(-(-(-cfg["rows_per_gpu"] // 4)) // 128) * 128)        # 4 chunks per slot, each a whole number of 128-row workgroups (RowShards.CHUNK_QUANTUM)
    want = 4e-9 * n_pad * ((256 if l1 == "wide" else 128) + r4(47))
    assert abs(pf["GB_received_per_rank"] - want) < 1e-6 * max(1.0, want) + 1e-9, (pf, want)
    assert len(out["_compact"]["layers"]) == len(out["layers"])


def test_bench_emulate_measures_every_rank_of_the_scaling_model():
    """--emulate 2,4: every rank of the 2- and 4-rank sharded forward timed on this one GPU with truth fills; every emulated rank's
    output equals the unsharded rows; the halo form on the re-partitioned clustered graph receives less than the all-gather."""
    out = _run([sys.executable, "bench.py", "--emulate", "2,4", "--scale", "0.02", "--steps", "2"])
    sm = out["scale_model"]
    assert set(sm) == {"allgather-narrow", "allgather-wide", "allgather-mixed0.5", "halo-lp"} and out["verified"] is True
    for form, o in sm.items():
        assert set(o["worlds"]) == {"2", "4"} and o["one_gpu_forward_ms"] > 0
        for N, w in o["worlds"].items():
            assert w["verified"] and len(w["ranks"]) == int(N) and w["max_kernel_ms"] > 0
            assert sum(r["rows"] for r in w["ranks"]) == o["nodes"] and sum(r["nnz"] for r in w["ranks"]) == o["nnz"]
    assert sm["halo-lp"]["worlds"]["4"]["max_GB_received_per_rank"] < sm["allgather-narrow"]["worlds"]["4"]["max_GB_received_per_rank"]
    assert sm["allgather-wide"]["worlds"]["4"]["max_GB_received_per_rank"] > sm["allgather-mixed0.5"]["worlds"]["4"]["max_GB_received_per_rank"] \
        > sm["allgather-narrow"]["worlds"]["4"]["max_GB_received_per_rank"]          # (mixed: half of the chunks travel wide)


def test_bench_xl_two_ranks_driver_launch_line():
    """--workload xl under the driver's launch line with 2 ranks (both on cuda:0 over gloo: test-only knobs): every rank generates its own
    shard, the input is replicated, the exchanges are REAL collectives (no emulation), weak scaling; every launch of the verification
    forward holds on both ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--workload", "xl", "--scale", "0.002", "--steps", "2", "--warmup", "1"],
               env={"GLNN_SINGLE_DEVICE": "1", "GLNN_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["verified"] is True
    cfg = out["config"]
    assert cfg["shards"] == 2 and cfg["nodes_total"] == 2 * cfg["rows_per_gpu"] and "EMULATED" not in cfg["parallelism"]
    assert out["per_forward"]["emulated_fill_ms"] is None and out["per_forward"]["collectives"] == 8
    assert abs(out["value"] - 2 * 3 * cfg["nnz_per_gpu"] * 2 / (out["ms_per_step"] * 2e-3)) < 1e-3 * out["value"]
