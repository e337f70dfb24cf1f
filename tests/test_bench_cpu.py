"""bench.py's pure-Python parts on CPU: the compact final line and the launch checker (round 4)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_compact_line_keeps_the_contract_and_the_headline_numbers_within_4_kb():
    """The driver's record keeps the TAIL of stdout: the last line must carry metric / value / verified / roofline / cpu_baseline /
    student of the run itself and stay below 4 KB whatever the detail object holds (round 3's 15 KB line lost them)."""
    import bench
    long = "x" * 3000
    r = {"metric": "m", "value": 1.0, "unit": "edges/s", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0, "higher_is_better": True,
         "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": True, "rccl_ranks": 1, "backend": None,
         "verify": {"ok": True, "max_abs_diff_vs_unfused_aggregate_first": 1e-6, "tolerance": 1e-4, "layer1_conservation_rel_err_fp64": 1e-9, "what": long},
         "config": {"workload": long, "nodes": 10, "nnz": 20, "graph": long, "parallelism": "1 GPU"},
         "roofline": {"bound": "hbm", "kernel": long, "achieved": 6800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.85, "traffic": 1.3e11,
                      "algorithmic_bytes_per_launch": 1.3e11, "avg_launch_ms": 19.0, "note": long,
                      "hbm_bytes_estimated": {"frac_of_peak_lower": 0.6, "frac_of_peak_upper": 0.9, "how": long},
                      "all_aggregation_launches": [{"d": d, "avg_ms": 1.0, "GBps": 5000.0, "kernel": long} for d in (47, 100, 256)],
                      "dense_projection_ms_per_forward": 0.0},
         "student": {"metric": long, "value": 1000.0, "unit": "steps/s", "ms_per_step": 0.93, "steps": 60, "tflops": 119.0, "frac_of_fp32_mfma_peak": 0.76},
         "students_small": [{"student": n, "ms_per_step": 0.06} for n in ("MLP", "MLP3w4", "products-MLP", "cora-MLP")],
         "teacher_training": {"value": 300.0, "ms_per_step": 3.3, "engine_alone_ms_per_step": 2.9, "metric": long},
         "cpu_baseline": {"value": 2.5e7, "unit": "edges/s", "cores": 128, "kind": "port", "sample": long, "student_steps_per_s": 3.5,
                          "student_threads_best": 16, "student_thread_sweep": [{"threads": t} for t in range(50)]},
         "roofline_reordered": {"edges_per_s": 1.0, "ms_per_step": 1.0, "frac": 0.8, "kernel": long},
         "roofline_clustered": {"edges_per_s": 1.0, "ms_per_step": 1.0, "frac": 1.0, "kernel": long},
         # round 5: BASELINE configs[4] and configs[1]+[2] ride on the default line (child processes of the products run)
         "xl": {"ms_per_step": 160.0, "value": 4.7e9, "verified": True, "per_forward": {"kernel_ms": 140.0}, "wall_s": 80.0, "verify": {"what": long},
                "config": {"rows_per_gpu": 12500000, "nnz_per_gpu": 250000000, "shards": 8, "workload": long},
                "layers": [{"layer": long, "kernel": long, "bound": "hbm", "ms": 24.0, "frac": 0.72}, {"layer": long, "GB_received_per_rank": 44.8},
                           {"layer": long, "bound": "mfma", "ms": 56.0, "frac": 0.74}, {"layer": long, "bound": "hbm", "ms": 46.0, "frac": 0.73},
                           {"layer": long, "GB_received_per_rank": 16.0}, {"layer": long, "bound": "hbm", "ms": 12.0, "frac": 0.55}]},
         "placement": [{"what": "SAGE.inference " + w, "rows": 2449029, "d": d, "ms": [18.5, 19.4, 18.1, 18.3, 18.6, 20.1, 18.2, 18.4], "chosen": 2}
                       for w, d in (("features", 100), ("y0", 256), ("proj1", 47), ("y0 reordered", 256), ("features clustered", 100))],
         "arxiv": {"ms_per_step": 0.89, "value": 8.4e9, "verified": True, "wall_s": 15.0, "config": {"nodes": 169343, "nnz": 2484941, "workload": long},
                   "student": {"metric": "student distill steps/s (MLP3w4 128-1024-1024-40, B=512 per rank, " + long, "ms_per_step": 0.135}}}
    c = bench.compact(r, None)
    line = json.dumps(c)
    assert len(line) <= 4096, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert c[k] == r[k]
    assert c["verified"] is True and c["verify"]["max_abs_diff_vs_unfused_aggregate_first"] == 1e-6
    assert c["roofline"]["frac"] == 0.85 and c["roofline"]["hbm_frac_bracket"] == [0.6, 0.9] and len(c["roofline"]["launches"]) == 3
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] == 128 and c["student"]["frac_of_fp32_mfma_peak"] == 0.76
    assert set(c["students_small"]) == {"MLP", "MLP3w4", "products-MLP", "cora-MLP"} and c["teacher_training"]["steps_per_s"] == 300.0 and c["teacher_training"]["engine_alone_ms"] == 2.9
    assert c["xl"]["ms"] == 160.0 and c["xl"]["verified"] is True and [l["frac"] for l in c["xl"]["layers"]] == [0.72, 0.74, 0.73, 0.55]
    assert len(c["placement"]) == 4 and c["placement"][1][1:] == [18.1, 20.1]
    assert c["arxiv"]["student"] == "MLP3w4" and c["arxiv"]["student_ms"] == 0.135 and c["arxiv"]["Gedges_per_s"] == 8.4
    r2 = dict(r, xl={"error": long, "wall_s": 1.0}, arxiv={"error": "rc 1: boom", "wall_s": 1.0})        # a failed leg is reported, the line survives
    c2 = bench.compact(r2, None)
    assert len(json.dumps(c2)) <= 4096 and "error" in c2["xl"] and c2["arxiv"]["error"] == "rc 1: boom" and c2["value"] == 1.0


def test_checked_backend_catches_a_wrong_launch_and_accepts_right_ones():
    """bench.CheckedBackend (the verifier of the XL forward): every launch re-derived in torch fp64 on a row sample, stand-alone
    aggregations re-launched as a row range and held to the fp64 conservation identity -- on the oracle-backed CPU stand-in: correct
    launches pass; a backend that drops one edge, or perturbs one GEMM element by 1e-3, is reported."""
    import bench
    from test_dist_cpu import OracleBackend
    from graphgen import random_graph
    n, d = 400, 12
    indptr, indices = random_graph(n, 8, seed=2, power=0.5, isolated=2, hub=150)
    ip, ix = torch.from_numpy(indptr), torch.from_numpy(indices)
    x = torch.from_numpy(np.random.RandomState(2).standard_normal((n, d)).astype(np.float32))
    w = torch.from_numpy(np.random.RandomState(3).standard_normal((7, d)).astype(np.float32))
    be = OracleBackend()
    chk = bench.CheckedBackend(be, sample=64, conservation=True)
    agg = chk.spmm(ip, ix, x, n, be.AGG_SAGE_GCN)
    chk.gemm(agg, w, ep_shift=torch.zeros(7), relu=True)
    assert chk.ok and len(chk.report) == 2 and chk.report[0]["row_range_relaunch_bit_equal"] and chk.report[0]["conservation_rel_err_fp64_all_rows"] < 1e-6

    class DropsAnEdge(OracleBackend):
        def spmm(self, indptr, indices, x, n_dst, mode, **kw):
            bad = indices.clone()
            e = int(indptr[n_dst // 2])                 # first edge of the middle row (inside the checker's sample)
            bad[e] = (int(bad[e]) + 1) % x.shape[0]
            return super().spmm(indptr, bad, x, n_dst, mode, **kw)

    c2 = bench.CheckedBackend(DropsAnEdge(), sample=n, conservation=True)
    c2.spmm(ip, ix, x, n, be.AGG_SAGE_GCN)
    assert not c2.ok

    class OffByALittle(OracleBackend):
        def gemm(self, a, w, **kw):
            y = super().gemm(a, w, **kw)
            y[a.shape[0] // 2, 0] += 1e-3
            return y

    c3 = bench.CheckedBackend(OffByALittle(), sample=n)
    c3.gemm(agg, w)
    assert not c3.ok and c3.report[0]["max_abs_diff_vs_fp64"] > 5e-4
