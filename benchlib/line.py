"""The two output lines of bench.py: the long DETAIL object, then ONE compact JSON line (<= 4 KB) -- the driver's record keeps only the tail of stdout."""
import json
import os

from . import common as C



def emit(result, args):
    """Print the long object first (one line, prefixed so that it is not mistaken for THE line), write it to the detail file, then
    ONE compact JSON line (<= 4 KB) last: the driver's record keeps only the tail of stdout."""
    path = args.detail_file
    if path is None and os.path.isdir(os.path.join(C.ROOT, "gpurun_out")):
        path = os.path.join(C.ROOT, "gpurun_out", "bench_detail.json")
    if path:
        try:
            with open(path, "w") as f:
                json.dump(result, f)
        except OSError:
            path = None
    print("DETAIL " + json.dumps(result), flush=True)
    print(json.dumps(compact(result, path)), flush=True)


def _short(v, n=160):
    """Strings longer than n are cut at the last word boundary in front of the limit (never mid-word)."""
    if not isinstance(v, str) or len(v) <= n:
        return v
    cut = v[:n - 3]
    sp = cut.rfind(" ")
    return (cut[:sp] if sp > n // 2 else cut) + "..."


def compact(r, detail_path):
    """The contract keys + the headline numbers of every object; long prose, per-launch lists and sweeps stay in the DETAIL line."""
    c = {k: r.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data", "verified", "rccl_ranks", "backend")}
    cfg = r.get("config", {})
    c["config"] = {k: _short(v, 200) for k, v in cfg.items() if k in ("workload", "nodes", "nnz", "graph_fingerprint", "edges_aggregated_per_step", "scale", "exchange",
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
            c["roofline"]["traffic_source"] = "static: " + C.PMC_FILE
        if "hbm_bytes_estimated" in rf:
            h = rf["hbm_bytes_estimated"]
            c["roofline"]["hbm_frac_bracket"] = [h["frac_of_peak_lower"], h["frac_of_peak_upper"]]
        if "all_aggregation_launches" in rf:
            c["roofline"]["launches"] = [{"d": l["d"], "ms": l["avg_ms"], "GBps": l["GBps"], "frac": round(l["GBps"] / C.HBM_PEAK_GBS, 4)}
                                         for l in rf["all_aggregation_launches"]]
            c["roofline"]["dense_ms"] = rf.get("dense_projection_ms_per_forward")
    for key in ("roofline_reordered", "roofline_clustered"):
        if key in r:
            c[key] = {"edges_per_s": r[key]["edges_per_s"], "ms_per_step": r[key]["ms_per_step"], "frac": r[key]["frac"]}
    if "layers" in r:
        c["layers"] = [{k: _short(l.get(k), 60) for k in ("layer", "bound", "ms", "achieved", "unit", "frac", "GB_received_per_rank", "emulated_fill_ms",
                                                           "modelled_link_ms") if l.get(k) is not None} for l in r["layers"]]
        c["per_forward"] = {k: r["per_forward"][k] for k in ("wall_ms", "kernel_ms", "emulated_fill_ms", "GB_received_per_rank")}
    if r.get("chunked"):      # the drop-in chunked sweep beside the whole-graph `value`
        c["chunked"] = {k: r["chunked"][k] for k in ("ms", "Gedges_per_s", "launches", "max_abs_diff_vs_whole_graph")}
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
        if "engine_alone_ms_per_step" in r["teacher_training"]:
            c["teacher_training"]["engine_alone_ms"] = round(r["teacher_training"]["engine_alone_ms_per_step"], 4)
    cb = r.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb["sample"], 200),
                             "student_steps_per_s": cb["student_steps_per_s"], "student_threads": cb["student_threads_best"]}
    if r.get("exchange"):
        ex = r["exchange"]
        c["exchange"] = {k: ex.get(k) for k in ("GB_received_per_rank_per_forward", "collectives_per_forward", "link_GBps_measured", "layer1_autotune_ms",
                                                "layer1_chosen", "chunks", "kernel_ms_max", "kernel_ms_mean", "wall_ms_max", "exchange_exposed_ms_max",
                                                "exchange_exposed_ms_mean")}
        if ex.get("ladder"):        # which rung of the fail-safe ladder ran, and what failed on the way down (strings cut to 100 chars)
            ld = ex["ladder"]
            c["exchange"]["ladder"] = {"teacher_rung": ld.get("teacher_rung"), "student_rung": ld.get("student_rung"),
                                       "errors": {k: _short(v, 100) for k, v in list((ld.get("errors") or {}).items())[:6]} or None}
            if ld.get("stage"):
                c["exchange"]["ladder"]["stage"] = ld["stage"]
        if isinstance(ex.get("layer1_autotune_ms"), dict):
            c["exchange"]["layer1_autotune_ms"] = {k: (_short(v, 60) if isinstance(v, str) else v) for k, v in ex["layer1_autotune_ms"].items()}
        if ex.get("link_probe"):
            c["exchange"]["link_probe_GBps"] = {k: [round(v["per_link_GBps"], 2), round(v["received_GBps"], 2)] for k, v in ex["link_probe"].items()}
        if ex.get("ranks"):       # per rank: [kernel ms, wall ms, exposed exchange ms]; per launch family: the slowest rank's ms
            c["exchange"]["ranks"] = [[round(q["kernel_ms"], 3), round(q["wall_ms"], 3), round(q["exchange_exposed_ms"], 3)] for q in ex["ranks"]]
            fam = {}
            for q in ex["ranks"]:
                for k, v in q["kernels"].items():
                    fam[k] = max(fam.get(k, 0.0), v)
            c["exchange"]["family_ms_max"] = {k: round(v, 3) for k, v in fam.items()}
    if r.get("error"):
        c["error"] = _short(r["error"], 200)
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
    if r.get("placement"):      # gathered matrices placed by probe: [what, best ms, worst ms] of the candidates tried (products graph only)
        c["placement"] = [[_short(q["what"], 40), min(q["ms"]), max(q["ms"])] for q in r["placement"][:4]]
    if "scale_model" in r:
        c["scale_model"] = {f: {"one_gpu_ms": o["one_gpu_forward_ms"],
                                **{N: {"max_kernel_ms": w["max_kernel_ms"], "GB": w["max_GB_received_per_rank"], "link_ms": w["modelled_link_ms"],
                                       "speedup": w["speedup_vs_one_gpu"]} for N, w in o["worlds"].items()}} for f, o in r["scale_model"].items()}
    c["detail"] = "the preceding stdout line (prefix 'DETAIL ')" + (f" and {os.path.relpath(detail_path, C.ROOT)}" if detail_path else "")
    return c
