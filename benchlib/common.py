"""Constants of the contract benchmark (bench.py), the workload switch (products | arxiv) and the small helpers every part uses."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")

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
XL_DIMS = [128, 256, 256, 47]      # 128-d features (BASELINE configs[4]); hidden 256 / 47 classes / BatchNorm as the products teacher (train.conf.yaml:196-204)
XGMI_LINK_GBS = (64.0, 77.0)       # effective one-direction rate of ONE xGMI link; 7 links per GPU, one per peer
PMC_FILE = os.path.join("profiles", "pmc_traffic.json")


def use_workload(name):
    """Switch the module-level workload description (read as `common.SAGE_DIMS` ... at call time by every part)."""
    global SAGE_DIMS, STUDENT, GRAPH, CPU_SAMPLE_SCALE
    if name == "arxiv":
        SAGE_DIMS, STUDENT, GRAPH, CPU_SAMPLE_SCALE = ARXIV["sage_dims"], ARXIV["student"], ARXIV["graph"], ARXIV["cpu_sample_scale"]



def agg_width(d_in, d_out):
    return d_out if d_in > d_out else d_in      # project-first when the layer narrows


def lanes_per_row(d):
    dv = (d + 3) // 4
    return 4 if dv <= 4 else 8 if dv <= 8 else 16 if dv <= 16 else 32 if dv <= 32 else 64


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


def kernel_breakdown_median(timing, steps):
    """As kernel_breakdown, but per launch family the MEDIAN over the forwards of the timed region instead of the mean: one stalled launch
    in one forward (seen twice in the emulation runs of round 5: a single 38 ms spmm among 0.6 ms ones) does not define a rank's time.
    Falls back to the mean when the records do not divide evenly into forwards."""
    import statistics
    if steps < 1 or len(timing) % steps:
        return kernel_breakdown(timing, steps)
    per_fwd = len(timing) // steps
    fams = [kernel_breakdown(timing[i * per_fwd:(i + 1) * per_fwd], 1) for i in range(steps)]
    return {k: statistics.median(f.get(k, 0.0) for f in fams) for k in fams[0]}

