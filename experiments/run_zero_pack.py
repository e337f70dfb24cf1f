import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzeropack.so"))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.sp_compress.argtypes = [vp, i64, i64, vp, vp, i64, vp]
lib.sp_gather.argtypes = [vp, vp, i64, vp, i64, vp, vp, i64, ctypes.c_int, vp]
dev = "cuda:0"
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = data.make_graph("ogbn-products", seed=0, device=dev, scale=scale)
n, nnz = g.n_dst, g.num_edges()
st = vp(torch.cuda.current_stream().cuda_stream)
# forms: "side" = round 3 (header in a side array, rows 256 floats apart), "slot" = round 5 (header in the row's own 288-float slot)
for zero_frac, form, depth in ((0.5, "slot", 1), (0.5, "slot", 2), (0.5, "slot", 3), (0.5, "side", 1), (0.4, "slot", 2), (0.6, "slot", 2), (0.7, "slot", 2), (0.0, "slot", 2)):
    x = torch.randn(n + 1, 256, device=dev)
    x = torch.where(torch.rand_like(x) < zero_frac, torch.zeros_like(x), x.abs() + 0.1)
    ldp = 256 if form == "side" else 288
    meta = torch.zeros((n + 1) * 8 * 2, dtype=torch.int32, device=dev) if form == "side" else None
    mp = meta.data_ptr() if meta is not None else None
    packed = torch.zeros(n + 2, ldp, device=dev)
    assert lib.sp_compress(x.data_ptr(), 256, n + 1, mp, packed.data_ptr(), ldp, st) == 0
    out_d = torch.empty(n, 256, device=dev); out_s = torch.empty(n, 256, device=dev)
    def run(sparse):
        src = packed if sparse else x
        assert lib.sp_gather(g.indptr.data_ptr(), g.indices.data_ptr(), n, src.data_ptr(), ldp if sparse else 256, mp,
                             (out_s if sparse else out_d).data_ptr(), 256, depth if sparse else 0, st) == 0
    def timeit(fn, reps=7):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
        return statistics.median(ts)
    td, ts_ = timeit(lambda: run(False)), timeit(lambda: run(True))
    tc = timeit(lambda: lib.sp_compress(x.data_ptr(), 256, n + 1, mp, packed.data_ptr(), ldp, st))
    err = float((out_d - out_s).abs().max())
    print(f"zeros {zero_frac:.1f} {form} U={4 + 4 * depth}: ratio {ts_ / td:.3f} dense gather {td:7.3f} ms | compressed gather {ts_:7.3f} ms | compress pass {tc:6.3f} ms | max|diff| {err:.3e} "
          f"| {nnz / ts_ / 1e6:.2f} vs {nnz / td / 1e6:.2f} G edges/s", flush=True)
