import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsparse.so"))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.sp_compress.argtypes = [vp, i64, i64, vp, vp, i64, vp]
lib.sp_gather.argtypes = [vp, vp, i64, vp, i64, vp, vp, i64, ctypes.c_int, vp]
dev = "cuda:0"
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = data.make_graph("ogbn-products", seed=0, device=dev, scale=scale)
n, nnz = g.n_dst, g.num_edges()
st = vp(torch.cuda.current_stream().cuda_stream)
for zero_frac in (0.5, 0.7, 0.9, 0.0):
    x = torch.randn(n + 1, 256, device=dev)
    x = torch.where(torch.rand_like(x) < zero_frac, torch.zeros_like(x), x.abs() + 0.1)
    meta = torch.zeros((n + 1) * 8 * 2, dtype=torch.int32, device=dev)
    packed = torch.zeros(n + 2, 256, device=dev)
    assert lib.sp_compress(x.data_ptr(), 256, n + 1, meta.data_ptr(), packed.data_ptr(), 256, st) == 0
    out_d = torch.empty(n, 256, device=dev); out_s = torch.empty(n, 256, device=dev)
    def run(sparse):
        src = packed if sparse else x
        assert lib.sp_gather(g.indptr.data_ptr(), g.indices.data_ptr(), n, src.data_ptr(), 256, meta.data_ptr(),
                             (out_s if sparse else out_d).data_ptr(), 256, 1 if sparse else 0, st) == 0
    def timeit(fn, reps=7):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
        return statistics.median(ts)
    td, ts_ = timeit(lambda: run(False)), timeit(lambda: run(True))
    tc = timeit(lambda: lib.sp_compress(x.data_ptr(), 256, n + 1, meta.data_ptr(), packed.data_ptr(), 256, st))
    err = float((out_d - out_s).abs().max())
    print(f"zeros {zero_frac:.1f}: dense gather {td:7.3f} ms | compressed gather {ts_:7.3f} ms | compress pass {tc:6.3f} ms | max|diff| {err:.3e} "
          f"| {nnz / ts_ / 1e6:.2f} vs {nnz / td / 1e6:.2f} G edges/s", flush=True)
