import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsplit.so"))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.sp_gather100.argtypes = [vp, vp, i64, vp, i64, vp, vp, i64, ctypes.c_int, vp]
dev = "cuda:0"
g = data.make_graph("ogbn-products", seed=0, device=dev)
n, nnz = g.n_dst, g.num_edges()
st = vp(torch.cuda.current_stream().cuda_stream)
x = torch.randn(n, 100, device=dev)
main = torch.zeros(n, 96, device=dev); main.copy_(x[:, :96])
rem = x[:, 96:].contiguous()
out_a = torch.empty(n, 100, device=dev); out_b = torch.empty(n, 100, device=dev)
def run(split):
    if split: rc = lib.sp_gather100(g.indptr.data_ptr(), g.indices.data_ptr(), n, main.data_ptr(), 96, rem.data_ptr(), out_b.data_ptr(), 100, 1, st)
    else: rc = lib.sp_gather100(g.indptr.data_ptr(), g.indices.data_ptr(), n, x.data_ptr(), 100, None, out_a.data_ptr(), 100, 0, st)
    assert rc == 0
def timeit(fn, reps=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return statistics.median(ts)
for _ in range(2):
    ta, tb = timeit(lambda: run(False)), timeit(lambda: run(True))
    print(f"D=100 gather: [N][100] {ta:.3f} ms | split 96+4 {tb:.3f} ms | max|diff| {float((out_a - out_b).abs().max()):.1e} | "
          f"{nnz / ta / 1e6:.2f} vs {nnz / tb / 1e6:.2f} G edges/s", flush=True)
