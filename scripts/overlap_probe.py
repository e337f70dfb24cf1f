"""Do the MFMA-bound and the HBM-bound launches of the MLP3w8 backward tail overlap when they are issued on two streams?
   serial (today):  dW2 -> dA1 -> bn_bwd(L1) -> dW1 -> Adam
   forked:          main: dA1 -> bn_bwd(L1) -> dW1 ; side: dW2 ; join ; Adam         (development aid, round 5)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
B, F, H = 4096, 100, 2048
dz2 = torch.randn(B, H, device=dev); h1 = torch.randn(B, H, device=dev); w2 = torch.randn(H, H, device=dev) * 0.02
z1 = torch.randn(B, H, device=dev); x = ops.as_feat(torch.randn(B, F, device=dev))
gamma = torch.rand(H, device=dev) + 0.5; mean = z1.mean(0); rstd = 1.0 / z1.std(0); a_scale = gamma * rstd; a_shift = -mean * a_scale
dw2 = torch.empty(H, H, device=dev); da1 = torch.empty(B, H, device=dev); dz1 = torch.empty(B, H, device=dev)
dw1 = torch.empty(H, F, device=dev); dg = torch.empty(H, device=dev); db = torch.empty(H, device=dev)
ws_a = torch.empty(1 << 24, device=dev); ws_b = torch.empty(1 << 24, device=dev); ws_bn = torch.empty((3 * 32 + 2) * H + 1024, device=dev)
params = [torch.randn(H, H, device=dev), torch.randn(H, F, device=dev)]
side = torch.cuda.Stream()
ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()

def main_chain():
    ops.gemm(dz2, w2, w_is_kn=True, out=da1)                                   # dA1 = dz2 . W2
    ops.bn_relu_bwd(da1, z1, gamma, mean, rstd, a_scale, a_shift, dz=dz1, dgamma=dg, dbeta=db, workspace=ws_bn, drop_p=0.2, drop_seed=7)
    ops.gemm_tn(dz1, x, out=dw1, workspace=ws_a)                               # dW1

def side_chain():
    ops.gemm_tn(dz2, h1, out=dw2, workspace=ws_b)                              # dW2

def tail():
    params[0].add_(dw2, alpha=-1e-3)                                           # (stand-in for Adam: an HBM-bound pass over W2)

def serial():
    side_chain(); main_chain(); tail()

def forked():
    cur = torch.cuda.current_stream()
    ev_fork.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(ev_fork)
        side_chain()
        ev_join.record(side)
    main_chain()
    cur.wait_event(ev_join)
    tail()

def forked_side_first_main():
    """the big dW2 on the main stream, the dependent chain on the side stream"""
    cur = torch.cuda.current_stream()
    ev_fork.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(ev_fork)
        main_chain()
        ev_join.record(side)
    side_chain()
    cur.wait_event(ev_join)
    tail()

def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for rnd in range(3):
    print(f"round {rnd}: serial {timeit(serial):.1f} us   forked {timeit(forked):.1f} us   forked(chain on side) {timeit(forked_side_first_main):.1f} us   "
          f"[parts: dW2 {timeit(side_chain):.1f}, chain {timeit(main_chain):.1f}, tail {timeit(tail):.1f}]", flush=True)
