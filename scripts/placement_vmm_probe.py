"""does a DETERMINISTIC allocation make every gathered matrix 'fast'?  scripts/placement_cause_probe.py showed the fused
D=256 launch taking 18.0 or 19.5 ms depending on the PHYSICAL backing of the matrix it gathers from (same virtual addresses, other speed
after small allocations in between).  Here the 2.5 GB source matrix is allocated (a) by torch's caching allocator, (b) by hipMalloc
directly, (c) through the HIP virtual-memory API -- hipMemCreate (one physical handle for the whole matrix, or 2 MB / recommended-granularity
handles) + hipMemAddressReserve (alignment 2 MB / 1 GB) + hipMemMap + hipMemSetAccess -- each several times, on a fresh pool and after the
pool has been fragmented by 64 MB allocations; the launch is timed on every buffer.
   python scripts/placement_vmm_probe.py            -> table on stdout"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import data, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments"))
import vmm_alloc as vmm
dev = "cuda:0"
g = data.make_graph("ogbn-products", seed=0, device=dev, scale=1.0)
n = g.n_dst
w2 = torch.randn(256, 256, device=dev) / 16
w3 = torch.randn(47, 256, device=dev) / 16
order = g.fused_tile_order()
o47 = ops.feat_empty(n, 47, dev)
src = torch.randn(n, 256, device=dev).relu_()


def timed(x, reps=3):
    f = lambda: ops.sage_fused(g.indptr, g.indices, x, n, w2, relu=True, x_self=x, w_next=w3, out_next=o47, want_out=False, tile_order=order)
    f()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def show(tag, make, k=6):
    ms, keep = [], []
    for _ in range(k):
        x = make()
        x.copy_(src)
        ms.append(timed(x))
        keep.append(x)
    print(f"{tag:64s} " + " ".join(f"{t:6.2f}" for t in ms), flush=True)
    return keep


print("granularity (min, recommended):", vmm.granularity(0, False), vmm.granularity(0, True), flush=True)
for phase in ("fresh pool", "after 40 x 64 MB allocations (fragmented pool)"):
    print("----", phase, flush=True)
    spacers = [] if phase == "fresh pool" else [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(40)]
    if spacers:
        del spacers[::2]                       # free every second one: holes
    k1 = show("torch.empty (caching allocator)", lambda: torch.empty(n, 256, device=dev))
    del k1
    torch.cuda.empty_cache()
    k2 = show("hipMalloc (direct)", lambda: vmm.hipmalloc_tensor(n, 256, dev))
    del k2
    for chunk, align in ((0, 2 << 20), (0, 1 << 30), (2 << 20, 2 << 20), (256 << 20, 1 << 30)):      # (recommended granularity = 4 KB here: 610 k handles per matrix, not run)
        name = f"VMM: {'one handle' if chunk == 0 else ('recommended-granularity handles' if chunk < 0 else str(chunk >> 20) + ' MB handles')}, VA aligned {align >> 20} MB"
        try:
            k3 = show(name, lambda: vmm.vmm_tensor(n, 256, dev, chunk_bytes=chunk, va_align=align))
            del k3
        except Exception as e:
            print(f"{name:64s} failed: {e}", flush=True)
    del spacers
    torch.cuda.empty_cache()
