"""Times glnn_classifier_loss_f32 alone on the MLP3w8 classifier shape (development aid; GLNN_LIB_PATH selects a library build):
   python scripts/cls_probe.py [rows k c p]  ->  us per launch for {transform, plain operand} x {with loss, logits only}
   (back-to-back launches on one stream, wall time / n: no per-launch events)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops

rows, k, c, p = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (4096, 2048, 47, 0.2)
dev = "cuda:0"
torch.manual_seed(0)
z = torch.randn(rows, k, device=dev)
sc = torch.rand(k, device=dev) + 0.5
sh = torch.randn(k, device=dev) * 0.3
w = torch.randn(c, k, device=dev) / k ** 0.5
b = torch.randn(c, device=dev) * 0.1
t = ops.as_feat(torch.log_softmax(torch.randn(rows, c, device=dev), 1))
logits = torch.empty(rows, c, device=dev); dl = torch.empty(rows, c, device=dev); lo = torch.empty(1, device=dev); ws = torch.empty(1024, device=dev)
act = ops.act_fwd(z, sc, sh, p, 7)
def run(xf, loss):
    kw = dict(a_scale=sc, a_shift=sh, drop_p=p, drop_seed=7) if xf else {}
    if xf == 1:
        kw["drop_p"] = 0.0
    ops.classifier_loss(z if xf else act, w, b, ops.LOSS_KL if loss else -1, 1.0, target_logp=t if loss else None, logits=logits,
                        dlogits=dl if loss else None, loss_out=lo if loss else None, workspace=ws if loss else None, **kw)
tag = os.environ.get("GLNN_LIB_PATH", "in-tree")
one = os.environ.get("CLS_PROBE_ONE")          # "xf loss": that combination only, 300 launches (under rocprofv3: scripts/cls_prof.sh)
if one:
    xf, loss = int(one.split()[0]), bool(int(one.split()[1]))
    z2 = z.clone()
    for _ in range(300):
        z.copy_(z2)                            # z as "fresh" as the producing GEMM leaves it
        run(xf, loss)
    torch.cuda.synchronize()
    sys.exit(0)
for xf in (2, 1, 0):
    for loss in (True, False):
        for _ in range(50):
            run(xf, loss)
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            run(xf, loss)
        torch.cuda.synchronize()
        print(f"{tag} {rows}x{k}x{c} p={p} xf={xf} loss={int(loss)}: {(time.perf_counter() - t0) / n * 1e6:.1f} us")
