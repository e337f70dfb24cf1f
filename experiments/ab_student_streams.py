"""A/B of the student step's one-stream and two-stream backward (GLNN_STUDENT_TWO_STREAMS) inside one process, interleaved;
first checks that both engines, started from the same state, hold bit-identical parameters after 5 steps (same kernels, same
order of arithmetic -- only the streams differ).  usage: python scripts/ab_student_streams.py [config ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine

CONFIGS = {
    "products-MLP3w8": dict(dims=[100, 2048, 2048, 47], B=4096, p=0.2, n=400000),
    "products-MLP": dict(dims=[100, 256, 256, 47], B=4096, p=0.5, n=400000),
    "arxiv-MLP3w4": dict(dims=[128, 1024, 1024, 40], B=512, p=0.5, n=169343),
}
dev = "cuda:0"
for name in (sys.argv[1:] or list(CONFIGS)):
    c = CONFIGS[name]; d = c["dims"]
    feats = ops.as_feat(torch.randn(c["n"], d[0], device=dev))
    out_t = ops.as_feat(torch.log_softmax(torch.randn(c["n"], d[-1], device=dev), 1))
    nb = c["n"] // c["B"]
    perm = torch.randperm(c["n"])[: nb * c["B"]].view(nb, -1).to(dev)
    engs = {}
    for mode in ("0", "1"):
        os.environ["GLNN_STUDENT_TWO_STREAMS"] = mode
        torch.manual_seed(0)
        model = Model(dict(model_name="MLP", num_layers=len(d) - 1, feat_dim=d[0], hidden_dim=d[1], label_dim=d[-1],
                           dropout_ratio=c["p"], norm_type="batch", device=dev))
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        engs[mode] = (StudentEngine(model, opt, c["B"]), model)
        assert (engs[mode][0].aux_stream is not None) == (mode == "1")
    for i in range(5):
        for mode in ("0", "1"):
            engs[mode][0].step(feats, perm[i], ops.LOSS_KL, out_t, 1.0)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(engs["0"][1].state_dict().values(), engs["1"][1].state_dict().values()))
    res = {"0": [], "1": []}
    for rnd in range(3):
        for mode in ("0", "1"):
            e = engs[mode][0]
            for i in range(20):
                e.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(300):
                e.step(feats, perm[i % nb], ops.LOSS_KL, out_t, 1.0)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 300 * 1e3)
    gf = 3 * 2 * c["B"] * sum(a * b for a, b in zip(d[:-1], d[1:])) / 1e9
    t0, t1 = min(res["0"]), min(res["1"])
    print(f"{name:18s} one stream {t0:.4f} ms ({gf / t0 / 157.3 * 100 / 1e0 / 1e0:.1f} % of fp32 MFMA peak)   two streams {t1:.4f} ms "
          f"({gf / t1 / 157.3 * 100:.1f} %)   x{t0 / t1:.3f}   bit-identical after 5 steps: {same}", flush=True)
