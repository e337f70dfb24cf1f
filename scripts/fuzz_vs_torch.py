"""Randomised check of the student step (default forms) against torch autograd in float64 on the same module: loss, logits and every
parameter gradient of one step, random shapes as in fuzz_small_step.py; dropout through the masks the engine's counter hash defines
(glnn_dropout_mask_u8 with the seeds of step 1)."""
import copy, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev = "cuda:0"


def run(seed=0, n_cases=40, verbose=True):
    rnd = random.Random(seed)
    out = []
    for case in range(n_cases):
        L = rnd.choice([1, 2, 3])
        feat = rnd.choice([7, 24, 50, 100, 128, 130, 257, 1433])
        hid = rnd.choice([8, 30, 64, 72, 100, 128, 200, 256, 260, 512])
        c = rnd.choice([2, 7, 40, 47, 64, 70])
        B = rnd.choice([2, 5, 31, 32, 33, 77, 140, 300, 512, 700, 1024, 1100, 4096])
        norm = rnd.choice(["batch", "none"])
        kind = rnd.choice(["kl", "nll"])
        lamb = rnd.choice([1.0, 0.3])
        p = rnd.choice([0.0, 0.0, 0.3, 0.5]) if L > 1 else 0.0
        torch.manual_seed(2000 + case)
        base = Model(dict(model_name="MLP", num_layers=L, feat_dim=feat, hidden_dim=hid, label_dim=c, dropout_ratio=p, norm_type=norm, device=dev))
        n = B + 17
        x = torch.randn(n, feat, device=dev)
        idx = torch.randperm(n, device=dev)[:B].contiguous()
        tgt = torch.randint(0, c, (n,), device=dev) if kind == "nll" else torch.log_softmax(torch.randn(n, c, device=dev), 1)
        m2 = copy.deepcopy(base); m2.train()
        eng = StudentEngine(m2, torch.optim.Adam(m2.parameters(), lr=0.01), B)
        eng.step_count = 1                                      # the seeds of the step about to run
        masks = [ops.dropout_mask(B, hid, p, eng._seed(l), dev).double() / (1.0 - p) for l in range(L - 1)] if p > 0 else None
        eng.step_count = 0
        # float64 reference on a copy of the module (training mode: batch statistics)
        ref = copy.deepcopy(base).double(); ref.train()
        h = x[idx].double()
        enc = ref.encoder
        for l, lay in enumerate(enc.layers):
            h = lay(h)
            if l != L - 1:
                if norm == "batch":
                    h = enc.norms[l](h)
                h = torch.relu(h)
                if masks is not None:
                    h = h * masks[l]
        logp = torch.log_softmax(h, 1)
        if kind == "nll":
            loss = torch.nn.functional.nll_loss(logp, tgt[idx])
        else:
            loss = torch.nn.functional.kl_div(logp, tgt[idx].double(), reduction="batchmean", log_target=True)
        (loss * lamb).backward()
        eng.step(ops.as_feat(x), idx, ops.LOSS_NLL if kind == "nll" else ops.LOSS_KL, tgt if kind == "nll" else ops.as_feat(tgt), lamb)
        torch.cuda.synchronize()
        err = abs(float(eng.loss_out) - float(loss.detach())) / max(1.0, abs(float(loss.detach())))
        err = max(err, float((eng.logits[:B, :c].double() - h.detach()).abs().max()) / (float(h.detach().abs().max()) + 1e-12))
        gscale = max(float(q.grad.abs().max()) for q in ref.parameters()) + 1e-12
        worst_t, worst_d = "", None
        for (nm, q), g in zip(ref.named_parameters(), eng.grads):
            e = float((g.double() - q.grad).abs().max()) / gscale
            if e > err:
                err, worst_t, worst_d = e, nm, g.double() - q.grad
        desc = f"case {case:3d}: L={L} dims {feat}-{hid}-{c} B={B} norm={norm} p={p} {kind} lamb={lamb}"
        # BatchNorm over a handful of rows is ill-conditioned in fp32 (two nearly equal samples: rstd ~ 1e3 and dz = dy - mean - xhat * ... cancels):
        # seen 3.3e-4 at B = 2; those cases are held to 2e-3
        tol = 2e-3 if (norm == "batch" and B < 8) else 1e-4
        # A weight-gradient difference that is rank 1 and confined to ONE row is one ReLU gate whose pre-activation lies within fp32 rounding of 0
        # (float64 opens it, fp32 closes it, or the other way round): a property of the input, not an arithmetic difference -- reported, not counted
        gate = None
        if err >= tol and worst_d is not None and worst_d.dim() == 2:
            sv = torch.linalg.svdvals(worst_d)
            rows = int((worst_d.abs().amax(1) > 0.05 * worst_d.abs().max()).sum())
            one_gate = rows == 1 and sv[1] < 1e-3 * sv[0]
            gate = (f"difference of {worst_t}: singular values {sv[0]:.2e} {sv[1]:.2e} {sv[2] if len(sv) > 2 else 0:.2e}"
                    + (" -- rank 1 in one row: ONE ReLU gate within rounding of 0, not an arithmetic difference" if one_gate else "")
                    + f"; rows above 5 % of its max: {rows} / {worst_d.shape[0]}")
            if one_gate:
                desc += f" [one ReLU gate at rounding: {err:.1e} in {worst_t}]"
                err = 0.0
        out.append((desc, err / tol * 1e-4))                     # normalised so that callers compare with 1e-4
        if verbose:
            print(f"{'ok ' if err < tol else 'BAD'} {desc}: max rel err vs float64 torch {err:.2e} {worst_t}", flush=True)
            if gate:
                print("      " + gate, flush=True)
    return out


if __name__ == "__main__":
    r = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    print(f"worst {max(e for _, e in r):.2e}")
