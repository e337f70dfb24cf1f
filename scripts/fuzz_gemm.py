"""Randomised check of glnn_gemm_f32 / glnn_gemm_tn_f32 against float64 torch over shapes that hit every dispatch: aligned and unaligned
operands (widths that are not multiples of 4: the dword-loading latency kernel / the generic kernel), row gathers, operand transforms
with dropout masks taken from glnn_dropout_mask_u8, epilogue scale / shift / ReLU, both weight layouts, split reductions."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops
dev = "cuda:0"


def run(seed=0, n_cases=60, verbose=True):
    rnd = random.Random(seed)
    out = []
    for case in range(n_cases):
        m = rnd.choice([1, 7, 33, 140, 512, 1000, 2485, 4096])
        k = rnd.choice([4, 7, 47, 100, 128, 130, 256, 257, 1433])
        n = rnd.choice([1, 7, 40, 47, 64, 100, 128, 256, 260])
        kn = rnd.random() < 0.35
        gather = rnd.random() < 0.3
        xf = rnd.choice([0, 0, 1, 2]) if k % 4 == 0 else 0           # the operand transform needs float4 scale / shift vectors
        eps, relu, bias = rnd.random() < 0.3, rnd.random() < 0.3, rnd.random() < 0.7
        pad_a = rnd.random() < 0.8                                 # as_feat layout (rows on a 16-byte grid) or a raw matrix
        torch.manual_seed(seed * 1000 + case)
        na = m + 13 if gather else m
        a = torch.randn(na, k, device=dev)
        a_in = ops.as_feat(a) if pad_a else a
        w = torch.randn((k, n) if kn else (n, k), device=dev) * 0.2
        rows = torch.randint(0, na, (m,), device=dev) if gather else None
        sc = torch.rand(k, device=dev) + 0.5 if xf else None
        sh = torch.randn(k, device=dev) * 0.3 if xf else None
        p = 0.3 if xf == 2 else 0.0
        es = torch.rand(n, device=dev) + 0.5 if eps else None
        eh = torch.randn(n, device=dev) if bias else None
        try:
            y = ops.gemm(a_in, w, w_is_kn=kn, a_rows=rows, a_scale=sc, a_shift=sh, ep_scale=es, ep_shift=eh, relu=relu, drop_p=p, drop_seed=77,
                         workspace=torch.empty(1 << 22, device=dev))
        except Exception as e:                                      # unsupported combinations must say so, not return garbage
            out.append((f"case {case}: raised {type(e).__name__}: {e}", 0.0, "Unsupported" in str(e) or "must" in str(e) or "needs" in str(e)))
            continue
        ad = (a[rows] if gather else a).double()
        if xf:
            ad = torch.relu(ad * sc.double() + sh.double())
            if xf == 2:
                ad = ad * ops.dropout_mask(m, k, p, 77, dev).double() / (1.0 - p)
        ref = ad @ (w.double() if kn else w.double().t())
        if es is not None:
            ref = ref * es.double()
        if eh is not None:
            ref = ref + eh.double()
        if relu:
            ref = torch.relu(ref)
        err = float((y[:, :n].double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-9)
        desc = f"case {case:3d}: m={m} k={k} n={n} kn={kn} gather={gather} xf={xf} eps={eps} relu={relu} bias={bias} padded_a={pad_a}"
        out.append((desc, err, True))
        if verbose:
            print(f"{'ok ' if err < 2e-5 else 'BAD'} {desc}: rel err {err:.2e}", flush=True)
    return out


if __name__ == "__main__":
    r = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
    print(f"worst {max(e for _, e, _ in r):.2e}; not ok: {[d for d, e, ok in r if not ok]}")
