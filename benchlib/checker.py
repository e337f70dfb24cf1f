"""Self-checks of the timed outputs: the unfused / unsharded recomputation of the products forward, and CheckedBackend -- every launch of a forward re-derived in torch fp64 on a row sample (the verifier of the XL rank-forward)."""
import torch

from . import common as C



def verify_single(g, feats, teacher, out_timed, ops):
    """Self-check of the TIMED output (N = 1): the same forward recomputed WITHOUT the fused kernel, the chained projection and
    project-first -- every layer as stand-alone aggregation (spmm_csr_kernel) + GEMM, aggregate first as dgl does -- must agree
    within 1e-4; and the layer-1 aggregation satisfies the conservation identity
    sum_v (deg_v + 1) * mean_v == sum_u (outdeg_u + 1) * x_u in fp64."""
    enc = teacher.encoder
    n = g.n_dst
    with torch.no_grad():
        x = feats
        cons = None
        for l, layer in enumerate(enc.layers):
            es, eh, relu = enc._tail(l)
            agg = ops.spmm(g.indptr, g.indices, x, n, ops.AGG_SAGE_GCN)
            if l == 0:
                deg, outdeg = g.in_degrees().double(), g.out_degrees().double()
                d = x.shape[1]
                lhs = ((deg + 1).unsqueeze(1) * agg[:, :d].double()).sum(0)
                rhs = ((outdeg + 1).unsqueeze(1) * x[:, :d].double()).sum(0)
                cons = float((lhs - rhs).abs().max() / rhs.abs().max().clamp(min=1))
            x = ops.gemm(agg, layer.fc_neigh.weight, ep_scale=es, ep_shift=eh, relu=relu)
            del agg
        c = enc.layers[-1].fc_neigh.weight.shape[0]
        diff = float((x[:, :c] - out_timed[:, :c]).abs().max())
        finite = bool(torch.isfinite(out_timed[:, :c]).all())
    return {"ok": bool(finite and diff <= 1e-4 and cons < 1e-5), "max_abs_diff_vs_unfused_aggregate_first": diff, "tolerance": 1e-4,
            "layer1_conservation_rel_err_fp64": cons, "finite": finite, "rows_checked": n,
            "what": "timed output vs stand-alone aggregation + GEMM per layer (no fused kernel, no chained projection, aggregate-first)"}


def verify_sharded(out_own, ref_own, dev, dist):
    """Self-check of the TIMED output (N > 1): every rank's rows of the sharded forward vs the unsharded forward of the same
    rows computed on that rank before the graph was sharded; max over ranks."""
    c = ref_own.shape[1]
    d = (out_own[:, :c] - ref_own).abs().max() if out_own.numel() else torch.zeros((), device=dev)
    bad = (~torch.isfinite(out_own[:, :c])).any().float() if out_own.numel() else torch.zeros((), device=dev)
    t = torch.stack([d.double(), bad.double()])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    diff, nonfinite = float(t[0]), bool(t[1] > 0)
    return {"ok": bool(diff <= 1e-4 and not nonfinite), "max_abs_diff_vs_unsharded": diff, "tolerance": 1e-4, "finite": not nonfinite,
            "what": "each rank's rows of the sharded forward vs the unsharded forward of those rows (max over ranks)"}


class CheckedBackend:
    """A proxy of glnn_amd.ops for ONE verification forward of a sharded teacher (never inside a timed region): every aggregation /
    fused / GEMM launch runs as usual, then `sample` of its rows are (a) recomputed independently with torch index arithmetic in
    fp64 from the launch's own inputs -- mean = (sum_{e in row} x[src_e] + x_self) / (deg + 1), projections as fp64 matmuls, the
    epilogue per column -- and compared within `tol`, and (b) for stand-alone aggregations re-launched as a row range of their own,
    which must reproduce the rows bit for bit.  Works whatever filled the input buffers (real collectives, truth fills, synthetic
    fills): each launch is checked against ITS inputs."""

    def __init__(self, be, sample=4096, tol=1e-4, conservation=False):
        """conservation: stand-alone SAGE aggregations (no ReLU) are also held to the identity over ALL their rows, in fp64:
        sum_v (deg_v + 1) * mean_v == sum_u (edges of the launch out of u) * x_u + sum_v x_self_v  (a full pass over x per launch)."""
        self.be, self.sample, self.tol, self.report, self.ok, self.conservation = be, int(sample), tol, [], True, conservation

    def _conservation(self, indptr, indices, x, n_dst, xs, out, ep_scale, ep_shift, self_rows=None):
        d, dev = x.shape[1], x.device
        e0, e1 = int(indptr[0]), int(indptr[n_dst])
        cnt = torch.bincount(indices[e0:e1].long(), minlength=x.shape[0]).double()
        deg1 = (indptr[1:n_dst + 1] - indptr[:n_dst]).double() + 1
        lhs = torch.zeros(d, dtype=torch.float64, device=dev)
        rhs = torch.zeros(d, dtype=torch.float64, device=dev)
        step = 1 << 20                                         # fp64 reductions in slabs (bounded temporaries)
        for s0 in range(0, n_dst, step):
            sl = slice(s0, min(n_dst, s0 + step))
            y = out[sl, :d].double()
            if ep_shift is not None:
                y = y - ep_shift.double()
            if ep_scale is not None:
                y = y / ep_scale.double()
            lhs += (deg1[sl].unsqueeze(1) * y).sum(0)
            rhs += (xs[sl, :d] if self_rows is None else xs[self_rows[sl].long()][:, :d]).double().sum(0)
        for s0 in range(0, x.shape[0], step):
            sl = slice(s0, min(x.shape[0], s0 + step))
            rhs += (cnt[sl].unsqueeze(1) * x[sl, :d].double()).sum(0)
        return float((lhs - rhs).abs().max() / rhs.abs().max().clamp(min=1))

    def __getattr__(self, name):
        return getattr(self.be, name)

    def _range(self, n):
        k = min(self.sample, n)
        r0 = (n - k) // 2
        return r0, k

    def _agg_ref(self, indptr, indices, x, r0, k, mode, x_self, row_scale=None, col_scale=None):
        d = x.shape[1]
        e0, e1 = int(indptr[r0]), int(indptr[r0 + k])
        idx = indices[e0:e1].long()
        deg = indptr[r0 + 1:r0 + k + 1] - indptr[r0:r0 + k]
        dst = torch.repeat_interleave(torch.arange(k, device=x.device), deg)
        rows = x[idx][:, :d].double()
        if col_scale is not None:
            rows = rows * col_scale[idx].double().unsqueeze(1)
        acc = torch.zeros(k, d, dtype=torch.float64, device=x.device).index_add_(0, dst, rows)
        if mode == self.be.AGG_SAGE_GCN:
            return (acc + x_self[r0:r0 + k, :d].double()) / (deg.double() + 1).unsqueeze(1)
        return acc * row_scale[r0:r0 + k].double().unsqueeze(1) if row_scale is not None else acc

    def _agg_ref_rows(self, indptr, indices, x, r0, k, mode, self_sel, row_scale=None, col_scale=None):
        """The SAGE-gcn reference with the k self rows given directly (self_sel[i] = the self row of destination r0 + i)."""
        d = x.shape[1]
        acc = self._agg_ref(indptr, indices, x, r0, k, self.be.AGG_SUM, None, None, col_scale)
        deg = indptr[r0 + 1:r0 + k + 1] - indptr[r0:r0 + k]
        return (acc + self_sel[:, :d].double()) / (deg.double() + 1).unsqueeze(1)

    @staticmethod
    def _epi(y, ep_scale, ep_shift, relu):
        if ep_scale is not None:
            y = y * ep_scale.double()
        if ep_shift is not None:
            y = y + ep_shift.double()
        return y.clamp(min=0) if relu else y

    def _note(self, what, diff, exact=None):
        good = bool(diff <= self.tol) and (exact is None or exact)
        self.ok = self.ok and good
        self.report.append({"launch": what, "max_abs_diff_vs_fp64": diff, **({} if exact is None else {"row_range_relaunch_bit_equal": exact})})

    def spmm(self, indptr, indices, x, n_dst, mode, row_scale=None, col_scale=None, ep_scale=None, ep_shift=None, relu=False, out=None,
             x_self=None, self_rows=None, **kw):
        # (kw: the hub plan of the HIP backend.  The row-range relaunch below runs WITHOUT one: plan and no plan must agree bit for bit)
        out = self.be.spmm(indptr, indices, x, n_dst, mode, row_scale=row_scale, col_scale=col_scale, ep_scale=ep_scale, ep_shift=ep_shift,
                           relu=relu, out=out, x_self=x_self, self_rows=self_rows, **kw)
        chunks = kw.pop("chunks", None)
        if n_dst and chunks is not None:
            # ONE launch over chunks: self rows and output rows of row v sit at the chunk's rows of the whole buffers -- checked through
            # index vectors (the row-range relaunch and the conservation identity below then see ordinary tensors)
            v = torch.arange(n_dst, device=x.device)
            starts = torch.tensor([chunks.row_start[c] for c in range(chunks.n_chunks)], device=x.device)
            c_ = torch.searchsorted(starts, v, right=True) - 1
            if self_rows is None:
                self_rows = torch.tensor([chunks.self_row[i] for i in range(chunks.n_chunks)], device=x.device)[c_] + v - starts[c_]
            out_view = out[torch.tensor([chunks.out_row[i] for i in range(chunks.n_chunks)], device=x.device)[c_] + v - starts[c_]]
            self._check_spmm(indptr, indices, x, n_dst, mode, row_scale, col_scale, ep_scale, ep_shift, relu, out_view, x_self, self_rows, kw)
            return out
        self._check_spmm(indptr, indices, x, n_dst, mode, row_scale, col_scale, ep_scale, ep_shift, relu, out, x_self, self_rows, kw)
        return out

    def _check_spmm(self, indptr, indices, x, n_dst, mode, row_scale, col_scale, ep_scale, ep_shift, relu, out, x_self, self_rows, kw):
        if n_dst:
            xs = x if x_self is None else x_self
            r0, k = self._range(n_dst)
            if self_rows is not None:      # the self row of destination v is xs[self_rows[v]]: a view the reference below can index by v - r0
                sel = xs[self_rows[r0:r0 + k]]
                xs_again, sr = xs, self_rows[r0:r0 + k].contiguous()
                ref = self._epi(self._agg_ref_rows(indptr, indices, x, r0, k, mode, sel, row_scale, col_scale), ep_scale, ep_shift, relu)
            else:
                xs_again, sr = xs[r0:r0 + k], None
                ref = self._epi(self._agg_ref(indptr, indices, x, r0, k, mode, xs, row_scale, col_scale), ep_scale, ep_shift, relu)
            diff = float((out[r0:r0 + k].double() - ref).abs().max())
            again = self.be.spmm(indptr[r0:r0 + k + 1], indices, x, k, mode, row_scale=None if row_scale is None else row_scale[r0:r0 + k],
                                 col_scale=col_scale, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, x_self=xs_again, **({} if sr is None else {"self_rows": sr}))
            self._note(f"spmm d={x.shape[1]} rows={n_dst}", diff, bool(torch.equal(again, out[r0:r0 + k])))
            if self.conservation and mode == self.be.AGG_SAGE_GCN and not relu:
                err = self._conservation(indptr, indices, x, n_dst, xs, out, ep_scale, ep_shift, self_rows)
                self.report[-1]["conservation_rel_err_fp64_all_rows"] = err
                self.ok = self.ok and err < 1e-5
        return out

    def gemm(self, a, w, ep_scale=None, ep_shift=None, relu=False, out=None, **kw):
        out = self.be.gemm(a, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out, **kw)
        if not kw and a.shape[0]:
            r0, k = self._range(a.shape[0])
            ref = self._epi(a[r0:r0 + k].double() @ w.detach().double().t(), ep_scale, ep_shift, relu)
            self._note(f"gemm m={a.shape[0]} k={a.shape[1]} n={w.shape[0]}", float((out[r0:r0 + k].double() - ref).abs().max()))
        return out

    def sage_fused(self, indptr, indices, x, n_dst, w, ep_scale=None, ep_shift=None, relu=False, out=None, x_self=None, w_packed=None,
                   w_next=None, out_next=None, want_out=True, tile_order=None, **kw):
        res = self.be.sage_fused(indptr, indices, x, n_dst, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out, x_self=x_self,
                                 w_packed=w_packed, w_next=w_next, out_next=out_next, want_out=want_out, tile_order=tile_order, **kw)
        if n_dst:
            xs = x if x_self is None else x_self
            r0, k = self._range(n_dst)
            o, o2 = (res, None) if w_next is None else res
            chunks = kw.get("chunks")
            if chunks is not None:      # ONE launch over chunks: self rows and output rows of row v sit at the chunk's rows of the whole buffers
                v = torch.arange(r0, r0 + k, device=x.device)
                starts = torch.tensor([chunks.row_start[c] for c in range(chunks.n_chunks)], device=x.device)
                c = torch.searchsorted(starts, v, right=True) - 1
                self_idx = torch.tensor([chunks.self_row[i] for i in range(chunks.n_chunks)], device=x.device)[c] + v - starts[c]
                out_idx = torch.tensor([chunks.out_row[i] for i in range(chunks.n_chunks)], device=x.device)[c] + v - starts[c]
                agg = self._agg_ref_rows(indptr, indices, x, r0, k, self.be.AGG_SAGE_GCN, xs[self_idx])
                o, o2 = (None if o is None else o[out_idx]), (None if o2 is None else o2[out_idx])
            else:
                agg = self._agg_ref(indptr, indices, x, r0, k, self.be.AGG_SAGE_GCN, xs)
                o, o2 = (None if o is None else o[r0:r0 + k]), (None if o2 is None else o2[r0:r0 + k])
            h = self._epi(agg @ w.detach().double().t(), ep_scale, ep_shift, relu)
            diff = 0.0
            if o is not None:
                diff = float((o.double() - h).abs().max())
            if w_next is not None:
                diff = max(diff, float((o2.double() - h @ w_next.detach().double().t()).abs().max()))
            self._note(f"sage_fused d={x.shape[1]}->{w.shape[0]}" + (f"->{w_next.shape[0]}" if w_next is not None else "") + f" rows={n_dst}", diff)
        return res
