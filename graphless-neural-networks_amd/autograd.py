"""torch.autograd.Function shims over the C-ABI kernels, so that `loss.backward()` /
`optimizer.step()` in the reference's generic loops (train_and_eval.py:12-56: `train`, `train_sage`)
keep working on HIP for callers that differentiate `Model.forward` themselves: the dense projections, the neighbour
aggregation and the norm/ReLU/dropout tails run on libglnn_hip.so in both directions.  The training loops of this package
do not go through autograd at all -- see student.py (StudentEngine) and teacher.py (TeacherEngine)."""
import torch

from . import ops


class _LinearFn(torch.autograd.Function):
    """y = x @ w.T (+ b) with w [out,in] (nn.Linear / fc_neigh layout), or y = x @ w (+ b) with w [in,out]
    (dgl GraphConv layout) when w_is_kn."""

    @staticmethod
    def forward(ctx, x, w, b, w_is_kn):
        x = ops.as_feat(x.detach())
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.kn = b is not None, w_is_kn
        return ops.gemm(x, w.detach(), w_is_kn=w_is_kn, ep_shift=None if b is None else b.detach())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = ops.as_feat(dy.contiguous())
        dx = dw = db = None
        n_out = w.shape[1] if ctx.kn else w.shape[0]
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy, w.detach(), w_is_kn=not ctx.kn)       # dy @ W ([out,in])  |  dy @ W^T ([in,out])
        if ctx.needs_input_grad[1] or ctx.has_bias:
            db_buf = torch.empty(n_out, dtype=torch.float32, device=w.device) if ctx.has_bias else None
            if ctx.kn:
                dw = ops.gemm_tn(x, dy)                             # x^T @ dy -> [in,out]
                if ctx.has_bias:
                    ops.col_sum(dy, out=db_buf)
            else:
                dw = ops.gemm_tn(dy, x, col_sum_a=db_buf)           # dy^T @ x -> [out,in]
            db = db_buf
        return dx, dw, db, None


def linear_fn(x, w, b, w_is_kn=False):
    return _LinearFn.apply(x, w, b, w_is_kn)


class SpmmFn(torch.autograd.Function):
    """Neighbour aggregation with its backward over the transposed graph (A^T dY on the SAME gather kernel, over
    glnn_csr_transpose): the teacher TRAINING direction (reference train_and_eval.py:12-56).
      AGG_SUM       y = row_scale * A (col_scale * x)          dx = col_scale * A^T (row_scale * dy)
      AGG_SAGE_GCN  y = (A x + x[:n_dst]) / (deg + 1)          dx = (A^T + I_dst) (dy / (deg + 1))"""

    @staticmethod
    def forward(ctx, graph, x, mode, row_scale=None, col_scale=None):
        ctx.graph, ctx.mode, ctx.n_src, ctx.rs, ctx.cs = graph, mode, x.shape[0], row_scale, col_scale
        return ops.spmm(graph.indptr, graph.indices, x.detach(), graph.num_dst_nodes(), mode, row_scale=row_scale, col_scale=col_scale)

    @staticmethod
    def backward(ctx, dy):
        if not ctx.needs_input_grad[1]:          # e.g. the outermost block: its input is feats[input_nodes]
            return None, None, None, None, None
        g = ctx.graph
        dy = ops.as_feat(dy.contiguous())
        if ctx.mode == ops.AGG_SUM:
            t = g.transposed(False)
            dx = ops.spmm(t.indptr, t.indices, dy, ctx.n_src, ops.AGG_SUM, row_scale=ctx.cs, col_scale=ctx.rs)
        else:
            t = g.transposed(True)
            dx = ops.spmm(t.indptr, t.indices, dy, ctx.n_src, ops.AGG_SUM, col_scale=g.inv_deg_plus1())
        return None, dx, None, None, None


def graphconv_fwd(g, a, w, b, relu):
    """dgl GraphConv(norm='both') forward on HIP (reference models.py:193): returns (y, mid, first).
    first (in > out): y = act(rs * A (cs * (a W)) + b), mid = a;  else: mid = rs * A (cs * a), y = act(mid W + b)."""
    rs, cs = g.degree_norms()
    n = g.num_dst_nodes()
    first = w.shape[0] > w.shape[1]
    if first:
        hw = ops.gemm(a, w, w_is_kn=True, row_scale=cs)
        return ops.spmm(g.indptr, g.indices, hw, n, ops.AGG_SUM, row_scale=rs, ep_shift=b, relu=relu), a, True
    mid = ops.spmm(g.indptr, g.indices, a, n, ops.AGG_SUM, row_scale=rs, col_scale=cs)
    return ops.gemm(mid, w, w_is_kn=True, ep_shift=b, relu=relu), mid, False


def graphconv_bwd(g, dz, mid, first, w, gw, want_da):
    """Backward of graphconv_fwd given dz = d/d(pre-activation): writes dW into gw, returns d/da (or None)."""
    rs, cs = g.degree_norms()
    n = g.num_dst_nodes()
    t = g.transposed(False)
    if first:
        dhw = ops.spmm(t.indptr, t.indices, dz, n, ops.AGG_SUM, row_scale=cs, col_scale=rs)      # cs * A^T (rs * dz)
        ops.gemm_tn(mid, dhw, out=gw)                                                       # a^T dhw -> [in, out]
        return ops.gemm(dhw, w) if want_da else None                                        # dhw W^T
    ops.gemm_tn(mid, dz, out=gw)
    if not want_da:
        return None
    return ops.spmm(t.indptr, t.indices, ops.gemm(dz, w), n, ops.AGG_SUM, row_scale=cs, col_scale=rs)


class GraphConvFn(torch.autograd.Function):
    """One dgl GraphConv(norm='both', activation=relu|None) layer as a differentiable op on the HIP path."""

    @staticmethod
    def forward(ctx, graph, feat, w, b, relu):
        a = ops.as_feat(feat.detach())
        y, mid, first = graphconv_fwd(graph, a, w.detach(), None if b is None else b.detach(), relu)
        ctx.graph, ctx.first, ctx.relu, ctx.has_bias = graph, first, relu, b is not None
        ctx.save_for_backward(mid, y, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        mid, y, w = ctx.saved_tensors
        dy = ops.as_feat(dy.contiguous())
        db = torch.empty(w.shape[1], dtype=torch.float32, device=w.device) if ctx.has_bias else None
        if ctx.relu:
            dz, _, _ = ops.bn_relu_bwd(dy, y, dz_col_sum=db)              # y = relu(z): y > 0 <=> z > 0
        else:
            dz = dy
            if db is not None:
                ops.col_sum(dz, out=db)
        gw = torch.empty_like(w)
        da = graphconv_bwd(ctx.graph, dz, mid, ctx.first, w.detach(), gw, ctx.needs_input_grad[1])
        return None, da, gw, db, None


class _NormActDropFn(torch.autograd.Function):
    """dropout(relu(BatchNorm_train(z))) (or dropout(relu(z)) without a norm) as ONE differentiable op on the HIP path:
    glnn_bn_stats_f32 (batch statistics + running-stat update) -> glnn_act_fwd_f32; backward = glnn_bn_relu_bwd_f32 with the
    same counter-based dropout seed.  Replaces `self.norms[l](h)` -> `self.activation(h)` -> `self.dropout(h)` of the
    reference's training-mode forwards (models.py:48-52, 113-117)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, p, seed, relu=True):
        z = ops.as_feat(z.detach())
        ctx.relu = relu
        ctx.layer_norm = isinstance(bn, torch.nn.LayerNorm)
        if ctx.layer_norm:
            y, mean, rstd = ops.layernorm_fwd(z, None if gamma is None else gamma.detach(), None if beta is None else beta.detach(),
                                              eps=bn.eps, relu=relu, drop_p=p, drop_seed=seed)
            ctx.save_for_backward(z, mean, rstd, *(t.detach() for t in (gamma, beta) if t is not None))
            ctx.has_bn, ctx.p, ctx.seed, ctx.affine = False, p, seed, gamma is not None
            return y
        if bn is not None:
            mean, rstd, a_scale, a_shift = ops.bn_stats(z, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                                                        bn.num_batches_tracked, eps=bn.eps, momentum=bn.momentum)
            ctx.save_for_backward(z, gamma.detach(), mean, rstd, a_scale, a_shift)
        else:
            a_scale = a_shift = None
            ctx.save_for_backward(z)
        ctx.has_bn, ctx.p, ctx.seed = bn is not None, p, seed
        return ops.act_fwd(z, a_scale, a_shift, drop_p=p, drop_seed=seed, relu=relu)

    @staticmethod
    def backward(ctx, dy):
        dy = ops.as_feat(dy.contiguous())
        if ctx.layer_norm:
            z, mean, rstd, *aff = ctx.saved_tensors
            gamma, beta = (aff + [None, None])[:2] if ctx.affine else (None, None)
            dz, dgamma, dbeta = ops.layernorm_bwd(dy, z, gamma, beta, mean, rstd, relu=ctx.relu, drop_p=ctx.p, drop_seed=ctx.seed)
            return dz, dgamma, dbeta, None, None, None, None
        if ctx.has_bn:
            z, gamma, mean, rstd, a_scale, a_shift = ctx.saved_tensors
            dz, dgamma, dbeta = ops.bn_relu_bwd(dy, z, gamma, mean, rstd, a_scale, a_shift, drop_p=ctx.p, drop_seed=ctx.seed, relu=ctx.relu)
            return dz, dgamma, dbeta, None, None, None, None
        (z,) = ctx.saved_tensors
        dz, _, _ = ops.bn_relu_bwd(dy, z, drop_p=ctx.p, drop_seed=ctx.seed, relu=ctx.relu)
        return dz, None, None, None, None, None, None


_drop_counter = [0]


def norm_act_drop(z, bn, p, relu=True):
    """Training-mode tail of a hidden layer: norm -> ReLU -> dropout (MLP / SAGE, reference models.py:48-52, 113-117) or, with
    relu=False, norm -> dropout (GCN, models.py:195-198: the ReLU sits inside the GraphConv).  bn: nn.BatchNorm1d (reference
    defaults), nn.LayerNorm, or None; p: dropout probability.
    The dropout stream is counter-based: seed = hash(torch.initial_seed(), call counter)."""
    _drop_counter[0] += 1
    seed = (int(torch.initial_seed()) * 0x9E3779B1 + _drop_counter[0] * 0x85EBCA77) & 0xFFFFFFFF if p > 0 else 0
    if bn is not None:
        return _NormActDropFn.apply(z, bn.weight, bn.bias, bn, float(p), seed, relu)
    return _NormActDropFn.apply(z, None, None, None, float(p), seed, relu)
