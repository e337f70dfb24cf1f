"""torch.autograd.Function shims over the C-ABI kernels, so that `loss.backward()` /
`optimizer.step()` in the reference's generic loops (train_and_eval.py:12-56: `train`, `train_sage`)
keep working on HIP: the dense projections and the neighbour aggregation run on libglnn_hip.so in both
directions.  (The student's hot loop does not go through autograd at all -- see student.py.)"""
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
                    db_buf = dy.sum(0)
            else:
                dw = ops.gemm_tn(dy, x, col_sum_a=db_buf)           # dy^T @ x -> [out,in]
            db = db_buf
        return dx, dw, db, None


def linear_fn(x, w, b, w_is_kn=False):
    return _LinearFn.apply(x, w, b, w_is_kn)


class SpmmFn(torch.autograd.Function):
    """Neighbour aggregation with a backward through the reversed graph (A^T dY): the teacher
    TRAINING direction (reference train_and_eval.py:12-56), same kernel on the transposed CSR."""

    @staticmethod
    def forward(ctx, graph, x, mode):
        ctx.graph, ctx.mode, ctx.n_src = graph, mode, x.shape[0]
        return ops.spmm(graph.indptr, graph.indices, x.detach(), graph.num_dst_nodes(), mode)

    @staticmethod
    def backward(ctx, dy):
        g = ctx.graph
        rev = g.reverse()
        dy = ops.as_feat(dy.contiguous())
        if ctx.mode == ops.AGG_SUM:
            dx = ops.spmm(rev.indptr, rev.indices, dy, ctx.n_src, ops.AGG_SUM)
        else:
            inv = 1.0 / (g.in_degrees().to(torch.float32) + 1.0)
            dx = ops.spmm(rev.indptr, rev.indices, dy, ctx.n_src, ops.AGG_SUM, col_scale=inv.contiguous())
            dx[: g.num_dst_nodes()] += dy * inv.unsqueeze(1)
        return None, dx, None


class _NormActDropFn(torch.autograd.Function):
    """dropout(relu(BatchNorm_train(z))) (or dropout(relu(z)) without a norm) as ONE differentiable op on the HIP path:
    glnn_bn_stats_f32 (batch statistics + running-stat update) -> glnn_act_fwd_f32; backward = glnn_bn_relu_bwd_f32 with the
    same counter-based dropout seed.  Replaces `self.norms[l](h)` -> `self.activation(h)` -> `self.dropout(h)` of the
    reference's training-mode forwards (models.py:48-52, 113-117)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, p, seed):
        z = ops.as_feat(z.detach())
        if bn is not None:
            mean, rstd, a_scale, a_shift = ops.bn_stats(z, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                                                        bn.num_batches_tracked, eps=bn.eps, momentum=bn.momentum)
            ctx.save_for_backward(z, gamma.detach(), mean, rstd, a_scale, a_shift)
        else:
            a_scale = a_shift = None
            ctx.save_for_backward(z)
        ctx.has_bn, ctx.p, ctx.seed = bn is not None, p, seed
        return ops.act_fwd(z, a_scale, a_shift, drop_p=p, drop_seed=seed)

    @staticmethod
    def backward(ctx, dy):
        dy = ops.as_feat(dy.contiguous())
        if ctx.has_bn:
            z, gamma, mean, rstd, a_scale, a_shift = ctx.saved_tensors
            dz, dgamma, dbeta = ops.bn_relu_bwd(dy, z, gamma, mean, rstd, a_scale, a_shift, drop_p=ctx.p, drop_seed=ctx.seed)
            return dz, dgamma, dbeta, None, None, None
        (z,) = ctx.saved_tensors
        dz, _, _ = ops.bn_relu_bwd(dy, z, drop_p=ctx.p, drop_seed=ctx.seed)
        return dz, None, None, None, None, None


_drop_counter = [0]


def norm_act_drop(z, bn, p):
    """Training-mode tail of a hidden layer.  bn: nn.BatchNorm1d (reference defaults) or None; p: dropout probability.
    The dropout stream is counter-based: seed = hash(torch.initial_seed(), call counter)."""
    _drop_counter[0] += 1
    seed = (int(torch.initial_seed()) * 0x9E3779B1 + _drop_counter[0] * 0x85EBCA77) & 0xFFFFFFFF if p > 0 else 0
    if bn is not None:
        return _NormActDropFn.apply(z, bn.weight, bn.bias, bn, float(p), seed)
    return _NormActDropFn.apply(z, None, None, None, float(p), seed)
