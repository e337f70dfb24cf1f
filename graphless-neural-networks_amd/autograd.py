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
