"""Node-range sharding of the hot path over the GPUs of one node (one process per GPU, RCCL over xGMI
through torch.distributed; backend "nccl" IS RCCL on ROCm).  The reference has no distributed code at all
(SURVEY.md section 2), so everything here is new: SURVEY.md section 8e is the design brief.

Teacher (layer-wise full-neighbour inference, reference models.py:121-148)
  * rank r owns the contiguous destination rows [r*rpr, (r+1)*rpr) of every layer: their CSR rows, their
    output rows.  Every dst row is independent, so a layer needs no collective while it runs.
  * one exchange per layer: the own slab of the layer's output is written straight into the rank's slot of
    a full-size activation buffer and an in-place all-gather completes it for the next layer's gathers.
    Layer-1 input features are static -> replicated once, outside the timed loop.
  * layers with in > out project FIRST (dense, row-parallel, weights replicated) and all-gather the narrow
    projected rows, then aggregate: the last products layer moves 47 floats per node, not 256.
  * layers with 2*in <= out exchange the narrow AGGREGATE and replicate the projection: the first products layer
    moves 100 floats per node, not 256.  Per forward the exchange is N*(100+47)*4 = 1.44 GB instead of 2.97 GB.
Student (reference train_and_eval.py:59-86): data parallel, gradients summed with ONE all-reduce over a flat
  gradient buffer before the fused Adam launch (glnn_amd.student.StudentEngine(grad_sync=...)).

The compute backend is a parameter (`be`): production passes glnn_amd.ops (HIP); the world_size-2 gloo tests
pass a CPU stand-in with the same signatures so the sharding / exchange logic is exercised without a GPU."""
import torch
import torch.distributed as dist


class RowShards:
    """Equal contiguous row ranges; the last ranks may be short (buffers are padded to world*rpr rows)."""

    def __init__(self, n, world, rank):
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        self.rpr = (self.n + self.world - 1) // self.world
        self.lo = min(self.n, self.rank * self.rpr)
        self.hi = min(self.n, self.lo + self.rpr)
        self.rows = self.hi - self.lo
        self.n_pad = self.rpr * self.world


def _storage_rows(buf):
    """The contiguous [rows, ld] tensor behind a feature view [rows, d] (ld = row stride >= d)."""
    if buf.is_contiguous():
        return buf
    return torch.as_strided(buf, (buf.shape[0], buf.stride(0)), (buf.stride(0), 1))


def all_gather_rows(buf, shards, group=None):
    """In-place all-gather of the [n_pad, d] feature buffer `buf` (row stride ld) whose slot
    [rank*rpr, (rank+1)*rpr) this rank has filled.  The collective runs on the contiguous padded storage, whole
    rows including the [d, ld) padding columns, so that every rank's slab is ONE contiguous block."""
    if shards.world == 1:
        return buf
    base = _storage_rows(buf)
    mine = base[shards.rank * shards.rpr:(shards.rank + 1) * shards.rpr]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(base, mine, group=group)
    else:   # gloo (CPU tests): list form
        tmp = [torch.empty_like(mine) for _ in range(shards.world)]
        dist.all_gather(tmp, mine.contiguous(), group=group)
        for r, t in enumerate(tmp):
            base[r * shards.rpr:(r + 1) * shards.rpr].copy_(t)
    return buf


class ShardedTeacher:
    """SAGE layer-wise inference over a row-sharded graph.  `graph_shard` = full graph's rows [lo,hi)
    (glnn_amd.graph.CSRGraph.row_range), column indices global."""

    def __init__(self, encoder, graph_shard, shards, be, group=None):
        self.enc, self.g, self.sh, self.be, self.group = encoder, graph_shard, shards, be, group
        self._bufs = {}

    def _full_buffer(self, key, d, device):
        k = (key, d)
        if k not in self._bufs:
            self._bufs[k] = self.be.feat_empty(self.sh.n_pad, d, device, zero=True)
        return self._bufs[k]

    def forward(self, x_full):
        """x_full: [>= n, F] replicated input features.  Returns this rank's rows of the logits [rows, C]."""
        enc, sh, be, g = self.enc, self.sh, self.be, self.g
        x = be.as_feat(x_full)
        L = enc.num_layers
        y_own = None
        for l, layer in enumerate(enc.layers):
            w = layer.fc_neigh.weight
            ep_scale, ep_shift, relu = enc._tail(l)
            d_in, d_out = w.shape[1], w.shape[0]
            last = l == L - 1
            if d_in > d_out:
                # project own rows, exchange the narrow rows, aggregate own rows
                hw = self._full_buffer(("hw", l), d_out, x.device)
                be.gemm(x[sh.lo:sh.hi], w, out=hw[sh.lo:sh.hi])
                all_gather_rows(hw, sh, self.group)
                out = be.feat_empty(sh.rows, d_out, x.device) if last else self._full_buffer(("y", l), d_out, x.device)[sh.lo:sh.hi]
                be.spmm(g.indptr, g.indices, hw, sh.rows, be.AGG_SAGE_GCN, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu,
                        out=out, x_self=hw[sh.lo:sh.hi])
            elif sh.world > 1 and not last and 2 * d_in <= d_out:
                # widening layer (products layer 1: 100 -> 256): exchange the NARROW aggregate and let every rank
                # project all rows itself -- the all-gather moves d_in instead of d_out floats per node (0.98 GB
                # instead of 2.5 GB on products) for the price of a replicated [N, d_in] x [d_in, d_out] GEMM.
                agg = self._full_buffer(("agg", l), d_in, x.device)
                be.spmm(g.indptr, g.indices, x, sh.rows, be.AGG_SAGE_GCN, out=agg[sh.lo:sh.hi], x_self=x[sh.lo:sh.hi])
                all_gather_rows(agg, sh, self.group)
                y_full = self._full_buffer(("y", l), d_out, x.device)
                be.gemm(agg, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=y_full)
                x = y_full
                y_own = y_full[sh.lo:sh.hi]
                continue
            else:
                out = be.feat_empty(sh.rows, d_out, x.device) if last else self._full_buffer(("y", l), d_out, x.device)[sh.lo:sh.hi]
                if hasattr(be, "sage_fused") and d_in <= 256 and d_out <= 256:
                    be.sage_fused(g.indptr, g.indices, x, sh.rows, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out,
                                  x_self=x[sh.lo:sh.hi])
                else:
                    agg = be.spmm(g.indptr, g.indices, x, sh.rows, be.AGG_SAGE_GCN, x_self=x[sh.lo:sh.hi])
                    be.gemm(agg, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out)
            y_own = out
            if not last:
                x = all_gather_rows(self._full_buffer(("y", l), d_out, x.device), sh, self.group)
        return y_own


def make_grad_sync(flat_grads, world, group=None, average=False):
    """Gradient exchange for the data-parallel student: one all-reduce over the engine's flat grad buffer."""
    if world == 1:
        return None

    def sync():
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat_grads.mul_(1.0 / world)
    return sync
