"""Teacher TRAINING step on libglnn_hip.so: forward + NLL + backward + Adam as an explicit launch sequence.

One `TeacherEngine.step_sage()` is the loop body of the reference's `train_sage` (reference train_and_eval.py:39-54)

    blocks = [blk.int().to(device) ...]; batch_feats = feats[input_nodes]; logits = model(blocks, batch_feats)
    out = logits.log_softmax(1); loss = criterion(out, labels[output_nodes]); total += loss.item()
    loss *= lamb; optimizer.zero_grad(); loss.backward(); optimizer.step()

and one `step_gcn()` is the reference's full-graph `train` (train_and_eval.py:12-29), with NO autograd graph and NO host
synchronisation:

  SAGE layer l   agg_l = (A_l h_l + h_l[:n_dst]) / (deg+1)          glnn_spmm_csr_f32 (SAGE_GCN); for the OUTERMOST block the
                                                                    gather goes straight into `feats` through the block's
                                                                    global source ids (feats[input_nodes] is never built)
                 z_l = agg_l W_l^T + b_l                            glnn_gemm_f32
                 h_{l+1} = dropout(relu(BN_train(z_l)))             glnn_bn_stats_f32 + glnn_act_fwd_f32
  loss           log_softmax + NLL, d(lamb*loss)/dlogits            glnn_softmax_loss_f32 (labels indexed by output_nodes)
  backward       dW_l = dz_l^T agg_l (+ db_l)                       glnn_gemm_tn_f32
                 dagg_l = dz_l W_l                                  glnn_gemm_f32
                 dh_l = (A_l^T + I_dst) (dagg_l / (deg+1))          glnn_spmm_csr_f32 over glnn_csr_transpose(add_self)
                 dz_{l-1} = BN / ReLU / dropout backward            glnn_bn_relu_bwd_f32 (also yields db_{l-1}, dgamma, dbeta)
  update         one fused multi-tensor Adam launch                 glnn_adam_step_f32

  GCN layer      GraphConv(norm='both'): weight first iff in > out (dgl), D_out^-1/2 / D_in^-1/2 folded into the GEMM row
                 scale and the aggregation's row / column scales, bias + ReLU in the epilogue; backward over the transposed
                 graph with the two scales swapped.

The engine works IN PLACE on the Model's parameters / BatchNorm buffers and on the torch optimizer's state tensors, exactly
like glnn_amd.student.StudentEngine, so state_dict(), early-stopping snapshots and optimizer.state_dict() keep working."""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .autograd import graphconv_bwd, graphconv_fwd
from .student import _mix32


def _is_relu(act):
    return act is F.relu or getattr(act, "__name__", "") == "relu"


def check_supported(model, criterion, optimizer):
    """Raise unless (model, criterion, optimizer) is what train_teacher.py:232-238 builds for a SAGE / GCN teacher."""
    enc = model.encoder
    name = model.model_name
    if "MLP" in name or not ("SAGE" in name or "GCN" in name):
        raise NotImplementedError(f"TeacherEngine: SAGE or GCN teachers only (got {name})")
    if not (isinstance(criterion, nn.NLLLoss) and criterion.reduction == "mean" and criterion.weight is None
            and criterion.ignore_index == -100):
        raise NotImplementedError("TeacherEngine: the criterion must be nn.NLLLoss() (reference train_teacher.py:237)")
    grp = optimizer.param_groups
    if type(optimizer) is not torch.optim.Adam or len(grp) != 1 or grp[0].get("amsgrad") or grp[0].get("maximize"):
        raise NotImplementedError("TeacherEngine: torch.optim.Adam, one param group, no amsgrad/maximize (train_teacher.py:234-236)")
    if "SAGE" in name:
        if enc.norm_type not in ("none", "batch") or not _is_relu(enc.activation):
            raise NotImplementedError("TeacherEngine: SAGE with norm_type none|batch and ReLU (the reference's configs)")
    else:
        if enc.norm_type not in ("none", "batch", "layer"):      # train.conf.yaml: cora-style GCN none, pokec / penn94 GCN batch
            raise NotImplementedError("TeacherEngine: GCN with norm_type none|batch|layer")
        for lay in enc.layers[:-1]:
            if not _is_relu(lay._activation):
                raise NotImplementedError("TeacherEngine: GraphConv(activation=F.relu) on hidden layers (models.py:170-187)")
    for bn in enc.norms:
        if isinstance(bn, nn.LayerNorm):
            if not bn.elementwise_affine or len(bn.normalized_shape) != 1:
                raise NotImplementedError("TeacherEngine: nn.LayerNorm(hidden_dim) with the reference's defaults")
        elif bn.momentum is None or not bn.affine or not bn.track_running_stats:
            raise NotImplementedError("TeacherEngine: BatchNorm1d with the reference's defaults")
    if next(model.parameters()).device.type != "cuda":
        raise RuntimeError("TeacherEngine needs the model on the GPU (HIP path only; the reference's --device -1 default "
                           "selects the CPU: pass --device 0)")


class _Arena:
    """Bump allocator over one device buffer (base None: sizing pass); 256-byte granules keep every sub-buffer float4-aligned."""

    def __init__(self, base):
        self.base, self.off = base, 0

    def take(self, nbytes):
        p = None if self.base is None else self.base + self.off
        self.off += (int(nbytes) + 255) // 256 * 256
        return p


class TeacherEngine:
    def __init__(self, model, optimizer):
        self.model, self.enc, self.opt = model, model.encoder, optimizer
        self.kind = "sage" if "SAGE" in model.model_name else "gcn"
        self.L = self.enc.num_layers
        self.bn = self.enc.norm_type == "batch"
        self.p = float(self.enc.dropout.p)
        params = list(model.parameters())
        self.params = params
        dev = params[0].device
        self.dev = dev
        f32 = dict(dtype=torch.float32, device=dev)
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        self.flat_grads = torch.zeros(sum(sizes), **f32)
        self.grads, off = [], 0
        for p, sz in zip(params, sizes):
            self.grads.append(self.flat_grads[off:off + p.numel()].view_as(p))
            off += sz
        for p, g in zip(params, self.grads):
            p.grad = g
        self._g = {id(p): g for p, g in zip(params, self.grads)}
        exp_avg, exp_avg_sq = [], []
        for p in params:
            st = optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            exp_avg.append(st["exp_avg"])
            exp_avg_sq.append(st["exp_avg_sq"])
        self.step_count = int(optimizer.state[params[0]]["step"])
        self.table = ops.TensorTable(params, self.grads, exp_avg, exp_avg_sq)
        self.loss_out = torch.zeros(1, **f32)
        self.loss_accum = torch.zeros(1, **f32)
        self._sage_desc = self._arena = self._arena_stream = None      # step_sage: persistent descriptor and scratch arena
        self.ws_loss = torch.empty(1024, **f32)
        # step_sage: the hidden layers' h = dropout(relu(norm(z))) is NOT written -- the next layer's aggregation applies that tail to the z
        # rows it gathers (glnn::spmm_csr_tail; same arithmetic per element, so the step is bit-identical to the materialised form, one pass
        # over the layer's activations and its buffer less).  GLNN_TEACHER_GATHER_TAIL=0 (read here, once) keeps the act_fwd launches.
        self.gather_tail = os.environ.get("GLNN_TEACHER_GATHER_TAIL", "1") != "0"
        self._one_call = os.environ.get("GLNN_TEACHER_ONE_CALL", "1") != "0"
        self.base_seed = int(torch.initial_seed()) & 0xFFFFFFFF
        self.grad_sync = None

    # ------------------------------------------------------------------------------------------
    def grad(self, p):
        return self._g[id(p)]

    def _seed(self, layer):
        return _mix32(self.base_seed ^ _mix32(self.step_count * 131 + layer + 1)) if self.p > 0 else 0

    def _adam(self):
        g = self.opt.param_groups[0]
        if self.grad_sync is not None:
            self.grad_sync()
        ops.adam_step(self.table, g["lr"], self.step_count, weight_decay=g["weight_decay"], beta1=g["betas"][0], beta2=g["betas"][1],
                      eps=g["eps"])

    def sync_optimizer_state(self):
        for p in self.params:
            self.opt.state[p]["step"] = torch.tensor(float(self.step_count))

    def _tail_fwd(self, l, z):
        """norms[l] -> relu -> dropout of hidden layer l (models.py:113-117), materialised for the next layer's gather."""
        seed = self._seed(l)
        if self.bn:
            bn = self.enc.norms[l]
            stats = ops.bn_stats(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, eps=bn.eps,
                                 momentum=bn.momentum)
        else:
            stats = (None, None, None, None)
        return ops.act_fwd(z, stats[2], stats[3], drop_p=self.p, drop_seed=seed), stats, seed

    def _tail_bwd(self, l, dh, z, stats, seed, bias_grad):
        """dz_l (in place on dh) + dgamma / dbeta + the bias gradient of the layer in front."""
        mean, rstd, a_scale, a_shift = stats
        if self.bn:
            bn = self.enc.norms[l]
            ops.bn_relu_bwd(dh, z, bn.weight, mean, rstd, a_scale, a_shift, dz=dh, dgamma=self.grad(bn.weight), dbeta=self.grad(bn.bias),
                            drop_p=self.p, drop_seed=seed, dz_col_sum=bias_grad)
        else:
            ops.bn_relu_bwd(dh, z, dz=dh, drop_p=self.p, drop_seed=seed, dz_col_sum=bias_grad)
        return dh

    # ------------------------------------------------------------------------------------------ GraphSAGE on blocks
    @torch.no_grad()
    def _sage_signature(self, enc):
        """Everything the persistent glnn_sage_step_desc captured once: the data pointers of every parameter, gradient buffer and BatchNorm
        buffer, and the scalars baked into it.  A step whose signature differs rebuilds the descriptor (only the weight / bias
        pointers used to be compared, so a replaced norm module or a changed dropout left the C call on stale or freed pointers)."""
        ptr = lambda t: None if t is None else t.data_ptr()
        sig = [float(enc.dropout.p), len(enc.layers)]
        for l, layer in enumerate(enc.layers):
            w, b = layer.fc_neigh.weight, layer.fc_neigh.bias
            sig += [ptr(w), ptr(b), ptr(self.grad(w)), ptr(self.grad(b))]
            if l != len(enc.layers) - 1 and self.bn:
                bn = enc.norms[l]
                sig += [ptr(bn.weight), ptr(bn.bias), ptr(self.grad(bn.weight)), ptr(self.grad(bn.bias)), ptr(bn.running_mean), ptr(bn.running_var),
                        ptr(bn.num_batches_tracked), bn.eps, bn.momentum]
        return sig + [ptr(self.ws_loss), ptr(self.loss_out), ptr(self.loss_accum)]

    def step_sage(self, blocks, feats, labels, output_nodes, lamb=1.0, input_nodes=None):
        """One optimisation step on a batch of sampled blocks (blocks[0] outermost).  `feats` is the GLOBAL feature matrix:
        when blocks[0] carries global source ids (glnn_amd.graph.NodeDataLoader) its aggregation gathers from it directly,
        otherwise feats[input_nodes] is gathered once (blocks that came from elsewhere).  The whole forward + loss + backward
        is ONE C call (glnn_sage_fwd_bwd_f32, csrc/sage_step.hip) over buffers allocated here; Adam follows."""
        enc, L = self.enc, self.L
        ops._need_cuda(feats, labels, output_nodes, input_nodes)
        if len(blocks) != L:
            raise ValueError(f"TeacherEngine.step_sage: {len(blocks)} blocks for {L} layers")
        if L > _lib.SAGE_MAX_LAYERS:
            raise NotImplementedError(f"TeacherEngine: at most {_lib.SAGE_MAX_LAYERS} layers")
        if labels.dtype != torch.int64 or output_nodes.dtype != torch.int64 or not output_nodes.is_contiguous():
            raise ValueError("TeacherEngine.step_sage: labels / output_nodes must be int64 (output_nodes contiguous)")
        x = ops.as_feat(feats)
        dev = self.dev
        if blocks[0].gindices is None:
            if input_nodes is None:
                raise ValueError("TeacherEngine.step_sage: blocks without global ids need input_nodes")
            x = ops.gather_rows(x, input_nodes)
        for l in range(1, L):
            if blocks[l].num_src_nodes() != blocks[l - 1].num_dst_nodes():
                raise ValueError(f"TeacherEngine.step_sage: block {l} has {blocks[l].num_src_nodes()} sources, block {l - 1} "
                                 f"{blocks[l - 1].num_dst_nodes()} destinations")
        if x.shape[0] < (blocks[0].num_src_nodes() if blocks[0].gindices is None else 1):
            raise ValueError("TeacherEngine.step_sage: the layer-0 source matrix is smaller than the outermost block")
        self.step_count += 1
        # The descriptor lives across steps: everything that does not depend on the batch (parameter / gradient / BatchNorm pointers, sizes,
        # loss buffers) is written once; a step only fills in the blocks, the dropout seeds and the scratch pointers.  (Building the
        # ~150 ctypes fields from nothing was 0.1 ms of the 0.25 ms of host time per step of a loop that is host-bound.)
        r4 = lambda c: (c + 3) // 4 * 4
        ptr = lambda t: None if t is None else t.data_ptr()
        d = self._sage_desc
        if d is None:
            d = self._sage_desc = _lib.SageStepDesc()
            self._sage_dims = [enc.layers[0].fc_neigh.weight.shape[1]] + [lay.fc_neigh.weight.shape[0] for lay in enc.layers]
            self.p = float(enc.dropout.p)
            d.num_layers, d.batchnorm, d.dropout_p = L, 1 if self.bn else 0, self.p
            for i, v in enumerate(self._sage_dims):
                d.dims[i] = v
            for l, layer in enumerate(enc.layers):
                y = d.layer[l]
                w, b = layer.fc_neigh.weight, layer.fc_neigh.bias
                y.w, y.b, y.gw, y.gb = ptr(w), ptr(b), ptr(self.grad(w)), ptr(self.grad(b))
                if l != L - 1 and self.bn:
                    bn = enc.norms[l]
                    d.bn_eps, d.bn_momentum = bn.eps, bn.momentum
                    y.gamma, y.beta, y.ggamma, y.gbeta = ptr(bn.weight), ptr(bn.bias), ptr(self.grad(bn.weight)), ptr(self.grad(bn.bias))
                    y.running_mean, y.running_var, y.nbt = ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked)
            d.ws_loss, d.ws_loss_floats = ptr(self.ws_loss), self.ws_loss.numel()
            d.loss_out, d.loss_accum = ptr(self.loss_out), ptr(self.loss_accum)
            self._sage_static = self._sage_signature(enc)
        elif self._sage_static != self._sage_signature(enc):
            self._sage_desc = None                          # something the descriptor captured was replaced (load_state_dict keeps the
                                                            # tensors; .to(), a new norm module, a changed dropout / eps / momentum do not)
            self.step_count -= 1
            return self.step_sage(blocks, feats, labels, output_nodes, lamb, input_nodes)
        dims = self._sage_dims
        d.lamb = float(lamb)
        # Every buffer of the step is scratch that only the C call touches, so they are carved as raw pointers out of ONE arena that
        # persists across steps (same-stream reuse is ordered; it grows when a batch needs more).  ONE pass hands out the pointers.

        def layout(A):
            max_rows, max_hidden = 1, 4
            for l, blk in enumerate(blocks):
                y = d.layer[l]
                n_dst, n_src, nnz = blk.num_dst_nodes(), blk.num_src_nodes(), blk.num_edges()
                y.agg, y.ld_agg = A.take(4 * n_dst * r4(dims[l])), r4(dims[l])
                y.z, y.ldz = A.take(4 * n_dst * r4(dims[l + 1])), r4(dims[l + 1])
                if l != L - 1:
                    if self.gather_tail and dims[l + 1] <= 256:      # (the tail-in-gather kernel holds a row in registers: d <= 256)
                        y.h, y.ldh = None, 0
                    else:
                        y.h, y.ldh = A.take(4 * n_dst * r4(dims[l + 1])), r4(dims[l + 1])
                    max_rows, max_hidden = max(max_rows, n_dst), max(max_hidden, dims[l + 1])
                    if self.bn:
                        y.mean, y.rstd, y.a_scale, y.a_shift = (A.take(4 * dims[l + 1]) for _ in range(4))
                if l >= 1 and getattr(blk, "t_indptr", None) is not None:      # transposed by the loader (NodeDataLoader.global_first_block)
                    y.t_indptr, y.t_indices, y.inv_deg = ptr(blk.t_indptr), ptr(blk.t_indices), ptr(blk.inv_deg)
                    y.tr_ws, y.tr_ws_bytes = None, 0
                elif l >= 1:
                    nnz_t = nnz + n_dst
                    wsb = int(_lib.lib().glnn_csr_transpose_workspace_bytes(n_src, nnz_t))
                    y.t_indptr, y.t_indices, y.inv_deg = A.take(8 * (n_src + 1)), A.take(4 * max(nnz_t, 1)), A.take(4 * n_dst)
                    y.tr_ws, y.tr_ws_bytes = A.take(wsb), (wsb + 7) // 8 * 8
            n_out = blocks[-1].num_dst_nodes()
            d.dlogits, d.ld_dlogits = A.take(4 * n_out * r4(dims[-1])), r4(dims[-1])
            if L > 1:
                wd = r4(max(dims[1:L]))
                d.dagg, d.ld_dagg = A.take(4 * max(b.num_dst_nodes() for b in blocks[1:]) * wd), wd
                d.dh, d.ld_dh = A.take(4 * max(b.num_src_nodes() for b in blocks[1:]) * wd), wd
            nchunks = (max_rows + 127) // 128
            d.ws_bn_floats = (3 * nchunks + 2 + 3 * ((nchunks + 63) // 64)) * max_hidden + 1024
            if L > 1 and self.bn:     # room for the outermost layer's BatchNorm backward without passes of its own (round 5)
                d.ws_bn_floats = max(d.ws_bn_floats, int(_lib.lib().glnn_sage_step_ws_bn_floats(blocks[0].num_dst_nodes(), dims[1])))
            # (x L: in the one-call step every layer's split slabs wait in ws_tn for the Adam launch to fold them -- csrc/sage_step.hip)
            d.ws_tn_floats = L * (64 * max(dims) + 256 * 128 * 128 + 2 * max(dims) * max(dims))
            # split-K slabs, sized as StudentEngine does; the GEMM only splits outputs of < 256 tiles, i.e. <= 512 slabs of 128 x 128
            d.ws_gemm_floats = min(max(16 * max(b.num_dst_nodes() for b in blocks) * min(dims[1:]), 1 << 20), 512 * 128 * 128)
            d.ws_bn, d.ws_tn, d.ws_gemm = A.take(4 * d.ws_bn_floats), A.take(4 * d.ws_tn_floats), A.take(4 * d.ws_gemm_floats)
            return A.off

        for l, blk in enumerate(blocks):       # the batch: blocks and dropout seeds
            y = d.layer[l]
            glob = l == 0 and blk.gindices is not None
            y.indptr, y.indices = ptr(blk.indptr), ptr(blk.gindices if glob else blk.indices)
            y.n_dst, y.n_src, y.nnz = blk.num_dst_nodes(), blk.num_src_nodes(), blk.num_edges()
            y.self_rows = ptr(blk.dst_nodes) if glob else None
            if l != L - 1:
                y.drop_seed = self._seed(l)
        cur = torch.cuda.current_stream(dev)
        if self._arena_stream is not None and self._arena_stream != cur:
            cur.wait_stream(self._arena_stream)            # the previous step used the arena on another stream
        self._arena_stream = cur
        arena = self._arena
        base = 0 if arena is None else (arena.data_ptr() + 255) // 256 * 256
        total = layout(_Arena(base))
        if arena is None or total + 256 > arena.numel():    # first step / a bigger batch: (re)allocate with headroom, lay out again
            arena = self._arena = torch.empty(int(total * 1.25) + 256, dtype=torch.uint8, device=dev)
            layout(_Arena((arena.data_ptr() + 255) // 256 * 256))
        d.x, d.ldx, d.x_rows = ptr(x), x.stride(0), x.shape[0]
        d.labels, d.label_rows = ptr(labels), ptr(output_nodes)
        d.ws_loss, d.ws_loss_floats = ptr(self.ws_loss), self.ws_loss.numel()
        d.loss_out, d.loss_accum = ptr(self.loss_out), ptr(self.loss_accum)
        keep = [x]                                              # alive until the call below is queued (same-stream reuse is ordered)
        ops.note_param_write()      # (running statistics are written through raw pointers; Adam follows)
        # nothing sits between the backward and Adam (no gradient exchange): the whole step is ONE C call and Adam folds the backward's
        # last partial sums itself (glnn_sage_train_step_f32, round 6; GLNN_TEACHER_ONE_CALL=0 keeps the two calls)
        one_call = self.grad_sync is None and self._one_call
        if one_call:
            g_ = self.opt.param_groups[0]
            ad = self.table.desc
            ad.lr, ad.beta1, ad.beta2, ad.eps, ad.weight_decay, ad.step = g_["lr"], g_["betas"][0], g_["betas"][1], g_["eps"], g_["weight_decay"], self.step_count
            rc = _lib.lib().glnn_sage_train_step_f32(ctypes.byref(d), ctypes.byref(ad), ops._stream())
        else:
            rc = _lib.lib().glnn_sage_fwd_bwd_f32(ctypes.byref(d), ops._stream())
        if rc != 0:
            self.step_count -= 1          # the step never happened: Adam's bias correction and the dropout seeds stay where they were
        _lib.check(rc, "glnn_sage_train_step_f32" if one_call else "glnn_sage_fwd_bwd_f32")
        if not one_call:
            self._adam()

    # ------------------------------------------------------------------------------------------ full-graph GCN
    @torch.no_grad()
    def step_gcn(self, g, feats, labels, idx_train, lamb=1.0):
        enc, L, p = self.enc, self.L, self.p
        ops._need_cuda(feats, labels, idx_train, g.indptr)
        if g.has_zero_in_degree() and not all(lay._allow_zero_in_degree for lay in enc.layers):
            raise RuntimeError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid "
                               "(dgl GraphConv semantics; add self-loops or set allow_zero_in_degree).")
        n = g.num_dst_nodes()
        self.step_count += 1
        try:
            self._step_gcn_body(g, feats, labels, idx_train, lamb, n)
        except Exception:
            self.step_count -= 1          # the step never happened (see step_sage)
            raise
        self._adam()

    def _step_gcn_body(self, g, feats, labels, idx_train, lamb, n):
        enc, L, p = self.enc, self.L, self.p
        a = ops.as_feat(feats)
        saved = []
        norm_kind = enc.norm_type
        for l, layer in enumerate(enc.layers):
            last = l == L - 1
            y, mid, first = graphconv_fwd(g, a, layer.weight, layer.bias, relu=not last)
            seed = self._seed(l)
            stats = None
            if not last:      # reference models.py:195-198: norms[l] -> dropout, NO ReLU behind the norm (it sits inside the conv)
                if norm_kind == "batch":
                    bn = enc.norms[l]
                    stats = ops.bn_stats(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, eps=bn.eps,
                                         momentum=bn.momentum)
                    a = ops.act_fwd(y, stats[2], stats[3], drop_p=p, drop_seed=seed, relu=False)
                elif norm_kind == "layer":
                    ln = enc.norms[l]
                    a, mu, rs = ops.layernorm_fwd(y, ln.weight, ln.bias, eps=ln.eps, relu=False, drop_p=p, drop_seed=seed)
                    stats = (mu, rs)
                else:
                    a = ops.act_fwd(y, drop_p=p, drop_seed=seed) if p > 0 else y       # y >= 0 already (ReLU inside the conv)
            saved.append((mid, first, y, seed, stats))
        logits = saved[-1][2]
        logits_tr = ops.gather_rows(logits, idx_train)                            # out[idx_train] (train_and_eval.py:22)
        _, dl = ops.softmax_loss(logits_tr, ops.LOSS_NLL, float(lamb), labels=labels, label_rows=idx_train, loss_out=self.loss_out,
                                 loss_accum=self.loss_accum, workspace=self.ws_loss)
        dz = ops.feat_empty(n, logits.shape[1], self.dev, zero=True)
        ops.scatter_rows(dl, idx_train, dz)
        ops.col_sum(dz, out=self.grad(enc.layers[-1].bias))
        for l in range(L - 1, -1, -1):
            layer = enc.layers[l]
            mid, first, _, _, _ = saved[l]
            da = graphconv_bwd(g, dz, mid, first, layer.weight, self.grad(layer.weight), want_da=l > 0)
            if l == 0:
                break
            _, _, y_prev, seed_prev, stats = saved[l - 1]
            if norm_kind == "batch":       # dropout + norm backward (no ReLU in this tail) -> grad wrt the conv's output y
                bn = enc.norms[l - 1]
                mean, rstd, a_sc, a_sh = stats
                ops.bn_relu_bwd(da, y_prev, bn.weight, mean, rstd, a_sc, a_sh, dz=da, dgamma=self.grad(bn.weight), dbeta=self.grad(bn.bias),
                                drop_p=p, drop_seed=seed_prev, relu=False)
                dz, _, _ = ops.bn_relu_bwd(da, y_prev, dz=da, dz_col_sum=self.grad(enc.layers[l - 1].bias))     # the ReLU inside conv l-1
            elif norm_kind == "layer":
                ln = enc.norms[l - 1]
                _, dg, db = ops.layernorm_bwd(da, y_prev, ln.weight, ln.bias, stats[0], stats[1], relu=False, drop_p=p, drop_seed=seed_prev, dz=da)
                self.grad(ln.weight).copy_(dg)
                self.grad(ln.bias).copy_(db)
                dz, _, _ = ops.bn_relu_bwd(da, y_prev, dz=da, dz_col_sum=self.grad(enc.layers[l - 1].bias))
            else:                          # dropout backward, then the ReLU inside conv l-1 (y > 0 <=> z > 0)
                dz, _, _ = ops.bn_relu_bwd(da, y_prev, dz=da, drop_p=p, drop_seed=seed_prev, dz_col_sum=self.grad(enc.layers[l - 1].bias))


def get_engine(model, optimizer):
    eng = getattr(model, "_glnn_teacher_engine", None)
    if eng is None or eng.opt is not optimizer or any(a is not b for a, b in zip(eng.params, model.parameters())):
        eng = TeacherEngine(model, optimizer)
        object.__setattr__(model, "_glnn_teacher_engine", eng)
    return eng
