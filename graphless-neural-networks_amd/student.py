"""Fused MLP student training step on libglnn_hip.so.

One `StudentEngine.step()` is the loop body of the reference's `train_mini_batch`
(reference train_and_eval.py:74-85):

    logits = model(None, feats[idx]); out = logits.log_softmax(1); loss = criterion(out, target[idx])
    total_loss += loss.item(); loss *= lamb; optimizer.zero_grad(); loss.backward(); optimizer.step()

executed as a fixed sequence of C-ABI kernel launches with NO autograd graph, NO intermediate
activation tensors beyond the pre-activation outputs z_l, and NO host synchronisation:

  forward   z_0 = gemm(feats[idx] gathered in the operand load) ; per hidden layer: bn_stats(z_l) ->
            (scale, shift) ; z_{l+1} = gemm(drop(relu(z_l*scale+shift)) formed in the operand load)
  loss      log_softmax + NLL | KL(log-target) -> loss (accumulated on device) and dlogits (x lamb/B)
  backward  dW_l = gemm_tn(dz_l, recomputed activation) (+ bias grad) ; da = gemm(dz_l, W_l) ;
            dz_{l-1} = BN/ReLU/dropout backward
  update    one fused multi-tensor Adam launch over all parameters (torch.optim.Adam semantics)

The engine works IN PLACE on the `Model`'s own parameters/buffers and on the torch optimizer's own state
tensors (exp_avg / exp_avg_sq / step), so `state_dict()`, early-stopping snapshots
(train_and_eval.py:588,596) and `optimizer.state_dict()` behave exactly as with the reference."""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib, ops


def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


class StudentEngine:
    def __init__(self, model, optimizer, max_batch, grad_sync=None, loss_scale_rows=None):
        """grad_sync: optional callable run between backward and Adam (data-parallel all-reduce over
        `self.flat_grads`, see glnn_amd.dist.make_grad_sync).  loss_scale_rows: the GLOBAL batch size when
        the batch is split over ranks (the loss mean and dlogits are taken over it)."""
        enc = model.encoder
        if "MLP" not in model.model_name or enc.norm_type not in ("none", "batch", "layer"):
            raise NotImplementedError("StudentEngine: MLP students with norm_type none|batch|layer (reference models.py:28-31)")
        if type(optimizer) is not torch.optim.Adam:
            raise NotImplementedError("StudentEngine: the reference uses torch.optim.Adam (train_student.py:275)")
        grp = optimizer.param_groups
        if len(grp) != 1 or grp[0].get("amsgrad") or grp[0].get("maximize"):
            raise NotImplementedError("StudentEngine: single param group, no amsgrad/maximize")
        self.model, self.enc, self.opt = model, enc, optimizer
        self.L = enc.num_layers
        self.bn = enc.norm_type == "batch"
        self.ln = enc.norm_type == "layer"          # per-row statistics; the tail is always materialised (glnn_layernorm_fwd_f32)
        self.p = float(enc.dropout.p)
        self.W = [l.weight for l in enc.layers]
        self.b = [l.bias for l in enc.layers]
        dev = self.W[0].device
        if dev.type != "cuda":
            raise RuntimeError("StudentEngine needs the model on the GPU (HIP path only)")
        self.dev = dev
        self.dims = [self.W[0].shape[1]] + [w.shape[0] for w in self.W]
        self.B = int(max_batch)
        f32 = dict(dtype=torch.float32, device=dev)
        B = self.B
        # pre-activation outputs z_l (l < L-1), logits, gradient buffers
        self.z = [ops.feat_empty(B, self.dims[l + 1], dev) for l in range(self.L - 1)]
        self.logits = ops.feat_empty(B, self.dims[-1], dev)
        self.dlogits = ops.feat_empty(B, self.dims[-1], dev)
        hmax = max(self.dims[1:-1]) if self.L > 1 else 4
        self.da = ops.feat_empty(B, hmax, dev)
        self.dz = ops.feat_empty(B, hmax, dev)
        self.stats = []      # per hidden layer: (mean, rstd, a_scale, a_shift)
        for l in range(self.L - 1):
            h = self.dims[l + 1]
            if self.bn:
                self.stats.append(tuple(torch.empty(h, **f32) for _ in range(4)))
            elif self.ln:            # per-ROW mean / rstd
                self.stats.append((torch.empty(B, **f32), torch.empty(B, **f32), None, None))
            else:
                self.stats.append((None, None, torch.ones(h, **f32), torch.zeros(h, **f32)))
        # parameters in torch order (model.parameters()): layers.{i}.weight, .bias ..., norms.{i}.weight, .bias ...
        params = list(model.parameters())
        self.params = params
        self.grad_sync, self.loss_scale_rows = grad_sync, loss_scale_rows
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]          # keep every view 16-byte aligned
        self.flat_grads = torch.zeros(sum(sizes), **f32)
        self.grads, off = [], 0
        for p, sz in zip(params, sizes):
            self.grads.append(self.flat_grads[off:off + p.numel()].view_as(p))
            off += sz
        for p, g in zip(params, self.grads):
            p.grad = g                               # optimizer.zero_grad() semantics: overwritten every step
        self._gW = {id(p): g for p, g in zip(params, self.grads)}
        # share the torch optimizer's state tensors
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
        self._one_call = os.environ.get("GLNN_STUDENT_ONE_CALL", "1") != "0"
        # workspaces
        hk = max(self.dims)
        n_chunks = (B + 127) // 128
        self.ws_bn = torch.empty(max((3 * n_chunks + 2) * hmax * max(self.L - 1, 1), 1024,   # x (L-1): per-layer slices when Adam folds the column sums
                                      max(4, 3 * (self.L - 1)) * ((B + 31) // 32) * hmax if B * hmax <= (1 << 20) else 0,   # mlp_lat.hip: 32-row tile partials (statistics: two layers in flight; backward: 3 per layer)
                                      _lib.lib().glnn_layernorm_bwd_workspace_floats(B, hmax) if self.ln else 0), **f32)
        # weight-gradient workspace: split-reduction slabs.  When affordable (<= 16 M floats) it holds the slabs of EVERY layer at once
        # (<= ceil(B/64) splits each): the batched weight-gradient launch needs that, and so does leaving the folds to Adam
        all_slabs = 64 * hk + ((B + 63) // 64) * sum(self.dims[l] * self.dims[l + 1] for l in range(self.L))
        if all_slabs > (1 << 24):
            # large steps: what glnn_gemm_tn_f32 really plans -- the reduction is split until ~1024 workgroups exist (64 x 64 tiles), never
            # below two 32-row k-tiles per split -- plus the classifier's row-chunk partials out of the BatchNorm backward (<= 64 classes).
            # Without room for ALL layers' slabs a later layer's product loses its split (vk_class MLP, 512 x 100 over 6754 rows: one
            # workgroup column of 211 k-tiles, 60 us instead of 10) because the earlier layers' slabs wait in ws_tn for Adam to fold them.
            def planned(l):
                t64 = -(-self.dims[l + 1] // 64) * -(-self.dims[l] // 64)
                return min(-(-1024 // t64) if t64 < 1024 else 1, (B + 63) // 64) * self.dims[l] * self.dims[l + 1]
            tight = 64 * hk + sum(planned(l) + 4 for l in range(self.L))
            if self.dims[-1] <= 64:
                tight += n_chunks * (self.dims[-1] * self.dims[-2] + self.dims[-1]) + 8
            all_slabs = tight if tight <= (1 << 26) else 0
        self.ws_tn = torch.empty(max(64 * hk + 256 * 128 * 128 + 2 * hk * hk, all_slabs), **f32)
        # (+ a float4-addressable shadow of W_0 when the feature rows are wide and unaligned: csrc/mlp_step.hip, GLNN_STUDENT_PAD_W0)
        shadow = self.dims[1] * ((self.dims[0] + 3) // 4 * 4) + 4 * B * self.dims[1] if self.dims[0] >= 512 and self.dims[0] % 4 else 0
        self.ws_gemm = torch.empty(max(16 * B * min(self.dims[1:]), 1 << 20) + shadow, **f32)
        self.ws_loss = torch.empty(256 * 65 + 1024, **f32)
        # fused finalizes (last workgroup folds the partials: 7 launches fewer per step): arxiv MLP 0.143 -> 0.136 ms, MLP3w4 0.193 ->
        # 0.185, products MLP 0.229 -> 0.226, MLP3w8 1.196 -> 1.203 (interleaved A/B) -> only for the small, latency-bound steps
        fused = os.environ.get("GLNN_STUDENT_FUSED_FINALIZE", "auto")
        self.sync_counters = torch.zeros(_lib.MLP_COUNTERS, dtype=torch.int32, device=dev) \
            if fused == "1" or (fused == "auto" and B * hmax <= (1 << 20)) else None
        # the tail of hidden layer l (norm -> ReLU -> dropout) is either recomputed inside the operand loads of the next GEMM and
        # of the weight gradient (nothing stored) or written once per step and read plain.  With dropout the recompute hashes
        # every staged element in every workgroup that stages it (n_next / 128 times): MLP3w8 forward GEMM 316 vs 260 us + a
        # 13 us pass, weight gradient 309 vs 277 us -> materialise when the next layer is wide and the batch large; small
        # latency-bound steps keep the recompute (one launch fewer per hidden layer).
        mat = os.environ.get("GLNN_STUDENT_MATERIALIZE_ACT", "auto")
        self.act = [ops.feat_empty(B, self.dims[l + 1], dev)
                    if self.ln or mat == "1" or (mat == "auto" and self._materialize_tail(l, B)) else None
                    for l in range(self.L - 1)]
        # feats[idx] copied once per step when the batch is long enough for the first layer's weight gradient to take the
        # pipelined kernel (>= 2048 reduction rows, > 64 feature columns); small batches (<= 1024 rows, <= 256 features: the latency
        # GEMM of csrc/mlp_lat.hip) get the buffer filled by the first layer's GEMM itself; the rest keep the gather in the operand loads
        pg = os.environ.get("GLNN_STUDENT_PREGATHER", "auto")
        self.xb = ops.feat_empty(B, self.dims[0], dev) if pg == "1" or (pg == "auto" and ((B >= 2048 and self.dims[0] > 64) or (B <= 1024 and self.dims[0] <= 256))) else None
        self.dz2 = None
        if self.dz2 is None and self.sync_counters is not None and self.L <= 3:
            self.dz2 = ops.feat_empty(B, hmax, dev)      # small steps defer their weight gradients to ONE batched launch (mlp_step.hip): dz_l must outlive the loop
        self.loss_out = torch.zeros(1, **f32)
        self.loss_accum = torch.zeros(1, **f32)
        self.base_seed = int(torch.initial_seed()) & 0xFFFFFFFF
        self._seed_arr = (ctypes.c_uint32 * _lib.MLP_MAX_LAYERS)()
        self.exchange = None
        self.desc = self._build_desc()

    def _materialize_tail(self, l, B):
        """Store dropout(relu(norm(z_l))) once per step instead of re-evaluating it in the operand loads of its consumers?  The
        dropout hash is the cost, and every column tile of the NEXT layer's GEMM re-evaluates it: worth a [B, h] write and
        re-read when the next layer is >= 512 wide (4+ column tiles).  Round 2 applied it from B * h >= 2^21 (MLP3w8's first
        hidden layer: 1.18 -> 1.00 ms); round 3 from 2^19, which adds MLP3w4 at B = 512 (0.164 -> 0.158 ms, interleaved A/B).
        A tail that feeds only the narrow output layer (one column tile) stays recomputed: materialising it as well costs
        MLP3w8 0.8 %."""
        if self.p <= 0:
            # no hash to save, but a stored tail is a PLAIN operand: its consumers take the pipelined kernels (148 TF) instead of the
            # operand-transform ones (127 TF forward, 117 TF weight gradient) -- MLP3w8 at B = 4096 without dropout: 1.02 -> 0.93 ms
            return self.dims[l + 2] >= 512 and B * self.dims[l + 1] >= (1 << 21)
        return self.dims[l + 2] >= 512 and B * self.dims[l + 1] >= (1 << 19)

    def enable_batch_split(self, world, rank, group=None):
        """Data-parallel over ONE batch: every rank steps on its slice of the batch rows.  BatchNorm statistics are
        taken over the whole batch through the exchange hook (two small all-gathers per BN layer and step), the loss
        is normalised by the global row count (`loss_scale_rows` is set per step by the caller or defaults to
        world * local rows), gradients are SUMMED over ranks -> the update equals the single-GPU step on the whole
        batch (tests/test_dist_gpu.py).  Dropout masks are drawn per rank."""
        from .dist import OverlappedGradSync, StatExchange
        if world <= 1:
            return
        hmax = max(self.dims[1:-1]) if self.L > 1 else 4
        self.exchange = StatExchange(world, rank, hmax, self.dev, group)
        d = self.desc
        d.world, d.rank = world, rank
        d.exchange = self.exchange.callback
        d.sync_send, d.sync_recv, d.sync_rows = self.exchange.send.data_ptr(), self.exchange.recv.data_ptr(), self.exchange.rows.data_ptr()
        self.overlap = OverlappedGradSync(self, world, group, average=False)   # sets self.grad_sync and the desc hook
        self.batch_split_world = world
        self.base_seed = _mix32(self.base_seed ^ (0x9E3779B9 * (rank + 1)))      # independent dropout masks per rank

    def _build_desc(self):
        d = _lib.MlpStepDesc()
        L = self.L
        if L > _lib.MLP_MAX_LAYERS:
            raise NotImplementedError(f"StudentEngine: at most {_lib.MLP_MAX_LAYERS} layers")
        d.num_layers, d.batchnorm, d.dropout_p, d.max_batch = L, 1 if self.bn else (2 if self.ln else 0), self.p, self.B
        for i, v in enumerate(self.dims):
            d.dims[i] = v
        ptr = lambda t: None if t is None else t.data_ptr()
        for l in range(L):
            if not (self.W[l].is_contiguous() and self.b[l].is_contiguous()):
                raise ValueError("StudentEngine: contiguous parameters required")
            d.w[l], d.b[l] = ptr(self.W[l]), ptr(self.b[l])
            d.gw[l], d.gb[l] = ptr(self._grad(self.W[l])), ptr(self._grad(self.b[l]))
        d.bn_eps, d.bn_momentum = 1e-5, 0.1
        for l in range(L - 1):
            mean, rstd, sc, sh = self.stats[l]
            d.mean[l], d.rstd[l], d.a_scale[l], d.a_shift[l] = ptr(mean), ptr(rstd), ptr(sc), ptr(sh)
            d.z[l], d.ldz[l] = ptr(self.z[l]), self.z[l].stride(0)
            if self.act[l] is not None:
                d.act[l], d.ld_act[l] = ptr(self.act[l]), self.act[l].stride(0)
            if self.bn:
                bn = self.enc.norms[l]
                if bn.momentum is None or not bn.affine or not bn.track_running_stats:
                    raise NotImplementedError("StudentEngine: BatchNorm1d with the reference's defaults")
                d.bn_eps, d.bn_momentum = bn.eps, bn.momentum
                d.gamma[l], d.beta[l] = ptr(bn.weight), ptr(bn.bias)
                d.ggamma[l], d.gbeta[l] = ptr(self._grad(bn.weight)), ptr(self._grad(bn.bias))
                d.running_mean[l], d.running_var[l], d.nbt[l] = ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked)
            elif self.ln:
                ln = self.enc.norms[l]
                if not isinstance(ln, nn.LayerNorm) or not ln.elementwise_affine or len(ln.normalized_shape) != 1:
                    raise NotImplementedError("StudentEngine: nn.LayerNorm(hidden_dim) with the reference's defaults")
                d.bn_eps = ln.eps
                d.gamma[l], d.beta[l] = ptr(ln.weight), ptr(ln.bias)
                d.ggamma[l], d.gbeta[l] = ptr(self._grad(ln.weight)), ptr(self._grad(ln.bias))
        d.logits, d.ld_logits = ptr(self.logits), self.logits.stride(0)
        d.dlogits, d.ld_dlogits = ptr(self.dlogits), self.dlogits.stride(0)
        d.da, d.ld_da, d.dz, d.ld_dz = ptr(self.da), self.da.stride(0), ptr(self.dz), self.dz.stride(0)
        d.ws_bn, d.ws_bn_floats = ptr(self.ws_bn), self.ws_bn.numel()
        d.ws_tn, d.ws_tn_floats = ptr(self.ws_tn), self.ws_tn.numel()
        d.ws_gemm, d.ws_gemm_floats = ptr(self.ws_gemm), self.ws_gemm.numel()
        d.ws_loss, d.ws_loss_floats = ptr(self.ws_loss), self.ws_loss.numel()
        d.loss_out, d.loss_accum = ptr(self.loss_out), ptr(self.loss_accum)
        if self.xb is not None:
            d.xb, d.ld_xb = ptr(self.xb), self.xb.stride(0)
        if self.sync_counters is not None and max(self.dims) <= 64 * (_lib.MLP_COUNTERS - 1):
            d.sync_counters = ptr(self.sync_counters)
        if self.dz2 is not None:
            d.dz2, d.ld_dz2 = ptr(self.dz2), self.dz2.stride(0)
        return d

    # ------------------------------------------------------------------------------------------
    def _grad(self, p):
        return self._gW[id(p)]

    def _seed(self, layer):
        return _mix32(self.base_seed ^ _mix32(self.step_count * 131 + layer + 1))

    def step(self, feats, idx, kind, target, lamb, target_rows=None):
        """One optimisation step on rows `idx` (int64 device vector, or None = rows 0..m-1 of feats)."""
        enc, L, p = self.enc, self.L, self.p
        m = idx.numel() if idx is not None else feats.shape[0]
        if m > self.B:
            raise ValueError(f"batch of {m} rows exceeds the engine's buffers ({self.B})")
        ops._need_cuda(feats, idx, target, target_rows)
        if feats.dtype != torch.float32 or feats.dim() != 2 or feats.stride(1) != 1 or feats.shape[1] != self.dims[0]:
            raise ValueError("StudentEngine.step: feats must be a row-major float32 [N, feat_dim] tensor")
        if idx is not None and (idx.dtype != torch.int64 or not idx.is_contiguous()):
            raise ValueError("StudentEngine.step: idx must be a contiguous int64 vector")
        if kind == ops.LOSS_NLL and target.dtype != torch.int64:
            raise ValueError("StudentEngine.step: NLL targets are int64 labels")
        if kind == ops.LOSS_KL and (target.dtype != torch.float32 or target.dim() != 2 or target.stride(1) != 1
                                    or target.shape[1] != self.dims[-1]):
            raise ValueError("StudentEngine.step: KL targets are float32 [N, C] teacher log-probabilities")
        lr = self.opt.param_groups[0]["lr"]
        wd = self.opt.param_groups[0]["weight_decay"]
        beta1, beta2 = self.opt.param_groups[0]["betas"]
        eps = self.opt.param_groups[0]["eps"]
        self.step_count += 1
        seeds = [self._seed(l) for l in range(L - 1)] if p > 0 else [0] * (L - 1)

        # ---- forward + loss + backward: ONE C call issuing the whole kernel sequence (csrc/mlp_step.hip) ----
        scale_rows = self.loss_scale_rows
        if scale_rows is None and self.exchange is not None:
            scale_rows = m * self.batch_split_world              # equal slices unless the caller says otherwise
        if scale_rows is not None:
            lamb = lamb * m / float(scale_rows)                 # dlogits scale becomes lamb / global_rows
        for i, sd in enumerate(seeds):
            self._seed_arr[i] = sd
        trow = idx if target_rows is None else target_rows
        # nothing sits between the backward and Adam (no gradient exchange): the whole step is ONE C call, and Adam folds the
        # backward's gradient partials itself (glnn_mlp_train_step_f32)
        ops.note_param_write()      # (the step writes parameters and BatchNorm buffers through raw pointers)
        one_call = self.grad_sync is None and self.exchange is None and getattr(self, "overlap", None) is None and self._one_call
        if one_call:
            ad = self.table.desc
            ad.lr, ad.beta1, ad.beta2, ad.eps, ad.weight_decay, ad.step = lr, beta1, beta2, eps, wd, self.step_count
            rc = _lib.lib().glnn_mlp_train_step_f32(
                ctypes.byref(self.desc), ops._p(feats), feats.stride(0), ops._p(idx), m, kind,
                ops._p(target) if kind == ops.LOSS_NLL else None,
                ops._p(target) if kind == ops.LOSS_KL else None, target.stride(0) if kind == ops.LOSS_KL else 0,
                ops._p(trow), float(lamb), self._seed_arr, ctypes.byref(ad), ops._stream())
        else:
            rc = _lib.lib().glnn_mlp_fwd_bwd_f32(
                ctypes.byref(self.desc), ops._p(feats), feats.stride(0), ops._p(idx), m, kind,
                ops._p(target) if kind == ops.LOSS_NLL else None,
                ops._p(target) if kind == ops.LOSS_KL else None, target.stride(0) if kind == ops.LOSS_KL else 0,
                ops._p(trow), float(lamb), self._seed_arr, ops._stream())
        if rc != 0:
            # a step that aborted midway: Adam's bias correction and the dropout seeds must not advance, and the arrival / fold
            # counters of the one-launch reductions may be left non-zero (the next launch would skip its wait and read stale partials)
            self.step_count -= 1
            if self.sync_counters is not None:
                self.sync_counters.zero_()
        if rc != 0 and self.exchange is not None and self.exchange.error is not None:
            err, self.exchange.error = self.exchange.error, None
            raise RuntimeError("glnn_mlp_fwd_bwd_f32: the batch-statistics exchange failed") from err
        overlap = getattr(self, "overlap", None)
        if rc != 0 and overlap is not None and overlap.error is not None:
            err, overlap.error = overlap.error, None
            raise RuntimeError("glnn_mlp_fwd_bwd_f32: a gradient all-reduce started inside the backward failed") from err
        _lib.check(rc, "glnn_mlp_train_step_f32" if one_call else "glnn_mlp_fwd_bwd_f32")
        if one_call:
            return

        # ---- (data-parallel) gradient exchange, then Adam ---------------------------------------
        if self.grad_sync is not None:
            self.grad_sync()
        ops.adam_step(self.table, lr, self.step_count, weight_decay=wd, beta1=beta1, beta2=beta2, eps=eps)

    def sync_optimizer_state(self):
        """Write the step counter back into the torch optimizer's state (host-side scalars)."""
        for p in self.params:
            self.opt.state[p]["step"] = torch.tensor(float(self.step_count))


def get_engine(model, optimizer, batch):
    eng = getattr(model, "_glnn_engine", None)
    if eng is None or eng.opt is not optimizer or eng.B < batch or any(a is not b for a, b in zip(eng.params, model.parameters())):
        eng = StudentEngine(model, optimizer, batch)
        object.__setattr__(model, "_glnn_engine", eng)
    return eng


def criterion_kind(criterion):
    """Map the reference's criterion objects (train_student.py:278-279) onto the fused loss kernel."""
    if isinstance(criterion, nn.NLLLoss) and criterion.reduction == "mean" and criterion.weight is None \
            and criterion.ignore_index == -100:
        return ops.LOSS_NLL
    if isinstance(criterion, nn.KLDivLoss) and criterion.reduction == "batchmean" and criterion.log_target:
        return ops.LOSS_KL
    return None


def student_supported(model, criterion, optimizer, feats=None, labels=None):
    """The ONE eligibility check of the fused student path (train_mini_batch and StudentEngine agree by construction):
    returns the loss kind, raises NotImplementedError / RuntimeError with the reason otherwise."""
    kind = criterion_kind(criterion)
    if kind is None:
        raise NotImplementedError("student step: the criterion must be nn.NLLLoss() or nn.KLDivLoss(reduction='batchmean', "
                                  "log_target=True) (reference train_student.py:278-279)")
    enc = model.encoder
    if "MLP" not in model.model_name or enc.norm_type not in ("none", "batch", "layer"):
        raise NotImplementedError("student step: MLP students with norm_type none|batch|layer (reference models.py:28-31)")
    if enc.num_layers > _lib.MLP_MAX_LAYERS:
        raise NotImplementedError(f"student step: at most {_lib.MLP_MAX_LAYERS} layers")
    grp = optimizer.param_groups
    if type(optimizer) is not torch.optim.Adam or len(grp) != 1 or grp[0].get("amsgrad") or grp[0].get("maximize"):
        raise NotImplementedError("student step: torch.optim.Adam with one param group, no amsgrad/maximize (train_student.py:275-277)")
    for bn in enc.norms:
        if isinstance(bn, nn.LayerNorm):
            if not bn.elementwise_affine or len(bn.normalized_shape) != 1:
                raise NotImplementedError("student step: nn.LayerNorm(hidden_dim) with the reference's defaults")
        elif bn.momentum is None or not bn.affine or not bn.track_running_stats:
            raise NotImplementedError("student step: BatchNorm1d with the reference's defaults")
    if next(model.parameters()).device.type != "cuda":
        raise RuntimeError("student step: the model must be on the GPU (HIP path only; pass --device 0, the reference's default "
                           "--device -1 selects the CPU)")
    for t, name in ((feats, "feats"), (labels, "labels")):
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"student step: {name} must be on the GPU (HIP path only)")
    return kind
