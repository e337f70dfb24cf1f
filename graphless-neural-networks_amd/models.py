"""The reference's model zoo surface (reference models.py) on the HIP hot path.

Kept byte-for-byte: class names, constructor arguments, `forward` return conventions ((h_list, h)),
`Model(conf)` substring dispatch ("MLP" tested first, models.py:355,409), `Model.forward /
forward_fitnet / inference`, and state_dict key names (encoder.layers.{i}.weight|bias |
.fc_neigh.weight|bias, encoder.norms.{i}.*).  What changed is where the arithmetic runs: every
Linear / SAGEConv / GraphConv / BatchNorm(eval) / ReLU goes through libglnn_hip.so.

GAT / APPNP (ablation-only teachers) are out of scope and raise."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .autograd import linear_fn, norm_act_drop
from .nn import GraphConv, SAGEConv


def _bn_eval_fold(bn, bias):
    """Per-column (scale, shift) of  BN_eval(x + bias):  y = x*s + ((bias - rm)*s + beta), s = gamma/sqrt(rv+eps)."""
    s = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    b = bias.detach() if bias is not None else 0.0
    return s.contiguous(), ((b - bn.running_mean) * s + bn.bias.detach()).contiguous()


def _need_hip(x, what):
    if not x.is_cuda:
        raise _lib.GlnnError(f"{what}: tensors must live on the GPU -- this package computes on libglnn_hip.so only, there is no "
                             "CPU path (the reference's default --device -1 selects the CPU: pass --device 0)")


def _check_tail(module):
    """The hidden-layer tails implemented on HIP: BatchNorm1d / LayerNorm with the reference's defaults or no norm, ReLU, dropout."""
    if module.norm_type not in ("none", "batch", "layer"):
        raise NotImplementedError(f"norm_type {module.norm_type!r}: the reference builds 'none', 'batch' or 'layer' (models.py:28-31)")
    act = getattr(module, "activation", F.relu)
    if act is not F.relu and getattr(act, "__name__", "") != "relu":
        raise NotImplementedError("only ReLU hidden activations (what the reference constructs, models.py:371,381)")
    for bn in module.norms:
        if isinstance(bn, nn.LayerNorm):
            if not bn.elementwise_affine or len(bn.normalized_shape) != 1:
                raise NotImplementedError("nn.LayerNorm(hidden_dim) with the reference's defaults (elementwise affine)")
        elif not (bn.affine and bn.track_running_stats and bn.momentum is not None):
            raise NotImplementedError("BatchNorm1d with the reference's defaults (affine, running stats, momentum 0.1)")


def _norm_of(module, l):
    return module.norms[l] if module.norm_type != "none" else None


def _eval_tail(module, l, z, relu=True):
    """Eval-mode `norms[l] -> relu -> dropout` of a hidden layer on a materialised z (relu=False: GCN's `norms[l] -> dropout`):
    one glnn_act_fwd_f32 / glnn_norm_drop_fwd_f32 / glnn_layernorm_fwd_f32."""
    if module.norm_type == "batch":
        a_scale, a_shift = _bn_eval_fold(module.norms[l], None)
        return ops.act_fwd(z, a_scale, a_shift, relu=relu)
    if module.norm_type == "layer":
        ln = module.norms[l]
        return ops.layernorm_fwd(z, ln.weight.detach(), ln.bias.detach(), eps=ln.eps, relu=relu, want_stats=False)[0]
    return ops.act_fwd(z) if relu else z


class MLP(nn.Module):
    """reference models.py:7-53"""

    def __init__(self, num_layers, input_dim, hidden_dim, output_dim, dropout_ratio, norm_type="none"):
        super().__init__()
        self.num_layers = num_layers
        self.norm_type = norm_type
        self.dropout = nn.Dropout(dropout_ratio)
        self.layers = nn.ModuleList()
        self.norms = nn.ModuleList()
        if num_layers == 1:
            self.layers.append(nn.Linear(input_dim, output_dim))
        else:
            self.layers.append(nn.Linear(input_dim, hidden_dim))
            self._add_norm(hidden_dim)
            for _ in range(num_layers - 2):
                self.layers.append(nn.Linear(hidden_dim, hidden_dim))
                self._add_norm(hidden_dim)
            self.layers.append(nn.Linear(hidden_dim, output_dim))

    def _add_norm(self, hidden_dim):
        if self.norm_type == "batch":
            self.norms.append(nn.BatchNorm1d(hidden_dim))
        elif self.norm_type == "layer":
            self.norms.append(nn.LayerNorm(hidden_dim))

    def forward(self, feats):
        _need_hip(feats, "MLP.forward")
        _check_tail(self)
        if not self.training:
            with torch.no_grad():
                return self._forward_hip_eval(feats)
        h = feats
        h_list = []
        for l, layer in enumerate(self.layers):
            h = linear_fn(h, layer.weight, layer.bias)
            if l != self.num_layers - 1:
                h_list.append(h)
                # norm -> relu -> dropout as one differentiable HIP op (same module state: running stats, affine)
                h = norm_act_drop(h, _norm_of(self, l), self.dropout.p)
        return h_list, h

    def _forward_hip_eval(self, feats, want_hidden=True):
        """Eval-mode chain: each Linear is one glnn_gemm_f32 (dropout is the identity in eval mode).
        want_hidden (what `MLP.forward` returns: h_list = the raw Linear outputs, reference models.py:44-53): BN(eval)+ReLU of
        layer l are folded into the OPERAND LOAD of layer l+1, so z_l is stored exactly as the reference returns it.
        Otherwise (Model.forward / inference / evaluate: only the logits are used) they go into the EPILOGUE of layer l itself,
        relu((x W^T) s + (b s + t)): the next GEMM then reads a plain operand and takes the pipelined kernel -- the 2048-wide
        middle layer of MLP3w8 over millions of rows is the whole cost of evaluating / serving the student."""
        h = ops.as_feat(feats)
        h_list = []
        a_scale = a_shift = None
        if self.norm_type == "layer":          # per-ROW statistics cannot ride in a GEMM's per-column operand transform / epilogue
            for l, layer in enumerate(self.layers):
                z = ops.gemm(h, layer.weight, ep_shift=layer.bias)
                if l != self.num_layers - 1:
                    if want_hidden:
                        h_list.append(z)
                    z = _eval_tail(self, l, z)
                h = z
            return h_list, h
        for l, layer in enumerate(self.layers):
            last = l == self.num_layers - 1
            if want_hidden or last:
                z = ops.gemm(h, layer.weight, a_scale=a_scale, a_shift=a_shift, ep_shift=layer.bias)
            else:
                if self.norm_type == "batch":
                    s, t = _bn_eval_fold(self.norms[l], layer.bias)      # BN_eval(x + bias) = x s + t
                    z = ops.gemm(h, layer.weight, ep_scale=s, ep_shift=t, relu=True)
                else:
                    z = ops.gemm(h, layer.weight, ep_shift=layer.bias, relu=True)
            if not last and want_hidden:
                h_list.append(z)
                if self.norm_type == "batch":
                    a_scale, a_shift = _bn_eval_fold(self.norms[l], None)
                else:
                    a_scale = torch.ones(z.shape[1], device=z.device)
                    a_shift = torch.zeros(z.shape[1], device=z.device)
            h = z
        return h_list, h


class SAGE(nn.Module):
    """reference models.py:62-148"""

    def __init__(self, num_layers, input_dim, hidden_dim, output_dim, dropout_ratio, activation, norm_type="none"):
        super().__init__()
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.norm_type = norm_type
        self.activation = activation
        self.dropout = nn.Dropout(dropout_ratio)
        self.layers = nn.ModuleList()
        self.norms = nn.ModuleList()
        if num_layers == 1:
            self.layers.append(SAGEConv(input_dim, output_dim, "gcn"))
        else:
            self.layers.append(SAGEConv(input_dim, hidden_dim, "gcn"))
            self._add_norm(hidden_dim)
            for _ in range(num_layers - 2):
                self.layers.append(SAGEConv(hidden_dim, hidden_dim, "gcn"))
                self._add_norm(hidden_dim)
            self.layers.append(SAGEConv(hidden_dim, output_dim, "gcn"))

    def _add_norm(self, hidden_dim):
        if self.norm_type == "batch":
            self.norms.append(nn.BatchNorm1d(hidden_dim))
        elif self.norm_type == "layer":
            self.norms.append(nn.LayerNorm(hidden_dim))

    def forward(self, blocks, feats):
        """Sampled-block forward (reference models.py:101-119): training mode through the differentiable HIP ops of
        glnn_amd.autograd, eval mode on the plain kernels (BatchNorm running stats folded into one activation pass)."""
        _need_hip(feats, "SAGE.forward")
        _check_tail(self)
        h = feats
        h_list = []
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            h_dst = h[: block.num_dst_nodes()]
            if self.training:
                h = layer(block, (h, h_dst))
                if l != self.num_layers - 1:
                    h_list.append(h)
                    h = norm_act_drop(h, _norm_of(self, l), self.dropout.p)
            else:
                with torch.no_grad():
                    h = layer(block, (h, h_dst))
                    if l != self.num_layers - 1:
                        h_list.append(h)
                        h = _eval_tail(self, l, h)
        return h_list, h

    CHAIN_NEXT_PROJECTION = True      # A/B switch of the chained projection in `inference`

    def _tail(self, l):
        """Fused eval tail of layer l: (ep_scale, ep_shift, relu) = BN(eval) o (+bias) o ReLU; dropout is a no-op."""
        bias = self.layers[l].fc_neigh.bias
        if l == self.num_layers - 1:
            return None, bias, False
        if self.activation is not F.relu and getattr(self.activation, "__name__", "") != "relu":
            raise NotImplementedError("SAGE.inference: the reference always passes activation=F.relu (models.py:371)")
        if self.norm_type == "batch":
            # the fold is five small launches per layer: remembered while the BatchNorm's tensors and the bias are unmodified (torch's
            # version counters + ops.PARAM_EPOCH, which this library's raw-pointer writers bump)
            bn = self.norms[l]
            ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var) + ((bias,) if bias is not None else ())
            key = (ops.PARAM_EPOCH, bn.eps) + tuple((t.data_ptr(), t._version) for t in ts)
            cache = self.__dict__.setdefault("_tail_cache", {})
            ent = cache.get(l)
            if ent is None or ent[0] != key:
                cache[l] = ent = (key,) + _bn_eval_fold(bn, bias)
            return ent[1], ent[2], True
        if self.norm_type == "none":
            return None, bias, True
        raise NotImplementedError("SAGE._tail: LayerNorm has per-row statistics and cannot be folded into a kernel epilogue "
                                  "(SAGE.inference applies it as its own pass; the sharded teachers do not support it)")

    # ---- placement of the matrices the whole-graph launches gather from (ops.placed_for_gather: which allocation holds a matrix
    #      decides whether the gather over it takes 18.1 or 19.4 ms).  Intermediate activations live in buffers that are placed once per
    #      (graph, layer) and reused by every later call -- they are internal: what inference RETURNS is always a fresh tensor; the input
    #      features are copied once into a better allocation if one is found (remembered while the same unmodified tensor comes back).
    def _placed_buffer(self, g, key, rows, d, device, probe=None):
        if not ops.placement_applies(rows, d):
            return None
        cache = self.__dict__.setdefault("_placed", {})
        k = (id(g), key, rows, d, str(device))
        ent = cache.get(k)
        if ent is None or ent[0]() is not g:
            import weakref
            buf = ops.placed_for_gather(rows, d, device, g.indptr, g.indices, g.num_dst_nodes(), what=f"SAGE.inference {key[0]}{key[1]}", zero=True,
                                        probe=probe)
            cache[k] = ent = (weakref.ref(g), buf)
        return ent[1]

    def release_placed(self):
        """Drop the placed buffers this encoder keeps across inference calls (the hidden layers' rows per (graph, layer) and the placed copy
        of the input features): a long-lived process that is done with a graph gets its memory back."""
        self.__dict__.pop("_placed", None)
        self.__dict__.pop("_placed_x", None)
        torch.cuda.empty_cache()

    def _whole_graph_layer(self, l, g, x, projected, place=True):
        """Layer l of the whole-graph sweep: (y, projected for layer l+1 or None).  place=False: plain allocations (the launch is being
        used as the PROBE that places the buffer it gathers from -- see _placed_buffer)."""
        layer = self.layers[l]
        post_ln = self.norm_type == "layer" and l != self.num_layers - 1
        ep_scale, ep_shift, relu = (None, layer.fc_neigh.bias, False) if post_ln else self._tail(l)
        n = g.num_dst_nodes()
        nxt = self.layers[l + 1] if l + 1 < self.num_layers else None
        if projected is not None:
            # the dense half of this layer already came out of the previous layer's kernel: aggregate + epilogue only
            return ops.spmm(g.indptr, g.indices, projected, n, ops.AGG_SAGE_GCN, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu), None
        if nxt is not None and layer.fused_eligible() and nxt._in_feats > nxt._out_feats and nxt._out_feats <= 256 \
                and SAGE.CHAIN_NEXT_PROJECTION and not post_ln:
            # layer l aggregates first in the fused kernel and layer l+1 projects first: chain W_{l+1} behind the
            # epilogue, so the hidden activations of layer l never reach HBM (products: 2.5 GB written + read)
            out_next = self._placed_buffer(g, ("proj", l), n, nxt._out_feats, x.device) if place else None
            _, proj = ops.sage_fused(g.indptr, g.indices, x, n, layer.fc_neigh.weight, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu,
                                     x_self=x[:n], w_next=nxt.fc_neigh.weight, want_out=False, tile_order=g.fused_tile_order(),
                                     out_next=out_next)
            return x, proj                                               # (y is not read: the next layer consumes `proj`)
        out = None
        if place and nxt is not None and not post_ln:
            # a hidden layer's rows are what the next layer gathers from: they go to a placed buffer kept across calls, chosen by timing
            # the NEXT layer's own launch on each candidate allocation
            out = self._placed_buffer(g, ("y", l), n, layer._out_feats, x.device,
                                      probe=lambda cand: self._whole_graph_layer(l + 1, g, cand, None, place=False))
        y = layer(g, (x, x[:n]), ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out)
        if post_ln:
            y = _eval_tail(self, l, y)
        return y, None

    def _placed_input(self, g, feats, x):
        if not ops.placement_applies(x.shape[0], x.shape[1]) or x.shape[0] < g.num_dst_nodes():
            return x
        import weakref
        ent = self.__dict__.get("_placed_x")
        sig = (id(g), feats.data_ptr(), tuple(feats.shape), feats._version)
        if ent is not None and ent[0]() is feats and ent[1] == sig:
            return ent[2]
        px = ops.place_for_gather(x, g.indptr, g.indices, g.num_dst_nodes(), what="SAGE.inference features")
        self.__dict__["_placed_x"] = (weakref.ref(feats), sig, px)
        return px

    def inference(self, dataloader, feats, whole_graph=True):
        """Layer-wise full-neighbour inference (reference models.py:121-148).

        `dataloader` is a glnn_amd.graph.FullNeighborLoader.  whole_graph=True aggregates every destination row
        of a layer in ONE launch over the resident CSR (each dst row is independent, so the result is identical
        to the chunked sweep); whole_graph=False walks the chunks exactly like the reference does
        (gather input rows -> block conv -> fused BN/ReLU -> scatter)."""
        _need_hip(feats, "SAGE.inference")
        whole_graph = whole_graph and getattr(dataloader, "graph", None) is not None     # loaders that do not sweep arange(N)
        with torch.no_grad():
            x = ops.as_feat(feats)
            if whole_graph:
                x = self._placed_input(dataloader.graph, feats, x)
            projected = None          # x @ W_l^T handed over by the previous (fused) layer when layer l projects first
            ln = self.norm_type == "layer"
            for l, layer in enumerate(self.layers):
                post_ln = ln and l != self.num_layers - 1      # LayerNorm -> ReLU as a pass of its own behind the conv (+ bias)
                ep_scale, ep_shift, relu = (None, layer.fc_neigh.bias, False) if post_ln else self._tail(l)
                if whole_graph:
                    y, projected = self._whole_graph_layer(l, dataloader.graph, x, projected)
                else:
                    d_out = self.hidden_dim if l != self.num_layers - 1 else self.output_dim
                    y = ops.feat_empty(x.shape[0], d_out, x.device, zero=True)           # models.py:129-132
                    wp = ops.pack_weight(layer.fc_neigh.weight) if layer.fused_eligible() else None     # once per layer, not per chunk
                    # Loaders that can hand out GLOBAL-id blocks (FullNeighborLoader.global_blocks, round 6) are asked to for this sweep: the
                    # chunk loop of models.py:133-145 stays -- one block, one conv, one slice of y per chunk -- but feats[input_nodes] and
                    # y[output_nodes] = h are not copies any more: the conv gathers its source rows from x itself and writes its rows of y
                    # (products: 711 -> ~60 ms per forward; a layer that projects first projects x once, not once per chunk)
                    engine = hasattr(dataloader, "global_blocks") and not post_ln
                    xp = ops.gemm(x, layer.fc_neigh.weight) if engine and layer._in_feats > layer._out_feats else None
                    if engine:
                        dataloader.global_blocks = True
                    # The chunks of a layer are independent (disjoint rows of y, x read-only) and SHORT -- 4096 rows are 128 tiles on 256 CUs,
                    # ~90 us of a mostly idle part per launch -- so the engine-mode sweep issues them round-robin on SWEEP_STREAMS streams
                    # (products: 182 -> ~60 ms per forward; the loop, the blocks and the results are what they were).
                    cur = torch.cuda.current_stream(x.device)
                    pool = _sweep_streams(x.device) if engine else []
                    for st in pool:
                        st.wait_stream(cur)
                    # ... and launched through ONE prepared call per layer (ops.RowRangeLaunch: the per-chunk ops call's checks and argument
                    # list done once; a chunk offsets three pointers) where the layer is a single launch
                    launch = None
                    if engine and getattr(dataloader, "graph", None) is not None and (xp is not None or layer.fused_eligible()):
                        tailed = ep_scale is not None or ep_shift is not None or relu
                        g_ = dataloader.graph
                        launch = (ops.RowRangeLaunch(g_.indptr, g_.indices, xp, y, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu) if xp is not None else
                                  ops.RowRangeLaunch(g_.indptr, g_.indices, x, y, w=layer.fc_neigh.weight, ep_scale=ep_scale,
                                                     ep_shift=ep_shift if tailed else layer.fc_neigh.bias, relu=relu, w_packed=wp))
                    try:
                        it = iter(dataloader)
                        k = 0
                        for input_nodes, output_nodes, blocks in it:
                            if input_nodes is not None:
                                break
                            block = blocks[0]
                            s_, e_ = block.dst_range
                            with torch.cuda.stream(pool[k % len(pool)]) if pool else contextlib.nullcontext():
                                if launch is not None:
                                    launch(s_, e_)
                                elif xp is not None:
                                    ops.spmm(block.indptr, block.indices, xp, e_ - s_, ops.AGG_SAGE_GCN, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu,
                                             out=y[s_:e_], x_self=xp[s_:e_])
                                else:
                                    layer(block, (x, x[s_:e_]), ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, w_packed=wp, out=y[s_:e_])
                            k += 1
                        else:
                            for st in pool:
                                cur.wait_stream(st)
                            x = y
                            continue
                    finally:
                        if engine:
                            dataloader.global_blocks = False
                    for st in pool:
                        cur.wait_stream(st)
                    for input_nodes, output_nodes, blocks in dataloader:
                        block = blocks[0].int().to(x.device)
                        h = ops.gather_rows(x, input_nodes)                              # feats[input_nodes]
                        h = layer(block, (h, h[: block.num_dst_nodes()]), ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, w_packed=wp)
                        if post_ln:
                            h = _eval_tail(self, l, h)
                        ops.scatter_rows(h, output_nodes, y)                             # y[output_nodes] = h
                x = y
            return x


SWEEP_STREAMS = int(__import__("os").environ.get("GLNN_SWEEP_STREAMS", "8"))      # streams of the engine-mode chunked sweep (<= 1: the caller's stream only)
_SWEEP_POOLS = {}


def _sweep_streams(device):
    if SWEEP_STREAMS <= 1 or device.type != "cuda":
        return []
    key = (device.index, SWEEP_STREAMS)
    if key not in _SWEEP_POOLS:
        _SWEEP_POOLS[key] = [torch.cuda.Stream(device=device) for _ in range(SWEEP_STREAMS)]
    return _SWEEP_POOLS[key]


class GCN(nn.Module):
    """reference models.py:151-199"""

    def __init__(self, num_layers, input_dim, hidden_dim, output_dim, dropout_ratio, activation, norm_type="none"):
        super().__init__()
        self.num_layers = num_layers
        self.norm_type = norm_type
        self.dropout = nn.Dropout(dropout_ratio)
        self.layers = nn.ModuleList()
        self.norms = nn.ModuleList()
        if num_layers == 1:
            self.layers.append(GraphConv(input_dim, output_dim, activation=activation))
        else:
            self.layers.append(GraphConv(input_dim, hidden_dim, activation=activation))
            self._add_norm(hidden_dim)
            for _ in range(num_layers - 2):
                self.layers.append(GraphConv(hidden_dim, hidden_dim, activation=activation))
                self._add_norm(hidden_dim)
            self.layers.append(GraphConv(hidden_dim, output_dim))

    def _add_norm(self, hidden_dim):
        if self.norm_type == "batch":
            self.norms.append(nn.BatchNorm1d(hidden_dim))
        elif self.norm_type == "layer":
            self.norms.append(nn.LayerNorm(hidden_dim))

    def forward(self, g, feats):
        """reference models.py:189-199: GraphConv (ReLU inside the conv on hidden layers) -> norms[l] -> dropout (NO ReLU behind
        the norm).  train.conf.yaml's cora-style sections use norm_type none, pokec / penn94 GCN use batch."""
        _need_hip(feats, "GCN.forward")
        _check_tail(self)
        h = feats
        h_list = []
        for l, layer in enumerate(self.layers):
            h = layer(g, h)
            if l != self.num_layers - 1:
                h_list.append(h)
                if self.training:
                    if self.norm_type != "none" or self.dropout.p > 0:
                        h = norm_act_drop(h, _norm_of(self, l), self.dropout.p, relu=False)
                elif self.norm_type != "none":
                    with torch.no_grad():
                        h = _eval_tail(self, l, h, relu=False)
        return h_list, h


class Model(nn.Module):
    """Wrapper of different models (reference models.py:347-429)."""

    def __init__(self, conf):
        super().__init__()
        self.model_name = conf["model_name"]
        common = dict(num_layers=conf["num_layers"], input_dim=conf["feat_dim"], hidden_dim=conf["hidden_dim"],
                      output_dim=conf["label_dim"], dropout_ratio=conf["dropout_ratio"])
        if "MLP" in conf["model_name"]:
            self.encoder = MLP(norm_type=conf["norm_type"], **common).to(conf["device"])
        elif "SAGE" in conf["model_name"]:
            self.encoder = SAGE(activation=F.relu, norm_type=conf["norm_type"], **common).to(conf["device"])
        elif "GCN" in conf["model_name"]:
            self.encoder = GCN(activation=F.relu, norm_type=conf["norm_type"], **common).to(conf["device"])
        elif "GAT" in conf["model_name"] or "APPNP" in conf["model_name"]:
            raise NotImplementedError(f"{conf['model_name']}: ablation-only teacher (reference models.py:202-344), "
                                      "outside the MI355X hot-path scope (SURVEY.md section 2 row 5)")
        else:
            raise ValueError(f"Unknown model_name {conf['model_name']}")

    def forward(self, data, feats):
        """data: a graph `g`, a list of blocks, or None for MLPs."""
        if "MLP" in self.model_name:
            if not self.encoder.training:           # logits only: the hidden Linear outputs need not be kept raw
                _need_hip(feats, "MLP.forward")
                _check_tail(self.encoder)
                with torch.no_grad():
                    return self.encoder._forward_hip_eval(feats, want_hidden=False)[1]
            return self.encoder(feats)[1]
        return self.encoder(data, feats)[1]

    def forward_fitnet(self, data, feats):
        if "MLP" in self.model_name:
            return self.encoder(feats)
        return self.encoder(data, feats)

    def inference(self, data, feats):
        if "SAGE" in self.model_name:
            return self.encoder.inference(data, feats)
        return self.forward(data, feats)
