"""Drop-in modules for the two dgl layers the reference's hot path uses, computing on libglnn_hip.so.

  SAGEConv(in, out, "gcn")(block, (h, h_dst))   <- dgl.nn.SAGEConv, reference models.py:84-99,112,138
  GraphConv(in, out, activation=)(g, h)          <- dgl.nn.GraphConv, reference models.py:170-187,193

Parameter names follow dgl 0.6.1 so that a reference `model.pth` loads: SAGEConv.fc_neigh.{weight,bias}
(weight [out,in], xavier_uniform gain=relu; no fc_self for "gcn"), GraphConv.{weight [in,out] xavier_uniform,
bias zeros}.  Only what the reference constructs is implemented; anything else raises."""

import torch
import torch.nn as nn

from . import ops
from .autograd import GraphConvFn, SpmmFn, graphconv_fwd, linear_fn


FUSED_SAGE_MAX_IN = 256   # aggregate-first layers with d_in, d_out <= 256 take the single-launch K1F kernel.  Interleaved
                          # A/B on one MI355X, products shape (scripts/ab_fused.py): 100->256 10.6 vs 12.5 ms,
                          # 128->256 11.4 vs 12.2 ms, 256->256 23.6 vs 25.1 ms (fused vs aggregation + GEMM)


class SAGEConv(nn.Module):
    def __init__(self, in_feats, out_feats, aggregator_type, bias=True):
        super().__init__()
        if aggregator_type != "gcn":
            raise NotImplementedError("the reference only builds SAGEConv(..., 'gcn') (models.py:84-99)")
        self._in_feats, self._out_feats = in_feats, out_feats
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=nn.init.calculate_gain("relu"))

    def fused_eligible(self):
        """Aggregate-first layers with d_in, d_out <= 256 run on the single-launch K1F kernel."""
        return self._in_feats <= self._out_feats and self._in_feats <= FUSED_SAGE_MAX_IN and self._out_feats <= 256

    def forward(self, graph, feat, ep_scale=None, ep_shift=None, relu=False, w_packed=None, out=None):
        """out = fc_neigh((sum_{u->v} h[u] + h_dst[v]) / (deg(v)+1)).  ep_* / relu: optional fused tail
        (eval-mode BatchNorm + ReLU of the caller) used by SAGE.inference; bias is folded by the caller then.
        w_packed: ops.pack_weight(fc_neigh.weight) of a caller that sweeps many blocks with the same weights (the chunked
        inference loop packs once per layer instead of once per chunk).  out (inference only): where the layer's rows go
        (SAGE.inference hands a placed buffer, ops.placed_for_gather, to the layers whose output the next layer gathers)."""
        h_src, h_dst = feat if isinstance(feat, tuple) else (feat, feat)
        n_dst = graph.num_dst_nodes()
        if h_dst.shape[0] != n_dst:
            raise ValueError("SAGEConv: h_dst must hold the block's destination rows")
        w, b = self.fc_neigh.weight, self.fc_neigh.bias
        needs_grad = torch.is_grad_enabled() and (h_src.requires_grad or w.requires_grad)
        if needs_grad:
            agg = SpmmFn.apply(graph, h_src, ops.AGG_SAGE_GCN)
            return linear_fn(agg, w, b)
        fused_tail = ep_scale is not None or ep_shift is not None or relu
        shift = ep_shift if fused_tail else b
        # (no ops.HubPlan here: a whole-graph launch is long enough to hide its hub rows behind the heaviest-first tile order -- measured on
        #  the arxiv-shaped graph the extra launch costs 10-60 us per layer and gains nothing; row shards use one: glnn_amd/dist.py)
        if self._in_feats > self._out_feats:
            # project first (linear commutes with the mean): aggregate at the narrower width
            hw = ops.gemm(ops.as_feat(h_src), w)
            return ops.spmm(graph.indptr, graph.indices, hw, n_dst, ops.AGG_SAGE_GCN, ep_scale=ep_scale,
                            ep_shift=shift, relu=relu, out=out)
        if self._in_feats <= FUSED_SAGE_MAX_IN and self._out_feats <= 256:
            # aggregation + projection + epilogue in one launch: the aggregated rows never reach HBM
            order = graph.fused_tile_order() if n_dst == graph.n_dst else None
            return ops.sage_fused(graph.indptr, graph.indices, h_src, n_dst, w, ep_scale=ep_scale, ep_shift=shift, relu=relu,
                                  x_self=h_dst, w_packed=w_packed, tile_order=order, out=out)
        agg = ops.spmm(graph.indptr, graph.indices, h_src, n_dst, ops.AGG_SAGE_GCN)
        return ops.gemm(agg, w, ep_scale=ep_scale, ep_shift=shift, relu=relu, out=out)


class GraphConv(nn.Module):
    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False):
        super().__init__()
        if norm != "both" or not weight:
            raise NotImplementedError("the reference only builds GraphConv(in, out, activation=...) (models.py:170-187)")
        self._in_feats, self._out_feats = in_feats, out_feats
        self._activation = activation
        self._allow_zero_in_degree = allow_zero_in_degree
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.empty(out_feats)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, graph, feat):
        if not self._allow_zero_in_degree and graph.has_zero_in_degree():
            raise RuntimeError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid "
                               "(dgl GraphConv semantics; add self-loops or set allow_zero_in_degree).")
        act = self._activation
        relu = act is not None and getattr(act, "__name__", "") == "relu"
        if act is not None and not relu:
            raise NotImplementedError("GraphConv: only activation=F.relu or None is used by the reference")
        if torch.is_grad_enabled() and (feat.requires_grad or self.weight.requires_grad):
            return GraphConvFn.apply(graph, feat, self.weight, self.bias, relu)
        return graphconv_fwd(graph, ops.as_feat(feat), self.weight, self.bias, relu)[0]
