"""Golden vectors for the TEACHER composition, produced by the reference's own Python.

The reference's graph arithmetic lives in dgl==0.6.1, which is absent here, so the *layer* arithmetic cannot be
pinned by the reference.  What CAN be pinned is everything the reference itself writes around those
layers: `models.SAGE.inference` (layer-wise sweep over a dataloader of 1-hop blocks, BatchNorm(eval) -> activation ->
dropout order, `y[output_nodes] = h`, last layer raw), `models.GCN.forward`, `Model` dispatch and `state_dict`
layout.  This script imports the reference's models.py with `dgl.nn.SAGEConv / GraphConv` replaced by small
torch.sparse stand-ins that implement dgl's PUBLISHED semantics (independent of this repo's kernels and of its C
oracle), runs the reference code on small seeded graphs and stores inputs + outputs.

    python tests/golden/make_teacher_golden.py        (build container only: needs /root/reference)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from graphgen import random_graph  # noqa: E402


class Block:
    """What the reference touches on a DGL block / graph: num_dst_nodes(), int(), to()."""

    def __init__(self, indptr, indices, n_dst, n_src):
        self.indptr, self.indices, self.n_dst, self.n_src = indptr, indices, n_dst, n_src

    def num_dst_nodes(self):
        return self.n_dst

    def int(self):
        return self

    def to(self, device):
        return self

    def adj(self):
        return torch.sparse_csr_tensor(torch.from_numpy(self.indptr), torch.from_numpy(self.indices.astype(np.int64)),
                                       torch.ones(len(self.indices), dtype=torch.float64), size=(self.n_dst, self.n_src))


class StubSAGEConv(nn.Module):
    """dgl 0.6.1 SAGEConv(in, out, 'gcn'): h = (sum_{u->v} h_src[u] + h_dst[v]) / (in_deg(v)+1); rst = fc_neigh(h)."""

    def __init__(self, in_feats, out_feats, aggregator_type):
        super().__init__()
        assert aggregator_type == "gcn"
        self.fc_neigh = nn.Linear(in_feats, out_feats)

    def forward(self, block, feat):
        h_src, h_dst = feat
        a = block.adj()
        neigh = (a @ h_src.double()).float()
        deg = torch.from_numpy(np.diff(block.indptr)).float().unsqueeze(1)
        return self.fc_neigh((neigh + h_dst) / (deg + 1))


class StubGraphConv(nn.Module):
    """dgl 0.6.1 GraphConv(in, out, norm='both', activation): D_out^-1/2 on sources, weight first when in > out,
    D_in^-1/2 on destinations, + bias, activation."""

    def __init__(self, in_feats, out_feats, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats))
        nn.init.xavier_uniform_(self.weight)
        self._act, self._in, self._out = activation, in_feats, out_feats

    def forward(self, g, feat):
        a = g.adj()
        in_deg = torch.from_numpy(np.diff(g.indptr)).double().clamp(min=1)
        out_deg = torch.bincount(torch.from_numpy(g.indices.astype(np.int64)), minlength=g.n_src).double().clamp(min=1)
        h = feat.double() * out_deg.pow(-0.5).unsqueeze(1)
        if self._in > self._out:
            rst = a @ (h @ self.weight.double())
        else:
            rst = (a @ h) @ self.weight.double()
        rst = (rst * in_deg.pow(-0.5).unsqueeze(1) + self.bias.double()).float()
        return self._act(rst) if self._act is not None else rst


def _stub_modules():
    dgl = types.ModuleType("dgl")
    dgl_nn = types.ModuleType("dgl.nn")
    dgl_nn.SAGEConv, dgl_nn.GraphConv = StubSAGEConv, StubGraphConv
    dgl_nn.APPNPConv = dgl_nn.GATConv = type("Unused", (), {})
    dgl.nn = dgl_nn
    sys.modules.update({"dgl": dgl, "dgl.nn": dgl_nn})


def full_neighbor_blocks(indptr, indices, n, batch_size):
    """The reference's dataloader_eval (train_and_eval.py:193-202): 1-hop full-neighbour blocks over arange(N),
    in node-id order, dst nodes first among the block's sources."""
    for s in range(0, n, batch_size):
        e = min(n, s + batch_size)
        lo, hi = indptr[s], indptr[e]
        src = indices[lo:hi].astype(np.int64)
        out_nodes = np.arange(s, e)
        extra = np.setdiff1d(np.unique(src), out_nodes)
        input_nodes = np.concatenate([out_nodes, extra])
        remap = np.full(n, -1, np.int64)
        remap[input_nodes] = np.arange(len(input_nodes))
        blk = Block((indptr[s:e + 1] - lo).astype(np.int64), remap[src].astype(np.int32), e - s, len(input_nodes))
        yield torch.from_numpy(input_nodes), torch.from_numpy(out_nodes), [blk]


def randomize_norms(model, rs):
    with torch.no_grad():
        for bn in model.encoder.norms:
            h = bn.weight.shape[0]
            bn.weight.copy_(torch.from_numpy(rs.uniform(.5, 1.5, h).astype(np.float32)))
            bn.bias.copy_(torch.from_numpy(rs.uniform(-.2, .2, h).astype(np.float32)))
            bn.running_mean.copy_(torch.from_numpy(rs.uniform(-.3, .3, h).astype(np.float32)))
            bn.running_var.copy_(torch.from_numpy(rs.uniform(.5, 1.5, h).astype(np.float32)))


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import models as ref_models        # the reference's models.py, unmodified
    torch.set_num_threads(1)
    out = {}
    # ---- SAGE.inference (reference models.py:121-148) on an arxiv-like small graph ----
    n, dims, bs = 900, [24, 48, 48, 10], 128
    indptr, indices = random_graph(n, 8, seed=21, power=0.6, isolated=6, hub=300)
    rs = np.random.RandomState(21)
    feats = rs.standard_normal((n, dims[0])).astype(np.float32)
    torch.manual_seed(21)
    conf = dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.5,
                norm_type="batch", device="cpu")
    model = ref_models.Model(conf)
    randomize_norms(model, rs)
    model.eval()
    with torch.no_grad():
        logits = model.inference(list(full_neighbor_blocks(indptr, indices, n, bs)), torch.from_numpy(feats))
    out.update({"sage.indptr": indptr, "sage.indices": indices, "sage.feats": feats, "sage.logits": logits.numpy(),
                "sage.dims": np.asarray(dims), "sage.batch_size": np.int64(bs)})
    for k, v in model.state_dict().items():
        out[f"sage.sd.{k}"] = v.numpy()
    # ---- GCN.forward (reference models.py:189-199), cora-like: in > out on both layers, no norm, self-loops ----
    n2, dims2 = 400, [120, 16, 5]
    ip2, ix2 = random_graph(n2, 3, seed=22, symmetric=True, self_loops=True)
    feats2 = (rs.standard_normal((n2, dims2[0])) * 0.3).astype(np.float32)
    torch.manual_seed(22)
    conf2 = dict(model_name="GCN", num_layers=2, feat_dim=dims2[0], hidden_dim=dims2[1], label_dim=dims2[-1], dropout_ratio=0.8,
                 norm_type="none", device="cpu")
    gcn = ref_models.Model(conf2)
    with torch.no_grad():
        for lay in gcn.encoder.layers:
            lay.bias.copy_(torch.randn_like(lay.bias) * 0.1)
    gcn.eval()
    with torch.no_grad():
        h_list, logits2 = gcn.forward_fitnet(Block(ip2, ix2, n2, n2), torch.from_numpy(feats2))
    out.update({"gcn.indptr": ip2, "gcn.indices": ix2, "gcn.feats": feats2, "gcn.logits": logits2.numpy(),
                "gcn.h0": h_list[0].numpy(), "gcn.dims": np.asarray(dims2)})
    for k, v in gcn.state_dict().items():
        out[f"gcn.sd.{k}"] = v.numpy()
    path = os.path.join(HERE, "teacher_composition.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; sage logits", logits.shape, "gcn logits", logits2.shape)


if __name__ == "__main__":
    main()
