"""Golden vectors for the teacher TRAINING step, produced by the reference's own Python.

Reference code that runs here, unmodified, imported from /root/reference:
  * train_and_eval.train_sage (:32-56) driving models.SAGE.forward (models.py:101-119) over FIXED sampled blocks:
    feats[input_nodes] -> per layer SAGEConv(block, (h, h[:n_dst])) -> BatchNorm1d(train) -> relu -> dropout(p=0) ->
    log_softmax -> NLLLoss -> .item() -> backward -> Adam.step(), two epochs over three batches;
  * train_and_eval.train (:12-29) driving models.GCN.forward (models.py:189-199) on a full graph, five steps.
As in make_teacher_golden.py the two dgl layers are differentiable torch stand-ins of dgl 0.6.1's published semantics
(dgl itself is absent), so what this pins is everything the reference writes around them: layer/BN/ReLU order, which rows
are h_dst, the loss and its scaling, Adam (L2 decay folded into the gradient) and the order of optimiser steps.  The
blocks are drawn by a small numpy sampler here and stored in the fixture: the sampler itself is not part of the parity
claim (dgl's RNG cannot be restated), the arithmetic on given blocks is.

    python tests/golden/make_teacher_train_golden.py        (build container only)
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_teacher_golden as mtg          # noqa: E402  (Block + import stubs)
from graphgen import random_graph          # noqa: E402


def _adj(block):
    n_dst = block.n_dst
    dst = np.repeat(np.arange(n_dst), np.diff(block.indptr))
    idx = torch.from_numpy(np.stack([dst, block.indices.astype(np.int64)]))
    return torch.sparse_coo_tensor(idx, torch.ones(len(dst)), (n_dst, block.n_src))


class TrainSAGEConv(nn.Module):
    """dgl 0.6.1 SAGEConv(in, out, 'gcn'), differentiable: rst = fc_neigh((A h_src + h_dst) / (in_deg + 1))."""

    def __init__(self, in_feats, out_feats, aggregator_type):
        super().__init__()
        assert aggregator_type == "gcn"
        self.fc_neigh = nn.Linear(in_feats, out_feats)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, block, feat):
        h_src, h_dst = feat
        deg = torch.from_numpy(np.diff(block.indptr)).float().unsqueeze(1)
        return self.fc_neigh((torch.sparse.mm(_adj(block), h_src) + h_dst) / (deg + 1))


class TrainGraphConv(nn.Module):
    """dgl 0.6.1 GraphConv(in, out, norm='both', activation), differentiable."""

    def __init__(self, in_feats, out_feats, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats))
        nn.init.xavier_uniform_(self.weight)
        self._act, self._in, self._out = activation, in_feats, out_feats

    def forward(self, g, feat):
        a = _adj(g)
        in_deg = torch.from_numpy(np.diff(g.indptr)).float().clamp(min=1)
        out_deg = torch.bincount(torch.from_numpy(g.indices.astype(np.int64)), minlength=g.n_src).float().clamp(min=1)
        h = feat * out_deg.pow(-0.5).unsqueeze(1)
        if self._in > self._out:
            rst = torch.sparse.mm(a, h @ self.weight)
        else:
            rst = torch.sparse.mm(a, h) @ self.weight
        rst = rst * in_deg.pow(-0.5).unsqueeze(1) + self.bias
        return self._act(rst) if self._act is not None else rst


def sample_blocks(indptr, indices, seeds, fanouts, rs):
    """MultiLayerNeighborSampler-shaped blocks (outermost first): per layer at most fanout in-neighbours per destination,
    without replacement; a block's sources = its destinations first, then the other sampled nodes in ascending id order."""
    blocks = []
    for fan in reversed(fanouts):
        ip, src = [0], []
        for v in seeds:
            nb = indices[indptr[v]:indptr[v + 1]]
            if len(nb) > fan:
                nb = nb[np.sort(rs.choice(len(nb), fan, replace=False))]
            src.append(nb.astype(np.int64))
            ip.append(ip[-1] + len(nb))
        src = np.concatenate(src) if src else np.zeros(0, np.int64)
        extra = np.setdiff1d(np.unique(src), seeds)
        input_nodes = np.concatenate([seeds, extra])
        remap = np.full(len(indptr) - 1, -1, np.int64)
        remap[input_nodes] = np.arange(len(input_nodes))
        blocks.insert(0, mtg.Block(np.asarray(ip, np.int64), remap[src].astype(np.int32), len(seeds), len(input_nodes)))
        seeds = input_nodes
    return seeds, blocks


def main():
    mtg._stub_modules()
    dgl_nn = sys.modules["dgl.nn"]
    dgl_nn.SAGEConv, dgl_nn.GraphConv = TrainSAGEConv, TrainGraphConv
    sys.modules["dgl"].function = None
    import types
    for name in ("dgl.function", "ogb", "ogb.nodeproppred"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["ogb.nodeproppred"].Evaluator = type("Evaluator", (), {})
    sys.modules["dgl.function"].copy_u = sys.modules["dgl.function"].sum = None
    sys.path.insert(0, mtg.REF)
    import models as ref_models            # noqa: reference, unmodified
    import train_and_eval as ref_te        # noqa
    torch.set_num_threads(1)
    out = {}

    # ---- train_sage over fixed blocks -----------------------------------------------------------------------------------
    n, dims, fanouts, bsz = 700, [20, 32, 32, 6], [4, 6, 8], 48
    indptr, indices = random_graph(n, 9, seed=41, power=0.5, isolated=5, hub=200)
    rs = np.random.RandomState(41)
    feats = rs.standard_normal((n, dims[0])).astype(np.float32)
    labels = rs.randint(0, dims[-1], n).astype(np.int64)
    train_ids = rs.permutation(n)[:3 * bsz]
    batches = []
    for b in range(3):
        seeds = train_ids[b * bsz:(b + 1) * bsz]
        input_nodes, blocks = sample_blocks(indptr, indices, seeds, fanouts, rs)
        batches.append((torch.from_numpy(input_nodes), torch.from_numpy(seeds), blocks))
        out[f"sage.b{b}.input_nodes"], out[f"sage.b{b}.output_nodes"] = input_nodes, seeds
        for l, blk in enumerate(blocks):
            out[f"sage.b{b}.l{l}.indptr"], out[f"sage.b{b}.l{l}.indices"] = blk.indptr, blk.indices
            out[f"sage.b{b}.l{l}.n_src"] = np.int64(blk.n_src)
    for norm, wd in (("batch", 0.0), ("none", 5e-4)):
        torch.manual_seed(41)
        conf = dict(model_name="SAGE", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                    norm_type=norm, device="cpu")
        model = ref_models.Model(conf)
        with torch.no_grad():
            for lay in model.encoder.layers:
                lay.fc_neigh.bias.copy_(torch.randn_like(lay.fc_neigh.bias) * 0.1)
        optimizer = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=wd)        # train_teacher.py:234-236
        step_losses = []
        base = nn.NLLLoss()                                                               # train_teacher.py:237

        def criterion(o, y):
            l = base(o, y)
            step_losses.append(float(l.item()))
            return l

        tag = f"sage.{norm}"
        for k, v in model.state_dict().items():
            out[f"{tag}.init.{k}"] = v.numpy().copy()
        tf, tl = torch.from_numpy(feats), torch.from_numpy(labels)
        epoch_losses = [ref_te.train_sage(model, batches, tf, tl, criterion, optimizer) for _ in range(2)]
        out[f"{tag}.epoch_losses"], out[f"{tag}.step_losses"] = np.asarray(epoch_losses), np.asarray(step_losses)
        for k, v in model.state_dict().items():
            out[f"{tag}.final.{k}"] = v.numpy().copy()
        out[f"{tag}.wd"] = np.float64(wd)
        # gradients of the very first step, from a fresh copy of the initial state
        model.load_state_dict({k[len(tag) + 6:]: torch.from_numpy(np.asarray(v)) for k, v in out.items() if k.startswith(f"{tag}.init.")})
        model.train()
        inp, outn, blks = batches[0]
        loss = base(model(blks, tf[inp]).log_softmax(dim=1), tl[outn])
        model.zero_grad()
        loss.backward()
        for pname, p in model.named_parameters():
            out[f"{tag}.grad0.{pname}"] = p.grad.numpy().copy()
    out["sage.indptr"], out["sage.indices"], out["sage.feats"], out["sage.labels"] = indptr, indices, feats, labels
    out["sage.dims"] = np.asarray(dims)

    # ---- train (full-graph GCN) ---------------------------------------------------------------------------------------
    n2, dims2 = 300, [60, 16, 5]
    ip2, ix2 = random_graph(n2, 3, seed=42, symmetric=True, self_loops=True)
    feats2 = (rs.standard_normal((n2, dims2[0])) * 0.5).astype(np.float32)
    labels2 = rs.randint(0, dims2[-1], n2).astype(np.int64)
    idx_train = np.sort(rs.permutation(n2)[:60]).astype(np.int64)
    torch.manual_seed(42)
    conf2 = dict(model_name="GCN", num_layers=2, feat_dim=dims2[0], hidden_dim=dims2[1], label_dim=dims2[-1], dropout_ratio=0.0,
                 norm_type="none", device="cpu")
    gcn = ref_models.Model(conf2)
    opt2 = torch.optim.Adam(gcn.parameters(), lr=0.01, weight_decay=1e-3)                  # cora GCN: train.conf.yaml:12-15
    for k, v in gcn.state_dict().items():
        out[f"gcn.init.{k}"] = v.numpy().copy()
    g2 = mtg.Block(ip2, ix2, n2, n2)
    losses = [ref_te.train(gcn, g2, torch.from_numpy(feats2), torch.from_numpy(labels2), nn.NLLLoss(), opt2, torch.from_numpy(idx_train))
              for _ in range(5)]
    out["gcn.losses"] = np.asarray(losses)
    for k, v in gcn.state_dict().items():
        out[f"gcn.final.{k}"] = v.numpy().copy()
    out.update({"gcn.indptr": ip2, "gcn.indices": ix2, "gcn.feats": feats2, "gcn.labels": labels2, "gcn.idx_train": idx_train,
                "gcn.dims": np.asarray(dims2)})

    # ---- train (full-graph GCN) WITH a norm layer (round 3): reference models.py:189-199 puts norms[l] BEHIND the GraphConv's ReLU
    # and applies no ReLU after it; train.conf.yaml:206-213, 231-238 (pokec / penn94 GCN) use norm_type batch, weight_decay 0.001
    for tag, norm in (("gcnbn", "batch"), ("gcnln", "layer")):
        dims3 = [60, 16, 16, 5]
        torch.manual_seed(7 if norm == "batch" else 8)
        conf3 = dict(model_name="GCN", num_layers=3, feat_dim=dims3[0], hidden_dim=dims3[1], label_dim=dims3[-1], dropout_ratio=0.0,
                     norm_type=norm, device="cpu")
        gcn3 = ref_models.Model(conf3)
        with torch.no_grad():
            for nm in gcn3.encoder.norms:
                nm.weight.uniform_(0.5, 1.5)
                nm.bias.uniform_(-0.2, 0.2)
        opt3 = torch.optim.Adam(gcn3.parameters(), lr=0.01, weight_decay=1e-3)
        for k, v in gcn3.state_dict().items():
            out[f"{tag}.init.{k}"] = v.numpy().copy()
        losses3 = [ref_te.train(gcn3, g2, torch.from_numpy(feats2), torch.from_numpy(labels2), nn.NLLLoss(), opt3, torch.from_numpy(idx_train))
                   for _ in range(5)]
        out[f"{tag}.losses"] = np.asarray(losses3)
        for k, v in gcn3.state_dict().items():
            out[f"{tag}.final.{k}"] = v.numpy().copy()
        gcn3.eval()
        with torch.no_grad():
            out[f"{tag}.eval_logits"] = gcn3(g2, torch.from_numpy(feats2)).numpy().copy()
        out[f"{tag}.dims"] = np.asarray(dims3)

    path = os.path.join(HERE, "teacher_training.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; sage epoch losses",
          out["sage.batch.epoch_losses"], out["sage.none.epoch_losses"], "gcn losses", losses)


if __name__ == "__main__":
    main()
