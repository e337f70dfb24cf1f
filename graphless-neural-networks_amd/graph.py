"""CSR graph / block container exposing the handful of DGLGraph methods the reference touches.

The reference passes `dgl.DGLGraph` objects and 1-hop "blocks" around (reference models.py:109,134-137,
train_and_eval.py:41,178-211, utils.py:162-163,176-185, dataloader.py:105).  This container keeps the
same call surface -- num_nodes(), number_of_nodes(), number_of_edges(), num_dst_nodes(), in_degrees(),
out_degrees(), ndata[...], int(), to(device), create_formats_(), subgraph(idx) -- over the ONE layout the
HIP kernels read: CSR over destination rows, int64 indptr, int32 indices, resident in HBM."""
import torch


class CSRGraph:
    def __init__(self, indptr, indices, n_dst, n_src=None):
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32
        assert indptr.numel() == n_dst + 1
        self.indptr = indptr.contiguous()
        self.indices = indices.contiguous()
        self.n_dst = int(n_dst)
        self.n_src = int(n_dst if n_src is None else n_src)
        self.ndata = {}
        self._nnz = None
        self._cache = {}

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_edges(cls, src, dst, num_nodes):
        """dgl.graph((src, dst)) equivalent: edge u->v means v aggregates from u."""
        src = torch.as_tensor(src, dtype=torch.int64)
        dst = torch.as_tensor(dst, dtype=torch.int64)
        order = torch.argsort(dst, stable=True)
        counts = torch.bincount(dst, minlength=num_nodes)
        indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dst.device)
        torch.cumsum(counts, 0, out=indptr[1:])
        return cls(indptr, src[order].to(torch.int32), num_nodes)

    # ---- DGL-like surface -----------------------------------------------------------------------
    @property
    def device(self):
        return self.indptr.device

    def num_nodes(self):
        return self.n_src

    number_of_nodes = num_nodes

    def num_src_nodes(self):
        return self.n_src

    def num_dst_nodes(self):
        return self.n_dst

    def num_edges(self):
        if self._nnz is None:
            self._nnz = int(self.indptr[-1].item())
        return self._nnz

    number_of_edges = num_edges

    def int(self):
        return self            # indices already int32 (reference: block.int(), models.py:134)

    def create_formats_(self):
        return None            # CSR-by-destination is the only format used

    def to(self, device):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if device == self.indptr.device:
            return self
        g = CSRGraph(self.indptr.to(device), self.indices.to(device), self.n_dst, self.n_src)
        g._nnz = self._nnz
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        return g

    def in_degrees(self):
        return self.indptr[1:] - self.indptr[:-1]

    def out_degrees(self):
        return torch.bincount(self.indices.long(), minlength=self.n_src)

    def degree_norms(self):
        """(in_deg.clamp(1)^-1/2, out_deg.clamp(1)^-1/2) as fp32 device vectors, computed once by
        glnn_degrees_f32 (GraphConv norm='both' / feature_prop, reference utils.py:178-179)."""
        if "norms" not in self._cache:
            if self.indptr.is_cuda:
                from . import ops
                in_deg, out_deg = ops.degrees(self.indptr, self.indices, self.n_dst, self.n_src, self.num_edges())
            else:
                in_deg, out_deg = self.in_degrees().float(), self.out_degrees().float()
            self._cache["norms"] = (in_deg.clamp(min=1).pow(-0.5), out_deg.clamp(min=1).pow(-0.5))
        return self._cache["norms"]

    def reverse(self):
        """Transposed graph (CSR over the original SOURCE nodes), cached: used by the aggregation backward."""
        if "rev" not in self._cache:
            dst_all = torch.repeat_interleave(torch.arange(self.n_dst, device=self.device), self.in_degrees())
            src = self.indices.long()
            order = torch.argsort(src, stable=True)
            counts = torch.bincount(src, minlength=self.n_src)
            indptr = torch.zeros(self.n_src + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(counts, 0, out=indptr[1:])
            self._cache["rev"] = CSRGraph(indptr, dst_all[order].to(torch.int32), self.n_src, self.n_dst)
        return self._cache["rev"]

    def row_range(self, start, stop):
        """Destination rows [start, stop) with ALL source columns kept (node-range shard / eval chunk)."""
        lo, hi = int(self.indptr[start].item()), int(self.indptr[stop].item())
        g = CSRGraph((self.indptr[start:stop + 1] - lo).contiguous(), self.indices[lo:hi], stop - start, self.n_src)
        g._nnz = hi - lo
        return g

    def subgraph(self, idx):
        """g.subgraph(idx_obs) (reference train_and_eval.py:324): induced subgraph, nodes relabelled in
        the order of idx."""
        idx = torch.as_tensor(idx, dtype=torch.int64, device=self.device)
        remap = torch.full((self.n_src,), -1, dtype=torch.int64, device=self.device)
        remap[idx] = torch.arange(idx.numel(), device=self.device)
        deg = self.in_degrees()
        dst_all = torch.repeat_interleave(torch.arange(self.n_dst, device=self.device), deg)
        src_new, dst_new = remap[self.indices.long()], remap[dst_all]
        keep = (src_new >= 0) & (dst_new >= 0)
        g = CSRGraph.from_edges(src_new[keep], dst_new[keep], idx.numel())
        g.ndata = {k: v[idx] for k, v in self.ndata.items()}
        return g


class FullNeighborLoader:
    """The reference's `dataloader_eval` (train_and_eval.py:193-202): MultiLayerFullNeighborSampler(1)
    over torch.arange(N), shuffle=False, drop_last=False.  Iterating yields
    (input_nodes, output_nodes, [block]) per chunk of `batch_size` destination nodes in node-id order,
    the chunk's dst nodes first among the block's sources.  `graph` exposes the whole CSR so that
    SAGE.inference can take the unchunked fast path (identical arithmetic: every dst row is independent)."""

    def __init__(self, graph, batch_size):
        self.graph = graph
        self.batch_size = int(batch_size)

    def __len__(self):
        return (self.graph.n_dst + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        g = self.graph
        dev = g.device
        n = g.n_dst
        for s in range(0, n, self.batch_size):
            e = min(n, s + self.batch_size)
            lo, hi = int(g.indptr[s].item()), int(g.indptr[e].item())
            src = g.indices[lo:hi].long()
            output_nodes = torch.arange(s, e, device=dev)
            is_out = torch.zeros(g.n_src, dtype=torch.bool, device=dev)
            is_out[s:e] = True
            uniq = torch.unique(src)
            extra = uniq[~is_out[uniq]]
            input_nodes = torch.cat([output_nodes, extra])
            remap = torch.empty(g.n_src, dtype=torch.int64, device=dev)
            remap[input_nodes] = torch.arange(input_nodes.numel(), device=dev)
            block = CSRGraph((g.indptr[s:e + 1] - lo).contiguous(), remap[src].to(torch.int32), e - s,
                             input_nodes.numel())
            block._nnz = hi - lo
            yield input_nodes, output_nodes, [block]


class MultiLayerNeighborSampler:
    """dgl.dataloading.MultiLayerNeighborSampler(fanouts) (reference train_and_eval.py:179-181): per layer, at most
    fanout[l] uniformly sampled in-neighbours per destination node, without replacement."""

    def __init__(self, fanouts):
        self.fanouts = [int(f) for f in fanouts]


class MultiLayerFullNeighborSampler:
    """dgl.dataloading.MultiLayerFullNeighborSampler(n_layers) (reference train_and_eval.py:193)."""

    def __init__(self, n_layers):
        self.n_layers = int(n_layers)


class NodeDataLoader:
    """dgl.dataloading.NodeDataLoader(g, nids, sampler, batch_size, shuffle, drop_last) (reference
    train_and_eval.py:182-202) over the resident CSR.  Yields (input_nodes, output_nodes, blocks) with
    blocks[0] the outermost (first-layer) block; every block lists its destination nodes first among its sources.
    Sampling runs on the GPU (glnn_sample_neighbors); shuffling uses torch's CPU generator like a torch DataLoader."""

    def __init__(self, g, nids, sampler, batch_size=1, shuffle=False, drop_last=False, num_workers=0, seed=None):
        self.g, self.nids, self.sampler = g, torch.as_tensor(nids, dtype=torch.int64), sampler
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self._epoch = 0
        self._seed = int(torch.initial_seed() if seed is None else seed) & 0x7FFFFFFF
        if isinstance(sampler, MultiLayerFullNeighborSampler) and sampler.n_layers == 1:
            self.graph = g        # lets SAGE.inference take the whole-graph path when nids covers every node

    def __len__(self):
        n = self.nids.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _block(self, seeds, fanout, rng_seed):
        from . import ops
        g, dev = self.g, self.g.device
        if fanout is None:                                   # full neighbourhood
            deg = (g.indptr[seeds + 1] - g.indptr[seeds])
            starts = g.indptr[seeds]
            indptr = torch.zeros(seeds.numel() + 1, dtype=torch.int64, device=dev)
            torch.cumsum(deg, 0, out=indptr[1:])
            pos = torch.arange(int(indptr[-1].item()), device=dev) - torch.repeat_interleave(indptr[:-1] - starts, deg)
            src = g.indices[pos].long()
        else:
            smp, cnt = ops.sample_neighbors(g.indptr, g.indices, seeds, fanout, rng_seed)
            indptr = torch.zeros(seeds.numel() + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt.long(), 0, out=indptr[1:])
            valid = torch.arange(fanout, device=dev).unsqueeze(0) < cnt.unsqueeze(1)
            src = smp[valid].long()                          # row-major: edges stay grouped by destination
        # sources of the block = seeds first, then the other referenced nodes in ascending id order (what sort-based
        # unique() gave before): one flag scatter + one prefix sum over the node range instead of a device sort per layer
        ns = seeds.numel()
        flag = torch.zeros(g.n_src, dtype=torch.int64, device=dev)
        flag[src] = 1
        flag[seeds] = 0
        remap = torch.cumsum(flag, 0)
        remap += ns - 1                                      # flagged node -> ns + (its rank among the flagged)
        extra = torch.nonzero(flag).squeeze(1)
        remap[seeds] = torch.arange(ns, device=dev)
        input_nodes = torch.cat([seeds, extra])
        block = CSRGraph(indptr, remap[src].to(torch.int32), ns, input_nodes.numel())
        block._nnz = int(src.numel())
        return input_nodes, block

    def __iter__(self):
        self._epoch += 1
        n = self.nids.numel()
        order = torch.randperm(n) if self.shuffle else torch.arange(n)
        dev = self.g.device
        fanouts = self.sampler.fanouts if isinstance(self.sampler, MultiLayerNeighborSampler) else [None] * self.sampler.n_layers
        for b, s in enumerate(range(0, n, self.batch_size)):
            idx = order[s:s + self.batch_size]
            if self.drop_last and idx.numel() < self.batch_size:
                break
            output_nodes = self.nids[idx].to(dev)
            seeds, blocks = output_nodes, []
            for l in reversed(range(len(fanouts))):          # last layer's block is sampled first
                rng = (self._seed * 1000003 + self._epoch * 7919 + b * 31 + l) & 0xFFFFFFFF
                seeds, blk = self._block(seeds, fanouts[l], rng)
                blocks.insert(0, blk)
            yield seeds, output_nodes, blocks
