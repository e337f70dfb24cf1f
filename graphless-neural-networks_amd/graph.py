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
        self.gindices = None       # outermost training block only: the edges' GLOBAL source ids (see NodeDataLoader)
        self.dst_nodes = None
        self.t_indptr = self.t_indices = self.inv_deg = None      # inner training blocks of an engine-mode loader: the transposed block (+ self) and 1/(deg+1)

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
            from . import ops
            self._cache["norms"] = ops.degrees(self.indptr, self.indices, self.n_dst, self.n_src, self.num_edges(),
                                               transform=ops.DEG_RSQRT_CLAMP1)
        return self._cache["norms"]

    def inv_deg_plus1(self):
        """1 / (in_deg + 1) per destination row (fp32, cached): the divisor of the SAGE-"gcn" aggregator, i.e. the
        col_scale of its backward over the transposed block."""
        if "inv1" not in self._cache:
            from . import ops
            self._cache["inv1"] = ops.degrees(self.indptr, None, self.n_dst, self.n_src, 0, want_out=False, transform=ops.DEG_INV_PLUS1)[0]
        return self._cache["inv1"]

    def fused_tile_order(self):
        """The 32-row tiles of a fused aggregate + project launch over ALL destination rows, heaviest row first (cached int32 permutation;
        None on graphs too small for it to matter): hub rows of a power-law graph start first (ops.sage_fused(tile_order=...))."""
        if self.n_dst < 4096 or not self.indptr.is_cuda:
            return None
        if "tile_order" not in self._cache:
            from . import ops
            self._cache["tile_order"] = ops.fused_tile_order(self.indptr, self.n_dst)
        return self._cache["tile_order"]

    def hub_plan(self):
        """ops.HubPlan over ALL destination rows (cached; None without hub rows or off the GPU): the rows of more than
        glnn_hub_row_threshold() in-edges are gathered segment by segment by one workgroup each in front of an aggregation launch
        (ops.spmm / ops.sage_fused(hub=...)) -- what matters for SHORT launches (an arxiv-sized graph, a row shard)."""
        if not self.indptr.is_cuda or self.n_dst == 0:
            return None
        if "hub_plan" not in self._cache:
            from . import ops
            self._cache["hub_plan"] = ops.hub_plan(self.indptr, self.n_dst)
        return self._cache["hub_plan"]

    def has_zero_in_degree(self):
        """dgl GraphConv's `(graph.in_degrees() == 0).any()` check, evaluated once per graph."""
        if "zero_in" not in self._cache:
            self._cache["zero_in"] = bool((self.in_degrees() == 0).any()) if self.n_dst else False
        return self._cache["zero_in"]

    def transposed(self, add_self=False):
        """The transposed graph (CSR over the original SOURCE nodes, rows sorted), cached: what the aggregation's
        backward gathers over (glnn_csr_transpose).  add_self: plus one entry u <- u per destination u < n_dst, the
        h_dst term of the SAGE-"gcn" aggregator."""
        key = ("tr", bool(add_self))
        if key not in self._cache:
            from . import ops
            t_indptr, t_indices = ops.csr_transpose(self.indptr, self.indices, self.n_dst, self.n_src, self.num_edges(), add_self)
            t = CSRGraph(t_indptr, t_indices, self.n_src, self.n_dst)
            t._nnz = self.num_edges() + (self.n_dst if add_self else 0)
            self._cache[key] = t
        return self._cache[key]

    def reverse(self):
        """dgl-style reversed graph (every edge u->v becomes v->u): the transposed CSR."""
        if self.indptr.is_cuda:
            return self.transposed(False)
        if "rev" not in self._cache:        # host-side index algebra for graphs that live on the CPU
            dst_all = torch.repeat_interleave(torch.arange(self.n_dst), self.in_degrees())
            src = self.indices.long()
            order = torch.argsort(src, stable=True)
            indptr = torch.zeros(self.n_src + 1, dtype=torch.int64)
            torch.cumsum(torch.bincount(src, minlength=self.n_src), 0, out=indptr[1:])
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
        self.global_blocks = False      # True (set by SAGE.inference for its chunked sweep): row-range blocks with global source ids, see __iter__

    def __len__(self):
        return (self.graph.n_dst + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        from . import ops
        g = self.graph
        if not g.indptr.is_cuda:
            raise RuntimeError("FullNeighborLoader: blocks are built on the GPU (move the graph with g.to(device))")
        n = g.n_dst
        bounds = list(range(0, n, self.batch_size)) + [n]
        key = ("loader_offs", self.batch_size)
        if key not in g._cache:                                                   # ONE read-back per (graph, batch size): a sweep per layer re-uses it
            g._cache[key] = g.indptr[torch.tensor(bounds, device=g.device)].tolist()
        offs = g._cache[key]
        for b in range(len(bounds) - 1):
            s, e = bounds[b], bounds[b + 1]
            if getattr(self, "global_blocks", False):
                output_nodes = None             # (= arange(s, e) = block.dst_range: not materialised -- one launch and one allocation per chunk saved)
                # engine mode (round 6, set by SAGE.inference for its sweep): the chunk's block IS rows [s, e) of the resident CSR -- absolute
                # row offsets into the one indices array, GLOBAL source ids -- so nothing is built, relabelled or gathered: the consumer's
                # aggregation reads its source rows straight from the layer's input matrix (input_nodes = None)
                block = CSRGraph(g.indptr[s:e + 1], g.indices, e - s, g.n_src)
                block._nnz = offs[b + 1] - offs[b]
                block.dst_range = (s, e)
                block._cache["tile_order"] = None      # (a 4096-row launch: the order would cost six small launches per chunk)
                yield None, output_nodes, [block]
                continue
            output_nodes = torch.arange(s, e, device=g.device)
            indptr, indices, _, input_nodes, nnz, n_src = ops.block_build(output_nodes, g.indptr, g.indices, nnz_cap=offs[b + 1] - offs[b],
                                                                          n_nodes=g.n_src)
            block = CSRGraph(indptr, indices, e - s, n_src)
            block._nnz = nnz
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


_SIDE_STREAMS = {}          # device index -> the stream batches are built on, shared by all loaders of that device


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
        self.prefetch, self._side = True, None
        # batches are built by a worker THREAD (below); GLNN_LOADER_THREAD=0 / .threaded = False: by the consumer's thread, one ahead
        self.threaded = __import__("os").environ.get("GLNN_LOADER_THREAD", "1") != "0"
        self.depth = 2                  # finished batches the worker may hold ready
        # True (set by train_and_eval.train_sage for its epoch): the OUTERMOST block is built as a global-id block only -- indptr + the
        # edges' global source ids; no frontier table, no local relabelling, `input_nodes` is yielded as None -- because TeacherEngine
        # gathers layer 0 straight from the feature matrix.  That block is the widest of a batch: 60 % of the sampler's device time.
        self.global_first_block = False
        # the whole-graph path of SAGE.inference is only valid when the loader sweeps EVERY node in id order
        if isinstance(sampler, MultiLayerFullNeighborSampler) and sampler.n_layers == 1 and not shuffle \
                and self.nids.numel() == g.num_dst_nodes() and bool((self.nids.cpu() == torch.arange(g.num_dst_nodes())).all()):
            self.graph = g

    def __len__(self):
        n = self.nids.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _block(self, seeds, fanout, rng_seed, want_global=False, global_only=False):
        """One 1-hop block over `seeds` on the device (glnn_sample_neighbors + glnn_block_build): work and memory are
        proportional to the frontier (ns * fanout), not to the graph; one host read-back (edge / source-node counts)."""
        from . import ops
        g = self.g
        if fanout is None:                                   # full neighbourhood: the edge count bounds the buffers
            deg_sum = int((g.indptr[seeds + 1] - g.indptr[seeds]).sum().item())
            indptr, indices, gidx, input_nodes, nnz, n_src = ops.block_build(seeds, g.indptr, g.indices, nnz_cap=deg_sum,
                                                                              want_global=want_global, n_nodes=g.n_src, global_only=global_only)
        else:
            smp, cnt = ops.sample_neighbors(g.indptr, g.indices, seeds, fanout, rng_seed)
            indptr, indices, gidx, input_nodes, nnz, n_src = ops.block_build(seeds, smp_src=smp, smp_cnt=cnt, want_global=want_global,
                                                                              n_nodes=g.n_src, global_only=global_only)
        if global_only:                                      # sources = the graph's nodes themselves: a CSR whose column ids are global ids
            n_src, indices = g.n_src, gidx
        block = CSRGraph(indptr, indices, seeds.numel(), n_src)
        block._nnz = nnz
        if want_global:
            block.gindices, block.dst_nodes = gidx, seeds       # the same edges with global source ids (TeacherEngine, layer 0)
        return input_nodes, block

    def _batch(self, b, idx, fanouts, epoch=None, global_first=None):
        """global_first: the value of `global_first_block` the ITERATOR was created under (snapshotted by __iter__, like `epoch`): a worker
        thread that is still building batches when the consumer flips the flag back keeps building the shape its consumer expects."""
        dev = self.g.device
        epoch = self._epoch if epoch is None else epoch
        gfb = self.global_first_block if global_first is None else global_first
        output_nodes = self.nids[idx].to(dev) if self.nids.device != dev else self.nids[idx.to(dev)]
        seeds, blocks = output_nodes, []
        for l in reversed(range(len(fanouts))):          # last layer's block is sampled first
            rng = (self._seed * 1000003 + epoch * 7919 + b * 31 + l) & 0xFFFFFFFF
            seeds, blk = self._block(seeds, fanouts[l], rng, want_global=(l == 0), global_only=(l == 0 and gfb))
            if l > 0 and gfb and blk.indptr.is_cuda:
                # the consumer is TeacherEngine (train_sage): what its backward needs of an inner block -- the block transposed with a self
                # entry per destination, 1 / (in-degree + 1) -- depends on the block alone, so it is built HERE, beside the previous step,
                # instead of by ten launches on the step's own stream (the same two library calls: the same bits)
                from . import ops
                blk.t_indptr, blk.t_indices = ops.csr_transpose(blk.indptr, blk.indices, blk.n_dst, blk.n_src, blk.num_edges(), add_self=True)
                blk.inv_deg, _ = ops.degrees(blk.indptr, None, blk.n_dst, blk.n_src, 0, want_out=False, transform=ops.DEG_INV_PLUS1)
            blocks.insert(0, blk)
        return seeds, output_nodes, blocks

    def __iter__(self):
        """Batches are built AHEAD on a side HIP stream (by a worker thread, up to `depth` of them; `threaded = False`: by the consumer's
        thread, one ahead): the sampler's kernels and its small host read-backs (edge / node
        counts of every block) overlap the consumer's work on the current stream -- e.g. TeacherEngine's forward / backward of
        the previous batch -- instead of draining it at every read-back (what the CPU workers of dgl's NodeDataLoader do
        for the reference, here as stream-level concurrency on the GPU)."""
        self._epoch += 1
        gfb = bool(self.global_first_block)      # snapshot: this iterator's batches keep the shape its consumer asked for (see _batch)
        n = self.nids.numel()
        from . import ops
        order = ops.randperm_cpu(n) if self.shuffle else torch.arange(n)
        fanouts = self.sampler.fanouts if isinstance(self.sampler, MultiLayerNeighborSampler) else [None] * self.sampler.n_layers
        chunks = [order[s:s + self.batch_size] for s in range(0, n, self.batch_size)]
        if self.drop_last and chunks and chunks[-1].numel() < self.batch_size:
            chunks.pop()
        if not self.g.indptr.is_cuda or not self.prefetch:
            for b, idx in enumerate(chunks):
                yield self._batch(b, idx, fanouts, global_first=gfb)
            return
        main = torch.cuda.current_stream(self.g.device)
        if self._side is None:
            # ONE side stream per device for every loader: the caching allocator keeps a pool per stream and never returns it, so a
            # stream per loader grew the reserved memory by ~2 GB per loader on the products configuration (scripts/loader_repeat_probe.py)
            key = torch.device(self.g.device).index
            key = torch.cuda.current_device() if key is None else key
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(self.g.device)
            self._side = _SIDE_STREAMS[key]
        side = self._side

        side.wait_stream(main)          # ONCE: the graph / nids may have been produced on the consumer's stream; later builds
                                        # depend on nothing the consumer queues, so they never wait for its kernels

        epoch = self._epoch

        def build(b):
            with torch.cuda.stream(side):
                batch = self._batch(b, chunks[b], fanouts, epoch, gfb)
                ev = torch.cuda.Event()
                ev.record(side)
            return batch, ev

        def hand_over(item):
            (input_nodes, output_nodes, blocks), ev = item
            # the consumer's stream is asked for at EVERY hand-over: a caller that steps inside its own `with torch.cuda.stream(...)`
            # gets the batch ordered on (and kept alive for) the stream it is actually running on, not the one __iter__ started on
            cur = torch.cuda.current_stream(self.g.device)
            cur.wait_event(ev)
            for t in [input_nodes, output_nodes] + [x for blk in blocks for x in (blk.indptr, blk.indices, blk.gindices, blk.dst_nodes,
                                                                                    blk.t_indptr, blk.t_indices, blk.inv_deg)]:
                if t is not None:
                    t.record_stream(cur)                     # allocated on the side stream, consumed on the current one
            return input_nodes, output_nodes, blocks

        if not self.threaded:
            nxt = build(0) if chunks else None
            for b in range(len(chunks)):
                cur_item = nxt
                nxt = build(b + 1) if b + 1 < len(chunks) else None
                yield hand_over(cur_item)
            return
        # A worker thread builds the batches (what dgl's NodeDataLoader workers do for the reference, train_and_eval.py:192-202): every block
        # costs one host read-back (its edge / source counts size the next block), and a consumer that builds batch b + 1 itself sits in
        # three of them, behind sampler kernels that share the GPU with its own step, before it can issue step b + 1 -- the step's
        # stream ran dry for ~0.3 ms of every 2.3 ms products step.  The worker holds up to `depth` finished batches; what a batch
        # contains depends on (seed, epoch, b) only, so the epochs are the same epochs (bit for bit) with and without the thread.
        import queue
        import threading
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        dev_index = torch.device(self.g.device).index
        dev_index = torch.cuda.current_device() if dev_index is None else dev_index

        def worker():
            try:
                torch.cuda.set_device(dev_index)
                for b in range(len(chunks)):
                    item = build(b)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.05)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return
            except BaseException as e:          # handed to the consumer, raised there
                while not stop.is_set():
                    try:
                        q.put(e, timeout=0.05)
                        break
                    except queue.Full:
                        pass

        th = threading.Thread(target=worker, name="glnn-node-loader", daemon=True)
        th.start()
        try:
            for b in range(len(chunks)):
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                yield hand_over(item)
        finally:                                # exhausted, closed early or failed: the worker is let go before the generator is
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(0.02)
