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
    """Contiguous destination-row ranges, one per rank.  `bounds` (world+1 row offsets) defaults to equal row counts;
    `balanced_bounds` cuts by WORK instead (a power-law graph whose node order correlates with degree -- e.g. the
    degree-ordered layout -- would otherwise load the ranks very unevenly).  Every rank's rows live in a SLOT of `rpr` rows
    (the longest range, rounded up to a multiple of `chunks`) of the padded activation buffers, so that the per-layer
    exchange is one equal-sized all-gather whatever the ranges are:
        layout "nat"  natural node ids (the replicated input features)
        layout "own"  rank-major slots      position(v) = rank(v) * rpr + (v - lo[rank(v)])      (== v for equal ranges)
        layout "cm"   chunk-major slots     [chunk][rank][cr]: what the chunked, overlapped exchange produces
    Column indices are relabelled ONCE per layout (ShardedTeacher._cols); activations are never re-packed."""

    CHUNK_QUANTUM = 128      # chunk sizes are whole 32-row tiles of the fused kernel AND whole 128-row workgroups of the stand-alone aggregation

    def __init__(self, n, world, rank, chunks=1, bounds=None, chunk_sizes=None, kinds=None):
        """chunk_sizes (round 6): rows per chunk of a slot, UNEQUAL chunks allowed (sum >= the longest range) -- the chunk-major layout is
        then [chunk k][rank][chunk_sizes[k]]; `kinds` labels the chunks (the mixed layer-1 exchange: "W" = exchanged as the layer's wide
        output, "N" = as its narrow aggregate).  Default: `chunks` equal chunks."""
        self.n, self.world, self.rank, self.chunks = int(n), int(world), int(rank), max(1, int(chunks))
        if bounds is None:
            per = (self.n + self.world - 1) // self.world
            bounds = [min(self.n, r * per) for r in range(self.world + 1)]
        self.bounds = [int(b) for b in bounds]
        if len(self.bounds) != self.world + 1 or self.bounds[0] != 0 or self.bounds[-1] != self.n \
                or any(a > b for a, b in zip(self.bounds[:-1], self.bounds[1:])):
            raise ValueError(f"RowShards: bounds must be {self.world + 1} non-decreasing offsets from 0 to n")
        longest = max(1, max(b - a for a, b in zip(self.bounds[:-1], self.bounds[1:])))
        if chunk_sizes is None:
            self.cr = (longest + self.chunks - 1) // self.chunks      # rows per chunk
            if self.chunks > 1:                                       # ... a whole number of the fused kernel's 32-row tiles, so that ONE launch
                q = self.CHUNK_QUANTUM                                # can cover all chunks of a rank (no tile straddles two chunks)
                self.cr = (self.cr + q - 1) // q * q
            self.csize = [self.cr] * self.chunks
        else:
            self.csize = [int(c) for c in chunk_sizes]
            self.chunks = len(self.csize)
            if self.chunks < 1 or min(self.csize) < 1 or sum(self.csize) < longest:
                raise ValueError("RowShards: chunk_sizes must be positive and cover the longest range")
            self.cr = max(self.csize)
        self.kinds = list(kinds) if kinds is not None else [None] * self.chunks
        self.coff = [0]
        for c in self.csize:
            self.coff.append(self.coff[-1] + c)                       # chunk k holds own-range offsets [coff[k], coff[k + 1])
        self.rpr = self.coff[-1]                                      # rows per slot
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.rows = self.hi - self.lo
        self.slot = self.rank * self.rpr
        self.n_pad = self.rpr * self.world
        self.uniform = all(b == min(self.n, r * self.rpr) for r, b in enumerate(self.bounds))   # "own" == "nat"

    @staticmethod
    def balanced_bounds(indptr, world, row_cost=2.0):
        """Row offsets that split cost(v) = in_degree(v) + row_cost evenly: per edge the aggregation moves one gathered
        row, per destination row about two more (self row + output row), hence row_cost = 2."""
        n = indptr.numel() - 1
        cost = indptr.to(torch.float64) + row_cost * torch.arange(n + 1, dtype=torch.float64, device=indptr.device)
        targets = cost[-1] * torch.arange(1, world, dtype=torch.float64, device=indptr.device) / world
        cuts = torch.searchsorted(cost, targets).clamp_(0, n).tolist()
        return [0] + [int(c) for c in cuts] + [n]

    def chunk_rows(self, c):
        """(offset inside the own range, row count) of chunk c of this rank."""
        off = self.coff[c]
        return off, max(0, min(self.rows - off, self.csize[c]))

    def chunk_block(self, c):
        """(first row, rows) of chunk c's block [world][csize[c]] in a chunk-major buffer."""
        return self.world * self.coff[c], self.world * self.csize[c]

    def chunk_slot(self, c, rank=None):
        """First row of `rank`'s (default: this rank's) slot inside chunk c's block of a chunk-major buffer."""
        return self.world * self.coff[c] + (self.rank if rank is None else rank) * self.csize[c]

    def mixed(self, fraction, quantum=64, by_chunk=True):
        """The shards of the MIXED layer-1 exchange.  by_chunk (default): the first round(fraction * chunks) chunks are "W" (exchanged as the
        layer's wide output), the rest "N" (narrow aggregate + replicated projection) -- the layout and the chunk count stay what they are,
        so the layers behind keep their launches (cutting every chunk in two doubled the launches of layers 2 and 3 and cost more than the
        mixed layer 1 saved: profiles/scale_model_r06_c2_mixed_twice_the_chunks.json).  by_chunk=False: every chunk is cut into a leading "W"
        part of ~fraction of its rows (a multiple of `quantum` rows) and an "N" part (unequal chunks, twice as many)."""
        if by_chunk:
            nw = min(self.chunks, max(0, int(round(fraction * self.chunks))))
            return RowShards(self.n, self.world, self.rank, bounds=self.bounds, chunk_sizes=list(self.csize),
                             kinds=["W"] * nw + ["N"] * (self.chunks - nw))
        sizes, kinds = [], []
        for c in self.csize:
            fw = min(c, max(0, int(round(fraction * c / quantum)) * quantum))
            for sz, kd in ((fw, "W"), (c - fw, "N")):
                if sz > 0:
                    sizes.append(sz)
                    kinds.append(kd)
        return RowShards(self.n, self.world, self.rank, bounds=self.bounds, chunk_sizes=sizes, kinds=kinds)

    def position(self, v, layout):
        """Row of node id(s) v (int64 tensor) in a buffer of the given layout."""
        if layout == "nat" or (layout == "own" and self.uniform):
            return v
        b = torch.tensor(self.bounds, dtype=torch.int64, device=v.device)
        r = torch.searchsorted(b[1:], v, right=True)
        i = v - b[r]
        if layout == "own":
            return r * self.rpr + i
        off = torch.tensor(self.coff, dtype=torch.int64, device=v.device)
        size = torch.tensor(self.csize, dtype=torch.int64, device=v.device)
        k = torch.searchsorted(off[1:], i, right=True)
        return self.world * off[k] + r * size[k] + (i - off[k])

    def cm_position(self, v):
        return self.position(v, "cm")


ONE_LAUNCH = __import__("os").environ.get("GLNN_ONE_LAUNCH", "1") != "0"      # the chunks of the fused layer as tile ranges of ONE launch with completion signals (ShardedTeacher)
CHUNK_STREAMS = int(__import__("os").environ.get("GLNN_CHUNK_STREAMS", "1"))      # producer chunk launches alternate between this many streams (ShardedTeacher)
EXCHANGE_STATS = {"collectives": 0, "floats_received": 0}      # per process; tests and bench read / reset it
FORCE_COLLECTIVES = False      # tests: issue the collectives even for world == 1 (a 1-rank RCCL group exercises the transport calls)


def _count(out_block):
    EXCHANGE_STATS["collectives"] += 1
    EXCHANGE_STATS["floats_received"] += out_block.numel()


class EmulatedPeers:
    """In-process stand-in for the process group of a `world`-rank job: THIS process plays rank `rank`, the peers do not run.
    Pass it as `group=` to ShardedTeacher / HaloShardedTeacher / HaloPlan (with RowShards(n, world, rank, ...)): every collective
    of this module then writes -- with local device copies -- the same number of bytes into the same places the real transport
    would, so that one GPU can time rank r's kernels of the N-rank forward (bench.py --emulate N, --workload xl).  Two sources:
      truth[tag]  the full [n, d] activation a boundary exchanges, natural node order, taken from an UNSHARDED forward
                  (`record_truth`): the peers' rows are the real ones and the emulated rank's output equals the unsharded rows;
      no truth    the peers' slots are filled with copies of this rank's own slab (right volume, made-up values): the
                  synthetic-XL shard, whose unsharded forward no single GPU can hold.
    Tags: ("agg", l) the aggregate of widening layer l, ("y", l) the output of layer l, ("hw", l) the projected rows of
    narrowing layer l.  `full_graph` (all rows, global ids) lets HaloPlan derive what the peers would request from this rank.
    `events` collects (tag, bytes, start, end) device events of the fills when the tensors live on a GPU."""

    def __init__(self, world, rank, truth=None, full_graph=None):
        self.world, self.rank = int(world), int(rank)
        self.truth = truth or {}
        self.full_graph = full_graph
        self.events = None          # set to a list to time the fills

    def _timed(self, tag, nbytes, like):
        if self.events is None or not like.is_cuda:
            return None
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        return (tag, nbytes, s)

    def _done(self, rec):
        if rec is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.append(rec + (e,))

    def fill_slots(self, base, mine, shards, tag, rows_per_slot, row_offset):
        """The all-gather of one [world * rows_per_slot, ld] block `base` whose slot `rank` (= `mine`) is filled: slot p receives
        rank p's rows [row_offset, row_offset + rows_per_slot) of its own range (clipped to the range)."""
        t = self.truth.get(tag)
        rec = self._timed(tag, (shards.world - 1) * mine.numel() * 4, base)
        my_nr = max(0, min(shards.rows - row_offset, rows_per_slot))         # valid rows of `mine` (the rest of the slot is padding)
        for p in range(shards.world):
            if p == shards.rank:
                continue
            dst = base[p * rows_per_slot:(p + 1) * rows_per_slot]
            lo = shards.bounds[p] + row_offset
            nr = max(0, min(shards.bounds[p + 1] - lo, rows_per_slot))
            if nr == 0:
                continue
            if t is not None:
                dst[:nr, :t.shape[1]].copy_(t[lo:lo + nr])
            elif my_nr == 0:
                dst[:nr].zero_()
            elif nr <= my_nr:
                dst[:nr].copy_(mine[:nr])
            else:                      # the peer's range is longer than mine: repeat my rows
                dst[:nr].copy_(mine[torch.arange(nr, device=mine.device) % my_nr])
        self._done(rec)

    def peer_requests(self, shards):
        """Per peer p: the sorted node ids inside THIS rank's range that p's rows reference (what p's HaloPlan would ask for)."""
        g = self.full_graph
        if g is None:
            raise RuntimeError("EmulatedPeers: the halo plan of an emulated rank needs full_graph (all rows, global ids)")
        out = []
        for p in range(shards.world):
            if p == shards.rank:
                out.append(torch.empty(0, dtype=torch.int64, device=g.indices.device))
                continue
            e0, e1 = int(g.indptr[shards.bounds[p]]), int(g.indptr[shards.bounds[p + 1]])
            u = torch.unique(g.indices[e0:e1].long())
            out.append(u[(u >= shards.lo) & (u < shards.hi)])
        return out

    def fill_halo(self, out, send, tag, ids):
        """The halo all-to-all: `out` [n_halo, w] receives the rows of node ids `ids` (ascending = grouped by owner)."""
        t = self.truth.get(tag)
        rec = self._timed(tag, out.numel() * 4, out)
        if out.shape[0]:
            if t is not None:
                out[:, :t.shape[1]].copy_(t[ids])
            elif send.shape[0]:
                out.copy_(send[torch.arange(out.shape[0], device=out.device) % send.shape[0]])
            else:
                out.zero_()
        self._done(rec)


def _emu(group):
    return group if isinstance(group, EmulatedPeers) else None


def record_truth(encoder, graph, x, be):
    """The unsharded layer-wise forward of `encoder` (SAGE, eval) over the whole `graph`, keeping what the sharded forms put on
    the wire: {("agg", l) | ("hw", l) | ("y", l): [n, d]} for EmulatedPeers(truth=...), and the logits.  Stand-alone aggregation +
    GEMM per layer (no fused kernel), project-first when the layer narrows -- the same arithmetic the sharded layers use."""
    truth = {}
    L = encoder.num_layers
    n = graph.n_dst
    h = be.as_feat(x)
    for l, layer in enumerate(encoder.layers):
        w = layer.fc_neigh.weight
        es, eh, relu = encoder._tail(l)
        d_out, d_in = w.shape
        if d_in > d_out:
            hw = be.gemm(h, w)
            truth[("hw", l)] = hw
            h = be.spmm(graph.indptr, graph.indices, hw, n, be.AGG_SAGE_GCN, ep_scale=es, ep_shift=eh, relu=relu)
        else:
            agg = be.spmm(graph.indptr, graph.indices, h, n, be.AGG_SAGE_GCN)
            if l < L - 1 and 2 * d_in <= d_out:
                truth[("agg", l)] = agg
            h = be.gemm(agg, w, ep_scale=es, ep_shift=eh, relu=relu)
        if l < L - 1:
            truth[("y", l)] = h
    return truth, h


# The bench's fail-safe ladder (benchlib/products.py): SAFE_LIST_FORM = True makes every RCCL all-gather take the out-of-place list
# form (what the gloo tests run) instead of the in-place all_gather_into_tensor on a slab of the receive buffer; INJECT_FAIL[kind] = n
# makes the next n collectives of that kind raise (tests only: "async" = the chunked overlapped all-gathers, "sync" = all_gather_rows,
# "grad_overlap" / "grad" = the student's gradient all-reduces, "probe" = the link probe, "signal" = the wait for a chunk's completion signal).
SAFE_LIST_FORM = False
INJECT_FAIL = {}


def _inject(kind):
    n = INJECT_FAIL.get(kind, 0)
    if n > 0:
        INJECT_FAIL[kind] = n - 1
        raise RuntimeError(f"injected failure of a '{kind}' collective (dist.INJECT_FAIL)")


def _storage_rows(buf):
    """The contiguous [rows, ld] tensor behind a feature view [rows, d] (ld = row stride >= d)."""
    if buf.is_contiguous():
        return buf
    return torch.as_strided(buf, (buf.shape[0], buf.stride(0)), (buf.stride(0), 1))


def all_gather_rows(buf, shards, group=None, tag=None):
    """In-place all-gather of the [n_pad, d] feature buffer `buf` (row stride ld) whose slot [rank*rpr, (rank+1)*rpr) this
    rank has filled.  The collective runs on the contiguous padded storage, whole rows including the [d, ld) padding
    columns, so that every rank's slab is ONE contiguous block.  RCCL: the in-place form (send buffer = this rank's slab of
    the receive buffer, ncclAllGather's documented sendbuff == recvbuff + rank * sendcount case) -- no staging copy."""
    if shards.world == 1 and not FORCE_COLLECTIVES and not _emu(group):
        return buf
    base = _storage_rows(buf)
    mine = base[shards.slot:shards.slot + shards.rpr]
    _count(base)
    if _emu(group):
        group.fill_slots(base, mine, shards, tag, shards.rpr, 0)
        return buf
    _inject("sync")
    if dist.get_backend(group) == "nccl" and not SAFE_LIST_FORM:
        dist.all_gather_into_tensor(base, mine, group=group)
    else:   # gloo (CPU tests): list form
        tmp = [torch.empty_like(mine) for _ in range(shards.world)]
        dist.all_gather(tmp, mine.contiguous(), group=group)
        for r, t in enumerate(tmp):
            base[r * shards.rpr:(r + 1) * shards.rpr].copy_(t)
    return buf


def _all_gather_block(out_block, mine, shards, group, tag=None, chunk=0):
    """Asynchronous all-gather of one contiguous [world*cr, ld] block from this rank's [cr, ld] slot inside it (in place).
    Returns a callable that makes the current stream wait for it."""
    _count(out_block)
    cs = shards.csize[chunk]
    if _emu(group):
        group.fill_slots(out_block, mine, shards, tag, cs, shards.coff[chunk])
        return lambda: None
    _inject("async")
    if dist.get_backend(group) == "nccl" and not SAFE_LIST_FORM:
        work = dist.all_gather_into_tensor(out_block, mine, group=group, async_op=True)
        return work.wait
    tmp = [torch.empty_like(mine) for _ in range(shards.world)]
    work = dist.all_gather(tmp, mine.contiguous(), group=group, async_op=True)

    def finish():
        work.wait()
        for r, t in enumerate(tmp):
            out_block[r * cs:(r + 1) * cs].copy_(t)
    return finish


def probe_link(world, rank, floats_per_rank, device, group=None, reps=3):
    """Measure what an in-place all-gather of `floats_per_rank` floats per rank achieves on THIS job's transport (RCCL over xGMI on
    the GPU box; gloo in the CPU / single-device tests): the figure the exchange forms of ShardedTeacher are priced with.  Every rank
    receives world - 1 slabs, one from each peer (on MI355X: one xGMI link per peer, all links in parallel), so
        per_link_GBps = slab bytes / time,     received_GBps = (world - 1) * slab bytes / time.
    Collective; returns the same dict on every rank (times are the maximum over ranks)."""
    import time as _time
    m = max(1, int(floats_per_rank))
    out = torch.empty(world * m, device=device)
    mine = out[rank * m:(rank + 1) * m]
    mine.fill_(float(rank))
    nccl = dist.get_backend(group) == "nccl"

    nccl = nccl and not SAFE_LIST_FORM
    _inject("probe")

    def once():
        if nccl:
            dist.all_gather_into_tensor(out, mine, group=group)
        else:
            tmp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(tmp, mine.contiguous(), group=group)
            for r, t in enumerate(tmp):
                out[r * m:(r + 1) * m].copy_(t)

    def sync():
        if out.is_cuda:
            torch.cuda.synchronize(device)

    once()                                       # communicator set-up, buffer registration
    sync()
    dist.barrier(group=group)
    times = []
    for _ in range(reps):
        sync()
        t0 = _time.perf_counter()
        once()
        sync()
        times.append(_time.perf_counter() - t0)
    ok = bool((out.view(world, m)[:, 0] == torch.arange(world, device=device, dtype=out.dtype)).all())
    t = torch.tensor([min(times), 0.0 if ok else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    sec = float(t[0].item())
    slab = 4.0 * m
    return {"slab_MB": slab / 1e6, "seconds": sec, "per_link_GBps": slab / sec / 1e9, "received_GBps": (world - 1) * slab / sec / 1e9,
            "correct": float(t[1].item()) == 0.0, "reps": reps, "backend": dist.get_backend(group)}


class ShardedTeacher:
    """SAGE layer-wise inference over a row-sharded graph.  `graph_shard` = full graph's rows [lo,hi)
    (glnn_amd.graph.CSRGraph.row_range), column indices global.

    With shards.chunks > 1 the exchange of a widening layer is CHUNKED AND OVERLAPPED: the own rows are aggregated
    chunk by chunk, each chunk's all-gather is issued asynchronously as soon as its aggregation is queued (it runs on
    the collective's stream while the next chunk aggregates), and the replicated projection consumes the chunks in
    arrival order.  The gathered activations then live in a chunk-major row order ([chunk][rank][rows]); the next
    layer reads them through a column-index array relabelled once at construction -- no data is ever re-packed."""

    def __init__(self, encoder, graph_shard, shards, be, group=None, widening_exchange="narrow", mixed_fraction=0.5, mixed_quantum=0):
        """widening_exchange: what a widening layer (2*d_in <= d_out: products layer 1, 100 -> 256) puts on the wire --
        "narrow": its d_in-wide aggregate, every rank then projects ALL rows itself (least bytes, replicated GEMM);
        "wide":   its d_out-wide output of the fused aggregate+project kernel on the own rows only (no replicated work,
                  d_out/d_in times the bytes);
        "mixed":  (round 6) a fraction `mixed_fraction` of every chunk's rows travels wide, the rest narrow -- the continuous form between
                  the two: `shards` is replaced by shards.mixed(mixed_fraction) (whole chunks of either kind).
        All are chunked and overlapped when shards.chunks > 1; results are identical."""
        if widening_exchange not in ("narrow", "wide", "mixed"):
            raise ValueError("ShardedTeacher: widening_exchange must be 'narrow', 'wide' or 'mixed'")
        if widening_exchange == "mixed":
            shards = shards.mixed(mixed_fraction, mixed_quantum or 64, by_chunk=not mixed_quantum)      # (mixed_quantum > 0: the cut-every-chunk form)
        self.enc, self.g, self.sh, self.be, self.group = encoder, graph_shard, shards, be, group
        if graph_shard.n_dst != shards.rows:
            raise ValueError(f"ShardedTeacher: the graph shard has {graph_shard.n_dst} rows, the shard range {shards.rows}")
        self.widening_exchange, self.mixed_fraction = widening_exchange, mixed_fraction
        self._bufs = {}
        self._col_cache = {}
        self.chunk_ready = None      # set to a list: (chunk, launch-start event, ready event) of every one-launch layer ((-1, ..) = the launch's end)

    # buffers a later layer GATHERS from, and the row order their columns are addressed in (the others feed a GEMM or are outputs)
    _GATHERED = {"ycm": "cm", "hwcm": "cm", "y": "own", "hw": "own"}

    def _full_buffer(self, key, d, device):
        k = (key, d)
        if k not in self._bufs:
            layout = self._GATHERED.get(key[0]) if isinstance(key, tuple) else None
            if layout is not None and hasattr(self.be, "placed_for_gather") and self.sh.rows > 0:
                # placed: the allocation in which a gather over this shard's edges runs fastest (ops.placed_for_gather, HIP backend only)
                self._bufs[k] = self.be.placed_for_gather(self.sh.n_pad, d, device, self.g.indptr, self._cols(layout), self.sh.rows,
                                                          what=f"ShardedTeacher {key[0]}{key[1]}", zero=True)
            else:
                self._bufs[k] = self.be.feat_empty(self.sh.n_pad, d, device, zero=True)
        return self._bufs[k]

    def _tile_order(self, off, nr):
        """Heaviest-tile-first order of the fused launch over own rows [off, off + nr) (cached; None without a HIP backend): a shard's
        chunk launch is short, and the graph's hub rows should start first, not wherever their ids put them."""
        if not hasattr(self.be, "fused_tile_order") or nr < 4096:
            return None
        k = ("order", off, nr)
        if k not in self._col_cache:
            self._col_cache[k] = self.be.fused_tile_order(self.g.indptr[off:off + nr + 1], nr)
        return self._col_cache[k]

    def _hub(self, off, nr):
        """The hub plan of a launch over own rows [off, off + nr) (cached; None without hub rows / without a HIP backend): a shard keeps the
        graph's hub rows, and a chunk launch is short -- their segments are gathered by one workgroup each (ops.HubPlan)."""
        if not hasattr(self.be, "hub_plan"):
            return None
        k = ("hub", off, nr)
        if k not in self._col_cache:
            self._col_cache[k] = self.be.hub_plan(self.g.indptr[off:off + nr + 1], nr)
        return self._col_cache[k]

    def _kw(self, off, nr, fused=False, d_in=None):
        """Launch extras of an aggregation over own rows [off, off + nr): hub plan (+ tile order for the fused kernel), HIP backend only.
        The plan is used where it was measured to pay on the products shard (profiles/scale_model_r05*.json): every stand-alone
        aggregation, and the fused kernel at d_in > 128 -- at 100 wide its heaviest-first tile order already hides the hub rows behind a
        chunk launch, and the extra launch costs ~30 us per chunk."""
        kw = {}
        if not fused or (d_in is not None and d_in > 128):
            hub = self._hub(off, nr)
            if hub is not None:
                kw["hub"] = hub
        if fused:
            kw["tile_order"] = self._tile_order(off, nr)
        return kw

    def _cols(self, layout):
        """Column indices of the shard's edges for a source matrix in `layout` (one-time relabelling, cached)."""
        if layout == "nat" or (layout == "own" and self.sh.uniform):
            return self.g.indices
        if layout not in self._col_cache:
            self._col_cache[layout] = self.sh.position(self.g.indices.long(), layout).to(torch.int32)
        return self._col_cache[layout]

    def _pieces(self, layout):
        """Own rows as (offset in own range, rows, slice into a buffer of `layout`) pieces."""
        sh = self.sh
        if layout == "nat":
            return [(0, sh.rows, slice(sh.lo, sh.hi))]
        if layout == "own":
            return [(0, sh.rows, slice(sh.slot, sh.slot + sh.rows))]
        out = []
        for c in range(sh.chunks):
            off, nr = sh.chunk_rows(c)
            if nr > 0:
                p0 = sh.chunk_slot(c)
                out.append((off, nr, slice(p0, p0 + nr)))
        return out

    def _chunk_self(self, x, layout, c):
        """The own rows of chunk c inside `x` (their self rows for the SAGE-gcn aggregator)."""
        sh = self.sh
        off, nr = sh.chunk_rows(c)
        if layout == "cm":
            p0 = sh.chunk_slot(c)
            return x[p0:p0 + nr]
        base = sh.lo if layout == "nat" else sh.slot
        return x[base + off:base + off + nr]

    def _own(self, key, d, device):
        """This rank's slot of the "own"-layout buffer `key` (where a layer writes its output rows)."""
        return self._full_buffer(key, d, device)[self.sh.slot:self.sh.slot + self.sh.rows]

    def _aggregate_project(self, x, layout, w, tail, out_own):
        """Aggregate-first layer on the own rows (piecewise when x is chunk-major)."""
        be, g = self.be, self.g
        ep_scale, ep_shift, relu = tail
        d_in, d_out = w.shape[1], w.shape[0]
        idx = self._cols(layout)
        for off, nr, sl in self._pieces(layout):
            ip = g.indptr[off:off + nr + 1]            # absolute offsets into the one indices array
            if hasattr(be, "sage_fused") and d_in <= 256 and d_out <= 256:
                be.sage_fused(ip, idx, x, nr, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out_own[off:off + nr], x_self=x[sl],
                              **self._kw(off, nr, fused=True, d_in=d_in))
            else:
                agg = be.spmm(ip, idx, x, nr, be.AGG_SAGE_GCN, x_self=x[sl], **self._kw(off, nr))
                be.gemm(agg, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out_own[off:off + nr])

    # Producer chunks on ALTERNATING streams (round 6, GLNN_CHUNK_STREAMS=2): a chunk launch is short (a rank's rows / chunks), and launched
    # back to back on one stream every one pays its ramp and its tail (N = 8: two launches of the fused layer 2 take 2.56 ms where one
    # over the same rows would take ~2.3; four take 2.74).  On two streams the next chunk's workgroups fill the CUs the previous chunk's
    # tail leaves idle; each chunk's all-gather is issued on ITS stream (the collective orders itself behind that stream only).
    def _chunk_stream(self, c, device):
        import contextlib
        n = CHUNK_STREAMS
        if n <= 1 or device.type != "cuda":
            return contextlib.nullcontext()
        pool = self.__dict__.setdefault("_streams", [torch.cuda.Stream(device=device) for _ in range(n)])
        st = pool[c % n]
        if c < n:
            st.wait_stream(torch.cuda.current_stream(device))          # the chunk's inputs were produced on the caller's stream
        return torch.cuda.stream(st)

    def _chunk_join(self, device):
        if CHUNK_STREAMS > 1 and device.type == "cuda" and "_streams" in self.__dict__:
            cur = torch.cuda.current_stream(device)
            for st in self._streams:
                cur.wait_stream(st)

    # ---- the chunks of a fused layer as tile ranges of ONE launch, each with a completion signal (round 6, ONE_LAUNCH) ----------------------
    def _one_launch_ok(self, x, c0, c1, d_in, d_mid, d_out2=0):
        be, sh = self.be, self.sh
        return (ONE_LAUNCH and x.is_cuda and hasattr(be, "ChunkSignals") and hasattr(be, "sage_fused") and 1 <= c1 - c0 <= be.ChunkSignals.MAX
                and d_in <= 256 and d_mid <= 256 and d_out2 <= 256 and all(o % 32 == 0 for o in sh.coff[:-1]) and sh.chunk_rows(c0)[1] > 0)

    def _launch_chunks(self, c0, c1, x, layout, w, tail, out=None, w_next=None, out_next=None, agg_out=None):
        """ONE launch over the own rows of chunks [c0, c1): self rows read from `x` (in `layout`), outputs written to the chunks' slots of
        the chunk-major buffers -- the fused aggregate + project kernel into `out` / `out_next` (tile order = the chunks' heaviest-first
        orders one after the other; completion follows it), or, with `agg_out`, the stand-alone aggregation into that buffer (w, tail
        unused).  Returns (signals, exchange stream, launch stream): `_after_signal` issues a chunk's exchange behind its signal."""
        be, g, sh, dev = self.be, self.g, self.sh, x.device
        off0 = sh.coff[c0]
        nr = max(0, min(sh.rows, sh.coff[c1]) - off0)
        key = ("signals", c0, c1, agg_out is not None)
        if key not in self._col_cache and agg_out is not None:
            self._col_cache[key] = (be.ChunkSignals([sh.coff[c] - off0 for c in range(c0, c1 + 1)], dev), None)
        if key not in self._col_cache:
            parts = []
            for c in range(c0, c1):
                off, n_c = sh.chunk_rows(c)
                if n_c > 0:
                    o = self._tile_order(off, n_c)
                    o = torch.arange((n_c + 31) // 32, dtype=torch.int32, device=dev) if o is None else o
                    parts.append(o + (off - off0) // 32)
            self._col_cache[key] = (be.ChunkSignals([sh.coff[c] - off0 for c in range(c0, c1 + 1)], dev), torch.cat(parts).to(torch.int32).contiguous())
        sig, order = self._col_cache[key]
        base = {"cm": None, "nat": sh.lo, "own": sh.slot}[layout]
        self_rows = [sh.chunk_slot(c) if base is None else base + sh.coff[c] for c in range(c0, c1)]
        desc = sig.launch(self_rows, [sh.chunk_slot(c) for c in range(c0, c1)], nr)
        cur = torch.cuda.current_stream(dev)
        side = self.__dict__.setdefault("_exchange_stream", torch.cuda.Stream(device=dev))
        side.wait_stream(cur)
        kw = {} if agg_out is not None else {"tile_order": order}
        if agg_out is not None or w.shape[1] > 128:        # (the hub plan where it was measured to pay: see _kw)
            hub = self._hub(off0, nr)
            if hub is not None:
                kw["hub"] = hub
        if self.chunk_ready is not None:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(cur)
        if agg_out is not None:
            be.spmm(g.indptr[off0:off0 + nr + 1], self._cols(layout), x, nr, be.AGG_SAGE_GCN, out=agg_out, x_self=x, chunks=desc, **kw)
        else:
            es, eh, rl = tail
            be.sage_fused(g.indptr[off0:off0 + nr + 1], self._cols(layout), x, nr, w, ep_scale=es, ep_shift=eh, relu=rl, x_self=x, out=out,
                          w_next=w_next, out_next=out_next, want_out=out is not None, chunks=desc, **kw)
        if self.chunk_ready is not None:
            self._t1 = torch.cuda.Event(enable_timing=True)
            self._t1.record(cur)
        return sig, side, cur

    def _after_signal(self, sig, k, side, issue):
        """Issue chunk k's exchange (`issue()` -> waiter) behind the chunk's completion signal: the exchange stream is held by the signal, the
        collective is enqueued from it.  (Emulated peers: the fills stay on the launch's own stream -- the model's kernel times are taken
        without them -- but the signal is still waited for, and `chunk_ready` records when it fired.)"""
        _inject("signal")
        if not sig.empty(k):
            sig.wait(side, k)
        if self.chunk_ready is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(side)
            self.chunk_ready.append((k, self._t0, e))
        if _emu(self.group) is not None:
            return issue()
        with torch.cuda.stream(side):
            return issue()

    def _launch_done(self, side, cur):
        if self.chunk_ready is not None:
            self.chunk_ready.append((-1, self._t0, self._t1))      # (the launch's own end)
        cur.wait_stream(side)

    def _widening_layer_overlapped(self, l, x, layout, w, tail):
        """2*d_in <= d_out, world > 1: per chunk, by its kind (RowShards.kinds):
             "N" / None   aggregate the own rows -> async all-gather of the d_in-wide aggregate -> REPLICATED GEMM of every rank's rows
                          (the "narrow" exchange: least bytes, every rank multiplies all rows)
             "W"          fused aggregate + project of the own rows -> async all-gather of the d_out-wide output (the "wide" exchange:
                          no replicated work, d_out / d_in times the bytes)
           Every chunk's all-gather is issued as soon as its producer is queued; the replicated GEMMs consume the narrow chunks in
           arrival order.  A MIXED exchange (round 6) is chunks of both kinds: the fraction of the rows that travels wide is the dial
           between link time and replicated MFMA work.  Returns the chunk-major gathered output (every rank's rows)."""
        be, g, sh = self.be, self.g, self.sh
        ep_scale, ep_shift, relu = tail
        d_in, d_out = w.shape[1], w.shape[0]
        kinds = [k or ("W" if self.widening_exchange == "wide" else "N") for k in sh.kinds]
        y = self._full_buffer(("ycm", l), d_out, x.device)       # chunk-major [chunk][rank][rows]
        ybase = _storage_rows(y)
        agg = abase = None
        if "N" in kinds:
            agg = self._full_buffer(("agg", l), d_in, x.device)
            abase = _storage_rows(agg)
        idx = self._cols(layout)
        works = []
        nw = 0                                                  # leading "W" chunks: ONE launch with a completion signal per chunk
        while nw < sh.chunks and kinds[nw] == "W":
            nw += 1
        if nw and self._one_launch_ok(x, 0, nw, d_in, d_out):
            sig, side, cur = self._launch_chunks(0, nw, x, layout, w, tail, out=y)
            for c in range(nw):
                b0, bn = sh.chunk_block(c)
                p0, cs = sh.chunk_slot(c), sh.csize[c]
                works.append(self._after_signal(sig, c, side, lambda: _all_gather_block(ybase[b0:b0 + bn], ybase[p0:p0 + cs], sh, self.group, ("y", l), c)))
            self._launch_done(side, cur)
        else:
            nw = 0
        n_first = nw                                            # the "N" chunks behind them: ONE launch of the stand-alone aggregation, likewise
        if (nw < sh.chunks and all(k == "N" for k in kinds[nw:]) and x.shape[1] <= 256 and getattr(be, "SELF_ROWS", False)
                and self._one_launch_ok(x, nw, sh.chunks, d_in, d_out)):
            sig, side, cur = self._launch_chunks(nw, sh.chunks, x, layout, w, tail, agg_out=agg)
            for c in range(nw, sh.chunks):
                b0, bn = sh.chunk_block(c)
                p0, cs = sh.chunk_slot(c), sh.csize[c]
                works.append(self._after_signal(sig, c - nw, side, lambda: _all_gather_block(abase[b0:b0 + bn], abase[p0:p0 + cs], sh, self.group,
                                                                                             ("agg", l), c)))
            self._launch_done(side, cur)
            n_first = sh.chunks
        for c in range(n_first, sh.chunks):
          with self._chunk_stream(c, x.device):
              off, nr = sh.chunk_rows(c)
              b0, bn = sh.chunk_block(c)
              p0, cs = sh.chunk_slot(c), sh.csize[c]
              xs = self._chunk_self(x, layout, c)
              ip = g.indptr[off:off + nr + 1]
              if kinds[c] == "N":
                  if nr > 0:
                      be.spmm(ip, idx, x, nr, be.AGG_SAGE_GCN, out=agg[p0:p0 + nr], x_self=xs, **self._kw(off, nr))
                  works.append(_all_gather_block(abase[b0:b0 + bn], abase[p0:p0 + cs], sh, self.group, ("agg", l), c))
              else:
                  if nr > 0:
                      if hasattr(be, "sage_fused") and d_in <= 256 and d_out <= 256:
                          be.sage_fused(ip, idx, x, nr, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=y[p0:p0 + nr], x_self=xs,
                                        **self._kw(off, nr, fused=True, d_in=d_in))
                      else:
                          a_ = be.spmm(ip, idx, x, nr, be.AGG_SAGE_GCN, x_self=xs, **self._kw(off, nr))
                          be.gemm(a_, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=y[p0:p0 + nr])
                  works.append(_all_gather_block(ybase[b0:b0 + bn], ybase[p0:p0 + cs], sh, self.group, ("y", l), c))
        self._chunk_join(x.device)
        for c in range(sh.chunks):
            works[c]()
            if kinds[c] == "N":
                b0, bn = sh.chunk_block(c)
                be.gemm(agg[b0:b0 + bn], w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=y[b0:b0 + bn])
        return y

    def _plain_then_narrow_overlapped(self, l, x, layout):
        """Aggregate-first layer l followed by a narrowing layer l+1, chunks > 1: the own rows of layer l are produced
        chunk by chunk, each chunk is projected with W_{l+1} at once (layer l+1 projects first) and the all-gather of
        its narrow rows is issued while the next chunk of layer l still aggregates -- the only data layer l+1 needs from
        other ranks.  Then layer l+1 aggregates the own rows from the chunk-major gathered buffer.  Returns its output."""
        enc, sh, be, g = self.enc, self.sh, self.be, self.g
        w1, w2 = enc.layers[l].fc_neigh.weight, enc.layers[l + 1].fc_neigh.weight
        tail1, tail2 = enc._tail(l), enc._tail(l + 1)
        d_mid, d_out = w1.shape[0], w2.shape[0]
        last = l + 1 == enc.num_layers - 1
        y_own = None      # (fallback branch only: the fused launch never writes layer l's rows; own rows, not an n_pad-row buffer)
        hw = self._full_buffer(("hwcm", l + 1), d_out, x.device)          # chunk-major [chunk][rank][rows]
        base = _storage_rows(hw)
        idx = self._cols(layout)
        works = []
        one_launch = self._one_launch_ok(x, 0, sh.chunks, w1.shape[1], d_mid, d_out)
        if one_launch:
            # ONE launch over all chunks (round 6): the chunks are tile ranges of the same launch, each signals its completion, and the
            # exchange stream holds every chunk's all-gather behind the chunk's signal -- the overlap of the chunked form without a short
            # launch's ramp and tail per chunk (N = 8: 2.53 ms in two launches, 2.74 in four, 2.42 in one).
            sig, side, cur = self._launch_chunks(0, sh.chunks, x, layout, w1, tail1, w_next=w2, out_next=hw)
            for c in range(sh.chunks):
                b0, bn = sh.chunk_block(c)
                p0 = sh.chunk_slot(c)
                works.append(self._after_signal(sig, c, side, lambda: _all_gather_block(base[b0:b0 + bn], base[p0:p0 + sh.csize[c]], sh, self.group,
                                                                                        ("hw", l + 1), c)))
            self._launch_done(side, cur)
        for c in (() if one_launch else range(sh.chunks)):
          with self._chunk_stream(c, x.device):
              off, nr = sh.chunk_rows(c)
              b0, bn = sh.chunk_block(c)
              p0 = sh.chunk_slot(c)
              if nr > 0:
                  xs = self._chunk_self(x, layout, c)
                  ip = g.indptr[off:off + nr + 1]
                  es, eh, rl = tail1
                  if hasattr(be, "sage_fused") and w1.shape[1] <= 256 and d_mid <= 256 and d_out <= 256:
                      # aggregate + project + tail + the NEXT layer's projection in one launch: layer l's rows never reach HBM
                      be.sage_fused(ip, idx, x, nr, w1, ep_scale=es, ep_shift=eh, relu=rl, x_self=xs, w_next=w2, out_next=hw[p0:p0 + nr],
                                    want_out=False, **self._kw(off, nr, fused=True, d_in=w1.shape[1]))
                  else:
                      if y_own is None:
                          if ("y_own", l) not in self._bufs:
                              self._bufs[("y_own", l)] = be.feat_empty(sh.rows, d_mid, x.device)
                          y_own = self._bufs[("y_own", l)]
                      agg = be.spmm(ip, idx, x, nr, be.AGG_SAGE_GCN, x_self=xs, **self._kw(off, nr))
                      be.gemm(agg, w1, ep_scale=es, ep_shift=eh, relu=rl, out=y_own[off:off + nr])
                      be.gemm(y_own[off:off + nr], w2, out=hw[p0:p0 + nr])
              works.append(_all_gather_block(base[b0:b0 + bn], base[p0:p0 + sh.csize[c]], sh, self.group, ("hw", l + 1), c))
        self._chunk_join(x.device)
        out = be.feat_empty(sh.rows, d_out, x.device) if last else self._own(("y", l + 1), d_out, x.device)
        for wk in works:
            wk()
        es, eh, rl = tail2
        idx_cm = self._cols("cm")
        if sh.rows > 0 and getattr(be, "SELF_ROWS", False):
            # ONE launch over all own rows (round 6): the output rows are contiguous, only the SELF rows sit in per-chunk slots of the
            # chunk-major buffer -- addressed through self_rows (the global-id form of the aggregation) instead of one launch per chunk
            # (N = 8, 4 chunks: 0.75 -> 0.60 ms per rank; a short launch pays its ramp and its tail once per launch)
            if "own_cm" not in self._col_cache:
                self._col_cache["own_cm"] = sh.position(torch.arange(sh.lo, sh.hi, device=g.indptr.device, dtype=torch.int64), "cm").contiguous()
            be.spmm(g.indptr, idx_cm, hw, sh.rows, be.AGG_SAGE_GCN, ep_scale=es, ep_shift=eh, relu=rl, out=out, x_self=hw,
                    self_rows=self._col_cache["own_cm"], **self._kw(0, sh.rows))
            return out
        for off, nr, sl in self._pieces("cm"):
            be.spmm(g.indptr[off:off + nr + 1], idx_cm, hw, nr, be.AGG_SAGE_GCN, ep_scale=es, ep_shift=eh, relu=rl,
                    out=out[off:off + nr], x_self=hw[sl], **self._kw(off, nr))
        return out

    def forward(self, x_full):
        """x_full: [>= n, F] replicated input features.  Returns this rank's rows of the logits [rows, C]."""
        enc, sh, be, g = self.enc, self.sh, self.be, self.g
        x = be.as_feat(x_full)
        layout = "nat"          # row order of x: natural node ids ("nat"), rank-major slots ("own") or chunk-major ("cm")
        complete = True         # x holds every node's row (False: only this rank's own rows are valid)
        L = enc.num_layers
        dims = [(lay.fc_neigh.weight.shape[1], lay.fc_neigh.weight.shape[0]) for lay in enc.layers]
        y_own = None
        multi = sh.world > 1 or FORCE_COLLECTIVES
        l = 0
        while l < L:
            w = enc.layers[l].fc_neigh.weight
            tail = enc._tail(l)
            ep_scale, ep_shift, relu = tail
            d_in, d_out = dims[l]
            last = l == L - 1
            next_narrow = not last and dims[l + 1][0] > dims[l + 1][1]
            if d_in > d_out:
                # narrowing layer: project own rows, exchange the narrow rows, aggregate own rows (products layer 3:
                # 47 floats per node on the wire instead of 256).  Needs only the OWN rows of x.
                hw = self._full_buffer(("hw", l), d_out, x.device)
                for off, nr, sl in self._pieces(layout):
                    be.gemm(x[sl], w, out=hw[sh.slot + off:sh.slot + off + nr])
                all_gather_rows(hw, sh, self.group, ("hw", l))
                out = be.feat_empty(sh.rows, d_out, x.device) if last else self._own(("y", l), d_out, x.device)
                be.spmm(g.indptr, self._cols("own"), hw, sh.rows, be.AGG_SAGE_GCN, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu,
                        out=out, x_self=hw[sh.slot:sh.slot + sh.rows], **self._kw(0, sh.rows))
            elif not complete:
                raise RuntimeError("ShardedTeacher: internal error, an aggregating layer needs every node's input row")
            elif multi and not last and 2 * d_in <= d_out and self.widening_exchange in ("wide", "mixed") and sh.chunks > 1 and not next_narrow:
                x = self._widening_layer_overlapped(l, x, layout, w, tail)    # exchange the wide output (of all / of the "W" chunks), chunked + overlapped
                layout, y_own = "cm", None
                l += 1
                continue
            elif multi and not last and 2 * d_in <= d_out and self.widening_exchange == "narrow":
                # widening layer (products layer 1: 100 -> 256): exchange the NARROW aggregate and let every rank
                # project all rows itself -- the all-gather moves d_in instead of d_out floats per node (0.98 GB
                # instead of 2.5 GB on products) for the price of a replicated [N, d_in] x [d_in, d_out] GEMM.
                if sh.chunks > 1:
                    x = self._widening_layer_overlapped(l, x, layout, w, tail)      # (every chunk "N")
                    layout = "cm"
                else:
                    agg = self._full_buffer(("agg", l), d_in, x.device)
                    (_, _, sl), = self._pieces(layout)
                    be.spmm(g.indptr, self._cols(layout), x, sh.rows, be.AGG_SAGE_GCN, out=agg[sh.slot:sh.slot + sh.rows], x_self=x[sl],
                            **self._kw(0, sh.rows))
                    all_gather_rows(agg, sh, self.group, ("agg", l))
                    y_full = self._full_buffer(("y", l), d_out, x.device)
                    be.gemm(agg, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=y_full)
                    x, layout = y_full, "own"
                y_own = None          # (a widening layer is never the last one)
                l += 1
                continue
            elif multi and sh.chunks > 1 and next_narrow:
                out = self._plain_then_narrow_overlapped(l, x, layout)      # layers l and l+1
                l += 1
                last = l == L - 1
                next_narrow = False
            else:
                out = be.feat_empty(sh.rows, d_out, x.device) if last else self._own(("y", l), d_out, x.device)
                self._aggregate_project(x, layout, w, tail, out)
            y_own = out
            if not last:
                d_cur = dims[l][1]
                buf = self._full_buffer(("y", l), d_cur, x.device)
                if dims[l + 1][0] > dims[l + 1][1]:
                    x, complete = buf, False          # the next (narrowing) layer projects its own rows only: no exchange
                else:
                    x, complete = all_gather_rows(buf, sh, self.group, ("y", l)), True
                layout = "own"
            l += 1
        return y_own


class HaloPlan:
    """Which remote rows this rank's shard references, and which of its own rows the peers reference: the index lists of
    the HALO exchange SURVEY.md 8(e) describes ("halo all-to-all of only the referenced remote rows, index lists
    precomputed at partition time").  Built once per (graph shard, RowShards): one unique() over the shard's column ids and
    two small all-to-alls (counts, then id lists).  Local activation buffers are laid out [own rows | halo rows], the halo
    rows in ascending node id (= grouped by owner rank, since ranges are contiguous); `cols` are the shard's column indices
    relabelled into that layout."""

    def __init__(self, graph_shard, shards, group=None):
        sh, dev = shards, graph_shard.indices.device
        idx = graph_shard.indices.long()
        ref = torch.unique(idx)                                               # sorted global ids this shard gathers
        remote = ref[(ref < sh.lo) | (ref >= sh.hi)]
        b = torch.tensor(sh.bounds, dtype=torch.int64, device=dev)
        owner = torch.searchsorted(b[1:], remote, right=True)
        self.recv_counts = torch.bincount(owner, minlength=sh.world).tolist()        # rows I receive from each rank
        self.n_halo = int(remote.numel())
        self.remote = remote                                                         # halo row i holds node remote[i]
        if _emu(group):                       # the peers do not run: derive their requests from the full graph
            req = group.peer_requests(sh)
            send_counts = [int(r.numel()) for r in req]
            wanted = torch.cat(req)
        else:
            send_counts = _all_to_all_counts(self.recv_counts, sh, dev, group)
            wanted = _all_to_all_rows(remote.unsqueeze(1).contiguous(), self.recv_counts, send_counts, sh, group).squeeze(1)
        self.send_counts = send_counts                                               # rows each rank wants from me
        if wanted.numel() and (int(wanted.min()) < sh.lo or int(wanted.max()) >= sh.hi):
            raise RuntimeError("HaloPlan: a peer asked for rows this rank does not own")
        self.send_rows = (wanted - sh.lo).contiguous()                               # local row offsets, grouped by requesting rank
        pos = torch.searchsorted(remote, idx)
        own = (idx >= sh.lo) & (idx < sh.hi)
        self.cols = torch.where(own, idx - sh.lo, sh.rows + pos).to(torch.int32)
        self.rows = sh.rows
        self.n_send = int(sum(send_counts))
        self._own_mask, self._indptr = own, graph_shard.indptr
        self._split = None

    def split_csr(self):
        """The shard's edges split by where their source row lives, for the OVERLAPPED exchange (built once, on first use):
          local  [rows x (rows)]          the local-source edges of row v plus one edge v <- v: pass 1 sums them from the own rows
                                          into P[v] = x_self[v] + sum_{u local} x[u] while the halo rows are still in flight;
          remote [rows x (rows + n_halo)] the remote-source edges of row v (columns = rows + halo position, as in `cols`) plus one
                                          edge v <- P[v] (column v): pass 2 sums them from the buffer [P | halo rows];
          inv_deg1                        1 / (in_degree + 1), the row scale of pass 2 (the SAGE-"gcn" mean).
        Returns (local_indptr, local_cols, remote_indptr, remote_cols, inv_deg1)."""
        if self._split is None:
            from .data import csr_from_edges
            ip, own, dev = self._indptr, self._own_mask, self.cols.device
            rows = self.rows
            deg = ip[1:] - ip[:-1]
            dst = torch.repeat_interleave(torch.arange(rows, device=dev), deg)
            me = torch.arange(rows, device=dev)
            cols = self.cols.long()
            loc = csr_from_edges(torch.cat([cols[own], me]), torch.cat([dst[own], me]), rows, rows)
            rem = csr_from_edges(torch.cat([cols[~own], me]), torch.cat([dst[~own], me]), rows, rows + self.n_halo)
            inv = 1.0 / (deg.to(torch.float32) + 1.0)
            self._split = (loc.indptr, loc.indices, rem.indptr, rem.indices, inv)
        return self._split


def _all_to_all_counts(counts, shards, dev, group):
    send = torch.tensor(counts, dtype=torch.int64, device=dev)
    if shards.world == 1:
        return [int(counts[0])]
    if dist.get_backend(group) == "nccl":
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
        return [int(v) for v in recv.tolist()]
    gathered = [torch.empty_like(send) for _ in range(shards.world)]              # gloo has no all-to-all: gather the table
    dist.all_gather(gathered, send, group=group)
    return [int(gathered[r][shards.rank]) for r in range(shards.world)]


def _all_to_all_rows(send, send_counts, recv_counts, shards, group, out=None, tag=None, ids=None):
    """Variable all-to-all of whole rows: `send` [sum(send_counts), w] grouped by destination rank -> [sum(recv_counts), w]
    grouped by source rank.  RCCL: one all_to_all_single with split sizes; gloo (CPU tests): pairwise isend / irecv."""
    n_recv = int(sum(recv_counts))
    if out is None:
        out = torch.empty((n_recv, send.shape[1]), dtype=send.dtype, device=send.device)
    EXCHANGE_STATS["collectives"] += 1
    EXCHANGE_STATS["floats_received"] += (n_recv - int(recv_counts[shards.rank])) * send.shape[1]
    if _emu(group):
        group.fill_halo(out, send, tag, ids)
        return out
    if shards.world == 1 and not FORCE_COLLECTIVES:
        out.copy_(send)
        return out
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(out, send, output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts],
                               group=group)
        return out
    s_off = [0] + list(torch.tensor(send_counts).cumsum(0).tolist())
    r_off = [0] + list(torch.tensor(recv_counts).cumsum(0).tolist())
    reqs, bufs = [], []
    for r in range(shards.world):
        if r == shards.rank:
            out[r_off[r]:r_off[r + 1]].copy_(send[s_off[r]:s_off[r + 1]])
            continue
        if recv_counts[r]:
            buf = torch.empty((int(recv_counts[r]), send.shape[1]), dtype=send.dtype)
            bufs.append((r, buf))
            reqs.append(dist.irecv(buf, src=r, group=group))
        if send_counts[r]:
            reqs.append(dist.isend(send[s_off[r]:s_off[r + 1]].cpu().contiguous(), dst=r, group=group))
    for q in reqs:
        q.wait()
    for r, buf in bufs:
        out[r_off[r]:r_off[r + 1]].copy_(buf)
    return out


def _all_to_all_rows_async(send, send_counts, recv_counts, shards, group, out, tag=None, ids=None):
    """_all_to_all_rows started without waiting for it: returns a callable that completes `out` (RCCL: the collective runs on
    the communicator's stream, the callable makes the current stream wait for it; gloo: the isend / irecv requests are posted
    here and reaped by the callable)."""
    n_recv = int(sum(recv_counts))
    EXCHANGE_STATS["collectives"] += 1
    EXCHANGE_STATS["floats_received"] += (n_recv - int(recv_counts[shards.rank])) * send.shape[1]
    if _emu(group):
        group.fill_halo(out, send, tag, ids)
        return lambda: None
    if shards.world == 1 and not FORCE_COLLECTIVES:
        out.copy_(send)
        return lambda: None
    if dist.get_backend(group) == "nccl":
        work = dist.all_to_all_single(out, send, output_split_sizes=[int(c) for c in recv_counts],
                                      input_split_sizes=[int(c) for c in send_counts], group=group, async_op=True)
        return work.wait
    s_off = [0] + list(torch.tensor(send_counts).cumsum(0).tolist())
    r_off = [0] + list(torch.tensor(recv_counts).cumsum(0).tolist())
    reqs, bufs = [], []
    for r in range(shards.world):
        if r == shards.rank:
            out[r_off[r]:r_off[r + 1]].copy_(send[s_off[r]:s_off[r + 1]])
            continue
        if recv_counts[r]:
            buf = torch.empty((int(recv_counts[r]), send.shape[1]), dtype=send.dtype)
            bufs.append((r, buf))
            reqs.append(dist.irecv(buf, src=r, group=group))
        if send_counts[r]:
            reqs.append(dist.isend(send[s_off[r]:s_off[r + 1]].cpu().contiguous(), dst=r, group=group))

    def finish():
        for q in reqs:
            q.wait()
        for r, buf in bufs:
            out[r_off[r]:r_off[r + 1]].copy_(buf)
    return finish


class HaloShardedTeacher:
    """SAGE layer-wise inference over a row-sharded graph with a HALO exchange: per layer boundary a rank receives only the
    remote rows its own edges reference (all-to-all over HaloPlan's index lists) instead of every row (ShardedTeacher's
    all-gather).  Activations live in local buffers [own rows | halo rows]; memory and wire volume scale with the shard's
    frontier, not with N.  The same narrow-side rules apply: a widening layer exchanges its d_in-wide aggregate and projects
    own + halo rows locally, a narrowing layer projects first and exchanges the d_out-wide rows, a layer in front of a
    narrowing layer exchanges nothing.  On a graph WITHOUT locality (the prescribed random-order generator) the halo of a
    rank is nearly every remote row and this degenerates to the all-gather's volume; it pays on locality-ordered graphs
    (glnn_amd.data.make_clustered_graph: 8 ranks, 90 % intra-community edges -> ~0.5x the all-gather's bytes)."""

    def __init__(self, encoder, graph_shard, shards, be, group=None, overlap=False):
        """overlap=True: every halo exchange is started asynchronously and hidden behind work that needs only the own rows --
        a widening layer projects its own rows while the halo rows of its aggregate travel, and an aggregation whose input is
        being exchanged runs in two passes over the split CSR of HaloPlan.split_csr (local-source edges + self first, then the
        remote-source edges once the halo has landed).  Same sums, local edges before remote ones (results equal to rounding)."""
        self.enc, self.g, self.sh, self.be, self.group, self.overlap = encoder, graph_shard, shards, be, group, overlap
        if graph_shard.n_dst != shards.rows:
            raise ValueError(f"HaloShardedTeacher: the graph shard has {graph_shard.n_dst} rows, the shard range {shards.rows}")
        self.plan = HaloPlan(graph_shard, shards, group)
        self._bufs = {}

    def _local(self, key, d, device):
        k = (key, d)
        if k not in self._bufs:
            self._bufs[k] = self.be.feat_empty(self.plan.rows + self.plan.n_halo, d, device, zero=True)
        return self._bufs[k]

    def _exchange(self, buf, tag=None):
        """Fill the halo rows of the local buffer `buf` (own rows valid) from their owners."""
        pl, be = self.plan, self.be
        base = _storage_rows(buf)
        own = base[:pl.rows]
        send = _storage_rows(be.gather_rows(buf[:pl.rows], pl.send_rows)) if hasattr(be, "gather_rows") else own[pl.send_rows]
        _all_to_all_rows(send.contiguous(), pl.send_counts, pl.recv_counts, self.sh, self.group, out=base[pl.rows:], tag=tag, ids=pl.remote)
        return buf

    def _exchange_async(self, own, dst, tag=None):
        """Start filling dst[rows:] (halo rows) from the owners' rows of `own` ([rows, d] view of a local buffer); returns finish()."""
        pl, be = self.plan, self.be
        base = _storage_rows(dst)
        send = _storage_rows(be.gather_rows(own, pl.send_rows)) if hasattr(be, "gather_rows") else _storage_rows(own)[pl.send_rows]
        return _all_to_all_rows_async(send.contiguous(), pl.send_counts, pl.recv_counts, self.sh, self.group, base[pl.rows:], tag=tag, ids=pl.remote)

    def _aggregate_two_pass(self, key, own, d, finish, tail=(None, None, False), out=None):
        """SAGE-"gcn" mean of the own rows from an input whose halo rows are still in flight into buffer `key2` = [P | halo]:
        pass 1 (local-source edges + self, from `own`) -> P; finish(); pass 2 (remote-source edges + P, row scale 1/(deg+1),
        optional epilogue).  Returns [rows, d]."""
        pl, be = self.plan, self.be
        lip, lcols, rip, rcols, inv = pl.split_csr()
        buf2 = self._local(key, d, own.device)
        es, eh, relu = tail
        be.spmm(lip, lcols, own, pl.rows, be.AGG_SUM, out=buf2[:pl.rows])
        finish()
        return be.spmm(rip, rcols, buf2, pl.rows, be.AGG_SUM, row_scale=inv, ep_scale=es, ep_shift=eh, relu=relu, out=out)

    def _forward_overlapped(self, x_full):
        enc, sh, be, g, pl = self.enc, self.sh, self.be, self.g, self.plan
        x = be.as_feat(x_full)
        L = enc.num_layers
        dims = [(lay.fc_neigh.weight.shape[1], lay.fc_neigh.weight.shape[0]) for lay in enc.layers]
        # state of the layer input: `own` = this rank's rows [rows, d]; either complete (`full` = a buffer every edge can gather
        # from with `cols`), or with its halo rows in flight (`finish` pending, to land in the [P | halo] buffer `key2`)
        full, cols, own = x, g.indices, x[sh.lo:sh.hi]
        pending = None
        out = None
        for l in range(L):
            w = enc.layers[l].fc_neigh.weight
            tail = enc._tail(l)
            ep_scale, ep_shift, relu = tail
            d_in, d_out = dims[l]
            last = l == L - 1
            if d_in > d_out:                       # narrowing: project the own rows, exchange them, aggregate in two passes
                if pending is not None:            # (cannot happen: the layer in front of a narrowing layer exchanges nothing)
                    pending[1]()
                hw = self._local(("hw", l), d_out, x.device)
                be.gemm(own, w, out=hw[:pl.rows])
                key2 = ("hw2", l)
                fin = self._exchange_async(hw[:pl.rows], self._local(key2, d_out, x.device), ("hw", l))
                nxt = None if last else self._local(("y", l), d_out, x.device)
                out = self._aggregate_two_pass(key2, hw[:pl.rows], d_out, fin, tail, out=None if last else nxt[:pl.rows])
                full, cols, own, pending = None, pl.cols, out, None
                if not last:
                    if not dims[l + 1][0] > dims[l + 1][1]:
                        pending = (("y2", l), self._exchange_async(own, self._local(("y2", l), d_out, x.device), ("y", l)))
                continue
            # aggregate-first layers need the mean of the own rows over ALL their edges
            widening = not last and 2 * d_in <= d_out
            a_buf = self._local(("agg", l), d_in, x.device) if widening else None
            a_out = a_buf[:pl.rows] if widening else None
            if pending is not None:
                agg = self._aggregate_two_pass(pending[0], own, d_in, pending[1], out=a_out)
                pending = None
            elif full is not None:
                agg = be.spmm(g.indptr, cols, full, pl.rows, be.AGG_SAGE_GCN, x_self=own, out=a_out)
            else:
                raise RuntimeError("HaloShardedTeacher: internal error, an aggregating layer needs its halo rows")
            if widening:                          # exchange the narrow aggregate, project own rows meanwhile, halo rows after
                fin = self._exchange_async(a_buf[:pl.rows], a_buf, ("agg", l))
                nxt = self._local(("y", l), d_out, x.device)
                be.gemm(a_buf[:pl.rows], w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=nxt[:pl.rows])
                fin()
                if pl.n_halo:
                    be.gemm(a_buf[pl.rows:], w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=nxt[pl.rows:])
                full, cols, own = nxt, pl.cols, nxt[:pl.rows]
                continue
            nxt = None if last else self._local(("y", l), d_out, x.device)
            out = be.feat_empty(pl.rows, d_out, x.device) if last else nxt[:pl.rows]
            be.gemm(agg, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out)
            if not last:
                own, full, cols = nxt[:pl.rows], None, pl.cols
                if not dims[l + 1][0] > dims[l + 1][1]:      # the next layer aggregates these rows: start their exchange now
                    pending = (("y2", l), self._exchange_async(own, self._local(("y2", l), d_out, x.device), ("y", l)))
        return out

    def forward(self, x_full):
        """x_full: [>= n, F] replicated input features.  Returns this rank's rows of the logits [rows, C]."""
        if self.overlap:
            return self._forward_overlapped(x_full)
        enc, sh, be, g, pl = self.enc, self.sh, self.be, self.g, self.plan
        x = be.as_feat(x_full)
        cols, x_self = g.indices, x[sh.lo:sh.hi]          # layer 1 gathers from the replicated input with the original ids
        complete = True
        L = enc.num_layers
        dims = [(lay.fc_neigh.weight.shape[1], lay.fc_neigh.weight.shape[0]) for lay in enc.layers]
        out = None
        for l in range(L):
            w = enc.layers[l].fc_neigh.weight
            ep_scale, ep_shift, relu = enc._tail(l)
            d_in, d_out = dims[l]
            last = l == L - 1
            if d_in > d_out:                                   # narrowing: project own rows, exchange d_out-wide rows, aggregate
                hw = self._local(("hw", l), d_out, x.device)
                be.gemm(x_self, w, out=hw[:pl.rows])
                self._exchange(hw, ("hw", l))
                out = be.feat_empty(pl.rows, d_out, x.device) if last else self._local(("y", l), d_out, x.device)[:pl.rows]
                be.spmm(g.indptr, pl.cols, hw, pl.rows, be.AGG_SAGE_GCN, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out,
                        x_self=hw[:pl.rows])
                nxt = None if last else self._local(("y", l), d_out, x.device)
            elif not complete:
                raise RuntimeError("HaloShardedTeacher: internal error, an aggregating layer needs its halo rows")
            elif not last and 2 * d_in <= d_out:               # widening: exchange the narrow aggregate, project own + halo rows
                agg = self._local(("agg", l), d_in, x.device)
                be.spmm(g.indptr, cols, x, pl.rows, be.AGG_SAGE_GCN, out=agg[:pl.rows], x_self=x_self)
                self._exchange(agg, ("agg", l))
                nxt = self._local(("y", l), d_out, x.device)
                be.gemm(agg, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=nxt)
                x, cols, x_self, complete = nxt, pl.cols, nxt[:pl.rows], True
                continue
            else:                                              # plain aggregate-first layer on the own rows
                nxt = None if last else self._local(("y", l), d_out, x.device)
                out = be.feat_empty(pl.rows, d_out, x.device) if last else nxt[:pl.rows]
                if hasattr(be, "sage_fused") and d_in <= 256 and d_out <= 256:
                    be.sage_fused(g.indptr, cols, x, pl.rows, w, ep_scale=ep_scale, ep_shift=ep_shift, relu=relu, out=out, x_self=x_self)
                else:
                    be.gemm(be.spmm(g.indptr, cols, x, pl.rows, be.AGG_SAGE_GCN, x_self=x_self), w, ep_scale=ep_scale, ep_shift=ep_shift,
                            relu=relu, out=out)
            if not last:
                if dims[l + 1][0] > dims[l + 1][1]:
                    complete = False                           # the next (narrowing) layer projects its own rows only
                else:
                    self._exchange(nxt, ("y", l))
                    complete = True
                x, cols, x_self = nxt, pl.cols, nxt[:pl.rows]
        return out


def make_grad_sync(flat_grads, world, group=None, average=False):
    """Gradient exchange for the data-parallel student: one all-reduce over the engine's flat grad buffer."""
    if world == 1:
        return None

    def sync():
        _inject("grad")
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat_grads.mul_(1.0 / world)
    return sync


class OverlappedGradSync:
    """Data-parallel gradient exchange of the student that starts INSIDE the backward: the C step calls `grad_ready(layer)`
    (glnn_mlp_step_desc.grad_ready) as soon as a layer's weight-gradient GEMM is enqueued; weight gradients of at least
    `min_bytes` are all-reduced right there, asynchronously on the communicator's stream (RCCL orders itself after the
    compute stream at the call), while the compute stream goes on with the input gradient, the BatchNorm backward and the
    layers in front.  `finish()` -- the engine's `grad_sync`, run between the step and Adam -- reduces the remaining small
    ranges of the flat buffer (biases, BatchNorm parameters, thin layers) and makes the compute stream wait for everything.
    MLP3w8: the 16 MB hidden-layer gradient rides under ~0.35 ms of remaining backward; ~1.2 MB are left for the end.
    Every rank issues the same collectives in the same order (the hook order is the backward's order)."""

    def __init__(self, engine, world, group=None, average=True, min_bytes=1 << 20):
        from . import _lib
        self.eng, self.world, self.group, self.average = engine, world, group, average
        flat = engine.flat_grads
        base = flat.data_ptr()
        self.big = {}                                   # layer -> view of the flat buffer
        taken = []
        for l, w in enumerate(engine.W):
            g = engine._grad(w)
            if g.numel() * 4 >= min_bytes:
                off = (g.data_ptr() - base) // 4
                self.big[l] = flat[off:off + g.numel()]
                taken.append((off, off + g.numel()))
        taken.sort()
        self.rest, pos = [], 0                          # the complement, as contiguous views
        for lo, hi in taken:
            if lo > pos:
                self.rest.append(flat[pos:lo])
            pos = hi
        if pos < flat.numel():
            self.rest.append(flat[pos:])
        self.works, self.error, self.calls, self.collectives = [], None, 0, 0
        self.callback = _lib.GRAD_READY_FN(self._hook)
        engine.desc.grad_ready = self.callback
        engine.grad_sync = self.finish

    def detach(self):
        """Take the hook out of the engine again (the bench's ladder falls back to one all-reduce after the backward)."""
        from . import _lib
        self.eng.desc.grad_ready = _lib.GRAD_READY_FN()
        self.eng.grad_sync = None
        self.works, self.error = [], None

    def _reduce(self, t, async_op):
        _inject("grad_overlap")
        if FORCE_COLLECTIVES or self.world > 1:
            self.collectives += 1                       # (not EXCHANGE_STATS: that is the teacher's feature exchange)
            if self.average and dist.get_backend(self.group) == "nccl":
                return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if self.average:                            # gloo has no AVG: scale after the (then synchronous) reduce
                if w is not None:
                    w.wait()
                t.mul_(1.0 / self.world)
                return None
            return w
        return None

    def _hook(self, ctx, layer, stream):
        try:
            self.calls += 1
            t = self.big.get(int(layer))
            if t is not None:
                # `stream` is the stream that orders gw[layer]: the step's own stream, or its aux stream when the weight
                # gradients run there (two-stream backward).  The collective must be ordered behind THAT stream.
                cur = torch.cuda.current_stream(t.device) if t.is_cuda else None
                if cur is not None and stream and int(stream) != cur.cuda_stream:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=t.device)):
                        w = self._reduce(t, True)
                else:
                    w = self._reduce(t, True)
                if w is not None:
                    self.works.append(w)
            return 0
        except Exception as e:                          # ctypes cannot propagate through the C frame
            self.error = e
            return 1

    def finish(self):
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError("OverlappedGradSync: a gradient all-reduce failed inside the backward") from err
        for t in self.rest:
            w = self._reduce(t, True)
            if w is not None:
                self.works.append(w)
        for w in self.works:
            w.wait()                                    # the compute stream waits for the communicator's stream
        self.works = []


class StatExchange:
    """The host side of glnn_exchange_fn (include/glnn_hip.h): the all-gather of per-rank BatchNorm sums that
    glnn_mlp_fwd_bwd_f32 asks for when one batch is split over ranks (global batch statistics keep the
    step identical to the single-GPU reference step).  Owns the send/recv/row-count device buffers the descriptor
    points at; `callback` is the C function pointer.  An exception inside the hook is kept in `.error` and reported
    to the library as a non-zero status (ctypes cannot propagate it through the C frame)."""

    def __init__(self, world, rank, max_hidden, device, group=None):
        from . import _lib
        self.world, self.rank, self.group = world, rank, group
        f32 = dict(dtype=torch.float32, device=device)
        self.send = torch.zeros(3 * max_hidden, **f32)
        self.recv = torch.zeros(world * 3 * max_hidden, **f32)
        self.rows = torch.zeros(1, **f32)
        self.error = None
        self.calls = 0
        self.callback = _lib.EXCHANGE_FN(self._hook)

    def _hook(self, ctx, send, recv, floats, stream):
        try:
            n = int(floats)
            if send != self.send.data_ptr() or recv != self.recv.data_ptr() or n > self.send.numel():
                raise RuntimeError("StatExchange: the library passed buffers this object does not own")
            src, dst = self.send[:n], self.recv[:self.world * n]
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(dst, src, group=self.group)        # RCCL, ordered after the current stream
            else:
                parts = [torch.empty_like(src) for _ in range(self.world)]
                dist.all_gather(parts, src.clone(), group=self.group)
                torch.cat(parts, out=dst)
            self.calls += 1
            return 0
        except BaseException as e:      # noqa: BLE001 -- must not unwind through the C caller
            self.error = e
            return 1
