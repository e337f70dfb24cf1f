"""Thin torch-tensor wrappers over the C ABI (include/glnn_hip.h).  torch is plumbing here: it owns
device memory and the current HIP stream; all arithmetic is in libglnn_hip.so.  Every function
raises on CPU tensors -- there is no fallback path."""
import ctypes

import torch

from . import _lib

AGG_SUM, AGG_SAGE_GCN = 0, 1
LOSS_NLL, LOSS_KL = 0, 1

# Optional per-launch timing (bench.py's roofline leg): a list that receives
# (kernel_name, info_dict, start_event, end_event) for every aggregation / GEMM launch, recorded on the
# stream the kernel is launched on.  None = no events (the default).
_TIMING = None


def set_timing(collector):
    global _TIMING
    _TIMING = collector


class _Timed:
    __slots__ = ("name", "info", "s")

    def __init__(self, name, **info):
        self.name, self.info, self.s = name, info, None

    def __enter__(self):
        if _TIMING is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *exc):
        if self.s is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            _TIMING.append((self.name, self.info, self.s, e))


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a C pointer.  torch.cuda.current_stream() builds a Stream object and
    resolves the device through several Python layers (~8 us per call: 50 us of host time per sampled batch); the raw getter
    torch itself uses for its compiled kernels returns the same handle in well under a microsecond."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GlnnError("glnn_amd.ops: tensors must live on the GPU (HIP path only, no CPU fallback)")


def _vec(t, n, name, dtype=torch.float32):
    if t is None:
        return None
    if t.dtype != dtype or t.dim() != 1 or t.numel() < n or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype} vector with >= {n} elements")
    return t


def _mat(t, name):
    if t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected 2-D float32 row-major tensor")
    return t


def _ld(t):
    # the stride of a size-1 dimension is arbitrary in torch/numpy: a single row is as wide as its columns
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def round4(d):
    return (d + 3) // 4 * 4


def feat_empty(n, d, device, zero=False):
    """[n, d] fp32 view of an [n, round4(d)] buffer: rows 16-byte aligned as the float4 kernels need."""
    ld = round4(d)
    buf = (torch.zeros if zero else torch.empty)((n, ld), dtype=torch.float32, device=device)
    return buf[:, :d]


_RANDPERM_LOCK = __import__("threading").Lock()


def randperm_cpu(n):
    """torch.randperm(n) from the global CPU generator (what the reference draws its mini-batches with, train_and_eval.py:66), issued
    with ONE intra-op thread: the CPU kernel is a serial shuffle, so the permutation is the same for any thread count
    (scripts/probe_randperm.py), but with the 128-thread pool of a GPU host awake the 0.3 ms job took 7-17 ms -- as long as the 177
    optimiser steps of an ogbn-arxiv pass.  Side effect, by design: torch's process-global intra-op thread count is 1 for the ~0.3 ms of
    the call (serialised by a lock; a CPU op running concurrently in another thread of the host program sees one thread meanwhile)."""
    with _RANDPERM_LOCK:
        k = torch.get_num_threads()
        if k <= 1:
            return torch.randperm(n)
        torch.set_num_threads(1)
        try:
            return torch.randperm(n)
        finally:
            torch.set_num_threads(k)


_PADDED = []            # [(weakref to the source tensor, (version, data_ptr, strides), padded copy)], most recent first, <= 2 live entries


def _pad_key(t):
    try:
        ver = t._version
    except RuntimeError:          # tensors created under torch.inference_mode() have no version counter: never cached
        return None
    return (ver, t.data_ptr(), tuple(t.stride()), tuple(t.shape))


def clear_pad_cache():
    """Drop the padded copies as_feat remembers (they are only ever handed out for the very same, unmodified source tensor)."""
    _PADDED.clear()


def as_feat(t):
    """Return t itself if its layout suits the float4 kernels, else a padded copy.
    The copy of a LARGE matrix is remembered while the very same tensor object is passed again unmodified (identity + torch's in-place
    version counter + data pointer and strides): the reference hands the same `feats` to every epoch's train / evaluate call, and re-padding
    a wide unaligned feature matrix (cora 1433, citeseer 3703 columns) was a zero fill and a copy per call.  Writes that bypass the version
    counter -- raw-pointer writes by this library's own kernels, `.data` assignments -- are NOT seen: such tensors are never inputs of
    as_feat in this package (parameters skip the cache through requires_grad).  Entries whose source died are evicted on every call.
    The result of a padding call is READ-ONLY for the caller (it may be handed out again); every call site in this package only reads it."""
    _mat(t, "as_feat")
    if t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t
    _PADDED[:] = [e for e in _PADDED if e[0]() is not None]
    key = _pad_key(t) if (t.numel() >= (1 << 20) and not t.requires_grad) else None
    if key is not None:
        for i, (ref, k, pad) in enumerate(_PADDED):
            if ref() is t and k == key and pad.device == t.device:
                if i:
                    _PADDED.insert(0, _PADDED.pop(i))
                return pad
    out = feat_empty(t.shape[0], t.shape[1], t.device, zero=True)
    out.copy_(t)
    if key is not None:
        import weakref
        _PADDED[:] = [e for e in _PADDED if e[0]() is not t][:1]
        _PADDED.insert(0, (weakref.ref(t), key, out))
    return out


# ---------------------------------------------------------------------------------------------
class HubPlan:
    """glnn_hub_plan for one (indptr, n_dst) launch range: the rows of more than glnn_hub_row_threshold() in-edges, cut into segments of
    glnn_hub_segment_edges() edges that one workgroup each gathers in front of the aggregation (include/glnn_hip.h, ABI 9).  Built once
    per range (one pass over the degrees and ONE host read-back: the segment count) and reused by every launch over that range; the slab
    is scratch of the launching stream (one stream at a time)."""

    def __init__(self, indptr, n_dst):
        thr, seg = _lib.lib().glnn_hub_row_threshold(), _lib.lib().glnn_hub_segment_edges()
        deg = indptr[1:n_dst + 1] - indptr[:n_dst]
        rows = torch.nonzero(deg > thr).flatten()
        self.n_hub = int(rows.numel())
        self.rows = rows.contiguous()
        segs = (deg[rows] + (seg - 1)) // seg
        self.seg_ptr = torch.zeros(self.n_hub + 1, dtype=torch.int32, device=indptr.device)
        if self.n_hub:
            self.seg_ptr[1:] = segs.cumsum(0).to(torch.int32)
        self.n_seg = int(self.seg_ptr[-1].item()) if self.n_hub else 0
        self.hub_edges = int(deg[rows].sum().item()) if self.n_hub else 0
        self.slab = None
        self.desc = _lib.HubPlanDesc()

    def desc_for(self, d, device):
        """The C descriptor with a slab wide enough for rows of d floats (None when the range has no hub row)."""
        if self.n_hub == 0:
            return None
        ld = (d + 3) // 4 * 4
        if self.slab is None or self.slab.shape[1] < ld:
            self.slab = torch.empty(8 * self.n_seg, ld, dtype=torch.float32, device=device)      # (row 8 s + w: wave w's piece of segment s)
        p = self.desc
        p.rows, p.seg_ptr, p.n_hub, p.n_seg = _p(self.rows), _p(self.seg_ptr), self.n_hub, self.n_seg
        p.slab, p.ld_slab, p.slab_rows = _p(self.slab), self.slab.shape[1], self.slab.shape[0]
        return ctypes.byref(p)


def hub_plan(indptr, n_dst):
    """A HubPlan for launches over rows [0, n_dst) of `indptr`, or None when it has no hub row (cache it per range: it costs a host sync)."""
    _need_cuda(indptr)
    plan = HubPlan(indptr, int(n_dst))
    return plan if plan.n_hub else None


def spmm(indptr, indices, x, n_dst, mode, row_scale=None, col_scale=None, ep_scale=None, ep_shift=None,
         relu=False, out=None, x_self=None, self_rows=None, hub=None, chunks=None):
    """K1/K2 glnn_spmm_csr_f32.  x: [n_src, d] feature tensor (see as_feat); returns [n_dst, d].
    chunks (ChunkSignals.launch(...), SAGE_GCN, d <= 256): ONE launch over the chunks of the row range -- `out` (and `x_self`, unless
    self_rows is given) are then WHOLE buffers addressed through the descriptor's per-chunk rows, and every chunk signals its completion.
    x_self (SAGE_GCN only): the destination rows' own features, default x[:n_dst] (a row shard passes its slice).
    self_rows (SAGE_GCN only, int64 [n_dst]): destination v's own row is x_self[self_rows[v]] (global-id blocks).
    hub (optional HubPlan of (indptr, n_dst)): the hub rows' segments are gathered by one workgroup each first (same result bit for bit)."""
    _need_cuda(indptr, indices, x, row_scale, col_scale, ep_scale, ep_shift, out, x_self, self_rows)
    x = as_feat(x)
    if self_rows is not None and (self_rows.dtype != torch.int64 or not self_rows.is_contiguous() or self_rows.numel() < n_dst):
        raise ValueError("spmm: self_rows must be a contiguous int64 vector of n_dst row ids")
    if x_self is None:
        x_self = x
    elif (self_rows is None and chunks is None and x_self.shape[0] < n_dst) or x_self.shape[1] != x.shape[1]:
        raise ValueError("spmm: x_self must hold n_dst rows of the same width as x")
    if chunks is not None and (out is None or mode != AGG_SAGE_GCN):
        raise ValueError("spmm(chunks=...): SAGE_GCN with the whole output buffer given (it is addressed by the chunks' rows)")
    n_src, d = x.shape
    if indptr.dtype != torch.int64 or indices.dtype != torch.int32:
        raise ValueError("spmm: indptr must be int64 and indices int32")
    if out is None:
        out = feat_empty(n_dst, d, x.device)
    _mat(out, "spmm out")
    with _Timed("spmm", d=d, n_dst=n_dst, mode=mode):
        rc = _spmm_call(indptr, indices, n_dst, n_src, x, d, mode, row_scale, col_scale, ep_scale, ep_shift, relu, out,
                        as_feat(x_self), self_rows, hub, chunks)
    _lib.check(rc, "glnn_spmm_csr_f32")
    return out


def _spmm_call(indptr, indices, n_dst, n_src, x, d, mode, row_scale, col_scale, ep_scale, ep_shift, relu, out, x_self,
               self_rows=None, hub=None, chunks=None):
    args = (_p(indptr), _p(indices), n_dst, n_src, _p(x), _ld(x), d, mode,
            _p(_vec(row_scale, n_dst, "row_scale")), _p(_vec(col_scale, n_src, "col_scale")),
            _p(x_self) if mode == AGG_SAGE_GCN else None, _ld(x_self), _p(self_rows),
            _p(_vec(ep_scale, d, "ep_scale")), _p(_vec(ep_shift, d, "ep_shift")), 1 if relu else 0,
            _p(out), _ld(out))
    plan = hub.desc_for(d, x.device) if hub is not None else None
    if chunks is not None:
        return _lib.lib().glnn_spmm_csr_chunks_f32(*args, plan, ctypes.byref(chunks), _stream())
    if plan is not None:
        return _lib.lib().glnn_spmm_csr_plan_f32(*args, plan, _stream())
    return _lib.lib().glnn_spmm_csr_f32(*args, _stream())


# Parameters and BatchNorm buffers are also written THROUGH RAW POINTERS by this library (the fused Adam launch, the statistics kernels, the
# one-call training steps): torch's version counters do not see those writes.  Every such call site bumps PARAM_EPOCH, and whatever is
# derived from parameters and remembered across calls (packed weights, folded eval-mode BatchNorm tails) is keyed by (identity, torch
# version, PARAM_EPOCH).
PARAM_EPOCH = 0


def note_param_write():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


_PACKED = []          # [(weakref(w), key, packed)], most recent first


def pack_weight(w):
    """glnn_pack_weight_f32: W [d_out, d_in] -> MFMA B-fragment order for sage_fused.  The packed copy of a weight is remembered while
    the same unmodified tensor comes back (an eval-mode teacher forward packs the same three weights on every call: three launches of
    the ~0.9 ms arxiv forward); READ-ONLY for the caller."""
    _need_cuda(w)
    key = (w.data_ptr(), tuple(w.shape), w.stride(0), w._version, PARAM_EPOCH)
    for i, (ref, k, wp) in enumerate(_PACKED):
        if ref() is w and k == key:
            if i:
                _PACKED.insert(0, _PACKED.pop(i))
            return wp
    wp = _pack_weight(w)
    import weakref
    _PACKED[:] = [e for e in _PACKED if e[0]() is not None and e[0]() is not w][:15]
    _PACKED.insert(0, (weakref.ref(w), key, wp))
    return wp


def _pack_weight(w):
    _mat(w, "pack_weight w")
    d_out, d_in = w.shape
    wp = torch.empty(_lib.lib().glnn_packed_weight_floats(d_out, d_in), dtype=torch.float32, device=w.device)
    rc = _lib.lib().glnn_pack_weight_f32(_p(w), _ld(w), d_out, d_in, _p(wp), _stream())
    _lib.check(rc, "glnn_pack_weight_f32")
    return wp


def fused_tile_order(indptr, n_dst):
    """The 32-row tiles of a fused launch over rows [0, n_dst) of `indptr`, heaviest row first (stable): int32 permutation for
    sage_fused(tile_order=...).  One-time index arithmetic per graph / row range (cache it)."""
    deg = (indptr[1:n_dst + 1] - indptr[:n_dst])
    tiles = (n_dst + 31) // 32
    pad = tiles * 32 - n_dst
    if pad:
        deg = torch.cat([deg, deg.new_zeros(pad)])
    return torch.argsort(deg.view(tiles, 32).amax(1), descending=True, stable=True).to(torch.int32)


class ChunkSignals:
    """Completion signals of ONE fused launch over the chunks of a row range (glnn_sage_fused_chunks_f32, round 6): per chunk a signal
    word a stream can wait on and an arrival counter.  `row_start`: first row of every chunk inside the launch's row range + the end
    (multiples of 32).  Per launch: sage_fused(..., chunks=sig.launch(self_rows, out_rows)) stamps a new epoch; then
    sig.wait(stream, c) holds `stream`'s next operation until chunk c's rows are stored and written back -- while the launch still runs.
    Chunks without rows are never signalled (`sig.empty(c)`: do not wait for them)."""
    MAX = _lib.MAX_CHUNKS

    def __init__(self, row_start, device):
        n = len(row_start) - 1
        if not 1 <= n <= self.MAX or row_start[0] != 0 or any(r % 32 for r in row_start[:-1]) or any(a > b for a, b in zip(row_start[:-1], row_start[1:])):
            raise ValueError(f"ChunkSignals: 1..{self.MAX} chunks whose first rows are ascending multiples of 32, starting at 0")
        self.row_start, self.n, self.device, self.epoch = [int(r) for r in row_start], n, device, 0
        self.arrivals = torch.zeros(n, dtype=torch.int32, device=device)
        self._signals = []
        for _ in range(n):
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().glnn_signal_alloc(ctypes.byref(p)), "glnn_signal_alloc")
            self._signals.append(p.value)
        self._n_dst = None

    def empty(self, c, n_dst=None):
        n_dst = self._n_dst if n_dst is None else n_dst
        return min(self.row_start[c + 1], n_dst) <= self.row_start[c]

    def launch(self, self_rows, out_rows, n_dst):
        """The descriptor of the next launch (a new epoch): self_rows[c] / out_rows[c] = the rows of x_self / of the outputs that hold the
        chunk's first row."""
        if len(self_rows) != self.n or len(out_rows) != self.n or self.row_start[-1] < n_dst:
            raise ValueError("ChunkSignals.launch: one self row and one output row per chunk, chunks covering the launch's rows")
        self.epoch += 1
        self._n_dst = int(n_dst)
        d = _lib.ChunkSignalsDesc()
        d.n_chunks = self.n
        for c in range(self.n + 1):
            d.row_start[c] = self.row_start[c]
        for c in range(self.n):
            d.self_row[c], d.out_row[c], d.signal[c] = int(self_rows[c]), int(out_rows[c]), self._signals[c]
        d.arrivals = self.arrivals.data_ptr()
        d.epoch = self.epoch
        return d

    def wait(self, stream, c):
        """`stream` (torch.cuda.Stream) runs nothing further until chunk c of the LAST launch is complete."""
        _lib.check(_lib.lib().glnn_stream_wait_value32(ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(self._signals[c]), self.epoch),
                   "glnn_stream_wait_value32")

    def value(self, c):
        v = ctypes.c_uint32()
        _lib.check(_lib.lib().glnn_signal_read(ctypes.c_void_p(self._signals[c]), ctypes.byref(v)), "glnn_signal_read")
        return v.value

    def __del__(self):
        try:
            for p in self._signals:
                _lib.lib().glnn_signal_free(ctypes.c_void_p(p))
        except Exception:
            pass


def sage_fused(indptr, indices, x, n_dst, w, ep_scale=None, ep_shift=None, relu=False, out=None, x_self=None, w_packed=None,
               w_next=None, out_next=None, want_out=True, tile_order=None, hub=None, chunks=None):
    """K1F glnn_sage_fused_f32: epi(((A x + x_self)/(deg+1)) @ w.T) in one launch (d_in, d_out <= 256).
    w_next [d_out2, d_out]: also returns (.. , out @ w_next.T) -- the projection of the NEXT layer when it projects first;
    with want_out=False the hidden rows themselves are not written at all (returns (None, projected)).
    chunks (ChunkSignals.launch(...)): the launch covers the chunks of a row range -- x_self / out / out_next are then WHOLE buffers
    addressed through the descriptor's per-chunk rows (they must be given), and every chunk signals its completion."""
    _need_cuda(indptr, indices, x, w, ep_scale, ep_shift, out, x_self, w_next, out_next)
    if chunks is not None and (x_self is None or (out is None and want_out) or (w_next is not None and out_next is None)):
        raise ValueError("sage_fused(chunks=...): x_self and the output buffers must be given (they are addressed by the chunks' rows)")
    x = as_feat(x)
    x_self = x if x_self is None else as_feat(x_self)
    n_src, d_in = x.shape
    d_out = w.shape[0]
    if w.shape[1] != d_in:
        raise ValueError("sage_fused: weight must be [d_out, d_in]")
    if tile_order is not None and (tile_order.dtype != torch.int32 or not tile_order.is_contiguous() or tile_order.numel() != (n_dst + 31) // 32):
        raise ValueError("sage_fused: tile_order must be a contiguous int32 permutation of the ceil(n_dst / 32) tile ids")
    if w_packed is None:
        w_packed = pack_weight(w)
    if out is None and (want_out or w_next is None):
        out = feat_empty(n_dst, d_out, x.device)
    w2p, d_out2 = None, 0
    if w_next is not None:
        if w_next.shape[1] != d_out:
            raise ValueError("sage_fused: w_next must be [d_out2, d_out]")
        w2p, d_out2 = pack_weight(w_next), w_next.shape[0]
        if out_next is None:
            out_next = feat_empty(n_dst, d_out2, x.device)
    with _Timed("sage_fused", d=d_in, n_dst=n_dst, d_out=d_out, d_chain=d_out2, d_written=(d_out if out is not None else 0) + d_out2):
        args = (_p(indptr), _p(indices), n_dst, n_src, _p(x), _ld(x), d_in, _p(x_self), _ld(x_self),
                _p(w_packed), d_out, _p(_vec(ep_scale, d_out, "ep_scale")),
                _p(_vec(ep_shift, d_out, "ep_shift")), 1 if relu else 0, _p(out), _ld(out) if out is not None else 0,
                _p(w2p), d_out2, _p(out_next), _ld(out_next) if out_next is not None else 0, _p(tile_order))
        plan = hub.desc_for(d_in, x.device) if hub is not None else None
        if chunks is not None:
            rc = _lib.lib().glnn_sage_fused_chunks_f32(*args, plan, ctypes.byref(chunks), _stream())
        elif plan is not None:
            rc = _lib.lib().glnn_sage_fused_plan_f32(*args, plan, _stream())
        else:
            rc = _lib.lib().glnn_sage_fused_f32(*args, _stream())
    _lib.check(rc, "glnn_sage_fused_f32")
    return out if w_next is None else (out, out_next)


class RowRangeLaunch:
    """A PREPARED SAGE-"gcn" launch over row ranges [s, e) of one resident CSR whose column ids are rows of `x` (the engine-mode chunked
    sweep of SAGE.inference, reference models.py:133-145): everything that does not change from chunk to chunk -- the checks of sage_fused /
    spmm, the packed weight, the C argument list -- is done ONCE here; a call only offsets three pointers (indptr + s, the self rows
    x[s:], the output rows out[s:]) and launches.  Same library entries, same arguments, same bits as the per-chunk ops calls (~8 us of
    host time per chunk instead of ~25).  w = None: the stand-alone aggregation (a layer that projected first), else the fused kernel."""

    def __init__(self, indptr, indices, x, out, w=None, ep_scale=None, ep_shift=None, relu=False, w_packed=None):
        _need_cuda(indptr, indices, x, out, w, ep_scale, ep_shift)
        x = as_feat(x)
        _mat(out, "RowRangeLaunch out")
        if indptr.dtype != torch.int64 or indices.dtype != torch.int32 or out.shape[0] < indptr.numel() - 1:
            raise ValueError("RowRangeLaunch: indptr int64, indices int32, one output row per CSR row")
        n_src, d_in = x.shape
        self.keep = (indptr, indices, x, out, w, ep_scale, ep_shift)          # (the pointers below stay valid while this object lives)
        self.n = indptr.numel() - 1
        self.ip0, self.x0, self.ldx, self.o0, self.ldo = indptr.data_ptr(), x.data_ptr(), _ld(x), out.data_ptr(), _ld(out)
        if w is None:
            d = d_in
            if out.shape[1] != d:
                raise ValueError("RowRangeLaunch: out must be as wide as x")
            self.fused, self.info = False, dict(d=d, mode=AGG_SAGE_GCN)
            self.tail = (_p(_vec(ep_scale, d, "ep_scale")), _p(_vec(ep_shift, d, "ep_shift")), 1 if relu else 0)
            self.head = (_p(indices),)
            self.mid = (n_src, self.x0, self.ldx, d, AGG_SAGE_GCN, None, None)
        else:
            d_out = w.shape[0]
            if w.shape[1] != d_in or d_in > 256 or d_out > 256 or out.shape[1] != d_out:
                raise ValueError("RowRangeLaunch: weight [d_out, d_in] with d_in, d_out <= 256, out [n, d_out]")
            wp = pack_weight(w) if w_packed is None else w_packed
            self.keep += (wp,)
            self.fused, self.info = True, dict(d=d_in, d_out=d_out, d_chain=0, d_written=d_out)
            self.head = (_p(indices),)
            self.mid = (n_src, self.x0, self.ldx, d_in)
            self.tail = (_p(wp), d_out, _p(_vec(ep_scale, d_out, "ep_scale")), _p(_vec(ep_shift, d_out, "ep_shift")), 1 if relu else 0)
        self.fn = _lib.lib().glnn_sage_fused_f32 if self.fused else _lib.lib().glnn_spmm_csr_f32

    def __call__(self, s, e):
        if not 0 <= s <= e <= self.n:
            raise ValueError("RowRangeLaunch: row range outside the CSR")
        n_dst = e - s
        ip, xs, o = self.ip0 + 8 * s, self.x0 + 4 * self.ldx * s, self.o0 + 4 * self.ldo * s
        if self.fused:
            args = (ip,) + self.head + (n_dst,) + self.mid + (xs, self.ldx) + self.tail + (o, self.ldo, None, 0, None, 0, None, _stream())
        else:
            args = (ip,) + self.head + (n_dst,) + self.mid + (xs, self.ldx, None) + self.tail + (o, self.ldo, _stream())
        if _TIMING is not None:
            with _Timed("sage_fused" if self.fused else "spmm", n_dst=n_dst, **self.info):
                rc = self.fn(*args)
        else:
            rc = self.fn(*args)
        if rc != 0:
            _lib.check(rc, "glnn_sage_fused_f32" if self.fused else "glnn_spmm_csr_f32")


# ---- placement of gathered matrices (round 5) ------------------------------------------------------------------------------------
# The same aggregation launch over the same graph runs 18.1 ... 19.4 ms (fused D=256, products shape) depending on WHICH allocation
# holds the matrix it gathers from -- stable per buffer, different from one buffer to the next inside one process, unaffected by the
# offset inside an allocation (profiles/r05_bimodal_launch.txt, scripts/bimodal_probe.py): where the driver put the buffer's pages in
# HBM.  A matrix that many launches will gather from is therefore PLACED: a few candidate allocations are timed with a gather over the
# very graph and the fastest one is kept.  One-time set-up work per (graph, matrix), like the tile order or a hub plan; results are
# untouched (it only chooses which memory holds the rows).  Only for matrices far beyond the caches (>= PLACEMENT_MIN_BYTES) that still
# fit many times (<= PLACEMENT_MAX_BYTES).
import os as _os
PLACEMENT_CANDIDATES = int(_os.environ.get("GLNN_PLACEMENT_CANDIDATES", "12"))     # <= 1: off (the first allocation is used)
# Candidates are made in groups of four with a BALLAST allocation between the groups (alive until the choice is made), so that the groups
# come from different regions of HBM: in some processes every allocation of the first region is slow (scripts/placement_ballast_probe.py:
# with three or four regions a 17.9-18.0 ms candidate was found in every one of five processes, the first region alone gave 17.96-18.45).
# Ballast = this fraction of the free device memory, at most PLACEMENT_BALLAST_MAX bytes per gap; 0 = none.
PLACEMENT_BALLAST_FRAC = float(_os.environ.get("GLNN_PLACEMENT_BALLAST_FRAC", "0.15"))
PLACEMENT_BALLAST_MAX = 48 << 30
PLACEMENT_MIN_BYTES = 1 << 30        # (matrices of 0.5-1 GB -- the products features, the 47-wide projection -- showed 1 % spreads: not worth a search)
PLACEMENT_MAX_BYTES = 8 << 30
PLACEMENT_LOG = []             # one record per tuned matrix: {"what", "rows", "d", "ms": [...], "chosen"} (bench.py prints it)


def placement_applies(rows, d):
    nbytes = 4 * rows * round4(d)
    return PLACEMENT_CANDIDATES > 1 and PLACEMENT_MIN_BYTES <= nbytes <= PLACEMENT_MAX_BYTES


def placed_for_gather(rows, d, device, indptr, indices, n_dst, what="", first=None, zero=False, probe=None):
    """An [rows, d] feature buffer for a matrix that launches over (indptr, indices) will gather from, in the allocation where that
    gather runs fastest among PLACEMENT_CANDIDATES tries (all alive at once, so that they are different memory; the losers are freed).
    `first` (optional, [rows, d]): an existing buffer that competes as candidate 0 (returned itself if it wins).  probe(candidate): the
    launch to time -- the caller's own consumer of the matrix where it has one (SAGE.inference times the next layer's launch); default: the
    stand-alone SAGE-gcn aggregation of the candidate.  (A candidate's CONTENT is irrelevant to the timing: zeros when `zero`, else
    whatever the allocator left.)"""
    if not placement_applies(rows, d):
        return first if first is not None else feat_empty(rows, d, device, zero=zero)
    # (What makes an allocation slow is its physical backing: the same virtual addresses gather in 18.0 ms when the request was served from
    #  an unfragmented free pool and in 19.0-19.5 ms when small allocations were made in between, or out of a re-used cached block with
    #  that history: scripts/placement_cause_probe.py, profiles/r05_placement_cause.txt.  Not predictable from here -- hence the probe.)
    # The search is an optimisation and must never be what runs a device out of memory: the candidates + the ballast together
    # stay below the free memory minus a headroom of two more matrices (the probe's own scratch / the next layer's output are allocated
    # while the candidates are alive), the probe loop is under the same handler as the allocations, and every failure falls back to the
    # plain allocation the caller would have made without the search.
    nbytes_m = 4 * rows * round4(d)
    cands, ballast = ([first] if first is not None else []), []
    scratch = None

    def fallback():
        del cands[:], ballast[:]
        torch.cuda.empty_cache()
        return first if first is not None else feat_empty(rows, d, device, zero=zero)

    try:
        budget = torch.cuda.mem_get_info(device)[0] - 2 * nbytes_m
        want = PLACEMENT_CANDIDATES
        while len(cands) < want:
            if budget < nbytes_m:
                break                                     # (a nearly full or shared device: choose among what fits)
            if len(cands) and len(cands) % 4 == 0 and PLACEMENT_BALLAST_FRAC > 0:
                free = torch.cuda.mem_get_info(device)[0]
                nb = int(min(PLACEMENT_BALLAST_FRAC * free, PLACEMENT_BALLAST_MAX))
                # (never squeeze the candidates themselves, nor the headroom)
                if nb >= (1 << 30) and free - nb > 8 * nbytes_m and budget - nb >= (want - len(cands)) * nbytes_m:
                    ballast.append(torch.empty(nb, dtype=torch.uint8, device=device))
                    budget -= nb
            cands.append(feat_empty(rows, d, device, zero=zero))
            budget -= nbytes_m
    except torch.cuda.OutOfMemoryError:
        # a device shared with other tenants (or nearly full): choose among the candidates that did fit
        del ballast[:]
        torch.cuda.empty_cache()
    try:
        if len(cands) <= 1:
            return fallback() if not cands else cands[0]
        if probe is None:
            scratch = feat_empty(n_dst, d, device)

            def probe(c):
                _lib.check(_spmm_call(indptr, indices, n_dst, rows, c, d, AGG_SAGE_GCN, None, None, None, None, False, scratch,
                                      c[:n_dst] if rows >= n_dst else c, None), "glnn_spmm_csr_f32 (placement probe)")
        ms = []
        for c in cands:
            probe(c)
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                probe(c)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1)
                best = t if best is None else min(best, t)
            ms.append(best)
    except torch.cuda.OutOfMemoryError:
        scratch = None
        return fallback()
    k = min(range(len(cands)), key=lambda i: ms[i])
    PLACEMENT_LOG.append({"what": what, "rows": int(rows), "d": int(d), "ms": [round(v, 3) for v in ms], "chosen": k})
    keep = cands[k]
    del cands, scratch, ballast
    torch.cuda.empty_cache()          # the losers and the ballast go back to the driver, not into the caching allocator's pool
    return keep


def place_for_gather(x, indptr, indices, n_dst, what="features"):
    """x itself, or a copy of it in a better-placed allocation (see placed_for_gather): for the static input matrix of a teacher forward."""
    x = as_feat(x)
    if not placement_applies(x.shape[0], x.shape[1]) or x.shape[0] < n_dst:
        return x
    best = placed_for_gather(x.shape[0], x.shape[1], x.device, indptr, indices, n_dst, what=what, first=x)
    if best is not x:
        _storage_copy(best, x)
    return best


def _storage_copy(dst, src):
    """dst[:, :] = src including the padding columns of the feature layout (both [rows, d] views of [rows, round4(d)] storage)."""
    d = src.shape[1]
    dst.copy_(src)
    if dst.stride(0) > d:                      # padding columns are zero by the layout's contract
        torch.as_strided(dst, (dst.shape[0], dst.stride(0) - d), (dst.stride(0), 1), dst.storage_offset() + d).zero_()


DEG_RAW, DEG_RSQRT_CLAMP1, DEG_INV_PLUS1 = 0, 1, 2
SELF_ROWS = True      # spmm(..., self_rows=) is available (dist.ShardedTeacher asks: the numpy test backend has no such argument)


def degrees(indptr, indices, n_dst, n_src, nnz, want_out=True, transform=DEG_RAW, want_in=True):
    """glnn_degrees_f32: (t(in_deg) [n_dst], t(out_deg) [n_src]) as fp32; transform: DEG_RAW | DEG_RSQRT_CLAMP1 | DEG_INV_PLUS1."""
    _need_cuda(indptr, indices)
    in_deg = torch.empty(n_dst, dtype=torch.float32, device=indptr.device) if want_in else None
    out_deg = torch.empty(n_src, dtype=torch.float32, device=indptr.device) if want_out else None
    rc = _lib.lib().glnn_degrees_f32(_p(indptr), _p(indices), n_dst, n_src, nnz, transform, _p(in_deg), _p(out_deg), _stream())
    _lib.check(rc, "glnn_degrees_f32")
    return in_deg, out_deg


_DEFAULT_WS = {}


def _default_ws(device):
    """One lazily allocated 64 MB split-K workspace per (device, stream) for callers that pass none: the partials of two GEMMs issued
    on different streams of one device must not share a buffer (reuse on ONE stream is ordered)."""
    key = (str(device), int(_stream().value or 0))
    if key not in _DEFAULT_WS:
        if len(_DEFAULT_WS) >= 8:
            _DEFAULT_WS.clear()                        # a pathological number of streams: start over rather than grow without bound
        _DEFAULT_WS[key] = torch.empty(1 << 24, dtype=torch.float32, device=device)
    return _DEFAULT_WS[key]


def gemm(a, w, w_is_kn=False, a_rows=None, a_scale=None, a_shift=None, row_scale=None, ep_scale=None,
         ep_shift=None, relu=False, out=None, m=None, drop_p=0.0, drop_seed=0, workspace=None):
    """K3 glnn_gemm_f32: out = epi(A' @ W^T) (w [n,k], torch Linear layout) or epi(A' @ W) (w [k,n])."""
    _need_cuda(a, w, a_rows, a_scale, a_shift, row_scale, ep_scale, ep_shift, out)
    _mat(a, "gemm a")
    _mat(w, "gemm w")
    k = a.shape[1]
    n = w.shape[1] if w_is_kn else w.shape[0]
    if (w.shape[0] if w_is_kn else w.shape[1]) != k:
        raise ValueError(f"gemm: inner dimensions differ: a {tuple(a.shape)} w {tuple(w.shape)} kn={w_is_kn}")
    if m is None:
        m = a_rows.numel() if a_rows is not None else a.shape[0]
    if a_rows is not None and (a_rows.dtype != torch.int64 or not a_rows.is_contiguous()):
        raise ValueError("gemm: a_rows must be contiguous int64")
    if out is None:
        out = feat_empty(m, n, a.device)
    _mat(out, "gemm out")
    if w.stride(0) % 4 and w.numel() <= (1 << 22) and (w_is_kn or k < 4 or n > 512 or m >= 1024):
        w = as_feat(w)                              # weights whose rows are not float4-addressable and that the unaligned-W latency kernel does not
                                                    # take, or takes badly ([k, n] with n % 4 != 0, [n, k] with k < 4, or thousands of rows of A):
                                                    # a padded copy per call (nn.Parameters are never cached by as_feat), then the tiled kernels
    if workspace is None and m * n <= (1 << 22) and k >= 2048:
        workspace = _default_ws(a.device)          # lets a deep, narrow product split its reduction (see gemm.hip)
    with _Timed("gemm", m=m, k=k, n=n):
        rc = _gemm_call(a, a_rows, a_scale, a_shift, drop_p, drop_seed, m, k, w, w_is_kn, n, row_scale, ep_scale, ep_shift, relu, out,
                        workspace)
    _lib.check(rc, "glnn_gemm_f32")
    return out


def _gemm_call(a, a_rows, a_scale, a_shift, drop_p, drop_seed, m, k, w, w_is_kn, n, row_scale, ep_scale, ep_shift, relu, out,
               workspace):
    return _lib.lib().glnn_gemm_f32(
        _p(a), _ld(a), _p(a_rows), _p(_vec(a_scale, k, "a_scale")), _p(_vec(a_shift, k, "a_shift")),
        float(drop_p), int(drop_seed) & 0xFFFFFFFF, m, k,
        _p(w), _ld(w), 1 if w_is_kn else 0, n, _p(_vec(row_scale, m, "row_scale")),
        _p(_vec(ep_scale, n, "ep_scale")), _p(_vec(ep_shift, n, "ep_shift")), 1 if relu else 0,
        _p(out), _ld(out), _p(workspace), workspace.numel() if workspace is not None else 0, _stream())


def gemm_tn(a, b, b_rows=None, b_scale=None, b_shift=None, out=None, col_sum_a=None, workspace=None, m=None,
            drop_p=0.0, drop_seed=0):
    """glnn_gemm_tn_f32: out[i,j] = sum_m a[m,i] * b'[m,j]  (weight gradient dW = dZ^T @ A_prev)."""
    _need_cuda(a, b, b_rows, b_scale, b_shift, out, col_sum_a, workspace)
    _mat(a, "gemm_tn a")
    _mat(b, "gemm_tn b")
    if m is None:
        m = a.shape[0]
    ka, nb = a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((ka, nb), dtype=torch.float32, device=a.device)
    if workspace is None:
        workspace = torch.empty(64 * ka + 256 * 128 * 128, dtype=torch.float32, device=a.device)
    rc = _lib.lib().glnn_gemm_tn_f32(
        _p(a), _ld(a), m, ka, _p(b), _ld(b), _p(b_rows), _p(_vec(b_scale, nb, "b_scale")),
        _p(_vec(b_shift, nb, "b_shift")), float(drop_p), int(drop_seed) & 0xFFFFFFFF, nb, _p(out), _ld(out), _p(col_sum_a), _p(workspace), workspace.numel(), _stream())
    _lib.check(rc, "glnn_gemm_tn_f32")
    return out


def softmax_loss(logits, kind, lamb, labels=None, label_rows=None, target_logp=None, target_rows=None,
                 dlogits=None, logprob_out=None, loss_out=None, loss_accum=None, workspace=None):
    """K4 glnn_softmax_loss_f32.  Returns (loss_out [1] device tensor, dlogits)."""
    _need_cuda(logits, labels, label_rows, target_logp, target_rows, dlogits, logprob_out, loss_out, loss_accum)
    _mat(logits, "softmax_loss logits")
    rows, c = logits.shape
    if dlogits is None:
        dlogits = torch.empty((rows, c), dtype=torch.float32, device=logits.device)
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=logits.device)
    if workspace is None:
        workspace = torch.empty(1024, dtype=torch.float32, device=logits.device)
    if labels is not None and labels.dtype != torch.int64:
        raise ValueError("softmax_loss: labels must be int64")
    rc = _lib.lib().glnn_softmax_loss_f32(
        _p(logits), _ld(logits), rows, c, kind, _p(labels), _p(label_rows),
        _p(target_logp), _ld(target_logp) if target_logp is not None else 0, _p(target_rows), float(lamb),
        _p(dlogits), _ld(dlogits), _p(logprob_out), _ld(logprob_out) if logprob_out is not None else 0,
        _p(loss_out), _p(loss_accum), _p(workspace), workspace.numel(), _stream())
    _lib.check(rc, "glnn_softmax_loss_f32")
    return loss_out, dlogits


def classifier_loss(a, w, bias, kind, lamb=1.0, a_scale=None, a_shift=None, drop_p=0.0, drop_seed=0, labels=None, label_rows=None,
                    target_logp=None, target_rows=None, logits=None, dlogits=None, loss_out=None, loss_accum=None, workspace=None):
    """glnn_classifier_loss_f32: logits = tail(a) . w^T + bias and, with kind >= 0, log_softmax + NLL / KL + dlogits behind it in the same
    launch (large batches in front of a narrow classifier; raises GlnnError(UNSUPPORTED) for other shapes).  kind = -1: logits only.
    Returns (logits, loss_out, dlogits)."""
    _need_cuda(a, w, bias, a_scale, a_shift, labels, label_rows, target_logp, target_rows, logits, dlogits, loss_out, loss_accum)
    _mat(a, "classifier_loss a")
    rows, k = a.shape
    c = w.shape[0]
    dev = a.device
    if logits is None:
        logits = torch.empty((rows, c), dtype=torch.float32, device=dev)
    if kind >= 0:
        if dlogits is None:
            dlogits = torch.empty((rows, c), dtype=torch.float32, device=dev)
        if loss_out is None:
            loss_out = torch.empty(1, dtype=torch.float32, device=dev)
        if workspace is None:
            workspace = torch.empty(1024, dtype=torch.float32, device=dev)
    rc = _lib.lib().glnn_classifier_loss_f32(
        _p(a), _ld(a), _p(a_scale), _p(a_shift), float(drop_p), int(drop_seed) & 0xFFFFFFFF, rows, k, _p(w), _ld(w), c, _p(bias),
        _p(logits), _ld(logits), kind, _p(labels), _p(label_rows), _p(target_logp), _ld(target_logp) if target_logp is not None else 0,
        _p(target_rows), float(lamb), _p(dlogits), _ld(dlogits) if dlogits is not None else 0, _p(loss_out), _p(loss_accum),
        _p(workspace), workspace.numel() if workspace is not None else 0, _stream())
    _lib.check(rc, "glnn_classifier_loss_f32")
    return logits, loss_out, dlogits


def log_softmax(logits, out=None):
    _need_cuda(logits, out)
    _mat(logits, "log_softmax logits")
    rows, c = logits.shape
    if out is None:
        out = torch.empty((rows, c), dtype=torch.float32, device=logits.device)
    rc = _lib.lib().glnn_log_softmax_f32(_p(logits), _ld(logits), rows, c, _p(out), _ld(out), _stream())
    _lib.check(rc, "glnn_log_softmax_f32")
    return out


def bn_stats(z, gamma, beta, running_mean, running_var, nbt, eps=1e-5, momentum=0.1, outs=None, workspace=None):
    """K5 glnn_bn_stats_f32.  Returns (mean, rstd, a_scale, a_shift)."""
    note_param_write()
    _need_cuda(z, gamma, beta, running_mean, running_var, nbt, workspace)
    _mat(z, "bn_stats z")
    rows, h = z.shape
    if outs is None:
        outs = tuple(torch.empty(h, dtype=torch.float32, device=z.device) for _ in range(4))
    mean, rstd, a_scale, a_shift = outs
    nchunks = (rows + 127) // 128
    need = 2 * nchunks * h + (3 * ((nchunks + 63) // 64) * h if nchunks > 256 else 0)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.float32, device=z.device)
    rc = _lib.lib().glnn_bn_stats_f32(_p(z), _ld(z), rows, h, _p(gamma), _p(beta), eps, momentum, _p(running_mean),
                                      _p(running_var), _p(nbt), _p(mean), _p(rstd), _p(a_scale), _p(a_shift),
                                      _p(workspace), workspace.numel(), _stream())
    _lib.check(rc, "glnn_bn_stats_f32")
    return mean, rstd, a_scale, a_shift


def linear_bn_stats(a, w, bias, gamma, beta, running_mean, running_var, nbt, eps=1e-5, momentum=0.1, out=None):
    """glnn_linear_bn_stats_f32: z = a w^T + bias and the BatchNorm1d training statistics of z, the statistics' first pass taken from
    the product kernel's epilogue where that kernel can leave it.  Returns (z, mean, rstd, a_scale, a_shift)."""
    note_param_write()
    _need_cuda(a, w, bias, gamma, beta, running_mean, running_var, nbt, out)
    a = as_feat(a)
    m, k = a.shape
    n = w.shape[0]
    if w.dim() != 2 or w.shape[1] < k or w.stride(1) != 1:
        raise ValueError("linear_bn_stats: w must be [n, k] row-major")
    z = feat_empty(m, n, a.device) if out is None else out
    mean, rstd, a_scale, a_shift = (torch.empty(n, dtype=torch.float32, device=a.device) for _ in range(4))
    nchunks = (m + 127) // 128
    ws_bn = torch.empty(max(2 * nchunks * n + (3 * ((nchunks + 63) // 64) * n if nchunks > 256 else 0), 3 * 128 * n), dtype=torch.float32, device=a.device)
    ws = _default_ws(a.device)
    rc = _lib.lib().glnn_linear_bn_stats_f32(_p(a), _ld(a), m, k, _p(w), w.stride(0), n, _p(bias), _p(z), _ld(z), _p(gamma), _p(beta), eps, momentum,
                                             _p(running_mean), _p(running_var), _p(nbt), _p(mean), _p(rstd), _p(a_scale), _p(a_shift),
                                             _p(ws), ws.numel(), _p(ws_bn), ws_bn.numel(), _stream())
    _lib.check(rc, "glnn_linear_bn_stats_f32")
    return z, mean, rstd, a_scale, a_shift


def bn_relu_bwd(da, z, gamma=None, mean=None, rstd=None, a_scale=None, a_shift=None, dz=None, dgamma=None,
                dbeta=None, workspace=None, drop_p=0.0, drop_seed=0, dz_col_sum=None, relu=True):
    """K5 glnn_bn_relu_bwd_f32 (relu=False: glnn_bn_bwd_f32, the norm -> dropout tail of GCN.forward).  Returns (dz, dgamma,
    dbeta); gamma=None => plain ReLU (+ dropout) backward.
    dz_col_sum: optional [h] output = column sums of dz (the bias gradient of the Linear in front)."""
    _need_cuda(da, z, gamma, mean, rstd, a_scale, a_shift, dz, dgamma, dbeta, workspace)
    _mat(da, "bn_relu_bwd da")
    _mat(z, "bn_relu_bwd z")
    rows, h = z.shape
    if dz is None:
        dz = torch.empty((rows, h), dtype=torch.float32, device=z.device)
    if gamma is not None:
        if dgamma is None:
            dgamma = torch.empty(h, dtype=torch.float32, device=z.device)
        if dbeta is None:
            dbeta = torch.empty(h, dtype=torch.float32, device=z.device)
    if workspace is None and (gamma is not None or dz_col_sum is not None):
        workspace = torch.empty((3 * ((rows + 127) // 128) + 2) * h, dtype=torch.float32, device=z.device)
    if relu:
        rc = _lib.lib().glnn_bn_relu_bwd_f32(_p(da), _ld(da), _p(z), _ld(z), rows, h, _p(gamma), _p(mean), _p(rstd),
                                             _p(a_scale), _p(a_shift), float(drop_p), int(drop_seed) & 0xFFFFFFFF,
                                             _p(dz), _ld(dz), _p(dgamma), _p(dbeta), _p(dz_col_sum),
                                             _p(workspace), workspace.numel() if workspace is not None else 0, _stream())
    else:
        rc = _lib.lib().glnn_bn_bwd_f32(_p(da), _ld(da), _p(z), _ld(z), rows, h, _p(gamma), _p(mean), _p(rstd),
                                        _p(a_scale), _p(a_shift), 0, float(drop_p), int(drop_seed) & 0xFFFFFFFF,
                                        _p(dz), _ld(dz), _p(dgamma), _p(dbeta), _p(dz_col_sum),
                                        _p(workspace), workspace.numel() if workspace is not None else 0, _stream())
    _lib.check(rc, "glnn_bn_relu_bwd_f32")
    return dz, dgamma, dbeta


def layernorm_fwd(z, gamma=None, beta=None, eps=1e-5, relu=True, drop_p=0.0, drop_seed=0, out=None, want_stats=True):
    """glnn_layernorm_fwd_f32: out = dropout(relu?(LayerNorm(z))); returns (out, mean_row, rstd_row) (stats None if not wanted)."""
    _need_cuda(z, gamma, beta, out)
    z = as_feat(z)
    rows, h = z.shape
    if out is None:
        out = feat_empty(rows, h, z.device)
    mean = torch.empty(rows, dtype=torch.float32, device=z.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=z.device) if want_stats else None
    rc = _lib.lib().glnn_layernorm_fwd_f32(_p(z), _ld(z), rows, h, _p(_vec(gamma, h, "gamma")), _p(_vec(beta, h, "beta")), float(eps),
                                           1 if relu else 0, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(out), _ld(out), _p(mean),
                                           _p(rstd), _stream())
    _lib.check(rc, "glnn_layernorm_fwd_f32")
    return out, mean, rstd


def layernorm_bwd(da, z, gamma, beta, mean, rstd, relu=True, drop_p=0.0, drop_seed=0, dz=None, dz_col_sum=None, want_param_grads=True):
    """glnn_layernorm_bwd_f32.  Returns (dz, dgamma, dbeta)."""
    _need_cuda(da, z, gamma, beta, mean, rstd, dz, dz_col_sum)
    da, z = as_feat(da), as_feat(z)
    rows, h = z.shape
    if dz is None:
        dz = feat_empty(rows, h, z.device)
    dgamma = torch.empty(h, dtype=torch.float32, device=z.device) if (want_param_grads and gamma is not None) else None
    dbeta = torch.empty(h, dtype=torch.float32, device=z.device) if dgamma is not None else None
    ws = None
    if dgamma is not None or dz_col_sum is not None:
        ws = torch.empty(_lib.lib().glnn_layernorm_bwd_workspace_floats(rows, h), dtype=torch.float32, device=z.device)
    rc = _lib.lib().glnn_layernorm_bwd_f32(_p(da), _ld(da), _p(z), _ld(z), rows, h, _p(_vec(gamma, h, "gamma")), _p(_vec(beta, h, "beta")),
                                           _p(mean), _p(rstd), 1 if relu else 0, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(dz), _ld(dz),
                                           _p(dgamma), _p(dbeta), _p(dz_col_sum), _p(ws), ws.numel() if ws is not None else 0, _stream())
    _lib.check(rc, "glnn_layernorm_bwd_f32")
    return dz, dgamma, dbeta


def col_sum(x, out=None):
    """glnn_col_sum_f32: column sums of a [rows, h] matrix (a bias gradient)."""
    _need_cuda(x, out)
    _mat(x, "col_sum x")
    rows, h = x.shape
    if out is None:
        out = torch.empty(h, dtype=torch.float32, device=x.device)
    ws = torch.empty(((rows + 127) // 128) * h, dtype=torch.float32, device=x.device)
    rc = _lib.lib().glnn_col_sum_f32(_p(x), _ld(x), rows, h, _p(out), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "glnn_col_sum_f32")
    return out


class TensorTable:
    """Device-side pointer table for the multi-tensor Adam (built once per optimiser)."""

    def __init__(self, params, grads, exp_avg, exp_avg_sq):
        dev = params[0].device
        for group in (params, grads, exp_avg, exp_avg_sq):
            for t in group:
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise ValueError("TensorTable: contiguous float32 CUDA tensors required")
        self.keep = (list(params), list(grads), list(exp_avg), list(exp_avg_sq))
        mk = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
        self.p, self.g, self.m, self.v = mk(params), mk(grads), mk(exp_avg), mk(exp_avg_sq)
        self.sizes = torch.tensor([t.numel() for t in params], dtype=torch.int64, device=dev)
        self.n = len(params)
        self.max_size = max(t.numel() for t in params)
        # glnn_adam_desc (the one-call train step): the same table + a HOST copy of the gradient pointers
        import ctypes
        self.g_host = (ctypes.c_void_p * self.n)(*[t.data_ptr() for t in grads])
        self.desc = _lib.AdamDesc()
        d = self.desc
        d.params, d.grads, d.exp_avg, d.exp_avg_sq, d.sizes = _p(self.p), _p(self.g), _p(self.m), _p(self.v), _p(self.sizes)
        d.grads_host = ctypes.cast(self.g_host, ctypes.c_void_p)
        d.num_tensors, d.max_size = self.n, self.max_size


def adam_step(table, lr, step, weight_decay=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    note_param_write()
    rc = _lib.lib().glnn_adam_step_f32(_p(table.p), _p(table.g), _p(table.m), _p(table.v), _p(table.sizes), table.n,
                                       table.max_size, lr, beta1, beta2, eps, weight_decay, step, _stream())
    _lib.check(rc, "glnn_adam_step_f32")


def act_fwd(z, a_scale=None, a_shift=None, drop_p=0.0, drop_seed=0, out=None, relu=True):
    """glnn_act_fwd_f32: out = dropout(relu(z * a_scale + a_shift)) (plain ReLU without a_scale/a_shift); relu=False:
    glnn_norm_drop_fwd_f32, the same without the ReLU (GCN's norm -> dropout tail)."""
    _need_cuda(z, a_scale, a_shift, out)
    _mat(z, "act_fwd z")
    rows, h = z.shape
    if out is None:
        out = feat_empty(rows, h, z.device)
    if relu:
        rc = _lib.lib().glnn_act_fwd_f32(_p(z), _ld(z), rows, h, _p(_vec(a_scale, h, "a_scale")), _p(_vec(a_shift, h, "a_shift")),
                                         float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(out), _ld(out), _stream())
    else:
        rc = _lib.lib().glnn_norm_drop_fwd_f32(_p(z), _ld(z), rows, h, _p(_vec(a_scale, h, "a_scale")), _p(_vec(a_shift, h, "a_shift")), 0,
                                               float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(out), _ld(out), _stream())
    _lib.check(rc, "glnn_act_fwd_f32")
    return out


def dropout_mask(rows, h, drop_p, drop_seed, device):
    mask = torch.empty((rows, h), dtype=torch.uint8, device=device)
    rc = _lib.lib().glnn_dropout_mask_u8(rows, h, float(drop_p), int(drop_seed) & 0xFFFFFFFF, _p(mask), _stream())
    _lib.check(rc, "glnn_dropout_mask_u8")
    return mask


def sample_neighbors(indptr, indices, seeds, fanout, rng_seed):
    """glnn_sample_neighbors: returns (src [n_seeds, fanout] int32, cnt [n_seeds] int32)."""
    _need_cuda(indptr, indices, seeds)
    n = seeds.numel()
    src = torch.empty((n, fanout), dtype=torch.int32, device=seeds.device)
    cnt = torch.empty(n, dtype=torch.int32, device=seeds.device)
    rc = _lib.lib().glnn_sample_neighbors(_p(indptr), _p(indices), _p(seeds.contiguous()), n, int(fanout), int(rng_seed) & 0xFFFFFFFF,
                                          _p(src), _p(cnt), _stream())
    _lib.check(rc, "glnn_sample_neighbors")
    return src, cnt


def _i32(n, device):
    return torch.empty(max(int(n), 1), dtype=torch.int32, device=device)[:int(n)]


def block_build(seeds, graph_indptr=None, graph_indices=None, smp_src=None, smp_cnt=None, nnz_cap=None, want_global=False, n_nodes=0,
                global_only=False):
    """glnn_block_build_ids: one 1-hop block over the destination nodes `seeds` (int64 device vector).  n_nodes: the size of the id
    universe (every id < n_nodes) when known -- lets wide blocks index their tables by the id itself; 0 = unknown.
    Sampled mode: smp_src [ns, fanout] / smp_cnt [ns] from sample_neighbors.  Full-neighbour mode: the graph CSR and
    nnz_cap (an upper bound of the block's edge count).  Returns (indptr [ns+1], indices [nnz] local ids,
    gindices [nnz] global ids or None, input_nodes [n_src], nnz, n_src); ONE host read-back (the two counts).
    global_only: the global-id block -- (indptr, None, gindices, None, nnz, None): no table, no relabelling (the outermost block of a
    training batch whose consumer gathers from the global feature matrix)."""
    _need_cuda(seeds, graph_indptr, graph_indices, smp_src, smp_cnt)
    dev = seeds.device
    if seeds.dtype != torch.int64 or not seeds.is_contiguous():
        raise ValueError("block_build: seeds must be a contiguous int64 vector")
    ns = seeds.numel()
    if smp_src is not None:
        fanout = smp_src.shape[1]
        if smp_src.dtype != torch.int32 or smp_cnt.dtype != torch.int32 or smp_src.shape[0] != ns or not smp_src.is_contiguous():
            raise ValueError("block_build: smp_src [ns, fanout] / smp_cnt [ns] must be contiguous int32")
        nnz_cap = ns * fanout
    else:
        fanout = 0
        if graph_indptr is None or graph_indices is None or nnz_cap is None:
            raise ValueError("block_build: full-neighbour mode needs the graph CSR and nnz_cap")
    nnz_cap = int(nnz_cap)
    want_global = want_global or global_only
    indptr = torch.empty(ns + 1, dtype=torch.int64, device=dev)
    indices = None if global_only else _i32(nnz_cap, dev)
    gindices = _i32(nnz_cap, dev) if want_global else None
    input_nodes = None if global_only else torch.empty(ns + nnz_cap, dtype=torch.int64, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    wsb = int(_lib.lib().glnn_block_workspace_bytes(ns, nnz_cap))
    ws = torch.empty((wsb + 7) // 8, dtype=torch.int64, device=dev)
    rc = _lib.lib().glnn_block_build_ids(_p(graph_indptr) if smp_src is None else None, _p(graph_indices) if smp_src is None else None,
                                         _p(seeds), ns, _p(smp_src), _p(smp_cnt), fanout, nnz_cap, _p(indptr), _p(indices), _p(gindices),
                                         _p(input_nodes), _p(counts), int(n_nodes), _p(ws), ws.numel() * 8, _stream())
    _lib.check(rc, "glnn_block_build_ids")
    nnz, n_src = (int(v) for v in counts.tolist())          # the only host sync of the block
    if nnz < 0:
        raise _lib.GlnnError(f"block_build: a seed or neighbour id lies outside [0, n_nodes = {int(n_nodes)})")
    if nnz > nnz_cap:
        raise _lib.GlnnError(f"block_build: the block has {nnz} edges, more than nnz_cap = {nnz_cap}")
    if global_only:
        return indptr, None, gindices[:nnz], None, nnz, None
    return indptr, indices[:nnz], (gindices[:nnz] if want_global else None), input_nodes[:n_src], nnz, n_src


def csr_transpose(indptr, indices, n_dst, n_src, nnz, add_self=False):
    """glnn_csr_transpose: (t_indptr [n_src+1], t_indices [nnz (+ n_dst)]) of the CSR-by-destination graph; rows sorted.
    add_self: one extra entry u <- u per destination u (the h_dst term of the SAGE-gcn aggregator's backward)."""
    _need_cuda(indptr, indices)
    dev = indptr.device
    nnz_out = int(nnz) + (int(n_dst) if add_self else 0)
    t_indptr = torch.empty(n_src + 1, dtype=torch.int64, device=dev)
    t_indices = _i32(nnz_out, dev)
    wsb = int(_lib.lib().glnn_csr_transpose_workspace_bytes(n_src, nnz_out))
    ws = torch.empty((wsb + 7) // 8, dtype=torch.int64, device=dev)
    rc = _lib.lib().glnn_csr_transpose(_p(indptr), _p(indices) if nnz else None, n_dst, n_src, int(nnz), 1 if add_self else 0,
                                       _p(t_indptr), _p(t_indices) if nnz_out else None, _p(ws), ws.numel() * 8, _stream())
    _lib.check(rc, "glnn_csr_transpose")
    return t_indptr, t_indices


def gather_rows(x, rows, out=None):
    _need_cuda(x, rows, out)
    x = as_feat(x)
    d = x.shape[1]
    if rows.dtype != torch.int64:
        raise ValueError("gather_rows: rows must be int64")
    if out is None:
        out = feat_empty(rows.numel(), d, x.device)
    rc = _lib.lib().glnn_gather_rows_f32(_p(x), _ld(x), _p(rows), rows.numel(), d, _p(out), _ld(out), _stream())
    _lib.check(rc, "glnn_gather_rows_f32")
    return out


def scatter_rows(x, rows, out):
    _need_cuda(x, rows, out)
    x = as_feat(x)
    d = x.shape[1]
    rc = _lib.lib().glnn_scatter_rows_f32(_p(x), _ld(x), _p(rows), rows.numel(), d, _p(out), _ld(out), _stream())
    _lib.check(rc, "glnn_scatter_rows_f32")
    return out
