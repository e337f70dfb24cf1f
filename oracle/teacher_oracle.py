"""CPU oracle for the GLNN teacher forward (SAGE-"gcn" / GraphConv / feature_prop).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.

PARITY STATUS: "parity unpinned" by the reference itself -- the arithmetic is in
the absent third-party dgl==0.6.1 and the reference has no tests.
Pinned instead by known answers + scipy.sparse + torch.sparse_csr in
tests/test_oracle_teacher.py.  The COMPOSITION below (sweep, BN/ReLU order, raw last
layer) is pinned by the reference's own models.py run with stubbed dgl layers:
tests/golden/teacher_composition.npz (made by tests/golden/make_teacher_golden.py).

The per-op arithmetic is the C restatement in oracle/glnn_oracle.c; this file
only composes it the way the reference's Python does:
  * sage_inference  <-  SAGE.inference,  reference models.py:121-148
  * gcn_forward     <-  GCN.forward,     reference models.py:189-199
  * feature_prop    <-  utils.feature_prop, reference utils.py:171-189
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    so = os.path.join(_HERE, "libglnn_oracle.so")
    src = os.path.join(_HERE, "glnn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libglnn_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_max_threads.restype = ctypes.c_int
    return _LIB


def max_threads():
    return int(lib().oracle_max_threads())


def _f(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _c32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _csr(indptr, indices):
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    return indptr, indices


def spmm_sum(indptr, indices, x, row_scale=None, col_scale=None, n_dst=None, threads=1):
    """out[v] = row_scale[v] * sum_{u->v} col_scale[u] * x[u]   (utils.py:185)."""
    indptr, indices = _csr(indptr, indices)
    x = _c32(x)
    n_dst = len(indptr) - 1 if n_dst is None else n_dst
    d = x.shape[1]
    out = np.empty((n_dst, d), dtype=np.float32)
    rs = None if row_scale is None else _c32(row_scale)
    cs = None if col_scale is None else _c32(col_scale)
    lib().oracle_spmm_sum_f32(
        indptr.ctypes.data_as(_i64p), indices.ctypes.data_as(_i32p), ctypes.c_int64(n_dst),
        _f(x), ctypes.c_int64(x.strides[0] // 4), ctypes.c_int(d), _f(rs), _f(cs),
        _f(out), ctypes.c_int64(d), ctypes.c_int(threads))
    return out


def sage_gcn_agg(indptr, indices, x, n_dst=None, threads=1):
    """(sum_{u->v} x[u] + x[v]) / (in_deg(v)+1); dst rows are the first n_dst src rows
    (models.py:109,137: h_dst = h[:block.num_dst_nodes()])."""
    indptr, indices = _csr(indptr, indices)
    x = _c32(x)
    n_dst = len(indptr) - 1 if n_dst is None else n_dst
    d = x.shape[1]
    out = np.empty((n_dst, d), dtype=np.float32)
    lib().oracle_sage_gcn_agg_f32(
        indptr.ctypes.data_as(_i64p), indices.ctypes.data_as(_i32p), ctypes.c_int64(n_dst),
        _f(x), ctypes.c_int64(x.strides[0] // 4), ctypes.c_int(d),
        _f(x), ctypes.c_int64(x.strides[0] // 4), _f(out), ctypes.c_int64(d), ctypes.c_int(threads))
    return out


def linear(x, w, b=None, w_is_in_by_out=False, threads=1):
    """x @ w.T + b (torch Linear layout [out,in]) or x @ w + b (GraphConv layout [in,out])."""
    x = _c32(x)
    w = _c32(w)
    b = None if b is None else _c32(b)
    m, k = x.shape
    n_out = w.shape[1] if w_is_in_by_out else w.shape[0]
    y = np.empty((m, n_out), dtype=np.float32)
    lib().oracle_linear_f32(_f(x), ctypes.c_int64(k), ctypes.c_int64(m), ctypes.c_int(k), _f(w),
                            ctypes.c_int64(w.shape[1]), ctypes.c_int(n_out),
                            ctypes.c_int(1 if w_is_in_by_out else 0), _f(b), _f(y),
                            ctypes.c_int64(n_out), ctypes.c_int(threads))
    return y


def bn_eval_relu_(x, bn=None, relu=True, eps=1e-5, threads=1):
    """In-place BatchNorm1d(eval) [bn = dict(weight,bias,running_mean,running_var)] + ReLU."""
    assert x.dtype == np.float32 and x.flags.c_contiguous
    m, d = x.shape
    if bn is None:
        mean = var = gamma = beta = None
    else:
        mean, var = _c32(bn["running_mean"]), _c32(bn["running_var"])
        gamma, beta = _c32(bn["weight"]), _c32(bn["bias"])
    lib().oracle_bn_eval_relu_f32(_f(x), ctypes.c_int64(d), ctypes.c_int64(m), ctypes.c_int(d),
                                  _f(mean), _f(var), _f(gamma), _f(beta), ctypes.c_float(eps),
                                  ctypes.c_int(1 if relu else 0), ctypes.c_int(threads))
    return x


def log_softmax_(x, threads=1):
    assert x.dtype == np.float32 and x.flags.c_contiguous
    m, c = x.shape
    lib().oracle_log_softmax_f32(_f(x), ctypes.c_int64(c), ctypes.c_int64(m), ctypes.c_int(c),
                                 ctypes.c_int(threads))
    return x


def degrees(indptr, indices, n_src=None):
    indptr, indices = _csr(indptr, indices)
    n_dst = len(indptr) - 1
    n_src = n_dst if n_src is None else n_src
    in_deg = np.empty(n_dst, np.float32)
    out_deg = np.empty(n_src, np.float32)
    lib().oracle_degrees(indptr.ctypes.data_as(_i64p), indices.ctypes.data_as(_i32p),
                         ctypes.c_int64(n_dst), ctypes.c_int64(n_src), _f(in_deg), _f(out_deg))
    return in_deg, out_deg


# ----------------------------------------------------------------------------------------------
# Compositions that follow the reference's Python
# ----------------------------------------------------------------------------------------------
def sage_conv_gcn(indptr, indices, h, w_neigh, b_neigh, n_dst=None, threads=1):
    """dgl 0.6.1 SAGEConv(in,out,'gcn')(block,(h,h_dst)) as called at models.py:112,138."""
    agg = sage_gcn_agg(indptr, indices, h, n_dst=n_dst, threads=threads)
    return linear(agg, w_neigh, b_neigh, threads=threads)


def sage_inference(indptr, indices, feats, layers, norms, batch_size=None, threads=1):
    """SAGE.inference (models.py:121-148): layer-wise full-neighbour sweep.

    layers: list of dict(weight [out,in], bias [out]) -- fc_neigh of each SAGEConv.
    norms : list (len L-1) of BN dicts or None entries (norm_type 'none').
    batch_size: if given, dst nodes are processed in chunks of `batch_size` in node-id
      order exactly like dataloader_eval (train_and_eval.py:193-202); each chunk's block has
      the chunk's dst nodes first, then the remaining unique sources (order irrelevant to the
      arithmetic).  None = whole graph at once; the result is identical row for row.
    Returns raw logits [N, C] (the caller applies log_softmax, train_and_eval.py:98).
    """
    indptr, indices = _csr(indptr, indices)
    n = len(indptr) - 1
    x = _c32(feats)
    num_layers = len(layers)
    for l, lay in enumerate(layers):
        d_out = lay["weight"].shape[0]
        y = np.zeros((n, d_out), np.float32)                     # models.py:129-132
        chunks = [(0, n)] if batch_size is None else [(s, min(n, s + batch_size)) for s in range(0, n, batch_size)]
        for s, e in chunks:
            if batch_size is None:
                h = sage_conv_gcn(indptr, indices, x, lay["weight"], lay["bias"], threads=threads)
            else:
                # build the 1-hop full-neighbour block of dst nodes [s,e)        (:134-137)
                lo, hi = indptr[s], indptr[e]
                src = indices[lo:hi]
                out_nodes = np.arange(s, e, dtype=np.int64)
                extra = np.setdiff1d(np.unique(src), out_nodes)
                input_nodes = np.concatenate([out_nodes, extra])
                remap = np.full(n, -1, np.int64)
                remap[input_nodes] = np.arange(len(input_nodes))
                b_indptr = (indptr[s:e + 1] - lo).astype(np.int64)
                b_indices = remap[src].astype(np.int32)
                h = sage_conv_gcn(b_indptr, b_indices, x[input_nodes], lay["weight"], lay["bias"],
                                  n_dst=e - s, threads=threads)
            if l != num_layers - 1:                               # :139-143 (dropout = no-op in eval)
                bn = norms[l] if norms else None
                bn_eval_relu_(h, bn, relu=True, threads=threads)
            y[s:e] = h                                            # :145
        x = y                                                     # :147
    return x


def graph_conv_both(indptr, indices, h, weight, bias, relu, threads=1):
    """dgl 0.6.1 GraphConv(in,out,norm='both',activation) on a square graph (models.py:193)."""
    in_deg, out_deg = degrees(indptr, indices)
    cs = np.power(np.maximum(out_deg, 1.0), -0.5).astype(np.float32)
    rs = np.power(np.maximum(in_deg, 1.0), -0.5).astype(np.float32)
    d_in, d_out = weight.shape
    if d_in > d_out:   # mult W first to reduce the feature size for aggregation
        hw = linear(_c32(h) * cs[:, None], weight, None, w_is_in_by_out=True, threads=threads)
        rst = spmm_sum(indptr, indices, hw, threads=threads)
    else:
        agg = spmm_sum(indptr, indices, _c32(h) * cs[:, None], threads=threads)
        rst = linear(agg, weight, None, w_is_in_by_out=True, threads=threads)
    rst = rst * rs[:, None]
    if bias is not None:
        rst = rst + bias[None, :]
    rst = np.ascontiguousarray(rst, np.float32)
    if relu:
        np.maximum(rst, 0.0, out=rst)
    return rst


def gcn_forward(indptr, indices, feats, layers, norms=None, threads=1):
    """GCN.forward in eval mode (models.py:189-199): relu inside every conv but the last."""
    h = _c32(feats)
    num_layers = len(layers)
    for l, lay in enumerate(layers):
        h = graph_conv_both(indptr, indices, h, lay["weight"], lay["bias"],
                            relu=(l != num_layers - 1), threads=threads)
        if l != num_layers - 1 and norms and norms[l] is not None:
            bn_eval_relu_(h, norms[l], relu=False, threads=threads)
    return h


def feature_prop(indptr, indices, feats, k, threads=1):
    """utils.feature_prop (utils.py:171-189): (D^-1/2 A D^-1/2)^k X with D = in-degree.clamp(1)."""
    in_deg, _ = degrees(indptr, indices)
    norm = np.power(np.maximum(in_deg, 1.0), -0.5).astype(np.float32)
    x = _c32(feats)
    for _ in range(k):
        x = x * norm[:, None]
        x = spmm_sum(indptr, indices, x, threads=threads)
        x = x * norm[:, None]
    return np.ascontiguousarray(x, np.float32)


def min_cut_loss(indptr, indices, logp):
    """compute_min_cut_loss (reference utils.py:159-168): tr(S^T A S) / tr(S^T D S), S = exp(out), D = diag(in-degree);
    the dense N x N adjacency of the reference replaced by sum(S * (A S)) (the trace of a product is orientation-free)."""
    s = np.exp(np.asarray(logp, dtype=np.float64))
    a_s = spmm_sum(indptr, indices, s.astype(np.float32)).astype(np.float64)
    deg = np.diff(np.asarray(indptr)).astype(np.float64)
    return float((s * a_s).sum() / (deg[:, None] * s * s).sum())
