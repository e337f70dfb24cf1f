"""CPU oracle for the teacher TRAINING step: sampled-block GraphSAGE (`train_sage`) and
full-graph GCN (`train`), forward + NLL loss + hand-written backward + torch-style Adam, numpy/scipy fp32.

TEST INFRASTRUCTURE ONLY -- imported by tests/ and __graft_entry__.smoke(); never by the product package.

PARITY STATUS: composition pinned, dgl layer arithmetic unpinned.  tests/golden/teacher_training.npz holds what the
reference's own train_sage / train (reference train_and_eval.py:12-56) and SAGE.forward / GCN.forward (models.py:101-119,
189-199) produced over fixed blocks with differentiable stand-ins of the two dgl 0.6.1 layers
(tests/golden/make_teacher_train_golden.py); tests/test_oracle_teacher.py checks this file against it.  dgl itself is
absent from the build container, so the formulas of SAGEConv('gcn') / GraphConv(norm='both') are restated from dgl
0.6.1's published source (oracle/DGL_SEMANTICS.md).

  * sage_forward / sage_backward   <- SAGE.forward over blocks            reference models.py:101-119
  * gcn_forward / gcn_backward     <- GCN.forward                         reference models.py:189-199
  * train_sage / train             <- the loops                           reference train_and_eval.py:12-56
  * BatchNorm1d(train), NLL on log_softmax, Adam: shared with oracle/student_oracle.py
"""
import numpy as np
import scipy.sparse as sp

from . import student_oracle as so

F32 = np.float32


def _adj(indptr, indices, n_src):
    n_dst = len(indptr) - 1
    return sp.csr_matrix((np.ones(len(indices), F32), np.asarray(indices, np.int64), np.asarray(indptr, np.int64)), shape=(n_dst, n_src))


class TeacherState:
    """Parameters keyed like the reference's state_dict; `kind` 'sage': encoder.layers.{i}.fc_neigh.{weight [out,in], bias};
    'gcn': encoder.layers.{i}.{weight [in,out], bias}; encoder.norms.{i}.* for norm_type 'batch'.  params() is
    model.parameters() order (all layers, then all norms), which is the order of the Adam state."""

    def __init__(self, state_dict, kind, num_layers, norm_type="none"):
        self.kind, self.L, self.norm_type = kind, num_layers, norm_type
        pre = "fc_neigh." if kind == "sage" else ""
        self.W = [np.array(state_dict[f"encoder.layers.{i}.{pre}weight"], F32) for i in range(num_layers)]
        self.b = [np.array(state_dict[f"encoder.layers.{i}.{pre}bias"], F32) for i in range(num_layers)]
        self.gamma, self.beta, self.rm, self.rv, self.nbt = [], [], [], [], []
        if norm_type == "batch":
            for i in range(num_layers - 1):
                self.gamma.append(np.array(state_dict[f"encoder.norms.{i}.weight"], F32))
                self.beta.append(np.array(state_dict[f"encoder.norms.{i}.bias"], F32))
                self.rm.append(np.array(state_dict[f"encoder.norms.{i}.running_mean"], F32))
                self.rv.append(np.array(state_dict[f"encoder.norms.{i}.running_var"], F32))
                self.nbt.append(int(state_dict[f"encoder.norms.{i}.num_batches_tracked"]))
        self.t = 0
        self.m = [np.zeros_like(p) for p in self.params()]
        self.v = [np.zeros_like(p) for p in self.params()]

    def params(self):
        ps = []
        for i in range(self.L):
            ps += [self.W[i], self.b[i]]
        for i in range(len(self.gamma)):
            ps += [self.gamma[i], self.beta[i]]
        return ps

    def state_dict(self):
        pre = "fc_neigh." if self.kind == "sage" else ""
        sd = {}
        for i in range(self.L):
            sd[f"encoder.layers.{i}.{pre}weight"], sd[f"encoder.layers.{i}.{pre}bias"] = self.W[i].copy(), self.b[i].copy()
        for i in range(len(self.gamma)):
            sd[f"encoder.norms.{i}.weight"], sd[f"encoder.norms.{i}.bias"] = self.gamma[i].copy(), self.beta[i].copy()
            sd[f"encoder.norms.{i}.running_mean"], sd[f"encoder.norms.{i}.running_var"] = self.rm[i].copy(), self.rv[i].copy()
            sd[f"encoder.norms.{i}.num_batches_tracked"] = np.int64(self.nbt[i])
        return sd


def _bn_train(st, l, z):
    B = z.shape[0]
    mean = z.mean(axis=0, dtype=np.float64).astype(F32)
    var = z.astype(np.float64).var(axis=0).astype(F32)
    st.rm[l] = ((1 - so.BN_MOMENTUM) * st.rm[l] + so.BN_MOMENTUM * mean).astype(F32)
    st.rv[l] = ((1 - so.BN_MOMENTUM) * st.rv[l] + so.BN_MOMENTUM * (var * F32(B / (B - 1)) if B > 1 else var)).astype(F32)
    st.nbt[l] += 1
    rstd = (1.0 / np.sqrt(var + so.BN_EPS)).astype(F32)
    xhat = ((z - mean) * rstd).astype(F32)
    return xhat, rstd, (xhat * st.gamma[l] + st.beta[l]).astype(F32)


def _tail_fwd(st, l, z, relu, keep, p, cache):
    """norm -> (relu) -> dropout of a hidden layer (models.py:113-117 / :195-198)."""
    if st.norm_type == "batch":
        xhat, rstd, y = _bn_train(st, l, z)
    else:
        xhat, rstd, y = None, None, z
    a = np.maximum(y, 0) if relu else y
    if keep is not None:
        a = (a * keep * F32(1.0 / (1.0 - p))).astype(F32)
    cache.append(dict(xhat=xhat, rstd=rstd, y=y, keep=keep, relu=relu))
    return a.astype(F32)


def _tail_bwd(st, l, da, c, p):
    if c["keep"] is not None:
        da = (da * c["keep"] * F32(1.0 / (1.0 - p))).astype(F32)
    dy = (da * (c["y"] > 0)).astype(F32) if c["relu"] else da
    if st.norm_type != "batch":
        return dy, None, None
    B = dy.shape[0]
    s1 = dy.sum(axis=0, dtype=np.float64)
    s2 = (dy.astype(np.float64) * c["xhat"]).sum(axis=0)
    dz = (st.gamma[l] * c["rstd"] * (dy - s1 / B - c["xhat"] * (s2 / B))).astype(F32)
    return dz, s2.astype(F32), s1.astype(F32)


# ------------------------------------------------------------------------------------------------ GraphSAGE on blocks
def sage_forward(st, blocks, x, masks=None, p=0.0):
    """blocks: list of (indptr, indices, n_src), outermost first; x = feats[input_nodes].  Returns (logits, cache)."""
    h = np.asarray(x, F32)
    cache = dict(h=[], agg=[], A=[], tails=[])
    for l, (ip, ix, n_src) in enumerate(blocks):
        A = _adj(ip, ix, n_src)
        n_dst = A.shape[0]
        deg1 = (np.diff(ip).astype(F32) + 1)[:, None]
        agg = ((A @ h + h[:n_dst]) / deg1).astype(F32)                    # (sum_{u->v} h[u] + h_dst[v]) / (deg + 1)
        z = (agg @ st.W[l].T + st.b[l]).astype(F32)
        cache["h"].append(h); cache["agg"].append(agg); cache["A"].append((A, deg1))
        if l != st.L - 1:
            h = _tail_fwd(st, l, z, True, None if masks is None else masks[l], p, cache["tails"])
        else:
            h = z
    return h, cache


def sage_backward(st, cache, dlogits, p=0.0):
    gW, gb = [None] * st.L, [None] * st.L
    gg, gbeta = [None] * len(st.gamma), [None] * len(st.gamma)
    dz = np.asarray(dlogits, F32)
    for l in range(st.L - 1, -1, -1):
        gW[l] = (dz.T @ cache["agg"][l]).astype(F32)
        gb[l] = dz.sum(axis=0, dtype=np.float64).astype(F32)
        if l == 0:
            break
        A, deg1 = cache["A"][l]
        dagg = (dz @ st.W[l] / deg1).astype(F32)
        dh = (A.T @ dagg).astype(F32)                                    # A^T (dAgg / (deg+1)) over the block's sources
        dh[: A.shape[0]] += dagg                                         # the self term h_dst = h[:n_dst]
        dz, g2, g1 = _tail_bwd(st, l - 1, dh, cache["tails"][l - 1], p)
        if g2 is not None:
            gg[l - 1], gbeta[l - 1] = g2, g1
    grads = []
    for i in range(st.L):
        grads += [gW[i], gb[i]]
    for i in range(len(st.gamma)):
        grads += [gg[i], gbeta[i]]
    return grads


def train_sage(st, batches, feats, labels, lr, weight_decay, lamb=1.0):
    """reference train_and_eval.py:32-56.  batches: list of (input_nodes, output_nodes, blocks)."""
    losses = []
    for input_nodes, output_nodes, blocks in batches:
        logits, cache = sage_forward(st, blocks, feats[input_nodes])
        loss, dlogits = so.loss_and_dlogits(logits, labels[output_nodes], "nll", lamb)
        losses.append(float(loss))
        so.adam_step(st, sage_backward(st, cache, dlogits), lr, weight_decay)
    return float(np.sum(losses) / len(batches)), losses


# ------------------------------------------------------------------------------------------------ full-graph GCN
def gcn_forward(st, indptr, indices, x, masks=None, p=0.0):
    """GCN.forward (models.py:189-199) with dgl GraphConv(norm='both', activation=relu on all but the last layer):
    h*outdeg^-1/2 -> (weight first iff in > out) -> sum over in-edges -> *indeg^-1/2 + bias -> relu | then norm, dropout."""
    n = len(indptr) - 1
    A = _adj(indptr, indices, n)
    cs = np.maximum(np.bincount(indices, minlength=n), 1).astype(F32) ** F32(-0.5)
    rs = np.maximum(np.diff(indptr), 1).astype(F32) ** F32(-0.5)
    h = np.asarray(x, F32)
    cache = dict(A=A, cs=cs[:, None], rs=rs[:, None], h=[], mid=[], tails=[], first=[])
    for l in range(st.L):
        hs = (h * cache["cs"]).astype(F32)
        first = st.W[l].shape[0] > st.W[l].shape[1]
        if first:
            mid = hs                                                     # operand of the weight GEMM
            rst = (A @ (hs @ st.W[l])).astype(F32)
        else:
            mid = (A @ hs).astype(F32)
            rst = (mid @ st.W[l]).astype(F32)
        z = (rst * cache["rs"] + st.b[l]).astype(F32)
        cache["h"].append(h); cache["mid"].append(mid); cache["first"].append(first)
        if l != st.L - 1:
            zr = np.maximum(z, 0)                                        # activation INSIDE the conv
            cache["tails"].append(dict(pre=z))
            h = _tail_fwd(st, l, zr, False, None if masks is None else masks[l], p, cache["tails"][-1].setdefault("t", []))
        else:
            h = z
    return h, cache


def gcn_backward(st, cache, dlogits, p=0.0):
    A = cache["A"]
    gW, gb = [None] * st.L, [None] * st.L
    gg, gbeta = [None] * len(st.gamma), [None] * len(st.gamma)
    dz = np.asarray(dlogits, F32)
    for l in range(st.L - 1, -1, -1):
        gb[l] = dz.sum(axis=0, dtype=np.float64).astype(F32)
        drst = (dz * cache["rs"]).astype(F32)
        if cache["first"][l]:
            dhw = (A.T @ drst).astype(F32)
            gW[l] = (cache["mid"][l].T @ dhw).astype(F32)
            dhs = (dhw @ st.W[l].T).astype(F32)
        else:
            gW[l] = (cache["mid"][l].T @ drst).astype(F32)
            dhs = (A.T @ (drst @ st.W[l].T)).astype(F32)
        if l == 0:
            break
        dh = (dhs * cache["cs"]).astype(F32)
        t = cache["tails"][l - 1]
        dzr, g2, g1 = _tail_bwd(st, l - 1, dh, t["t"][0], p)
        if g2 is not None:
            gg[l - 1], gbeta[l - 1] = g2, g1
        dz = (dzr * (t["pre"] > 0)).astype(F32)
    grads = []
    for i in range(st.L):
        grads += [gW[i], gb[i]]
    for i in range(len(st.gamma)):
        grads += [gg[i], gbeta[i]]
    return grads


def train(st, indptr, indices, feats, labels, idx_train, lr, weight_decay, lamb=1.0):
    """reference train_and_eval.py:12-29: one full-graph step, loss on idx_train."""
    logits, cache = gcn_forward(st, indptr, indices, feats)
    loss, dl = so.loss_and_dlogits(logits[idx_train], labels[idx_train], "nll", lamb)
    dlogits = np.zeros_like(logits)
    np.add.at(dlogits, idx_train, dl)
    so.adam_step(st, gcn_backward(st, cache, dlogits), lr, weight_decay)
    return float(loss)
