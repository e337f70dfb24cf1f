"""GPU parity of every C-ABI kernel against the CPU oracle (bar: 1e-4 abs fp32, north star).
All calls go through glnn_amd.ops -> ctypes -> libglnn_hip.so."""
import numpy as np
import pytest
import torch

from graphgen import csr_from_edges, random_graph
from oracle import student_oracle as so
from oracle import teacher_oracle as to

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def g2d(indptr, indices):
    return dev(indptr), dev(indices)


# ------------------------------------------------------------------------------------------- K1/K2
@pytest.mark.parametrize("d", [1, 7, 16, 40, 47, 64, 100, 128, 256, 300])
def test_spmm_sage_gcn_vs_oracle(d):
    from glnn_amd import ops
    n = 3000
    indptr, indices = random_graph(n, 12, seed=d, power=0.6, isolated=7, hub=2500)   # hub > 512 -> long-row role
    x = np.random.RandomState(d).standard_normal((n, d)).astype(np.float32)
    want = to.sage_gcn_agg(indptr, indices, x)
    ip, ix = g2d(indptr, indices)
    got = ops.spmm(ip, ix, dev(x), n, ops.AGG_SAGE_GCN)
    assert got.shape == (n, d)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)
    # padding columns of the output buffer are written as zero
    if got.stride(0) > d:
        base = torch.as_strided(got, (n, got.stride(0)), (got.stride(0), 1))
        assert float(base[:, d:].abs().max()) == 0.0


def test_spmm_known_answers_on_gpu():
    from glnn_amd import ops
    src = np.array([0, 2, 2, 3, 1]); dst = np.array([1, 1, 1, 3, 0])
    indptr, indices = csr_from_edges(src, dst, 5)
    x = np.array([[1, 10], [2, 20], [3, 30], [4, 40], [5, 50]], np.float32)
    got = ops.spmm(dev(indptr), dev(indices), dev(x), 5, ops.AGG_SAGE_GCN).cpu().numpy()
    want = np.array([[1.5, 15], [2.25, 22.5], [3, 30], [4, 40], [5, 50]], np.float32)
    np.testing.assert_array_equal(got, want)


def test_spmm_block_ndst_lt_nsrc():
    from glnn_amd import ops
    indptr = np.array([0, 3, 4], np.int64); indices = np.array([2, 3, 1, 3], np.int32)
    x = np.array([[1.0], [2.0], [4.0], [8.0]], np.float32)
    got = ops.spmm(dev(indptr), dev(indices), dev(x), 2, ops.AGG_SAGE_GCN).cpu().numpy()
    np.testing.assert_array_equal(got, np.array([[15 / 4], [5.0]], np.float32))


@pytest.mark.parametrize("d,use_rs,use_cs", [(64, True, True), (7, True, False), (128, False, True), (33, False, False),
                                              (700, True, True), (3703, False, False)])      # > 256 columns: one launch, blockIdx.y = column tile
def test_spmm_sum_scaled_vs_oracle(d, use_rs, use_cs):
    from glnn_amd import ops
    n = 2500
    indptr, indices = random_graph(n, 9, seed=100 + d, power=0.5, symmetric=True, self_loops=True, hub=900)
    rs_ = np.random.RandomState(5)
    x = rs_.standard_normal((n, d)).astype(np.float32)
    rs = rs_.uniform(0.1, 1, n).astype(np.float32) if use_rs else None
    cs = rs_.uniform(0.1, 1, n).astype(np.float32) if use_cs else None
    want = to.spmm_sum(indptr, indices, x, rs, cs)
    got = ops.spmm(dev(indptr), dev(indices), dev(x), n, ops.AGG_SUM,
                   row_scale=None if rs is None else dev(rs), col_scale=None if cs is None else dev(cs))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("d", [47, 601])
def test_spmm_epilogue_scale_shift_relu(d):
    from glnn_amd import ops
    n = 1200
    indptr, indices = random_graph(n, 20, seed=3, power=0.5, hub=700)
    r = np.random.RandomState(1)
    x = r.standard_normal((n, d)).astype(np.float32)
    sc, sh = r.uniform(0.5, 1.5, d).astype(np.float32), r.standard_normal(d).astype(np.float32)
    want = np.maximum(to.sage_gcn_agg(indptr, indices, x) * sc + sh, 0)
    got = ops.spmm(dev(indptr), dev(indices), dev(x), n, ops.AGG_SAGE_GCN, ep_scale=dev(sc), ep_shift=dev(sh), relu=True)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)


def test_spmm_empty_rows_and_wide_features():
    from glnn_amd import ops
    n, d = 300, 1433          # raw cora feature width: 6 column tiles
    indptr, indices = random_graph(n, 3, seed=9, isolated=100)
    x = np.random.RandomState(2).standard_normal((n, d)).astype(np.float32)
    got = ops.spmm(dev(indptr), dev(indices), dev(x), n, ops.AGG_SUM)
    np.testing.assert_allclose(got.cpu().numpy(), to.spmm_sum(indptr, indices, x), atol=TOL, rtol=0)


@pytest.mark.parametrize("n,d_in,d_out", [(3000, 100, 256), (1001, 128, 256), (777, 256, 256), (500, 20, 32), (333, 64, 40), (64, 7, 9)])
def test_sage_fused_vs_oracle(n, d_in, d_out):
    """K1F: aggregation + projection + scale/shift/ReLU in one launch, incl. a hub row (> 512 in-edges ->
    cooperative pass), isolated rows, a row count that is not a multiple of the 32-row tile, odd d_in/d_out."""
    from glnn_amd import ops
    indptr, indices = random_graph(n, 10, seed=n, power=0.6, isolated=5, hub=700 if n > 800 else 0)
    r = np.random.RandomState(n)
    x = r.standard_normal((n, d_in)).astype(np.float32)
    w = (r.standard_normal((d_out, d_in)) / np.sqrt(d_in)).astype(np.float32)
    sc, sh = r.uniform(.5, 1.5, d_out).astype(np.float32), r.standard_normal(d_out).astype(np.float32)
    want = np.maximum(to.linear(to.sage_gcn_agg(indptr, indices, x), w) * sc + sh, 0)
    got = ops.sage_fused(dev(indptr), dev(indices), dev(x), n, dev(w), ep_scale=dev(sc), ep_shift=dev(sh), relu=True)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)
    plain = ops.sage_fused(dev(indptr), dev(indices), dev(x), n, dev(w))
    np.testing.assert_allclose(plain.cpu().numpy(), to.linear(to.sage_gcn_agg(indptr, indices, x), w), atol=TOL, rtol=0)


@pytest.mark.parametrize("n,d_in,d_out,d2", [(3000, 256, 256, 47), (1001, 128, 256, 40), (500, 20, 32, 5), (333, 100, 200, 64), (70, 7, 9, 3)])
def test_sage_fused_chained_projection_vs_oracle(n, d_in, d_out, d2):
    """K1F with the NEXT layer's projection chained behind its epilogue (hidden rows never leave the workgroup):
    out2 = relu(bn(agg @ W^T)) @ W2^T, with and without also writing the hidden rows."""
    from glnn_amd import ops
    indptr, indices = random_graph(n, 8, seed=d_in + d2, power=0.6, isolated=3, hub=500)
    r = np.random.RandomState(d_out)
    x = r.standard_normal((n, d_in)).astype(np.float32)
    w = (r.standard_normal((d_out, d_in)) / np.sqrt(d_in)).astype(np.float32)
    w2 = (r.standard_normal((d2, d_out)) / np.sqrt(d_out)).astype(np.float32)
    sc, sh = r.uniform(.5, 1.5, d_out).astype(np.float32), r.uniform(-.5, .5, d_out).astype(np.float32)
    agg = to.sage_gcn_agg(indptr, indices, x)
    hid = np.maximum(to.linear(agg, w) * sc + sh, 0)
    want2 = to.linear(hid, w2)
    h, p2 = ops.sage_fused(dev(indptr), dev(indices), dev(x), n, dev(w), ep_scale=dev(sc), ep_shift=dev(sh), relu=True, w_next=dev(w2))
    np.testing.assert_allclose(h.cpu().numpy(), hid, atol=TOL, rtol=1e-5)
    np.testing.assert_allclose(p2.cpu().numpy(), want2, atol=TOL, rtol=1e-5)
    none, p3 = ops.sage_fused(dev(indptr), dev(indices), dev(x), n, dev(w), ep_scale=dev(sc), ep_shift=dev(sh), relu=True, w_next=dev(w2),
                              want_out=False)
    assert none is None and torch.equal(p3, p2)


def test_sage_fused_heaviest_tile_first_order_changes_nothing_but_the_schedule():
    """glnn_sage_fused_f32(tile_order): workgroup i takes tile tile_order[i] -- the heaviest-row-first permutation
    (ops.fused_tile_order, CSRGraph.fused_tile_order) starts a power-law graph's hub rows first (round 4: the D=256 layer of the products
    forward 18.55 -> 18.25 ms, a rank-of-8 chunk launch -7 %).  Every tile's arithmetic is untouched: identical bits, with and without
    the chained projection, ragged last tile; a permutation of the wrong length is refused."""
    from glnn_amd import ops
    n, d_in, d_out, d2 = 5003, 100, 256, 47
    indptr, indices = random_graph(n, 12, seed=4, power=0.8, isolated=5, hub=4000)
    rs = np.random.RandomState(4)
    x = dev(rs.standard_normal((n, d_in)).astype(np.float32))
    w = dev((rs.standard_normal((d_out, d_in)) / 10).astype(np.float32))
    w2 = dev((rs.standard_normal((d2, d_out)) / 16).astype(np.float32))
    b = dev(rs.standard_normal(d_out).astype(np.float32))
    ip, ix = g2d(indptr, indices)
    order = ops.fused_tile_order(ip, n)
    assert order.dtype == torch.int32 and sorted(order.tolist()) == list(range((n + 31) // 32))
    deg = np.diff(indptr)
    assert deg[order[0].item() * 32:(order[0].item() + 1) * 32].max() == deg.max()           # the hub's tile goes first
    plain = ops.sage_fused(ip, ix, x, n, w, ep_shift=b, relu=True)
    assert torch.equal(ops.sage_fused(ip, ix, x, n, w, ep_shift=b, relu=True, tile_order=order), plain)
    o1, p1 = ops.sage_fused(ip, ix, x, n, w, ep_shift=b, relu=True, w_next=w2)
    o2, p2 = ops.sage_fused(ip, ix, x, n, w, ep_shift=b, relu=True, w_next=w2, tile_order=order)
    assert torch.equal(o1, o2) and torch.equal(p1, p2)
    with pytest.raises(ValueError):
        ops.sage_fused(ip, ix, x, n, w, tile_order=order[:-1].contiguous())


@pytest.mark.parametrize("chain", [True, False])
def test_sage_fused_one_launch_over_chunks_equals_the_chunk_launches_and_signals_every_chunk(chain):
    """glnn_sage_fused_chunks_f32 (round 6): the chunks of a row range as tile ranges of ONE launch -- self rows and outputs of chunk c at
    the chunk's own rows of whole buffers (the chunk-major slots of the sharded forward), heaviest-first tile order per chunk -- gives
    the bits of one launch per chunk; and every chunk SIGNALS: a second stream held by glnn_stream_wait_value32 copies chunk c's rows out
    while the launch may still be running, and the copy already holds the final rows (stores written back before the signal).  Two
    launches in a row (epochs 1, 2: the arrival counters come back to zero); an empty chunk is never signalled; bad descriptors are
    refused."""
    from glnn_amd import ops
    n, d_in, d_out, d2 = 21003, 256, 256, 47
    indptr, indices = random_graph(n, 14, seed=9, power=0.8, isolated=7, hub=6000)
    rs = np.random.RandomState(9)
    ip, ix = g2d(indptr, indices)
    x = dev(rs.standard_normal((n, d_in)).astype(np.float32))
    w = dev((rs.standard_normal((d_out, d_in)) / 16).astype(np.float32))
    w2 = dev((rs.standard_normal((d2, d_out)) / 16).astype(np.float32)) if chain else None
    b = dev(rs.standard_normal(d_out).astype(np.float32))
    row_start = [0, 6400, 6400 + 7168, 6400 + 7168 + 8000, 6400 + 7168 + 8000 + 64]      # 4 chunks, the last one EMPTY (behind n)
    assert row_start[3] >= n
    slots = [40000, 10016, 25000, 60000]                      # where the chunks' rows live in the whole buffers (out of order, apart)
    d_o = d2 if chain else d_out
    xs = torch.zeros(70000, d_in, device="cuda")             # self rows: chunk c's rows at slots[c]
    want = torch.zeros(70000, d_o, device="cuda")
    orders = []
    for c in range(3):
        r0, r1 = row_start[c], min(row_start[c + 1], n)
        xs[slots[c]:slots[c] + r1 - r0] = x[r0:r1]
        o = ops.fused_tile_order(ip[r0:r1 + 1], r1 - r0)
        orders.append(o + r0 // 32)
        kw = dict(ep_shift=b, relu=True, x_self=x[r0:r1], tile_order=o)
        if chain:
            ops.sage_fused(ip[r0:r1 + 1], ix, x, r1 - r0, w, w_next=w2, out_next=want[slots[c]:slots[c] + r1 - r0], want_out=False, **kw)
        else:
            ops.sage_fused(ip[r0:r1 + 1], ix, x, r1 - r0, w, out=want[slots[c]:slots[c] + r1 - r0], **kw)
    order = torch.cat(orders).to(torch.int32).contiguous()
    sig = ops.ChunkSignals(row_start, x.device)
    side = torch.cuda.Stream()
    for epoch in (1, 2):
        got = torch.zeros(70000, d_o, device="cuda")
        early = torch.zeros(70000, d_o, device="cuda")
        torch.cuda.synchronize()
        desc = sig.launch(slots, slots, n)
        assert sig.epoch == epoch and sig.empty(3) and not sig.empty(2)
        if chain:
            ops.sage_fused(ip, ix, x, n, w, ep_shift=b, relu=True, x_self=xs, w_next=w2, out_next=got, want_out=False, tile_order=order, chunks=desc)
        else:
            ops.sage_fused(ip, ix, x, n, w, ep_shift=b, relu=True, x_self=xs, out=got, tile_order=order, chunks=desc)
        for c in range(3):
            sig.wait(side, c)
            with torch.cuda.stream(side):
                nr = min(row_start[c + 1], n) - row_start[c]
                early[slots[c]:slots[c] + nr].copy_(got[slots[c]:slots[c] + nr])
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert torch.equal(early, want)
        assert [sig.value(c) for c in range(4)] == [epoch, epoch, epoch, 0]
        assert int(sig.arrivals.abs().sum()) == 0
    with pytest.raises(ValueError):
        ops.ChunkSignals([0, 100, 200], x.device)            # not a multiple of the 32-row tile
    with pytest.raises(ValueError):
        ops.sage_fused(ip, ix, x, n, w, chunks=sig.launch(slots, slots, n))      # the whole buffers must be given


@pytest.mark.parametrize("d,with_plan", [(100, False), (100, True), (47, False), (256, True)])
def test_spmm_one_launch_over_chunks_equals_the_chunk_launches_and_signals_every_chunk(d, with_plan):
    """glnn_spmm_csr_chunks_f32 (round 6): the stand-alone SAGE aggregation over the chunks of a row range in ONE launch -- row workgroups
    that never straddle a chunk, the long-row role walking the chunks in order, outputs (and self rows) at the chunks' rows of whole
    buffers -- gives the bits of one launch per chunk, with and without a hub plan; a second stream held by the chunk's signal copies the
    chunk out while the launch may still run.  Two epochs; an empty chunk is never signalled."""
    from glnn_amd import ops
    n = 30011
    indptr, indices = random_graph(n, 16, seed=d, power=0.9, isolated=9, hub=7000)
    assert (np.diff(indptr) > 128).sum() > 20                      # long rows in every chunk, one hub row
    rs = np.random.RandomState(d)
    ip, ix = g2d(indptr, indices)
    x = dev(rs.standard_normal((n, d)).astype(np.float32))
    row_start = [0, 9600, 9600 + 10112, 9600 + 10112 + 12000, 9600 + 10112 + 12000 + 32]     # the last chunk is empty (behind n)
    assert row_start[3] >= n
    out_slots, self_slots = [50000, 1024, 20000, 70000], [30016, 64, 50048, 90000]
    xs = ops.feat_empty(100000, d, x.device, zero=True)
    want = ops.feat_empty(100000, d, x.device, zero=True)
    for c in range(3):
        r0, r1 = row_start[c], min(row_start[c + 1], n)
        xs[self_slots[c]:self_slots[c] + r1 - r0] = x[r0:r1]
        hub = ops.hub_plan(ip[r0:r1 + 1], r1 - r0) if with_plan else None
        ops.spmm(ip[r0:r1 + 1], ix, x, r1 - r0, ops.AGG_SAGE_GCN, out=want[out_slots[c]:out_slots[c] + r1 - r0], x_self=x[r0:r1],
                 **({"hub": hub} if hub is not None else {}))
    hub = ops.hub_plan(ip, n) if with_plan else None
    assert not with_plan or hub is not None
    sig = ops.ChunkSignals(row_start, x.device)
    side = torch.cuda.Stream()
    for epoch in (1, 2):
        got = ops.feat_empty(100000, d, x.device, zero=True)
        early = ops.feat_empty(100000, d, x.device, zero=True)
        torch.cuda.synchronize()
        ops.spmm(ip, ix, x, n, ops.AGG_SAGE_GCN, out=got, x_self=xs, chunks=sig.launch(self_slots, out_slots, n), **({"hub": hub} if hub is not None else {}))
        for c in range(3):
            sig.wait(side, c)
            with torch.cuda.stream(side):
                nr = min(row_start[c + 1], n) - row_start[c]
                early[out_slots[c]:out_slots[c] + nr].copy_(got[out_slots[c]:out_slots[c] + nr])
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert torch.equal(early, want)
        assert [sig.value(c) for c in range(4)] == [epoch, epoch, epoch, 0]
        assert int(sig.arrivals.abs().sum()) == 0
    with pytest.raises(ValueError):
        ops.spmm(ip, ix, x, n, ops.AGG_SUM, out=got, chunks=sig.launch(self_slots, out_slots, n))


def test_degrees():
    from glnn_amd import ops
    n = 1000
    indptr, indices = random_graph(n, 6, seed=4, power=0.7, isolated=11)
    i_want, o_want = to.degrees(indptr, indices)
    i_got, o_got = ops.degrees(dev(indptr), dev(indices), n, n, int(indptr[-1]))
    np.testing.assert_array_equal(i_got.cpu().numpy(), i_want)
    np.testing.assert_array_equal(o_got.cpu().numpy(), o_want)


# ------------------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("m,k,n,kn", [(1000, 100, 256, False), (517, 256, 47, False), (130, 1433, 64, True),
                                      (64, 7, 7, True), (2048, 128, 256, False), (300, 33, 130, False),
                                      (4096, 2048, 47, False), (1, 5, 3, False)])
def test_gemm_vs_oracle(m, k, n, kn):
    from glnn_amd import ops
    r = np.random.RandomState(m + k + n)
    a = r.standard_normal((m, k)).astype(np.float32)
    w = (r.standard_normal((k, n) if kn else (n, k)) / np.sqrt(k)).astype(np.float32)
    want = to.linear(a, w, None, w_is_in_by_out=kn)
    got = ops.gemm(ops.as_feat(dev(a)), dev(w), w_is_kn=kn)
    assert got.shape == (m, n)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("form,m,k,n", [("nt", 4096, 2048, 2048), ("nt", 300, 64, 200), ("nt", 129, 32, 129), ("nt", 1000, 96, 256),
                                        ("kn", 4096, 2048, 2048), ("kn", 300, 64, 200), ("kn", 517, 160, 132), ("kn", 128, 32, 68),
                                        ("tn", 4096, 2048, 2048), ("tn", 320, 200, 130), ("tn", 8192, 256, 256), ("tn", 64, 129, 257),
                                        ("tn", 2080, 130, 100), ("tn", 2077, 256, 256), ("tn", 60001, 256, 128), ("tn", 4096, 2048, 100),
                                        ("tn", 2049, 129, 65), ("tn", 2064, 128, 128), ("tn", 2144, 130, 100), ("tn", 4128, 256, 256),
                                        ("nt", 300, 96, 200), ("kn", 200, 32, 132), ("kn", 2100, 160, 256), ("nt", 5000, 224, 384),
                                        ("tn", 256, 130, 100), ("tn", 200, 256, 256), ("tn", 96, 2048, 300), ("nt", 700, 256, 260),
                                        ("nt", 4096, 4096, 2048), ("kn", 1000, 1024, 512), ("nt", 260, 288, 130), ("nt", 200, 544, 256)])
def test_pipelined_gemm_kernels_equal_the_compiler_scheduled_ones_bit_for_bit(form, m, k, n, monkeypatch):
    """gemm_kernel_pipe / gemm_tn_kernel_pipe (hand-scheduled main loop, buffer loads, plain operands, K % 32 == 0) keep the tiles,
    the k order and the MFMA order of gemm_kernel_fast / gemm_tn_kernel_t: same bits, ragged tiles, epilogues and splits included --
    for reductions of up to 256 terms.  Longer ones are accumulated in blocks of 256 k by the pipelined kernels (round 4: one MFMA
    chain over K = 2048 - 4096 carried 2.2 - 3.1 x the rounding noise of the oracle's blocked sgemm): there the two paths must agree
    to rounding, both must sit within the fp64 product's tolerance, and from 1024 terms on the pipelined result must be the closer
    one by a clear margin (rms error <= 0.75 x).  For "tn" (m, k, n) = (reduction rows, ka, nb)."""
    from glnn_amd import ops
    r = np.random.RandomState(m + k + n)
    outs = []
    if form == "tn":
        a = ops.as_feat(dev((r.standard_normal((m, k)) / 8).astype(np.float32)))
        b = ops.as_feat(dev(r.standard_normal((m, n)).astype(np.float32)))
        want = a[:, :k].double().t() @ b[:, :n].double()
    else:
        a = ops.as_feat(dev(r.standard_normal((m, k)).astype(np.float32)))
        w = dev((r.standard_normal((k, n) if form == "kn" else (n, k)) / np.sqrt(k)).astype(np.float32))
        if form == "kn":
            w = ops.as_feat(w)
        rs, es, eh = (dev(r.uniform(0.5, 2, m).astype(np.float32)), dev(r.uniform(0.5, 1.5, n).astype(np.float32)),
                      dev(r.standard_normal(n).astype(np.float32)))
        ww = w[:, :n] if form == "kn" else w.t()
        want = torch.relu((a[:, :k].double() @ ww.double()) * rs.double()[:, None] * es.double() + eh.double())
    for mode in ("1", "0"):
        monkeypatch.setenv("GLNN_GEMM_PIPE", mode)
        if form == "tn":
            outs.append(ops.gemm_tn(a, b).clone())
        else:
            outs.append(ops.gemm(a, w, w_is_kn=(form == "kn"), row_scale=rs, ep_scale=es, ep_shift=eh, relu=True).clone())
    # the weight-gradient launcher picks other tiles / reduction splits for few-tile outputs over >= 2048 rows when the pipelined
    # kernel is available: a different summation order across splits, so only closeness can be asked there
    red = m if form == "tn" else k              # terms per output element
    same_order = red <= 256 and not (form == "tn" and m >= 2048 and ((k + 127) // 128) * ((n + 127) // 128) <= 64)
    if same_order:
        assert torch.equal(outs[0], outs[1])
    else:
        np.testing.assert_allclose(outs[1][:, :want.shape[1]].cpu().numpy(), want.cpu().numpy(), atol=TOL * max(1.0, red / 512) ** 0.5, rtol=1e-5)
    np.testing.assert_allclose(outs[0][:, :want.shape[1]].cpu().numpy(), want.cpu().numpy(), atol=TOL * max(1.0, red / 512) ** 0.5, rtol=1e-5)
    unsplit = not (form == "tn" and ((k + 127) // 128) * ((n + 127) // 128) <= 64)
    if red >= 1024 and unsplit:      # (split reductions are blocked by their splits on both paths)
        full = want.shape[0] * want.shape[1] >= 2048 * 2048          # enough tiles that the launcher does not split K: the pipelined kernel runs
        if full or not torch.equal(outs[0], outs[1]):
            e = [float((o[:, :want.shape[1]].double() - want).pow(2).mean().sqrt()) for o in outs]
            assert e[0] <= 0.75 * e[1], e


@pytest.mark.parametrize("m,k,n,epi", [(5000, 100, 256, True), (4096, 100, 2048, True), (100003, 128, 256, False), (2049, 36, 96, True),
                                         (70001, 64, 300, True), (2048, 124, 130, False), (9000, 40, 128, True)])
def test_rowpanel_gemm_equals_the_tiled_kernels_bit_for_bit(m, k, n, epi, monkeypatch):
    """K3r (csrc/gemm_rowpanel.hip): short reductions over many rows -- persistent workgroups, the weight panel resident in LDS, row
    tiles walked with a cross-tile software pipeline, all global traffic through per-tile buffer descriptors.  Same k order as the
    tiled kernels -> identical bits; ragged last row tile, column panels past n, k tails inside the last k-group (k % 8 == 4), a
    padded output (ldc > n) whose padding must stay untouched; and closeness to the fp64 product."""
    from glnn_amd import ops
    r = np.random.RandomState(m + k + n)
    a = ops.as_feat(dev(r.standard_normal((m, k)).astype(np.float32)))
    w = dev((r.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    es = dev(r.uniform(0.5, 1.5, n).astype(np.float32)) if epi else None
    eh = dev(r.standard_normal(n).astype(np.float32)) if epi else None
    outs = []
    for mode in ("1", "2", "0"):       # the wave-walk kernel (round 5) | the workgroup-tile kernel (round 4) | the tiled kernels
        monkeypatch.setenv("GLNN_GEMM_ROWPANEL", mode)
        out = ops.feat_empty(m, n, DEV)
        base = torch.as_strided(out, (m, out.stride(0)), (out.stride(0), 1))
        base.fill_(-7.0)
        ops.gemm(a, w, ep_scale=es, ep_shift=eh, relu=epi, out=out)
        outs.append(base.clone())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])
    if outs[0].shape[1] > n:
        assert float((outs[0][:, n:] + 7.0).abs().max()) == 0.0          # padding columns: not written by either path
    want = a[:, :k].double() @ w.double().t()
    if epi:
        want = torch.relu(want * es.double() + eh.double())
    np.testing.assert_allclose(outs[0][:, :n].cpu().numpy(), want.cpu().numpy(), atol=TOL, rtol=1e-5)


def test_aggregation_of_a_row_shard_keeps_the_hub_rows_busy():
    """A row SHARD of a power-law graph keeps the graph's hub rows: the long-row role of spmm_csr_kernel gets one workgroup per 512-row
    scan chunk (round 4; it was n_dst / 4096, which left 18 workgroups with a fifth of the edges of a 76 k-row products shard).  Same
    sums as before -- deterministic LDS fold -- so a shard's rows equal the whole-graph launch bit for bit, for several shard sizes."""
    from glnn_amd import ops
    n, d = 40000, 100
    indptr, indices = random_graph(n, 20, seed=3, power=0.8, isolated=5, hub=9000)
    x = dev(np.random.RandomState(3).standard_normal((n, d)).astype(np.float32))
    ip, ix = g2d(indptr, indices)
    full = ops.spmm(ip, ix, x, n, ops.AGG_SAGE_GCN)
    np.testing.assert_allclose(full.cpu().numpy(), to.sage_gcn_agg(indptr, indices, x.cpu().numpy()), atol=TOL, rtol=0)
    for lo, hi in ((0, 513), (1000, 6000), (20000, 40000), (39000, 40000)):
        part = ops.spmm(ip[lo:hi + 1], ix, x, hi - lo, ops.AGG_SAGE_GCN, x_self=x[lo:hi])
        assert torch.equal(part, full[lo:hi])


@pytest.mark.parametrize("d", [47, 100, 256])
def test_hub_rows_split_over_workgroups_give_the_same_bits(d):
    """ABI 9 (round 5): rows of more than glnn_hub_row_threshold() in-edges are summed segment by segment; with an ops.HubPlan the
    segments are gathered by one workgroup each in a launch in front of the aggregation (the tail of a shard's short launches), without
    one by the row's own workgroup -- the same numbers bit for bit: stand-alone aggregation (SAGE-gcn, and SUM with both scales), the
    fused aggregate + project kernel with its chained projection, whole graph and row range, and all of them against the oracle.  The
    graph holds several hub rows: one of ~9000 edges, a few of 1100-3000 (one exactly at the threshold: not a hub), ragged last segments."""
    from glnn_amd import _lib, ops
    thr, seg = _lib.lib().glnn_hub_row_threshold(), _lib.lib().glnn_hub_segment_edges()
    n = 30000
    rs = np.random.RandomState(d)
    indptr0, indices0 = random_graph(n, 12, seed=d, power=0.7, isolated=3, hub=9000)
    deg0 = np.diff(indptr0)
    dst = np.repeat(np.arange(n), deg0)
    extra_rows = rs.choice(n, 6, replace=False)
    extra_deg = [thr + 1, thr + seg - 1, 2 * seg + 77, 3000, thr, 1500]          # (thr: stays with its owner)
    src_x = [rs.randint(0, n, size=max(0, k - int(deg0[r]))) for r, k in zip(extra_rows, extra_deg)]
    dst_x = [np.full(len(sx), r) for r, sx in zip(extra_rows, src_x)]
    indptr, indices = csr_from_edges(np.concatenate([indices0] + src_x), np.concatenate([dst] + dst_x), n)
    deg = np.diff(indptr)
    assert (deg > thr).sum() >= 5 and deg.max() >= 9000
    ip, ix = g2d(indptr, indices)
    x = dev(rs.standard_normal((n, d)).astype(np.float32))
    plan = ops.hub_plan(ip, n)
    assert plan.n_hub == int((deg > thr).sum()) and plan.n_seg == int(((deg[deg > thr] + seg - 1) // seg).sum())
    assert plan.rows.tolist() == np.flatnonzero(deg > thr).tolist()
    # stand-alone SAGE-gcn aggregation
    a0 = ops.spmm(ip, ix, x, n, ops.AGG_SAGE_GCN)
    a1 = ops.spmm(ip, ix, x, n, ops.AGG_SAGE_GCN, hub=plan)
    assert torch.equal(a0, a1)
    np.testing.assert_allclose(a1[:, :d].cpu().numpy(), to.sage_gcn_agg(indptr, indices, x.cpu().numpy()), atol=TOL, rtol=0)
    # a row range holding the big hub: its own plan, same rows
    hub_row = int(np.argmax(deg))
    lo, hi = max(0, hub_row - 700), min(n, hub_row + 900)
    sub = ops.hub_plan(ip[lo:hi + 1], hi - lo)
    part = ops.spmm(ip[lo:hi + 1], ix, x, hi - lo, ops.AGG_SAGE_GCN, x_self=x[lo:hi], hub=sub)
    assert sub is not None and torch.equal(part, a0[lo:hi])
    # SUM with row and column scales (GraphConv norm = both)
    rsc, csc = dev(rs.uniform(.5, 1.5, n).astype(np.float32)), dev(rs.uniform(.5, 1.5, n).astype(np.float32))
    s0 = ops.spmm(ip, ix, x, n, ops.AGG_SUM, row_scale=rsc, col_scale=csc)
    s1 = ops.spmm(ip, ix, x, n, ops.AGG_SUM, row_scale=rsc, col_scale=csc, hub=plan)
    assert torch.equal(s0, s1)
    # fused aggregate + project (+ chained projection): plan == no plan, and close to aggregate-then-project in fp64
    w = dev((rs.standard_normal((64, d)) / np.sqrt(d)).astype(np.float32))
    w2 = dev((rs.standard_normal((10, 64)) / 8).astype(np.float32))
    es, eh = dev(rs.uniform(.5, 1.5, 64).astype(np.float32)), dev(rs.standard_normal(64).astype(np.float32))
    f0, c0 = ops.sage_fused(ip, ix, x, n, w, ep_scale=es, ep_shift=eh, relu=True, w_next=w2)
    f1, c1 = ops.sage_fused(ip, ix, x, n, w, ep_scale=es, ep_shift=eh, relu=True, w_next=w2, hub=plan, tile_order=ops.fused_tile_order(ip, n))
    assert torch.equal(f0, f1) and torch.equal(c0, c1)
    want = torch.relu(a0[:, :d].double() @ w.double().t() * es.double() + eh.double())
    np.testing.assert_allclose(f1[:, :64].cpu().numpy(), want.cpu().numpy(), atol=TOL, rtol=1e-5)
    np.testing.assert_allclose(c1[:, :10].cpu().numpy(), (want @ w2.double().t()).cpu().numpy(), atol=TOL, rtol=1e-5)
    # a graph without hub rows has no plan
    ip2, _ = g2d(*random_graph(2000, 5, seed=1))
    assert ops.hub_plan(ip2, 2000) is None


def test_gemm_transpose_detecting():
    # A = I (padded) against an ASYMMETRIC B catches swapped C layouts
    from glnn_amd import ops
    k = n = 96
    a = np.eye(k, dtype=np.float32)
    w = np.arange(n * k, dtype=np.float32).reshape(n, k) / 100.0
    got = ops.gemm(dev(a), dev(w)).cpu().numpy()
    np.testing.assert_allclose(got, w.T, atol=1e-6, rtol=0)


def test_gemm_full_epilogue_and_operand_transform():
    from glnn_amd import ops
    r = np.random.RandomState(11)
    nrows, m, k, n = 5000, 1500, 100, 200
    x = r.standard_normal((nrows, k)).astype(np.float32)
    rows = r.randint(0, nrows, m).astype(np.int64)
    a_sc, a_sh = r.uniform(0.5, 1.5, k).astype(np.float32), r.standard_normal(k).astype(np.float32) * 0.3
    w = (r.standard_normal((n, k)) / 10).astype(np.float32)
    rs = r.uniform(0.5, 2, m).astype(np.float32)
    e_sc, e_sh = r.uniform(0.5, 1.5, n).astype(np.float32), r.standard_normal(n).astype(np.float32)
    ap = np.maximum(x[rows] * a_sc + a_sh, 0)
    want = np.maximum((to.linear(ap, w) * rs[:, None]) * e_sc + e_sh, 0)
    got = ops.gemm(dev(x), dev(w), a_rows=dev(rows), a_scale=dev(a_sc), a_shift=dev(a_sh), row_scale=dev(rs),
                   ep_scale=dev(e_sc), ep_shift=dev(e_sh), relu=True)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("m,ka,nb", [(512, 40, 256), (4096, 47, 300), (700, 256, 128), (333, 130, 100), (4096, 256, 256)])
def test_gemm_tn_vs_numpy(m, ka, nb):
    from glnn_amd import ops
    r = np.random.RandomState(m + ka)
    a = (r.standard_normal((m, ka)) / 8).astype(np.float32)
    b = r.standard_normal((m, nb)).astype(np.float32)
    want = (a.astype(np.float64).T @ b.astype(np.float64))
    colsum = torch.empty(ka, device=DEV)
    got = ops.gemm_tn(ops.as_feat(dev(a)), ops.as_feat(dev(b)), col_sum_a=colsum)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=1e-5)
    np.testing.assert_allclose(colsum.cpu().numpy(), a.astype(np.float64).sum(0), atol=TOL, rtol=1e-5)


def test_gemm_tn_gather_and_transform():
    from glnn_amd import ops
    r = np.random.RandomState(8)
    nrows, m, ka, nb = 3000, 1024, 64, 100
    dz = (r.standard_normal((m, ka)) / 8).astype(np.float32)
    x = r.standard_normal((nrows, nb)).astype(np.float32)
    rows = r.randint(0, nrows, m).astype(np.int64)
    sc, sh = r.uniform(0.5, 1.5, nb).astype(np.float32), r.standard_normal(nb).astype(np.float32) * .2
    want = dz.astype(np.float64).T @ np.maximum(x[rows] * sc + sh, 0).astype(np.float64)
    got = ops.gemm_tn(dev(dz), dev(x), b_rows=dev(rows), b_scale=dev(sc), b_shift=dev(sh), m=m)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("m,ka,nb,rows_on", [(1000, 64, 48, True), (777, 130, 200, True), (2048, 256, 256, False), (333, 40, 60, False)])
def test_gemm_tn_dropout_transform_all_instantiations(m, ka, nb, rows_on):
    """TN weight gradient with the full operand transform (BN affine + ReLU + dropout) with and without gathered source
    rows, for both column-tile widths: dW = dz^T @ (mask * relu(x[rows]*sc+sh) / (1-p)), the mask being the kernel's own
    counter-based one (glnn_dropout_mask_u8), rows m not a multiple of the 32-row k-tile."""
    from glnn_amd import ops
    r = np.random.RandomState(m + nb)
    nrows = 2 * m
    dz = (r.standard_normal((m, ka)) / 8).astype(np.float32)
    x = r.standard_normal((nrows if rows_on else m, nb)).astype(np.float32)
    rows = r.randint(0, nrows, m).astype(np.int64) if rows_on else None
    sc, sh = r.uniform(0.5, 1.5, nb).astype(np.float32), r.standard_normal(nb).astype(np.float32) * .2
    p, seed = 0.35, 4242
    mask = ops.dropout_mask(m, nb, p, seed, DEV).cpu().numpy().astype(np.float64)
    xs = x[rows] if rows_on else x
    act = np.maximum(xs.astype(np.float64) * sc + sh, 0) * mask / (1 - p)
    want = dz.astype(np.float64).T @ act
    got = ops.gemm_tn(ops.as_feat(dev(dz)), ops.as_feat(dev(x)), b_rows=dev(rows) if rows_on else None, b_scale=dev(sc),
                      b_shift=dev(sh), m=m, drop_p=p, drop_seed=seed)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=1e-5)
    assert 0.55 < mask.mean() < 0.75


@pytest.mark.parametrize("rows,h,bn,p", [(1000, 47, True, 0.3), (37, 256, True, 0.0), (4096, 100, False, 0.5), (1, 5, False, 0.0)])
def test_act_fwd_and_its_backward_pair(rows, h, bn, p):
    """glnn_act_fwd_f32 = dropout(relu(z*scale+shift)) with the kernels' counter-based mask, and glnn_bn_relu_bwd_f32 as its
    backward (same seed): checked against float64 numpy through the mask glnn_dropout_mask_u8 reports."""
    from glnn_amd import ops
    r = np.random.RandomState(rows + h)
    z = r.standard_normal((rows, h)).astype(np.float32)
    seed = 99
    mask = ops.dropout_mask(rows, h, p, seed, DEV).cpu().numpy().astype(np.float64) if p > 0 else np.ones((rows, h))
    zd = ops.as_feat(dev(z))
    if bn:
        gamma, beta = r.uniform(.5, 1.5, h).astype(np.float32), (r.standard_normal(h) * .2).astype(np.float32)
        rm, rv, nbt = torch.zeros(h, device=DEV), torch.ones(h, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        mean, rstd, sc, sh = ops.bn_stats(zd, dev(gamma), dev(beta), rm, rv, nbt)
        z64 = z.astype(np.float64)
        mu, var = z64.mean(0), z64.var(0)
        xhat = (z64 - mu) / np.sqrt(var + 1e-5)
        pre = xhat * gamma + beta
    else:
        sc = sh = None
        pre = z.astype(np.float64)
    want = np.maximum(pre, 0) * mask / (1 - p)
    y = ops.act_fwd(zd, sc, sh, drop_p=p, drop_seed=seed)
    assert y.stride(0) % 4 == 0
    np.testing.assert_allclose(y.cpu().numpy(), want, atol=TOL, rtol=1e-5)
    dy = r.standard_normal((rows, h)).astype(np.float32)
    dpre = dy.astype(np.float64) * mask / (1 - p) * (pre > 0)
    if bn and rows > 1:
        dz, dg, db = ops.bn_relu_bwd(ops.as_feat(dev(dy)), zd, dev(gamma), mean, rstd, sc, sh, drop_p=p, drop_seed=seed)
        dxhat = dpre * gamma
        want_dz = (dxhat - dxhat.mean(0) - xhat * (dxhat * xhat).mean(0)) / np.sqrt(var + 1e-5)
        np.testing.assert_allclose(dz.cpu().numpy(), want_dz, atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(dg.cpu().numpy(), (dpre * xhat).sum(0), atol=2e-3, rtol=1e-4)
        np.testing.assert_allclose(db.cpu().numpy(), dpre.sum(0), atol=2e-3, rtol=1e-4)
    elif not bn:
        dz, _, _ = ops.bn_relu_bwd(ops.as_feat(dev(dy)), zd, drop_p=p, drop_seed=seed)
        np.testing.assert_allclose(dz.cpu().numpy(), dpre, atol=TOL, rtol=1e-5)


# ------------------------------------------------------------------------------------------- K4
@pytest.mark.parametrize("rows,c", [(512, 40), (4096, 47), (140, 7), (37, 100),
                                    (5003, 2), (70001, 7), (4096, 8), (33333, 1)])      # >= 4096 rows of <= 8 classes: eight rows per wavefront
@pytest.mark.parametrize("kind", ["nll", "kl"])
def test_softmax_loss_vs_oracle(rows, c, kind):
    from glnn_amd import ops
    r = np.random.RandomState(rows + c)
    z = (r.standard_normal((rows, c)) * 2).astype(np.float32)
    lamb = 0.37
    if kind == "nll":
        y = r.randint(0, c, rows).astype(np.int64)
        loss_w, dz_w = so.loss_and_dlogits(z, y, "nll", lamb)
        loss, dz = ops.softmax_loss(dev(z), ops.LOSS_NLL, lamb, labels=dev(y))
    else:
        t = so.log_softmax(r.standard_normal((rows, c)).astype(np.float32))
        loss_w, dz_w = so.loss_and_dlogits(z, t, "kl", lamb)
        loss, dz = ops.softmax_loss(dev(z), ops.LOSS_KL, lamb, target_logp=dev(t))
    assert abs(float(loss.item()) - float(loss_w)) < TOL
    np.testing.assert_allclose(dz.cpu().numpy(), dz_w, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("n,rows,c", [(900, 256, 40), (9000, 4500, 2)])
def test_softmax_loss_indexed_targets_and_accum(n, rows, c):
    from glnn_amd import ops
    r = np.random.RandomState(0)
    z = r.standard_normal((rows, c)).astype(np.float32)
    labels_all = r.randint(0, c, n).astype(np.int64)
    t_all = so.log_softmax(r.standard_normal((n, c)).astype(np.float32))
    idx = r.permutation(n)[:rows].astype(np.int64)
    acc = torch.zeros(1, device=DEV)
    l1, d1 = ops.softmax_loss(dev(z), ops.LOSS_NLL, 1.0, labels=dev(labels_all), label_rows=dev(idx), loss_accum=acc)
    l2, d2 = ops.softmax_loss(dev(z), ops.LOSS_KL, 0.5, target_logp=dev(t_all), target_rows=dev(idx), loss_accum=acc)
    w1, g1 = so.loss_and_dlogits(z, labels_all[idx], "nll", 1.0)
    w2, g2 = so.loss_and_dlogits(z, t_all[idx], "kl", 0.5)
    assert abs(l1.item() - w1) < TOL and abs(l2.item() - w2) < TOL and abs(acc.item() - (w1 + w2)) < TOL
    np.testing.assert_allclose(d1.cpu().numpy(), g1, atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(d2.cpu().numpy(), g2, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("rows,k,c,p", [(4096, 2048, 47, 0.2), (4001, 256, 40, 0.5), (1030, 512, 7, 0.0), (2500, 1024, 48, 0.3), (4096, 768, 1, 0.1)])
@pytest.mark.parametrize("kind", ["nll", "kl"])
def test_classifier_loss_one_launch_vs_oracle_and_the_two_launch_form(rows, k, c, p, kind):
    """glnn_classifier_loss_f32 (cls_block.hip): logits = dropout(relu(z * a_scale + a_shift)) . W^T + b by 16-row workgroups with the
    criterion behind it in the same launch.  (a) logits against float64 on the engine's own dropout mask (1e-4), loss and dlogits against the
    oracle; (b) the stored-tail form (plain operand) gives the same logits bit for bit as the operand transform; (c) loss and dlogits
    are the BITS glnn_softmax_loss_f32 produces from the same logits (one wave per row, the same expressions and shuffle trees, the
    same per-four-rows partials); ragged row counts, 1 ... 48 classes, indexed targets."""
    from glnn_amd import ops
    r = np.random.RandomState(rows + k + c)
    z = dev(r.standard_normal((rows, k)).astype(np.float32))
    sc = dev((0.5 + r.rand(k)).astype(np.float32))
    sh = dev((0.3 * r.standard_normal(k)).astype(np.float32))
    w = dev((r.standard_normal((c, k)) / np.sqrt(k)).astype(np.float32))
    b = dev((0.1 * r.standard_normal(c)).astype(np.float32))
    n_all = rows + 37
    idx = dev(r.permutation(n_all)[:rows].astype(np.int64))
    seed, lamb = 12345 + rows, 0.7
    if kind == "nll":
        y_all = r.randint(0, c, n_all).astype(np.int64)
        kw = dict(labels=dev(y_all), label_rows=idx)
        knd = ops.LOSS_NLL
    else:
        t_all = so.log_softmax(r.standard_normal((n_all, c)).astype(np.float32))
        kw = dict(target_logp=ops.as_feat(dev(t_all)), target_rows=idx)
        knd = ops.LOSS_KL
    logits, loss, dl = ops.classifier_loss(z, w, b, knd, lamb, a_scale=sc, a_shift=sh, drop_p=p, drop_seed=seed, **kw)
    # (a)
    act = ops.act_fwd(z, sc, sh, p, seed)
    ref = act.double() @ w.double().t() + b.double()
    assert float((logits.double() - ref).abs().max()) < TOL
    tgt = y_all[idx.cpu().numpy()] if kind == "nll" else t_all[idx.cpu().numpy()]
    loss_w, dz_w = so.loss_and_dlogits(logits.cpu().numpy(), tgt, kind, lamb)
    assert abs(float(loss.item()) - float(loss_w)) < TOL
    np.testing.assert_allclose(dl.cpu().numpy(), dz_w, atol=1e-6, rtol=1e-4)
    # (b)
    logits_b, loss_b, dl_b = ops.classifier_loss(act, w, b, knd, lamb, **kw)
    assert torch.equal(logits_b, logits) and torch.equal(dl_b, dl) and float(loss_b) == float(loss)
    only, _, _ = ops.classifier_loss(z, w, b, -1, a_scale=sc, a_shift=sh, drop_p=p, drop_seed=seed)
    assert torch.equal(only, logits)
    # (c)
    loss_2, dl_2 = ops.softmax_loss(logits, knd, lamb, **kw)
    assert torch.equal(dl_2, dl)
    assert float(loss_2) == float(loss)


def test_classifier_loss_refuses_other_shapes():
    from glnn_amd import ops
    from glnn_amd._lib import GlnnError
    for rows, k, c in [(512, 2048, 47), (4096, 200, 47), (4096, 2048, 64)]:
        z = torch.randn(rows, k, device=DEV)
        w = torch.randn(c, k, device=DEV)
        with pytest.raises(GlnnError):
            ops.classifier_loss(z, w, None, -1)


@pytest.mark.parametrize("rows,c", [(1000, 47), (50001, 2), (4096, 7)])          # the last two: eight rows per wavefront
def test_log_softmax(rows, c):
    from glnn_amd import ops
    z = (np.random.RandomState(1).standard_normal((rows, c)) * 3).astype(np.float32)
    np.testing.assert_allclose(ops.log_softmax(dev(z)).cpu().numpy(), so.log_softmax(z), atol=1e-5, rtol=0)


# ------------------------------------------------------------------------------------------- K5
@pytest.mark.parametrize("rows,h", [(512, 256), (4096, 300), (100, 64), (129, 70), (40000, 70), (7000, 256)])   # 40000 rows: two-level folds
def test_bn_stats_and_backward_vs_oracle(rows, h):
    from glnn_amd import ops
    r = np.random.RandomState(rows)
    z = (r.standard_normal((rows, h)) * 1.7 + r.standard_normal(h) * 3).astype(np.float32)   # offset mean: var stability
    gamma, beta = r.uniform(.5, 1.5, h).astype(np.float32), r.standard_normal(h).astype(np.float32) * .2
    rm, rv = r.standard_normal(h).astype(np.float32) * .1, r.uniform(.5, 1.5, h).astype(np.float32)
    sd = {"encoder.layers.0.weight": np.eye(h, dtype=np.float32), "encoder.layers.0.bias": np.zeros(h, np.float32),
          "encoder.layers.1.weight": np.zeros((1, h), np.float32), "encoder.layers.1.bias": np.zeros(1, np.float32),
          "encoder.norms.0.weight": gamma, "encoder.norms.0.bias": beta, "encoder.norms.0.running_mean": rm,
          "encoder.norms.0.running_var": rv, "encoder.norms.0.num_batches_tracked": np.int64(3)}
    st = so.MLPState(sd, 2, "batch")
    _, cache = so.mlp_forward(st, z, training=True)          # layer 0 is the identity: z is the BN input
    t_rm, t_rv, t_nbt = dev(rm.copy()), dev(rv.copy()), torch.tensor([3], device=DEV)
    mean, rstd, a_sc, a_sh = ops.bn_stats(dev(z), dev(gamma), dev(beta), t_rm, t_rv, t_nbt)
    np.testing.assert_allclose(rstd.cpu().numpy(), cache["rstd"][0], rtol=1e-5)
    y = z * a_sc.cpu().numpy() + a_sh.cpu().numpy()
    np.testing.assert_allclose(y, cache["bn_out"][0], atol=TOL, rtol=0)
    np.testing.assert_allclose(t_rm.cpu().numpy(), st.rm[0], atol=1e-5, rtol=0)
    np.testing.assert_allclose(t_rv.cpu().numpy(), st.rv[0], atol=1e-5, rtol=1e-5)
    assert int(t_nbt.item()) == 4 == st.nbt[0]
    # backward: dlogits = ones through W1 = 0 would be zero, so drive mlp_backward pieces directly
    da = r.standard_normal((rows, h)).astype(np.float32)
    dy = da * (cache["bn_out"][0] > 0)
    xhat = cache["xhat"][0]
    s1, s2 = dy.sum(0, dtype=np.float64), (dy.astype(np.float64) * xhat).sum(0)
    dz_w = gamma * cache["rstd"][0] * (dy - s1 / rows - xhat * (s2 / rows))
    dzsum = torch.empty(h, device=DEV)
    dz, dg, db = ops.bn_relu_bwd(dev(da), dev(z), dev(gamma), mean, rstd, a_sc, a_sh, dz_col_sum=dzsum)
    # the column sums of dz are mathematically ZERO behind a BatchNorm: what is compared is fp32 summation noise, ~ rows * 1e-8
    np.testing.assert_allclose(dzsum.cpu().numpy(), dz_w.astype(np.float64).sum(0), atol=2e-4 * max(1.0, rows / 4096), rtol=0)
    np.testing.assert_allclose(dg.cpu().numpy(), s2, atol=2e-4 * max(1.0, rows / 4096), rtol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), s1, atol=2e-4 * max(1.0, rows / 4096), rtol=1e-5)
    np.testing.assert_allclose(dz.cpu().numpy(), dz_w, atol=TOL, rtol=0)
    dz2, _, _ = ops.bn_relu_bwd(dev(da), dev(z), dz_col_sum=dzsum)
    np.testing.assert_array_equal(dz2.cpu().numpy(), da * (z > 0))
    np.testing.assert_allclose(dzsum.cpu().numpy(), (da * (z > 0)).astype(np.float64).sum(0), atol=2e-4 * max(1.0, rows / 4096), rtol=1e-5)


@pytest.mark.parametrize("m,k,n", [(4096, 100, 2048), (5000, 100, 256), (70001, 128, 300), (2049, 36, 96), (100003, 64, 130),      # row-panel kernel
                                   (4096, 2048, 2048), (9000, 256, 256), (8321, 160, 1100), (16385, 512, 257),                     # pipelined kernel
                                   (300, 64, 200), (1000, 100, 64), (3000, 1024, 128)])                                            # neither: two-call form
def test_linear_bn_stats_equals_the_two_call_form(m, k, n, monkeypatch):
    """glnn_linear_bn_stats_f32: the BatchNorm statistics whose first pass comes out of the product kernel's epilogue -- the row-panel
    kernel's per-workgroup (count, mean, M2) triples accumulated along its walk, the pipelined kernel's per-tile mean / M2 -- against
    (a) the two-call form (GLNN_GEMM_STATS=0: same z bit for bit, statistics to rounding) and (b) float64 statistics of the stored z.
    Ragged last row tiles, column panels past n, columns with a large offset mean (the one-pass variance's shift at work)."""
    from glnn_amd import ops
    r = np.random.RandomState(m + k + n)
    a = ops.as_feat(dev(r.standard_normal((m, k)).astype(np.float32)))
    w = dev((r.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    bias = dev((r.standard_normal(n) * 5).astype(np.float32))                 # |mean| / std up to ~15
    gamma, beta = dev(r.uniform(.5, 1.5, n).astype(np.float32)), dev((r.standard_normal(n) * .2).astype(np.float32))
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GLNN_GEMM_STATS", mode)
        rm, rv, nbt = dev(np.zeros(n, np.float32)), dev(np.ones(n, np.float32)), torch.tensor([0], device=DEV)
        z, mean, rstd, a_sc, a_sh = ops.linear_bn_stats(a, w, bias, gamma, beta, rm, rv, nbt)
        outs.append((z[:, :n].clone(), mean.clone(), rstd.clone(), a_sc.clone(), a_sh.clone(), rm, rv, int(nbt.item())))
    assert torch.equal(outs[0][0], outs[1][0])
    zd = outs[0][0].double()
    mu, var = zd.mean(0), zd.var(0, unbiased=False)
    for o in outs:
        np.testing.assert_allclose(o[1].cpu().numpy(), mu.cpu().numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(o[2].cpu().numpy(), (1.0 / torch.sqrt(var + 1e-5)).cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(o[5].cpu().numpy(), (0.1 * mu).cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o[6].cpu().numpy(), (0.9 + 0.1 * zd.var(0, unbiased=True)).cpu().numpy(), rtol=1e-5)
        assert o[7] == 1
    for x, y in zip(outs[0][1:5], outs[1][1:5]):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("m,k,n", [(5000, 100, 256), (40000, 128, 128), (4096, 2048, 256)])
def test_linear_bn_stats_of_columns_far_from_zero(m, k, n, monkeypatch):
    """Columns with |mean| / std of several hundred: the statistics out of the product kernels' epilogues -- the wave-walk
    kernel's per-tile sums around the tile's first value merged tile by tile (Chan), the pipelined kernel's two passes over its accumulators
    -- keep the variance of the stored z to fp32 rounding, as the two-call form does."""
    from glnn_amd import ops
    r = np.random.RandomState(m + n)
    a = ops.as_feat(dev(r.standard_normal((m, k)).astype(np.float32)))
    w = dev((r.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    bias = dev((r.choice([-1.0, 1.0], n) * r.uniform(100, 400, n)).astype(np.float32))
    gamma, beta = dev(np.ones(n, np.float32)), dev(np.zeros(n, np.float32))
    for mode in ("1", "0"):
        monkeypatch.setenv("GLNN_GEMM_STATS", mode)
        rm, rv, nbt = dev(np.zeros(n, np.float32)), dev(np.ones(n, np.float32)), torch.tensor([0], device=DEV)
        z, mean, rstd, _, _ = ops.linear_bn_stats(a, w, bias, gamma, beta, rm, rv, nbt)
        zd = z[:, :n].double()
        mu, var = zd.mean(0), zd.var(0, unbiased=False)
        np.testing.assert_allclose(mean.cpu().numpy(), mu.cpu().numpy(), rtol=1e-6)
        np.testing.assert_allclose(rstd.cpu().numpy(), (1.0 / torch.sqrt(var + 1e-5)).cpu().numpy(), rtol=2e-4)


@pytest.mark.parametrize("rows,h,relu,p", [(100, 48, True, 0.0), (513, 512, True, 0.3), (64, 2048, True, 0.0), (33, 2500, False, 0.2),
                                           (40, 70, False, 0.0), (1000, 1, True, 0.0), (7, 4096, True, 0.5)])
def test_layernorm_forward_and_backward_vs_torch(rows, h, relu, p):
    """glnn_layernorm_fwd_f32 / glnn_layernorm_bwd_f32 = dropout(relu?(nn.LayerNorm(h)(z))) and its backward (reference
    models.py:30-31, 48-52), against torch.nn.functional.layer_norm in float64 with THIS library's dropout mask as an input:
    odd widths, rows held in registers (<= 2048 columns) and re-read (wider), one / two / four column quads per backward thread,
    ragged last row chunk, the column sums (dgamma, dbeta, bias gradient) and in-place dz == da."""
    import torch.nn.functional as F
    from glnn_amd import ops
    r = np.random.RandomState(rows + h)
    z = torch.from_numpy((r.standard_normal((rows, h)) * 1.5 + r.standard_normal((rows, 1))).astype(np.float32))
    gamma = torch.from_numpy(r.uniform(.5, 1.5, h).astype(np.float32))
    beta = torch.from_numpy((r.standard_normal(h) * .3).astype(np.float32))
    da = torch.from_numpy(r.standard_normal((rows, h)).astype(np.float32))
    seed = 4242
    keep = ops.dropout_mask(rows, h, p, seed, DEV).cpu().double() if p > 0 else torch.ones(rows, h, dtype=torch.float64)
    zd, gd, bd = z.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    y_ref = F.layer_norm(zd, (h,), gd, bd, 1e-5)
    if relu:
        y_ref = F.relu(y_ref)
    y_ref = y_ref * keep / (1.0 - p)
    y_ref.backward(da.double())
    y, mean, rstd = ops.layernorm_fwd(dev(z.numpy()), dev(gamma.numpy()), dev(beta.numpy()), relu=relu, drop_p=p, drop_seed=seed)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(mean.cpu().numpy(), z.double().mean(1).numpy(), atol=1e-5, rtol=0)
    if y.stride(0) > h:
        base = torch.as_strided(y, (rows, y.stride(0)), (y.stride(0), 1))
        assert float(base[:, h:].abs().max()) == 0.0
    dzsum = torch.empty(h, device=DEV)
    dz, dg, db = ops.layernorm_bwd(dev(da.numpy()), dev(z.numpy()), dev(gamma.numpy()), dev(beta.numpy()), mean, rstd, relu=relu, drop_p=p,
                                   drop_seed=seed, dz_col_sum=dzsum)
    scale = max(1.0, float(zd.grad.abs().max()))
    np.testing.assert_allclose(dz.cpu().numpy(), zd.grad.numpy(), atol=5e-5 * scale, rtol=1e-4)
    np.testing.assert_allclose(dg.cpu().numpy(), gd.grad.numpy(), atol=2e-4 * max(1.0, rows / 512), rtol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bd.grad.numpy(), atol=2e-4 * max(1.0, rows / 512), rtol=1e-4)
    np.testing.assert_allclose(dzsum.cpu().numpy(), zd.grad.sum(0).numpy(), atol=2e-4 * max(1.0, rows / 512) * scale, rtol=1e-4)
    buf = ops.as_feat(dev(da.numpy()))
    dz2, _, _ = ops.layernorm_bwd(buf, dev(z.numpy()), dev(gamma.numpy()), dev(beta.numpy()), mean, rstd, relu=relu, drop_p=p, drop_seed=seed,
                                  dz=buf, want_param_grads=False)
    assert torch.equal(dz2, dz)


def test_norm_tail_without_relu_forward_and_backward():
    """glnn_norm_drop_fwd_f32 / glnn_bn_bwd_f32 with relu = 0: y = dropout(BatchNorm_train(z)) (GCN's norm -> dropout tail) and its
    backward vs torch in float64 (this library's dropout mask as an input)."""
    import torch.nn.functional as F
    from glnn_amd import ops
    rows, h, p, seed = 700, 96, 0.4, 99
    r = np.random.RandomState(5)
    z = torch.from_numpy((r.standard_normal((rows, h)) * 2 + 1).astype(np.float32))
    gamma, beta = torch.from_numpy(r.uniform(.5, 1.5, h).astype(np.float32)), torch.from_numpy(r.standard_normal(h).astype(np.float32))
    da = torch.from_numpy(r.standard_normal((rows, h)).astype(np.float32))
    keep = ops.dropout_mask(rows, h, p, seed, DEV).cpu().double()
    zd, gd, bd = z.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    y_ref = F.batch_norm(zd, None, None, gd, bd, True, 0.1, 1e-5) * keep / (1 - p)
    y_ref.backward(da.double())
    mean, rstd, a_sc, a_sh = ops.bn_stats(dev(z.numpy()), dev(gamma.numpy()), dev(beta.numpy()), torch.zeros(h, device=DEV), torch.ones(h, device=DEV),
                                          torch.zeros(1, dtype=torch.int64, device=DEV))
    y = ops.act_fwd(dev(z.numpy()), a_sc, a_sh, drop_p=p, drop_seed=seed, relu=False)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), atol=2e-5, rtol=1e-5)
    assert float(y.min()) < 0        # no ReLU
    dz, dg, db = ops.bn_relu_bwd(dev(da.numpy()), dev(z.numpy()), dev(gamma.numpy()), mean, rstd, a_sc, a_sh, drop_p=p, drop_seed=seed, relu=False)
    np.testing.assert_allclose(dz.cpu().numpy(), zd.grad.numpy(), atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(dg.cpu().numpy(), gd.grad.numpy(), atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bd.grad.numpy(), atol=5e-4, rtol=1e-4)


# ------------------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("wd", [0.0, 5e-4])
def test_adam_vs_oracle(wd):
    from glnn_amd import ops
    r = np.random.RandomState(3)
    shapes = [(256, 100), (256,), (47, 256), (47,), (1,), (1000, 33)]
    ps = [r.standard_normal(s).astype(np.float32) for s in shapes]

    class St:       # minimal state the oracle's adam_step needs
        pass
    st = St(); st.t = 0
    st.params = lambda: st._p
    st._p = [p.copy() for p in ps]; st.m = [np.zeros_like(p) for p in ps]; st.v = [np.zeros_like(p) for p in ps]
    tp = [dev(p) for p in ps]; tg = [torch.zeros_like(p) for p in tp]
    tm = [torch.zeros_like(p) for p in tp]; tv = [torch.zeros_like(p) for p in tp]
    table = ops.TensorTable(tp, tg, tm, tv)
    for step in range(1, 6):
        gs = [(r.standard_normal(s) * (10.0 ** r.randint(-4, 1))).astype(np.float32) for s in shapes]
        so.adam_step(st, gs, lr=0.01, weight_decay=wd)
        for t, g in zip(tg, gs):
            t.copy_(dev(g))
        ops.adam_step(table, 0.01, step, weight_decay=wd)
        for a, b in zip(tp, st._p):
            np.testing.assert_allclose(a.cpu().numpy(), b, atol=2e-6, rtol=1e-5)
    for a, b in zip(tm, st.m):
        np.testing.assert_allclose(a.cpu().numpy(), b, atol=1e-7, rtol=1e-5)


# ------------------------------------------------------------------------------------------- K7
def test_gather_scatter_rows():
    from glnn_amd import ops
    r = np.random.RandomState(0)
    x = r.standard_normal((1000, 100)).astype(np.float32)
    rows = r.permutation(1000)[:300].astype(np.int64)
    got = ops.gather_rows(dev(x), dev(rows))
    np.testing.assert_array_equal(got.cpu().numpy(), x[rows])
    y = ops.feat_empty(1000, 100, DEV, zero=True)
    ops.scatter_rows(got, dev(rows), y)
    want = np.zeros_like(x); want[rows] = x[rows]
    np.testing.assert_array_equal(y.cpu().numpy(), want)


# ------------------------------------------------------------------------------------------- edge cases
def test_edge_cases_empty_single_and_giant_rows():
    from glnn_amd import ops
    # (1) zero destination rows: a no-op that must not fault
    ip0 = torch.zeros(1, dtype=torch.int64, device=DEV); ix0 = torch.zeros(0, dtype=torch.int32, device=DEV)
    x = torch.randn(5, 8, device=DEV)
    assert ops.spmm(ip0, ix0, x, 0, ops.AGG_SAGE_GCN).shape == (0, 8)
    assert ops.sage_fused(ip0, ix0, x, 0, torch.randn(16, 8, device=DEV)).shape == (0, 16)
    # (2) a graph with no edges at all: SAGE-gcn returns the self features, SUM returns zeros
    ip = torch.zeros(6, dtype=torch.int64, device=DEV)
    np.testing.assert_array_equal(ops.spmm(ip, ix0, x, 5, ops.AGG_SAGE_GCN).cpu().numpy(), x.cpu().numpy())
    assert float(ops.spmm(ip, ix0, x, 5, ops.AGG_SUM).abs().max()) == 0.0
    # (3) one destination row with 20,000 in-edges (ogbn-products' max degree is 17,481) next to empty rows:
    #     the whole-workgroup long-row role in both the stand-alone and the fused kernel
    n, d = 4000, 100
    r = np.random.RandomState(0)
    src = r.randint(0, n, 20000); dst = np.full(20000, 1234)
    indptr, indices = csr_from_edges(src, dst, n)
    xx = r.standard_normal((n, d)).astype(np.float32)
    want = to.sage_gcn_agg(indptr, indices, xx)
    got = ops.spmm(dev(indptr), dev(indices), dev(xx), n, ops.AGG_SAGE_GCN)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=TOL, rtol=0)
    w = (r.standard_normal((64, d)) / 10).astype(np.float32)
    gotf = ops.sage_fused(dev(indptr), dev(indices), dev(xx), n, dev(w))
    np.testing.assert_allclose(gotf.cpu().numpy(), to.linear(want, w), atol=TOL, rtol=0)
    # (4) bit-for-bit run-to-run determinism (no float atomics anywhere)
    again = ops.spmm(dev(indptr), dev(indices), dev(xx), n, ops.AGG_SAGE_GCN)
    assert torch.equal(got, again)
    assert torch.equal(gotf, ops.sage_fused(dev(indptr), dev(indices), dev(xx), n, dev(w)))


def test_argument_errors_are_raised_not_crashes():
    from glnn_amd import GlnnError, ops
    x = torch.randn(10, 6, device=DEV)[:, :5]            # row stride 6: not a multiple of 4 -> as_feat copies; fine
    ip = torch.zeros(11, dtype=torch.int64, device=DEV); ix = torch.zeros(0, dtype=torch.int32, device=DEV)
    assert ops.spmm(ip, ix, x, 10, ops.AGG_SUM).shape == (10, 5)
    with pytest.raises(ValueError):
        ops.spmm(ip.int(), ix, x, 10, ops.AGG_SUM)        # indptr must be int64
    with pytest.raises(ValueError):
        ops.gemm(torch.randn(4, 8, device=DEV), torch.randn(3, 7, device=DEV))
    with pytest.raises(GlnnError):
        ops.sage_fused(ip, ix, torch.randn(10, 300, device=DEV), 10, torch.randn(16, 300, device=DEV))   # d_in > 256
    with pytest.raises(GlnnError):
        ops.spmm(ip, ix, torch.randn(10, 8, device=DEV), 10, 7)                                        # unknown mode


def test_full_size_student_pass_properties():
    """A full products-sized soft-label pass (597 steps of B=4096 over 2,449,029 rows) of the student bench.py times --
    MLP3w8, 100-2048-2048-47, BatchNorm, dropout 0.2 (reference train.conf.yaml:187-194) -- through train_mini_batch:
    finite decreasing loss, BatchNorm counters advanced by exactly the step count, Adam step counter in sync."""
    from glnn_amd import data
    from glnn_amd import train_and_eval as te
    from glnn_amd.models import Model
    torch.manual_seed(0)
    n = 2449029
    feats, labels, out_t, _ = data.make_node_data("ogbn-products", seed=0, device=DEV, n=n)
    w = torch.randn(100, 47, device=DEV)
    out_t = torch.log_softmax(feats @ w, dim=1)               # a learnable teacher
    model = Model(dict(model_name="MLP3w8", num_layers=3, feat_dim=100, hidden_dim=2048, label_dim=47, dropout_ratio=0.2,
                       norm_type="batch", device=DEV))
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0)
    crit = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    l1 = te.train_mini_batch(model, feats, out_t, 4096, crit, opt, 1.0)
    l2 = te.train_mini_batch(model, feats, out_t, 4096, crit, opt, 1.0)
    steps = n // 4096
    assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1
    assert int(model.encoder.norms[0].num_batches_tracked) == 2 * steps
    assert int(opt.state[next(model.parameters())]["step"]) == 2 * steps
    out, loss, score = te.evaluate_mini_batch(model, feats, labels, torch.nn.NLLLoss(), 4096, lambda o, y: 0.0)
    assert out.shape == (n, 47) and torch.isfinite(out).all()
    assert float((out.exp().sum(1) - 1).abs().max()) < 1e-4
    agree = (out.argmax(1) == out_t.argmax(1)).float().mean().item()
    assert agree > 0.5, agree                                  # the student tracks the teacher after two passes


# ------------------------------------------------------------------------------------------- randomized sweeps
def test_randomized_shape_sweep_vs_oracle():
    """40 seeded random (graph, width, mode) combinations through both aggregation kernels and the three GEMM forms."""
    from glnn_amd import ops
    import os
    base = int(os.environ.get("GLNN_SHAPE_SEED", "2024"))       # other seeds: ad-hoc soak runs (scripts/soak.sh)
    rs = np.random.RandomState(base)
    for it in range(40):
        n = int(rs.randint(1, 2500))
        deg = float(rs.choice([0.5, 3, 12, 40]))
        d = int(rs.choice([1, 3, 8, 31, 47, 64, 100, 129, 200, 256]))
        indptr, indices = random_graph(n, deg, seed=it + (0 if base == 2024 else 40 * base), power=float(rs.choice([0.0, 0.5, 0.9])), isolated=int(min(n // 3, rs.randint(0, 5))),
                                       hub=int(rs.choice([0, 0, 300])) if n > 50 else 0)
        x = rs.standard_normal((n, d)).astype(np.float32)
        ip, ix, xd = dev(indptr), dev(indices), dev(x)
        want = to.sage_gcn_agg(indptr, indices, x)
        np.testing.assert_allclose(ops.spmm(ip, ix, xd, n, ops.AGG_SAGE_GCN).cpu().numpy(), want, atol=TOL, rtol=0, err_msg=f"it={it} n={n} d={d}")
        rsc = rs.uniform(.2, 1, n).astype(np.float32)
        np.testing.assert_allclose(ops.spmm(ip, ix, xd, n, ops.AGG_SUM, row_scale=dev(rsc), col_scale=dev(rsc)).cpu().numpy(),
                                   to.spmm_sum(indptr, indices, x, rsc, rsc), atol=TOL, rtol=1e-5, err_msg=f"sum it={it}")
        d_out = int(rs.choice([1, 5, 32, 47, 100, 256]))
        w = (rs.standard_normal((d_out, d)) / np.sqrt(d)).astype(np.float32)
        got = ops.sage_fused(ip, ix, xd, n, dev(w))
        np.testing.assert_allclose(got.cpu().numpy(), to.linear(want, w), atol=TOL, rtol=0, err_msg=f"fused it={it} n={n} d={d} d_out={d_out}")
        # GEMM forms on the same operands
        np.testing.assert_allclose(ops.gemm(ops.as_feat(xd), dev(w)).cpu().numpy(), to.linear(x, w), atol=TOL, rtol=1e-5)
        wk = np.ascontiguousarray(w.T)
        np.testing.assert_allclose(ops.gemm(ops.as_feat(xd), dev(wk), w_is_kn=True).cpu().numpy(), to.linear(x, w), atol=TOL, rtol=1e-5)
        dz = rs.standard_normal((n, d_out)).astype(np.float32) / 4
        np.testing.assert_allclose(ops.gemm_tn(ops.as_feat(dev(dz)), ops.as_feat(xd)).cpu().numpy(),
                                   dz.astype(np.float64).T @ x.astype(np.float64), atol=2e-4, rtol=1e-5, err_msg=f"tn it={it}")


def test_randomized_gemm_sweep_all_variants():
    """30 seeded random GEMM problems across the tile regimes (64x64 latency tiles, 128x64, 128x128), the three forms,
    operand transforms (BN affine + ReLU, + dropout through the kernel's own mask), row gathers, epilogues and split-K
    (workspace on/off); float64 numpy is the reference."""
    from glnn_amd import ops
    import os
    rs = np.random.RandomState(int(os.environ.get("GLNN_SWEEP_SEED", "77")))     # other seeds: ad-hoc soak runs
    for it in range(30):
        m = int(rs.choice([1, 37, 512, 700, 4096, 9000, 20011]))
        k = int(rs.choice([4, 20, 100, 128, 256, 516, 1024]))
        n = int(rs.choice([1, 7, 47, 64, 100, 256, 300, 1024]))
        xf = int(rs.randint(0, 3))
        gather = bool(rs.randint(0, 2))
        use_ws = bool(rs.randint(0, 2))
        nsrc = m + int(rs.randint(0, 50))
        a = rs.standard_normal((nsrc, k)).astype(np.float32)
        w = (rs.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        rows = rs.randint(0, nsrc, m).astype(np.int64) if gather else None
        a_eff = (a[rows] if gather else a[:m]).astype(np.float64)
        kw = {}
        p, seed = 0.3, 1000 + it
        if xf:
            sc, sh = rs.uniform(.5, 1.5, k).astype(np.float32), (rs.standard_normal(k) * .2).astype(np.float32)
            a_eff = np.maximum(a_eff * sc + sh, 0)
            kw.update(a_scale=dev(sc), a_shift=dev(sh))
            if xf == 2:
                a_eff = a_eff * ops.dropout_mask(m, k, p, seed, DEV).cpu().numpy() / (1 - p)
                kw.update(drop_p=p, drop_seed=seed)
        es, eh = rs.uniform(.5, 1.5, n).astype(np.float32), rs.standard_normal(n).astype(np.float32)
        relu = bool(rs.randint(0, 2))
        want = a_eff @ w.astype(np.float64).T * es + eh
        if relu:
            want = np.maximum(want, 0)
        ws = torch.empty(1 << 23, device=DEV) if use_ws else None
        ad = ops.as_feat(dev(a))
        tag = f"it={it} m={m} k={k} n={n} xf={xf} gather={gather} ws={use_ws}"
        got = ops.gemm(ad, ops.as_feat(dev(w)), a_rows=dev(rows) if gather else None, m=m, ep_scale=dev(es), ep_shift=dev(eh), relu=relu,
                       workspace=ws, **kw)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-4, rtol=1e-5, err_msg="NT " + tag)
        got = ops.gemm(ad, ops.as_feat(dev(np.ascontiguousarray(w.T))), w_is_kn=True, a_rows=dev(rows) if gather else None, m=m,
                       ep_scale=dev(es), ep_shift=dev(eh), relu=relu, workspace=ws, **kw)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-4, rtol=1e-5, err_msg="KN " + tag)
        # TN: dW[n_out, k] = dz^T @ a_eff   (the transform / gather / dropout sit on the second operand)
        dz = (rs.standard_normal((m, n)) / 8).astype(np.float32)
        kt = {}
        if xf:
            kt.update(b_scale=kw["a_scale"], b_shift=kw["a_shift"])
            if xf == 2:
                kt.update(drop_p=p, drop_seed=seed)
        colsum = torch.empty(n, device=DEV)
        got = ops.gemm_tn(ops.as_feat(dev(dz)), ad, b_rows=dev(rows) if gather else None, m=m, col_sum_a=colsum, **kt)
        np.testing.assert_allclose(got.cpu().numpy(), dz.astype(np.float64).T @ a_eff, atol=3e-4, rtol=1e-5, err_msg="TN " + tag)
        np.testing.assert_allclose(colsum.cpu().numpy(), dz.astype(np.float64).sum(0), atol=2e-4, rtol=1e-5, err_msg="colsum " + tag)


@pytest.mark.parametrize("seed", [0, 1])
def test_gemm_dispatch_on_random_shapes(seed):
    """scripts/fuzz_gemm.py: 60 random glnn_gemm_f32 problems per seed against float64 torch -- widths that are not multiples of 4 (the
    dword-loading latency kernel, the generic kernel), row gathers, operand transform + counter-hash dropout, epilogue scale / shift /
    ReLU, both weight layouts, 1 ... 4096 rows: every kernel the dispatcher can pick."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_gemm
    bad = [(d, e) for d, e, ok in fuzz_gemm.run(seed, 60, verbose=False) if not ok or e >= 2e-5]
    assert not bad, bad
