"""The REAL sharded HIP path with 2 ranks: both processes share cuda:0 and exchange through gloo (RCCL refuses
two ranks on one device), so everything except the collective's transport is what an N-GPU run executes:
row_range shards, x_self offsets, slot writes into the padded activation buffers, in-place all-gather,
project-first last layer.  The gathered logits must equal the single-GPU Model.inference."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, dims, chunks, balanced, q, halo=False, variant=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd import data, ops
        from glnn_amd.dist import HaloShardedTeacher, RowShards, ShardedTeacher, make_grad_sync
        from glnn_amd.graph import FullNeighborLoader
        from glnn_amd.models import Model
        dev = "cuda:0"
        torch.manual_seed(0)
        g = data.make_graph("ogbn-arxiv", seed=0, device=dev, scale=n / 169343)
        nn_ = g.n_dst
        x = torch.randn(nn_, dims[0], device=dev)
        model = Model(dict(model_name="SAGE", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                           dropout_ratio=0.5, norm_type="batch", device=dev))
        with torch.no_grad():
            for bn in model.encoder.norms:
                bn.running_mean.uniform_(-.3, .3); bn.running_var.uniform_(.5, 1.5); bn.weight.uniform_(.5, 1.5); bn.bias.uniform_(-.2, .2)
        model.eval()
        want = model.inference(FullNeighborLoader(g, 1024), x)
        sh = RowShards(nn_, world, rank, chunks=chunks, bounds=RowShards.balanced_bounds(g.indptr, world) if balanced else None)
        with torch.no_grad():
            if halo:
                y = HaloShardedTeacher(model.encoder, g.row_range(sh.lo, sh.hi), sh, ops, overlap=(variant == "overlap")).forward(x)
            else:
                y = ShardedTeacher(model.encoder, g.row_range(sh.lo, sh.hi), sh, ops, widening_exchange=variant or "narrow").forward(x)
        err = float((y - want[sh.lo:sh.hi]).abs().max())
        flat = torch.full((8,), float(rank + 1), device=dev)
        make_grad_sync(flat, world, average=True)()
        q.put((rank, sh.lo, sh.hi, err, flat.cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,chunks,balanced", [([128, 256, 256, 40], 1, False), ([100, 256, 256, 47], 4, False), ([100, 256, 256, 47], 4, True),
                                                  ([128, 256, 256, 40], 1, True)])
def test_sharded_teacher_hip_two_ranks_one_gpu(dims, chunks, balanced):
    world, n = 2, 9001
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, dims, chunks, balanced, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    covered = 0
    for rank, lo, hi, err, flat in res:
        assert err < 1e-4, (rank, err)
        covered += hi - lo
        np.testing.assert_allclose(flat, np.full(8, 1.5))
    assert covered >= n - 2


@pytest.mark.parametrize("halo,variant,chunks,dims", [(True, None, 1, [100, 256, 256, 47]), (True, "overlap", 1, [100, 256, 256, 47]),
                                                     (True, "overlap", 1, [64, 64, 64, 64]), (False, "wide", 4, [100, 256, 256, 47]),
                                                     (False, "wide", 1, [100, 256, 256, 47])])
def test_halo_sharded_teacher_hip_two_ranks_one_gpu(halo, variant, chunks, dims):
    """The halo exchange with the real kernels: glnn_gather_rows_f32 packs the rows the peer references, the relabelled
    [own | halo] column ids feed the fused / stand-alone aggregation kernels; its OVERLAPPED form (asynchronous exchange, two-pass
    aggregation over the split CSR with the AGG_SUM kernel + row scale); and the all-gather teacher exchanging the WIDE output
    of the first layer (bench.py --layer1-exchange wide)."""
    world, n = 2, 9001
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, dims, chunks, True, q, halo, variant)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, lo, hi, err, flat in res:
        assert err < 1e-4, (rank, err)


def _rccl_worker(port, q):
    """ONE rank, backend "nccl" (= RCCL): the production transport branches of glnn_amd.dist -- the in-place
    all_gather_into_tensor of a slot of the padded buffer, its async chunked form, the flat-gradient all_reduce, the
    StatExchange hook -- executed for real on the GPU (a 1-rank communicator; FORCE_COLLECTIVES keeps the world == 1
    short-cuts from skipping them).  Values must be unchanged: with one rank every collective is the identity."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from glnn_amd import data, ops
        from glnn_amd import dist as gdist
        from glnn_amd.graph import FullNeighborLoader
        from glnn_amd.models import Model
        gdist.FORCE_COLLECTIVES = True
        torch.manual_seed(0)
        g = data.make_graph("ogbn-arxiv", seed=0, device=dev, scale=0.05)
        n = g.n_dst
        errs = {}
        for dims, chunks in (([100, 256, 256, 47], 4), ([128, 256, 256, 40], 1), ([64, 32, 16], 2)):
            x = torch.randn(n, dims[0], device=dev)
            model = Model(dict(model_name="SAGE", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                               dropout_ratio=0.5, norm_type="batch", device=dev))
            model.eval()
            want = model.inference(FullNeighborLoader(g, 1024), x)
            sh = gdist.RowShards(n, 1, 0, chunks=chunks)
            gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
            with torch.no_grad():
                y = gdist.ShardedTeacher(model.encoder, g, sh, ops).forward(x)
            torch.cuda.synchronize()
            errs[(tuple(dims), chunks)] = (float((y - want).abs().max()), gdist.EXCHANGE_STATS["collectives"])
            gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
            with torch.no_grad():          # the halo form: all_to_all_single with split sizes on the 1-rank communicator
                yh = gdist.HaloShardedTeacher(model.encoder, g, sh, ops).forward(x)
            torch.cuda.synchronize()
            errs[(tuple(dims), "halo")] = (float((yh - want).abs().max()), gdist.EXCHANGE_STATS["collectives"])
            gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
            with torch.no_grad():          # its overlapped form: async all_to_all_single (work.wait) + the two-pass aggregation
                yo = gdist.HaloShardedTeacher(model.encoder, g, sh, ops, overlap=True).forward(x)
            torch.cuda.synchronize()
            errs[(tuple(dims), "halo-overlap")] = (float((yo - want).abs().max()), gdist.EXCHANGE_STATS["collectives"])
            gdist.EXCHANGE_STATS.update(collectives=0, floats_received=0)
            with torch.no_grad():          # the all-gather teacher exchanging the wide first-layer output, chunked + async
                yw = gdist.ShardedTeacher(model.encoder, g, sh, ops, widening_exchange="wide").forward(x)
            torch.cuda.synchronize()
            errs[(tuple(dims), "wide")] = (float((yw - want).abs().max()), gdist.EXCHANGE_STATS["collectives"])
            if chunks > 1 and dims[1] == 256:
                # round 6: the fused layers as ONE launch over the chunks, every chunk's all-gather held behind its completion signal
                # (hipStreamWaitValue32 on the exchange stream) -- the same bits as one launch per chunk, narrow and wide
                for form, got in (("narrow", y), ("wide", yw)):
                    gdist.ONE_LAUNCH = False
                    with torch.no_grad():
                        per_chunk = gdist.ShardedTeacher(model.encoder, g, sh, ops, widening_exchange=form).forward(x)
                    gdist.ONE_LAUNCH = True
                    torch.cuda.synchronize()
                    errs[(tuple(dims), "one-launch == per-chunk launches, " + form)] = (0.0 if torch.equal(got, per_chunk) else 1.0, 1)
        # the gradient all-reduce started from inside the backward (grad_ready hook -> async RCCL all-reduce on the communicator's
        # stream -> wait before Adam): with one rank every reduce is the identity, so the steps must equal the plain engine's
        import copy
        from glnn_amd.student import StudentEngine
        torch.manual_seed(1)
        base = Model(dict(model_name="MLP", num_layers=3, feat_dim=100, hidden_dim=512, label_dim=47, dropout_ratio=0.2,
                          norm_type="batch", device=dev))
        xs = ops.as_feat(torch.randn(4096, 100, device=dev))
        tg = ops.as_feat(torch.log_softmax(torch.randn(4096, 47, device=dev), 1))
        states, hook_calls, colls = [], 0, 0
        for overlap in (False, True):
            m2 = copy.deepcopy(base); m2.train()
            o2 = torch.optim.Adam(m2.parameters(), lr=0.01)
            eng = StudentEngine(m2, o2, 1024)
            if overlap:
                eng.overlap = gdist.OverlappedGradSync(eng, 1, average=True)
                assert sorted(eng.overlap.big) == [1] and len(eng.overlap.rest) == 2          # the 512 x 512 gradient; before / after it
            for i in range(3):
                eng.step(xs, torch.arange(i * 5, i * 5 + 1024, device=dev), ops.LOSS_KL, tg, 1.0)
            torch.cuda.synchronize()
            if overlap:
                hook_calls, colls = eng.overlap.calls, eng.overlap.collectives
            states.append([t.detach().clone() for t in m2.state_dict().values()])
        overlap_equal = all(torch.equal(a, b) for a, b in zip(*states))
        errs[("student-overlap", "hooks")] = (0.0 if (overlap_equal and hook_calls == 9) else 1.0, colls // 3)   # 3 collectives per step
        flat = torch.arange(16, dtype=torch.float32, device=dev)
        gdist.make_grad_sync(flat, 2, average=True)()            # "world 2" arithmetic over the 1-rank communicator: sum / 2
        ex = gdist.StatExchange(1, 0, 8, dev)
        ex.send[:24] = torch.arange(24, dtype=torch.float32, device=dev)
        rc = ex.callback(None, ex.send.data_ptr(), ex.recv.data_ptr(), 24, None)
        torch.cuda.synchronize()
        q.put((errs, flat.cpu().numpy().copy(), rc, repr(ex.error), ex.recv[:24].cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_rccl_transport_branches_execute_on_one_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    errs, flat, rc, err, recv = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    for key, (e, ncoll) in errs.items():
        assert e < 1e-4 and ncoll >= 1, (key, e, ncoll)
    np.testing.assert_allclose(flat, np.arange(16) / 2)
    assert rc == 0 and err == "None"
    np.testing.assert_array_equal(recv, np.arange(24, dtype=np.float32))


def _student_setup(dims, norm, dropout, dev, seed=3, big=False):
    from glnn_amd.models import Model
    torch.manual_seed(seed)
    model = Model(dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1],
                       dropout_ratio=dropout, norm_type=norm, device=dev))
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    g = torch.Generator().manual_seed(seed)
    n = 30000 if big else 3000
    feats = torch.randn(n, dims[0], generator=g).to(dev)
    labels = torch.randint(0, dims[-1], (n,), generator=g).to(dev)
    out_t = torch.log_softmax(torch.randn(n, dims[-1], generator=g), 1).to(dev)
    sizes = (24000, 24000, 20001, 24000) if big else (512, 512, 300, 512)                  # 300 / 20001: unequal, odd chunk tail
    batches = [torch.randperm(n, generator=g)[:b].to(dev) for b in sizes]
    return model, opt, feats, labels, out_t, batches


def _run_student(model, opt, feats, labels, out_t, batches, world, rank, group_enabled):
    from glnn_amd import ops
    from glnn_amd.student import StudentEngine
    model.train()
    eng = StudentEngine(model, opt, max(b.numel() for b in batches))
    if group_enabled:
        eng.enable_batch_split(world, rank)
    first_grads = None
    for i, idx in enumerate(batches):
        if group_enabled:                                  # uneven split on purpose: rank 0 takes ~60 %
            cut = (idx.numel() * 3) // 5
            mine = idx[:cut] if rank == 0 else idx[cut:]
            eng.loss_scale_rows = idx.numel()
        else:
            mine = idx
        mine = mine.contiguous()
        if i % 2 == 0:
            eng.step(feats, mine, ops.LOSS_NLL, labels, 0.3)
        else:
            eng.step(feats, mine, ops.LOSS_KL, out_t, 0.7)
        if i == 0:
            first_grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    return sd, first_grads, (eng.exchange.calls if eng.exchange else 0)


def _student_worker(rank, world, port, dims, norm, q, big=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        setup = _student_setup(dims, norm, 0.0, "cuda:0", big=big)
        sd, grads, calls = _run_student(*setup, world, rank, True)
        q.put((rank, sd, grads, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,norm,big", [([100, 256, 256, 47], "batch", False), ([128, 192, 40], "batch", False), ([50, 64, 64, 7], "none", False),
                                           ([100, 512, 512, 47], "batch", True)])
def test_student_batch_split_two_ranks_equals_single_gpu_step(dims, norm, big):
    """a batch split over ranks (uneven slices, global BatchNorm statistics through the exchange hook,
    summed gradients) must take the same optimisation steps as one GPU on the whole batch -- parameters, BN running
    statistics and num_batches_tracked after 4 mixed NLL/KL steps.  big: batches of 24000 rows (14400 / 9600 per rank): the products run on
    the row-panel and the pipelined kernels, whose epilogues leave the statistics' first pass as per-workgroup triples / per-tile partials
    that the rank-level combine and exchange then start from (glnn::ColStats under a BnGroup)."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_student_worker, args=(r, world, port, dims, norm, q, big)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want, want_grads, _ = _run_student(*_student_setup(dims, norm, 0.0, "cuda:0", big=big), 1, 0, False)
    n_bn = len(dims) - 2 if norm == "batch" else 0
    L = len(dims) - 1

    def gauge(k):     # bias in front of a BatchNorm (zero true gradient: rounding noise that Adam amplifies) and the
        if norm != "batch":                      # running_mean that tracks it -- see tests/parity_rules.py
            return False
        if "running_mean" in k:
            return True
        return ".layers." in k and k.endswith(".bias") and int(k.split(".")[2]) < L - 1

    for rank, sd, grads, calls in res:
        assert calls == 4 * 2 * n_bn, (rank, calls)          # 4 steps x (fwd + bwd) x BN layers
        for k, gw in want_grads.items():                      # step-1 gradients (after the all-reduce): linear, tight
            if not gauge(k):
                scale = np.abs(gw).max() + 1e-12
                assert np.abs(grads[k] - gw).max() <= 2e-5 * scale + 1e-9, (rank, k, np.abs(grads[k] - gw).max(), scale)
        assert sd.keys() == want.keys()
        for k in want:
            if k.endswith("num_batches_tracked"):
                assert sd[k] == want[k], k
            elif not gauge(k):
                d = np.abs(sd[k].astype(np.float64) - want[k])
                # max: one lr-sized Adam flip.  big: gradients of 24000-row batches are an order smaller against the same fp32 summation
                # noise, and Adam normalises them away: mean 6e-5 with AND without the statistics epilogues (GLNN_GEMM_STATS=0)
                assert d.mean() <= (1e-4 if big else 1e-5) and d.max() <= 0.01, (rank, k, d.mean(), d.max())
    for k in want:            # the two ranks hold bit-identical models (same gathered sums, same order)
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)
