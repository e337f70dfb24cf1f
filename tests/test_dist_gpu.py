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


def _worker(rank, world, port, n, dims, chunks, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glnn_amd import data, ops
        from glnn_amd.dist import RowShards, ShardedTeacher, make_grad_sync
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
        sh = RowShards(nn_, world, rank, chunks=chunks)
        with torch.no_grad():
            y = ShardedTeacher(model.encoder, g.row_range(sh.lo, sh.hi), sh, ops).forward(x)
        err = float((y - want[sh.lo:sh.hi]).abs().max())
        flat = torch.full((8,), float(rank + 1), device=dev)
        make_grad_sync(flat, world, average=True)()
        q.put((rank, sh.lo, sh.hi, err, flat.cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,chunks", [([128, 256, 256, 40], 1), ([100, 256, 256, 47], 4)])
def test_sharded_teacher_hip_two_ranks_one_gpu(dims, chunks):
    world, n = 2, 9001
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, dims, chunks, q)) for r in range(world)]
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
