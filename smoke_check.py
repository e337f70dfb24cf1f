"""__graft_entry__.smoke(): one small invocation of each half of the hot path on cuda:0 -- teacher forward, student
distillation step, one sampled-block teacher training step -- checked against the CPU oracle.  Lives at the repo root, NOT in
the product package: it imports `oracle/` (the oracle is the checker here, never the product path)."""
import os
import sys

import numpy as np
import torch


def smoke():
    root = os.path.dirname(os.path.abspath(__file__))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import student_oracle as so
    from oracle import teacher_oracle as to
    from oracle import teacher_train_oracle as tt

    from glnn_amd import data, ops
    from glnn_amd.graph import FullNeighborLoader, MultiLayerNeighborSampler, NodeDataLoader
    from glnn_amd.models import Model
    from glnn_amd.student import StudentEngine
    from glnn_amd.teacher import TeacherEngine

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (MI355X)")
    dev = "cuda:0"
    torch.manual_seed(0)
    # ---- teacher: arxiv-shaped (scaled) SAGE layer-wise inference -------------------------------
    g = data.make_graph("ogbn-arxiv", seed=0, device="cpu", scale=0.02)
    n = g.n_dst
    feats = torch.randn(n, 128)
    model = Model(dict(model_name="SAGE", num_layers=3, feat_dim=128, hidden_dim=256, label_dim=40, dropout_ratio=0.2,
                       norm_type="batch", device=dev))
    model.eval()
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    layers = [dict(weight=sd[f"encoder.layers.{i}.fc_neigh.weight"], bias=sd[f"encoder.layers.{i}.fc_neigh.bias"]) for i in range(3)]
    norms = [dict(weight=sd[f"encoder.norms.{i}.weight"], bias=sd[f"encoder.norms.{i}.bias"],
                  running_mean=sd[f"encoder.norms.{i}.running_mean"], running_var=sd[f"encoder.norms.{i}.running_var"]) for i in range(2)]
    want = to.sage_inference(g.indptr.numpy(), g.indices.numpy(), feats.numpy(), layers, norms)
    got = model.inference(FullNeighborLoader(g.to(dev), 512), feats.to(dev)).cpu().numpy()
    err_t = float(np.abs(got - want).max())
    assert err_t < 1e-4, f"teacher forward mismatch {err_t}"
    # ---- student: one fused distillation step (KL) ------------------------------------------------
    student = Model(dict(model_name="MLP", num_layers=3, feat_dim=128, hidden_dim=256, label_dim=40, dropout_ratio=0.0,
                         norm_type="batch", device=dev))
    sd0 = {k: v.cpu().numpy() for k, v in student.state_dict().items()}
    opt = torch.optim.Adam(student.parameters(), lr=0.01, weight_decay=0.0)
    out_t = torch.log_softmax(torch.randn(n, 40), dim=1)
    bsz = 512
    eng = StudentEngine(student, opt, bsz)
    idx = torch.randperm(n)[:bsz]
    student.train()
    eng.step(ops.as_feat(feats.to(dev)), idx.to(dev), ops.LOSS_KL, ops.as_feat(out_t.to(dev)), 1.0)
    st = so.MLPState(sd0, 3, "batch")
    logits, cache = so.mlp_forward(st, feats.numpy()[idx.numpy()], training=True)
    loss, dlogits = so.loss_and_dlogits(logits, out_t.numpy()[idx.numpy()], "kl", 1.0)
    grads = so.mlp_backward(st, cache, dlogits)
    err_l = abs(float(loss) - eng.loss_out.item())
    err_g = max(float(np.abs(p.grad.cpu().numpy() - gr).max()) for p, gr in zip(student.parameters(), grads))
    assert err_l < 1e-4 and err_g < 1e-4, f"student step mismatch loss {err_l} grad {err_g}"
    # ---- teacher TRAINING: one sampled-block GraphSAGE step (blocks built on the device) vs the numpy oracle on the same blocks
    gd = g.to(dev)
    tm = Model(dict(model_name="SAGE", num_layers=2, feat_dim=128, hidden_dim=64, label_dim=40, dropout_ratio=0.0,
                    norm_type="batch", device=dev))
    sd_t = {k: v.cpu().numpy() for k, v in tm.state_dict().items()}
    topt = torch.optim.Adam(tm.parameters(), lr=0.01, weight_decay=5e-4)
    labels = torch.randint(0, 40, (n,))
    loader = NodeDataLoader(gd, torch.arange(256), MultiLayerNeighborSampler([5, 5]), batch_size=256)
    input_nodes, output_nodes, blocks = next(iter(loader))
    tm.train()
    teng = TeacherEngine(tm, topt)
    teng.step_sage(blocks, ops.as_feat(feats.to(dev)), labels.to(dev), output_nodes, 1.0, input_nodes=input_nodes)
    st_t = tt.TeacherState(sd_t, "sage", 2, "batch")
    nb = [(b.indptr.cpu().numpy(), b.indices.cpu().numpy(), b.num_src_nodes()) for b in blocks]
    loss_t, _ = tt.train_sage(st_t, [(input_nodes.cpu().numpy(), output_nodes.cpu().numpy(), nb)], feats.numpy(), labels.numpy(), 0.01, 5e-4)
    err_tl = abs(loss_t - teng.loss_out.item())
    err_tw = max(float(np.abs(p.detach().cpu().numpy() - w).max()) for p, w in zip(tm.parameters(), st_t.params()))
    assert err_tl < 1e-4 and err_tw < 1e-4, f"teacher training step mismatch loss {err_tl} params {err_tw}"
    print(f"smoke ok: teacher max|err| {err_t:.2e}, student loss err {err_l:.2e}, grad err {err_g:.2e}, "
          f"teacher-train loss err {err_tl:.2e}, param err {err_tw:.2e}")
