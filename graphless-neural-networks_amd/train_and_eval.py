"""The reference's train / eval / distill loops (reference train_and_eval.py) with the same function
names, arguments and return values, re-pointed at the HIP hot path.

  * train_mini_batch / evaluate_mini_batch (the student's loops, :59-86 / :108-136) run on the fused
    StudentEngine; the (model, criterion, optimizer) triple must be what train_student.py:274-279 builds --
    anything else raises (one shared eligibility check: glnn_amd.student.student_supported).
  * evaluate (:89-105) calls Model.inference -> SAGE.inference on the aggregation kernel.
  * The per-step `loss.item()` host sync of the reference (:80) is replaced by a device-side running sum
    read ONCE per pass; the returned mean of per-batch losses is the same number.
  * Teacher TRAINING (`train_sage` on fan-out-sampled blocks from glnn_amd.graph.NodeDataLoader, `train` for the
    full-graph GCN) runs on glnn_amd.teacher.TeacherEngine: blocks built on the device, forward, loss, backward over the
    transposed blocks and Adam as one launch sequence without autograd or host syncs.
  There is no eager / CPU path in this module: inputs that are not on the GPU raise."""
import copy

import numpy as np
import torch

from . import ops
from .graph import FullNeighborLoader, MultiLayerFullNeighborSampler, MultiLayerNeighborSampler, NodeDataLoader
from . import teacher
from .student import get_engine, student_supported
from .utils import set_seed


EVAL_BLOCK_ROWS = 1 << 19


def train(model, data, feats, labels, criterion, optimizer, idx_train, lamb=1):
    """GNN full-batch training step (reference train_and_eval.py:12-29; `data` is the whole graph): one
    TeacherEngine.step_gcn -- forward, NLL over idx_train, backward over the transposed graph, Adam -- on libglnn_hip.so.
    Returns the unscaled loss like the reference's `loss.item()`."""
    teacher.check_supported(model, criterion, optimizer)
    if "GCN" not in model.model_name:
        raise NotImplementedError("train(): the full-graph step is implemented for the GCN teacher (SAGE trains with train_sage)")
    model.train()
    eng = teacher.get_engine(model, optimizer)
    eng.step_gcn(data, feats, labels, idx_train, float(lamb))
    eng.sync_optimizer_state()
    return eng.loss_out.item()


def train_sage(model, dataloader, feats, labels, criterion, optimizer, lamb=1):
    """Sampled-block GraphSAGE training (reference train_and_eval.py:32-56): one TeacherEngine.step_sage per batch of
    `dataloader` (glnn_amd.graph.NodeDataLoader: blocks sampled and relabelled on the device; the outermost block gathers
    straight from `feats`, so `feats[input_nodes]` is never materialised).  The per-step `loss.item()` of the reference
    (:49) is a device-side running sum read ONCE per epoch; the returned mean of the per-batch losses is the same number."""
    teacher.check_supported(model, criterion, optimizer)
    if "SAGE" not in model.model_name:
        raise NotImplementedError("train_sage(): GraphSAGE teachers only")
    model.train()
    eng = teacher.get_engine(model, optimizer)
    eng.loss_accum.zero_()
    steps = 0
    # the engine gathers layer 0 straight from `feats` through the outermost block's global ids: our loader then builds that block
    # without its frontier table / local relabelling and yields input_nodes = None (60 % of the sampler's device time; other loaders: no-op)
    had = getattr(dataloader, "global_first_block", None)
    if had is not None:
        dataloader.global_first_block = True
    try:
        for input_nodes, output_nodes, blocks in dataloader:
            eng.step_sage(blocks, feats, labels, output_nodes, float(lamb), input_nodes=input_nodes)
            steps += 1
    finally:
        if had is not None:
            dataloader.global_first_block = had
    eng.sync_optimizer_state()
    return eng.loss_accum.item() / max(steps, 1)


def _batch_indices(n, batch_size):
    """reference train_and_eval.py:65-71: CPU randperm, remainder dropped, [nb, B] view."""
    num_batches = max(1, n // batch_size)
    idx_batch = ops.randperm_cpu(n)[: num_batches * batch_size]
    idx_batch = idx_batch.view(1, -1) if num_batches == 1 else idx_batch.view(num_batches, batch_size)
    return num_batches, idx_batch


def train_mini_batch(model, feats, labels, batch_size, criterion, optimizer, lamb=1):
    """One pass of the MLP over `feats` in random mini-batches (reference train_and_eval.py:59-86) on the fused
    StudentEngine.  `labels` is int64 [N] with NLLLoss or fp32 teacher log-probs [N, C] with KLDivLoss(log_target).
    Anything other than the (MLP, NLLLoss | KLDivLoss(batchmean, log_target), Adam) triple train_student.py:274-279
    builds raises: there is no generic / eager loop behind this function."""
    kind = student_supported(model, criterion, optimizer, feats, labels)
    model.train()
    num_batches, idx_batch = _batch_indices(feats.shape[0], batch_size)
    eng = get_engine(model, optimizer, idx_batch.shape[1])
    x = ops.as_feat(feats)
    target = labels if kind == ops.LOSS_NLL else ops.as_feat(labels)
    idx_dev = idx_batch.to(feats.device)           # one H2D copy per pass
    eng.loss_accum.zero_()
    for i in range(num_batches):
        eng.step(x, idx_dev[i], kind, target, float(lamb))
    eng.sync_optimizer_state()
    return eng.loss_accum.item() / num_batches      # the only host sync of the pass


def evaluate(model, data, feats, labels, criterion, evaluator, idx_eval=None):
    """reference train_and_eval.py:89-105"""
    model.eval()
    with torch.no_grad():
        logits = model.inference(data, feats)
        out = ops.log_softmax(logits)
        if idx_eval is None:
            loss = _apply_criterion(criterion, out, labels)
            score = evaluator(out, labels)
        else:
            loss = _apply_criterion(criterion, out[idx_eval], labels[idx_eval])
            score = evaluator(out[idx_eval], labels[idx_eval])
    return out, loss.item(), score


def _apply_criterion(criterion, out, labels):
    """criterion(out, labels) -- the reference passes torch.nn.NLLLoss() (train_teacher.py / train_student.py).  torch's CUDA kernel for it
    reduces a [N, C] matrix in ONE workgroup: 1.46 ms for pokec's 1.6 M rows, a third of an evaluate call.  A plain default NLLLoss is
    therefore evaluated as -mean(out[i, labels[i]]) with a gather and a grid-wide mean (same value to fp32 summation order); anything else
    (weights, ignore_index hits, another reduction, another callable) goes to the callable itself."""
    if (type(criterion) is torch.nn.NLLLoss and criterion.reduction == "mean" and criterion.weight is None and out.dim() == 2 and
            labels.dim() == 1 and labels.dtype == torch.int64 and out.shape[0] >= 65536):
        if not bool((labels == criterion.ignore_index).any()):
            return -out.gather(1, labels.view(-1, 1)).mean()
    return criterion(out, labels)


def evaluate_mini_batch(model, feats, labels, criterion, batch_size, evaluator, idx_eval=None):
    """reference train_and_eval.py:108-136.  Eval-mode rows are independent, so the matrix goes through ONE chain of GEMMs
    per big row block instead of ceil(N/B) mini-batches; the output is the same [N, C]."""
    ops._need_cuda(feats)
    model.eval()
    with torch.no_grad():
        # row blocks bounded so that a wide student over millions of rows does not materialise tens of GB of activations
        blk = max(int(batch_size), EVAL_BLOCK_ROWS)
        out_all = torch.empty((feats.shape[0], model.encoder.layers[-1].out_features), dtype=torch.float32, device=feats.device)
        x = ops.as_feat(feats)                     # padded ONCE (and remembered across calls), not per row block: the slices of x are float4 rows
        for s0 in range(0, feats.shape[0], blk):
            ops.log_softmax(model.inference(None, x[s0:s0 + blk]), out=out_all[s0:s0 + blk])
        if idx_eval is None:
            loss = _apply_criterion(criterion, out_all, labels)
            score = evaluator(out_all, labels)
        else:
            loss = _apply_criterion(criterion, out_all[idx_eval], labels[idx_eval])
            score = evaluator(out_all[idx_eval], labels[idx_eval])
    return out_all, loss.item(), score


def _early_stop_loop(conf, model, logger, loss_and_score, train_epoch, eval_epoch):
    """Shared epoch loop of the reference's run_* / distill_run_* drivers: evaluate every
    `eval_interval`, keep the best-validation state_dict in RAM, stop after `patience` bad evals
    (reference train_and_eval.py:215-271, :558-594)."""
    best_epoch, best_score_val, count = 0, 0, 0
    state = copy.deepcopy(model.state_dict())
    for epoch in range(1, conf["max_epoch"] + 1):
        loss = train_epoch()
        if epoch % conf["eval_interval"] == 0:
            row, score_val, msg = eval_epoch()
            logger.debug(f"Ep {epoch:3d} | loss: {loss:.4f} | {msg}")
            loss_and_score += [[epoch] + row]
            if score_val >= best_score_val:
                best_epoch, best_score_val = epoch, score_val
                state = copy.deepcopy(model.state_dict())
                count = 0
            else:
                count += 1
        if count == conf["patience"] or epoch == conf["max_epoch"]:
            break
    model.load_state_dict(state)
    return best_epoch


def run_transductive(conf, model, g, feats, labels, indices, criterion, evaluator, optimizer, logger, loss_and_score):
    """Teacher (or plain-MLP) training + eval, transductive (reference train_and_eval.py:144-287).
    SAGE: training on fan-out-sampled blocks (GPU sampler), evaluation = layer-wise full-neighbour inference."""
    set_seed(conf["seed"])
    device = conf["device"]
    batch_size = conf["batch_size"]
    idx_train, idx_val, idx_test = indices
    feats, labels = feats.to(device), labels.to(device)
    idx_train, idx_val, idx_test = idx_train.to(device), idx_val.to(device), idx_test.to(device)
    is_mlp, is_sage = "MLP" in model.model_name, "SAGE" in model.model_name
    if is_sage:
        # the reference's two loaders (train_and_eval.py:176-205), sampling on the GPU-resident CSR
        g = g.to(device)
        sampler = MultiLayerNeighborSampler([int(fanout) for fanout in str(conf["fan_out"]).split(",")])
        data = NodeDataLoader(g, idx_train, sampler, batch_size=batch_size, shuffle=True, drop_last=False)
        data_eval = NodeDataLoader(g, torch.arange(g.num_nodes()), MultiLayerFullNeighborSampler(1), batch_size=batch_size,
                                   shuffle=False, drop_last=False)
    elif is_mlp:
        feats_train, labels_train = feats[idx_train], labels[idx_train]
        feats_val, labels_val = feats[idx_val], labels[idx_val]
        feats_test, labels_test = feats[idx_test], labels[idx_test]
    else:
        g = g.to(device)
        data = data_eval = g

    def train_epoch():
        if is_sage:
            return train_sage(model, data, feats, labels, criterion, optimizer)
        if is_mlp:
            return train_mini_batch(model, feats_train, labels_train, batch_size, criterion, optimizer)
        return train(model, data, feats, labels, criterion, optimizer, idx_train)

    def eval_epoch():
        if is_mlp:
            _, l_tr, s_tr = evaluate_mini_batch(model, feats_train, labels_train, criterion, batch_size, evaluator)
            _, l_va, s_va = evaluate_mini_batch(model, feats_val, labels_val, criterion, batch_size, evaluator)
            _, l_te, s_te = evaluate_mini_batch(model, feats_test, labels_test, criterion, batch_size, evaluator)
        else:
            out, l_tr, s_tr = evaluate(model, data_eval, feats, labels, criterion, evaluator, idx_train)
            l_va, s_va = criterion(out[idx_val], labels[idx_val]).item(), evaluator(out[idx_val], labels[idx_val])
            l_te, s_te = criterion(out[idx_test], labels[idx_test]).item(), evaluator(out[idx_test], labels[idx_test])
        return [l_tr, l_va, l_te, s_tr, s_va, s_te], s_va, f"s_train: {s_tr:.4f} | s_val: {s_va:.4f} | s_test: {s_te:.4f}"

    best_epoch = _early_stop_loop(conf, model, logger, loss_and_score, train_epoch, eval_epoch)
    if is_mlp:
        out, _, score_val = evaluate_mini_batch(model, feats, labels, criterion, batch_size, evaluator, idx_val)
    else:
        out, _, score_val = evaluate(model, data_eval, feats, labels, criterion, evaluator, idx_val)
    score_test = evaluator(out[idx_test], labels[idx_test])
    logger.info(f"Best valid model at epoch: {best_epoch: 3d}, score_val: {score_val :.4f}, score_test: {score_test :.4f}")
    return out, score_val, score_test


def distill_run_transductive(conf, model, feats, labels, out_t_all, distill_indices, criterion_l, criterion_t,
                             evaluator, optimizer, logger, loss_and_score):
    """Distillation, transductive (reference train_and_eval.py:520-606): per epoch a hard-label pass
    weighted lamb, then a soft-label pass weighted 1-lamb, each with its own randperm and optimiser steps."""
    set_seed(conf["seed"])
    device = conf["device"]
    batch_size = conf["batch_size"]
    lamb = conf["lamb"]
    idx_l, idx_t, idx_val, idx_test = [i.to(device) for i in distill_indices]
    feats, labels, out_t_all = feats.to(device), labels.to(device), out_t_all.to(device)
    feats_l, labels_l = feats[idx_l], labels[idx_l]
    feats_t, out_t = feats[idx_t], out_t_all[idx_t]
    feats_val, labels_val = feats[idx_val], labels[idx_val]
    feats_test, labels_test = feats[idx_test], labels[idx_test]

    def train_epoch():
        loss_l = train_mini_batch(model, feats_l, labels_l, batch_size, criterion_l, optimizer, lamb)
        loss_t = train_mini_batch(model, feats_t, out_t, batch_size, criterion_t, optimizer, 1 - lamb)
        return loss_l + loss_t

    def eval_epoch():
        _, l_l, s_l = evaluate_mini_batch(model, feats_l, labels_l, criterion_l, batch_size, evaluator)
        _, l_va, s_va = evaluate_mini_batch(model, feats_val, labels_val, criterion_l, batch_size, evaluator)
        _, l_te, s_te = evaluate_mini_batch(model, feats_test, labels_test, criterion_l, batch_size, evaluator)
        return [l_l, l_va, l_te, s_l, s_va, s_te], s_va, f"s_l: {s_l:.4f} | s_val: {s_va:.4f} | s_test: {s_te:.4f}"

    best_epoch = _early_stop_loop(conf, model, logger, loss_and_score, train_epoch, eval_epoch)
    out, _, score_val = evaluate_mini_batch(model, feats, labels, criterion_l, batch_size, evaluator, idx_val)
    score_test = evaluator(out[idx_test], labels_test)
    logger.info(f"Best valid model at epoch: {best_epoch: 3d}, score_val: {score_val :.4f}, score_test: {score_test :.4f}")
    return out, score_val, score_test


def run_inductive(conf, model, g, feats, labels, indices, criterion, evaluator, optimizer, logger, loss_and_score):
    """Teacher / plain-MLP training under the inductive ("production") setting (reference
    train_and_eval.py:290-512): train on the observed subgraph `obs_g = g.subgraph(idx_obs)`, evaluate on the
    observed test nodes AND, with the full graph, on the held-out inductive test nodes."""
    set_seed(conf["seed"])
    device = conf["device"]
    batch_size = conf["batch_size"]
    obs_idx_train, obs_idx_val, obs_idx_test, idx_obs, idx_test_ind = [i.to(device) for i in indices]
    feats, labels = feats.to(device), labels.to(device)
    obs_feats, obs_labels = feats[idx_obs], labels[idx_obs]
    is_mlp, is_sage = "MLP" in model.model_name, "SAGE" in model.model_name
    if is_mlp:
        feats_train, labels_train = obs_feats[obs_idx_train], obs_labels[obs_idx_train]
        feats_val, labels_val = obs_feats[obs_idx_val], obs_labels[obs_idx_val]
        feats_tt, labels_tt = obs_feats[obs_idx_test], obs_labels[obs_idx_test]
        feats_ti, labels_ti = feats[idx_test_ind], labels[idx_test_ind]
    else:
        g = g.to(device)
        obs_g = g.subgraph(idx_obs)
        if is_sage:
            sampler = MultiLayerNeighborSampler([int(fanout) for fanout in str(conf["fan_out"]).split(",")])
            obs_data = NodeDataLoader(obs_g, obs_idx_train, sampler, batch_size=batch_size, shuffle=True, drop_last=False)
            full = MultiLayerFullNeighborSampler(1)
            obs_data_eval = NodeDataLoader(obs_g, torch.arange(obs_g.num_nodes()), full, batch_size=batch_size)
            data_eval = NodeDataLoader(g, torch.arange(g.num_nodes()), full, batch_size=batch_size)
        else:
            obs_data = obs_data_eval = obs_g
            data_eval = g

    def train_epoch():
        if is_sage:
            return train_sage(model, obs_data, obs_feats, obs_labels, criterion, optimizer)
        if is_mlp:
            return train_mini_batch(model, feats_train, labels_train, batch_size, criterion, optimizer)
        return train(model, obs_data, obs_feats, obs_labels, criterion, optimizer, obs_idx_train)

    def eval_epoch():
        if is_mlp:
            _, l_tr, s_tr = evaluate_mini_batch(model, feats_train, labels_train, criterion, batch_size, evaluator)
            _, l_va, s_va = evaluate_mini_batch(model, feats_val, labels_val, criterion, batch_size, evaluator)
            _, l_tt, s_tt = evaluate_mini_batch(model, feats_tt, labels_tt, criterion, batch_size, evaluator)
            _, l_ti, s_ti = evaluate_mini_batch(model, feats_ti, labels_ti, criterion, batch_size, evaluator)
        else:
            obs_out, l_tr, s_tr = evaluate(model, obs_data_eval, obs_feats, obs_labels, criterion, evaluator, obs_idx_train)
            l_va, s_va = criterion(obs_out[obs_idx_val], obs_labels[obs_idx_val]).item(), evaluator(obs_out[obs_idx_val], obs_labels[obs_idx_val])
            l_tt, s_tt = criterion(obs_out[obs_idx_test], obs_labels[obs_idx_test]).item(), evaluator(obs_out[obs_idx_test], obs_labels[obs_idx_test])
            _, l_ti, s_ti = evaluate(model, data_eval, feats, labels, criterion, evaluator, idx_test_ind)
        return ([l_tr, l_va, l_tt, l_ti, s_tr, s_va, s_tt, s_ti], s_va,
                f"s_train: {s_tr:.4f} | s_val: {s_va:.4f} | s_tt: {s_tt:.4f} | s_ti: {s_ti:.4f}")

    best_epoch = _early_stop_loop(conf, model, logger, loss_and_score, train_epoch, eval_epoch)
    if is_mlp:
        obs_out, _, score_val = evaluate_mini_batch(model, obs_feats, obs_labels, criterion, batch_size, evaluator, obs_idx_val)
        out, _, score_test_ind = evaluate_mini_batch(model, feats, labels, criterion, batch_size, evaluator, idx_test_ind)
    else:
        obs_out, _, score_val = evaluate(model, obs_data_eval, obs_feats, obs_labels, criterion, evaluator, obs_idx_val)
        out, _, score_test_ind = evaluate(model, data_eval, feats, labels, criterion, evaluator, idx_test_ind)
    score_test_tran = evaluator(obs_out[obs_idx_test], obs_labels[obs_idx_test])
    out[idx_obs] = obs_out
    logger.info(f"Best valid model at epoch: {best_epoch :3d}, score_val: {score_val :.4f}, score_test_tran: {score_test_tran :.4f}, score_test_ind: {score_test_ind :.4f}")
    return out, score_val, score_test_tran, score_test_ind


def distill_run_inductive(conf, model, feats, labels, out_t_all, distill_indices, criterion_l, criterion_t, evaluator,
                          optimizer, logger, loss_and_score):
    """Distillation under the inductive setting (reference train_and_eval.py:609-742): the student only sees
    the observed nodes' features and soft labels; it is scored on observed-test and inductive-test nodes."""
    set_seed(conf["seed"])
    device = conf["device"]
    batch_size = conf["batch_size"]
    lamb = conf["lamb"]
    obs_idx_l, obs_idx_t, obs_idx_val, obs_idx_test, idx_obs, idx_test_ind = [i.to(device) for i in distill_indices]
    feats, labels, out_t_all = feats.to(device), labels.to(device), out_t_all.to(device)
    obs_feats, obs_labels, obs_out_t = feats[idx_obs], labels[idx_obs], out_t_all[idx_obs]
    feats_l, labels_l = obs_feats[obs_idx_l], obs_labels[obs_idx_l]
    feats_t, out_t = obs_feats[obs_idx_t], obs_out_t[obs_idx_t]
    feats_val, labels_val = obs_feats[obs_idx_val], obs_labels[obs_idx_val]
    feats_tt, labels_tt = obs_feats[obs_idx_test], obs_labels[obs_idx_test]
    feats_ti, labels_ti = feats[idx_test_ind], labels[idx_test_ind]

    def train_epoch():
        loss_l = train_mini_batch(model, feats_l, labels_l, batch_size, criterion_l, optimizer, lamb)
        loss_t = train_mini_batch(model, feats_t, out_t, batch_size, criterion_t, optimizer, 1 - lamb)
        return loss_l + loss_t

    def eval_epoch():
        _, l_l, s_l = evaluate_mini_batch(model, feats_l, labels_l, criterion_l, batch_size, evaluator)
        _, l_va, s_va = evaluate_mini_batch(model, feats_val, labels_val, criterion_l, batch_size, evaluator)
        _, l_tt, s_tt = evaluate_mini_batch(model, feats_tt, labels_tt, criterion_l, batch_size, evaluator)
        _, l_ti, s_ti = evaluate_mini_batch(model, feats_ti, labels_ti, criterion_l, batch_size, evaluator)
        return ([l_l, l_va, l_tt, l_ti, s_l, s_va, s_tt, s_ti], s_va,
                f"s_l: {s_l:.4f} | s_val: {s_va:.4f} | s_tt: {s_tt:.4f} | s_ti: {s_ti:.4f}")

    best_epoch = _early_stop_loop(conf, model, logger, loss_and_score, train_epoch, eval_epoch)
    obs_out, _, score_val = evaluate_mini_batch(model, obs_feats, obs_labels, criterion_l, batch_size, evaluator, obs_idx_val)
    out, _, score_test_ind = evaluate_mini_batch(model, feats, labels, criterion_l, batch_size, evaluator, idx_test_ind)
    score_test_tran = evaluator(obs_out[obs_idx_test], labels_tt)
    out[idx_obs] = obs_out
    logger.info(f"Best valid model at epoch: {best_epoch: 3d} score_val: {score_val :.4f}, score_test_tran: {score_test_tran :.4f}, score_test_ind: {score_test_ind :.4f}")
    return out, score_val, score_test_tran, score_test_ind
