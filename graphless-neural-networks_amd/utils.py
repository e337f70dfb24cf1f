"""Host-side helpers with the reference's names and behaviour (reference utils.py).  Pure index /
config logic stays Python; `feature_prop` runs on the HIP aggregation kernel."""
import logging
import os
import random
import shutil
from datetime import datetime

import numpy as np
import torch
import yaml

from . import ops

CPF_data = ["cora", "citeseer", "pubmed", "a-computer", "a-photo"]
OGB_data = ["ogbn-arxiv", "ogbn-products"]


def set_seed(seed):
    """Seed every RNG the hot path draws from (reference utils.py:19-26): torch CPU (randperm of train_mini_batch,
    parameter init), numpy, python; all CUDA generators when a GPU is present."""
    for seeder in (torch.manual_seed, np.random.seed, random.seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_training_config(config_path, model_name, dataset):
    """Hyper-parameters of (dataset, model): the YAML's `global` section overlaid by [dataset][model_name], plus
    `model_name` itself (reference utils.py:29-41).  An empty model section means "global only"."""
    with open(config_path) as fh:
        cfg = yaml.safe_load(fh)
    merged = dict(cfg["global"])
    merged.update(cfg[dataset][model_name] or {})
    merged["model_name"] = model_name
    return merged


def check_writable(path, overwrite=True):
    """Make sure `path` is a directory we can write to; overwrite=True empties an existing one."""
    if overwrite and os.path.exists(path):
        shutil.rmtree(path)
    os.makedirs(path, exist_ok=True)


def check_readable(path):
    if not os.path.exists(path):
        raise ValueError(f"No such file or directory! {path}")


def _pacific_now():
    try:
        import pytz
        return datetime.now(pytz.timezone("US/Pacific"))
    except Exception:  # pragma: no cover - pytz missing: local time
        return datetime.now()


def get_logger(filename, console_log=False, log_level=logging.INFO):
    """File logger (+ optional console) with the reference's line format and US/Pacific clock (utils.py:59-85);
    handlers of a previous run are dropped so repeated runs do not duplicate lines."""
    logger = logging.getLogger(__name__)
    logger.propagate = False
    logger.setLevel(log_level)
    logger.handlers.clear()
    fmt = logging.Formatter("%(asctime)s: %(message)s", datefmt="%b%d %H-%M-%S")
    fmt.converter = lambda *_: _pacific_now().timetuple()
    sinks = [logging.FileHandler(filename)] + ([logging.StreamHandler()] if console_log else [])
    for sink in sinks:
        sink.setFormatter(fmt)
        logger.addHandler(sink)
    return logger


def idx_split(idx, ratio, seed=0):
    """Random two-way split of `idx` into int(n*ratio) and the rest, seeded (reference utils.py:88-100: the split is
    a torch.randperm drawn right after set_seed(seed), so it is reproducible across the teacher and student runs)."""
    set_seed(seed)
    order = torch.randperm(len(idx))
    cut = int(len(idx) * ratio)
    return idx[order[:cut]], idx[order[cut:]]


def graph_split(idx_train, idx_val, idx_test, rate, seed):
    """Inductive ("production") split (reference utils.py:103-127): `rate` of the test nodes are hidden
    (idx_test_ind); the observed graph holds train + val + remaining test nodes IN THAT ORDER, so the obs_* index
    sets are simply consecutive ranges of the observed numbering.
    Returns (obs_idx_train, obs_idx_val, obs_idx_test, idx_obs, idx_test_ind)."""
    idx_test_ind, idx_test_tran = idx_split(idx_test, rate, seed)
    idx_obs = torch.cat([idx_train, idx_val, idx_test_tran])
    n_tr, n_va = len(idx_train), len(idx_val)
    obs = torch.arange(len(idx_obs))
    return obs[:n_tr], obs[n_tr:n_tr + n_va], obs[n_tr + n_va:], idx_obs, idx_test_ind


def get_evaluator(dataset):
    """Plain argmax accuracy for every dataset: the EFFECTIVE reference evaluator -- its second definition
    (utils.py:151-156) shadows the OGB one (utils.py:130-148)."""
    def evaluator(out, labels):
        return (out.argmax(1) == labels).float().mean().item()
    return evaluator


def feature_prop(feats, g, k):
    """(D^-1/2 A D^-1/2)^k X, D = in-degree.clamp(1) (reference utils.py:171-189), one fused
    glnn_spmm_csr_f32 per hop (row_scale = col_scale = D^-1/2)."""
    assert feats.shape[0] == g.num_nodes()
    norm, _ = g.degree_norms()
    x = ops.as_feat(feats)
    for _ in range(k):
        x = ops.spmm(g.indptr, g.indices, x, g.num_dst_nodes(), ops.AGG_SUM, row_scale=norm, col_scale=norm)
    return x


def compute_min_cut_loss(g, out):
    """tr(S^T A S) / tr(S^T D S) (reference utils.py:159-168) WITHOUT the dense N x N adjacency:
    tr(S^T A S) = sum(S * (A S)) with A S from the aggregation kernel; D = in-degrees."""
    s = ops.as_feat(out.exp().to(g.device))
    a_s = ops.spmm(g.indptr, g.indices, s, g.num_dst_nodes(), ops.AGG_SUM)
    num = (s[: g.num_dst_nodes()] * a_s).sum()
    den = (g.in_degrees().to(torch.float32).unsqueeze(1) * s[: g.num_dst_nodes()] ** 2).sum()
    return (num / den).item()
