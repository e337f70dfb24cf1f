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
    """reference utils.py:19-26"""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_training_config(config_path, model_name, dataset):
    """reference utils.py:29-41: `global` section overlaid by [dataset][model_name]; injects model_name."""
    with open(config_path, "r") as conf:
        full_config = yaml.load(conf, Loader=yaml.FullLoader)
    dataset_specific_config = full_config["global"]
    model_specific_config = full_config[dataset][model_name]
    if model_specific_config is not None:
        specific_config = dict(dataset_specific_config, **model_specific_config)
    else:
        specific_config = dataset_specific_config
    specific_config["model_name"] = model_name
    return specific_config


def check_writable(path, overwrite=True):
    if not os.path.exists(path):
        os.makedirs(path)
    elif overwrite:
        shutil.rmtree(path)
        os.makedirs(path)


def check_readable(path):
    if not os.path.exists(path):
        raise ValueError(f"No such file or directory! {path}")


def get_logger(filename, console_log=False, log_level=logging.INFO):
    """reference utils.py:64-85 (US/Pacific timestamps when pytz is importable)."""
    logger = logging.getLogger(__name__)
    logger.propagate = False
    logger.setLevel(log_level)
    for hdlr in logger.handlers[:]:
        logger.removeHandler(hdlr)
    formatter = logging.Formatter("%(asctime)s: %(message)s", datefmt="%b%d %H-%M-%S")
    try:
        import pytz
        tz = pytz.timezone("US/Pacific")
        formatter.converter = lambda *a: datetime.now(tz).timetuple()
    except Exception:  # pragma: no cover
        pass
    file_handler = logging.FileHandler(filename)
    file_handler.setFormatter(formatter)
    logger.addHandler(file_handler)
    if console_log:
        console_handler = logging.StreamHandler()
        console_handler.setFormatter(formatter)
        logger.addHandler(console_handler)
    return logger


def idx_split(idx, ratio, seed=0):
    """reference utils.py:88-100"""
    set_seed(seed)
    n = len(idx)
    cut = int(n * ratio)
    idx_idx_shuffle = torch.randperm(n)
    idx1_idx, idx2_idx = idx_idx_shuffle[:cut], idx_idx_shuffle[cut:]
    return idx[idx1_idx], idx[idx2_idx]


def graph_split(idx_train, idx_val, idx_test, rate, seed):
    """reference utils.py:103-127: hide `rate` of the test nodes for the inductive evaluation."""
    idx_test_ind, idx_test_tran = idx_split(idx_test, rate, seed)
    idx_obs = torch.cat([idx_train, idx_val, idx_test_tran])
    n1, n2 = idx_train.shape[0], idx_val.shape[0]
    obs_idx_all = torch.arange(idx_obs.shape[0])
    return obs_idx_all[:n1], obs_idx_all[n1:n1 + n2], obs_idx_all[n1 + n2:], idx_obs, idx_test_ind


def get_evaluator(dataset):
    """The EFFECTIVE reference evaluator: the second definition (utils.py:151-156) shadows the OGB one."""
    def evaluator(out, labels):
        pred = out.argmax(1)
        return pred.eq(labels).float().mean().item()
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
