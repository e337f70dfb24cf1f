"""Command-line surface of the reference's two entry points, kept flag-for-flag:
`train_teacher.py` (reference train_teacher.py:21-149, run :152-314) and `train_student.py`
(reference train_student.py:22-165, run :168-384).  Same flags and defaults, same YAML-overrides-CLI merge
(`conf = dict(args.__dict__, **conf)`, train_teacher.py:225-229), same output directory layout and artefacts
(out.npz = log-probs of ALL nodes, loss_and_score.npz, model.pth, min_cut_loss, exp_results).

The flag tables below are data; both parsers are built from them."""
import argparse
from pathlib import Path

import numpy as np
import torch
from torch import optim

from .dataloader import load_data, load_out_t
from .models import Model
from .train_and_eval import distill_run_inductive, distill_run_transductive, run_inductive, run_transductive
from .utils import (check_readable, check_writable, compute_min_cut_loss, feature_prop, get_evaluator, get_logger,
                    get_training_config, graph_split, set_seed)

_DEFAULT_CONF = str(Path(__file__).resolve().parent.parent.joinpath("train.conf.yaml"))

# (flag, type-or-"store_true", default, help)
_COMMON = [
    ("device", int, -1, "CUDA device, -1 means CPU"), ("seed", int, 0, "Random seed"),
    ("log_level", int, 20, "Logger levels for run {10: DEBUG, 20: INFO, 30: WARNING}"),
    ("console_log", "store_true", False, "Set to True to display log info in console"),
    ("output_path", str, "outputs", "Path to save outputs"), ("num_exp", int, 1, "Repeat how many experiments"),
    ("exp_setting", str, "tran", "Experiment setting, one of [tran, ind]"),
    ("eval_interval", int, 1, "Evaluate once per how many epochs"),
    ("save_results", "store_true", False, "Save the loss curves, trained model, and min-cut loss"),
    ("dataset", str, "cora", "Dataset"), ("data_path", str, "./data", "Path to data"),
    ("labelrate_train", int, 20, "How many labeled data per class as train set"),
    ("labelrate_val", int, 30, "How many labeled data per class in valid set"),
    ("split_idx", int, 0, "For Non-Homo datasets only, one of [0,1,2,3,4]"),
    ("model_config_path", str, _DEFAULT_CONF, "Path to model configuration"),
    ("teacher", str, "SAGE", "Teacher model"), ("num_layers", int, 2, "Model number of layers"),
    ("hidden_dim", int, 128, "Model hidden layer dimensions"), ("dropout_ratio", float, 0, ""),
    ("norm_type", str, "none", "One of [none, batch, layer]"), ("batch_size", int, 512, ""),
    ("fan_out", str, "5,5", "Number of samples for each layer in SAGE. Length = num_layers"),
    ("num_workers", int, 0, "Number of workers for sampler"), ("learning_rate", float, 0.01, ""),
    ("weight_decay", float, 0.0005, ""), ("max_epoch", int, 500, "Maximum number of epochs"),
    ("patience", int, 50, "Early stop if the validation score does not improve for this many epochs"),
    ("feature_noise", float, 0, "add white noise to features for analysis, value in [0, 1]"),
    ("split_rate", float, 0.2, "Rate for graph split, see graph_split"),
    ("compute_min_cut", "store_true", False, "Compute and store the min-cut loss"),
    ("feature_aug_k", int, 0, "Augment node features by aggregating feature_aug_k-hop neighbour features"),
]
_STUDENT_ONLY = [
    ("student", str, "MLP", "Student model"),
    ("lamb", float, 0, "Parameter balancing hard-label loss (lamb) and teacher soft-label loss (1-lamb), in [0, 1]"),
    ("out_t_path", str, "outputs", "Path to load teacher outputs"),
]


def _parser(rows, description):
    p = argparse.ArgumentParser(description=description)
    for name, typ, default, hlp in rows:
        if typ == "store_true":
            p.add_argument(f"--{name}", action="store_true", help=hlp)
        else:
            p.add_argument(f"--{name}", type=typ, default=default, help=hlp)
    return p


def get_teacher_args(argv=None):
    return _parser(_COMMON, "GLNN teacher (HIP hot path)").parse_args(argv)


def get_student_args(argv=None):
    return _parser(_COMMON + _STUDENT_ONLY, "GLNN student distillation (HIP hot path)").parse_args(argv)


def _device(args):
    """The reference maps --device -1 (its default) to the CPU (train_teacher.py:162-165).  This build computes on
    libglnn_hip.so only, so a CPU selection fails HERE -- before any output directory is created -- instead of mid-run."""
    if args.device < 0 or not torch.cuda.is_available():
        raise SystemExit("glnn_amd runs on the MI355X HIP path only and has no CPU path: pass --device 0 (the reference's "
                         f"default --device -1 selects the CPU; torch.cuda.is_available() = {torch.cuda.is_available()})")
    return torch.device("cuda:" + str(args.device))


def _setting_dir(args, root, leaf):
    if args.exp_setting == "tran":
        return Path.cwd().joinpath(root, "transductive", args.dataset, leaf, f"seed_{args.seed}")
    if args.exp_setting == "ind":
        return Path.cwd().joinpath(root, "inductive", f"split_rate_{args.split_rate}", args.dataset, leaf, f"seed_{args.seed}")
    raise ValueError(f"Unknown experiment setting! {args.exp_setting}")


def _load(args, logger, model_name):
    g, labels, idx_train, idx_val, idx_test = load_data(args.dataset, args.data_path, split_idx=args.split_idx, seed=args.seed,
                                                        labelrate_train=args.labelrate_train, labelrate_val=args.labelrate_val)
    logger.info(f"Total {g.number_of_nodes()} nodes.")
    logger.info(f"Total {g.number_of_edges()} edges.")
    feats = g.ndata["feat"]
    args.feat_dim = feats.shape[1]
    args.label_dim = labels.int().max().item() + 1
    if 0 < args.feature_noise <= 1:
        feats = (1 - args.feature_noise) * feats + args.feature_noise * torch.randn_like(feats)
    conf = {}
    if args.model_config_path is not None:
        conf = get_training_config(args.model_config_path, model_name, args.dataset.replace("synthetic-", "").split("@")[0])
    conf = dict(args.__dict__, **conf)          # YAML wins over CLI flags, as in the reference
    return g, feats, labels, (idx_train, idx_val, idx_test), conf


def _save(args, output_dir, out, loss_and_score, model, g):
    np.savez(output_dir.joinpath("out"), out.detach().cpu().numpy())
    if args.save_results:
        np.savez(output_dir.joinpath("loss_and_score"), np.array(loss_and_score))
        torch.save(model.state_dict(), output_dir.joinpath("model.pth"))
    if args.exp_setting == "tran" and args.compute_min_cut:
        with open(output_dir.parent.joinpath("min_cut_loss"), "a+") as f:
            f.write(f"{compute_min_cut_loss(g.to(out.device), out) :.4f}\n")


def _prop(feats, g, k, device):
    return feature_prop(feats.to(device), g.to(device), k).cpu() if k > 0 else feats


def run_teacher(args):
    set_seed(args.seed)
    device = _device(args)
    if args.feature_noise != 0 and args.seed == 0:
        args.output_path = Path.cwd().joinpath(args.output_path, "noisy_features", f"noise_{args.feature_noise}")
    if args.feature_aug_k > 0 and args.seed == 0:
        args.output_path = Path.cwd().joinpath(args.output_path, "aug_features", f"aug_hop_{args.feature_aug_k}")
        args.teacher = f"GA{args.feature_aug_k}{args.teacher}"
    output_dir = _setting_dir(args, args.output_path, args.teacher)
    args.output_dir = output_dir
    check_writable(output_dir, overwrite=False)
    logger = get_logger(output_dir.joinpath("log"), args.console_log, args.log_level)
    logger.info(f"output_dir: {output_dir}")
    g, feats, labels, (idx_train, idx_val, idx_test), conf = _load(args, logger, args.teacher)
    conf["device"] = device
    logger.info(f"conf: {conf}")
    model = Model(conf)
    optimizer = optim.Adam(model.parameters(), lr=conf["learning_rate"], weight_decay=conf["weight_decay"])
    criterion = torch.nn.NLLLoss()
    evaluator = get_evaluator(conf["dataset"])
    loss_and_score = []
    if args.exp_setting == "tran":
        feats = _prop(feats, g, args.feature_aug_k, device)
        out, score_val, score_test = run_transductive(conf, model, g, feats, labels, (idx_train, idx_val, idx_test), criterion,
                                                      evaluator, optimizer, logger, loss_and_score)
        score_lst = [score_test]
    else:
        indices = graph_split(idx_train, idx_val, idx_test, args.split_rate, args.seed)
        if args.feature_aug_k > 0:
            idx_obs = indices[3]
            obs_feats = _prop(feats[idx_obs], g.subgraph(idx_obs), args.feature_aug_k, device)
            feats = _prop(feats, g, args.feature_aug_k, device)
            feats[idx_obs] = obs_feats
        out, score_val, score_tt, score_ti = run_inductive(conf, model, g, feats, labels, indices, criterion, evaluator, optimizer,
                                                           logger, loss_and_score)
        score_lst = [score_tt, score_ti]
    logger.info(f"num_layers: {conf['num_layers']}. hidden_dim: {conf['hidden_dim']}. dropout_ratio: {conf['dropout_ratio']}")
    logger.info(f"# params {sum(p.numel() for p in model.parameters())}")
    _save(args, output_dir, out, loss_and_score, model, g)
    return score_lst


def run_student(args):
    set_seed(args.seed)
    device = _device(args)
    if args.feature_noise != 0 and args.seed == 0:
        args.output_path = Path.cwd().joinpath(args.output_path, "noisy_features", f"noise_{args.feature_noise}")
        args.out_t_path = args.output_path      # the teacher is assumed trained on the same noisy features
    if args.feature_aug_k > 0 and args.seed == 0:
        args.output_path = Path.cwd().joinpath(args.output_path, "aug_features", f"aug_hop_{args.feature_aug_k}")
        args.student = f"GA{args.feature_aug_k}{args.student}"
    output_dir = _setting_dir(args, args.output_path, f"{args.teacher}_{args.student}")
    out_t_dir = _setting_dir(args, args.out_t_path, args.teacher)
    args.output_dir = output_dir
    check_writable(output_dir, overwrite=False)
    check_readable(out_t_dir)
    logger = get_logger(output_dir.joinpath("log"), args.console_log, args.log_level)
    logger.info(f"output_dir: {output_dir}")
    logger.info(f"out_t_dir: {out_t_dir}")
    g, feats, labels, (idx_train, idx_val, idx_test), conf = _load(args, logger, args.student)     # student section of the YAML
    conf["device"] = device
    logger.info(f"conf: {conf}")
    model = Model(conf)
    optimizer = optim.Adam(model.parameters(), lr=conf["learning_rate"], weight_decay=conf["weight_decay"])
    criterion_l = torch.nn.NLLLoss()
    criterion_t = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    evaluator = get_evaluator(conf["dataset"])
    out_t = load_out_t(out_t_dir)
    for nm, idx in (("train", idx_train), ("val", idx_val), ("test", idx_test)):
        logger.debug(f"teacher score on {nm} data: {evaluator(out_t[idx], labels[idx])}")
    loss_and_score = []
    if args.exp_setting == "tran":
        distill_indices = (idx_train, torch.cat([idx_train, idx_val, idx_test]), idx_val, idx_test)
        feats = _prop(feats, g, args.feature_aug_k, device)
        out, score_val, score_test = distill_run_transductive(conf, model, feats, labels, out_t, distill_indices, criterion_l,
                                                              criterion_t, evaluator, optimizer, logger, loss_and_score)
        score_lst = [score_test]
    else:
        obs_idx_train, obs_idx_val, obs_idx_test, idx_obs, idx_test_ind = graph_split(idx_train, idx_val, idx_test, args.split_rate, args.seed)
        distill_indices = (obs_idx_train, torch.cat([obs_idx_train, obs_idx_val, obs_idx_test]), obs_idx_val, obs_idx_test,
                           idx_obs, idx_test_ind)
        if args.feature_aug_k > 0:
            obs_feats = _prop(feats[idx_obs], g.subgraph(idx_obs), args.feature_aug_k, device)
            feats = _prop(feats, g, args.feature_aug_k, device)
            feats[idx_obs] = obs_feats
        out, score_val, score_tt, score_ti = distill_run_inductive(conf, model, feats, labels, out_t, distill_indices, criterion_l,
                                                                   criterion_t, evaluator, optimizer, logger, loss_and_score)
        score_lst = [score_tt, score_ti]
    logger.info(f"num_layers: {conf['num_layers']}. hidden_dim: {conf['hidden_dim']}. dropout_ratio: {conf['dropout_ratio']}")
    logger.info(f"# params {sum(p.numel() for p in model.parameters())}")
    _save(args, output_dir, out, loss_and_score, model, g)
    return score_lst


def _main(args, run):
    if args.num_exp == 1:
        score = run(args)
        score_str = "".join([f"{s : .4f}\t" for s in score])
    else:
        scores = []
        for seed in range(args.num_exp):
            args.seed = seed
            scores.append(run(args))
        scores = np.array(scores)
        score_str = "".join([f"{s : .4f}\t" for s in scores.mean(axis=0)] + [f"{s : .4f}\t" for s in scores.std(axis=0)])
    with open(args.output_dir.parent.joinpath("exp_results"), "a+") as f:
        f.write(f"{score_str}\n")
    print(score_str)      # for collecting aggregated results


def teacher_main(argv=None):
    _main(get_teacher_args(argv), run_teacher)


def student_main(argv=None):
    _main(get_student_args(argv), run_student)
