"""Golden vectors for the reference's DRIVER loops around the student path:
epoch order (hard pass, soft pass), evaluation every eval_interval, best-validation snapshot, patience-based early stop,
final evaluation of the restored state -- produced by the reference's own distill_run_transductive /
distill_run_inductive / run_transductive (MLP branch) (reference train_and_eval.py:144-287, 520-742), with
utils.graph_split providing the inductive indices.  All of it is pure PyTorch on the student side, so it is imported
with the same import-only dgl/ogb stubs as make_student_golden.py.

Configs are small and WITHOUT the Adam gauge freedom (no norm or weight_decay > 0), so that tens of optimiser steps stay
comparable at 1e-4.  The reference draws its permutations with torch.randperm after set_seed(conf["seed"]) -- the HIP
mirror does exactly the same on the CPU, so no permutation needs to be replayed.

    python tests/golden/make_driver_golden.py        (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_student_golden as msg          # noqa: E402  (stubs + input generators)


class ListLogger:
    def __init__(self):
        self.lines = []

    def debug(self, m):
        self.lines.append(m)

    info = debug


CASES = {
    "tran_nonorm": dict(dims=[16, 32, 5], norm="none", wd=5e-3, lamb=0.3, B=64, n=420, max_epoch=14, patience=3, seed=11),
    "tran_bn_wd": dict(dims=[12, 24, 24, 4], norm="batch", wd=5e-4, lamb=0.0, B=50, n=360, max_epoch=10, patience=2, seed=12),
    "ind_nonorm": dict(dims=[16, 32, 5], norm="none", wd=5e-3, lamb=0.5, B=64, n=420, max_epoch=12, patience=3, seed=13, ind=True),
    "plain_mlp_tran": dict(dims=[16, 32, 5], norm="none", wd=5e-3, lamb=1.0, B=64, n=420, max_epoch=10, patience=3, seed=14, plain=True),
}


def main():
    msg._stub_modules()
    sys.path.insert(0, msg.REF)
    import models as ref_models            # noqa
    import train_and_eval as ref_te        # noqa
    import utils as ref_utils              # noqa
    torch.set_num_threads(1)
    out = {}
    for name, c in CASES.items():
        dims, n = c["dims"], c["n"]
        feats, labels, out_t, _ = msg.make_inputs(c["seed"], n, dims[0], dims[-1], 10)
        # labels that depend on the features, so that validation scores move and early stopping is exercised
        w = np.random.RandomState(c["seed"]).standard_normal((dims[0], dims[-1])).astype(np.float32)
        labels = (feats @ w).argmax(1).astype(np.int64)
        sd0 = msg.make_state(c["seed"], dims, c["norm"])
        perm = np.random.RandomState(c["seed"] + 1).permutation(n)
        idx_train, idx_val, idx_test = (torch.from_numpy(perm[a:b].astype(np.int64)) for a, b in ((0, 120), (120, 200), (200, n)))
        conf = dict(model_name="MLP", num_layers=len(dims) - 1, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.0,
                    norm_type=c["norm"], device="cpu", seed=c["seed"], batch_size=c["B"], lamb=c["lamb"], max_epoch=c["max_epoch"],
                    patience=c["patience"], eval_interval=1)
        model = ref_models.Model(conf)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd0.items()})
        optimizer = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=c["wd"])
        crit_l = torch.nn.NLLLoss()
        crit_t = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
        evaluator = ref_utils.get_evaluator("cora")
        logger, las = ListLogger(), []
        tf, tl, tt = torch.from_numpy(feats), torch.from_numpy(labels), torch.from_numpy(out_t)
        if c.get("plain"):
            res = ref_te.run_transductive(conf, model, None, tf, tl, (idx_train, idx_val, idx_test), crit_l, evaluator, optimizer, logger, las)
        elif c.get("ind"):
            obs_tr, obs_va, obs_te, idx_obs, idx_ti = ref_utils.graph_split(idx_train, idx_val, idx_test, 0.25, c["seed"])
            obs_idx_t = torch.cat([obs_tr, obs_va, obs_te])                       # train_student.py:318
            res = ref_te.distill_run_inductive(conf, model, tf, tl, tt, (obs_tr, obs_idx_t, obs_va, obs_te, idx_obs, idx_ti), crit_l, crit_t,
                                               evaluator, optimizer, logger, las)
        else:
            idx_t = torch.cat([idx_train, idx_val, idx_test])                      # train_student.py:297-299
            res = ref_te.distill_run_transductive(conf, model, tf, tl, tt, (idx_train, idx_t, idx_val, idx_test), crit_l, crit_t, evaluator,
                                                  optimizer, logger, las)
        out[f"{name}.out"] = res[0].numpy()
        out[f"{name}.scores"] = np.asarray(res[1:], np.float64)
        out[f"{name}.loss_and_score"] = np.asarray(las, np.float64)
        out[f"{name}.last_log"] = np.asarray(logger.lines[-1])
        for k in ("wd", "lamb", "B", "n", "max_epoch", "patience", "seed"):
            out[f"{name}.cfg.{k}"] = np.float64(c[k])
        out[f"{name}.cfg.dims"], out[f"{name}.cfg.norm"] = np.asarray(dims), np.asarray(c["norm"])
        out[f"{name}.cfg.kind"] = np.asarray("plain" if c.get("plain") else "ind" if c.get("ind") else "tran")
        for k, v in model.state_dict().items():
            out[f"{name}.final.{k}"] = v.numpy().copy()
        print(name, "epochs run", len(las), "|", logger.lines[-1])
    path = os.path.join(HERE, "driver_loops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
