"""One distillation epoch through the preserved driver (distill_run_transductive: hard pass + soft pass + three evaluations,
reference train_and_eval.py:520-606) on ogbn-arxiv-shaped synthetic inputs; seconds per epoch and where the host time goes."""
import cProfile, logging, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glnn_amd import ops, train_and_eval as te
from glnn_amd.models import Model
dev = "cuda:0"
n, n_l, n_va, n_te = 169343, 90941, 29799, 48603
which = sys.argv[1] if len(sys.argv) > 1 else "MLP"
dims = {"MLP": [128, 256, 256, 40], "MLP3w4": [128, 1024, 1024, 40]}[which]
torch.manual_seed(0)
perm = torch.randperm(n)
idx_l, idx_va, idx_te = perm[:n_l], perm[n_l:n_l + n_va], perm[n_l + n_va:]
feats = torch.randn(n, dims[0]); labels = torch.randint(0, dims[-1], (n,)); out_t = torch.log_softmax(torch.randn(n, dims[-1]), 1)
model = Model(dict(model_name="MLP", num_layers=3, feat_dim=dims[0], hidden_dim=dims[1], label_dim=dims[-1], dropout_ratio=0.2 if which == "MLP" else 0.5,
                   norm_type="batch", device=dev))
opt = torch.optim.Adam(model.parameters(), lr=0.01)
epochs = int(os.environ.get("EPOCHS", "20"))
conf = dict(seed=0, device=dev, batch_size=512, lamb=0.0 if os.environ.get("LAMB") is None else float(os.environ["LAMB"]), max_epoch=epochs, eval_interval=1, patience=1000)
evaluator = lambda out, y: float((out.argmax(1) == y).float().mean().item())
logger = logging.getLogger("bench"); logger.setLevel(logging.ERROR)
args = (conf, model, feats, labels, out_t, (idx_l, torch.arange(n), idx_va, idx_te), torch.nn.NLLLoss(), torch.nn.KLDivLoss(reduction="batchmean", log_target=True),
        evaluator, opt, logger, [])
conf["max_epoch"] = 2
te.distill_run_transductive(*args)                      # warm-up (engine construction, first launches)
conf["max_epoch"] = epochs
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
te.distill_run_transductive(*args)
pr.disable()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
steps = epochs * (n_l // 512 + n // 512)
print(f"arxiv {which}: {epochs} epochs in {dt:.3f} s = {dt / epochs * 1e3:.1f} ms per epoch ({steps / epochs} optimiser steps + 3 evaluations each)", flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
