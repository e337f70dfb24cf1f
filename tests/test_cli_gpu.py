"""End-to-end: the preserved command lines train a GCN teacher and distil an MLP student on a cora-shaped
synthetic graph through the HIP path, and hand over out.npz exactly like the reference (config 0)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("teacher,setting", [("GCN", "tran"), ("SAGE", "ind")])
def test_teacher_then_student_cli(tmp_path, teacher, setting):
    common = ["--dataset", "synthetic-cora", "--teacher", teacher, "--device", "0", "--max_epoch", "6", "--patience", "3",
              "--exp_setting", setting, "--save_results"]
    out = _run("train_teacher.py", common + ["--compute_min_cut"], tmp_path)
    base = tmp_path / "outputs" / ("transductive" if setting == "tran" else "inductive/split_rate_0.2") / "synthetic-cora"
    tdir = base / teacher / "seed_0"
    out_t = np.load(tdir / "out.npz")["arr_0"]
    assert out_t.shape == (2485, 7) and out_t.dtype == np.float32
    np.testing.assert_allclose(np.exp(out_t).sum(1), 1.0, atol=1e-4)          # log-probabilities of ALL nodes
    assert (tdir / "model.pth").exists() and (tdir / "loss_and_score.npz").exists() and (tdir / "log").exists()
    assert len(out.split()) == (1 if setting == "tran" else 2)
    _run("train_student.py", common + ["--student", "MLP", "--lamb", "0.5"], tmp_path)
    sdir = base / f"{teacher}_MLP" / "seed_0"
    out_s = np.load(sdir / "out.npz")["arr_0"]
    assert out_s.shape == (2485, 7) and np.isfinite(out_s).all()
    ls = np.load(sdir / "loss_and_score.npz")["arr_0"]
    assert ls.shape[1] == (7 if setting == "tran" else 9) and ls.shape[0] >= 1
