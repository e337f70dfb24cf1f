"""The C restatement under AddressSanitizer + UBSan (SURVEY.md section 5 asks for a sanitizer build of the CPU restatement): oracle/selftest.c
drives every exported function of oracle/glnn_oracle.c on hand-made graphs with written-down answers (path / star pieces, an isolated node, a
duplicate edge, a self-loop, a block with fewer destinations than sources, both Linear weight layouts, the norm-free BatchNorm path, empty
inputs) out of heap buffers of exactly the needed size."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_known_answers_under_address_and_undefined_behaviour_sanitizers():
    if not shutil.which("gcc") and not shutil.which("cc"):
        pytest.skip("no C compiler")
    build = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "selftest_asan"], capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip("this toolchain has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([os.path.join(ROOT, "oracle", "selftest_asan")], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "all known answers met" in run.stdout
