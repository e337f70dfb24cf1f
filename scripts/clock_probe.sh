#!/bin/bash
# Samples sclk / power while (i) the register-only MFMA loop and (ii) the MLP3w8 NT GEMM run back to back: tells a
# power-limited clock from an issue-limited kernel.  gpurun -- 'scripts/clock_probe.sh'
cd "$(dirname "$0")/.."
sample() {  # $1 = label, runs until the file gpurun_out/.probe_stop appears
  while [ ! -e gpurun_out/.probe_stop ]; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | sed "s/^/[$1] /"
    sleep 0.4
  done
}
mkdir -p gpurun_out; rm -f gpurun_out/.probe_stop
echo "== idle"; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' '
echo "== mfma register loop"
sample mfma & SP=$!
for i in 1 2 3 4 5 6; do experiments/mfma_peak | grep "rep 3" | head -2; done
touch gpurun_out/.probe_stop; wait $SP; rm -f gpurun_out/.probe_stop
echo "== NT gemm loop"
sample gemm & SP=$!
python - <<'P'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from glnn_amd import ops
dev = "cuda:0"; m, k, n = 4096, 2048, 2048
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
out = ops.feat_empty(m, n, dev); ws = torch.empty(1 << 24, device=dev)
for fn, name in ((lambda: ops.gemm(x, w, out=out, workspace=ws), "ours NT"), (lambda: torch.matmul(x, w.t()), "torch NT")):
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(200): fn()
        torch.cuda.synchronize(); it += 200
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / it * 1e6:.1f} us/launch sustained over {dt:.1f} s = {2.0 * m * k * n * it / dt / 1e12:.1f} TF", flush=True)
P
touch gpurun_out/.probe_stop; wait $SP; rm -f gpurun_out/.probe_stop
