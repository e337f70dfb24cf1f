#!/bin/bash
# usage (GPU box, repo root): scripts/rowpanel_pmc.sh -> gpurun_out/rowpanel_pmc/{times.txt,sq.csv}: where do the wave cycles of K3r go?
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/rowpanel_pmc
mkdir -p "$OUT"
python scripts/rowpanel_probe.py > "$OUT/times.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq" -- python scripts/rowpanel_probe.py > "$OUT/sq.log" 2>&1
python - "$OUT" <<'PY' | tee "$OUT/sq_summary.txt"
import collections, csv, glob, sys
files = sorted(glob.glob(sys.argv[1] + "/sq/*/*counter_collection.csv"))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[-1])):
    n = r["Kernel_Name"]
    if "gemm" in n:
        key = n.replace("(anonymous namespace)::", "")[:70] + " grid=" + r.get("Grid_Size", "?")
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in vals.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k, "launches", max(len(v) for v in cs.values()))
    print("   ", {c: f"{v:.4g} ({v / wc:.3f} of wave cycles)" for c, v in m.items()})
PY
cat "$OUT/times.txt"; tail -n 2 "$OUT/sq.log"
rm -rf "$OUT"/sq/*/*kernel_trace.csv; find "$OUT" -name "*.db" -delete 2>/dev/null
