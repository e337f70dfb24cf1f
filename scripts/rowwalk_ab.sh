#!/bin/bash
# usage (GPU box, repo root): scripts/rowwalk_ab.sh -> gpurun_out/rowwalk/{times.txt,clock.csv,...}: the K3w variants against K3r and the tiled kernels
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/rowwalk
mkdir -p "$OUT"
python scripts/rowwalk_probe.py default 1,2 2>&1 | grep -v amdgpu.ids | tee "$OUT/times.txt"
for v in variants/libglnn_rw_*.so; do
  [ -e "$v" ] || continue
  GLNN_LIB_PATH=$PWD/$v python scripts/rowwalk_probe.py "$(basename $v .so)" 1 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/times.txt"
done
# effective clock and SQ activity of the kernels of the default build (separate counter passes)
(cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/clk" -- python $OLDPWD/scripts/rowwalk_probe.py default 1,2 > "$OUT/clk.log" 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python $OLDPWD/scripts/rowwalk_probe.py default 1,2 > "$OUT/sq.log" 2>&1)
python - "$OUT" <<'PY' | tee "$OUT/pmc_summary.txt"
import collections, csv, glob, sys
out = sys.argv[1]
for sub in ("clk", "sq"):
    files = sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True))
    if not files:
        print(sub, "no counter file"); continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[-1])):
        n = r["Kernel_Name"]
        if "gemm" in n:
            key = n.replace("(anonymous namespace)::", "")[:60] + " grid=" + r.get("Grid_Size", "?")
            vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
                vals[key]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, cs in vals.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        extra = ""
        if "ns" in m and "GRBM_GUI_ACTIVE" in m:
            extra = f"  -> {m['GRBM_GUI_ACTIVE'] / m['ns']:.3f} GHz over {m['ns'] / 1e3:.1f} us"
        wc = m.get("SQ_WAVE_CYCLES")
        print(sub, k, "launches", max(len(v) for v in cs.values()), extra)
        print("     ", {c: (f"{v:.4g}" + (f" ({v / wc:.3f})" if wc else "")) for c, v in m.items()})
PY
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null; find "$OUT" -name "*agent_info.csv" -delete 2>/dev/null
ls -la "$OUT"
