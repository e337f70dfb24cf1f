#!/bin/bash
# usage (GPU box, repo root): scripts/train_sage_timeline.sh TAG -> gpurun_out/train_sage_timeline_TAG.txt: ms/step + one steady-state engine step of the
# products teacher-training configuration, launch by launch
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/train_sage_timeline_$TAG.txt
python scripts/trace_train_sage_step.py 2>&1 | grep "ms;" > "$OUT"
rm -rf /tmp/tr_sage; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_sage -- python scripts/trace_train_sage_step.py 8 > /dev/null 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob("/tmp/tr_sage/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last-but-one Adam launch to the last one
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[adam[-2] + 1:adam[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0
with open(sys.argv[1], "a") as out:
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        out.write(f"{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:5.1f} dur {(e - s) / 1e3:6.1f}  {name}\n")
        prev = e
    out.write(f"step span {(prev - t0) / 1e3:.1f} us, {len(seg)} launches\n")
PY
cat "$OUT"
