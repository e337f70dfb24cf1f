#!/bin/bash
# usage (GPU box, repo root): scripts/request_size.sh   -> gpurun_out/request_size/{times.txt,summary.txt}
# which load forms make the L2 ask the fabric for less than a 128-byte line?  experiments/request_size.hip under
# the per-size request counters (one --pmc pass, --kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/request_size
mkdir -p "$OUT"
BIN=$PWD/experiments/request_size
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 experiments/request_size.hip -o "$BIN"
"$BIN" > "$OUT/times.txt" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/rd" -- "$BIN" > "$OUT/rd.log" 2>&1
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import collections, csv, glob, re, sys
root = sys.argv[1]
files = sorted(glob.glob(f"{root}/rd/*/*counter_collection.csv"))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[-1])):
    m = re.search(r"gather_kernel<(\d+), (\d+), (\d+)>", r["Kernel_Name"])
    if m:
        vals[tuple(map(int, m.groups()))][r["Counter_Name"]].append(float(r["Counter_Value"]))
pol = ["plain", "nt", "sc0", "sc1", "sc0 sc1", "nt sc0 sc1", "2 x dwordx2", "4 x dword"]
print("per launch (mean of 3): fabric read requests by size; row bytes = 64 M rows x D x 4")
for (lpr, body, tail), cs in vals.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = 47 if lpr == 16 else 100
    n32, n64, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0)
    byt = 32 * n32 + 64 * n64 + 128 * n128
    print(f"D={d:3d} body={pol[body]:11s} tail={pol[tail]:11s} RDREQ {m.get('TCC_EA0_RDREQ_sum', 0):.4g}  32B {n32:.4g}  64B {n64:.4g}  128B {n128:.4g}  "
          f"bytes {byt / 1e9:.2f} GB = {byt / (64e6 * d * 4):.3f} x row bytes")
PY
cat "$OUT/times.txt"
tail -n 3 "$OUT/rd.log"
rm -rf "$OUT"/rd/*/*kernel_trace.csv; find "$OUT" -name "*.db" -delete 2>/dev/null
