#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_train_sage.sh TAG -> gpurun_out/pmc_train_TAG/: kernel stats + HBM traffic (FETCH_SIZE x2 + WRITE_SIZE,
# separate --pmc passes, --kernel-trace only) of the sampled-block teacher training on the products config (scripts/bench_train_sage.py).
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
export GLNN_BENCH_EPOCHS=2
OUT=$PWD/gpurun_out/pmc_train_$TAG
mkdir -p "$OUT"
CMD="python scripts/bench_train_sage.py ogbn-products"
$CMD > "$OUT/train_plain.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
f=$(ls "$OUT"/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_train_sage_products_kernel_stats.csv"
python - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, sys
root, tag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])     # kernel -> [launches, fetch bytes, write bytes, ns]
for sub, col in (("fetch", 1), ("write", 2)):
    files = sorted(glob.glob(f"{root}/{sub}/*/*counter_collection.csv"))
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
        v = float(r["Counter_Value"])
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950: FETCH_SIZE tallies 128-byte requests as 64 (MI355X_MICROARCH.md) -> x2
        tot[k][col] += v * 1024 * (2 if sub == "fetch" else 1)
        if sub == "fetch":
            tot[k][0] += 1
            tot[k][3] += float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0))
rows = sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
with open(f"{root}/{tag}_pmc_train_sage.csv", "w") as f:
    f.write("kernel,launches,fetch_GB_total,write_GB_total,GB_per_launch,profiled_ms_total,GBps_under_profiler\n")
    for k, (n, fe, wr, ns) in rows[:40]:
        f.write(f"\"{k}\",{n},{fe / 1e9:.3f},{wr / 1e9:.3f},{(fe + wr) / 1e9 / max(n, 1):.4f},{ns / 1e6:.2f},{(fe + wr) / max(ns, 1):.1f}\n")
print(open(f"{root}/{tag}_pmc_train_sage.csv").read()[:3500])
PY
tail -n 4 "$OUT/train_plain.log"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
