"""The LAST `count` kernels of a rocprofv3 kernel trace, every kernel, with gaps (the tail of a script = its steady-state loop)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
count = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seg = rows[-count:]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    print(f"{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:5.1f} dur {(e - s) / 1e3:6.1f}  {name}")
    prev = e
