"""Print a steady-state slice of a rocprofv3 kernel trace: start, duration, gap to the previous kernel."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
skip, count = int(sys.argv[2]), int(sys.argv[3])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "anonymous namespace" in r["Kernel_Name"]]
seg = ours[skip:skip + count]
t0 = int(seg[0]["Start_Timestamp"]); prev = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:8.1f}us dur {(e - s) / 1e3:6.1f} gap {gap:6.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} {name}")
    prev = e
