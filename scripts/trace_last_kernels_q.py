"""The LAST `count` kernels of a rocprofv3 kernel trace with queue ids: start, end, duration, queue, kernel."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
count = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seg = rows[-count:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50]
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f} q{r.get('Queue_Id', '?'):>3}  {name}")
