"""Steady-state slice of a rocprofv3 kernel trace with the queue id: start, end, duration, queue, kernel."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
skip, count = int(sys.argv[2]), int(sys.argv[3])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "anonymous namespace" in r["Kernel_Name"] or "act_fwd" in r["Kernel_Name"]]
seg = ours[skip:skip + count]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
