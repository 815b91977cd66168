"""Last step of a rocprofv3 kernel trace as a timeline: python tools/timeline/trace_timeline.py <kernel_trace.csv> <first kernel substring>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void amp::", "").replace("amp::", "")[:60]
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}  grid {r.get('Grid_Size', '')}")
