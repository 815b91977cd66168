"""Kernel totals of a rocprofv3 kernel trace: python tools/timeline/trace_summary.py <kernel_trace.csv> [top]"""
import csv, sys
from collections import defaultdict
tot, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
    tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[n] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for n, t in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{t:10.1f} us  {cnt[n]:6d} x {t / cnt[n]:8.1f}  {n}")
