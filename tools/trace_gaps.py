#!/usr/bin/env python3
"""Per-kernel durations AND the gap in front of each kernel (start - previous end) from a rocprofv3 --kernel-trace CSV,
over the last `tail` launches (the steady-state replays)."""
import csv, sys
from collections import defaultdict

path, tail = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 800
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path, newline=""))))
rows = rows[-tail:]
dur, gap = defaultdict(list), defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    n = n.replace("void ", "").replace("amp::", "").split("(")[0]
    dur[n].append(e - s)
    if i:
        gap[n].append(s - rows[i - 1][1])
print("| kernel | calls | avg us | min us | gap before, avg us |")
print("|---|---|---|---|---|")
tot = 0.0
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = gap.get(n, [0])
    tot += (sum(v) / len(v) + sum(g) / len(g)) / 1e3
    print(f"| {n} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {sum(g) / len(g) / 1e3:.2f} |")
print(f"\nsum of (avg duration + avg gap) over the kernels: {tot:.1f} us")
