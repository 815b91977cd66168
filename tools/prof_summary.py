#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, avg / min / max duration.

    python tools/prof_summary.py gpurun_out/prof_r1/*/*_kernel_trace.csv > profiles/r01_bench_kernel_trace.md

bench.py runs the hot path at two shard sizes (65 536 and 4 096 envs), so rocprofv3's own --stats table mixes two
launch shapes per kernel; grouping by grid size separates them.  The average is over EVERY launch of the run (warm-up, the
launches next to the fp32-engine / calibration sections included); the median is what the timed region's launches take.
"""
import csv
import sys
from collections import defaultdict


def main(path):
    groups = defaultdict(list)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            for junk in ("void ", "amp::"):
                name = name.replace(junk, "")
            name = name.split("(")[0]
            grid = int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1)
            groups[(name, grid, int(row["VGPR_Count"]), int(row["LDS_Block_Size"]))].append(
                int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    total = sum(sum(v) for v in groups.values())
    print(f"source: {path}\n")
    print("| kernel | workgroups | VGPR | LDS B | calls | avg us | median us | min us | max us | % of GPU time |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for (name, grid, vgpr, lds), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if name.startswith("at::") or name.startswith("__amd"):
            name = name[:60]
        med = sorted(v)[len(v) // 2]
        print(f"| {name} | {grid} | {vgpr} | {lds} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {med / 1e3:.2f} | {min(v) / 1e3:.2f} | "
              f"{max(v) / 1e3:.2f} | {100.0 * sum(v) / total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
