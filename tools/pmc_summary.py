#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs (one or more passes).

    python tools/pmc_summary.py gpurun_out/pmc_sq/*/*_counter_collection.csv [more.csv ...]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    for junk in ("void ", "amp::"):
        name = name.replace(junk, "")
    return name.split("(")[0][:48]


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        per_dispatch = defaultdict(dict)
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                key = (row["Dispatch_Id"], short(row["Kernel_Name"]), int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
                per_dispatch[key][row["Counter_Name"]] = per_dispatch[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for (_, name, grid), counters in per_dispatch.items():
            for c, v in counters.items():
                acc[(name, grid)][c].append(v)
    counters = sorted({c for d in acc.values() for c in d})
    print("| kernel | workgroups | launches | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    for (name, grid), d in sorted(acc.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
        if name.startswith("at::") or name.startswith("__amd"):
            continue
        n = max(len(v) for v in d.values())
        print(f"| {name} | {grid} | {n} | " + " | ".join(f"{sum(d[c]) / len(d[c]):.4g}" if c in d else "" for c in counters) + " |")


def traffic_json(paths, out):
    """{kernel@workgroups: {fetch_bytes, write_bytes, hbm_bytes}} per launch.  FETCH_SIZE / WRITE_SIZE are in KiB;
    on gfx950 FETCH_SIZE reports exactly half of a wide (16 B / lane) coalesced read stream (MI355X_MICROARCH.md,
    HBM section): it is doubled here.  WRITE_SIZE is exact for 16-B-per-lane stores."""
    import json

    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        per_dispatch = defaultdict(dict)
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                key = (row["Dispatch_Id"], short(row["Kernel_Name"]), int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
                per_dispatch[key][row["Counter_Name"]] = per_dispatch[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for (_, name, grid), counters in per_dispatch.items():
            for c, v in counters.items():
                acc[f"{name}@{grid}"][c].append(v)
    res = {}
    for k, d in acc.items():
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 * 2 if d.get("FETCH_SIZE") else None
        w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024 if d.get("WRITE_SIZE") else None
        res[k] = {"fetch_bytes_x2_corrected": f, "write_bytes": w, "hbm_bytes": (f or 0) + (w or 0) if f is not None and w is not None else None}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "--traffic-json":
        traffic_json(sys.argv[3:], sys.argv[2])
    else:
        main(sys.argv[1:])
