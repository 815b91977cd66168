"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/kernel_stats_top.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':72s} {'calls':>6s} {'total us':>10s} {'avg us':>8s} {'%':>5s}")
for r in rows[:n]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e3:10.1f} {float(r['AverageNs'])/1e3:8.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}")
