"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/prof_top.py <dir> [n]"""
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print(f"{r['Name'][:72]:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.1f} ms")
