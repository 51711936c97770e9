"""Per-kernel durations and idle gaps from a rocprofv3 --kernel-trace csv: python tools/trace_gaps.py <dir> [skip_first_n_per_kernel]"""
import csv, glob, sys, collections
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t_min = min(int(r["Start_Timestamp"]) for r in rows)
print(f"{'kernel':58s} {'n':>4s} {'dur us':>8s} {'gap us':>8s} {'period us':>9s}")
for name, iv in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    iv.sort()
    iv = iv[skip:]
    if len(iv) < 3:
        continue
    dur = sum(e - s for s, e in iv) / len(iv) / 1e3
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
    per = [iv[i + 1][0] - iv[i][0] for i in range(len(iv) - 1)]
    gaps.sort(); per.sort()
    print(f"{name[:58]:58s} {len(iv):4d} {dur:8.1f} {gaps[len(gaps)//2]/1e3:8.1f} {per[len(per)//2]/1e3:9.1f}")
# the last launch of every kernel: when it started / ended relative to the last eq launch
last = {name: max(iv) for name, iv in by.items()}
ref = min(s for s, e in last.values())
print("last launch of each kernel (us after the earliest of them):")
for name, (s, e) in sorted(last.items(), key=lambda kv: kv[1][0]):
    print(f"  {name[:58]:58s} start {(s - ref)/1e3:9.1f} end {(e - ref)/1e3:9.1f}")
# full timeline between the 30th and the 32nd launch of the EQ kernel
eq = sorted(iv for name, ivs in by.items() if "eq_systolic" in name for iv in ivs)
if len(eq) > 33:
    lo, hi = eq[30][0], eq[32][0]
    ev = []
    for name, ivs in by.items():
        for s, e in ivs:
            if lo - 200000 <= s < hi:
                ev.append((s, e, name))
    print("timeline (us after EQ launch 30):")
    for s, e, name in sorted(ev):
        short = name.replace("af::(anonymous namespace)::", "").replace("void ", "")[:40]
        print(f"  {(s - lo)/1e3:9.1f} .. {(e - lo)/1e3:9.1f}  {short}")
