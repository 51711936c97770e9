"""Where the chain stream idles inside one bench step: python tools/step_timeline.py <rocprofv3 --kernel-trace dir> [windows per step]
For the LAST step of the trace: every chain launch with the idle gap in front of it and when the same window's EQ launch (its
producer) ended; the sum of the gaps is what the step lasts beyond the chain's own launches."""
import csv, glob, sys, collections
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
W = int(sys.argv[2]) if len(sys.argv) > 2 else 54
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def last(sub, n=W):
    ivs = sorted(iv for name, v in by.items() if sub in name for iv in v)
    return ivs[-n:]
chain, eq, pre, syn = last("chain_ring"), last("eq_systolic"), last("supp_prefilter"), last("supp_synth")
t0 = min(pre[0][0], chain[0][0])
print(f"step: first pre-pass launch at 0, last chain launch ends at {(chain[-1][1] - t0) / 1e6:.2f} ms; chain launches sum {sum(e - s for s, e in chain) / 1e6:.2f} ms")
print(f"{'w':>3s} {'pre end':>8s} {'synth end':>9s} {'eq end':>8s} {'chain start':>11s} {'chain end':>9s} {'idle before':>11s} {'wait on eq':>10s}")
idle = 0.0
for w in range(len(chain)):
    s, e = chain[w]
    gap = (s - (chain[w - 1][1] if w else t0)) / 1e6
    idle += gap
    eq_end = eq[w][1] if w < len(eq) else 0
    print(f"{w:3d} {(pre[w][1] - t0) / 1e6:8.2f} {(syn[w][1] - t0) / 1e6:9.2f} {(eq_end - t0) / 1e6:8.2f} {(s - t0) / 1e6:11.2f} {(e - t0) / 1e6:9.2f} {gap:11.3f} {(s - eq_end) / 1e6:10.3f}")
print(f"idle on the chain stream: {idle:.2f} ms")
if len(sys.argv) > 3:  # every launch that starts in the first <ms> of the step
    horizon = float(sys.argv[3]) * 1e6
    ev = []
    for name, ivs in by.items():
        for s, e in ivs:
            if t0 <= s < t0 + horizon:
                ev.append((s, e, name))
    print(f"launches starting in the first {sys.argv[3]} ms:")
    for s, e, name in sorted(ev):
        short = name.replace("af::(anonymous namespace)::", "").replace("void ", "").replace("af::", "")[:34]
        print(f"  {(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e6:6.3f})  {short}")
