"""Which window the one-launch chain waits for: python tools/supply_line.py <rocprofv3 --kernel-trace dir> [frames per step] [chain ms per frame]
For the last step of the trace: when every window's ready count was published (chain_publish_ready_kernel), the frames it
covers (the schedule bench.py's default call produces), and the finish time the chain could reach if that window were the
only constraint: publish time + (frames from the window's first one to the end) x chain rate.  The largest is the binding one."""
import csv, glob, sys, collections
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
chain = sorted(iv for n, v in by.items() if "chain_ring" in n for iv in v)[-1]
pub = sorted(iv for n, v in by.items() if "publish_ready" in n for iv in v)
pub = [p for p in pub if p[0] >= chain[0] - 1_000_000]
pre = sorted(iv for n, v in by.items() if "supp_prefilter" in n for iv in v)
pre = [p for p in pre if p[0] >= chain[0] - 2_000_000]
t0 = min(chain[0], pre[0][0])
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
W = len(pub)
up = [4, 8, 16]  # the default schedule: 4, 8, 16, 20 ..., 16, 8, 4
body = total - 2 * sum(up)
sched = up + [20] * (body // 20) + ([body % 20] if body % 20 else []) + up[::-1]
if len(sched) != W:
    print(f"schedule guess has {len(sched)} windows, trace {W}: the frames column is a guess")
    sched = (sched + [0] * W)[:W]
print(f"chain launch {(chain[0] - t0) / 1e6:.2f} .. {(chain[1] - t0) / 1e6:.2f} ms ({(chain[1] - chain[0]) / 1e6:.2f} ms)")
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1827  # ms of chain work per frame
f0 = 0
worst = (0, 0)
for w, (s, e) in enumerate(pub):
    bound = (e - t0) / 1e6 + (total - f0) * rate
    if bound > worst[0]:
        worst = (bound, w)
    print(f"{w:3d} frames {f0:4d}+{sched[w]:2d} published {(e - t0) / 1e6:8.2f} ms  -> chain could end at {bound:8.2f}")
    f0 += sched[w]
print(f"binding window: {worst[1]} ({worst[0]:.2f} ms)")
