"""Device occupancy of a batch run from a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory: over the
LAST `frac` of the trace (steady state), the time in which at least one kernel ran, the time in which >= 2 ran, the sum of
kernel durations by kernel, and the same for copies.
  python tools/batch_timeline.py <dir> [frac=0.5]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ker, cop = [], []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|msak::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0][:36]))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[-14:]))
ker.sort()
if not ker:
    sys.exit("no kernels")
t_end = max(e for _, e, _ in ker)
t_beg = ker[0][0]
lo = t_end - (t_end - t_beg) * frac
win = [(max(s, lo), e, n) for s, e, n in ker if e > lo]
span = t_end - lo
ev = sorted([(s, 1) for s, e, n in win] + [(e, -1) for s, e, n in win])
depth, last, busy1, busy2 = 0, lo, 0, 0
for t, dlt in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += dlt
    last = t
tot = defaultdict(lambda: [0, 0])
for s, e, n in win:
    tot[n][0] += e - s
    tot[n][1] += 1
print(f"window {span / 1e6:.2f} ms: some kernel running {busy1 / span:.3f} of it, two or more {busy2 / span:.3f}; sum of kernel durations {sum(v[0] for v in tot.values()) / span:.2f} x the window")
for n, (ns, k) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:38s} {k:6d} launches  {ns / 1e6:9.3f} ms total  {ns / k / 1e3:9.1f} us each  {ns / span:6.3f} of the window")
cw = [(max(s, lo), e, n) for s, e, n in cop if e > lo]
ctot = defaultdict(lambda: [0, 0])
for s, e, n in cw:
    ctot[n][0] += e - s
    ctot[n][1] += 1
for n, (ns, k) in sorted(ctot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:38s} {k:6d} copies    {ns / 1e6:9.3f} ms total  {ns / k / 1e3:9.1f} us each")
