"""Timeline of one bench step (the middle one, or the k-th as third argument) from a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory:
every kernel / copy with its start offset, duration and queue, and the time in which the device ran nothing.
  python tools/step_timeline.py <dir> [first kernel of a step, default prep_planes_kernel]"""
import csv, glob, os, re, sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "prep_planes_kernel"
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|msak::|void ", "", r["Kernel_Name"]).split("(")[0][:40], "q" + r.get("Queue_Id", "?")))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[-12:] + " " + r.get("Size", r.get("Bytes", "")), "dma"))
ev.sort()
starts = [i for i, e in enumerate(ev) if first in e[2]]
if len(starts) < 2:
    sys.exit("no two steps found")
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) // 2
lo, hi = starts[k - 1], starts[k]
t0 = ev[lo][0]
busy_end = t0
idle = 0
for s, e, name, q in ev[lo:hi]:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {q:>4}  {name}" + (f"   (device idle {gap / 1e3:.1f} us before)" if gap > 2000 else ""))
    busy_end = max(busy_end, e)
print(f"step {(ev[hi][0] - t0) / 1e3:.1f} us, of which nothing running {(idle + max(0, ev[hi][0] - busy_end)) / 1e3:.1f} us")
