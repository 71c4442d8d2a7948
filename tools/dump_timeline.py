"""Every kernel and copy of a rocprofv3 --kernel-trace --memory-copy-trace csv directory behind the LAST `marker` kernel
(default: the last gap_counts launch's step), with start offsets:  python tools/dump_timeline.py <dir> [count]"""
import csv, glob, os, re, sys
d = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 70
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|msak::|void ", "", r["Kernel_Name"]).split("(")[0][:40], "q" + r.get("Queue_Id", "?")))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[-16:] + " " + r.get("Size", r.get("Bytes", "")), "dma"))
ev.sort()
ev = ev[-count:]
t0 = ev[0][0]
for s, e, name, q in ev:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {q:>4}  {name}")
