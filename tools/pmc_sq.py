"""Summarise the SQ / GRBM counter passes of tools/gpu_pmc_sq.sh into profiles/r01_pmc_sq_c3.txt."""
import csv, glob, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out")
rows = defaultdict(list)
dur = defaultdict(list)
for d in ("prof_sq", "prof_sq2"):
    files = sorted(glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not files:
        sys.exit("missing " + d)
    with open(files[-1], newline="") as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            if "msak::" not in name:
                continue
            rows[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
kernels = sorted({k for k, _ in rows}, key=lambda k: -sum(dur[k]) / len(dur[k]))
counters = sorted({c for _, c in rows})
lines = ["# rocprofv3 --pmc <SQ/GRBM counters> --kernel-trace, python3 bench.py --steps 3 --warmup 1 (C3); averages per dispatch.",
         "# SQ_* cycle counters count quad-cycles summed over waves (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs:",
         "# effective clock = GRBM_GUI_ACTIVE / 8 / kernel time."]
for k in kernels[:6]:
    ms = sum(dur[k]) / len(dur[k])
    lines.append(f"{k}  ({ms:.3f} ms per dispatch under the profiler)")
    for c in counters:
        v = rows.get((k, c))
        if v:
            a = sum(v) / len(v)
            extra = ""
            if c == "GRBM_GUI_ACTIVE":
                extra = f"   -> effective clock {a / 8 / (ms * 1e-3) / 1e9:.2f} GHz"
            lines.append(f"    {c:<22} {a:>16.0f}{extra}")
text = "\n".join(lines) + "\n"
open(os.path.join(ROOT, "profiles", "r01_pmc_sq_c3.txt"), "w").write(text)
print(text)
