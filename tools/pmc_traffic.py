"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_profile.sh into
profiles/r01_pmc_hbm_traffic_c3.txt and profiles/traffic.json (bytes per launch, gfx950 corrections per
MI355X_MICROARCH.md: FETCH_SIZE counts half of the bytes of wide coalesced reads -> x2; units are KB)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out")


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not files:
        sys.exit("no file matches " + pattern)
    return files[-1]


def collect(path):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
            if "msak::" in name:
                acc[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


rows = {}
for d in ("prof_fetch", "prof_write"):
    rows.update(collect(newest(os.path.join(out, d, "*", "*_counter_collection.csv"))))
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 bench.py --steps 3 --warmup 1 (C3)",
         "# units: KB per dispatch (avg).  gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> corrected = 2x"]
avg = {}
for (name, counter), vals in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    avg[(name, counter)] = sum(vals) / len(vals)
    lines.append("%-48s %-11s n=%d avg_KB=%.1f" % (name, counter, len(vals), avg[(name, counter)]))


def traffic(kernel):
    f = [v for (n, c), v in avg.items() if kernel in n and c == "FETCH_SIZE"]
    w = [v for (n, c), v in avg.items() if kernel in n and c == "WRITE_SIZE"]
    return int((2 * sum(f) + sum(w)) * 1024)


t = {"C3:simnum": traffic("similarity_num_kernel"),
     "C3:simden": traffic("sim_den_kernel") + traffic("sim_den2_kernel") + traffic("den_pairmask_kernel"),
     "C3:pairs": traffic("pair_counts_kernel"), "C3:gaps": traffic("gap_counts_kernel")}
t["C3:sim"] = t["C3:simnum"] + t["C3:simden"]
t["_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch; 'sim' = numerator + denominator kernel; "
              "see profiles/r01_pmc_hbm_traffic_c3.txt")
lines.append("# bytes per launch (corrected): " + json.dumps({k: v for k, v in t.items() if k != "_note"}))
open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_c3.txt"), "w").write("\n".join(lines) + "\n")
json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("\n".join(lines))
