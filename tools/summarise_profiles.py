"""gpurun_out/<tag> (tools/gpu_profiles.sh <tag>) -> profiles/<tag>_*: bench lines, rocprofv3 kernel stats per workload,
PMC HBM traffic per kernel and workload (-> profiles/traffic.json, read by bench.py as `roofline.traffic`), SQ counters."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", TAG)
DST = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"[<(].*", "", name).strip()


def counters(d):
    # (the newest file: gpurun merges a run's output into what earlier runs of the same tag left behind)
    files = sorted(glob.glob(os.path.join(SRC, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    acc = defaultdict(list)
    if files:
        for r in csv.DictReader(open(files[-1])):
            if "msak::" in r["Kernel_Name"]:
                acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


family = {  # kernel -> the name bench.py's roofline uses
    "msak::similarity_lg_kernel": "sim", "msak::pair_counts_kernel": "pairs", "msak::pair_counts_pipe_kernel": "pairs", "msak::pair_counts_pipe16_kernel": "pairs", "msak::sim_lists_fused_kernel": "encode",
    "msak::gap_counts_kernel": "gaps",
    "msak::prep_planes_kernel": "prep", "msak::identity_rows_kernel": "idstats", "msak::sim_encode_cm_kernel": "encode",
    "msak::bx_compact_kernel": "encode", "msak::cluster_mis_kernel": "cluster", "msak::cluster_adjacency_kernel": "cluster",
}
traffic = {"_source": "profiles/" + TAG + "_pmc_hbm_traffic.txt (builder PMC passes of tools/gpu_profiles.sh, not measured in the bench run)",
           "_note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch -- per pass for the similarity kernel, which a pass launches several times -- (gfx950: FETCH_SIZE counts half of the bytes of wide "
                    "coalesced reads, MI355X_MICROARCH.md); Infinity-Cache hits are counted by these counters"}
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over python3 bench.py --workload W",
         "# KB per dispatch (avg); corrected bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"]
for w in ("C3", "C2", "C4", "C5"):
    f, wr = counters(f"fetch_{w}"), counters(f"write_{w}")
    per = defaultdict(lambda: [0.0, 0.0])
    for (k, c), v in f.items():
        if c == "FETCH_SIZE":
            per[k][0] = sum(v) / len(v)
    for (k, c), v in wr.items():
        if c == "WRITE_SIZE":
            per[k][1] = sum(v) / len(v)
    fam = defaultdict(float)
    for k, (fe, wrr) in sorted(per.items()):
        b = (2 * fe + wrr) * 1024
        lines.append(f"{w} {k:<40} FETCH_SIZE {fe:>12.1f} KB  WRITE_SIZE {wrr:>12.1f} KB  corrected {b / 1e6:>10.2f} MB")
        if k in family:
            fam[family[k]] += b
    for name, b in fam.items():
        traffic[f"{w}:{name}"] = int(b)
    # a similarity pass is several launches of the kernel from 1800 rows on (six rounds each, lg_rounds_per_launch): bench.py's
    # `achieved` is per pass, so is this
    m_of = {"C3": 2000, "C2": 500, "C4": 5000, "C5": 1000}[w]
    rounds = (m_of - 1 + 63) // 64
    passes = (rounds + 5) // 6 if m_of >= 1800 and rounds > 6 else 1
    if f"{w}:sim" in traffic and passes > 1:
        traffic[f"{w}:sim"] *= passes
        lines.append(f"{w} (a similarity pass = {passes} launches: {traffic[f'{w}:sim'] / 1e6:.1f} MB per pass)")
open(os.path.join(DST, TAG + "_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)

for w in ("C3", "C2", "C4", "C5"):
    files = sorted(glob.glob(os.path.join(SRC, f"stats_{w}", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if files:
        shutil.copy(files[-1], os.path.join(DST, f"{TAG}_rocprofv3_kernel_stats_{w.lower()}.csv"))
    b = os.path.join(SRC, f"bench_{w}.json")
    if os.path.exists(b):
        last = [ln for ln in open(b).read().splitlines() if ln.startswith("{")]
        if last:
            open(os.path.join(DST, f"{TAG}_bench_{w.lower()}.json"), "w").write(last[-1] + "\n")

out = ["# rocprofv3 --pmc <SQ / GRBM counters> --kernel-trace over python3 bench.py --steps 3 --warmup 1; averages per dispatch",
       "# (a similarity pass at C3 is six dispatches of similarity_lg_kernel: DESIGN 5.8).",
       "# SQ_* cycle counters count quad-cycles summed over waves (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs:",
       "# kernel cycles = GRBM_GUI_ACTIVE / 8.  VALU busy = SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel cycles)."]
for tag, dirs in (("C3", ("sq1_C3", "sq2_C3", "sq3_C3")), ("C4", ("sq1_C4",))):
    acc = {}
    for d in dirs:
        acc.update(counters(d))
    kernels = sorted({k for k, _ in acc})
    for k in kernels:
        vals = {c: sum(v) / len(v) for (kk, c), v in acc.items() if kk == k}
        if vals.get("SQ_INSTS_VALU", 0) < 1e6:
            continue
        out.append(f"{tag} {k}")
        for c in sorted(vals):
            out.append(f"    {c:<24} {vals[c]:>16.0f}")
        if "GRBM_GUI_ACTIVE" in vals and "SQ_INSTS_VALU" in vals:
            cyc = vals["GRBM_GUI_ACTIVE"] / 8
            out.append(f"    -> kernel cycles {cyc:.3g}, VALU busy {vals['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.2f}")
        if "SQ_INSTS_VMEM_RD" in vals and "SQ_INSTS_VALU" in vals and "similarity_" in k:
            out.append(f"    -> vector loads {vals['SQ_INSTS_VMEM_RD']:.3g} (one per partner step + the ordered rows), VALU / load "
                       f"{vals['SQ_INSTS_VALU'] / vals['SQ_INSTS_VMEM_RD']:.2f}, SALU / load "
                       f"{vals.get('SQ_INSTS_SALU', 0) / vals['SQ_INSTS_VMEM_RD']:.2f}")
        if "TA_TA_BUSY_sum" in vals and "GRBM_GUI_ACTIVE" in vals:
            cyc = vals["GRBM_GUI_ACTIVE"] / 8
            out.append(f"    -> texture addresser busy {vals['TA_TA_BUSY_sum'] / 256 / cyc:.2f} of the kernel (sum over 256 CUs), "
                       f"texture data {vals.get('TD_TD_BUSY_sum', 0) / 256 / cyc:.2f}, LDS {vals.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / cyc:.2f}")
open(os.path.join(DST, TAG + "_pmc_sq.txt"), "w").write("\n".join(out) + "\n")
ub = os.path.join(SRC, "ubench_wstream.txt")
if os.path.exists(ub):
    shutil.copy(ub, os.path.join(DST, TAG + "_ubench_wstream.txt"))
for name in ("ubench_wform.txt", "sim_shapes.jsonl", "sim_shapes_one_wave_per_column.jsonl", "small_batch.jsonl", "c5_engine.jsonl", "c5_counts.jsonl",
             "small_latency.jsonl", "small_latency_ordinary_launch_sequence.jsonl", "flat_sweep.jsonl", "fixtures_time.jsonl", "small_kernel_stats.txt", "cold_upload.jsonl",
             "front_pairs_ab.jsonl", "small_batch_engine_kinds.jsonl", "ubench_lstrip.txt"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, TAG + "_" + name))
for name in ("bx_stamps.jsonl", "ab_switches.txt", "timeline_C3.txt", "timeline_C2.txt", "timeline_C4.txt", "reference_shape.jsonl",
             "c5_batch.jsonl", "upload.txt", "sim_by_data.jsonl", "c5_timeline.txt", "bench_REF.json", "pmc_sim.txt", "sim_fixture_stamps.txt", "sim_overlap.jsonl"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, TAG + "_" + name))
if os.path.exists(os.path.join(SRC, "c5_collective.jsonl")):  # (profiles/<tag>_c5_collective.jsonl is the hand-assembled A/B with round 5's path)
    shutil.copy(os.path.join(SRC, "c5_collective.jsonl"), os.path.join(DST, TAG + "_c5_collective_final_build.jsonl"))
print("\n".join(lines[-40:]))
print("\n".join(out))
