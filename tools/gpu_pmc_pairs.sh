# SQ counters of the pair pass at C4 (5000 x 5000): dense codes against the raw planes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pairs
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for dense in 1 0; do
  export MSA_PAIR_DENSE=$dense
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1_$dense -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C4 > $OUT/sq1_$dense.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_VMEM TA_TA_BUSY_sum SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2_$dense -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C4 > $OUT/sq2_$dense.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f_$dense -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C4 > $OUT/f_$dense.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_pairs"
for dense in ("1", "0"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/*_" + dense + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pair_counts" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = {k: sum(x)/len(x) for k, x in acc.items()}
    print("MSA_PAIR_DENSE=" + dense)
    for k in sorted(v): print("    %-28s %16.0f" % (k, v[k]))
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc: print("    kernel cycles %.3g  VALU busy %.2f  waves %d  wave-cycles/wave %.0f" % (cyc, v["SQ_INSTS_VALU"]*4/1024/cyc, v.get("SQ_WAVES", 0), v["SQ_WAVE_CYCLES"]*4/max(v.get("SQ_WAVES", 1),1)))
PY
