ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run sq3 TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
find $OUT -name "*kernel_trace.csv" -size +4M -delete
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc3"
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "similarity_lg" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x)/len(x) for k, x in acc.items()}
for k in sorted(v): print("%-28s %16.0f" % (k, v[k]))
cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
if cyc:
    print("kernel cycles %.3g  VALU busy %.2f  SALU+SMEM per CU-cycle %.2f  TA busy %.2f  TD busy %.2f  LDS %.2f" % (cyc, v["SQ_INSTS_VALU"]*4/1024/cyc, (v["SQ_INSTS_SALU"]+v["SQ_INSTS_SMEM"])/256/cyc, v.get("TA_TA_BUSY_sum",0)/256/cyc, v.get("TD_TD_BUSY_sum",0)/256/cyc, v.get("SQ_LDS_IDX_ACTIVE",0)/256/cyc))
PY
