#!/bin/bash
# PMC passes over the similarity kernel (bench.py --steps 3): SQ instruction mix / waits, LDS, L1 (TCP) and L2 (TCC) traffic.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lg
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVES
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_WAVE32_LDS
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_READ_sum TCC_TAG_STALL_sum
run ta TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUFFER_LOAD_WAVEFRONTS_sum
find $OUT -name "*kernel_trace.csv" -size +4M -delete
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_lg"
for d in sorted(glob.glob(out + "/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "similarity" not in k: continue
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(d.rstrip("/")), k)
        for c, v in sorted(cs.items()):
            print("    %-36s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
