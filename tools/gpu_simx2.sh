#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/bx_stamps.py > $OUT/bx_stamps.log 2>&1; cat $OUT/bx_stamps.log | tail -12
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_bx1 $OUT/prof_bx2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/prof_bx1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_bx1.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/prof_bx2 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_bx2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/prof_bx3 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_bx3.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for d in ("prof_bx1", "prof_bx2", "prof_bx3"):
    files = sorted(glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")))
    if not files:
        print("missing", d); continue
    rows = defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "similarity_bx" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(rows.items()):
        print(d, k, "%.4g" % (sum(v) / len(v)))
PY
