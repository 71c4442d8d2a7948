#!/bin/bash
# SQ counters of the binade-exact kernel (two passes), the program under the profiler is tools/bx_stamps.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_bx1 $OUT/prof_bx2
BX_STAMP_R0=8 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/prof_bx1 -- python3 $ROOT/tools/bx_stamps.py > $OUT/prof_bx1.log 2>&1
BX_STAMP_R0=8 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/prof_bx2 -- python3 $ROOT/tools/bx_stamps.py > $OUT/prof_bx2.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, os
from collections import defaultdict
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for d in ("prof_bx1", "prof_bx2"):
    files = sorted(glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")))
    if not files:
        print("missing", d, open(os.path.join(out, d + ".log")).read()[-600:]); continue
    rows = defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "similarity_bx" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(rows.items()):
        print(d, k, "%.4g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
