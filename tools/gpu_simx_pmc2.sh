#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_bx3
BX_STAMP_R0=8 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/prof_bx3 -- python3 $ROOT/tools/bx_stamps.py > $OUT/prof_bx3.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, os
from collections import defaultdict
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for d in ("prof_bx3",):
    files = sorted(glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")))
    if not files:
        print("missing", d, open(os.path.join(out, d + ".log")).read()[-1500:]); continue
    rows = defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "similarity_bx" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(rows.items()):
        print(d, k, "%.4g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
