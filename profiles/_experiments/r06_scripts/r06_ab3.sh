#!/bin/bash
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab3
mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
for rep in 1 2; do
  echo "== default (16 rows from 513)" >> $OUT/pairs_shapes.txt; CHECK=0 REPS=6 timeout 300 python tools/sim_shapes.py 3000 8000 5 3583 7287 1003 2000 10000 1003 1500 6000 7 2>/dev/null | cut -c1-200 >> $OUT/pairs_shapes.txt
  echo "== MSA_PAIR_TI=8" >> $OUT/pairs_shapes.txt; MSA_PAIR_TI=8 CHECK=0 REPS=6 timeout 300 python tools/sim_shapes.py 3000 8000 5 3583 7287 1003 2000 10000 1003 1500 6000 7 2>/dev/null | cut -c1-200 >> $OUT/pairs_shapes.txt
done
for t in 4 8 4 8; do echo "== THREADS=$t" >> $OUT/c5_threads.txt; THREADS=$t timeout 300 python tools/c5_counts.py >> $OUT/c5_threads.txt 2>/dev/null; done
timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/c5_collective.py 15 > $OUT/c5_collective.jsonl 2> $OUT/c5_collective.err; echo "collective rc=$?"
export TMPDIR=/tmp; cd /tmp
for v in "" 8; do
  export MSA_PAIR_TI=$v; [ -z "$v" ] && unset MSA_PAIR_TI
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_C3_ti$v -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C3 > $OUT/fetch_C3_ti$v.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_C3_ti$v -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C3 > $OUT/write_C3_ti$v.log 2>&1
done
unset MSA_PAIR_TI
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06_ab3")
for d in sorted(glob.glob(out + "/fetch_C3_*")) + sorted(glob.glob(out + "/write_C3_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pair_counts" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(os.path.basename(d), k, "avg KB per launch %.1f" % (sum(v) / len(v)), "launches", len(v))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/pairs_shapes.txt; cat $OUT/c5_threads.txt; cat $OUT/c5_collective.jsonl
