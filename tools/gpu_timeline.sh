#!/bin/bash
# kernel + copy timeline of one steady-state step per workload (tools/step_timeline.py reads the csv)
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in C3 C2 C4; do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$w -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload $w > $OUT/$w.log 2>&1
  python3 $ROOT/tools/step_timeline.py $OUT/$w > $OUT/timeline_$w.txt 2>&1
  find $OUT/$w -name "*.csv" -size +2M -delete
done
cd $ROOT
for s in "1000 4000 2000" "500 2000 1002"; do timeout 120 python tools/bx_stamps.py $s 2>/dev/null | grep sim_ms; done > $OUT/stamps_small.jsonl
cat $OUT/timeline_C3.txt
