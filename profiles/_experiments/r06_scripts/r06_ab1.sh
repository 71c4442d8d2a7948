#!/bin/bash
# round 6, first A/B on the GPU box: the narrow front kernel and the sixteen-row pair tiles against round 5's kernels
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab1
mkdir -p $OUT; cd $ROOT
timeout 900 python tools/front_pairs_ab.py check > $OUT/check.txt 2>&1; echo "check rc=$?"; tail -3 $OUT/check.txt
timeout 900 python tools/front_pairs_ab.py time > $OUT/time.jsonl 2>$OUT/time.err; echo "time rc=$?"
timeout 200 python tests/fuzz/fuzz_trim.py 90 61 tall > $OUT/fuzz_tall.txt 2>&1; echo "fuzz tall rc=$?"; tail -2 $OUT/fuzz_tall.txt
timeout 100 python tests/fuzz/fuzz_trim.py 40 62 > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz.txt
for rep in 1 2; do
  echo "== default" >> $OUT/c5_counts.txt; timeout 300 python tools/c5_counts.py >> $OUT/c5_counts.txt 2>/dev/null
  echo "== round 5 kernels" >> $OUT/c5_counts.txt; MSA_FRONT_CW=64 MSA_PAIR_TI=8 timeout 300 python tools/c5_counts.py >> $OUT/c5_counts.txt 2>/dev/null
  echo "== new front only" >> $OUT/c5_counts.txt; MSA_PAIR_TI=8 timeout 300 python tools/c5_counts.py >> $OUT/c5_counts.txt 2>/dev/null
  echo "== new pairs only" >> $OUT/c5_counts.txt; MSA_FRONT_CW=64 timeout 300 python tools/c5_counts.py >> $OUT/c5_counts.txt 2>/dev/null
done
cat $OUT/time.jsonl | cut -c1-400
cat $OUT/c5_counts.txt
