#!/bin/bash
export MSA_DIAGNOSTICS=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab6
mkdir -p $OUT; cd $ROOT
for t in 3 4 5 6 4 5; do echo "== THREADS=$t" >> $OUT/c5_threads.txt; THREADS=$t timeout 300 python tools/c5_counts.py 2>/dev/null | head -4 >> $OUT/c5_threads.txt; done
cat $OUT/c5_threads.txt
bash tools/gpu_fuzz.sh 150 120 400
cp gpurun_out/fuzz.txt $OUT/fuzz.txt
