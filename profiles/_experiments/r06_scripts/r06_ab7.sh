#!/bin/bash
export MSA_DIAGNOSTICS=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab7
mkdir -p $OUT; cd $ROOT
for rep in 1 2; do
for v in "" "MSA_FRONT_NT=512" "MSA_FRONT_NT=256" "MSA_FRONT_CW=32 MSA_FRONT_NT=256" "MSA_PAIR_K=4" "MSA_FRONT_NT=256 MSA_PAIR_K=4"; do
  echo "== $v" >> $OUT/c5_nt.txt; env $v timeout 300 python tools/c5_counts.py 2>/dev/null | head -4 >> $OUT/c5_nt.txt
done; done
cat $OUT/c5_nt.txt
