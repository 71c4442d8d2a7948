#!/bin/bash
# kernel timeline of the C5 batch on one GPU through the native batch path (tools/batch_timeline.py reads the csv)
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-4}
OUT=$ROOT/gpurun_out/c5tl
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/w$W -- python3 $ROOT/tools/c5_batch.py $W > $OUT/w$W.log 2>&1
python3 $ROOT/tools/batch_timeline.py $OUT/w$W 0.5 > $OUT/timeline_w$W.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/w$W.log | tail -2; cat $OUT/timeline_w$W.txt
