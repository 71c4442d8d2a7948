#!/bin/bash
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab4
mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "engine or batch" > $OUT/pytest_engine.txt 2>&1; echo "pytest engine rc=$?"; tail -15 $OUT/pytest_engine.txt
timeout 400 python tests/fuzz/fuzz_batch.py 150 611 > $OUT/fuzz_batch.txt 2>&1; echo "fuzz_batch rc=$?"; tail -c 1500 $OUT/fuzz_batch.txt
for meth in gappyout overlap representative noduplicateseqs strict; do
  timeout 300 python tools/small_batch.py 1024 100 1000 $meth 2>/dev/null >> $OUT/small_batch.jsonl
  MSA_BATCH_ENGINE=0 timeout 300 python tools/small_batch.py 1024 100 1000 $meth 2>/dev/null >> $OUT/small_batch.jsonl
done
for meth in overlap representative; do
  timeout 300 python tools/small_batch.py 256 300 1200 $meth 2>/dev/null >> $OUT/small_batch.jsonl
  MSA_BATCH_ENGINE=0 timeout 300 python tools/small_batch.py 256 300 1200 $meth 2>/dev/null >> $OUT/small_batch.jsonl
done
cat $OUT/small_batch.jsonl
