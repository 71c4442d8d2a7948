#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "similarity or binade" > $OUT/simx_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/simx_pytest.log
tail -4 $OUT/simx_pytest.log
BX_RECORDS=1 BX_STAMP_R0=${BX_STAMP_R0:-8,4} timeout 600 python tools/bx_stamps.py > $OUT/bx_stamps.log 2>&1; tail -8 $OUT/bx_stamps.log
