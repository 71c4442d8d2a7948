#!/bin/bash
# SQ / GRBM counter pass over the C3 bench (own run: no stats, no PMC of other blocks)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_sq $OUT/prof_sq2
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/prof_sq -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_sq.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/prof_sq2 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_sq2.log 2>&1
cd $ROOT
tail -2 $OUT/prof_sq.log | cut -c1-200
ls $OUT/prof_sq/*/ $OUT/prof_sq2/*/ 2>/dev/null | head
