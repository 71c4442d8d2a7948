#!/bin/bash
# first checks of the binade-exact similarity kernel: parity tests, then C3 timings of the three kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "similarity or binade or random_small or nucleotide or wide" > $OUT/simx_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/simx_pytest.log
tail -15 $OUT/simx_pytest.log
for k in "" chain; do
  MSA_SIM_KERNEL=$k timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/simx_bench_$k.log 2>&1
  echo "kernel=[$k]"; tail -1 $OUT/simx_bench_$k.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'], d['config'].get('kept_columns'))"
done
for c in 1 2; do
  MSA_BX_COLS=$c timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/simx_bench_c$c.log 2>&1
  echo "bx cols=$c"; tail -1 $OUT/simx_bench_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'])"
done
