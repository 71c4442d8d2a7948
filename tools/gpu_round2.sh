#!/bin/bash
# round-2 GPU check: all gpu tests, smoke(), the bench line of every workload, rocprofv3 kernel stats per workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
for w in C3 C2 C4 C5; do
  timeout 900 python bench.py --steps 10 --warmup 2 --workload $w > $OUT/bench_$w.log 2>&1; echo "bench $w rc=$?"
  tail -1 $OUT/bench_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step','value_host_rows','kernels_ms','speedup_vs_cpu_1core','speedup_vs_cpu_all_cores')}); print(d.get('cpu_baseline',{}).get('flavours'))" 2>&1 | cut -c1-700
done
export TMPDIR=/tmp
cd /tmp
for w in C3 C2 C4 C5; do
  rm -rf $OUT/prof_stats_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$w -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $w > $OUT/prof_stats_$w.log 2>&1
  f=$(find $OUT/prof_stats_$w -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
done
