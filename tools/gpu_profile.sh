#!/bin/bash
# full GPU check: all gpu tests, smoke(), bench, rocprofv3 kernel stats, PMC HBM traffic passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log | cut -c1-400
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1
cd $ROOT
find $OUT/prof_stats -name "*kernel_stats*.csv" | head -2
f=$(find $OUT/prof_stats -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -20 "$f"
ls -R $OUT | head -40
