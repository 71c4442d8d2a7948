#!/bin/bash
# the pieces of tools/gpu_profiles.sh that the front kernel's block size touches, again on the final build
export MSA_DIAGNOSTICS=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final
rm -rf $OUT; mkdir -p $OUT; cd $ROOT
timeout 900 python bench.py --steps 20 --warmup 3 --workload C5 > $OUT/bench_C5.json 2> $OUT/bench_C5.err; echo "bench C5 rc=$?"
timeout 900 python bench.py --steps 20 --warmup 3 --workload C3 > $OUT/bench_C3.json 2> $OUT/bench_C3.err; echo "bench C3 rc=$?"
timeout 300 python tools/c5_counts.py > $OUT/c5_counts.jsonl 2>/dev/null
timeout 300 python tools/small_latency.py > $OUT/small_latency.jsonl 2>/dev/null
timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29547 tools/c5_collective.py 15 > $OUT/c5_collective.jsonl 2> $OUT/c5_collective.err
timeout 300 python tests/measure/fixtures_time.py > $OUT/fixtures_time.jsonl 2>/dev/null
export TMPDIR=/tmp
( cd /tmp; : > $OUT/small_kernel_stats.txt
  for a in "500 2000 strict" "1000 4000 automated1" "600 2500 automated1"; do
    rm -rf /tmp/small_prof
    timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/small_prof -- python3 $ROOT/tools/small_one.py $a 200 > /tmp/small_prof.log 2>&1
    grep "per upload" /tmp/small_prof.log >> $OUT/small_kernel_stats.txt
    f=$(find /tmp/small_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 $f | cut -d, -f1-4 | cut -c1-160 >> $OUT/small_kernel_stats.txt
  done )
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C5 -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload C5 > $OUT/stats_C5.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +4M -delete
cat $OUT/c5_counts.jsonl; tail -2 $OUT/c5_collective.jsonl; cat $OUT/small_kernel_stats.txt
