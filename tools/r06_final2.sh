#!/bin/bash
# behind the staggered halves of the similarity kernel's columns: the GPU suite, the A/B by shape, C3 with and without, the by-shape table
export MSA_DIAGNOSTICS=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final2
rm -rf $OUT; mkdir -p $OUT; cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
timeout 1500 python tools/tall_halves_ab.py > $OUT/tall_halves_ab.jsonl 2>/dev/null; echo "halves ab rc=$?"
for rep in 1 2; do for v in "MSA_LG_HALVES=1" "MSA_LG_HALVES=0"; do
  echo "== C3 $v" >> $OUT/c3_halves.txt
  env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('ms_per_step_median_of_5_regions'), r.get('ms_per_step_resident'), r.get('kernels_ms'))" >> $OUT/c3_halves.txt
done; done
timeout 900 python bench.py --steps 20 --warmup 3 --workload C3 > $OUT/bench_C3.json 2> $OUT/bench_C3.err; echo "bench C3 rc=$?"
timeout 600 python bench.py --workload REF --out $OUT/reference_shape.jsonl > $OUT/bench_REF.json 2> $OUT/bench_REF.err
timeout 600 python tools/sim_shapes.py > $OUT/sim_shapes.jsonl 2>/dev/null
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload C3 > $OUT/stats_C3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_C3 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C3 > $OUT/fetch_C3.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_C3 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C3 > $OUT/write_C3.log 2>&1
cd $ROOT
find $OUT -name "*kernel_trace.csv" -size +4M -delete
cat $OUT/c3_halves.txt; cut -c1-200 $OUT/tall_halves_ab.jsonl
