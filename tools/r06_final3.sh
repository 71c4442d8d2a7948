#!/bin/bash
export MSA_DIAGNOSTICS=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final3
rm -rf $OUT; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_parity.py -x -q -k "dispatch or tall or many_rows or beyond or variants" > $OUT/pytest_sel.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_sel.txt
timeout 2400 python tools/tall_halves_ab.py 3000 8000 6 3583 7287 1003 4500 6000 4 2000 10000 1003 2000 3000 2 5000 5000 1004 8000 3000 5 3000 1500 9 4000 2000 3 8000 1500 6 9000 640 5 10000 700 1 12000 800 2 14000 700 6 16000 600 2 16000 1000 6 20000 1200 7 24000 900 8 30000 1200 9 10000 500 1 12000 500 3 20000 500 3 30000 400 8 40000 300 4 > $OUT/tall_halves_ab.jsonl 2>/dev/null; echo "ab rc=$?"
for r in 2 3 4 6; do echo "# ROUNDS=$r (rounds per launch of both legs forced)" >> $OUT/tall_halves_rounds.jsonl; ROUNDS=$r timeout 900 python tools/tall_halves_ab.py 8000 1500 6 12000 800 2 16000 1000 6 20000 1200 7 20000 500 3 >> $OUT/tall_halves_rounds.jsonl 2>/dev/null; done
timeout 600 python tools/sim_shapes.py > $OUT/sim_shapes.jsonl 2>/dev/null
CHECK=0 REPS=3 timeout 600 python tools/sim_shapes.py 12000 800 2 20000 1200 7 30000 1200 9 > $OUT/sim_shapes_tall_wide.jsonl 2>/dev/null
timeout 600 python bench.py --workload REF --out $OUT/reference_shape.jsonl > $OUT/bench_REF.json 2> $OUT/bench_REF.err
cut -c1-150 $OUT/sim_shapes_tall_wide.jsonl
