#!/bin/bash
export MSA_DIAGNOSTICS=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab5
mkdir -p $OUT; cd $ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
for rep in 1 2; do
  for v in "MSA_LISTS_FUSED=1" "MSA_LISTS_FUSED=0"; do
    echo "== C3 $v" >> $OUT/lists.txt
    env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('ms_per_step_resident'), r.get('kernels_ms'))" >> $OUT/lists.txt
    echo "== shapes $v" >> $OUT/lists.txt
    env $v CHECK=0 REPS=5 timeout 300 python tools/sim_shapes.py 3583 7287 1003 1500 1500 7 8000 3000 5 2>/dev/null | cut -c1-160 >> $OUT/lists.txt
  done
done
cat $OUT/lists.txt
