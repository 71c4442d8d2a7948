python -m pytest tests/test_gpu_dispatch.py -x -q -m gpu 2>&1 | tail -4
python tools/upload_time.py 2>&1 | grep -v amdgpu.ids
MSA_UPLOAD_DIRECT=0 python tools/upload_time.py 2>&1 | grep -v amdgpu.ids | grep pageable
python bench.py --workload C4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: r[k] for k in r if 'ms_per_step' in k})"
