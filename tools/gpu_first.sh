#!/bin/bash
# first GPU contact: parity tests, instruction-timing probes, a short bench
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/rocminfo.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 120 ./tools/ubench > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log
timeout 600 python bench.py --steps 3 --warmup 1 --workload C2 > gpurun_out/bench_c2.log 2>&1; tail -3 gpurun_out/bench_c2.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_c3.log 2>&1; tail -3 gpurun_out/bench_c3.log
