#!/bin/bash
# iteration loop on the GPU box: parity tests, then the bench (C2 quick, C3 headline)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_c3.log 2>&1; tail -2 gpurun_out/bench_c3.log
