python -m pytest tests/test_gpu_trimmers.py -x -q -m gpu -k "one_wait" 2>&1 | tail -3
