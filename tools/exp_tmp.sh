timeout 700 python tests/fuzz/fuzz_trim.py 600 4242 2>/dev/null | tail -1
timeout 400 python tests/fuzz/fuzz_batch.py 300 777 2>/dev/null | tail -1
timeout 400 python tools/cross_check.py 400 99 2>/dev/null | tail -1
