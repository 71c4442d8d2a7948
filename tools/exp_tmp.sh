timeout 400 python tests/fuzz/fuzz_batch.py 150 11 2>&1 | tail -1
timeout 400 python tests/fuzz/fuzz_batch.py 100 12 2>&1 | tail -1
MSA_BATCH_SORT=0 timeout 200 python tests/fuzz/fuzz_batch.py 60 11 2>&1 | tail -1
