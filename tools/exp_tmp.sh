RESIDENT=1 THREADS=3,5,3,7,8,3 python tools/sim_overlap.py 2>/dev/null
python tools/c5_batch.py 3 5 3 4 2>/dev/null
