"""The CPU port (the oracle) on 16..128 host threads: columns/s of oracle.trim('automated1') over column slices of C3."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np, oracle
from multiprocessing.pool import ThreadPool
from pytrimal_amd.synth import synth_msa
a = synth_msa(2000, 10000, 1003)
per = 256
base = [np.ascontiguousarray(a[:, i * per:(i + 1) * per]) for i in range(39)]
for threads in (16, 32, 48, 64, 96, 128):
    sl = [base[i % 39] for i in range(threads)]
    t = time.perf_counter()
    with ThreadPool(threads) as pool: pool.map(lambda x: oracle.trim(x, method="automated1"), sl)
    dt = time.perf_counter() - t
    print(threads, "threads: %.2f s -> %.0f columns/s" % (dt, threads * per / dt), flush=True)
