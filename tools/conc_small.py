import sys, time
sys.path.insert(0, "/root/repo")
import torch
from multiprocessing.pool import ThreadPool
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.synth import synth_msa
cases = []
for seed in range(18):
    a = synth_msa(120 + 37 * (seed % 5), 300 + 64 * (seed % 4), 8800 + seed)
    cases.append(Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]))
tr = AutomaticTrimmer("strict", platform="hip")
for threads in (1, 2, 4):
    with ThreadPool(threads) as pool:
        t = time.perf_counter(); pool.map(tr.trim, cases); t1 = time.perf_counter() - t
        t = time.perf_counter(); pool.map(tr.trim, cases); t2 = time.perf_counter() - t
    print(threads, "threads: first pass %.3f s, second %.3f s" % (t1, t2), flush=True)
