"""Do similarity kernels of different contexts (= different HIP streams) overlap on the chip?  T threads, each with its own
context and its own 1000 x 4000 alignment, run `reps` similarity passes (pair pass and layout kernels included) back to back;
printed: wall time per pass by number of threads.  If the kernels of different streams ran side by side, two threads would
need less than twice the time of one per pass each (one alignment's 3683 waves leave a quarter of the wave slots free, and
half of them in its tail).   python tools/sim_overlap.py [m n reps]
RESIDENT=1: no upload between the passes -- W stays, a pass is the layout kernels (0.03 ms) and the similarity kernel alone."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

m, n, reps = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1000, 4000, 40)))
vhash, dist = SimilarityMatrix.aa()._device_arrays()
resident = os.environ.get("RESIDENT", "0") == "1"


def worker(k, barrier, out):
    ctx = _lib.Context(0)
    a = synth_msa(m, n, 2000 + k)
    ctx.upload(a, ord("X"))
    ctx.similarity(vhash, dist)
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(reps):
        if not resident:
            ctx.upload(a, ord("X"))  # (drops W: the pass runs the pair pass and the layouts again)
        ctx.similarity(vhash, dist)
    out[k] = time.perf_counter() - t0
    ctx.close()


for T in [int(x) for x in os.environ.get("THREADS", "1,2,3,4,6").split(",")]:
    out = [0.0] * T
    barrier = threading.Barrier(T)
    th = [threading.Thread(target=worker, args=(k, barrier, out)) for k in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(out)
    print(json.dumps({"threads": T, "m": m, "n": n, "passes": T * reps, "wall_ms": round(wall * 1e3, 2), "ms_per_pass": round(wall * 1e3 / (T * reps), 4), "resident": resident}), flush=True)
