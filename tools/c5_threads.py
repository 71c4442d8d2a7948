"""C5-style batch on one GPU: threads x tile width (diagnostics)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from multiprocessing.pool import ThreadPool
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.synth import synth_msa

batch = [synth_msa(1000, 4000, 2000 + k) for k in range(16)]
alis = [Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]) for a in batch]
trimmer = AutomaticTrimmer("automated1", platform="hip")
for threads in (1, 2, 3, 4, 6, 8):
    with ThreadPool(threads) as pool:
        pool.map(trimmer.trim, alis[:threads])
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            pool.map(trimmer.trim, alis)
            best = min(best, time.perf_counter() - t)
    print(json.dumps({"tcols": os.environ.get("MSA_SIM_TCOLS", "64"), "hwq": os.environ.get("GPU_MAX_HW_QUEUES", "-"),
                      "threads": threads, "ms": round(best * 1e3, 2), "columns_per_s": round(len(alis) * 4000 / best)}), flush=True)
