"""The C5 batch (64 x 1000 x 4000, AutomaticTrimmer('automated1')) on one GPU through the native batch path
(`msa_trim_batch`) by number of worker threads: ms per batch from host rows, columns/s, and the device-busy fraction
(sum of the similarity + pair kernel times of the batch / wall time is NOT it: use the rocprofv3 timeline, tools/gpu_timeline.sh).
   python tools/c5_batch.py [workers ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch  # noqa: E402,F401
from pytrimal_amd import Alignment, AutomaticTrimmer  # noqa: E402
from pytrimal_amd.batch import trim_batch  # noqa: E402
from pytrimal_amd.synth import synth_msa  # noqa: E402

alis = []
for k in range(64):
    a = synth_msa(1000, 4000, 2000 + k)
    alis.append(Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]))
trimmer = AutomaticTrimmer("automated1", platform="hip")
for workers in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 6, 8, 12]:
    trim_batch(trimmer, alis, threads=workers)
    times = []
    for _ in range(5):
        t = time.perf_counter()
        out = trim_batch(trimmer, alis, threads=workers)
        times.append(time.perf_counter() - t)
    best = min(times)
    print(json.dumps({"workers": workers, "ms_per_batch_best": round(best * 1e3, 2), "ms_per_batch_median": round(sorted(times)[2] * 1e3, 2),
                      "columns_per_s": round(64 * 4000 / best), "kept_columns": int(sum(sum(t.residues_mask) for t in out))}), flush=True)
