"""Similarity trim over a sweep of shapes around the numerator kernel's residency limits (m <= 2016: 18 rounds of
codes in registers, m <= 4032: 36 rounds, above: codes streamed): cycles per pair step of both chain kernels."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
from pytrimal_amd import Alignment, ManualTrimmer, _lib
from pytrimal_amd.synth import synth_msa

shapes = [(int(x.split("x")[0]), int(x.split("x")[1])) for x in sys.argv[1:]] or [(1000, 4000), (2000, 4000), (2016, 4000),
          (2100, 4000), (3000, 4000), (4032, 4000), (4100, 4000), (6000, 4000), (8000, 2000)]
tr = ManualTrimmer(similarity_threshold=0.5, platform="hip")
for m, n in shapes:
    a = synth_msa(m, n, m + n)
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
    tr.trim(ali)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); tr.trim(ali); ts.append(time.perf_counter() - t)
    ctx = _lib.thread_context()
    ctx.prof_enable(True); ctx.prof_reset(); tr.trim(ali)
    k = {nm: ctx.prof_get(nm)[0] / max(1, ctx.prof_get(nm)[1]) for nm in ("pairs", "sim", "simnum", "simden")}
    ctx.prof_enable(False)
    steps = m * (m - 1) / 2
    print(json.dumps({"m": m, "n": n, "trim_ms": round(float(np.median(ts)) * 1e3, 2), "pairs_ms": round(k["pairs"], 3),
                      "sim_ms": round(k["sim"], 2), "num_ms": round(k["simnum"], 2), "den_ms": round(k["simden"], 2),
                      "num_ns_per_step": round(k["simnum"] * 1e6 / steps, 2), "den_ns_per_step": round(k["simden"] * 1e6 / steps, 2)}), flush=True)
