"""First upload of rows the runtime has never seen (a fresh array per sample) against re-uploads of one array, by upload path:
MSA_UPLOAD_DIRECT=1 (the runtime's pitched copy from pageable memory) / 0 (packed into pinned pieces by the calling thread and
the helper threads).   python tools/cold_upload.py [m n]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa

m, n = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (2000, 10000)))
a = synth_msa(m, n, 5)
for direct in ("1", "0"):
    os.environ["MSA_UPLOAD_DIRECT"] = direct
    ctx = _lib.Context(0)
    ctx.upload(a, ord("X"))
    ctx.gaps()
    fresh, again = [], []
    for _ in range(8):
        b = a.copy()
        t = time.perf_counter(); ctx.upload(b, ord("X")); fresh.append(time.perf_counter() - t)
        t = time.perf_counter(); ctx.upload(b, ord("X")); again.append(time.perf_counter() - t)
    ctx.close()
    print(json.dumps({"m": m, "n": n, "MSA_UPLOAD_DIRECT": direct, "first_upload_of_a_fresh_array_ms": round(float(np.median(fresh)) * 1e3, 3),
                      "second_upload_ms": round(float(np.median(again)) * 1e3, 3)}), flush=True)
