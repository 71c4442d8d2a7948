"""Host rows -> device by shape, ms per upload: page-locked rows (one pitched DMA copy), pageable rows (the runtime's
pitched copy when the rows are 16-byte multiples, else packed pinned pieces), and the packed pieces forced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np, torch
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa
for (m, n, seed) in ((2000, 10000, 1003), (1000, 4000, 2000), (5000, 5000, 1004), (500, 2000, 1002), (209, 1227, 5)):
    a = synth_msa(m, n, seed)
    ctx = _lib.Context(0)
    for pin in (False, True):
        for _ in range(3): ctx.upload(a, ord("X"), pin=pin)
        t = time.perf_counter()
        for _ in range(20): ctx.upload(a, ord("X"), pin=pin)
        print(m, n, "page-locked" if pin else ("pageable (MSA_UPLOAD_DIRECT=%s)" % os.environ.get("MSA_UPLOAD_DIRECT", "1")), "upload ms",
              round((time.perf_counter() - t) / 20 * 1e3, 4), flush=True)
        g = ctx.gaps(); assert np.array_equal(g, (a == ord("-")).sum(axis=0))
    ctx.close()
    del a
