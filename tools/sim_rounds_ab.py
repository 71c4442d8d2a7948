"""The similarity kernel by rounds per launch (MSA_LG_ROUNDS: 0 = one launch) and shape: kernel ms from the context's HIP
events (all launches of a pass together), Q compared bit for bit with the one-launch run.   python tools/sim_rounds_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

vhash, dist = SimilarityMatrix.aa()._device_arrays()
shapes = [(2000, 10000, 1003), (3000, 8000, 7), (3583, 7287, 1003), (5000, 5000, 1004), (8000, 3000, 5)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in sys.argv[1:4])]
for m, n, seed in shapes:
    a = synth_msa(m, n, seed)
    rec, q0 = {"m": m, "n": n}, None
    for per in os.environ.get("PERS", "0 1 2 3 4 6").split():
        os.environ["MSA_LG_ROUNDS"] = per
        ctx = _lib.Context(0)
        for _ in range(2):
            ctx.upload(a, ord("X"))
            ctx.similarity(vhash, dist)
        ctx.prof_enable(True)
        ctx.lib.msa_prof_reset(ctx.h)
        for _ in range(4):
            ctx.upload(a, ord("X"))
            mdk, q = ctx.similarity(vhash, dist)
        ms, k = ctx.prof_get("sim")
        rec["sim_ms_rounds_" + per] = round(ms / k, 3)
        bits = q.view(np.uint32).copy()
        if q0 is None:
            q0 = bits
        rec["q_equal_" + per] = bool(np.array_equal(bits, q0))
        ctx.close()
    print(json.dumps(rec), flush=True)
