"""Denominator kernel: cycles per pair step against the row count (diagnostics)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import numpy as np
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa
from pytrimal_amd.matrix import SimilarityMatrix
mx = SimilarityMatrix.aa()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for m in (500, 1000, 1500, 2000, 2016, 2100, 2500, 3000, 3583, 4096):
    a = synth_msa(m, n, 5)
    ctx = _lib.Context(0)
    ctx.upload(a, ord("X"))
    ctx.similarity(mx._vhash, mx._dist)
    ctx.prof_enable(True); ctx.prof_reset()
    ctx.similarity(mx._vhash, mx._dist)
    buf = (ctypes.c_uint64 * 64)()
    ctx.lib.msa_debug_sim_stamps(buf)
    den_ms = ctx.prof_get("simden")[0]; num_ms = ctx.prof_get("simnum")[0]
    steps = m * (m - 1) / 2
    print("m %5d n %5d  den %.2f ms (wave 0: %.2f ticks/step)  num %.2f ms  -> %.2f / %.2f ns per pair step" % (
        m, n, den_ms, buf[56] / max(1, buf[57]), num_ms, den_ms * 1e6 / steps, num_ms * 1e6 / steps), flush=True)
    ctx.close()
