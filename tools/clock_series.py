import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
os.environ["MSA_SIM_MODE"] = "64"
import numpy as np, torch
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa
mx = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32); dist = np.ascontiguousarray(mx._dist, dtype=np.float32)
a = synth_msa(2000, 10000, 1003)
lib = _lib.load()
lib.msa_debug_bx_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
ctx = _lib.Context(0)
ctx.upload(a, ord("X")); ctx.similarity(vhash, dist)
time.sleep(1.0)
buf = (ctypes.c_ulonglong * 16)()
lib.msa_debug_bx_stamps(buf, 1)
ctx.prof_enable(2)
out = []
for i in range(24):
    ctx.prof_reset()
    ctx.upload(a, ord("X")); ctx.similarity(vhash, dist)
    lib.msa_debug_bx_stamps(buf, 1)
    ms = ctx.prof_get("sim")[0]
    clk = (buf[0] + buf[1] + buf[2]) / max(buf[6], 1) / 10
    out.append((round(ms, 3), round(clk, 3)))
print(out)
