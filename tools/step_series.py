"""The first forty C3 trims of a process one by one (ms): the ramp after idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa
mx = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32); dist = np.ascontiguousarray(mx._dist, dtype=np.float32)
m, n = 2000, 10000
P = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data, len(mx))
P.method = _lib.METHOD_CODES["automated1"]
a = synth_msa(m, n, 1003); ld = (n + 63) // 64 * 64
dev = torch.zeros((m, ld), dtype=torch.uint8, device="cuda:0"); dev[:, :n] = torch.from_numpy(a).to("cuda:0"); torch.cuda.synchronize()
ctx = _lib.Context(0)
ts = []
for i in range(40):
    t = time.perf_counter(); ctx.attach(dev.data_ptr(), m, n, ld, ord("X")); ctx.trim(P); ts.append(round((time.perf_counter() - t) * 1e3, 3))
print(ts)
