"""The first forty C3 trims of a process one by one: ms per trim and the similarity kernel's own time (HIP events) --
the ramp after idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
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
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))  # idle before the first trim
ctx = _lib.Context(0)
ctx.attach(dev.data_ptr(), m, n, ld, ord("X")); ctx.trim(P)  # (allocations)
if len(sys.argv) > 2:  # seconds of unrelated dense products right in front of the series
    kind = sys.argv[3] if len(sys.argv) > 3 else "matmul"
    spin = torch.randn((4096, 4096), device="cuda:0")
    big_a = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0") if kind == "copy" else None  # 1 GiB
    big_b = torch.empty_like(big_a) if kind == "copy" else None
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(sys.argv[2]):
        for _ in range(8):
            if kind == "copy":
                big_b.copy_(big_a)
            else:
                spin = torch.nn.functional.normalize(spin @ spin, dim=1)
        torch.cuda.synchronize()
ctx.prof_enable(2)
ts, sims, pairs = [], [], []
for i in range(40):
    ctx.prof_reset()
    t = time.perf_counter(); ctx.attach(dev.data_ptr(), m, n, ld, ord("X")); ctx.trim(P); ts.append(round((time.perf_counter() - t) * 1e3, 3))
    sims.append(round(ctx.prof_get("sim")[0], 3)); pairs.append(round(ctx.prof_get("pairs")[0], 3))
print("trim ms", ts)
print("sim ms ", sims)
print("pairs  ", pairs)
