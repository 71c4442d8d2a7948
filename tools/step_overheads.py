"""ms per step of the bench workloads under alternating settings of one diagnostic switch, same process, same box
(A B A B): python tools/step_overheads.py [C3 C2 C4 C5x1] [--switch MSA_PIPELINE=1,0].  Without --switch: with and
without the per-kernel event pairs (msa_prof_enable) -- what the measurement itself costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

W = {"C3": (2000, 10000, 1003, "automated1"), "C2": (500, 2000, 1002, None), "C4": (5000, 5000, 1004, None), "C5": (1000, 4000, 2000, "automated1")}
mx = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mx._dist, dtype=np.float32)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
switch = None
for i, a in enumerate(sys.argv):
    if a == "--switch":
        switch = sys.argv[i + 1]
        args.remove(switch)
for name in (args or ["C3", "C2", "C4", "C5"]):
    m, n, seed, method = W[name]
    P = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data, len(mx))
    if method:
        P.method = _lib.METHOD_CODES[method]
    elif name == "C2":
        P.gap_threshold = float(np.float32(1) - np.float32(0.5))
        P.similarity_threshold = 0.5
    else:
        P.max_identity = 0.5
    a = synth_msa(m, n, seed)
    ld = (n + 63) // 64 * 64
    dev = torch.zeros((m, ld), dtype=torch.uint8, device="cuda:0")
    dev[:, :n] = torch.from_numpy(a).to("cuda:0")
    torch.cuda.synchronize()
    def timed(ctx):
        for _ in range(3):
            ctx.attach(dev.data_ptr(), m, n, ld, ord("X"))
            ctx.trim(P)
        t = time.perf_counter()
        for _ in range(30):
            ctx.attach(dev.data_ptr(), m, n, ld, ord("X"))
            ctx.trim(P)
        return round((time.perf_counter() - t) / 30 * 1e3, 4)

    out = {}
    if switch:
        var, values = switch.split("=")
        for value in values.split(",") * 3:
            os.environ[var] = value
            ctx = _lib.Context(0)  # (the switches are read when a context is created)
            out.setdefault(value, []).append(timed(ctx))
            ctx.close()
        print(name, "ms/step", {f"{var}={k}": v for k, v in out.items()}, flush=True)
        continue
    ctx = _lib.Context(0)
    for prof in (False, True, False, True):
        ctx.prof_enable(prof)
        out.setdefault(prof, []).append(timed(ctx))
    print(name, "ms/step without events", out[False], "with events", out[True], flush=True)
    ctx.close()
