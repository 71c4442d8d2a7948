"""Sweep MSA_SIM_TCOLS (columns per similarity workgroup) at the C3 size; one subprocess per value."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "x":
    import numpy as np
    import torch  # noqa: F401  (first HIP runtime mapped)
    from pytrimal_amd import _lib
    from pytrimal_amd.synth import synth_msa
    from pytrimal_amd.matrix import SimilarityMatrix
    m, n = int(sys.argv[2]), int(sys.argv[3])
    a = synth_msa(m, n, 1003)
    ctx = _lib.Context(0)
    ctx.upload(a, ord("X"))
    mx = SimilarityMatrix.aa()
    mdk, q = ctx.similarity(mx._vhash, mx._dist)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(3): ctx.similarity(mx._vhash, mx._dist)
    ms, k = ctx.prof_get("sim")
    import zlib
    print("tcols", os.environ.get("MSA_SIM_TCOLS", "auto"), "m", m, "n", n, "sim ms", round(ms / k, 3), "crc", zlib.crc32(q.tobytes()))
else:
    for (m, n) in ((2000, 10000), (1000, 4000)):
        for t in ("64", "56", "48", "40", "32", "24", "16", "8", ""):
            env = dict(os.environ)
            if t: env["MSA_SIM_TCOLS"] = t
            else: env.pop("MSA_SIM_TCOLS", None)
            subprocess.run([sys.executable, __file__, "x", str(m), str(n)], env=env)
