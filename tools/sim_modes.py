import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    from pytrimal_amd import _lib
    from pytrimal_amd.synth import synth_msa
    from pytrimal_amd.matrix import SimilarityMatrix
    m, n = 2000, 10000
    a = synth_msa(m, n, 1003)
    ctx = _lib.Context(0)
    ctx.upload(a, ord("X"))
    mx = SimilarityMatrix.aa()
    ctx.similarity(mx._vhash, mx._dist)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(3): ctx.similarity(mx._vhash, mx._dist)
    ms, k = ctx.prof_get("sim")
    print("mode", os.environ.get("MSA_SIM_MODE", "0"), "kernel", os.environ.get("MSA_SIM_KERNEL", "default"), "sim ms", round(ms / k, 3),
          " ".join("%s %.3f" % (nm, ctx.prof_get(nm)[0] / max(1, ctx.prof_get(nm)[1])) for nm in ("simnum", "simden", "encode")))
    import ctypes
    buf = (ctypes.c_uint64 * 64)()
    ctx.lib.msa_debug_sim_stamps(buf)
    if buf[57]:
        print(" den wave 0: %.2f ticks/step over %d steps, %.0f ticks total" % (buf[56] / buf[57], buf[57], buf[56]))
        dt = (ctypes.c_uint64 * 1024)()
        ctx.lib.msa_debug_den_ticks(dt)
        v = np.array(dt[: (n + 31) // 32], dtype=np.float64) / 1e6
        print(" den waves Mticks: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
        print(" per WG (4 waves) max:", " ".join("%.0f" % x for x in v[: len(v) // 4 * 4].reshape(-1, 4).max(axis=1)[:80]))
    if int(os.environ.get("MSA_SIM_MODE", "0")) & 64:
        import ctypes
        buf = (ctypes.c_uint64 * 64)()
        ctx.lib.msa_debug_sim_stamps(buf)
        r = buf[2]
        print(" consumer: work %.0f  barrier %.0f cycles/round (rounds %d)" % (buf[0] / r, buf[1] / r, r))
        for p in range(1, 8):
            v = [buf[p * 8 + k] / r for k in range(5)]
            if os.environ.get("MSA_SIM_KERNEL", "") == "pc":
                print(" producer %d: settle %.0f fetch %.0f produce %.0f refresh %.0f barrier %.0f" % (p - 1, *v))
            else:
                print(" producer %d: gather %.0f emit+refresh %.0f barrier %.0f" % (p - 1, *v[:3]))
else:
    for mode in (0, 64):
        env = dict(os.environ, MSA_SIM_MODE=str(mode))
        subprocess.run([sys.executable, __file__, "x"], env=env)
