"""Phase cycles of the binade-exact similarity kernel (MSA_SIM_MODE=64): per wave averages of the ordered
prologue, the round loops and the stitching, for several columns-per-wave / prologue settings."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

m, n, seed = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2000, 10000, 1003)))
a = synth_msa(m, n, seed)
mat = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mat._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mat._dist, dtype=np.float32)
lib = _lib.load()
lib.msa_debug_bx_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.msa_debug_bx_records.argtypes = [ctypes.c_void_p, ctypes.c_int]
COLS = os.environ.get("BX_STAMP_COMPACT", "0").split(",")
R0S = os.environ.get("BX_STAMP_R0", "").split(",")
for cols in COLS:
    for r0 in R0S:
        os.environ.update(MSA_SIM_MODE="64", MSA_BX_COMPACT=cols, **({"MSA_BX_R0": r0} if r0 != "" else {}))
        wv = os.environ.get("MSA_BX_ASM", "")
        ctx = _lib.Context(0)
        ctx.upload(a, ord("X"))
        ctx.similarity(vhash, dist)
        buf = (ctypes.c_ulonglong * 16)()
        lib.msa_debug_bx_stamps(buf, 1)
        ctx.prof_enable(True)
        t = time.perf_counter()
        ctx.upload(a, ord("X"))
        ctx.similarity(vhash, dist)
        wall = time.perf_counter() - t
        lib.msa_debug_bx_stamps(buf, 1)
        ms, k = ctx.prof_get("sim")
        w = max(buf[3], 1)
        print(json.dumps({"asm": wv, "compact": int(cols), "r0": r0, "ordered_rows_per_wave": round(buf[10] / w, 1), "sim_ms": round(ms / max(k, 1), 3), "waves": buf[3],
                          "rounds_per_wave": round(buf[4] / w, 1), "dual_chains_per_round": round(buf[5] / max(buf[4], 1), 2),
                          "max_rounds": buf[8], "shortened_per_wave": round(buf[9] / w, 2), "wave_ms_avg": round(buf[6] / w / 1e5, 3), "longest_wave_kcycles": round(buf[7] / 1e3, 1),
                          "clock_GHz": round((buf[0] + buf[1] + buf[2]) / max(buf[6], 1) / 10, 3), "kcycles_per_wave": {
                              "prologue": round(buf[0] / w / 1e3, 1), "loops": round(buf[1] / w / 1e3, 1),
                              "stitch": round(buf[2] / w / 1e3, 1), "of_which_ordered_rows": round(buf[11] / w / 1e3, 1)}}), flush=True)
        if os.environ.get("BX_RECORDS"):
            nw = int(buf[3])
            rec = (ctypes.c_uint * (8 * nw))()
            lib.msa_debug_bx_records(rec, nw)
            r = np.frombuffer(rec, dtype=np.uint32).reshape(nw, 8).astype(np.int64)
            tot = (r[:, 1] + r[:, 2] + r[:, 3]) * 64
            order = np.argsort(-tot)
            g = (a == ord("-")).sum(axis=0)
            for i in list(order[:6]) + list(order[-3:]):
                c = int(r[i, 0])
                print("  wave", int(i), "col", c, "gaps", int(g[c]) if c < n else -1, "kcycles pro/loop/stitch",
                      int(r[i, 1]) * 64 // 1000, int(r[i, 2]) * 64 // 1000, int(r[i, 3]) * 64 // 1000, "rounds", int(r[i, 4]),
                      "short", int(r[i, 5]), "dualslots", int(r[i, 6]))
            print("  percentiles of wave kcycles:", [int(x) // 1000 for x in np.percentile(tot, [1, 25, 50, 75, 95, 99, 100])])
        ctx.close()
