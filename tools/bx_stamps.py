"""Cycle stamps of the similarity kernel (MSA_SIM_MODE=64): per wave averages of the ordered prologue, the round loops
and the stitching, ordered rows per wave, the shader clock under the kernel.  The pass runs as ONE launch here (MSA_LG_ROUNDS=0
unless the variable is set): with a launch every six rounds a "wave" would be a column's share of one launch.
   python tools/bx_stamps.py [m n seed]      (default: the C3 alignment)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa


def stamped_similarity(a, indet=ord("X"), matrix=None):
    """-> (mdk, q, record) of one stamped launch (a context created under MSA_SIM_MODE=64)"""
    mat = matrix or SimilarityMatrix.aa()
    vhash, dist = mat._device_arrays()
    lib = _lib.load()
    lib.msa_debug_bx_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    os.environ["MSA_SIM_MODE"] = "64"
    rounds = os.environ.get("MSA_LG_ROUNDS")
    os.environ["MSA_LG_ROUNDS"] = rounds if rounds is not None else "0"  # (one launch: the stamps are per wave = per column then)
    ctx = _lib.Context(0)
    os.environ.pop("MSA_SIM_MODE")
    if rounds is None:
        os.environ.pop("MSA_LG_ROUNDS")
    try:
        ctx.upload(a, indet)
        ctx.similarity(vhash, dist)
        buf = (ctypes.c_ulonglong * 16)()
        lib.msa_debug_bx_stamps(buf, 1)
        ctx.prof_enable(True)
        ctx.upload(a, indet)
        mdk, q = ctx.similarity(vhash, dist)
        lib.msa_debug_bx_stamps(buf, 1)
        ms, k = ctx.prof_get("sim")
    finally:
        ctx.close()
    w = max(buf[3], 1)
    return mdk, q, {"sim_ms": round(ms / max(k, 1), 3), "waves": int(buf[3]), "ordered_rows_per_wave": round(buf[10] / w, 1), "of_which_sparse_rows": round(buf[9] / w, 1), "of_which_same_row_in_both_chains": round(2 * buf[12] / w, 1),
                    "rounds_per_wave": round(buf[4] / w, 1), "max_rounds": int(buf[8]), "wave_ms_avg": round(buf[6] / w / 1e5, 3),
                    "longest_wave_kcycles": round(buf[7] / 1e3, 1), "clock_GHz": round((buf[0] + buf[1] + buf[2]) / max(buf[6], 1) / 10, 3),
                    "kcycles_per_wave": {"prologue": round(buf[0] / w / 1e3, 1), "loops": round(buf[1] / w / 1e3, 1),
                                         "stitch": round(buf[2] / w / 1e3, 1), "of_which_ordered_rows": round(buf[11] / w / 1e3, 1)},
                    "prologue_parts_kcycles": {"histogram": round(buf[13] / w / 1e3, 1), "G": round(buf[14] / w / 1e3, 1), "first_rows": round(buf[5] / w / 1e3, 1)}}


if __name__ == "__main__":
    m, n, seed = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2000, 10000, 1003)))
    _, _, rec = stamped_similarity(synth_msa(m, n, seed))
    print(json.dumps({"m": m, "n": n, **rec}), flush=True)
