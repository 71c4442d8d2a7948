"""Counters of the pipelined split-column kernel (similarity_lg_pipe_body) for one pass at default settings, and -- with the stamped
barrier kernel (MSA_SIM_MODE=64, the pipe off) -- the ordered rows of the same pass for comparison:
   python tools/pipe_stats.py m n seed [m n seed ...]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

vhash, dist = SimilarityMatrix.aa()._device_arrays()
lib = _lib.load()
lib.msa_debug_bx_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
args = [int(x) for x in sys.argv[1:]]
for m, n, seed in [tuple(args[i:i + 3]) for i in range(0, len(args), 3)]:
    a = synth_msa(m, n, seed)
    rec = {"m": m, "n": n}
    for name, env in (("pipe", {"MSA_LG_PIPE": "2"}), ("barrier_stamped", {"MSA_LG_PIPE": "0", "MSA_SIM_MODE": "64"})):
        os.environ.update(env)
        ctx = _lib.Context(0)
        for k in env:
            os.environ.pop(k)
        buf = (ctypes.c_ulonglong * 16)()
        ctx.upload(a, ord("X")); ctx.similarity(vhash, dist)
        lib.msa_debug_bx_stamps(buf, 1)
        ctx.prof_enable(True)
        ctx.upload(a, ord("X")); ctx.similarity(vhash, dist)
        lib.msa_debug_bx_stamps(buf, 1)
        ms, k = ctx.prof_get("sim")
        P = ctx.last_paths()
        ctx.close()
        w = max(buf[3], 1)
        rec[name] = {"sim_ms": round(ms / max(k, 1), 3), "column_launches": int(buf[3]), "rounds": int(buf[4]), "ordered_rows": int(buf[10]),
                     "ordered_rows_per_round": round(buf[10] / max(buf[4], 1), 4), "waves_per_column": int(P["sim_waves_per_column"])}
        if name == "pipe":
            rec[name].update({"service_polls_for_deposits_per_round": round(buf[12] / max(buf[4], 1), 1),
                              "loop_wave_polls_for_E_per_round": round(buf[13] / max(buf[4], 1), 2), "loop_wave_polls_for_table_per_round": round(buf[14] / max(buf[4], 1), 2)})
    print(json.dumps(rec), flush=True)
