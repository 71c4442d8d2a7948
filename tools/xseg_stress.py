"""The XCD-per-segment kernel under uneven load: one thread runs its similarity pass again and again (forced at a size that takes a few
milliseconds, every result compared bit for bit with the first), other threads keep the chip busy with other contexts' work (strict trims
with many columns, RepresentativeTrimmer's pair pass, a second tall pass that has to take the barrier scheme while the first holds the
kernel).  Prints the passes, the mismatches and the passes that took longer than 150 ms (a pass that gave up and was redone).
   python tools/xseg_stress.py [seconds]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MSA_DIAGNOSTICS"] = "1"
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
mat = SimilarityMatrix.aa()
vhash, dist = mat._device_arrays()
vh = np.ascontiguousarray(mat._vhash, dtype=np.int32)
dm = np.ascontiguousarray(mat._dist, dtype=np.float32)


def params(method=None, max_identity=-1.0):
    P = _lib.TrimParams(_lib.METHOD_CODES[method] if method else 0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vh.ctypes.data, dm.ctypes.data, len(mat))
    if max_identity >= 0:
        P.max_identity = max_identity
    return P


stop = time.time() + seconds
rec = {"xseg_passes": 0, "xseg_mismatches": 0, "xseg_slow_passes": 0, "second_tall_passes": 0, "second_tall_mismatches": 0, "second_tall_kernels": {},
       "strict_trims": 0, "strict_mismatches": 0, "representative_trims": 0}


def tall(forced, key, shape, seed):
    if forced:
        os.environ["MSA_LG_XSEG"] = "2"
    ctx = _lib.Context(0)
    os.environ.pop("MSA_LG_XSEG", None)
    a = synth_msa(*shape, seed)
    ctx.upload(a, ord("X"))
    ref = ctx.similarity(vhash, dist)[1].view(np.uint32).copy()
    while time.time() < stop:
        t = time.perf_counter()
        ctx.upload(a, ord("X"))
        q = ctx.similarity(vhash, dist)[1].view(np.uint32)
        dt = time.perf_counter() - t
        rec[key + "_passes"] += 1
        rec[key + "_mismatches"] += int(not np.array_equal(q, ref))
        if key == "xseg":
            rec["xseg_slow_passes"] += int(dt > 0.15)
        else:
            k = ctx.last_paths()["sim_kernel"]
            rec["second_tall_kernels"][k] = rec["second_tall_kernels"].get(k, 0) + 1
    ctx.close()


def strict():
    ctx = _lib.Context(0)
    a = synth_msa(2000, 3000, 5)
    P = params("strict")
    ctx.upload(a, ord("X"))
    ref = ctx.trim(P)[0].copy()
    while time.time() < stop:
        ctx.upload(a, ord("X"))
        res = ctx.trim(P)[0]
        rec["strict_trims"] += 1
        rec["strict_mismatches"] += int(not np.array_equal(res, ref))
    ctx.close()


def representative():
    ctx = _lib.Context(0)
    a = synth_msa(3000, 2000, 6)
    P = params(None, 0.5)
    while time.time() < stop:
        ctx.upload(a, ord("X"))
        ctx.trim(P)
        rec["representative_trims"] += 1
    ctx.close()


threads = [threading.Thread(target=tall, args=(True, "xseg", (6000, 300), 1)), threading.Thread(target=tall, args=(False, "second_tall", (10000, 120), 2)),
           threading.Thread(target=strict), threading.Thread(target=representative)]
for th in threads:
    th.start()
for th in threads:
    th.join()
rec["seconds"] = seconds
print(json.dumps(rec), flush=True)
