import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from pytrimal_amd import _lib, Alignment, RepresentativeTrimmer, AutomaticTrimmer
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

def phases(name, a, trimmer):
    t0 = time.perf_counter(); ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]); t1 = time.perf_counter()
    ty = ali._alignment_type(); t2 = time.perf_counter()
    ctx = _lib.thread_context()
    ctx.upload(a, ord("X")); ctx.upload(a, ord("X")); t3 = time.perf_counter()
    ctx.upload(a, ord("X")); t4 = time.perf_counter()
    mx = SimilarityMatrix.aa(); t5 = time.perf_counter()
    params = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, None, None, 0)
    trimmer._configure(params)
    vh = np.ascontiguousarray(mx._vhash); d = np.ascontiguousarray(mx._dist)
    params.vhash = vh.ctypes.data; params.dist = d.ctypes.data; params.npos = 20
    ctx.trim(params); ctx.upload(a, ord("X")); t6 = time.perf_counter()
    r = ctx.trim(params); t7 = time.perf_counter()
    out = trimmer.trim(ali); t8 = time.perf_counter()
    out = trimmer.trim(ali); t9 = time.perf_counter()
    print(f"{name}: Alignment() {t1-t0:.4f}  type {t2-t1:.4f}  upload {t4-t3:.4f}  SimilarityMatrix.aa {t5-t4:.4f}  msa_trim {t7-t6:.4f}  trimmer.trim total {t9-t8:.4f}")

phases("C3", synth_msa(2000, 10000, 1003), AutomaticTrimmer("automated1", platform="hip"))
phases("C4", synth_msa(5000, 5000, 1004), RepresentativeTrimmer(identity_threshold=0.5, platform="hip"))
phases("C5x1", synth_msa(1000, 4000, 2000), AutomaticTrimmer("automated1", platform="hip"))
