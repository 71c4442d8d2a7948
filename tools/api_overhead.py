"""Where the interpreter-side time of a public-API trim goes: cProfile over repeated `trimmer.trim(alignment)` calls at a
size where the device work is small (500 x 2000), beside the same trims through the C ABI alone.
   python tools/api_overhead.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, _lib
from pytrimal_amd.synth import synth_msa

m, n = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (500, 2000)
a = synth_msa(m, n, 1002)
ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
tr = ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform="hip")
for _ in range(5):
    tr.trim(ali)
N = 200
names, dense, indet, params, keep = tr._prepare(ali)
ctx = _lib.thread_context()
for rep in range(3):  # (alternating: the order must not matter)
    t = time.perf_counter()
    for _ in range(N):
        tr.trim(ali)
    api = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for _ in range(N):
        ctx.upload(dense, indet)
        ctx.trim(params)
    abi = (time.perf_counter() - t) / N
    print(f"{m} x {n}: public API {api * 1e3:.3f} ms per trim, C ABI (upload + msa_trim through ctypes) {abi * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.trim(ali)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
