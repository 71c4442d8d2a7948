"""Interpreter-side cost of trim_batch on sixteen 1000 x 4000 alignments: cProfile by own time (single thread), and the
batch's wall time with the cyclic garbage collector on / off (its passes run under the interpreter lock and walk every
container the process holds -- the alignments' thousands of names and sequences included)."""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.batch import trim_batch
from pytrimal_amd.synth import synth_msa
alis = []
for k in range(32):
    a = synth_msa(1000, 4000, 2000 + k)
    alis.append(Alignment([b"s%d" % i for i in range(1000)], [bytes(r) for r in a]))
tr = AutomaticTrimmer("automated1", platform="hip")
for threads in (1, 4):
    trim_batch(tr, alis, threads=threads)
    for label in ("gc on", "gc off", "gc on", "gc off"):
        if label == "gc off":
            gc.disable()
        t = time.perf_counter()
        for _ in range(4):
            trim_batch(tr, alis, threads=threads)
        dt = (time.perf_counter() - t) / 4
        gc.enable()
        print(f"threads {threads} {label}: {dt * 1e3:.2f} ms per batch of {len(alis)}, {len(alis) * 4000 / dt / 1e6:.2f} M columns/s", flush=True)
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): trim_batch(tr, alis, threads=1)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(10)
