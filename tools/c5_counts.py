import json, os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.batch import trim_batch
from pytrimal_amd.synth import synth_msa
alis = []
for k in range(64):
    a = synth_msa(1000, 4000, 2000 + k)
    alis.append(Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]))
tr = AutomaticTrimmer("automated1", platform="hip")
THREADS = int(os.environ.get("THREADS", "4"))
trim_batch(tr, alis, threads=THREADS)
for count in (64, 32, 16, 8, 4, 2, 1):
    sub = alis[:count]
    ts = []
    for _ in range(7):
        t = time.perf_counter(); trim_batch(tr, sub, threads=THREADS); ts.append(time.perf_counter() - t)
    print(json.dumps({"alignments": count, "ms_best": round(min(ts)*1e3, 2), "ms_median": round(sorted(ts)[3]*1e3, 2), "ms_per_alignment": round(min(ts)*1e3/count, 3), "threads": THREADS}), flush=True)
