"""One small alignment trimmed again and again through the C ABI (upload + msa_trim): the command a profiler is pointed at.
   python tools/small_one.py [m n [method [count]]]      e.g.  rocprofv3 --kernel-trace --stats -d out -- python3 tools/small_one.py 46 1181 strict 300
   method: an AutomaticTrimmer method, or overlap / representative (OverlapTrimmer(80, 0.8), RepresentativeTrimmer(identity_threshold=0.75))"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer, _lib
from pytrimal_amd.synth import synth_msa

method = sys.argv[3] if len(sys.argv) > 3 else "strict"
count = int(sys.argv[4]) if len(sys.argv) > 4 else 300
if len(sys.argv) > 2 and not sys.argv[1].isdigit():  # a FASTA file instead of a shape:  small_one.py tests/golden/data/X.fasta - strictplus
    ali = Alignment.load(sys.argv[1], "fasta")
    m, n = len(ali.sequences), len(ali.sequences[0])
else:
    m, n = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (46, 1181)
    a = synth_msa(m, n, 77 + m)
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
if method == "overlap":          # OverlapTrimmer(80, 0.8): front kernel + two launches behind it
    from pytrimal_amd import OverlapTrimmer
    tr = OverlapTrimmer(80.0, 0.8, platform="hip")
elif method == "representative":  # RepresentativeTrimmer(identity_threshold=0.75)
    from pytrimal_amd import RepresentativeTrimmer
    tr = RepresentativeTrimmer(identity_threshold=0.75, platform="hip")
else:
    tr = AutomaticTrimmer(method, platform="hip")
for _ in range(5):
    tr.trim(ali)
names, dense, indet, params, keep = tr._prepare(ali)
ctx = _lib.thread_context()
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(count):
        ctx.upload(dense, indet)
        ctx.trim(params)
    best = min(best, (time.perf_counter() - t) / count)
ctx.prof_enable(True)
ctx.lib.msa_prof_reset(ctx.h)
for _ in range(20):
    ctx.upload(dense, indet)
    ctx.trim(params)
kern = {}
for k in ("front", "gaps", "prep", "pairs", "idstats", "encode", "sim", "overlap", "cluster"):
    ms, cnt = ctx.prof_get(k)
    if cnt:
        kern[k] = round(ms / cnt, 4)
ctx.prof_enable(False)
print(f"{m} x {n} {method}: {best * 1e3:.4f} ms per upload + msa_trim", kern, flush=True)
ctx.close()  # (not at interpreter exit: under a profiler the runtime may be gone by then)
