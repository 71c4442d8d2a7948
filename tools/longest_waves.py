import ctypes, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np, torch
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa
from bx_stamps import stamped_similarity
m, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = synth_msa(m, n, seed)
_, _, rec = stamped_similarity(a)
lib = _lib.load()
nw = rec["waves"]
buf = (ctypes.c_uint * (8 * nw))()
lib.msa_debug_bx_records.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.msa_debug_bx_records(buf, nw)
r = np.frombuffer(buf, dtype=np.uint32).reshape(nw, 8)
tot = (r[:, 1] + r[:, 2] + r[:, 3]) * 64
order = np.argsort(-tot.astype(np.int64))
print(rec)
print("ordered rows: mean %.1f  p50 %d  p90 %d  max %d" % (r[:, 6].mean(), np.percentile(r[:, 6], 50), np.percentile(r[:, 6], 90), r[:, 6].max()))
for i in order[:8]:
    c = int(r[i, 0]); col = a[:, c]
    print("col %5d: %7d cycles (pro %d loop %d stitch %d) ordered %d  valid rows %d  distinct residues %d" % (c, tot[i], r[i,1]*64, r[i,2]*64, r[i,3]*64, r[i,6], (col != ord('-')).sum(), len(set(col.tolist()))))
