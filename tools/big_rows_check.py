"""The row-count boundary of the binade-exact similarity kernel (m < 32 000: 16-bit row indices, 32-bit W offsets): at
m = 31 999 the default kernel and the chain kernels (independent implementations of one bit-exact statistic) must agree
bit for bit; at m = 32 001 only the chain kernels run -- checked to be deterministic and finite.  ~13 GB of device memory.
  python tools/big_rows_check.py [columns=8]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mx = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mx._dist, dtype=np.float32)


def run(m, kernel):
    for k in ("MSA_SIM_KERNEL",):
        os.environ.pop(k, None)
    if kernel:
        os.environ["MSA_SIM_KERNEL"] = kernel
    a = synth_msa(m, n, 4242)
    ctx = _lib.Context(0)
    t = time.perf_counter()
    ctx.upload(a, ord("X"))
    mdk, q = ctx.similarity(vhash, dist)
    g = ctx.gaps()
    dt = time.perf_counter() - t
    ctx.close()
    assert np.array_equal(g, (a == ord("-")).sum(axis=0))
    return np.asarray(q, dtype=np.float32).view(np.uint32), np.asarray(mdk, dtype=np.float32), dt


out = {}
q_lg, mdk_lg, t_lg = run(31999, "")
q_ch, mdk_ch, t_ch = run(31999, "chain")
out["m_31999"] = {"lg_s": round(t_lg, 2), "chain_s": round(t_ch, 2), "q_bits_equal": bool(np.array_equal(q_lg, q_ch)),
                  "mdk_finite": bool(np.isfinite(mdk_lg).all())}
q1, mdk1, t1 = run(32001, "")
q2, mdk2, t2 = run(32001, "")
out["m_32001"] = {"seconds": round(t1, 2), "deterministic": bool(np.array_equal(q1, q2)), "mdk_finite": bool(np.isfinite(mdk1).all()),
                  "mdk_range": [float(mdk1.min()), float(mdk1.max())]}
print(json.dumps(out))
sys.exit(0 if out["m_31999"]["q_bits_equal"] and out["m_32001"]["deterministic"] else 1)
