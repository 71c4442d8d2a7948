"""Round 6: the compact pipeline's front kernel and pair pass from 513 sequences on, by variant -- contexts created under
MSA_FRONT_CW / MSA_FRONT_NT / MSA_FRONT_XCD / MSA_PAIR_TI / MSA_PAIR_K, alternating inside one process on one box.
  1. every variant's trims against the round-5 kernels (MSA_FRONT_CW=64 MSA_PAIR_TI=8) and the oracle: masks, selectMethod's
     means, cut points -- random alignments of 513 ... 1024 sequences;
  2. ms per upload + msa_trim of one 1000 x 4000 `automated1` trim and its kernels by HIP events, A B C ... A B C.
   python tools/front_pairs_ab.py [check|time|both] > profiles/r06_front_pairs_ab.txt"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

SWITCHES = ("MSA_FRONT_CW", "MSA_FRONT_NT", "MSA_FRONT_XCD", "MSA_PAIR_TI", "MSA_PAIR_K", "MSA_FRONT_FROM_M")
BASE = dict(MSA_FRONT_CW="64", MSA_PAIR_TI="8")
VARIANTS = [("round 5", BASE)]
for cw in ("16", "32"):
    for nt in ("256", "512", "1024"):
        VARIANTS.append((f"front {cw} x {nt}", dict(MSA_FRONT_CW=cw, MSA_FRONT_NT=nt, MSA_PAIR_TI="8")))
VARIANTS.append(("front 16 x 1024, blocks as they lie", dict(MSA_FRONT_CW="16", MSA_FRONT_NT="1024", MSA_FRONT_XCD="0", MSA_PAIR_TI="8")))
for k in ("2", "4", "8"):
    VARIANTS.append((f"pairs 16 rows, K = {k}", dict(MSA_FRONT_CW="64", MSA_PAIR_TI="16", MSA_PAIR_K=k)))
for k in ("2", "8"):
    VARIANTS.append((f"pairs 8 rows, K = {k}", dict(MSA_FRONT_CW="64", MSA_PAIR_TI="8", MSA_PAIR_K=k)))
VARIANTS.append(("default build", dict()))
if os.environ.get("AB_SMALL"):  # the narrow front kernel below 513 sequences
    VARIANTS = [("round 5", BASE), ("front 16 x 256 from 130", dict(MSA_FRONT_CW="16", MSA_FRONT_NT="256", MSA_FRONT_FROM_M="130", MSA_PAIR_TI="8")),
                ("front 16 x 512 from 130", dict(MSA_FRONT_CW="16", MSA_FRONT_NT="512", MSA_FRONT_FROM_M="130", MSA_PAIR_TI="8")),
                ("front 32 x 512 from 130", dict(MSA_FRONT_CW="32", MSA_FRONT_NT="512", MSA_FRONT_FROM_M="130", MSA_PAIR_TI="8")),
                ("front 32 x 1024 from 130", dict(MSA_FRONT_CW="32", MSA_FRONT_NT="1024", MSA_FRONT_FROM_M="130", MSA_PAIR_TI="8")),
                ("pairs 16 rows", dict(MSA_FRONT_CW="64", MSA_PAIR_TI="16")), ("pairs 16 rows, K = 4", dict(MSA_FRONT_CW="64", MSA_PAIR_TI="16", MSA_PAIR_K="4"))]


def context(env):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    c = _lib.Context(0)
    for k in SWITCHES:
        os.environ.pop(k, None)
    return c


mx = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mx._dist, dtype=np.float32)


def params(method):
    p = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data, len(mx))
    p.method = _lib.METHOD_CODES[method]
    return p


what = sys.argv[1] if len(sys.argv) > 1 else "both"
ctxs = [(name, env, context(env)) for name, env in VARIANTS]

if what in ("check", "both"):
    import oracle
    rng = np.random.default_rng(6)
    AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    cases = 0
    for m, n in [(513, 70), (600, 333), (640, 31), (777, 129), (1000, 500), (1023, 65), (1024, 64), (1024, 257), (900, 5200)]:
        for rep in range(2):
            keep = float(rng.choice([0.3, 0.6, 0.9]))
            root = AA[rng.integers(0, 20, n)]
            a = np.where(rng.random((m, n)) < keep, root[None, :], AA[rng.integers(0, 20, (m, n))])
            g = rng.beta(0.6, 1.8, n) if rep == 0 else np.where(rng.random(n) < 0.3, 0.9, 0.05)
            a[rng.random((m, n)) < g[None, :]] = ord("-")
            a[(rng.random((m, n)) < 0.01) & (a != ord("-"))] = ord("X")
            if rep:
                a[rng.integers(0, m)] = ord("-")
                a[:, rng.integers(0, n)] = ord("-")
                sel = rng.random((m, n)) < 0.03
                a[sel & (a >= 65) & (a <= 90) & (a != ord("X"))] += 32  # (a lower-case x is no indetermination: both sides raise)
            a = np.ascontiguousarray(a, dtype=np.uint8)
            for method in ("strict", "automated1"):
                ores = None
                if n <= 600:
                    ores, oseq, oinfo = oracle.trim(a, indet=ord("X"), matrix=(vhash, dist), method=method)
                ref = None
                for name, env, c in ctxs:
                    c.upload(a, ord("X"))
                    kr, ks, info = c.trim(params(method))
                    got = (kr.tobytes(), ks.tobytes(), info.selected_method, np.float32(info.avg_seq).tobytes(), np.float32(info.max_seq).tobytes(),
                           info.gap_cut, np.float32(info.sim_cut).tobytes())
                    if ref is None:
                        ref = got
                        if ores is not None:
                            assert np.array_equal(kr, ores) and np.array_equal(ks, oseq), ("oracle", name, m, n, method)
                    assert got == ref, ("variant differs from round 5's kernels", name, m, n, method, c.last_paths())
                cases += 1
        # a bad residue must be reported the same way (first bad residue: smallest column, then row)
        b = a.copy()
        b[m // 2, n // 3] = ord("J")
        b[m // 3, n // 3] = ord("O")
        errs = []
        for name, env, c in ctxs:
            c.upload(b, ord("X"))
            try:
                c.trim(params("strict"))
                errs.append(None)
            except Exception as e:  # noqa: BLE001
                errs.append(str(e))
        assert errs[0] is not None and all(e == errs[0] for e in errs), errs
    print(json.dumps({"checked_cases": cases, "variants": len(ctxs), "all_identical": True}), flush=True)

if what in ("time", "both"):
    shapes = [(1000, 4000), (600, 2500), (1024, 8000)]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in x.split("x")) for x in os.environ["SHAPES"].split(",")]
    for m, n in shapes:
        a = synth_msa(m, n, 77 + m)
        p = params("automated1")
        res = {name: [] for name, _, _ in ctxs}
        kern = {}
        for rnd in range(3):
            for name, env, c in ctxs:
                for _ in range(5):
                    c.upload(a, ord("X"))
                    c.trim(p)
                t = time.perf_counter()
                N = 150
                for _ in range(N):
                    c.upload(a, ord("X"))
                    c.trim(p)
                res[name].append((time.perf_counter() - t) / N * 1e3)
                if rnd == 2:
                    c.prof_enable(True)
                    c.prof_reset()
                    for _ in range(30):
                        c.upload(a, ord("X"))
                        c.trim(p)
                    k = {}
                    for key in ("front", "pairs", "idstats", "sim"):
                        ms, cnt = c.prof_get(key)
                        if cnt:
                            k[key] = round(ms / cnt * 1e3, 1)
                    c.prof_enable(False)
                    kern[name] = k
        for name, env, c in ctxs:
            print(json.dumps({"shape": [m, n], "variant": name, "env": env, "trim_ms": [round(x, 4) for x in res[name]], "kernels_us": kern[name],
                              "paths": c.last_paths() if hasattr(c, "last_paths") else None}), flush=True)
for _, _, c in ctxs:
    c.close()
