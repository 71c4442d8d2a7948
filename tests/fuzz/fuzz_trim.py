"""Randomised whole-trim check: random shapes, compositions, alphabets and trimmer settings through msa_trim (contexts under
the default switches, the serial flow, the side stream at any size, row-index lists, the sequential similarity kernel, host
exponentials + packed uploads) against
the oracle's trim -- masks, selected method, identity means, cut points.
  python tests/fuzz/fuzz_trim.py [seconds=120] [seed=1] [tall]     (prints one JSON line; exit code 1 on the first mismatch)
`tall` (round 6): 700 ... 2300 sequences x 16 ... 96 columns, dense around the dispatch thresholds at 1024 / 1025 (the compact
pipeline hands over to the ordinary one) and 1799 / 1800 (six rounds of the similarity kernel per launch) and around 512 / 513 (the
narrow front kernel, the sixteen-row pair tiles): the paths the BASELINE's C3 and C5 take, on random data, every trimmer kind."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
import oracle
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
TALL = len(sys.argv) > 3 and sys.argv[3] == "tall"
rng = np.random.default_rng(seed)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
EXTRA = np.frombuffer(b"BZJUO", dtype=np.uint8)  # in no default matrix: a strict trim must raise where the oracle does

# (the default context takes small alignments through the compact pipeline with the flat similarity kernel up to 128 sequences;
# the others: the flat kernel at every size it takes, the compact pipeline with the wave-per-column kernel, the ordinary launch
# sequence with the rows copied to the device, and the switches of the wave-per-column kernel)
CONTEXTS = [dict(), dict(MSA_FLAT_MAX_M="512"), dict(MSA_FLAT_MAX_M="0", MSA_MDK_HOST="1"), dict(MSA_COMPACT="0", MSA_ZEROCOPY_KB="0"),
            dict(MSA_PIPELINE="0"), dict(MSA_PIPELINE="3"), dict(MSA_LG_BIG="1"), dict(MSA_SIM_KERNEL="seq"),
            dict(MSA_MDK_HOST="1", MSA_UPLOAD_DIRECT="0"), dict(MSA_LG_ROUNDS="1"), dict(MSA_LG_SPLIT="3"),
            dict(MSA_LG_SPLIT="8", MSA_LG_ROUNDS="2"),
            # (round 6) the front kernel and the pair tiles of 513 ... 1024 sequences: round 5's kernels, and the new ones in other shapes
            dict(MSA_FRONT_CW="64", MSA_PAIR_TI="8"), dict(MSA_FRONT_CW="32", MSA_FRONT_NT="256", MSA_PAIR_K="2"),
            dict(MSA_FRONT_CW="16", MSA_FRONT_NT="512", MSA_FRONT_XCD="0", MSA_PAIR_TI="16", MSA_PAIR_K="8", MSA_FRONT_FROM_M="130"),
            dict(MSA_LISTS_FUSED="0", MSA_COMPACT="0"), dict(MSA_LG_HALVES="2", MSA_LG_ROUNDS="2", MSA_COMPACT="0"),
            dict(MSA_LG_HALVES="0"),
            # (round 6, late) a forced split runs as loop waves + a service wave up to twelve; the barrier scheme, and the pipelined kernel wherever a column is split
            dict(MSA_LG_SPLIT="5", MSA_LG_PIPE="0"), dict(MSA_LG_PIPE="0"), dict(MSA_LG_SPLIT="12", MSA_LG_ROUNDS="1"),
            dict(MSA_LG_SPLIT="4", MSA_LG_PIPE_K="4"),
            # the XCD-per-segment kernel of tall alignments at any size, and its pass giving up at once
            dict(MSA_COMPACT="0", MSA_LG_XSEG="2"), dict(MSA_COMPACT="0", MSA_LG_XSEG="3", MSA_LG_ROUNDS="3")]
SWITCHES = ("MSA_PIPELINE", "MSA_LG_BIG", "MSA_SIM_KERNEL", "MSA_MDK_HOST", "MSA_UPLOAD_DIRECT", "MSA_LG_ROUNDS", "MSA_LG_SPLIT", "MSA_COMPACT",
            "MSA_FLAT_MAX_M", "MSA_ZEROCOPY_KB", "MSA_FRONT_CW", "MSA_FRONT_NT", "MSA_FRONT_XCD", "MSA_FRONT_FROM_M", "MSA_PAIR_TI", "MSA_PAIR_K", "MSA_LISTS_FUSED", "MSA_LG_HALVES", "MSA_LG_PIPE", "MSA_LG_PIPE_K", "MSA_LG_XSEG", "MSA_LG_XSEG_KX")
ctxs = []
for env in CONTEXTS:
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    ctxs.append(_lib.Context(0))
for k in SWITCHES:
    os.environ.pop(k, None)

mx = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mx._dist, dtype=np.float32)


def alignment():
    m = int(rng.choice([2, 3, 7, 21, 40, 64, 65, 130, 300, 700])) + int(rng.integers(0, 9))
    n = int(rng.choice([1, 5, 31, 33, 64, 100, 257, 600, 1500])) + int(rng.integers(0, 7))
    if m <= 140 and rng.random() < 0.03:
        n = 5121 + int(rng.integers(0, 64))  # (more columns than the chip has wave slots: the compact pipeline sorts them)
    if TALL:
        r = rng.random()
        if r < 0.2:
            m = int(rng.choice([1023, 1024, 1025, 1026]))
        elif r < 0.4:
            m = int(rng.choice([1798, 1799, 1800, 1801]))
        elif r < 0.5:
            m = int(rng.choice([511, 512, 513, 514]))
        else:
            m = int(rng.integers(700, 2301))
        n = int(rng.integers(16, 97))
    keep = float(rng.choice([0.2, 0.45, 0.6, 0.7, 0.85, 0.97]))
    root = AA[rng.integers(0, 20, n)]
    a = np.where(rng.random((m, n)) < keep, root[None, :], AA[rng.integers(0, 20, (m, n))])
    style = rng.integers(0, 4)
    if style == 0:
        g = rng.beta(0.6, 1.8, n)
    elif style == 1:
        g = np.full(n, 0.02)
    elif style == 2:
        g = np.where(rng.random(n) < 0.3, 0.9, 0.05)
    else:
        g = rng.random(n)
    a[rng.random((m, n)) < g[None, :]] = ord("-")
    a[(rng.random((m, n)) < 0.01) & (a != ord("-"))] = ord("X")
    if rng.random() < 0.15:  # whole rows / columns of gaps
        a[rng.integers(0, m)] = ord("-")
        a[:, rng.integers(0, n)] = ord("-")
    if rng.random() < 0.1:  # lower case and rare letters
        sel = rng.random((m, n)) < 0.05
        a[sel & (a >= 65) & (a <= 90)] += 32
    if rng.random() < 0.05:
        a[rng.integers(0, m), rng.integers(0, n)] = EXTRA[rng.integers(0, len(EXTRA))]
    if rng.random() < 0.1 and m > 4:  # duplicated rows
        a[rng.integers(0, m)] = a[rng.integers(0, m)]
    return np.ascontiguousarray(a, dtype=np.uint8)


def settings(m, n):
    kind = rng.integers(0, 10)
    if kind < 4:
        return dict(method=str(rng.choice(["automated1", "strict", "strictplus", "gappyout", "nogaps", "noallgaps", "noduplicateseqs"])))
    if kind < 8:
        kw = {}
        if rng.random() < 0.7:
            kw["gap_threshold"] = float(rng.choice([0.1, 0.5, 0.8, 0.95]))
        if rng.random() < 0.7 or not kw:
            kw["similarity_threshold"] = float(rng.choice([0.05, 0.2, 0.5, 0.8]))
        if rng.random() < 0.4:
            kw["conservation_percentage"] = float(rng.choice([20, 50, 80]))
        if rng.random() < 0.3 and n >= 16:
            kw[str(rng.choice(["window", "gap_window", "similarity_window"]))] = int(rng.integers(1, max(2, min(6, n // 4))))
        return kw
    if kind == 8 and m >= 2 and TALL and rng.random() < 0.5:  # (clusters=K probes the clustering a dozen times: every other case)
        return dict(identity_threshold=float(rng.choice([0.2, 0.4, 0.6, 0.9])))
    if kind == 8 and m >= 2:
        return dict(identity_threshold=float(rng.choice([0.2, 0.4, 0.6, 0.9]))) if rng.random() < 0.6 else dict(clusters=int(rng.integers(1, m + 1)))
    return dict(residue_overlap=float(rng.choice([0.3, 0.6, 0.9])), sequence_overlap=float(rng.choice([20, 50, 80])))


def params_of(kw):
    p = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data, len(mx))
    if "method" in kw:
        p.method = _lib.METHOD_CODES[kw["method"]]
    if "gap_threshold" in kw:
        p.gap_threshold = float(np.float32(1) - np.float32(kw["gap_threshold"]))
    for name in ("similarity_threshold", "conservation_percentage", "window", "gap_window", "similarity_window", "residue_overlap",
                 "sequence_overlap", "clusters"):
        if name in kw:
            setattr(p, name, kw[name])
    if "identity_threshold" in kw:
        p.max_identity = kw["identity_threshold"]
    return p


t0 = time.time()
cases = raised = 0
while time.time() - t0 < budget:
    a = alignment()
    m, n = a.shape
    kw = settings(m, n)
    try:
        res, seq, oinfo = oracle.trim(a, indet=ord("X"), matrix=(vhash, dist), **kw)
        expect = None
    except oracle.OracleError as e:
        expect = e
    p = params_of(kw)
    for ctx, env in zip(ctxs, CONTEXTS):
        ctx.upload(a, ord("X"))
        try:
            keep_res, keep_seq, info = ctx.trim(p)
            got = None
        except Exception as e:  # noqa: BLE001
            got = e
        ok = (expect is None) == (got is None)
        if ok and expect is None:
            ok = np.array_equal(keep_res, res) and np.array_equal(keep_seq, seq)
            if ok and kw.get("method") == "automated1" and m > 1:
                ok = info.selected_method == oinfo.selected and np.float32(info.avg_seq).view(np.uint32) == np.float32(oinfo.avg_seq).view(np.uint32)
        if not ok:
            print(json.dumps({"mismatch": True, "case": cases, "seed": seed, "shape": [m, n], "settings": kw, "context": env,
                              "oracle_raised": repr(expect), "device_raised": repr(got)}))
            np.save(os.path.join(ROOT, "gpurun_out", "fuzz_failure.npy"), a) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
            sys.exit(1)
    cases += 1
    raised += expect is not None
print(json.dumps({"mismatch": False, "cases": cases, "contexts_per_case": len(ctxs), "cases_where_both_raise": int(raised),
                  "seconds": round(time.time() - t0, 1), "seed": seed, "mode": "tall" if TALL else "default"}))
