"""Four threads, each with its own contexts, trimming random alignments against the oracle at the same time (what
ThreadPool.map(trimmer.trim, alignments) does to the library): python tests/fuzz/fuzz_threads.py [seconds=60] [threads=4]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
import oracle
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, OverlapTrimmer, RepresentativeTrimmer, SimilarityMatrix

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
failures, counts = [], [0] * nthreads


def worker(k):
    rng = np.random.default_rng(1000 + k)
    t0 = time.time()
    while time.time() - t0 < budget and not failures:
        m = int(rng.choice([5, 40, 130, 400, 900, 1700])) + int(rng.integers(0, 9))
        n = int(rng.choice([33, 100, 600, 2000])) + int(rng.integers(0, 7))
        keep = float(rng.choice([0.3, 0.6, 0.9]))
        flavour = int(rng.integers(0, 6))  # 0-2 protein, 3 DNA, 4 RNA, 5 nucleotides with degenerate letters
        alpha = AA if flavour < 3 else np.frombuffer({3: b"ACGT", 4: b"ACGU", 5: b"ACGTRYKMSWN"}[flavour], dtype=np.uint8)
        root = alpha[rng.integers(0, len(alpha), n)]
        a = np.where(rng.random((m, n)) < keep, root[None, :], alpha[rng.integers(0, len(alpha), (m, n))])
        a[rng.random((m, n)) < rng.beta(0.6, 1.8, n)[None, :]] = ord("-")
        if rng.random() < 0.3:
            a[(rng.random((m, n)) < 0.02) & (a != ord("-"))] = ord("X") if flavour < 3 else ord("N")
        if rng.random() < 0.15:
            low = (rng.random((m, n)) < 0.1) & (a >= 65) & (a <= 90)
            a = np.where(low, a + 32, a)
        a = np.ascontiguousarray(a, dtype=np.uint8)
        kind = int(rng.integers(0, 8))
        if kind <= 2:
            method = str(rng.choice(["automated1", "strict", "strictplus", "gappyout", "nogaps", "noallgaps", "noduplicateseqs"]))
            kw, tr = dict(method=method), AutomaticTrimmer(method, platform="hip")
        elif kind <= 4:
            kw = dict(gap_threshold=float(rng.choice([0.3, 0.6, 0.9])), similarity_threshold=float(rng.choice([0.1, 0.2, 0.6])))
            if rng.random() < 0.4 and n >= 40:
                kw[str(rng.choice(["window", "gap_window", "similarity_window"]))] = int(rng.integers(1, 5))
            if rng.random() < 0.3:
                kw["conservation_percentage"] = float(rng.choice([30, 60]))
            tr = ManualTrimmer(platform="hip", **kw)
        elif kind == 5:
            kw = dict(identity_threshold=float(rng.choice([0.3, 0.5, 0.8])))
            tr = RepresentativeTrimmer(platform="hip", **kw)
        elif kind == 6:
            kw = dict(clusters=int(rng.integers(1, m + 1)))
            tr = RepresentativeTrimmer(platform="hip", **kw)
        else:
            kw = dict(residue_overlap=float(rng.choice([0.3, 0.6, 0.9])), sequence_overlap=float(rng.choice([20, 50, 80])))
            tr = OverlapTrimmer(kw["sequence_overlap"], kw["residue_overlap"], platform="hip")
        ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
        okw, custom = dict(kw), None
        if rng.random() < 0.15:  # a custom similarity matrix over a random alphabet (letters may be missing from it)
            letters = "".join(sorted(set(rng.choice(list("ABCDEFGHIKLMNPQRSTUVWXYZ"), size=int(rng.integers(4, 25))))))
            sim = rng.integers(-4, 10, (len(letters), len(letters))).astype(np.float32)
            sim = (sim + sim.T) / 2
            custom = SimilarityMatrix(sim.tolist(), alphabet=letters)
            okw["matrix"] = oracle.make_matrix(sim, letters)
        try:
            res, seq, _ = oracle.trim(a, **okw)
            expect = None
        except oracle.OracleError as e:  # (e.g. a tiny alignment detected as nucleotides: its letters are not in that matrix)
            expect = e
        try:
            t = tr.trim(ali, custom) if custom is not None else tr.trim(ali)
            got = None
        except (ValueError, RuntimeError) as e:
            got = e
        if (expect is None) != (got is None):
            failures.append({"thread": k, "shape": [m, n], "settings": kw, "oracle": repr(expect), "device": repr(got)})
        elif expect is None and (t.residues_mask != [bool(x) for x in res] or t.sequences_mask != [bool(x) for x in seq]):
            failures.append({"thread": k, "shape": [m, n], "settings": kw})
        elif expect is None and rng.random() < 0.3:
            # TrimmedAlignment.terminal_only against the oracle: reading 2 -- the gap vector of the original alignment, the windowed
            # counts of the trim when it fetched them (checked against the oracle's own window), else the counts --
            hw = kw.get("window", kw.get("gap_window", 0))
            ogw = oracle.gaps_window(oracle.gaps(a)[0], hw)
            cached = getattr(t, "_gaps_w", None)
            if cached is not None and not np.array_equal(cached, ogw):
                failures.append({"thread": k, "shape": [m, n], "settings": kw, "cached_gap_vector": True})
            # (a result without gap statistics -- a sequence trimmer's whose trim fetched no counts -- counts over the sequences
            # it holds: reading 0)
            shared = cached is not None or getattr(t, "_gap_stats", False)
            want = oracle.terminal_only(a, res, seq, reading=2 if shared else 0, gaps_w=ogw if cached is not None else None)
            try:
                have = t.terminal_only().residues_mask
            except RuntimeError:
                have = None
            if (want is None) != (have is None) or (want is not None and have != [bool(x) for x in want]):
                failures.append({"thread": k, "shape": [m, n], "settings": kw, "terminal_only": True})
        if expect is None and got is None and not failures and rng.random() < 0.25 and res.any() and seq.any():
            # a second trim of the trimmed alignment: the reference materialises the kept part first (_trimal.pyx:1324-1327)
            sub = np.ascontiguousarray(a[np.flatnonzero(seq)][:, np.flatnonzero(res)])
            kw2 = dict(method=str(rng.choice(["gappyout", "strict", "automated1", "noallgaps"])))
            tr2 = AutomaticTrimmer(kw2["method"], platform="hip")
            try:
                res2, seq2, _ = oracle.trim(sub, **kw2)
                expect2 = None
            except oracle.OracleError as e:
                expect2 = e
            try:
                t2 = tr2.trim(t)
                got2 = None
            except (ValueError, RuntimeError) as e:
                got2 = e
            if (expect2 is None) != (got2 is None) or (expect2 is None and (t2.residues_mask != [bool(x) for x in res2] or
                                                                           t2.sequences_mask != [bool(x) for x in seq2])):
                failures.append({"thread": k, "shape": list(sub.shape), "settings": [kw, kw2], "chained": True,
                                 "oracle": repr(expect2), "device": repr(got2)})
        counts[k] += 1


ts = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(json.dumps({"mismatch": bool(failures), "failures": failures[:3], "trims_per_thread": counts, "threads": nthreads, "seconds": budget}))
sys.exit(1 if failures else 0)
