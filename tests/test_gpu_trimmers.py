"""The reference's trimmer tests with platform="hip": every test class of
src/pytrimal/tests/test_{automatic,manual,overlap,representative}_trimmer.py, run through the
public API of this package against the same golden files (those that survive in the reference
checkout) and, where the reference's fixture is a dangling symlink, against the oracle.
"""
import json
import pickle
import warnings

import numpy as np
import pytest

import oracle
from conftest import EXAMPLE_001, EXAMPLE_001_NAMES, GOLDEN, OVERLAP_EXAMPLE, OVERLAP_NAMES, data_path, edge_msa
from pytrimal_amd import (
    Alignment,
    AutomaticTrimmer,
    ManualTrimmer,
    OverlapTrimmer,
    RepresentativeTrimmer,
    SimilarityMatrix,
    TrimmedAlignment,
)
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa

pytestmark = pytest.mark.gpu
PLATFORM = "hip"


def load(name, fmt="fasta"):
    with open(data_path(name), "rb") as f:
        return Alignment.load(f, format=fmt)


def assert_trimmed_equal(trimmed, expected):
    assert len(trimmed.names) == len(expected.names)
    assert len(trimmed.sequences) == len(expected.sequences)
    assert trimmed.names == expected.names
    for s1, s2 in zip(trimmed.sequences, expected.sequences):
        assert s1 == s2


def assert_matches_oracle(trimmed, a, **kw):
    res, seq, _ = oracle.trim(a, **kw)
    assert trimmed.residues_mask == [bool(x) for x in res]
    assert trimmed.sequences_mask == [bool(x) for x in seq]


@pytest.fixture(scope="module")
def enog_ali():
    return load("ENOG411BWBU.seq40.res60.fasta")  # == ENOG411BWBU.fasta


# --- TestManualTrimmer ---------------------------------------------------------------------------

@pytest.mark.parametrize("gt,cons", [(0.9, 60), (0.4, 40)])
def test_gap_threshold(enog_ali, gt, cons):
    expected = load("ENOG411BWBU.cons%02d.gt%02d.fasta" % (cons, int(gt * 100)))
    trimmed = ManualTrimmer(gap_threshold=gt, conservation_percentage=cons, platform=PLATFORM).trim(enog_ali)
    assert_trimmed_equal(trimmed, expected)


def test_window():
    ali = Alignment(EXAMPLE_001_NAMES, EXAMPLE_001)
    expected = load("example.001.gt90.w3.clw", "clustal")
    trimmed = ManualTrimmer(gap_threshold=0.9, window=3, platform=PLATFORM).trim(ali)
    assert trimmed.names == expected.names
    assert list(trimmed.sequences) == list(expected.sequences)


def test_large_window():
    ali = Alignment([b"seq1", b"seq2"], ["M-KKV", "MY-KV"])
    with pytest.raises(Exception):
        ManualTrimmer(gap_threshold=0.9, window=100, platform=PLATFORM).trim(ali)


@pytest.mark.parametrize("kw", [dict(gap_threshold=0.5, similarity_threshold=0.5),
                                dict(similarity_threshold=0.3, conservation_percentage=50),
                                dict(gap_threshold=0.8, window=2),
                                dict(gap_threshold=0.7, similarity_threshold=0.2, gap_window=2, similarity_window=3),
                                dict(gap_absolute_threshold=20)])
def test_manual_against_oracle(enog_ali, kw):
    trimmed = ManualTrimmer(platform=PLATFORM, **kw).trim(enog_ali)
    a = oracle.pack(list(enog_ali.sequences))
    assert_matches_oracle(trimmed, a, **kw)


# --- TestAutomaticTrimmer ------------------------------------------------------------------------

def test_noduplicateseqs_method(enog_ali):
    trimmed = AutomaticTrimmer("noduplicateseqs", platform=PLATFORM).trim(enog_ali)
    assert_trimmed_equal(trimmed, load("ENOG411BWBU.noduplicateseqs.fasta"))


@pytest.mark.parametrize("method", ["strict", "strictplus", "gappyout", "automated1", "nogaps", "noallgaps"])
def test_automatic_methods_against_oracle(enog_ali, method):
    # the reference's golden files for these methods are dangling symlinks in its checkout
    trimmed = AutomaticTrimmer(method, platform=PLATFORM).trim(enog_ali)
    assert_matches_oracle(trimmed, oracle.pack(list(enog_ali.sequences)), method=method)


def test_automated1_docstring_example():
    ali = Alignment(EXAMPLE_001_NAMES, EXAMPLE_001)
    trimmed = AutomaticTrimmer("automated1", platform=PLATFORM).trim(ali)
    assert list(trimmed.sequences) == ["VWLFPWNGLQIHMMGII", "EWFFAWLGLEINMMVII", "AAANAWLGLEINMMAQI",
                                       "SWYLAWLGLEINMMAII", "TWFQLWQGLDLNKMPVF", "AWFQAWGGLEINKQAIL"]


def test_strictplus_readme_example():
    ali = Alignment(EXAMPLE_001_NAMES, EXAMPLE_001)
    trimmed = AutomaticTrimmer("strictplus", platform=PLATFORM).trim(ali)
    assert list(trimmed.sequences) == ["GIVLVWLFPWNGLQIHMMGII", "VIMLEWFFAWLGLEINMMVII", "GLFLAAANAWLGLEINMMAQI",
                                       "GIYLSWYLAWLGLEINMMAII", "GFLLTWFQLWQGLDLNKMPVF", "GLHMAWFQAWGGLEINKQAIL"]


def test_automated2_is_refused_by_the_c_abi(enog_ali):
    """The Python constructor refuses the method (tests/test_api.py); a caller of the C ABI gets a return code."""
    import ctypes

    ctx = _lib.Context(0)
    names, seqs = enog_ali.names, list(enog_ali.sequences)
    ctx.upload(oracle.pack(seqs), ord("X"))
    P = _lib.TrimParams(_lib.METHOD_CODES["automated2"], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, None, None, 0)
    with pytest.raises(_lib.MsaError) as err:
        ctx.trim(P)
    assert err.value.code == _lib.E_NOT_IMPLEMENTED
    ctx.close()


def test_custom_similarity_matrix(enog_ali):
    with open(data_path("pam70.json")) as f:
        pam70 = SimilarityMatrix(**json.load(f))
    trimmed = AutomaticTrimmer("strict", platform=PLATFORM).trim(enog_ali, pam70)
    a = oracle.pack(list(enog_ali.sequences))
    assert_matches_oracle(trimmed, a, method="strict", matrix=(pam70._vhash, pam70._dist))


def test_invalid_characters():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ali = Alignment([b"seq1", b"seq2"], ["MKKBO", "MKKAY"])
        with pytest.raises(ValueError):
            AutomaticTrimmer("strict", platform=PLATFORM).trim(ali)


def test_pickle_automatic():
    t = AutomaticTrimmer("automated1", platform=PLATFORM)
    p = pickle.loads(pickle.dumps(t))
    ali = Alignment(OVERLAP_NAMES, OVERLAP_EXAMPLE)
    assert_trimmed_equal(p.trim(ali), t.trim(ali))


def test_repr_on_gpu_host():
    assert repr(AutomaticTrimmer("strict")) == "AutomaticTrimmer('strict')"
    assert repr(AutomaticTrimmer("noduplicateseqs", platform=None)) == "AutomaticTrimmer('noduplicateseqs', platform=None)"
    assert repr(ManualTrimmer(window=5, platform=None)) == "ManualTrimmer(window=5, platform=None)"
    assert repr(OverlapTrimmer(30, 0.25, platform=None)) == "OverlapTrimmer(30.0, 0.25, platform=None)"
    with pytest.raises(RuntimeError):
        AutomaticTrimmer("strict", platform=None).trim(Alignment(EXAMPLE_001_NAMES, EXAMPLE_001))


# --- TestOverlapTrimmer --------------------------------------------------------------------------

@pytest.mark.parametrize("seq,res", [(80, 80), (40, 60)])
def test_overlap(enog_ali, seq, res):
    expected = load("ENOG411BWBU.seq%d.res%d.fasta" % (seq, res))
    trimmed = OverlapTrimmer(sequence_overlap=seq, residue_overlap=res / 100, platform=PLATFORM).trim(enog_ali)
    assert_trimmed_equal(trimmed, expected)


def test_overlap_docstring_example():
    ali = Alignment(OVERLAP_NAMES, OVERLAP_EXAMPLE)
    trimmed = OverlapTrimmer(40.0, 0.5, platform=PLATFORM).trim(ali)
    assert trimmed.names == [b"Sp17", b"Sp10", b"Sp26"]
    assert list(trimmed.sequences) == ["APDLLL-IGFLLKTV-ATFGDTWFQLWQGLD", "DPAVL--FVIMLGTI-TKFSSEWFFAWLGLE",
                                       "AAALLTYLGLFLGTDYENFAAAAANAWLGLE"]


# --- TestRepresentativeTrimmer -------------------------------------------------------------------

@pytest.mark.parametrize("thr,fname", [(0.75, "maxidentity75"), (0.7, "id70"), (0.5, "id50")])
def test_identity_threshold(enog_ali, thr, fname):
    trimmed = RepresentativeTrimmer(identity_threshold=thr, platform=PLATFORM).trim(enog_ali)
    assert_trimmed_equal(trimmed, load("ENOG411BWBU.%s.fasta" % fname))


@pytest.mark.parametrize("clusters", [1, 2, 5, 10, 50, 209])
def test_clusters_bounds(enog_ali, clusters):
    trimmed = RepresentativeTrimmer(clusters=clusters, platform=PLATFORM).trim(enog_ali)
    assert len(trimmed.sequences) <= max(clusters, 1)
    assert_matches_oracle(trimmed, oracle.pack(list(enog_ali.sequences)), clusters=clusters)


# --- beyond the reference's tests ----------------------------------------------------------------

def test_trimming_a_trimmed_alignment(enog_ali):
    first = ManualTrimmer(gap_threshold=0.9, conservation_percentage=60, platform=PLATFORM).trim(enog_ali)
    assert isinstance(first, TrimmedAlignment)
    second = AutomaticTrimmer("gappyout", platform=PLATFORM).trim(first)
    dense = oracle.pack(list(first.sequences))
    assert_matches_oracle(second, dense, method="gappyout")


def test_golden_vectors(enog_ali):
    vec = np.load(GOLDEN + "/vectors.npz")
    for cname, make in {"strict": lambda: AutomaticTrimmer("strict", platform=PLATFORM),
                        "gt50st50": lambda: ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform=PLATFORM),
                        "ov60_50": lambda: OverlapTrimmer(60, 0.5, platform=PLATFORM),
                        "id50": lambda: RepresentativeTrimmer(identity_threshold=0.5, platform=PLATFORM)}.items():
        t = make().trim(enog_ali)
        assert np.array_equal(np.packbits(np.array(t.residues_mask)), vec[f"enog.{cname}.res"]), cname
        assert np.array_equal(np.packbits(np.array(t.sequences_mask)), vec[f"enog.{cname}.seq"]), cname


@pytest.mark.parametrize("method", ["automated1", "strictplus", "gappyout"])
def test_c2_config_masks(method):
    # BASELINE configs[1] size (500 x 2000): masks identical to the oracle's
    a = synth_msa(500, 2000, 1002)
    ali = Alignment([b"s%d" % i for i in range(500)], [bytes(r) for r in a])
    assert_matches_oracle(AutomaticTrimmer(method, platform=PLATFORM).trim(ali), a, method=method)


def test_threads_are_independent(enog_ali):
    from multiprocessing.pool import ThreadPool

    trimmer = AutomaticTrimmer("strict", platform=PLATFORM)
    alis = [enog_ali, Alignment(EXAMPLE_001_NAMES, EXAMPLE_001), load("halorhodopsin.afa")] * 2
    with ThreadPool(3) as pool:  # README.md:136-152 usage
        out = pool.map(trimmer.trim, alis)
    for ali, t in zip(alis, out):
        assert_matches_oracle(t, oracle.pack(list(ali.sequences)), method="strict")


def test_trimal_warnings_become_runtime_warnings(caplog):
    """What trimAl reports as a warning surfaces as `RuntimeWarning`, the category the reference gives to trimAl's
    warnings (src/trimal/source/reportsystem.cpp:132-173): one per sequence removed because the trimming left it with
    gaps only.  The library's other `MSA_W_*` bits have no trimAl counterpart: they go to the `pytrimal_amd` logger, never
    to `warnings` (a caller running with -W error sees exactly the exceptions the reference would raise)."""
    # the only residues of the third sequence sit in a gappy column: trimming it away leaves gaps only
    ali = Alignment([b"a", b"b", b"c"], ["ACDEF-", "ACDEF-", "-----W"])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        t = ManualTrimmer(gap_threshold=0.5, platform=PLATFORM).trim(ali)
    assert t.sequences_mask == [True, True, False]
    assert [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)] == ["Removing sequence 'c' composed only by gaps"]
    # two such sequences: two warnings, in row order
    ali = Alignment([b"a", b"b", b"c", b"d"], ["ACDEF--", "ACDEF--", "-----W-", "------Y"])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        t = ManualTrimmer(gap_threshold=0.5, platform=PLATFORM).trim(ali)
    assert t.sequences_mask == [True, True, False, False]
    assert [str(w.message) for w in caught] == ["Removing sequence 'c' composed only by gaps", "Removing sequence 'd' composed only by gaps"]
    # two sequences without a single residue: no column counts for the pair, its identity is undefined
    ali = Alignment([b"a", b"b", b"c"], ["ACDEFGHIKLMNPQ", "--------------", "----X---------"])
    import logging

    with warnings.catch_warnings(record=True) as caught, caplog.at_level(logging.INFO, logger="pytrimal_amd"):
        warnings.simplefilter("always")
        RepresentativeTrimmer(identity_threshold=0.9, platform=PLATFORM).trim(ali)
    assert not [w for w in caught if "identity" in str(w.message)]
    assert any("identity" in r.getMessage() for r in caplog.records)
    # and nothing is raised for an ordinary alignment
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        AutomaticTrimmer("gappyout", platform=PLATFORM).trim(Alignment(EXAMPLE_001_NAMES, EXAMPLE_001))


def test_terminal_only_against_oracle(enog_ali):
    """`TrimmedAlignment.terminal_only` (Cleaner::removeOnlyTerminal, _trimal.pyx:1144-1157; semantics [R], see
    oracle/msa_oracle.c): ENOG after two trimmers, the synthetic edge alignment, and the no-gap single sequence."""
    from conftest import edge_msa
    from pytrimal_amd import TrimmedAlignment

    cases = [AutomaticTrimmer("strict", platform=PLATFORM).trim(enog_ali),
             OverlapTrimmer(80, 0.8, platform=PLATFORM).trim(enog_ali)]
    e = edge_msa(40, 300, 5)
    e[:, 60:240][:, ::7] = np.where(e[:, 60:240][:, ::7] == ord("-"), ord("A"), e[:, 60:240][:, ::7])  # some gap-free columns
    cases.append(ManualTrimmer(gap_threshold=0.9, platform=PLATFORM).trim(
        Alignment([b"s%d" % i for i in range(e.shape[0])], [bytes(r) for r in e])))
    cases.append(TrimmedAlignment([b"a"], ["ACDEFG"], residues_mask=[False, True, False, True, True, False]))
    for t in cases:
        a = oracle.pack(list(t.original_alignment().sequences))
        # reading 2 when the trim computed gap statistics (its own counts over the original alignment), reading 0 -- counts
        # over the kept sequences -- for a result without them (OverlapTrimmer, an object built from masks)
        gw = getattr(t, "_gaps_w", None)
        if gw is not None:
            assert np.array_equal(gw, oracle.gaps(a)[0])  # (no window in these trims)
        want = oracle.terminal_only(a, t.residues_mask, t.sequences_mask, reading=2 if gw is not None else 0, gaps_w=gw)
        if want is None:
            with pytest.raises(RuntimeError):
                t.terminal_only()
            continue
        got = t.terminal_only()
        assert got.residues_mask == [bool(x) for x in want]
        assert got.sequences_mask == t.sequences_mask
    assert cases[-1].terminal_only().residues_mask == [True] * 6  # a sequence without gaps: everything comes back
    with pytest.raises(RuntimeError):  # no column without gaps
        TrimmedAlignment([b"a", b"b"], ["A-C", "-D-"], residues_mask=[True, False, True]).terminal_only()


@pytest.mark.parametrize("clusters", [3, 40, 400])
def test_clusters_device_search(monkeypatch, clusters):
    """clusters=K with the threshold search probing the device clustering (the default from 2000 sequences on;
    forced here on a smaller alignment as well) against the oracle's getCutPointClusters + greedy clustering."""
    from pytrimal_amd.synth import synth_msa

    from pytrimal_amd import _lib

    monkeypatch.setenv("MSA_DEVICE_CLUSTERS", "1")
    _lib.reset_thread_context()  # the switches are read once per context
    try:
        for m, n, seed in ((300, 120, 31), (2050, 90, 32)):
            a = synth_msa(m, n, seed)
            a[5] = a[4]  # identical sequences: identity 1
            ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
            trimmed = RepresentativeTrimmer(clusters=clusters, platform=PLATFORM).trim(ali)
            assert_matches_oracle(trimmed, a, clusters=clusters)
    finally:
        monkeypatch.delenv("MSA_DEVICE_CLUSTERS")
        _lib.reset_thread_context()


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.9])
def test_representatives_device_kernels(monkeypatch, thr):
    """identity_threshold clustering on the device (forced below its 2000-sequence default as well) at several row
    counts, masks against the oracle's greedy clustering."""
    from pytrimal_amd.synth import synth_msa

    from pytrimal_amd import _lib

    monkeypatch.setenv("MSA_DEVICE_CLUSTERS", "1")
    _lib.reset_thread_context()
    try:
        for m, n, seed in ((70, 90, 41), (2100, 60, 42), (4200, 40, 43)):
            a = synth_msa(m, n, seed)
            a[7] = a[3]
            ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
            trimmed = RepresentativeTrimmer(identity_threshold=thr, platform=PLATFORM).trim(ali)
            assert_matches_oracle(trimmed, a, identity_threshold=thr)
    finally:
        monkeypatch.delenv("MSA_DEVICE_CLUSTERS")
        _lib.reset_thread_context()


# --- randomized sweep: every trimmer family on small random alignments, masks against the oracle's trim ---------


def _random_alignment(seed):
    r = np.random.default_rng(seed)
    m = int(r.integers(4, 60))
    n = int(r.integers(8, 160))
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    root = alpha[r.integers(0, 20, n)]
    a = np.where(r.random((m, n)) < r.uniform(0.2, 0.9), root, alpha[r.integers(0, 20, (m, n))]).astype(np.uint8)
    a[r.random((m, n)) < float(r.choice([0.05, 0.3, 0.6]))] = ord("-")
    a[r.random((m, n)) < 0.01] = ord("X")
    if r.random() < 0.4:
        a[:, int(r.integers(0, n))] = ord("-")
    if r.random() < 0.3:
        a[int(r.integers(1, m))] = a[0]
    return np.ascontiguousarray(a)


_SWEEP = [
    (lambda: AutomaticTrimmer("strict", platform=PLATFORM), dict(method="strict")),
    (lambda: AutomaticTrimmer("strictplus", platform=PLATFORM), dict(method="strictplus")),
    (lambda: AutomaticTrimmer("gappyout", platform=PLATFORM), dict(method="gappyout")),
    (lambda: AutomaticTrimmer("automated1", platform=PLATFORM), dict(method="automated1")),
    (lambda: AutomaticTrimmer("nogaps", platform=PLATFORM), dict(method="nogaps")),
    (lambda: AutomaticTrimmer("noallgaps", platform=PLATFORM), dict(method="noallgaps")),
    (lambda: ManualTrimmer(gap_threshold=0.7, platform=PLATFORM), dict(gap_threshold=0.7)),
    (lambda: ManualTrimmer(similarity_threshold=0.3, conservation_percentage=40, platform=PLATFORM),
     dict(similarity_threshold=0.3, conservation_percentage=40)),
    (lambda: ManualTrimmer(gap_threshold=0.6, similarity_threshold=0.2, window=1, platform=PLATFORM),
     dict(gap_threshold=0.6, similarity_threshold=0.2, window=1)),
    (lambda: OverlapTrimmer(50.0, 0.6, platform=PLATFORM), dict(sequence_overlap=50.0, residue_overlap=0.6)),
    (lambda: RepresentativeTrimmer(identity_threshold=0.6, platform=PLATFORM), dict(identity_threshold=0.6)),
    (lambda: RepresentativeTrimmer(clusters=3, platform=PLATFORM), dict(clusters=3)),
]


@pytest.mark.parametrize("seed", range(12))
def test_random_alignments_all_trimmers(seed):
    a = _random_alignment(7000 + seed)
    ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
    for make, kw in _SWEEP:
        trimmer = make()
        try:
            expected = oracle.trim(a, **kw)
        except oracle.OracleError:
            with pytest.raises((ValueError, RuntimeError)):
                trimmer.trim(ali)
            continue
        trimmed = trimmer.trim(ali)
        assert trimmed.residues_mask == [bool(x) for x in expected[0]], (seed, repr(trimmer))
        assert trimmed.sequences_mask == [bool(x) for x in expected[1]], (seed, repr(trimmer))


# --- nucleotide alignments: type detection picks 'N' as the indetermination symbol and the NT / degenerate-NT matrix


def _nt_alignment(seed, alphabet, m=40, n=150):
    r = np.random.default_rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    root = alpha[r.integers(0, len(alpha), n)]
    a = np.where(r.random((m, n)) < 0.7, root, alpha[r.integers(0, len(alpha), (m, n))]).astype(np.uint8)
    a[r.random((m, n)) < 0.2] = ord("-")
    a[r.random((m, n)) < 0.02] = ord("N")
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("alphabet", [b"ACGT", b"ACGU", b"ACGTRYKMSW"])
def test_nucleotide_alignments(alphabet):
    a = _nt_alignment(len(alphabet), alphabet)
    ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
    for make, kw in _SWEEP:
        trimmer = make()
        try:
            expected = oracle.trim(a, **kw)
        except oracle.OracleError:
            with pytest.raises((ValueError, RuntimeError)):
                trimmer.trim(ali)
            continue
        trimmed = trimmer.trim(ali)
        assert trimmed.residues_mask == [bool(x) for x in expected[0]], (alphabet, repr(trimmer))
        assert trimmed.sequences_mask == [bool(x) for x in expected[1]], (alphabet, repr(trimmer))


def test_concurrent_contexts_keep_parity():
    """Several threads trimming different alignments at once (one device context and two HIP streams each):
    every result still equals the oracle's."""
    from multiprocessing.pool import ThreadPool

    cases = []
    for seed in range(18):
        a = synth_msa(120 + 37 * (seed % 5), 300 + 64 * (seed % 4), 8800 + seed)
        res, seq, _ = oracle.trim(a, method="strict")
        cases.append((a, [bool(x) for x in res], [bool(x) for x in seq]))
    trimmer = AutomaticTrimmer("strict", platform=PLATFORM)

    def run(case):
        a, res, seq = case
        ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
        out = trimmer.trim(ali)
        return out.residues_mask == res and out.sequences_mask == seq

    with ThreadPool(4) as pool:
        for _ in range(3):
            assert all(pool.map(run, cases))


def test_a_trimmer_reconfigured_after_its_first_trim():
    """The parameter block of a trimmer is cached per matrix: an attribute written after the first trim, or `__setstate__` on
    a used object, must configure the next trim (round 4's advisor: the cache kept the old method and thresholds)."""
    a = synth_msa(60, 400, 91)
    ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
    t = AutomaticTrimmer("strict", platform=PLATFORM)
    assert_matches_oracle(t.trim(ali), a, method="strict")
    t.method = "gappyout"
    assert_matches_oracle(t.trim(ali), a, method="gappyout")
    t.__setstate__({"method": "nogaps", "platform": PLATFORM})
    assert_matches_oracle(t.trim(ali), a, method="nogaps")
    mt = ManualTrimmer(gap_threshold=0.3, platform=PLATFORM)
    assert_matches_oracle(mt.trim(ali), a, gap_threshold=0.3)
    state = mt.__getstate__()
    state["gap_threshold"] = float(np.float32(1) - np.float32(0.8))  # (the state holds the maximum gap FRACTION, as the reference's)
    mt.__setstate__(state)
    assert_matches_oracle(mt.trim(ali), a, gap_threshold=0.8)


@pytest.mark.parametrize("shape", [(46, 1181), (209, 1227), (400, 900), (9, 3), (16, 1029), (1024, 301), (1025, 300), (1000, 600)])
@pytest.mark.parametrize("args", [(80, 0.8), (40, 0.6), (95, 0.95), (10, 0.1)])
def test_overlap_trimmer_one_wait_on_small_alignments(shape, args):
    """OverlapTrimmer on alignments the compact front kernel takes (up to 1024 sequences; one shape beyond): counts, the sequences
    that stay (decided on the device as the host decides it) and the column counts over them come back behind ONE wait -- three
    launches that store into pinned host memory themselves -- against the oracle, whether sequences go or not; with indeterminations,
    a row count that is no multiple of the sixteen waves of the column counts, a width that is no multiple of four."""
    m, n = shape
    a = synth_msa(m, n, 300 + m)
    a[3, : n // 2] = ord("-")  # (a sequence that overlaps little)
    a[7] = ord("-")
    a[7, : min(5, n)] = a[6, : min(5, n)]
    a[np.random.default_rng(m + n).random((m, n)) < 0.03] = ord("X")
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
    seq_ov, res_ov = args
    trimmed = OverlapTrimmer(seq_ov, res_ov, platform=PLATFORM).trim(ali)
    assert_matches_oracle(trimmed, a, sequence_overlap=seq_ov, residue_overlap=res_ov)
