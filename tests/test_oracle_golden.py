"""The CPU oracle against every fixture the reference's own tests hold for this path
(pins P1..P9).  This is what makes the oracle trustworthy as the checker
of the HIP path; the reference itself cannot be built here (its trimAl submodule is empty).
"""
import os
import zlib

import numpy as np
import pytest

import oracle
from conftest import EXAMPLE_001, EXAMPLE_001_NAMES, GOLDEN, OVERLAP_EXAMPLE, OVERLAP_NAMES, data_path, edge_msa
from pytrimal_amd.synth import synth_msa


def kept(names, a, res, seq):
    return ([n for n, k in zip(names, seq) if k], [bytes(a[i][res]) for i in range(a.shape[0]) if seq[i]])


def assert_fixture(names, a, res, seq, fname):
    en, es = oracle.read_fasta(data_path(fname))
    gn, gs = kept(names, a, res, seq)
    assert gn == en
    assert gs == es


# --- P1: gaps + calcCutPoint + cleanByCutValueOverpass incl. recovery (tests/test_manual_trimmer.py:13-35)
@pytest.mark.parametrize("gt,cons", [(0.9, 60), (0.4, 40)])
def test_p1_gap_threshold_fixtures(enog, gt, cons):
    names, _, a = enog
    res, seq, _ = oracle.trim(a, gap_threshold=gt, conservation_percentage=cons)
    assert_fixture(names, a, res, seq, "ENOG411BWBU.cons%02d.gt%02d.fasta" % (cons, int(gt * 100)))


# --- P2: overlap (tests/test_overlap_trimmer.py:14-29)
@pytest.mark.parametrize("so,ro", [(80, 80), (40, 60)])
def test_p2_overlap_fixtures(enog, so, ro):
    names, _, a = enog
    res, seq, _ = oracle.trim(a, sequence_overlap=so, residue_overlap=ro / 100)
    assert_fixture(names, a, res, seq, "ENOG411BWBU.seq%d.res%d.fasta" % (so, ro))


# --- P3: identity + greedy clustering (tests/test_representative_trimmer.py:70-72; id50/id70 unreferenced)
@pytest.mark.parametrize("thr,fname", [(0.75, "maxidentity75"), (0.7, "id70"), (0.5, "id50")])
@pytest.mark.parametrize("sort_mode", [0, 1])
def test_p3_representative_fixtures(enog, thr, fname, sort_mode):
    names, _, a = enog
    res, seq, _ = oracle.trim(a, identity_threshold=thr, sort_mode=sort_mode)
    assert_fixture(names, a, res, seq, "ENOG411BWBU.%s.fasta" % fname)


# --- P4: removeDuplicates keeps the later duplicate (tests/test_automatic_trimmer.py:60-62)
def test_p4_noduplicateseqs_fixture(enog):
    names, _, a = enog
    res, seq, _ = oracle.trim(a, method="noduplicateseqs")
    assert_fixture(names, a, res, seq, "ENOG411BWBU.noduplicateseqs.fasta")
    assert names[int(np.flatnonzero(~seq)[0])] == b"39947.LOC_Os07g06262.1"


# --- P5: example.001 (6 x 46): automated1, strictplus, window
def test_p5_example001_automated1_docstring():
    a = oracle.pack(EXAMPLE_001)
    res, seq, info = oracle.trim(a, method="automated1")
    assert info.selected == oracle.STRICT
    # src/pytrimal/_trimal.pyx:33-38
    expected = ["VWLFPWNGLQIHMMGII", "EWFFAWLGLEINMMVII", "AAANAWLGLEINMMAQI", "SWYLAWLGLEINMMAII",
                "TWFQLWQGLDLNKMPVF", "AWFQAWGGLEINKQAIL"]
    assert [bytes(r[res]).decode() for r in a] == expected and seq.all()


def test_p5_example001_strictplus_readme():
    a = oracle.pack(EXAMPLE_001)
    res, seq, _ = oracle.trim(a, method="strictplus")
    # README.md:118-123
    expected = ["GIVLVWLFPWNGLQIHMMGII", "VIMLEWFFAWLGLEINMMVII", "GLFLAAANAWLGLEINMMAQI",
                "GIYLSWYLAWLGLEINMMAII", "GFLLTWFQLWQGLDLNKMPVF", "GLHMAWFQAWGGLEINKQAIL"]
    assert [bytes(r[res]).decode() for r in a] == expected


def test_p5_example001_window_fixture():
    a = oracle.pack(EXAMPLE_001)
    res, seq, _ = oracle.trim(a, gap_threshold=0.9, window=3)
    names, seqs = oracle.read_clustal(data_path("example.001.gt90.w3.clw"))
    assert names == EXAMPLE_001_NAMES
    assert [bytes(r[res]) for r in a] == seqs


def test_p5_known_answers_survey_b2():
    a = oracle.pack(EXAMPLE_001)
    g, hist, mx, tot = oracle.gaps(a)
    assert g[:28].tolist() == [5, 5, 4, 4, 4, 2, 2, 0, 0, 0, 0, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 5, 5]
    assert not g[28:].any() and tot == 43
    hit, dst = oracle.pair_counts(a)
    assert [(int(hit[0, j]), int(dst[0, j])) for j in range(1, 6)] == [(16, 40), (12, 46), (16, 40), (9, 44), (10, 41)]
    sel, avg, mxs = oracle.select_method(oracle.identities(hit, dst))
    assert sel == oracle.STRICT and abs(avg - 0.33040193) < 1e-7 and abs(mxs - 0.40969133) < 1e-7
    mdk, _ = oracle.similarity(a, oracle.weights(hit, dst), g, *oracle.aa_matrix())
    head = "00000000 00000000 35489d59 3f800000 34cfc5d8 397bff77 399bf5d2 3801cf42 380fde44 3b439374".split()
    assert ["%08x" % x for x in mdk.view(np.uint32)[:10]] == head
    assert mdk[26] == 0 and mdk[27] == 0  # 5/6 gaps >= 0.8


# --- P6: the OverlapTrimmer docstring example (_trimal.pyx:1676-1701)
def test_p6_overlap_docstring():
    a = oracle.pack(OVERLAP_EXAMPLE)
    res, seq, _ = oracle.trim(a, sequence_overlap=40.0, residue_overlap=0.5)
    n, s = kept(OVERLAP_NAMES, a, res, seq)
    assert n == [b"Sp17", b"Sp10", b"Sp26"]
    assert s == [b"APDLLL-IGFLLKTV-ATFGDTWFQLWQGLD", b"DPAVL--FVIMLGTI-TKFSSEWFFAWLGLE", b"AAALLTYLGLFLGTDYENFAAAAANAWLGLE"]


# --- P7 / P8: error cases
def test_p7_invalid_residue_is_an_error():
    a = oracle.pack(["MKKBO", "MKKAY"])
    with pytest.raises(oracle.OracleError) as e:
        oracle.trim(a, method="strict")
    assert e.value.code == oracle.E_UNDEFINED_SYMBOL


def test_p8_window_too_big():
    a = oracle.pack(["M-KKV", "MY-KV"])
    with pytest.raises(oracle.OracleError) as e:
        oracle.trim(a, gap_threshold=0.9, window=100)
    assert e.value.code == oracle.E_WINDOW_TOO_BIG


# --- P9: SimilarityMatrix known answers
def test_p9_matrix_known_answers():
    vh, d = oracle.aa_matrix()
    A = oracle.AA_ALPHABET
    assert d.shape == (20, 20) and d[A.index("A"), A.index("A")] == 0
    assert d[A.index("A"), A.index("R")].view(np.uint32) == 0x412D1104
    assert abs(d.max() - 21.354156) < 1e-5
    vh5, d5 = oracle.nt_matrix()
    assert d5.shape == (5, 5) and d5[0, 0] == 0 and d5[0, 3] > 0
    vh15, d15 = oracle.nt_matrix(True)
    assert d15.shape == (15, 15) and abs(d15[0, 3] - 1.5184) < 1e-4  # _trimal.pyx:2042-2046
    sim = np.array([[5, 0, 0, 4], [0, 5, 4, 0], [0, 4, 5, 0], [4, 0, 0, 5]], dtype=np.float32)
    vh4, d4 = oracle.make_matrix(sim, "ATCG")  # tests/test_similarity_matrix.py:10-18
    assert vh4[ord("A") - 65] == 0 and vh4[ord("G") - 65] == 3 and vh4[ord("B") - 65] == -1


# --- known answers of SURVEY appendix B.3 (integers, exact)
def test_enog_known_answers(enog):
    _, _, a = enog
    assert a.shape == (209, 1227)
    g, hist, mx, tot = oracle.gaps(a)
    assert tot == 102944 and mx == 208 and zlib.crc32(g.astype("<i4").tobytes()) == 0x43A2585E
    assert int(((g / 209) >= 0.8).sum()) == 396
    hit, dst = oracle.pair_counts(a)
    iu = np.triu_indices(209, 1)
    assert int(hit[iu].sum()) == 11040910 and int(dst[iu].sum()) == 17878297
    assert (int(hit[0, 1]), int(dst[0, 1])) == (22, 207)
    assert zlib.crc32(hit[iu].astype("<u4").tobytes()) == 0x17D4D68F
    assert zlib.crc32(dst[iu].astype("<u4").tobytes()) == 0xD58539F5
    sel, avg, mxs = oracle.select_method(oracle.identities(hit, dst))
    assert sel == oracle.GAPPYOUT and abs(avg - 0.6148056) < 1e-6 and abs(mxs - 0.846132) < 1e-6
    sizes = {0.5: 13, 0.7: 59, 0.75: 78}
    for thr, k in sizes.items():
        _, seq, _ = oracle.trim(a, identity_threshold=thr)
        assert int(seq.sum()) == k


# --- the committed golden vectors still describe the oracle
def test_golden_vectors_are_current(enog):
    vec = np.load(os.path.join(GOLDEN, "vectors.npz"))
    a = enog[2]
    g, _, _, _ = oracle.gaps(a)
    assert np.array_equal(vec["enog.gaps"], g)
    hit, dst = oracle.pair_counts(a)
    assert vec["enog.hit_crc"] == np.uint32(zlib.crc32(hit.tobytes()))
    mdk, q = oracle.similarity(a, oracle.weights(hit, dst), g, *oracle.aa_matrix())
    assert np.array_equal(vec["enog.mdk_bits"], mdk.view(np.uint32))
    for cname, kw in {"strict": dict(method="strict"), "gappyout": dict(method="gappyout"),
                      "gt50st50": dict(gap_threshold=0.5, similarity_threshold=0.5)}.items():
        res, seq, _ = oracle.trim(a, **kw)
        assert np.array_equal(vec[f"enog.{cname}.res"], np.packbits(res))
    s = synth_msa(64, 256, 64256)
    res, seq, _ = oracle.trim(s, method="automated1")
    assert np.array_equal(vec["synth64x256.automated1.res"], np.packbits(res))


# --- properties of the oracle itself
def test_overlap_closed_form_matches_definition():
    # hit count = (#valid - 1) for a valid residue, (#same symbol - 1) otherwise
    a = edge_msa(40, 90, 3)
    m, n = a.shape
    ov = oracle.overlap(a, 0.5)
    need = int(np.ceil(np.float32(0.5) * np.float32(m - 1)))
    good = np.zeros(m, dtype=np.int64)
    for c in range(n):
        col = a[:, c]
        valid = (col != ord("-")) & (col != ord("X"))
        for i in range(m):
            hit = int(valid.sum()) - 1 if valid[i] else int((col == col[i]).sum()) - 1
            good[i] += hit >= need
    assert np.array_equal(ov, (good.astype(np.float32) / np.float32(n)))


def test_alignment_type_detection():
    assert oracle.alignment_type(oracle.pack(EXAMPLE_001)) == 4
    assert oracle.alignment_type(oracle.pack(OVERLAP_EXAMPLE)) == 4
    assert oracle.alignment_type(oracle.pack(["ACGTACGTAC", "ACGT-CGTAC"])) == 1
    assert oracle.alignment_type(oracle.pack(["ACGUACGUAC", "ACGU-CGUAC"])) == 2


def test_terminal_only_readings():
    """The readings of Cleaner::removeOnlyTerminal the oracle restates ([R], unverified; the product follows reading 2:
    the gap vector of the original alignment, every sequence)."""
    a = oracle.pack(["A-CDEF-H", "ABC-EFGH", "ABCDEFG-"])
    keep = np.array([1, 0, 1, 0, 0, 1, 0, 1], dtype=bool)
    seqs = np.ones(3, dtype=bool)
    # columns without gaps: 0, 2, 4, 5 -> everything from 0 to 5 is restored, 6 and 7 keep the trimmer's decision
    assert oracle.terminal_only(a, keep, seqs, reading=0).tolist() == [True] * 6 + [False, True]
    # first / last kept column: 0 .. 7
    assert oracle.terminal_only(a, keep, seqs, reading=1).tolist() == [True] * 8
    # dropping the second sequence makes column 3 gap free and column 1 not
    assert oracle.terminal_only(a, keep, np.array([1, 0, 1], dtype=bool), reading=0).tolist() == [True] * 6 + [False, True]
    assert oracle.terminal_only(oracle.pack(["A-", "-B"]), [True, False], [True, True], reading=0) is None
    # reading 2: the dropped sequence still counts (column 3 keeps its gap); with all sequences kept it equals reading 0
    assert oracle.terminal_only(a, keep, np.array([1, 0, 1], dtype=bool), reading=2).tolist() == [True] * 6 + [False, True]
    assert oracle.terminal_only(a, keep, seqs, reading=2).tolist() == oracle.terminal_only(a, keep, seqs, reading=0).tolist()
    b = oracle.pack(["AB-D-", "A-CDE", "ABCDE"])
    only_last = np.array([0, 0, 1], dtype=bool)
    assert oracle.terminal_only(b, [False] * 5, only_last, reading=0).tolist() == [True] * 5   # the kept sequence holds no gap
    # the original alignment: columns 0 and 3 are free of gaps -> 0 .. 3 come back, column 4 keeps the trimmer's decision
    assert oracle.terminal_only(b, [False] * 5, only_last, reading=2).tolist() == [True] * 4 + [False]
    assert oracle.terminal_only(b, [False] * 5, only_last, reading=2, gaps_w=[1, 0, 0, 1, 1]).tolist() == [False, True, True, False, False]



@pytest.mark.skipif(oracle.lib_avx2() is None, reason="host CPU without AVX2")
@pytest.mark.parametrize("shape", [(2, 5), (37, 301), (130, 64), (64, 8200)])
def test_avx2_flavour_equals_scalar(shape):
    """The AVX2 pair counts and similarity (the CPU baseline's SIMD flavour) against the scalar restatement:
    integers equal, float32 sums bit-identical, errors at the same residue."""
    m, n = shape
    a = synth_msa(m, n, 77 + m)
    hit, dst = oracle.pair_counts(a)
    hit2, dst2 = oracle.pair_counts(a, avx2=True)
    assert np.array_equal(hit, hit2) and np.array_equal(dst, dst2)
    g, _, _, _ = oracle.gaps(a)
    w = oracle.weights(hit, dst)
    mdk, q = oracle.similarity(a, w, g, *oracle.aa_matrix())
    mdk2, q2 = oracle.similarity(a, w, g, *oracle.aa_matrix(), avx2=True)
    assert np.array_equal(q.view(np.uint32), q2.view(np.uint32))
    assert np.array_equal(mdk.view(np.uint32), mdk2.view(np.uint32))
    if n > 20:
        b = a.copy()
        b[m // 2, 17] = ord("B")  # not in BLOSUM62's alphabet
        b[0, 19] = ord("?")
        for flavour in (False, True):
            with pytest.raises(oracle.OracleError) as e:
                oracle.similarity(b, w, None, *oracle.aa_matrix(), avx2=flavour)
            assert e.value.detail[:2] == (m // 2, 17)
