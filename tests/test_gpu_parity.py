"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bit-exact for every integer output (gap counts, hit/dst, masks) and for the float32 identity /
weight matrices and the similarity quotient Q (the reference's sequential accumulation order is
reproduced) and for MDK = min(1, (float)exp(-(double)Q)): the device only keeps an exponential that
provably rounds to the same float as the host's (sim_finish_kernel), the others are evaluated on the host.
No float tolerance anywhere in this suite.
"""
import os

import numpy as np
import pytest

import oracle
from conftest import EXAMPLE_001, OVERLAP_EXAMPLE, data_path, edge_msa
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa

pytestmark = pytest.mark.gpu



@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture
def ctx_with(monkeypatch):
    """Factory of contexts created under given MSA_* diagnostic switches (the library reads them once, when a
    context is created)."""
    made = []

    def make(**env):
        for name in ("MSA_SIM_KERNEL", "MSA_LG_R0", "MSA_LG_BIG", "MSA_LG_ROUNDS", "MSA_LG_SPLIT", "MSA_MDK_HOST", "MSA_PIPELINE", "MSA_UPLOAD_DIRECT",
                     "MSA_COMPACT", "MSA_FLAT_MAX_M", "MSA_FLAT_U", "MSA_ZEROCOPY_KB", "MSA_FRONT_CW", "MSA_FRONT_NT", "MSA_FRONT_XCD",
                     "MSA_FRONT_FROM_M", "MSA_PAIR_TI", "MSA_PAIR_K", "MSA_LISTS_FUSED", "MSA_LG_HALVES", "MSA_LG_PIPE", "MSA_LG_PIPE_K", "MSA_LG_XSEG", "MSA_LG_XSEG_KX"):
            monkeypatch.delenv(name, raising=False)
        for name, value in env.items():
            if value:
                monkeypatch.setenv(name, value)
        c = _lib.Context(0)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def load_fixture(name):
    _, seqs = oracle.read_fasta(data_path(name))
    return oracle.pack(seqs)


def all_stats(ctx, a, indet=ord("X"), matrix=None):
    """Run every statistic on the device and on the oracle; compare."""
    m, n = a.shape
    ctx.upload(a, indet)
    g, x = ctx.gaps(with_indet=True)
    og, _, _, _ = oracle.gaps(a)
    assert np.array_equal(g, og)
    assert np.array_equal(x, (a == indet).sum(axis=0))
    hit, dst = ctx.pair_counts()
    ohit, odst = oracle.pair_counts(a, indet)
    assert np.array_equal(hit, ohit)
    assert np.array_equal(dst, odst)
    ident, w = ctx.identities()
    assert np.array_equal(bits(ident), bits(oracle.identities(ohit, odst)))
    ow = oracle.weights(ohit, odst)
    assert np.array_equal(bits(w), bits(ow))
    if m > 1:
        avg, mx = ctx.identity_stats()
        _, oavg, omx = oracle.select_method(oracle.identities(ohit, odst))
        assert bits(avg) == bits(oavg) and bits(mx) == bits(omx)
    vhash, dist = matrix if matrix is not None else oracle.aa_matrix()
    try:
        omdk, oq = oracle.similarity(a, ow, og, vhash, dist, indet)
    except oracle.OracleError as err:
        with pytest.raises(ValueError):
            ctx.similarity(vhash, dist)
        return err
    mdk, q = ctx.similarity(vhash, dist)
    assert np.array_equal(bits(q), bits(oq)), "similarity quotient must be bit-exact (reference order)"
    assert np.array_equal(bits(mdk), bits(omdk)), "MDK must be bit-exact"
    for thr in (0.5, 0.8):
        ov = ctx.overlap(thr)
        assert np.array_equal(bits(ov), bits(oracle.overlap(a, thr, indet)))
    return None


def test_example_001(ctx):
    all_stats(ctx, oracle.pack(EXAMPLE_001))


def test_overlap_docstring_example(ctx):
    a = oracle.pack(OVERLAP_EXAMPLE)
    err = all_stats(ctx, a)
    assert err is None


def test_enog(ctx, enog):
    all_stats(ctx, enog[2])


def test_lowercase_fixture(ctx):
    # PF12574.full.afa has lower-case residues: raw-byte identity vs upper-cased similarity
    names, seqs = oracle.read_fasta(data_path("PF12574.full.afa"))
    a = oracle.pack(seqs)
    all_stats(ctx, a)


def test_halorhodopsin(ctx):
    all_stats(ctx, load_fixture("halorhodopsin.afa"))


def test_edge_symbols_raise_like_the_oracle(ctx):
    # B / Z / lower-case x are not in the 20-letter BLOSUM62 alphabet -> UndefinedSymbol
    err = all_stats(ctx, edge_msa())
    assert err is not None and err.code == oracle.E_UNDEFINED_SYMBOL


def test_edge_symbols_with_wide_matrix(ctx):
    # same data with an alphabet that defines every letter: full numeric comparison
    alphabet = "ABCDEFGHIKLMNPQRSTVWXYZ"
    r = np.random.default_rng(5)
    sim = r.integers(-4, 9, (len(alphabet), len(alphabet))).astype(np.float32)
    sim = (sim + sim.T) / 2
    a = edge_msa()
    a[a == ord("x")] = ord("X")
    err = all_stats(ctx, a, matrix=oracle.make_matrix(sim, alphabet))
    assert err is None


@pytest.mark.parametrize("m,n", [(1, 1), (1, 40), (2, 1), (2, 5), (3, 31), (3, 33), (5, 64), (63, 100),
                                 (64, 64), (65, 129), (127, 200), (129, 95), (130, 257), (200, 33)])
def test_ragged_sizes(ctx, m, n):
    a = synth_msa(m, n, 100 + m * 7 + n)
    all_stats(ctx, a)


def test_all_gap_and_single_residue_columns(ctx):
    a = synth_msa(40, 120, 3)
    a[:, 5] = ord("-")
    a[:, 6] = ord("-")
    a[3, 6] = ord("A")       # one residue only: den == 0 -> MDK 0
    a[:, 7] = ord("X")
    a[:, 8] = ord("A")       # fully conserved: Q == 0 -> MDK 1
    all_stats(ctx, a)


def test_incorrect_symbol_position(ctx):
    a = synth_msa(20, 90, 4)
    a[a == ord("X")] = ord("A")
    a[7, 40] = ord("*")      # not a letter -> IncorrectSymbol
    a[9, 40] = ord("B")
    a[2, 60] = ord("B")
    ctx.upload(a, ord("X"))
    vhash, dist = oracle.aa_matrix()
    hit, dst = oracle.pair_counts(a)
    g, _, _, _ = oracle.gaps(a)
    with pytest.raises(oracle.OracleError) as o:
        oracle.similarity(a, oracle.weights(hit, dst), g, vhash, dist)
    assert o.value.code == oracle.E_INCORRECT_SYMBOL and o.value.detail[:2] == (7, 40)
    with pytest.raises(ValueError, match="incorrect"):
        ctx.similarity(vhash, dist)


def test_windowed_gap_cut_vector(ctx):
    a = synth_msa(50, 300, 21)
    ctx.upload(a, ord("X"))
    g = ctx.gaps()
    gw = oracle.gaps_window(g, 3)
    vhash, dist = oracle.aa_matrix()
    hit, dst = oracle.pair_counts(a)
    omdk, oq = oracle.similarity(a, oracle.weights(hit, dst), gw, vhash, dist)
    mdk, q = ctx.similarity(vhash, dist, gw)
    assert np.array_equal(bits(q), bits(oq))
    assert np.array_equal(bits(mdk), bits(omdk))


def test_c2_full_size(ctx):
    # BASELINE config 2: 500 x 2000, every statistic against the oracle
    all_stats(ctx, synth_msa(500, 2000, 1002))


def test_mdk_exponentials_device_and_host_agree(ctx_with):
    """MDK values the device vouches for (sim_finish_kernel's rounding test) against the same values with every
    exponential handed to the host (MSA_MDK_HOST=1): identical bits, and both equal to the oracle's."""
    a = synth_msa(300, 6000, 77)
    vhash, dist = oracle.aa_matrix()
    out = []
    for env in ({}, {"MSA_MDK_HOST": "1"}):
        c = ctx_with(**env)
        c.upload(a, ord("X"))
        out.append(c.similarity(vhash, dist))
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and np.array_equal(bits(out[0][1]), bits(out[1][1]))
    hit, dst = oracle.pair_counts(a)
    omdk, _ = oracle.similarity(a, oracle.weights(hit, dst), oracle.gaps(a)[0], vhash, dist)
    assert np.array_equal(bits(out[0][0]), bits(omdk))


def test_attach_device_buffer(ctx):
    torch = pytest.importorskip("torch")
    a = synth_msa(70, 333, 9)
    ld = 384
    buf = torch.zeros((70, ld), dtype=torch.uint8, device="cuda:0")
    buf[:, :333] = torch.from_numpy(a).to("cuda:0")
    buf[:, 333:] = 0x41  # garbage in the padding must not matter
    torch.cuda.synchronize()
    ctx.attach(buf.data_ptr(), 70, 333, ld, ord("X"))
    assert np.array_equal(ctx.gaps(), oracle.gaps(a)[0])
    hit, dst = ctx.pair_counts()
    ohit, odst = oracle.pair_counts(a)
    assert np.array_equal(hit, ohit) and np.array_equal(dst, odst)


# --- BASELINE.json full sizes: size-independent properties + oracle on slices -------------------

def test_c3_full_size_properties(ctx):
    """configs[2] (2000 x 10000): the oracle needs ~10 s for the whole thing, so integers are
    checked through identities that hold at any size and the similarity on a column slice whose
    W comes from the (separately verified) device pair counts."""
    m, n = 2000, 10000
    a = synth_msa(m, n, 1003)
    ctx.upload(a, ord("X"))
    g, x = ctx.gaps(with_indet=True)
    assert np.array_equal(g, (a == ord("-")).sum(axis=0))           # exact, numpy on the host
    assert np.array_equal(x, (a == ord("X")).sum(axis=0))
    hit, dst = ctx.pair_counts()
    assert np.array_equal(hit, hit.T) and np.array_equal(dst, dst.T)
    assert not hit.diagonal().any() and not dst.diagonal().any()
    assert (hit <= dst).all() and int(dst.max()) <= n
    valid = (a != ord("-")) & (a != ord("X"))
    nv = valid.sum(axis=1).astype(np.int64)
    # dst[i][j] = |valid_i or valid_j| = nv_i + nv_j - |valid_i and valid_j|: check row sums exactly
    both = valid.astype(np.float32) @ valid.astype(np.float32).T     # exact: counts < 2^24
    want_dst = (nv[:, None] + nv[None, :] - both.astype(np.int64))
    np.fill_diagonal(want_dst, 0)
    assert np.array_equal(dst.astype(np.int64), want_dst)
    # hit on a row slice against the oracle
    rows = np.r_[0:40, 990:1010, 1960:2000]
    ohit, odst = oracle.pair_counts(a[rows])
    assert np.array_equal(hit[np.ix_(rows, rows)], ohit)
    ident, w = ctx.identities()
    with np.errstate(invalid="ignore", divide="ignore"):
        want = np.where(dst > 0, hit.astype(np.float32) / dst.astype(np.float32), np.float32(0))
    assert np.array_equal(bits(ident), bits(want))
    assert np.array_equal(bits(w), bits(np.where(np.eye(m, dtype=bool), np.float32(0), np.float32(1) - want)))
    # similarity: 3 x 64 columns against the oracle (sequential float32 order), bit-exact Q
    vhash, dist = oracle.aa_matrix()
    mdk, q = ctx.similarity(vhash, dist)
    for c0 in (0, 4992, 9936):
        sl = slice(c0, c0 + 64)
        omdk, oq = oracle.similarity(np.ascontiguousarray(a[:, sl]), w, g[sl], vhash, dist)
        assert np.array_equal(bits(q[sl]), bits(oq))
        assert np.array_equal(bits(mdk[sl]), bits(omdk))
    assert (mdk[(g / np.float32(m)) >= np.float32(0.8)] == 0).all()
    assert ((mdk >= 0) & (mdk <= 1)).all()
    avg, mx = ctx.identity_stats()
    _, oavg, omx = oracle.select_method(ident)
    assert bits(avg) == bits(oavg) and bits(mx) == bits(omx)


def test_c4_full_size_pair_counts(ctx):
    """configs[3] (5000 x 5000): pair counts on row slices against the oracle + global identities."""
    m, n = 5000, 5000
    a = synth_msa(m, n, 1004)
    ctx.upload(a, ord("X"))
    hit, dst = ctx.pair_counts()
    assert np.array_equal(hit, hit.T) and np.array_equal(dst, dst.T) and (hit <= dst).all()
    for rows in (np.r_[0:64], np.r_[2470:2530], np.r_[4900:5000], np.arange(3, 5000, 97)):
        ohit, odst = oracle.pair_counts(a[rows])
        assert np.array_equal(hit[np.ix_(rows, rows)], ohit)
        assert np.array_equal(dst[np.ix_(rows, rows)], odst)
    # overlap through its closed form (the O(n m^2) oracle loop would take a minute here; the
    # closed form is checked against that definition in tests/test_oracle_golden.py)
    ov = ctx.overlap(0.5)
    is_gap, is_x = a == ord("-"), a == ord("X")
    ng, nx = is_gap.sum(axis=0), is_x.sum(axis=0)
    need = int(np.ceil(np.float32(0.5) * np.float32(m - 1)))
    hits = np.where(is_gap, ng[None, :] - 1, np.where(is_x, nx[None, :] - 1, (m - ng - nx)[None, :] - 1))
    want = (hits >= need).sum(axis=1).astype(np.float32) / np.float32(n)
    assert np.array_equal(bits(ov), bits(want))


@pytest.mark.parametrize("shape", [(2, 5), (9, 33), (65, 64), (130, 31), (513, 97), (700, 300), (1030, 70)])
def test_pair_kernel_shapes(ctx, shape):
    """the software-pipelined pair-count loop on the triangle-only grid against the oracle (odd and even chunk counts,
    rows that end inside a tile)"""
    m, n = shape
    a = synth_msa(m, n, 300 + m)
    ohit, odst = oracle.pair_counts(a, ord("X"))
    ctx.upload(a, ord("X"))
    hit, dst = ctx.pair_counts()
    assert np.array_equal(hit, ohit) and np.array_equal(dst, odst)
    ident, w = ctx.identities()
    assert np.array_equal(bits(ident), bits(oracle.identities(ohit, odst)))
    assert np.array_equal(bits(w), bits(oracle.weights(ohit, odst)))


def test_pair_kernel_two_rows_per_lane(ctx):
    """m large enough for the second regime of the pair pass (two rows j per lane, the plain loop): row slices against
    the oracle"""
    m, n = 4300, 40
    a = synth_msa(m, n, 77)
    ctx.upload(a, ord("X"))
    hit, dst = ctx.pair_counts()
    assert np.array_equal(hit, hit.T) and np.array_equal(dst, dst.T)
    for rows in (np.r_[0:70], np.r_[2100:2200], np.r_[4200:4300], np.arange(5, m, 61)):
        ohit, odst = oracle.pair_counts(a[rows], ord("X"))
        assert np.array_equal(hit[np.ix_(rows, rows)], ohit)
        assert np.array_equal(dst[np.ix_(rows, rows)], odst)


@pytest.mark.parametrize("shape", [(1, 1), (3, 31), (65, 64), (209, 1227), (700, 5000), (2100, 4100), (300, 1000), (64, 4000)])
def test_upload_paths_agree(ctx, ctx_with, shape):
    """packed matrix (the runtime's pitched copy, or re-pitched in pinned pieces), array of row pointers, already pitched
    matrix, page-locked rows (one DMA copy from where they lie): the same residues on the device (gap counts and a
    strict trim's similarity values as the witnesses), also on a context that held another shape in between"""
    m, n = shape
    a = synth_msa(m, n, 40 + m)
    vhash, dist = oracle.aa_matrix()
    ctx.upload(a, ord("X"))
    g0, x0 = ctx.gaps(with_indet=True)
    assert np.array_equal(g0, (a == ord("-")).sum(axis=0)) and np.array_equal(x0, (a == ord("X")).sum(axis=0))
    q0 = ctx.similarity(vhash, dist)[1] if m > 1 else None
    ctx.upload_rows([bytes(r) for r in a], ord("X"))
    g1, x1 = ctx.gaps(with_indet=True)
    assert np.array_equal(g1, g0) and np.array_equal(x1, x0)
    if m > 1:
        assert np.array_equal(bits(ctx.similarity(vhash, dist)[1]), bits(q0))
    ld = (n + 63) // 64 * 64
    pitched = np.zeros((m, ld), dtype=np.uint8)
    pitched[:, :n] = a
    check = _lib.check
    check(ctx.lib, ctx.h, ctx.lib.msa_upload_packed(ctx.h, _lib.ptr(pitched), m, n, ld, ord("X")))
    ctx.shape = (m, n)
    g2, x2 = ctx.gaps(with_indet=True)
    assert np.array_equal(g2, g0) and np.array_equal(x2, x0)
    locked = a.copy()
    other = synth_msa(37, 130, 3)
    staged = ctx_with(MSA_UPLOAD_DIRECT="0")
    for c, pin in ((ctx, True), (staged, False), (ctx, True)):
        c.upload(other, ord("X"))  # (another shape first: the padding columns of the device rows are dirty)
        assert np.array_equal(c.gaps(), (other == ord("-")).sum(axis=0))
        c.upload(locked, ord("X"), pin=pin)
        g3, x3 = c.gaps(with_indet=True)
        assert np.array_equal(g3, g0) and np.array_equal(x3, x0)
        if m > 1:
            assert np.array_equal(bits(c.similarity(vhash, dist)[1]), bits(q0))
    del locked  # (its finalizer unregisters the rows)


def _alphabet_case(m, n, letters, seed):
    r = np.random.default_rng(seed)
    alpha = np.frombuffer(letters, dtype=np.uint8)
    a = alpha[r.integers(0, len(alpha), (m, n))].copy()
    a[r.random((m, n)) < 0.15] = ord("-")
    a[r.random((m, n)) < 0.03] = ord("X")
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("letters", [b"AC", b"ACDEFGHIKLMNPQRSTVWY", b"ACDEFGHIKLMNPQRSTVWYBZJUO*.?",
                                     b"ACDEFGHIKLMNPQRSTVWYacdefghiklmnpqrstvwy", bytes(range(33, 127))])
@pytest.mark.parametrize("shape", [(70, 200), (300, 97)])
def test_pair_counts_by_alphabet_size(ctx, letters, shape):
    """the pair pass compares raw bytes: alphabets from two symbols to every printable character, and a second
    alignment with another alphabet on the same context"""
    m, n = shape
    a = _alphabet_case(m, n, letters, len(letters) + m)
    ohit, odst = oracle.pair_counts(a, ord("X"))
    for _ in range(2):
        ctx.upload(a, ord("X"))
        hit, dst = ctx.pair_counts()
        assert np.array_equal(hit, ohit) and np.array_equal(dst, odst)
    b = _alphabet_case(m, n, b"ACGT", 5)
    ctx.upload(b, ord("X"))
    hit, dst = ctx.pair_counts()
    bhit, bdst = oracle.pair_counts(b, ord("X"))
    assert np.array_equal(hit, bhit) and np.array_equal(dst, bdst)


def _sim_parity(ctx, a, indet=ord("X")):
    ctx.upload(a, indet)
    og, _, _, _ = oracle.gaps(a)
    ohit, odst = oracle.pair_counts(a, indet)
    ow = oracle.weights(ohit, odst)
    vhash, dist = oracle.aa_matrix()
    omdk, oq = oracle.similarity(a, ow, og, vhash, dist, indet)
    mdk, q = ctx.similarity(vhash, dist)
    assert np.array_equal(bits(q), bits(oq)), "similarity quotient must be bit-exact (reference order)"
    assert np.array_equal(bits(mdk), bits(omdk))


# The similarity kernel (binade-exact, per-lane grids) in its two instantiations -- 32-bit byte offsets of the W rows in
# the lists (default up to 32768 rows) and row indices multiplied out on the scalar unit (MSA_LG_BIG=1 forces it at
# any size) -- and the plain sequential kernel (MSA_SIM_KERNEL=seq: one lane per column, the reference's two loops).
# lg-rounds: the similarity kernel in several launches of one round each (what it does by itself from ~3000 rows on, two
# rounds per launch), the columns' state passed through memory
# lg-split-S: S waves of a workgroup share one column, each on a segment of every round's partner list (what the launcher
# picks by itself for tall alignments: fewer columns than wave slots); the increment pairs of the segments compose exactly
# (the default path of a small alignment is the compact pipeline -- three launches -- with the flat similarity kernel up to 128
# rows and the wave-per-column kernel from there to 512; MSA_COMPACT=0 is the ordinary launch sequence at every size; MSA_FLAT_U:
# terms per lane and scan of the flat kernel, by itself 8 below 112 rows and 16 from there on)
KERNELS = [dict(), dict(MSA_COMPACT="0"), dict(MSA_FLAT_MAX_M="0"), dict(MSA_FLAT_MAX_M="512"), dict(MSA_FLAT_MAX_M="512", MSA_FLAT_U="4"),
           dict(MSA_FLAT_MAX_M="512", MSA_FLAT_U="8"), dict(MSA_FLAT_MAX_M="512", MSA_FLAT_U="16"), dict(MSA_LG_BIG="1"), dict(MSA_LG_ROUNDS="1"),
           dict(MSA_LG_SPLIT="2"), dict(MSA_LG_SPLIT="4", MSA_LG_ROUNDS="1"), dict(MSA_LG_SPLIT="8"), dict(MSA_LG_SPLIT="16", MSA_LG_BIG="1"),
           dict(MSA_COMPACT="0", MSA_LISTS_FUSED="0"), dict(MSA_COMPACT="0", MSA_FRONT_CW="64", MSA_PAIR_TI="8"),
           dict(MSA_FRONT_CW="32", MSA_FRONT_NT="256", MSA_FRONT_FROM_M="2", MSA_PAIR_TI="16"),
           dict(MSA_COMPACT="0", MSA_LG_HALVES="2", MSA_LG_ROUNDS="2"), dict(MSA_COMPACT="0", MSA_LG_HALVES="2", MSA_LG_ROUNDS="3", MSA_LG_SPLIT="4"),
           # (round 6, late) a forced split up to twelve now runs as loop waves + a service wave (similarity_lg_pipe_body); MSA_LG_PIPE=0: the
           # barrier scheme of rounds 4 - 6
           dict(MSA_LG_SPLIT="2", MSA_LG_PIPE="0"), dict(MSA_LG_SPLIT="8", MSA_LG_PIPE="0", MSA_LG_ROUNDS="2"), dict(MSA_LG_SPLIT="3", MSA_LG_ROUNDS="0"),
           dict(MSA_LG_SPLIT="7", MSA_LG_ROUNDS="2"), dict(MSA_LG_SPLIT="12", MSA_LG_BIG="1"), dict(MSA_LG_SPLIT="5", MSA_LG_PIPE_K="3"),
           dict(MSA_LG_SPLIT="2", MSA_LG_PIPE="3", MSA_LG_PIPE_K="8", MSA_LG_ROUNDS="1"),
           # (round 6, late) the XCD-per-segment kernel of tall alignments forced at any size: 16 / 8 / 32 loop waves per column, a launch per
           # two rounds; and the pass giving up at once (the gated barrier-scheme launches behind it do the work)
           dict(MSA_COMPACT="0", MSA_LG_XSEG="2"), dict(MSA_COMPACT="0", MSA_LG_XSEG="2", MSA_LG_XSEG_KX="1", MSA_LG_ROUNDS="2"),
           dict(MSA_COMPACT="0", MSA_LG_XSEG="2", MSA_LG_XSEG_KX="4", MSA_LG_BIG="1"), dict(MSA_COMPACT="0", MSA_LG_XSEG="3"),
           dict(MSA_COMPACT="0", MSA_LG_XSEG="2", MSA_LG_XSEG_KX="8", MSA_LG_ROUNDS="3"),  # (four segments per column: two XCDs share one)
           dict(MSA_COMPACT="0", MSA_LG_HALVES="2", MSA_LG_ROUNDS="3", MSA_LG_SPLIT="4", MSA_LG_PIPE="0"),
           dict(MSA_SIM_KERNEL="seq")]
KERNEL_IDS = ["default", "lg", "compact-lg", "compact-flat-512", "compact-flat-512-u4", "compact-flat-512-u8", "compact-flat-512-u16", "lg-big", "lg-rounds", "lg-split-2", "lg-split-4-rounds", "lg-split-8",
              "lg-split-16-big", "lg-lists-in-two-passes", "lg-round5-front-pairs", "narrow-front-32-pairs-16", "lg-halves", "lg-halves-split-4",
              "lg-barrier-split-2", "lg-barrier-split-8-rounds", "lg-pipe-3-one-launch", "lg-pipe-7-rounds", "lg-pipe-12-big", "lg-pipe-5-subrounds-3",
              "lg-pipe-2-lockstep-subrounds-8", "xseg-16", "xseg-8-rounds-2", "xseg-32-big", "xseg-gives-up", "xseg-4-rounds-3", "lg-barrier-halves-split-4", "seq"]


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("shape", [(2, 70), (9, 33), (65, 64), (113, 200), (127, 1), (128, 300), (129, 70), (225, 96), (337, 130), (640, 257)])
def test_similarity_kernel_variants(ctx_with, kernel, shape):
    """Every similarity path against the oracle, at row counts on both sides of the round boundaries (64 rows per
    round) and with ragged numbers of columns."""
    m, n = shape
    _sim_parity(ctx_with(**kernel), synth_msa(m, n, 4242 + m))


@pytest.mark.parametrize("big", ["", "1"])
@pytest.mark.parametrize("r0", ["0", "1", "3", "8", "40", "64", "70", "200"])
def test_lane_grid_kernel_ordered_prefix(ctx_with, big, r0):
    """Per-lane grids: the number of rows evaluated in order before the first round (which then starts in the middle
    of a 64-row block, or several blocks in) must not matter."""
    _sim_parity(ctx_with(MSA_LG_BIG=big, MSA_LG_R0=r0), synth_msa(300, 150, 31))
    _sim_parity(ctx_with(MSA_LG_BIG=big, MSA_LG_R0=r0), _conserved_case(200, 70, 5))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_lane_grid_kernel_adversarial_predictions(ctx_with, kernel):
    ctx = ctx_with(**kernel)
    """Columns whose sums the per-lane predictor cannot foresee: the top half of the rows identical sequences (all
    their mutual weights zero, then a jump), a block of gaps in the middle of every column, residues sorted by row."""
    r = np.random.default_rng(5)
    a = synth_msa(400, 96, 123)
    a[:200, :] = a[0, :]                       # W = 0 among the first 200 rows
    a[120:260, 10:40] = ord("-")               # no valid row for 140 rows
    a[:, 50:60] = np.sort(a[:, 50:60], axis=0)  # residues (and gaps) sorted along the column
    a[r.random(a.shape) < 0.01] = ord("X")
    _sim_parity(ctx, np.ascontiguousarray(a))


@pytest.mark.parametrize("kernel", KERNELS[:-1], ids=KERNEL_IDS[:-1])
@pytest.mark.parametrize("shape", [(2016, 40), (2017, 33), (2100, 72), (4040, 20), (9000, 8)])
def test_similarity_many_rows(ctx_with, kernel, shape):
    """thousands of rows, a handful of columns (a single partial workgroup)"""
    m, n = shape
    _sim_parity(ctx_with(**kernel), synth_msa(m, n, 77 + m))


def test_pipelined_kernel_that_gives_up_is_an_error(ctx_with):
    """The waits between the waves of a split column are bounded: a wait that never ends (a protocol error) raises a word in LDS, the
    service wave writes NaN sums and the host turns them into an error -- loud, never a hang, never a silently wrong value.
    MSA_LG_PIPE=4 makes every workgroup give up at once."""
    a = synth_msa(2100, 40, 5)
    vhash, dist = oracle.aa_matrix()
    ctx = ctx_with(MSA_LG_PIPE="4")
    ctx.upload(a, ord("X"))
    with pytest.raises(RuntimeError):
        ctx.similarity(vhash, dist)
    ctx.close()
    ok = ctx_with()
    ok.upload(a, ord("X"))
    _sim_parity(ok, a)  # (the next context of the process is unharmed)


@pytest.mark.parametrize("shape", [(20000, 500), (40000, 300)])
def test_similarity_tall_alignments_split_columns(ctx_with, shape):
    """Pfam-style shapes: far fewer columns than the chip has wave slots, so the launcher gives every column a workgroup of
    8 / 16 waves (lg_split), each on a segment of every round's partner list.  Q and MDK of that default path against the
    CPU oracle on two 16-column slices, and against the plain sequential kernel (one lane per column, the reference's two
    loops) and the wave-per-column path, bit for bit, every column."""
    m, n = shape
    a = synth_msa(m, n, 20000 + n)
    vhash, dist = oracle.aa_matrix()
    ctx = ctx_with()
    ctx.upload(a, ord("X"))
    g = ctx.gaps()
    mdk, q = ctx.similarity(vhash, dist)
    paths = ctx.last_paths()  # (the default dispatch: a workgroup of eight waves per column, row indices beyond 32768 rows)
    # (round 6, late: wave w of every column on XCD w -- eight loop waves per column, sixteen while 17 x columns waves fit the chip)
    assert paths["sim_waves_per_column"] == (16 if m > 32768 else 8) and paths["sim_kernel"] == ("lg_big_xseg" if m > 32768 else "lg_xseg"), paths
    # the CPU oracle itself on two 16-column slices of the workgroup-per-column path at the sizes it was built for (W from the
    # device's pair pass -- the oracle's own would take minutes here; the device's W against the oracle on row slices)
    assert np.array_equal(g, (a == ord("-")).sum(axis=0))
    _, w = ctx.identities(want_ident=False)
    ctx.close()
    rows = np.r_[0:40, m // 2 - 20:m // 2 + 20, m - 40:m]
    ohit, odst = oracle.pair_counts(a[rows])
    assert np.array_equal(bits(w[np.ix_(rows, rows)]), bits(oracle.weights(ohit, odst)))
    for c0 in (0, n - 16):
        sl = slice(c0, c0 + 16)
        omdk, oq = oracle.similarity(np.ascontiguousarray(a[:, sl]), w, g[sl], vhash, dist)
        assert np.array_equal(bits(q[sl]), bits(oq)), f"Q of columns {c0}..{c0 + 15} differs from the oracle"
        assert np.array_equal(bits(mdk[sl]), bits(omdk)), f"MDK of columns {c0}..{c0 + 15} differs from the oracle"
    del w
    for env in (dict(MSA_SIM_KERNEL="seq"), dict(MSA_LG_SPLIT="1"), dict(MSA_LG_XSEG="0")):  # (... the sequential kernel, a wave per column, the barrier scheme)
        other = ctx_with(**env)
        other.upload(a, ord("X"))
        mdk2, q2 = other.similarity(vhash, dist)
        other.close()
        assert np.array_equal(bits(q), bits(q2)) and np.array_equal(bits(mdk), bits(mdk2)), env


def test_xseg_passes_of_several_contexts_at_once(ctx_with):
    """The XCD-per-segment kernel admits one pass at a time per process (its 9 or 17 waves per column must all be resident): contexts
    of other threads that ask meanwhile take the barrier scheme.  Four threads, a context each, the kernel forced at a size the
    sequential kernel checks in a moment; every thread's Q and MDK against the sequential kernel's, bit for bit, twice over."""
    import threading

    vhash, dist = oracle.aa_matrix()
    cases = [synth_msa(2300 + 64 * t, 40 + 7 * t, 900 + t) for t in range(4)]
    want = []
    for a in cases:
        seq = ctx_with(MSA_SIM_KERNEL="seq")
        seq.upload(a, ord("X"))
        want.append(seq.similarity(vhash, dist))
        seq.close()
    ctxs = [ctx_with(MSA_LG_XSEG="2") for _ in cases]
    bad, kinds = [], []

    def run(t):
        for _ in range(6):
            ctxs[t].upload(cases[t], ord("X"))
            mdk, q = ctxs[t].similarity(vhash, dist)
            kinds.append(ctxs[t].last_paths()["sim_kernel"])
            if not (np.array_equal(bits(q), bits(want[t][1])) and np.array_equal(bits(mdk), bits(want[t][0]))):
                bad.append(t)

    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for c in ctxs:
        c.close()
    assert not bad, f"threads {sorted(set(bad))} differ from the sequential kernel"
    assert "lg_xseg" in kinds, kinds  # (at least the first pass of all runs the kernel; the others take whatever is free)


def test_similarity_beyond_32768_rows(ctx_with):
    """40 000 sequences x 16 columns: past the 32-bit byte offsets of the W rows (m * ldw * 4 > 2^32), the lists hold row
    indices and the kernel multiplies them out.  Against the oracle -- with W taken from the device's pair pass, as in
    test_c3_full_size_properties: the oracle's own pair pass would take minutes here -- and against the plain
    sequential kernel, bit for bit.  (The pair counts themselves are checked against the oracle on row slices.)"""
    m, n = 40000, 16
    a = synth_msa(m, n, 40000)
    vhash, dist = oracle.aa_matrix()
    ctx = ctx_with()
    ctx.upload(a, ord("X"))
    g = ctx.gaps()
    assert np.array_equal(g, (a == ord("-")).sum(axis=0))
    mdk, q = ctx.similarity(vhash, dist)
    assert ctx.last_paths()["sim_kernel"] == "lg_big_xseg"
    seq = ctx_with(MSA_SIM_KERNEL="seq")
    seq.upload(a, ord("X"))
    mdk2, q2 = seq.similarity(vhash, dist)
    assert np.array_equal(bits(q), bits(q2)) and np.array_equal(bits(mdk), bits(mdk2))
    seq.close()
    _, w = ctx.identities(want_ident=False)
    rows = np.r_[0:40, 19990:20030, 39960:40000]
    ohit, odst = oracle.pair_counts(a[rows])
    assert np.array_equal(bits(w[np.ix_(rows, rows)]), bits(oracle.weights(ohit, odst)))
    omdk, oq = oracle.similarity(a, w, g, vhash, dist)
    assert np.array_equal(bits(q), bits(oq)) and np.array_equal(bits(mdk), bits(omdk))


def _conserved_case(m, n, seed):
    """Columns that are fully conserved (numerator stays zero), nearly conserved, all gaps but two rows, and
    identical sequences (weights zero): the sums that never leave zero or leave it late."""
    a = synth_msa(m, n, seed)
    r = np.random.default_rng(seed)
    a[:, 0] = ord("A")
    a[:, 1] = ord("A")
    a[m // 2, 1] = ord("W")
    a[:, 2] = ord("-")
    a[m - 2:, 2] = ord("K")
    a[:, 3] = ord("-")
    a[0, 3] = ord("K")
    a[m - 1, 3] = ord("R")
    a[1, :] = a[0, :]
    a[m - 1, 5:] = a[m - 2, 5:]
    a[:, 4] = np.where(r.random(m) < 0.97, ord("-"), ord("C"))
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("shape", [(70, 40), (200, 70), (513, 66)])
def test_similarity_zero_and_late_sums(ctx_with, kernel, shape):
    m, n = shape
    _sim_parity(ctx_with(**kernel), _conserved_case(m, n, 700 + m))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("letters", ["ACGT", "ABCDEFGHIKLMNOPQRSTUVWYZ", "ABCDEFGHIJKLMNOPQRSTUVWYZ"])
def test_similarity_alphabet_sizes(ctx_with, kernel, letters):
    """Custom matrices over 4, 24 and 25 letters (indetermination X): the per-wave LDS tables hold alphabet + 1 rows."""
    r = np.random.default_rng(len(letters))
    npos = len(letters)
    sim = r.integers(-4, 9, (npos, npos)).astype(np.float32)
    sim = (sim + sim.T) / 2
    matrix = oracle.make_matrix(sim, letters)
    alpha = np.frombuffer(letters.encode(), dtype=np.uint8)
    a = alpha[r.integers(0, npos, (190, 150))].copy()
    a[r.random(a.shape) < 0.2] = ord("-")
    a[r.random(a.shape) < 0.02] = ord("X")
    a = np.ascontiguousarray(a)
    ctx = ctx_with(**kernel)
    ctx.upload(a, ord("X"))
    og, _, _, _ = oracle.gaps(a)
    ohit, odst = oracle.pair_counts(a, ord("X"))
    omdk, oq = oracle.similarity(a, oracle.weights(ohit, odst), og, *matrix, ord("X"))
    mdk, q = ctx.similarity(*matrix)
    assert np.array_equal(bits(q), bits(oq))
    assert np.array_equal(bits(mdk), bits(omdk))


@pytest.mark.parametrize("seed", [21, 22])
def test_similarity_kernels_agree_on_random_shapes(seed):
    """tools/cross_check.py: the binade-exact kernel (both list formats) against the plain sequential kernel on random
    shapes and compositions (conserved, sorted, gap blocks, m = 2 .. 3000): independent implementations of one
    bit-exact statistic must agree bit for bit."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "cross_check.py"), "60", str(seed)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def _random_case(seed):
    """Small random alignments with extreme compositions: very gappy / fully gapped rows and columns,
    identical rows, indeterminations, lower case."""
    r = np.random.default_rng(seed)
    m = int(r.integers(2, 48))
    n = int(r.integers(1, 140))
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    a = alpha[r.integers(0, 20, (m, n))].copy()
    gap_rate = float(r.choice([0.0, 0.1, 0.5, 0.9]))
    a[r.random((m, n)) < gap_rate] = ord("-")
    a[r.random((m, n)) < 0.03] = ord("X")
    if r.random() < 0.5:
        a[int(r.integers(0, m)), :] = ord("-")                 # a sequence of gaps only
    if r.random() < 0.5:
        a[:, int(r.integers(0, n))] = ord("-")                 # an all-gap column
    if r.random() < 0.5 and m > 2:
        a[int(r.integers(1, m)), :] = a[0, :]                  # duplicated sequence
    if r.random() < 0.3:
        low = r.random((m, n)) < 0.2
        a[low & (a >= 65) & (a <= 90)] += 32                   # lower-case residues
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("seed", range(40))
def test_random_small_alignments(ctx, seed):
    all_stats(ctx, _random_case(9000 + seed))


@pytest.mark.parametrize("degenerate", [False, True])
def test_nucleotide_statistics(ctx, degenerate):
    """DNA alignments: indetermination 'N', the built-in nucleotide matrices."""
    r = np.random.default_rng(17 + degenerate)
    alpha = np.frombuffer(b"ACGTRYKM" if degenerate else b"ACGT", dtype=np.uint8)
    a = alpha[r.integers(0, len(alpha), (70, 210))].copy()
    a[r.random(a.shape) < 0.25] = ord("-")
    a[r.random(a.shape) < 0.03] = ord("N")
    err = all_stats(ctx, np.ascontiguousarray(a), indet=ord("N"), matrix=oracle.nt_matrix(degenerate))
    assert err is None


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_wide_alignment_many_workgroups(ctx_with, kernel):
    """Many more columns than wave slots: several waves of workgroups."""
    _sim_parity(ctx_with(**kernel), synth_msa(60, 40000, 4321))


def test_more_columns_than_a_grid_dimension(ctx):
    """trimAl has no column limit: an alignment with more 64-column groups than a launch's y / z dimensions hold
    (65 535 x 64 columns) -- every statistic and a whole trim against the oracle."""
    from pytrimal_amd import Alignment, AutomaticTrimmer

    n = 65535 * 64 + 257
    a = synth_msa(6, n, 99)
    all_stats(ctx, a)
    ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
    out = AutomaticTrimmer("strict", platform="hip").trim(ali)
    res, seq, _ = oracle.trim(a, method="strict")
    assert np.array_equal(np.asarray(out.residues_mask, dtype=bool), res.astype(bool))
    assert np.array_equal(np.asarray(out.sequences_mask, dtype=bool), seq.astype(bool))


def test_rows_are_the_callers_again_when_a_trim_returns():
    """`msa_upload_packed_async` + `msa_trim`: when the trim returns the copy is complete, also when the trim itself had
    nothing to compute (no threshold set: every column and sequence stays) -- the caller overwrites its rows right
    behind the call and the context must still hold what was uploaded."""
    c = _lib.Context(0)
    try:
        a = synth_msa(4000, 10000, 77)
        want = (a == ord("-")).sum(axis=0).astype(np.int32)
        nothing = _lib.TrimParams(_lib.METHOD_CODES[None], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, None, None, 0)
        for _ in range(3):
            rows = a.copy()
            c.upload(rows, ord("X"), pin=True, wait=False)
            keep_res, keep_seq, _ = c.trim(nothing)
            rows[:] = ord("-")
            assert keep_res.all() and keep_seq.all()
            assert np.array_equal(c.gaps(), want)
            c.upload(a.copy(), ord("X"), wait=False)  # an upload nobody waits for, released by the next one
            c.upload(a, ord("X"), wait=False)
            assert np.array_equal(c.gaps(), want)
    finally:
        c.close()


def _conserved_columns(m, n, seed):
    """Columns on both sides of the predictor's two gates: the commonest residue at 85 ... 100 % of the valid
    rows, 0 ... 20 rows of other residues (the list of a `SparseCol` holds 16) placed at the top, the bottom, one round apart or at
    random, with and without gaps -- beside ordinary columns."""
    r = np.random.default_rng(seed)
    a = synth_msa(m, n, seed)
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    for c in range(0, n, 2):  # (every other column stays as synth_msa made it)
        own = alpha[r.integers(0, 20)]
        a[:, c] = own
        others = int(r.choice([0, 1, 2, 3, 8, 15, 16, 17, 20, max(1, m // 12), max(1, m // 9), max(1, m // 7)]))
        place = r.integers(0, 4)
        if place == 0:
            rows = np.arange(others)                      # at the top: the sum starts large
        elif place == 1:
            rows = np.arange(m - others, m)               # at the bottom: nothing but zeros for a long time
        elif place == 2:
            rows = (np.arange(others) * 64 + 5) % m        # one per round
        else:
            rows = r.choice(m, size=min(others, m), replace=False)
        a[rows, c] = alpha[r.integers(0, 20, len(rows))]
        if r.random() < 0.5:
            a[r.random(m) < r.choice([0.05, 0.3, 0.6]), c] = ord("-")
        if r.random() < 0.2:
            a[r.integers(0, m), c] = ord("X")
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("kernel", [dict(MSA_COMPACT="0"), dict(MSA_COMPACT="0", MSA_LG_ROUNDS="1"), dict(MSA_LG_SPLIT="2", MSA_LG_ROUNDS="1"),
                                    dict(MSA_LG_SPLIT="5"), dict(MSA_LG_BIG="1"), dict()],
                         ids=["lg", "lg-rounds", "lg-split-2-rounds", "lg-split-5", "lg-big", "default"])
@pytest.mark.parametrize("shape", [(70, 90), (200, 120), (333, 64), (640, 48)])
def test_predictor_on_conserved_columns(ctx_with, kernel, shape):
    """The predictor's paths for conserved columns (exact counts of the rows behind, the non-zero products themselves, ordered
    rows from a handful of terms) in every instantiation that carries them -- a wave per column, one round per launch (the list
    travels through the state), several waves per column (every wave needs the list) -- against the oracle."""
    m, n = shape
    _sim_parity(ctx_with(**kernel), _conserved_columns(m, n, 1000 + m))


@pytest.mark.parametrize("kernel", [dict(), dict(MSA_LG_SPLIT="3")], ids=["default", "lg-split-3"])
def test_predictor_on_conserved_columns_many_rows(ctx_with, kernel):
    """... at 2100 rows: six rounds per launch, the columns' state (R, the list of a SparseCol) through memory; against the plain
    sequential kernel, bit for bit (the oracle would take a minute)."""
    a = _conserved_columns(2100, 40, 77)
    vhash, dist = oracle.aa_matrix()
    ctx = ctx_with(**kernel)
    ctx.upload(a, ord("X"))
    mdk, q = ctx.similarity(vhash, dist)
    ref = ctx_with(MSA_SIM_KERNEL="seq")
    ref.upload(a, ord("X"))
    mdk2, q2 = ref.similarity(vhash, dist)
    assert np.array_equal(bits(q), bits(q2)) and np.array_equal(bits(mdk), bits(mdk2))
    ohit, odst = oracle.pair_counts(a[:, :4])  # (... and the oracle itself on the first columns, with its own W of those columns' alignment)
    sub = np.ascontiguousarray(a[:, :4])
    small = ctx_with(**kernel)
    small.upload(sub, ord("X"))
    smdk, sq = small.similarity(vhash, dist)
    omdk, oq = oracle.similarity(sub, oracle.weights(ohit, odst), oracle.gaps(sub)[0], vhash, dist)
    assert np.array_equal(bits(sq), bits(oq)) and np.array_equal(bits(smdk), bits(omdk))
