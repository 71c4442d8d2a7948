"""What ships is the DEFAULT dispatch: which upload path, launch sequence, similarity kernel and pair kernel a given (m, n)
takes with no MSA_* variable set.  Every path is tested through its switch elsewhere (tests/test_gpu_parity.py KERNELS); here
the table below pins, shape by shape on both sides of every size threshold, the path the library takes by itself
(`msa_debug_last_paths`, include/msastat.h) AND compares the result of that very call with the CPU oracle.  An edit of a
threshold in pytrimal_amd/csrc fails this file until the table is updated with it.

Thresholds covered (DESIGN.md section 11 lists them): rows read in place up to 96 KB; flat similarity kernel up to 128 rows
(96 beyond 2560 columns: both waves of every column resident); compact pipeline up to 1024 rows, from 513 rows or 5121 columns on
with the columns dealt by weight; side stream from m^2 n = 2 * 10^9; a launch every six rounds from 1800 rows; a workgroup per column from 2048 rows when the columns leave
wave slots free; the columns of a multi-launch similarity pass as two staggered halves where that pays; front kernel alone for gap-only trims up to 1024 rows and 4 MB; sixteen rows i per tile of the pair pass from 513 rows,
two rows j per lane from 4096 rows.

The boundaries in columns are multiples of the device's compute units (10 and 20 per unit): the table is written for the MI355X's
256 and is skipped on a device (or a partition of one) that reports another count."""
import numpy as np
import pytest

import oracle
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa

pytestmark = pytest.mark.gpu

MSA_SWITCHES = ("MSA_SIM_KERNEL", "MSA_SIM_MODE", "MSA_LG_R0", "MSA_LG_BIG", "MSA_LG_ROUNDS", "MSA_LG_SPLIT", "MSA_MDK_HOST", "MSA_PIPELINE",
                "MSA_UPLOAD_DIRECT", "MSA_COMPACT", "MSA_FLAT_MAX_M", "MSA_FLAT_U", "MSA_ZEROCOPY_KB",
                "MSA_DEVICE_CLUSTERS", "MSA_TRACE", "MSA_FRONT_CW", "MSA_FRONT_NT", "MSA_FRONT_XCD", "MSA_FRONT_FROM_M", "MSA_PAIR_TI", "MSA_PAIR_K", "MSA_LISTS_FUSED", "MSA_LG_HALVES", "MSA_LG_PIPE", "MSA_LG_PIPE_K", "MSA_LG_XSEG", "MSA_LG_XSEG_KX")


def _compute_units():
    import torch

    return torch.cuda.get_device_properties(0).multi_processor_count


@pytest.fixture
def default_ctx(monkeypatch):
    if _compute_units() != 256:
        pytest.skip("the dispatch table is written for 256 compute units (its column boundaries are 10 and 20 per unit)")
    for name in MSA_SWITCHES:
        monkeypatch.delenv(name, raising=False)
    c = _lib.Context(0)
    yield c
    c.close()


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def params(method=None, **kw):
    vhash, dist = oracle.aa_matrix()
    vhash = np.ascontiguousarray(vhash, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    p = _lib.TrimParams(_lib.METHOD_CODES[method], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data,
                        dist.shape[0])
    for k, v in kw.items():
        setattr(p, k, v)
    p._keep = (vhash, dist)
    return p


# (m, n, expected entries of Context.last_paths() after upload + strict trim).  cus = 256: n <= 2560 / 5120 are "cus * 10 / 20".
LG1 = dict(sim_kernel="lg", sim_waves_per_column=1)
STRICT = [
    # --- flat kernel <-> wave-per-column kernel inside the compact pipeline
    (64, 300, dict(upload="in_place", pipeline="compact", sim_kernel="flat", sim_writes_mdk=1, sim_launches=1)),
    (64, 5000, dict(upload="packed", pipeline="compact", sim_kernel="flat")),
    (96, 2561, dict(upload="packed", pipeline="compact", sim_kernel="flat")),       # up to 96 rows: flat at any column count
    (97, 2560, dict(upload="linear", pipeline="compact", sim_kernel="flat")),       # 97 .. 128 rows: flat while n <= cus * 10
    (97, 2561, dict(upload="packed", pipeline="compact", sim_writes_mdk=1, sim_launches=1, **LG1)),
    (127, 600, dict(upload="in_place", pipeline="compact", sim_kernel="flat")),
    (128, 600, dict(upload="in_place", pipeline="compact", sim_kernel="flat")),
    (129, 600, dict(upload="in_place", pipeline="compact", sim_writes_mdk=1, **LG1)),
    (128, 2561, dict(pipeline="compact", sim_writes_mdk=1, **LG1)),
    # --- rows read in place up to 96 KB (m x the 64-byte padded row)
    (96, 1024, dict(upload="in_place", pipeline="compact", sim_kernel="flat")),
    (97, 1024, dict(upload="linear", pipeline="compact", sim_kernel="flat")),
    # --- compact pipeline: columns as they lie up to 512 rows and cus * 20 columns (every wave resident), dealt by weight beyond
    #     either; the ordinary pipeline from 1025 rows
    (512, 1000, dict(upload="packed", pipeline="compact", sim_writes_mdk=1, pair_kernel="pipe", **LG1)),
    (513, 1000, dict(upload="packed", pipeline="compact_sorted", sim_writes_mdk=1, sim_launches=1, pair_kernel="pipe16", **LG1)),
    (300, 5120, dict(upload="linear", pipeline="compact", sim_writes_mdk=1, **LG1)),
    (300, 5121, dict(upload="repitched", pipeline="compact_sorted", sim_writes_mdk=1, **LG1)),   # 1.5 MB of 5121-byte rows: one linear copy + a kernel
    (200, 5121, dict(upload="packed", pipeline="compact_sorted", sim_writes_mdk=1, **LG1)),      # (below a megabyte: packed pieces)
    (64, 20000, dict(pipeline="compact", sim_kernel="flat", sim_writes_mdk=1)),
    (1000, 4000, dict(pipeline="compact_sorted", sim_writes_mdk=1, sim_launches=1, pair_kernel="pipe16", pair_waves_per_tile=8, **LG1)),
    (1024, 2000, dict(pipeline="compact_sorted", sim_writes_mdk=1, **LG1)),
    (1024, 5121, dict(pipeline="compact_sorted", sim_writes_mdk=1, **LG1)),
    # --- side stream from m^2 n = 2e9
    (1025, 1903, dict(pipeline="one_stream", **LG1)),
    (1025, 1904, dict(pipeline="two_streams", **LG1)),
    (1025, 2000, dict(pipeline="two_streams", pair_kernel="pipe16", pair_waves_per_tile=4, **LG1)),
    # --- a launch every six rounds from 1800 rows (29 rounds: one launch, or five)
    (1799, 1800, dict(pipeline="two_streams", sim_launches=1, **LG1)),
    (1800, 1800, dict(pipeline="two_streams", sim_launches=5, **LG1)),
    # --- few columns, up to 9000 rows (round 6): S loop waves + a service wave per column, no barrier -- 12 up to a column per compute
    #     unit (and a quarter), 7 up to two (and a half), 3 up to six (evaluated columns: ~92 % of n here), none with less than 256 rows; one launch up to 3600 rows
    (1799, 900, dict(pipeline="two_streams", sim_kernel="lg_pipe", sim_waves_per_column=3, sim_launches=1)),
    (1500, 300, dict(sim_kernel="lg_pipe", sim_waves_per_column=3, sim_launches=1)),
    (2047, 1000, dict(pipeline="two_streams", sim_kernel="lg_pipe", sim_waves_per_column=3, sim_launches=1)),
    (2048, 500, dict(pipeline="two_streams", sim_kernel="lg_pipe", sim_waves_per_column=7, sim_launches=1)),
    (2048, 1500, dict(pipeline="two_streams", sim_kernel="lg_pipe", sim_waves_per_column=3, sim_launches=1)),
    (2048, 1800, dict(pipeline="two_streams", sim_kernel="lg", sim_waves_per_column=2, sim_launches=12)),   # (more than six columns per compute unit: the barrier scheme's two, as two staggered halves)
    (3328, 200, dict(sim_kernel="lg_pipe", sim_waves_per_column=12, sim_launches=1)),
    (3700, 500, dict(sim_kernel="lg_pipe", sim_waves_per_column=7, sim_launches=10)),
    (3700, 1000, dict(sim_kernel="lg_pipe", sim_waves_per_column=3, sim_launches=21)),   # (... as two staggered halves: 10 + 11)
    # --- beyond 9000 rows with up to 568 columns: wave w of every column on XCD w (round 6, late) -- eight loop waves per column, sixteen while
    #     17 x columns waves fit the chip; with more columns a workgroup per column and a barrier per round
    (9216, 64, dict(sim_kernel="lg_pipe", sim_waves_per_column=12)),    # (a compute unit per column: the pipelined kernel up to 11000 rows)
    (11100, 64, dict(sim_kernel="lg_xseg", sim_waves_per_column=16)),
    (9216, 340, dict(sim_kernel="lg_xseg", sim_waves_per_column=8)),
    (9216, 800, dict(sim_kernel="lg_xseg", sim_waves_per_column=4)),   # (700 .. ~970 columns up to 16000 rows: four segments, two XCDs share one)
    # --- the columns as two staggered halves on two streams (7 + 8 launches at 2600 rows): a wave per column when the columns outnumber the
    #     wave slots (2560 ... 4608 rows)
    (2048, 6000, dict(pipeline="two_streams", sim_kernel="lg", sim_waves_per_column=1, sim_launches=6)),
    (2600, 5400, dict(pipeline="two_streams", sim_kernel="lg", sim_waves_per_column=1, sim_launches=7)),
    (2600, 5700, dict(pipeline="two_streams", sim_kernel="lg", sim_waves_per_column=1, sim_launches=15)),
    # --- pair pass: eight rows i per tile up to 512 rows, sixteen from 513 on (K waves per tile while tiles x K <= 10240); one row j
    #     per lane below 4096 rows (m_pad / 128 * ceil(m / 8) / 2 < 8192), two from there on
    (2000, 2100, dict(pair_kernel="pipe16", pair_waves_per_tile=4, sim_kernel="lg")),
    (4088, 64, dict(pair_kernel="pipe16", pair_waves_per_tile=1, sim_kernel="lg_pipe")),
    (4096, 64, dict(pair_kernel="two_rows", pair_waves_per_tile=1, sim_kernel="lg_pipe")),
]


def expect(paths, want, what):
    got = {k: paths[k] for k in want}
    assert got == want, f"{what}: default dispatch took {paths}"


@pytest.mark.parametrize("m,n,want", STRICT, ids=[f"{m}x{n}" for m, n, _ in STRICT])
def test_default_dispatch_of_a_strict_trim(default_ctx, m, n, want):
    a = synth_msa(m, n, 9000 + m + n)
    ctx = default_ctx
    ctx.upload(a, ord("X"))
    res, seq, info = ctx.trim(params("strict"))
    expect(ctx.last_paths(), want, f"strict trim of {m} x {n}")
    ores, oseq, oinfo = oracle.trim(a, method="strict")
    assert np.array_equal(res, ores) and np.array_equal(seq, oseq), "masks differ from the oracle"
    assert info.gap_cut == oinfo.gap_cut and bits(info.sim_cut) == bits(oinfo.sim_cut)


# (m, n, pipeline of a gappyout trim): the front kernel alone up to 1024 rows and 4 MB of rows
GAPS = [(1024, 1000, "compact_gaps"), (1025, 1000, "none"), (500, 8192, "compact_gaps"), (520, 8192, "none"), (46, 1181, "compact_gaps")]


@pytest.mark.parametrize("m,n,pipe", GAPS, ids=[f"{m}x{n}" for m, n, _ in GAPS])
def test_default_dispatch_of_a_gap_only_trim(default_ctx, m, n, pipe):
    a = synth_msa(m, n, 9100 + m)
    ctx = default_ctx
    ctx.upload(a, ord("X"))
    res, seq, info = ctx.trim(params("gappyout"))
    expect(ctx.last_paths(), dict(pipeline=pipe, sim_kernel="none", pair_kernel="none"), f"gappyout trim of {m} x {n}")
    ores, oseq, _ = oracle.trim(a, method="gappyout")
    assert np.array_equal(res, ores) and np.array_equal(seq, oseq)


# msa_similarity by itself: the compact pipeline where it applies, else its own serial launch sequence
SIMILARITY = [(100, 700, dict(pipeline="compact", sim_kernel="flat")), (400, 700, dict(pipeline="compact", sim_writes_mdk=1, **LG1)),
              (513, 700, dict(pipeline="compact_sorted", sim_writes_mdk=1, **LG1)), (1025, 700, dict(pipeline="serial", sim_writes_mdk=0, sim_kernel="lg_pipe", sim_waves_per_column=3)),
              (400, 5121, dict(pipeline="compact_sorted", sim_writes_mdk=1, **LG1))]


@pytest.mark.parametrize("m,n,want", SIMILARITY, ids=[f"{m}x{n}" for m, n, _ in SIMILARITY])
def test_default_dispatch_of_msa_similarity(default_ctx, m, n, want):
    a = synth_msa(m, n, 9200 + m)
    ctx = default_ctx
    ctx.upload(a, ord("X"))
    vhash, dist = oracle.aa_matrix()
    mdk, q = ctx.similarity(vhash, dist)
    expect(ctx.last_paths(), want, f"msa_similarity of {m} x {n}")
    og = oracle.gaps(a)[0]
    ohit, odst = oracle.pair_counts(a)
    omdk, oq = oracle.similarity(a, oracle.weights(ohit, odst), og, vhash, dist)
    assert np.array_equal(bits(q), bits(oq)) and np.array_equal(bits(mdk), bits(omdk))


def test_pitched_upload_of_aligned_rows(default_ctx):
    """rows of a multiple of 16 bytes at a 16-byte aligned address go up in ONE pitched copy straight from where they lie"""
    m, n = 700, 1008
    a = synth_msa(m, n, 77)
    store = np.empty(m * n + 64, dtype=np.uint8)
    off = (-store.ctypes.data) % 64
    b = store[off:off + m * n].reshape(m, n)
    b[:] = a
    ctx = default_ctx
    ctx.upload(b, ord("X"))
    assert np.array_equal(ctx.gaps(), oracle.gaps(a)[0])
    assert ctx.last_paths()["upload"] == "pitched"
    c = store[off + 1:off + 1 + m * n].reshape(m, n)  # the same rows at an odd address: packed pieces
    c[:] = a
    ctx.upload(c, ord("X"))
    assert np.array_equal(ctx.gaps(), oracle.gaps(a)[0])
    assert ctx.last_paths()["upload"] == "packed"


def test_linear_copy_and_repitch_of_odd_sized_rows(default_ctx):
    """a contiguous matrix of a megabyte or more whose rows are no multiple of 16 bytes (the BASELINE's 5000 x 5000) goes up in ONE
    linear copy and a kernel lays the rows out at the device pitch -- also through a view a few columns narrower than its rows;
    a view much narrower than its rows, and anything below a megabyte, takes the packed pieces"""
    m, n = 1100, 1001
    a = synth_msa(m, n, 78)
    ctx = default_ctx
    ctx.upload(a, ord("X"))
    assert ctx.last_paths()["upload"] == "repitched"
    assert np.array_equal(ctx.gaps(), oracle.gaps(a)[0])
    res, seq, _ = ctx.trim(params("strict"))
    ores, oseq, _ = oracle.trim(a, method="strict")
    assert np.array_equal(res, ores) and np.array_equal(seq, oseq)
    wide = np.full((m, n + 5), ord("A"), dtype=np.uint8)
    wide[:, :n] = a

    def upload_view(cols):  # (the C ABI itself: rows `n + 5` bytes apart, `cols` of them used -- Context.upload would copy the view)
        _lib.check(ctx.lib, ctx.h, ctx.lib.msa_upload_packed(ctx.h, _lib.ptr(wide), m, cols, n + 5, ord("X")))
        ctx.shape = (m, cols)

    upload_view(n)
    assert ctx.last_paths()["upload"] == "repitched"
    assert np.array_equal(ctx.gaps(), oracle.gaps(a)[0])
    upload_view(101)  # a tenth of every row: packed pieces
    assert ctx.last_paths()["upload"] == "packed"
    assert np.array_equal(ctx.gaps(), oracle.gaps(np.ascontiguousarray(a[:, :101]))[0])
    big = synth_msa(5000, 5000, 1004)[:, :4999]   # (C4's shape, one column short: rows of 4999 bytes, 5000 apart)
    ctx.upload(big, ord("X"))
    assert ctx.last_paths()["upload"] == "repitched"
    assert np.array_equal(ctx.gaps(), (big == ord("-")).sum(axis=0))


def test_windowed_similarity_after_a_compact_trim_on_the_same_upload(default_ctx):
    """A compact trim leaves W on the device but not the mean weights of the ordinary kernel's predictor (its own kernel divides
    the pair pass's row sums): a later msa_similarity with a gap window on the SAME upload runs the ordinary kernel on that W
    and must compute them first (round 4's advisor: it read an uninitialised buffer)."""
    m, n = 300, 900
    a = synth_msa(m, n, 4711)
    ctx = default_ctx
    ctx.upload(a, ord("X"))
    res, seq, _ = ctx.trim(params("strict"))
    assert ctx.last_paths()["pipeline"] == "compact"
    ores, oseq, _ = oracle.trim(a, method="strict")
    assert np.array_equal(res, ores) and np.array_equal(seq, oseq)
    vhash, dist = oracle.aa_matrix()
    og = oracle.gaps(a)[0]
    gw = oracle.gaps_window(og, 3)
    for _ in range(2):
        mdk, q = ctx.similarity(vhash, dist, gaps_windowed=gw)
        paths = ctx.last_paths()
        assert paths["pipeline"] == "serial" and paths["sim_kernel"] == "lg" and paths["pair_kernel"] == "none", paths  # (W reused: no second pair pass)
        ohit, odst = oracle.pair_counts(a)
        omdk, oq = oracle.similarity(a, oracle.weights(ohit, odst), gw, vhash, dist)
        assert np.array_equal(bits(q), bits(oq)) and np.array_equal(bits(mdk), bits(omdk))
