"""CPU oracle for the MSA statistics path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes/numpy front-end of ``oracle/msa_oracle.c`` (see that file's header for provenance
and pinning).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; ``pytrimal_amd`` never does.

The similarity-matrix tables below restate what the reference gets from its
``scoring-matrices`` dependency (``/root/reference/src/pytrimal/_trimal.pyx:1879-1885``:
BLOSUM62 re-ordered to trimAl's ``aminoAcidResidues``) and from trimAl's built-in
nucleotide matrices (``_trimal.pyx:1887-1911``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, E_WINDOW_TOO_BIG, E_INCORRECT_SYMBOL, E_UNDEFINED_SYMBOL, E_NOT_IMPLEMENTED, E_NOMEM = range(6)
GAPPYOUT, STRICT = 1, 2

METHODS = {None: 0, "strict": 1, "strictplus": 2, "gappyout": 3, "nogaps": 4, "noallgaps": 5,
           "automated1": 6, "automated2": 7, "noduplicateseqs": 8}

AA_ALPHABET = "ARNDCQEGHILKMFPSTWYV"  # trimAl aminoAcidResidues (tests/test_similarity_matrix.py:27-33)
NT_ALPHABET = "ACGTU"
NT_DEG_ALPHABET = "ACGTURYKMSWBDHV"

# standard BLOSUM62 integers, alphabet order AA_ALPHABET
BLOSUM62 = np.array([
    [4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0],
    [-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3],
    [-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3],
    [-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3],
    [0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1],
    [-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2],
    [-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2],
    [0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3],
    [-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3],
    [-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3],
    [-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1],
    [-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2],
    [-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1],
    [-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1],
    [-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2],
    [1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2],
    [0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0],
    [-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3],
    [-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1],
    [0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4],
], dtype=np.float32)

# trimAl defaultNTSimMatrix [R]: identity with T == U.  Pinned only by
# nt().similarity('A','A') == 1, ('A','T') == 0 (_trimal.pyx:2005-2009).
NT_MATRIX = np.array([
    [1, 0, 0, 0, 0],
    [0, 1, 0, 0, 0],
    [0, 0, 1, 0, 0],
    [0, 0, 0, 1, 1],
    [0, 0, 0, 1, 1],
], dtype=np.float32)


def _deg_matrix():
    """trimAl defaultNTDegeneratedSimMatrix [R; values not in the reference tree]: identity on the
    diagonal, otherwise IUPAC base-set overlap / (|x| * |y|) / 2.  This is the reading that
    reproduces the one pin, nt(True).distance('A','T') ~ 1.5184 (_trimal.pyx:2042-2046)."""
    sets = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "U", "R": "AG", "Y": "CT", "K": "GT",
            "M": "AC", "S": "CG", "W": "AT", "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG"}
    k = len(NT_DEG_ALPHABET)
    out = np.zeros((k, k), dtype=np.float32)
    for i, x in enumerate(NT_DEG_ALPHABET):
        for j, y in enumerate(NT_DEG_ALPHABET):
            sx, sy = set(sets[x]), set(sets[y])
            out[i, j] = 1.0 if x == y else len(sx & sy) / float(len(sx) * len(sy)) / 2.0
    return out


def lib():
    """Load (building on first use) libmsa_oracle.so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # MSA_ORACLE_ASAN=1: the AddressSanitizer / UBSan build (run the CPU suite under LD_PRELOAD=libasan.so; GPU
    # sanitizers are not available on the pool, so the CPU restatement is what gets this check)
    name = "libmsa_oracle_asan.so" if os.environ.get("MSA_ORACLE_ASAN") else "libmsa_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "msa_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, name], check=True, capture_output=True)
    L = ctypes.CDLL(path)
    L.orc_gaps_cutpoint.restype = ctypes.c_double
    L.orc_sim_cutpoint.restype = ctypes.c_double
    L.orc_comb_simcut.restype = ctypes.c_float
    L.orc_cutpoint_clusters.restype = ctypes.c_float
    _LIB = L
    return L


def _p(arr):
    return arr.ctypes.data_as(ctypes.c_void_p) if arr is not None else None


def pack(rows):
    """List of equal-length bytes/str rows -> C-contiguous uint8 [m, n]."""
    if isinstance(rows, np.ndarray):
        return np.ascontiguousarray(rows, dtype=np.uint8)
    rows = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in rows]
    n = len(rows[0]) if rows else 0
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), n).copy()


class OracleError(Exception):
    def __init__(self, code, detail=None):
        super().__init__(f"oracle error {code} {detail}")
        self.code, self.detail = code, detail


# --- similarity matrices ------------------------------------------------------------------

def make_matrix(sim, alphabet):
    """(vhash[26] int32, dist[npos,npos] f32) from a similarity table, as
    SimilarityMatrix.__init__ (_trimal.pyx:1973-1997)."""
    sim = np.ascontiguousarray(sim, dtype=np.float32)
    npos = len(alphabet)
    assert sim.shape == (npos, npos)
    vhash = np.full(26, -1, dtype=np.int32)
    for i, ch in enumerate(alphabet):
        vhash[ord(ch) - 65] = i
    dist = np.zeros((npos, npos), dtype=np.float32)
    lib().orc_distmat(_p(sim), npos, _p(dist))
    return vhash, dist


def aa_matrix():
    return make_matrix(BLOSUM62, AA_ALPHABET)


def nt_matrix(degenerated=False):
    return make_matrix(_deg_matrix(), NT_DEG_ALPHABET) if degenerated else make_matrix(NT_MATRIX, NT_ALPHABET)


# --- statistics ---------------------------------------------------------------------------

def gaps(a):
    a = pack(a)
    m, n = a.shape
    g = np.zeros(n, dtype=np.int32)
    hist = np.zeros(m + 2, dtype=np.int32)
    mx = ctypes.c_int32(0)
    tot = ctypes.c_int64(0)
    lib().orc_gaps(_p(a), m, n, n, _p(g), _p(hist), ctypes.byref(mx), ctypes.byref(tot))
    return g, hist, mx.value, tot.value


def gaps_window(g, hw):
    g = np.ascontiguousarray(g, dtype=np.int32)
    out = np.zeros_like(g)
    rc = lib().orc_gaps_window(_p(g), len(g), int(hw), _p(out))
    if rc:
        raise OracleError(rc)
    return out


def window_f32(v, hw):
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros_like(v)
    rc = lib().orc_window_f32(_p(v), len(v), int(hw), _p(out))
    if rc:
        raise OracleError(rc)
    return out


def gaps_cutpoint(hist, m, n, base_line, gap_threshold):
    hist = np.ascontiguousarray(hist, dtype=np.int32)
    return lib().orc_gaps_cutpoint(_p(hist), m, n, ctypes.c_float(base_line), ctypes.c_float(gap_threshold))


def cutpoint_2nd_slope(hist, m, n, max_gaps):
    hist = np.ascontiguousarray(hist, dtype=np.int32)
    return lib().orc_gaps_cutpoint_2nd_slope(_p(hist), m, n, max_gaps)


_AVX2 = None


def lib_avx2():
    """The AVX2 flavour of the two pairwise passes (msa_oracle_avx2.c), or None when the host CPU has no AVX2.
    Same results as the scalar functions, bit for bit; it exists for the CPU baseline of bench.py."""
    global _AVX2
    if _AVX2 is None:
        path = os.path.join(_HERE, "libmsa_oracle_avx2.so")
        src = os.path.join(_HERE, "msa_oracle_avx2.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE, "libmsa_oracle_avx2.so"], check=True, capture_output=True)
        L = ctypes.CDLL(path)
        _AVX2 = L if L.orc_avx2_supported() else False
    return _AVX2 or None


def pair_counts(a, indet=ord("X"), avx2=False):
    a = pack(a)
    m, n = a.shape
    hit = np.zeros((m, m), dtype=np.uint32)
    dst = np.zeros((m, m), dtype=np.uint32)
    fn = lib_avx2().orc_pair_counts_avx2 if avx2 else lib().orc_pair_counts
    fn(_p(a), m, n, n, ctypes.c_uint8(indet), _p(hit), _p(dst))
    return hit, dst


def identities(hit, dst):
    m = hit.shape[0]
    out = np.zeros((m, m), dtype=np.float32)
    lib().orc_identities(_p(np.ascontiguousarray(hit)), _p(np.ascontiguousarray(dst)), m, _p(out))
    return out


def weights(hit, dst):
    m = hit.shape[0]
    out = np.zeros((m, m), dtype=np.float32)
    lib().orc_weights(_p(np.ascontiguousarray(hit)), _p(np.ascontiguousarray(dst)), m, _p(out))
    return out


def similarity(a, w, gaps_w, vhash, dist, indet=ord("X"), avx2=False):
    """-> (mdk f32[n], q f32[n]).  Raises OracleError(E_INCORRECT_SYMBOL / E_UNDEFINED_SYMBOL)."""
    a = pack(a)
    m, n = a.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    gw = None if gaps_w is None else np.ascontiguousarray(gaps_w, dtype=np.int32)
    vhash = np.ascontiguousarray(vhash, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    mdk = np.zeros(n, dtype=np.float32)
    q = np.zeros(n, dtype=np.float32)
    err = np.zeros(3, dtype=np.int32)
    fn = lib_avx2().orc_similarity_avx2 if avx2 else lib().orc_similarity
    rc = fn(_p(a), m, n, n, ctypes.c_uint8(indet), _p(w), _p(gw), _p(vhash), _p(dist), dist.shape[0], _p(mdk), _p(q),
            _p(err))
    if rc:
        raise OracleError(rc, tuple(int(x) for x in err))
    return mdk, q


def overlap(a, residue_overlap, indet=ord("X")):
    a = pack(a)
    m, n = a.shape
    out = np.zeros(m, dtype=np.float32)
    lib().orc_overlap(_p(a), m, n, n, ctypes.c_uint8(indet), ctypes.c_float(residue_overlap), _p(out))
    return out


def select_method(ident):
    ident = np.ascontiguousarray(ident, dtype=np.float32)
    avg = ctypes.c_float(0)
    mx = ctypes.c_float(0)
    r = lib().orc_select_method(_p(ident), ident.shape[0], ctypes.byref(avg), ctypes.byref(mx))
    return r, np.float32(avg.value), np.float32(mx.value)


def alignment_type(a):
    a = pack(a)
    m, n = a.shape
    return lib().orc_alignment_type(_p(a), m, n, n)


def indet_for(a):
    return ord("X") if (alignment_type(a) & 4) else ord("N")


# --- selection logic ----------------------------------------------------------------------

def clean_overpass(gw, cut, base_line):
    gw = np.ascontiguousarray(gw, dtype=np.int32)
    save = np.zeros(len(gw), dtype=np.int32)
    lib().orc_clean_overpass(_p(gw), len(gw), ctypes.c_double(cut), ctypes.c_float(base_line), _p(save))
    return save != -1


def clean_fallbehind(vw, cut, base_line):
    vw = np.ascontiguousarray(vw, dtype=np.float32)
    save = np.zeros(len(vw), dtype=np.int32)
    lib().orc_clean_fallbehind(_p(vw), len(vw), ctypes.c_float(cut), ctypes.c_float(base_line), _p(save))
    return save != -1


def clean_both(gw, vw, cut_g, cut_v, base_line):
    gw = np.ascontiguousarray(gw, dtype=np.int32)
    vw = np.ascontiguousarray(vw, dtype=np.float32)
    save = np.zeros(len(gw), dtype=np.int32)
    lib().orc_clean_both(_p(gw), _p(vw), len(gw), ctypes.c_double(cut_g), ctypes.c_float(cut_v),
                         ctypes.c_float(base_line), _p(save))
    return save != -1


def sim_cutpoint(mdkw, base_line, sim_threshold):
    mdkw = np.ascontiguousarray(mdkw, dtype=np.float32)
    return lib().orc_sim_cutpoint(_p(mdkw), len(mdkw), ctypes.c_float(base_line), ctypes.c_float(sim_threshold))


def comb_simcut(gw, mdkw, gap_cut):
    gw = np.ascontiguousarray(gw, dtype=np.int32)
    mdkw = np.ascontiguousarray(mdkw, dtype=np.float32)
    return np.float32(lib().orc_comb_simcut(_p(gw), _p(mdkw), len(gw), int(gap_cut)))


def clean_strict(gw, mdkw, gap_cut, sim_cut, variable):
    gw = np.ascontiguousarray(gw, dtype=np.int32)
    mdkw = np.ascontiguousarray(mdkw, dtype=np.float32)
    save = np.zeros(len(gw), dtype=np.int32)
    lib().orc_clean_strict(_p(gw), _p(mdkw), len(gw), int(gap_cut), ctypes.c_float(sim_cut), int(variable), _p(save))
    return save != -1


def representatives(a, ident, max_ident, sort_mode=0):
    a = pack(a)
    m, n = a.shape
    ident = np.ascontiguousarray(ident, dtype=np.float32)
    save = np.zeros(m, dtype=np.int32)
    nc = lib().orc_representatives(_p(a), m, n, n, _p(ident), ctypes.c_float(max_ident), int(sort_mode), _p(save))
    return save != -1, nc


def cutpoint_clusters(a, ident, k, sort_mode=0):
    a = pack(a)
    m, n = a.shape
    ident = np.ascontiguousarray(ident, dtype=np.float32)
    return np.float32(lib().orc_cutpoint_clusters(_p(a), m, n, n, _p(ident), int(k), int(sort_mode)))


class Params(ctypes.Structure):
    _fields_ = [
        ("method", ctypes.c_int32),
        ("gap_threshold", ctypes.c_float),
        ("gap_absolute_threshold", ctypes.c_int32),
        ("similarity_threshold", ctypes.c_float),
        ("conservation_percentage", ctypes.c_float),
        ("window", ctypes.c_int32),
        ("gap_window", ctypes.c_int32),
        ("similarity_window", ctypes.c_int32),
        ("residue_overlap", ctypes.c_float),
        ("sequence_overlap", ctypes.c_float),
        ("clusters", ctypes.c_int32),
        ("max_identity", ctypes.c_float),
        ("indet", ctypes.c_uint8),
        ("sort_mode", ctypes.c_int32),
    ]


class Info(ctypes.Structure):
    _fields_ = [
        ("selected", ctypes.c_int32),
        ("avg_seq", ctypes.c_float),
        ("max_seq", ctypes.c_float),
        ("gap_cut", ctypes.c_int32),
        ("sim_cut", ctypes.c_float),
        ("err", ctypes.c_int32 * 3),
    ]


def trim(a, method=None, gap_threshold=None, gap_absolute_threshold=None, similarity_threshold=None,
         conservation_percentage=None, window=None, gap_window=None, similarity_window=None,
         residue_overlap=None, sequence_overlap=None, clusters=None, identity_threshold=None,
         matrix=None, indet=None, sort_mode=0):
    """Whole `BaseTrimmer.trim` equivalent (_trimal.pyx:1291-1365) with the keyword meaning of
    the four trimmer constructors.  -> (residues_mask bool[n], sequences_mask bool[m], Info)."""
    a = pack(a)
    m, n = a.shape
    if indet is None:
        indet = indet_for(a) if m and n else ord("X")
    if matrix is None:
        t = alignment_type(a) if m and n else 4
        matrix = aa_matrix() if (t & 4) else nt_matrix(bool(t & 8))
    vhash, dist = matrix
    vhash = np.ascontiguousarray(vhash, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)

    def opt(v, d=-1):
        return d if v is None else v

    p = Params(METHODS[method],
               -1.0 if gap_threshold is None else float(np.float32(1) - np.float32(gap_threshold)),
               opt(gap_absolute_threshold), opt(similarity_threshold, -1.0),
               opt(conservation_percentage, -1.0), opt(window), opt(gap_window), opt(similarity_window),
               opt(residue_overlap, -1.0), opt(sequence_overlap, -1.0), opt(clusters),
               opt(identity_threshold, -1.0), indet, sort_mode)
    save_res = np.zeros(n, dtype=np.int32)
    save_seq = np.zeros(m, dtype=np.int32)
    info = Info()
    rc = lib().orc_trim(_p(a), m, n, n, ctypes.byref(p), _p(vhash), _p(dist), dist.shape[0],
                        _p(save_res), _p(save_seq), ctypes.byref(info))
    if rc:
        raise OracleError(rc, tuple(info.err))
    return save_res != -1, save_seq != -1, info


def terminal_only(a, residues_mask, sequences_mask, reading=2, gaps_w=None):
    """`TrimmedAlignment.terminal_only` (Cleaner::removeOnlyTerminal, _trimal.pyx:1144-1157; [R], see the C source):
    -> the new residues mask, or None when no column is free of gaps (upstream reports an error).  reading 2 (the
    product's): boundaries from the gap vector of the ORIGINAL alignment (`gaps_w`: the trim's windowed counts, or
    None = counted over all sequences); 0: gaps over the kept sequences; 1: first / last kept column."""
    a = pack(a)
    m, n = a.shape
    save_res = np.where(np.asarray(residues_mask, dtype=bool), np.arange(n), -1).astype(np.int32)
    save_seq = np.where(np.asarray(sequences_mask, dtype=bool), np.arange(m), -1).astype(np.int32)
    gw = None if gaps_w is None else np.ascontiguousarray(gaps_w, dtype=np.int32)
    ok = lib().orc_terminal_only(_p(a), m, n, n, _p(save_seq), _p(save_res), int(reading), _p(gw))
    return (save_res != -1) if ok else None


# --- tiny readers for the fixtures (tests only) ---------------------------------------------

def read_fasta(path):
    names, seqs = [], []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                names.append(line[1:].split()[0] if line[1:].split() else b"")
                seqs.append([])
            elif names and line:
                seqs[-1].append(line.replace(b" ", b""))
    return names, [b"".join(s) for s in seqs]


def read_clustal(path):
    names, seqs = [], {}
    with open(path, "rb") as f:
        first = True
        for line in f:
            line = line.rstrip(b"\r\n")
            if first:
                first = False
                continue
            if not line.strip() or line[:1] in (b" ", b"\t"):
                continue
            parts = line.split()
            if len(parts) < 2:
                continue
            if parts[0] not in seqs:
                names.append(parts[0])
                seqs[parts[0]] = []
            seqs[parts[0]].append(parts[1])
    return names, [b"".join(seqs[k]) for k in names]
