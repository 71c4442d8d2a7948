"""`Alignment` / `TrimmedAlignment`: the host-side containers of the trim path.

API-compatible with ``pytrimal.Alignment`` and ``pytrimal.TrimmedAlignment``
(``/root/reference/src/pytrimal/_trimal.pyx:416-1165``): same constructors, properties, views,
error messages and mask semantics.  Residues are held as one packed ``uint8[m, n]`` matrix -- the
layout the device upload wants -- instead of the reference's per-row ``std::string`` array
(``include/trimal/alignment.pxd:22``).  Only the FASTA and Clustal formats are read/written here:
trimAl's ten-format `FormatHandling` layer is I/O, not statistics.
"""
import io
import os

import numpy as np

_GAP = ord("-")
# characters trimAl's Alignment::fillMatrices accepts besides letters
_EXTRA_VALID = frozenset(b"-.?*")

SEQUENCE_TYPES = {"protein": 4, "dna": 1, "rna": 2}  # SequenceTypes bits: DNA 1, RNA 2, AA 4, DEG 8


def _is_valid_table():
    t = np.zeros(256, dtype=bool)
    for c in range(256):
        ch = bytes([c])
        t[c] = ch.isalpha() or c in _EXTRA_VALID
    return t


_VALID = _is_valid_table()


def _class_table():
    """byte -> bit flags: 1 letter (not '-', '.', '?'), 2 DNA letter, 4 RNA letter, 8 degenerate."""
    t = np.zeros(256, dtype=np.uint8)
    for c in range(256):
        up = chr(c).upper()
        flags = 0 if chr(c) in "-.?" else 1
        if up in "AGCTN":
            flags |= 2
        if up in "AGCUN":
            flags |= 4
        if up in "RYKMSWBDHV":
            flags |= 8
        t[c] = flags
    return t


_CLASS = _class_table()


def _first100_counts(block):
    """Per row of `block`: (letters seen, DNA hits, RNA hits, degenerate hits) among the first
    100 non-gap letters."""
    cls = _CLASS[block]
    letter = (cls & 1).astype(bool)
    first100 = letter & (np.cumsum(letter, axis=1, dtype=np.int32) <= 100)
    cls = np.where(first100, cls, 0)
    return (first100.sum(axis=1), ((cls >> 1) & 1).sum(axis=1), ((cls >> 2) & 1).sum(axis=1),
            ((cls >> 3) & 1).sum(axis=1))


def detect_alignment_type(matrix):
    """trimAl ``utils::checkAlignmentType`` (behind ``Alignment::getAlignmentType``,
    ``_trimal.pyx:891``): look at the first 100 non-gap letters of every sequence; an alignment
    is amino-acid as soon as one sequence has < 70 % nucleotide letters.  Only a prefix of the
    columns is scanned (widened for the rows that have not shown 100 letters yet)."""
    m, n = matrix.shape
    if m > 128:
        # upstream returns AA at the first sequence that qualifies: a protein alignment is decided by its first rows (the
        # scan of all 2000 rows of a 2000 x 10000 alignment was 2.5 ms of every first trim)
        if detect_alignment_type(matrix[:64]) == 4:
            return 4
    k = np.zeros(m, dtype=np.int64)
    hd, hr, dg = k.copy(), k.copy(), k.copy()
    rows = np.arange(m)
    width = min(n, 192)
    while True:
        kk, a, b, c = _first100_counts(matrix[rows, :width] if len(rows) < m else matrix[:, :width])
        k[rows], hd[rows], hr[rows], dg[rows] = kk, a, b, c
        rows = rows[(kk < 100)]
        if width >= n or len(rows) == 0:
            break
        width = min(n, width * 4)
    seen = k > 0
    kf = np.maximum(k, 1).astype(np.float32)
    # (upstream compares the float32 quotient with the DOUBLE literal 0.7: 14 / 20 = 0.7f = 0.69999998... is below it.
    # numpy would compare in float32, where 0.7 rounds to the same 0.7f and the test fails)
    protein = (seen & (((hd + dg).astype(np.float32) / kf).astype(np.float64) < 0.7) &
               (((hr + dg).astype(np.float32) / kf).astype(np.float64) < 0.7))
    if protein.any():  # upstream returns AA at the first such sequence
        return 4
    g_rna = int((seen & (hr > hd) & (dg == 0)).sum())
    g_dna = int((seen & (hr < hd) & (dg == 0)).sum())
    ext_rna = int((seen & (hr > hd) & (dg != 0)).sum())
    ext_dna = int((seen & (hr < hd) & (dg != 0)).sum())
    if ext_dna != 0 and ext_dna > ext_rna:
        return 1 | 8
    if ext_rna != 0 and ext_dna < ext_rna:
        return 2 | 8
    if g_rna > g_dna:
        return 2
    return 1


class _View:
    """Read-only view with zero-copy slicing (``AlignmentSequences`` / ``AlignmentResidues``)."""

    def __init__(self, owner, indices):
        self._owner = owner
        self._indices = indices

    def __len__(self):
        return len(self._indices)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __getitem__(self, index):
        if isinstance(index, slice):
            return type(self)(self._owner, self._indices[index])
        i = int(index)
        if i < 0:
            i += len(self._indices)
        if i < 0 or i >= len(self._indices):
            raise IndexError(index)
        return self._get(int(self._indices[i]))

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented


class AlignmentSequences(_View):
    """A read-only view over the sequences of an alignment."""

    def _get(self, row):
        o = self._owner
        return o._matrix[row, o._res_idx].tobytes().decode("ascii")


class AlignmentResidues(_View):
    """A read-only view over the residues (columns) of an alignment."""

    def _get(self, col):
        o = self._owner
        return o._matrix[o._seq_idx, col].tobytes().decode("ascii")


class Alignment:
    """A multiple sequence alignment."""

    def __init__(self, names, sequences, sequence_type=None):
        if names is None or sequences is None:
            raise TypeError("`names` and `sequences` must not be None")
        if len(names) != len(sequences):
            raise ValueError(f"`Alignment` given {len(names)!r} names but {len(sequences)!r} sequences")
        if sequence_type is not None and sequence_type not in SEQUENCE_TYPES:
            raise ValueError(
                f"invalid `sequence_type`: {sequence_type!r} (expected one of 'protein', 'rna', 'dna' or None)")
        validate = not isinstance(sequences, AlignmentSequences)
        rows, nres = [], 0
        self._names = []
        for i, (name, seq) in enumerate(zip(names, sequences)):
            if not isinstance(name, (bytes, bytearray)):
                raise TypeError(f"expected bytes, found {type(name).__name__}")
            raw = seq.encode("ascii") if isinstance(seq, str) else bytes(seq)
            if not nres:
                nres = len(raw)
            if len(raw) != nres:
                raise ValueError(f"Sequence length mismatch in sequence {i}: {len(raw)} != {nres}")
            self._names.append(bytes(name))
            rows.append(raw)
        m = len(rows)
        if m == 0:
            nres = 0
        self._matrix = (np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(m, nres).copy()
                        if m and nres else np.zeros((m, nres), dtype=np.uint8))
        if validate and nres:
            bad = ~_VALID[self._matrix]
            if bad.any():
                r, c = np.argwhere(bad)[0]
                raise ValueError(
                    f"The sequence \"{self._names[r].decode('ascii', 'replace')}\" has an unknown "
                    f"({int(self._matrix[r, c])}) character")
        self._datatype = SEQUENCE_TYPES.get(sequence_type, 0)
        self._seq_mask = np.ones(m, dtype=bool)
        self._res_mask = np.ones(nres, dtype=bool)
        self._reindex()

    # --- internals ------------------------------------------------------------

    @classmethod
    def _from_parts(cls, names, matrix, datatype=0, seq_mask=None, res_mask=None):
        self = cls.__new__(cls)
        # (the names of an alignment are never changed in place: a list handed over by another alignment is shared)
        self._names = names if type(names) is list else list(names)
        self._matrix = matrix
        self._datatype = datatype
        m, n = matrix.shape
        self._seq_mask = np.ones(m, dtype=bool) if seq_mask is None else np.asarray(seq_mask, dtype=bool).copy()
        self._res_mask = np.ones(n, dtype=bool) if res_mask is None else np.asarray(res_mask, dtype=bool).copy()
        self._reindex()
        return self

    def _reindex(self):
        self._seq_idx = np.flatnonzero(self._seq_mask)
        self._res_idx = np.flatnonzero(self._res_mask)
        self._detected_type = None

    def _dense(self):
        """The visible (kept) residues as a C-contiguous uint8 matrix."""
        # (index arrays as long as the matrix: every sequence and residue is visible -- no reduction over the masks per call)
        if len(self._seq_idx) == self._matrix.shape[0] and len(self._res_idx) == self._matrix.shape[1]:
            return self._matrix
        return np.ascontiguousarray(self._matrix[np.ix_(self._seq_idx, self._res_idx)])

    def _alignment_type(self):
        if self._datatype:
            return self._datatype
        cached = getattr(self, "_detected_type", None)
        if cached is None:
            dense = self._dense()
            cached = self._detected_type = 0 if dense.size == 0 else detect_alignment_type(dense)
        return cached

    # --- parser / loader --------------------------------------------------------

    @classmethod
    def load(cls, file, format=None):
        """Load a multiple sequence alignment from a path or a binary file-like object."""
        if file is None:
            raise TypeError("`file` must not be None")
        if isinstance(file, (str, bytes, os.PathLike)):
            path = os.fspath(file)
            if os.path.isdir(path):
                raise IsADirectoryError(file)
            with open(path, "rb") as f:
                data = f.read()
            fmt = format
        else:
            ty = type(file).__name__
            if not hasattr(file, "seek") or not file.seekable():
                raise TypeError(f"{ty!r} object is not seekable.")
            if not hasattr(file, "readinto") and not hasattr(file, "read"):
                raise TypeError(f"{ty!r} object has no attribute 'read'.")
            if format is None:
                raise ValueError("Format must be specified when loading from a file-like object")
            data = file.read()
            if isinstance(data, str):
                raise TypeError(f"{ty!r} object is not open in binary mode.")
            fmt = format
        if fmt is None:
            fmt = _sniff_format(data)
        fmt = fmt.lower()
        if fmt in ("fasta", "clustal"):
            fast = _load_native(cls, data, file, fmt)
            if fast is not None:
                return fast
            try:
                names, seqs = (_parse_fasta if fmt == "fasta" else _parse_clustal)(data)
            except ValueError as err:
                raise RuntimeError(f"Failed to recognize format {format!r} in {file!r}") from err
        elif fmt in _PARSERS:
            try:
                names, seqs = _PARSERS[fmt](data)
            except (ValueError, IndexError) as err:
                raise RuntimeError(f"Failed to recognize format {format!r} in {file!r}") from err
        else:
            raise ValueError(f"Unknown alignment format: {format!r}")
        if not names:
            raise RuntimeError(f"Failed to load alignment from {file!r}.")
        out = cls.__new__(cls)
        Alignment.__init__(out, names, seqs)
        return out

    def dump(self, file, format="fasta"):
        """Dump the alignment to a path or a binary file-like object."""
        text = self.dumps(format).encode("ascii")
        if isinstance(file, (str, bytes, os.PathLike)):
            with open(os.fspath(file), "wb") as f:
                f.write(text)
        else:
            file.write(text)

    def dumps(self, format="fasta", encoding="utf-8"):
        """Dump the alignment to a string in one of the formats of the reference's writer
        (``/root/reference/src/pytrimal/_trimal.pyx:604-731``): clustal, fasta, html, mega, nexus,
        phylip / phylip40, phylip32, phylippaml, nbrf / pir, and the ``_m10`` variants (names cut to
        10 characters) of fasta, nexus, phylippaml, phylip32 and phylip40."""
        fmt = format.lower()
        short = fmt.endswith("_m10")
        base = fmt[:-4] if short else fmt
        writer = _WRITERS.get(base)
        if writer is None or (short and base not in _M10_FORMATS):
            raise ValueError(f"Could not recognize alignment format: {format!r}")
        names = [n.decode(encoding) for n in self.names]
        if short:
            names = [x[:10] for x in names]
        fast = _FAST_WRITERS.get(base)
        if fast is not None and names:  # whole-matrix writers: the same bytes as the line-by-line ones below
            text = fast(names, self._dense())
            if text is not None:
                return text
        out = io.StringIO()
        writer(out, names, list(self.sequences), self._alignment_type() if names else 0)
        return out.getvalue()

    # --- magic ------------------------------------------------------------------

    def __repr__(self):
        return f"{type(self).__name__}(names={self.names!r}, sequences={list(self.sequences)!r})"

    def __copy__(self):
        return self.copy()

    def __len__(self):
        return len(self._seq_idx)

    # --- properties -------------------------------------------------------------

    @property
    def sequence_type(self):
        ty = self._alignment_type()
        if ty & 1:
            return "dna"
        if ty & 2:
            return "rna"
        if ty & 4:
            return "protein"
        return None

    @property
    def names(self):
        if len(self._seq_idx) == len(self._names):  # every sequence visible: no per-element indexing through numpy ints
            return list(self._names)
        return [self._names[i] for i in self._seq_idx.tolist()]

    @property
    def sequences(self):
        return AlignmentSequences(self, self._seq_idx)

    @property
    def residues(self):
        return AlignmentResidues(self, self._res_idx)

    def copy(self):
        return type(self)._from_parts(self._names, self._matrix.copy(), self._datatype, self._seq_mask,
                                      self._res_mask)


class TrimmedAlignment(Alignment):
    """A multiple sequence alignment that has been trimmed (masks over the original)."""

    @classmethod
    def load(cls, file, format=None):
        ali = Alignment.load(file, format)
        return cls._from_parts(ali._names, ali._matrix, ali._datatype)

    def __init__(self, names, sequences, sequences_mask=None, residues_mask=None):
        super().__init__(names, sequences)
        m, n = self._matrix.shape
        if sequences_mask is not None:
            if len(sequences_mask) != m:
                raise ValueError("Sequences mask must have the same length as the sequences list")
            self._seq_mask = np.array([bool(x) for x in sequences_mask], dtype=bool)
        if residues_mask is not None:
            if len(residues_mask) != n:
                raise ValueError("Sequences mask must have the same length as the sequences list")
            self._res_mask = np.array([bool(x) for x in residues_mask], dtype=bool)
        self._reindex()

    @property
    def residues_mask(self):
        """sequence of `bool`: Which residues are kept in the alignment."""
        return [bool(x) for x in self._res_mask]

    @property
    def sequences_mask(self):
        """sequence of `bool`: Which sequences are kept in the alignment."""
        return [bool(x) for x in self._seq_mask]

    def original_alignment(self):
        """Rebuild the original alignment from which this object was obtained."""
        return Alignment._from_parts(self._names, self._matrix.copy(), self._datatype)

    def terminal_only(self):
        """Get a trimmed alignment where only the terminal residues are removed.

        ``Cleaner::removeOnlyTerminal`` (``/root/reference/src/pytrimal/_trimal.pyx:1144-1157``,
        ``include/trimal/cleaner.pxd:38``).  **Unverified**: the body is not in the reference tree and no fixture
        of the reference exercises it; this follows the recalled upstream behaviour [R].  The
        boundaries are the first and the last column without gaps in the gap statistics of this object:

        * a trim that computed gap statistics hands them on -- the trimmed alignment shares the statistics object of
          the alignment it was trimmed from (``statistics.pxd:47-51``): the (windowed) gap vector of the ORIGINAL
          alignment, every sequence, whatever the trimmer kept ("reading 2");
        * a result without them (RepresentativeTrimmer, OverlapTrimmer, ``noduplicateseqs``, an object built from
          masks) computes them when asked, over the sequences it still holds -- upstream's lazily built ``Gaps``
          skips the sequences a trim removed ("reading 0").

        Every column between the two boundaries is restored, the columns outside keep the trimmer's decision.
        `RuntimeError` when no column is free of gaps (upstream reports an error).  No device work: the counts are
        the ones the trim fetched, or a host count over the bytes.
        """
        gaps = getattr(self, "_gaps_w", None)
        if gaps is None:
            # (a column trimmer's result whose vector was not kept -- `trim_batch` -- counts over every sequence, as the cached
            # vector would; a sequence trimmer's result, or an object built from masks, over the sequences it holds)
            shared = getattr(self, "_gap_stats", False)
            kept = self._matrix[self._seq_mask] if not (shared or self._seq_mask.all()) else self._matrix
            n = self._matrix.shape[1]
            gaps = (kept == _GAP).sum(axis=0, dtype=np.int32) if kept.shape[0] else np.zeros(n, dtype=np.int32)
            hw = getattr(self, "_gap_hw", 0)
            if hw > 0 and n:  # the window of the trim that produced this object (pure host function of the library)
                from . import _lib

                out = np.empty(n, dtype=np.int32)
                gaps = np.ascontiguousarray(gaps, dtype=np.int32)
                if _lib.load().msa_window_i32(_lib.ptr(gaps), n, hw, _lib.ptr(out)) == 0:
                    gaps = out
        res = self._res_mask.copy()
        free = np.flatnonzero(np.asarray(gaps) == 0)
        if free.size == 0:
            raise RuntimeError("the alignment has no column without gaps: terminal-only trimming is not possible")
        res[free[0]:free[-1] + 1] = True
        out = TrimmedAlignment._from_parts(self._names, self._matrix.copy(), self._datatype, self._seq_mask, res)
        out._gaps_w, out._gap_hw, out._gap_stats = getattr(self, "_gaps_w", None), getattr(self, "_gap_hw", 0), getattr(self, "_gap_stats", False)
        return out

    def copy(self):
        out = TrimmedAlignment._from_parts(self._names, self._matrix.copy(), self._datatype, self._seq_mask, self._res_mask)
        out._gaps_w, out._gap_hw, out._gap_stats = getattr(self, "_gaps_w", None), getattr(self, "_gap_hw", 0), getattr(self, "_gap_stats", False)
        return out


# --- minimal readers --------------------------------------------------------------------------

def _load_native(cls, data, file, fmt):
    """FASTA / Clustal text -> Alignment through the native ingest of libmsastat (`msa_fasta_scan` / `msa_fasta_fill`,
    `msa_clustal_scan` / `msa_clustal_fill`): one pass to size the matrix, one to fill it and validate the residues,
    no per-sequence Python objects except the names.  Returns None when the library is not built or does not
    recognise the text (the pure-Python parser takes over and reports the error)."""
    import ctypes

    from . import _lib

    try:
        L = _lib.load()
    except (RuntimeError, OSError):
        return None
    buf = np.frombuffer(data, dtype=np.uint8)
    m, n = ctypes.c_int32(0), ctypes.c_int32(0)
    scan, fill = (L.msa_fasta_scan, L.msa_fasta_fill) if fmt == "fasta" else (L.msa_clustal_scan, L.msa_clustal_fill)
    if scan(buf.ctypes.data, buf.size, ctypes.byref(m), ctypes.byref(n)) != 0:
        return None
    m, n = m.value, n.value
    if m == 0:
        if fmt != "fasta":
            return None
        raise RuntimeError(f"Failed to load alignment from {file!r}.")
    matrix = np.empty((m, n), dtype=np.uint8)
    off = np.empty(m, dtype=np.int64)
    ln = np.empty(m, dtype=np.int32)
    valid = _VALID.view(np.uint8)
    detail = _lib.ErrDetail()
    rc = fill(buf.ctypes.data, buf.size, m, n, matrix.ctypes.data, off.ctypes.data, ln.ctypes.data, valid.ctypes.data,
              ctypes.byref(detail))
    if rc not in (0, _lib.E_LENGTH_MISMATCH, _lib.E_BAD_RESIDUE):
        return None
    names = [bytes(data[o:o + k]) for o, k in zip(off.tolist(), ln.tolist())]
    if rc == _lib.E_LENGTH_MISMATCH:
        raise ValueError(f"Sequence length mismatch in sequence {detail.row}: {detail.col} != {n}")
    if rc == _lib.E_BAD_RESIDUE:
        raise ValueError(f"The sequence \"{names[detail.row].decode('ascii', 'replace')}\" has an unknown "
                         f"({detail.byte}) character")
    if rc != 0:
        return None
    out = cls.__new__(cls)
    out._names = names
    out._matrix = matrix
    out._datatype = 0
    out._seq_mask = np.ones(m, dtype=bool)
    out._res_mask = np.ones(n, dtype=bool)
    out._reindex()
    return out


def _parse_fasta(data):
    names, seqs = [], []
    for line in data.splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith(b">"):
            fields = line[1:].split()
            names.append(fields[0] if fields else b"")
            seqs.append([])
        elif names:
            seqs[-1].append(line.replace(b" ", b""))
    return names, [b"".join(s) for s in seqs]


def _sniff_format(data):
    """Content-based format detection, the role of trimAl's FormatManager::CheckAlignment chain
    (format_handling.pxd:11-32) for the formats of the reference's own loader tests."""
    head = data.lstrip()
    up = head[:8].upper()
    if up.startswith(b"CLUSTAL"):
        return "clustal"
    if up.startswith(b"#NEXUS"):
        return "nexus"
    if up.startswith(b"#MEGA"):
        return "mega"
    if head[:1] == b">":
        return "pir" if len(head) > 3 and head[3:4] == b";" and head[1:3].isalnum() else "fasta"
    first = head.split(b"\n", 1)[0].split()
    if len(first) == 2 and first[0].isdigit() and first[1].isdigit():
        try:
            _parse_phylip(data)
            return "phylip"
        except (ValueError, IndexError):
            return "phylip32"
    return "fasta"


def _phylip_header(lines):
    for k, line in enumerate(lines):
        parts = line.split()
        if parts:
            if len(parts) < 2 or not (parts[0].isdigit() and parts[1].isdigit()):
                raise ValueError("not a PHYLIP header")
            return int(parts[0]), int(parts[1]), k + 1
    raise ValueError("empty file")


def _parse_phylip(data):
    """PHYLIP 4.0, interleaved: the first block carries the names, later blocks residues only."""
    lines = data.splitlines()
    m, n, at = _phylip_header(lines)
    body = [ln for ln in lines[at:] if ln.strip()]
    if m == 0 or len(body) % m:
        raise ValueError("interleaved PHYLIP needs a multiple of the sequence count of lines")
    names, seqs = [], [[] for _ in range(m)]
    for k, line in enumerate(body):
        parts = line.split()
        if k < m:
            names.append(parts[0])
            parts = parts[1:]
        seqs[k % m].append(b"".join(parts))
    out = [b"".join(x) for x in seqs]
    if any(len(x) != n for x in out):
        raise ValueError("sequence lengths do not match the PHYLIP header")
    return names, out


def _parse_phylip32(data):
    """PHYLIP 3.2, sequential: each sequence (name first) runs on until it has its residues."""
    lines = data.splitlines()
    m, n, at = _phylip_header(lines)
    tokens = [ln.split() for ln in lines[at:] if ln.strip()]
    names, out, k = [], [], 0
    for _ in range(m):
        names.append(tokens[k][0])
        got = [b"".join(tokens[k][1:])]
        k += 1
        while sum(map(len, got)) < n:
            got.append(b"".join(tokens[k]))
            k += 1
        out.append(b"".join(got))
    if any(len(x) != n for x in out):
        raise ValueError("sequence lengths do not match the PHYLIP header")
    return names, out


def _parse_nexus(data):
    """NEXUS data block: the (possibly interleaved) MATRIX up to the closing ';'."""
    import re

    text = re.sub(rb"\[[^\]]*\]", b"", data)  # [comments]
    lines = text.splitlines()
    start = next(k for k, ln in enumerate(lines) if ln.strip().upper() == b"MATRIX")
    names, seqs = [], {}
    for line in lines[start + 1:]:
        stripped = line.strip()
        if stripped.startswith(b";"):
            break
        parts = stripped.rstrip(b";").split()
        if len(parts) < 2:
            if stripped.endswith(b";"):
                break
            continue
        if parts[0] not in seqs:
            names.append(parts[0])
            seqs[parts[0]] = []
        seqs[parts[0]].append(b"".join(parts[1:]))
        if stripped.endswith(b";"):
            break
    if not names:
        raise ValueError("empty NEXUS matrix")
    return names, [b"".join(seqs[k]) for k in names]


def _parse_pir(data):
    """NBRF/PIR: '>P1;name', a description line, residues up to '*'."""
    names, seqs = [], []
    lines = iter(data.splitlines())
    for line in lines:
        line = line.strip()
        if not line.startswith(b">"):
            continue
        names.append(line.split(b";", 1)[1].split()[0] if b";" in line else line[1:].split()[0])
        next(lines, None)  # description
        chunks = []
        for body in lines:
            body = body.strip()
            done = body.endswith(b"*")
            chunks.append(body.rstrip(b"*").replace(b" ", b""))
            if done:
                break
        seqs.append(b"".join(chunks))
    if not names:
        raise ValueError("no PIR records")
    return names, seqs


def _parse_clustal(data):
    names, seqs = [], {}
    lines = data.splitlines()
    head = 0
    while head < len(lines) and not lines[head].strip():
        head += 1
    if head == len(lines) or not lines[head].strip().upper().startswith((b"CLUSTAL", b"MUSCLE")):
        raise ValueError("not a Clustal file")
    for line in lines[head + 1:]:
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


def _parse_mega(data):
    """MEGA: '#MEGA', '!' command statements (each runs to its ';'), then '#name residues' lines,
    interleaved or not."""
    lines = data.splitlines()
    if not lines or not lines[0].strip().upper().startswith(b"#MEGA"):
        raise ValueError("not a MEGA file")
    names, seqs, in_command, current = [], {}, False, None
    for line in lines[1:]:
        stripped = line.strip()
        if in_command or stripped.startswith(b"!"):
            in_command = not stripped.endswith(b";")
            continue
        if not stripped:
            continue
        if stripped.startswith(b"#"):
            parts = stripped[1:].split()
            if not parts:
                raise ValueError("MEGA record without a name")
            current = parts[0]
            if current not in seqs:
                names.append(current)
                seqs[current] = []
            seqs[current].append(b"".join(parts[1:]))
        elif current is not None:
            seqs[current].append(stripped.replace(b" ", b""))
    if not names:
        raise ValueError("no MEGA records")
    return names, [b"".join(seqs[k]) for k in names]


_PARSERS = {"clustal": _parse_clustal, "phylip": _parse_phylip, "phylip40": _parse_phylip, "phylip32": _parse_phylip32,
            "phylippaml": _parse_phylip32, "nexus": _parse_nexus, "pir": _parse_pir, "nbrf": _parse_pir, "mega": _parse_mega}
for _name in ("nexus", "phylippaml", "phylip32", "phylip40", "phylip"):
    _PARSERS[_name + "_m10"] = _PARSERS[_name]


# --- writers (the formats of trimAl's FormatManager, format_handling.pxd:11-32; layouts follow the public format
# definitions -- blocks of 60 residues in groups of 10 where the format is free -- and every one is read back by
# the loader above) ---------------------------------------------------------------------------------------------

def _groups(seq, width=10):
    return " ".join(seq[k:k + width] for k in range(0, len(seq), width))


def _is_nucleotide(datatype):
    return bool(datatype & 3) and not datatype & 4  # SequenceTypes bits DNA | RNA without AA


def _write_fasta(out, names, seqs, datatype):
    for name, seq in zip(names, seqs):
        out.write(f">{name}\n")
        for k in range(0, len(seq), 60):
            out.write(seq[k:k + 60] + "\n")


# The two formats the reference's tests compare byte for byte (FASTA, Clustal) also have whole-matrix writers: the
# residues never become Python strings row by row, the line breaks are one reshape of the dense matrix (2000 x 10000:
# 150 -> 50 ms, 240 -> 85 ms).  They return None where the line-by-line writer must do it (names that are not ASCII: padding
# counts characters, not bytes).
def _fast_fasta(names, a):
    m, n = a.shape
    full, rem = divmod(n, 60)
    nl = np.full((m, max(full, 1), 1), 10, dtype=np.uint8)
    body = np.concatenate([a[:, :full * 60].reshape(m, full, 60), nl[:, :full]], axis=2).reshape(m, full * 61) if full else None
    tail = np.concatenate([a[:, full * 60:], nl[:, 0]], axis=1) if rem else None
    parts = []
    for i, name in enumerate(names):
        parts.append((">" + name + "\n").encode("utf-8"))
        if body is not None:
            parts.append(body[i].tobytes())
        if tail is not None:
            parts.append(tail[i].tobytes())
    return b"".join(parts).decode("utf-8")


def _fast_clustal(names, a):
    if not all(x.isascii() for x in names):
        return None
    m, n = a.shape
    width = max(len(x) for x in names) + 5
    prefix = np.frombuffer("".join(x.ljust(width) for x in names).encode("ascii"), dtype=np.uint8).reshape(m, width)
    nl = np.full((m, 1), 10, dtype=np.uint8)
    parts = [b"CLUSTAL multiple sequence alignment\n\n"]
    for k in range(0, n, 60):
        parts.append(np.concatenate([prefix, a[:, k:k + 60], nl], axis=1).tobytes())
        parts.append(b"\n\n")
    return b"".join(parts).decode("ascii")


def _write_clustal(out, names, seqs, datatype):
    # Name column = longest name + 5, blocks of 60 residues, two empty lines behind every block: byte for byte the
    # body of the one trimAl-written Clustal file in the reference tree (tests/data/example.001.gt90.w3.clw).  That
    # file's first line, "CLUSTAL 2.0.12 multiple sequence alignment", is the header of the ClustalW file it was
    # trimmed from: trimAl carries the input's header line over when input and output format agree; an alignment
    # built in memory has none, and gets trimAl's default line.
    out.write("CLUSTAL multiple sequence alignment\n\n")
    width = max((len(x) for x in names), default=0) + 5
    n = len(seqs[0]) if seqs else 0
    for k in range(0, n, 60):
        for name, seq in zip(names, seqs):
            out.write(name.ljust(width) + seq[k:k + 60] + "\n")
        out.write("\n\n")


def _write_phylip40(out, names, seqs, datatype):
    n = len(seqs[0]) if seqs else 0
    width = max(max((len(x) for x in names), default=0), 10) + 3
    out.write(f" {len(names)} {n}\n")
    for k in range(0, max(n, 1), 60):
        for name, seq in zip(names, seqs):
            out.write((name if k == 0 else "").ljust(width) + _groups(seq[k:k + 60]) + "\n")
        out.write("\n")


def _write_phylip32(out, names, seqs, datatype):
    n = len(seqs[0]) if seqs else 0
    width = max(max((len(x) for x in names), default=0), 10) + 3
    out.write(f" {len(names)} {n}\n")
    for name, seq in zip(names, seqs):
        for k in range(0, max(n, 1), 60):
            out.write((name if k == 0 else "").ljust(width) + _groups(seq[k:k + 60]) + "\n")
        out.write("\n")


def _write_phylippaml(out, names, seqs, datatype):
    n = len(seqs[0]) if seqs else 0
    width = max(max((len(x) for x in names), default=0), 10) + 3
    out.write(f" {len(names)} {n}\n")
    for name, seq in zip(names, seqs):
        out.write(name.ljust(width) + seq + "\n")


def _write_nexus(out, names, seqs, datatype):
    n = len(seqs[0]) if seqs else 0
    width = max((len(x) for x in names), default=0) + 4
    kind = "DNA" if _is_nucleotide(datatype) else "PROTEIN"
    out.write("#NEXUS\nBEGIN DATA;\n")
    out.write(f" DIMENSIONS NTAX={len(names)} NCHAR={n};\n")
    out.write(f"FORMAT DATATYPE={kind} INTERLEAVE=yes GAP=-;\n")
    for name in names:
        out.write(f"[Name: {name.ljust(width)}Len: {n}]\n")
    out.write("\nMATRIX\n")
    for k in range(0, max(n, 1), 50):
        for name, seq in zip(names, seqs):
            out.write(name.ljust(width) + _groups(seq[k:k + 50]) + "\n")
        out.write("\n")
    out.write(";\nEND;\n")


def _write_mega(out, names, seqs, datatype):
    n = len(seqs[0]) if seqs else 0
    width = max((len(x) for x in names), default=0) + 4
    kind = "DNA" if _is_nucleotide(datatype) else "protein"
    out.write("#MEGA\n!Title alignment;\n")
    out.write(f"!Format DataType={kind} NSeqs={len(names)} Nsites={n} indel=- CodeTable=Standard;\n\n")
    for k in range(0, max(n, 1), 50):
        for name, seq in zip(names, seqs):
            out.write(("#" + name).ljust(width + 1) + _groups(seq[k:k + 50]) + "\n")
        out.write("\n")


def _write_pir(out, names, seqs, datatype):
    code = "DL" if _is_nucleotide(datatype) else "P1"
    for name, seq in zip(names, seqs):
        out.write(f">{code};{name}\n{name} {len(seq)} bases\n")
        body = seq + "*"
        for k in range(0, len(body), 50):
            out.write("  " + _groups(body[k:k + 50]) + "\n")
        out.write("\n")


_HTML_CLASSES = {  # Clustal-style residue classes
    "AVFPMILW": "hyd", "DE": "neg", "RK": "pos", "STYHCNGQ": "pol",
}


def _write_html(out, names, seqs, datatype):
    import html

    n = len(seqs[0]) if seqs else 0
    width = max((len(x) for x in names), default=0) + 4
    cls = {c: k for letters, k in _HTML_CLASSES.items() for c in letters}
    out.write("<!DOCTYPE html>\n<html><head><meta charset=\"utf-8\"><title>alignment</title>\n<style>\n"
              "pre{font-family:monospace} .hyd{background:#f9a19a} .neg{background:#d6a5f7} .pos{background:#9ab8f9}"
              " .pol{background:#a5f7b0}\n</style></head><body><pre>\n")
    for k in range(0, max(n, 1), 120):
        for name, seq in zip(names, seqs):
            out.write(html.escape(name).ljust(width))
            for c in seq[k:k + 120]:
                tag = cls.get(c.upper())
                out.write(f'<span class="{tag}">{c}</span>' if tag else html.escape(c))
            out.write("\n")
        out.write("\n")
    out.write("</pre></body></html>\n")


_WRITERS = {"fasta": _write_fasta, "clustal": _write_clustal, "phylip": _write_phylip40, "phylip40": _write_phylip40,
            "phylip32": _write_phylip32, "phylippaml": _write_phylippaml, "nexus": _write_nexus, "mega": _write_mega,
            "pir": _write_pir, "nbrf": _write_pir, "html": _write_html}
_M10_FORMATS = {"fasta", "nexus", "phylippaml", "phylip32", "phylip40", "phylip"}
_FAST_WRITERS = {"fasta": _fast_fasta, "clustal": _fast_clustal}
