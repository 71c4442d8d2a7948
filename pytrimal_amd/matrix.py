"""`SimilarityMatrix`: the constant input of the similarity statistic.

Mirrors ``pytrimal.SimilarityMatrix`` (``/root/reference/src/pytrimal/_trimal.pyx:1867-2060``):
same constructor, class methods, lookups and error messages.  The reference derives from
``scoring_matrices.ScoringMatrix`` (absent from this environment); only what the trim path
and the reference's tests use is provided here.
"""
import math

import numpy as np

AA_ALPHABET = "ARNDCQEGHILKMFPSTWYV"   # trimAl `aminoAcidResidues`
NT_ALPHABET = "ACGTU"                   # trimAl `nucleotideResidues`
NT_DEG_ALPHABET = "ACGTURYKMSWBDHV"     # trimAl `degenerateNucleotideResidues`

_BLOSUM62 = """
 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0
-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3
-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3
-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3
 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1
-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2
-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2
 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3
-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3
-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3
-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1
-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2
-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1
-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1
-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2
 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0
-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3
-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1
 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4
"""

_IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "U", "R": "AG", "Y": "CT", "K": "GT", "M": "AC",
          "S": "CG", "W": "AT", "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG"}


def _deg_score(x, y):
    if x == y:
        return 1.0
    sx, sy = set(_IUPAC[x]), set(_IUPAC[y])
    return len(sx & sy) / float(len(sx) * len(sy)) / 2.0


def _blosum62():
    return [[float(x) for x in line.split()] for line in _BLOSUM62.strip().splitlines()]


class SimilarityMatrix:
    """A similarity matrix for biological sequence characters."""

    DEFAULT_ALPHABET = AA_ALPHABET

    # --- Class methods ------------------------------------------------------

    @classmethod
    def aa(cls):
        """Create a default amino-acid similarity matrix (BLOSUM62)."""
        return cls(_blosum62(), alphabet=AA_ALPHABET, name="BLOSUM62")

    @classmethod
    def nt(cls, degenerated=False):
        """Create a default nucleotide similarity matrix.

        The values of trimAl's built-in tables are not in the reference tree; they are
        restated here (identity with T == U; for the degenerate alphabet, IUPAC set overlap
        halved whenever a degenerate symbol takes part) and pinned only by the three known
        answers of ``_trimal.pyx:2005-2009,2042-2046`` -- the degenerate table is the one
        reading of them that reproduces ``distance('A', 'T') == 1.5184``.
        """
        if degenerated:
            alphabet = NT_DEG_ALPHABET
            matrix = [[_deg_score(x, y) for y in alphabet] for x in alphabet]
        else:
            alphabet = NT_ALPHABET
            matrix = [[1.0 if (x == y or {x, y} == {"T", "U"}) else 0.0 for y in alphabet] for x in alphabet]
        return cls(matrix, alphabet=alphabet)

    @classmethod
    def from_name(cls, name="BLOSUM62"):
        if name.upper() == "BLOSUM62":
            order = "".join(sorted(AA_ALPHABET))
            b = _blosum62()
            idx = [AA_ALPHABET.index(ch) for ch in order]
            return cls([[b[i][j] for j in idx] for i in idx], alphabet=order, name="BLOSUM62")
        raise ValueError(f"unknown scoring matrix: {name!r} (only BLOSUM62 is built in)")

    # --- Magic methods ------------------------------------------------------

    def __init__(self, matrix, alphabet=AA_ALPHABET, name=None):
        if matrix is None or alphabet is None:
            raise TypeError("`matrix` and `alphabet` must not be None")
        rows = [list(r) for r in matrix]
        size = len(alphabet)
        if len(rows) != size or any(len(r) != size for r in rows):
            raise ValueError(f"Matrix must be square and match the alphabet length ({size})")
        if len(set(alphabet)) != size:
            raise ValueError(f"Duplicate symbols in alphabet: {alphabet!r}")
        # check alphabet constraints (_trimal.pyx:1967-1981)
        if not alphabet.isupper():
            raise ValueError("Alphabet must only contain uppercase letters")
        if size > 28:
            raise ValueError(f"Cannot use alphabet of more than 28 symbols: {alphabet!r}")
        self.alphabet = alphabet
        self.name = name
        self._size = size
        self._sim = np.array(rows, dtype=np.float32).reshape(size, size)
        self._vhash = np.full(26, -1, dtype=np.int32)
        for i, letter in enumerate(alphabet):
            j = ord(letter) - ord("A")
            if j < 0 or j >= 26:
                raise ValueError(f"Invalid symbol in alphabet: {letter!r}")
            self._vhash[j] = i
        # Euclidean distance with a float32 accumulator (_trimal.pyx:1987-1997)
        sim = self._sim
        dist = np.zeros((size, size), dtype=np.float32)
        for j in range(size):
            for i in range(j + 1, size):
                diff = sim[:, j] - sim[:, i]          # float32
                sq = diff * diff                      # float32, each product rounded
                total = np.float32(0)
                for k in range(size):
                    total = np.float32(total + sq[k])
                dist[i, j] = dist[j, i] = np.float32(math.sqrt(float(total)))
        self._dist = dist

    def _device_arrays(self):
        """(vhash int32[26], dist float32[size * size]), C-contiguous: what `msa_trim_params` points into (built once)."""
        arrs = getattr(self, "_c_arrays", None)
        if arrs is None:
            arrs = self._c_arrays = (np.ascontiguousarray(self._vhash, dtype=np.int32),
                                     np.ascontiguousarray(self._dist, dtype=np.float32))
        return arrs

    def __len__(self):
        return self._size

    def __repr__(self):
        name = f", name={self.name!r}" if self.name is not None else ""
        return f"{type(self).__name__}({self.matrix!r}, alphabet={self.alphabet!r}{name})"

    def __reduce__(self):
        return (type(self), (self.matrix, self.alphabet, self.name))

    def __eq__(self, other):
        return (isinstance(other, SimilarityMatrix) and self.alphabet == other.alphabet
                and np.array_equal(self._sim, other._sim))

    __hash__ = None

    # --- Properties ---------------------------------------------------------

    @property
    def matrix(self):
        return [[float(x) for x in row] for row in self._sim]

    # --- Functions ----------------------------------------------------------

    def _index(self, ch):
        if not isinstance(ch, str):
            raise TypeError(f"expected str, found {type(ch).__name__}")
        if len(ch) != 1:  # Cython's `ord` on a str argument raises ValueError for these
            raise ValueError(f"only single character strings can be used as symbols, got length {len(ch)}")
        code = ord(ch)
        if code < ord("A") or code > ord("Z"):
            raise ValueError(f"the symbol {ch!r} is incorrect")
        idx = int(self._vhash[code - ord("A")])
        if idx == -1:
            raise ValueError(f"the symbol {ch!r} accesing the matrix is not defined in this object")
        return idx

    def similarity(self, a, b):
        """Return the similarity between two sequence characters."""
        return float(self._sim[self._index(a), self._index(b)])

    def distance(self, a, b):
        """Return the distance between two sequence characters."""
        # similarityMatrix::getDistance upper-cases its arguments first
        up = lambda ch: ch.upper() if isinstance(ch, str) and len(ch) == 1 and ch.isalpha() else ch
        return float(self._dist[self._index(up(a)), self._index(up(b))])
