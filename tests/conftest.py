import os
import sys

import numpy as np
import pytest

# torch's wheel bundles its own libamdhip64; it has to be the first HIP runtime mapped into the
# process, otherwise torch later reports "No HIP GPUs are available" (two runtimes, one KFD).
# libmsastat_hip.so then binds to the already-loaded runtime (same SONAME).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the library honours its MSA_* diagnostic switches (tests/test_gpu_parity.py KERNELS, the fuzzers) only when this one is set;
# with none of the others set it changes nothing: the default-dispatch tests still run what ships
os.environ.setdefault("MSA_DIAGNOSTICS", "1")

DATA = os.path.join(ROOT, "tests", "golden", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")

EXAMPLE_001_NAMES = [b"Sp8", b"Sp10", b"Sp26", b"Sp6", b"Sp17", b"Sp33"]
# example.001.AA.clw, reproduced in the reference at src/pytrimal/_trimal.pyx:11-19
EXAMPLE_001 = [
    "-----GLGKVIV-YGIVLGTKSDQFSNWVVWLFPWNGLQIHMMGII",
    "-------DPAVL-FVIMLGTIT-KFS--SEWFFAWLGLEINMMVII",
    "AAAAAAAAALLTYLGLFLGTDYENFA--AAAANAWLGLEINMMAQI",
    "-----ASGAILT-LGIYLFTLCAVIS--VSWYLAWLGLEINMMAII",
    "--FAYTAPDLL-LIGFLLKTVA-TFG--DTWFQLWQGLDLNKMPVF",
    "-------PTILNIAGLHMETDI-NFS--LAWFQAWGGLEINKQAIL",
]
# the OverlapTrimmer docstring example, src/pytrimal/_trimal.pyx:1676-1684
OVERLAP_NAMES = [b"Sp8", b"Sp17", b"Sp10", b"Sp26"]
OVERLAP_EXAMPLE = [
    "LG-----------TKSD---NNNNNNNNNNNNNNNNWV----------",
    "APDLLL-IGFLLKTV-ATFG-----------------DTWFQLWQGLD",
    "DPAVL--FVIMLGTI-TKFS-----------------SEWFFAWLGLE",
    "AAALLTYLGLFLGTDYENFA-----------------AAAANAWLGLE",
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # trimAl's "Removing sequence ... composed only by gaps" is API behaviour the randomised tests trigger hundreds of
    # times; the tests that are about it use pytest.warns, which sees it regardless of this filter
    config.addinivalue_line("filterwarnings", "ignore:Removing sequence:RuntimeWarning")


def data_path(name):
    return os.path.join(DATA, name)


@pytest.fixture(scope="session")
def enog():
    """The untrimmed ENOG411BWBU alignment (209 x 1227): the reference's input fixture is a
    dangling symlink, but ENOG411BWBU.seq40.res60.fasta is the same alignment."""
    import oracle

    names, seqs = oracle.read_fasta(data_path("ENOG411BWBU.seq40.res60.fasta"))
    return names, seqs, oracle.pack(seqs)


def edge_msa(m=64, n=256, seed=11):
    """Seeded synthetic with X, B, Z, lower-case letters and all-gap columns."""
    r = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    a = alpha[r.integers(0, 20, (m, n))].copy()
    a[r.random((m, n)) < 0.3] = ord("-")
    a[r.random((m, n)) < 0.02] = ord("X")
    a[r.random((m, n)) < 0.01] = ord("B")
    a[r.random((m, n)) < 0.01] = ord("Z")
    low = r.random((m, n)) < 0.05
    a[low & (a >= 65) & (a <= 90)] += 32
    a[:, r.integers(0, n, max(1, n // 40))] = ord("-")
    a[:, r.integers(0, n, max(1, n // 60))] = ord("X")
    return np.ascontiguousarray(a)
