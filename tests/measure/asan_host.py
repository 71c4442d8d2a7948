# host-only entry points of the library (ingest, windows, cut points, cleaners) from an ASan/UBSan build of msastat_host.cpp
import ctypes, os, sys, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import oracle
L = ctypes.CDLL("/tmp/libmsahost_asan.so")
vp, i32, i64, f32, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double
class Err(ctypes.Structure): _fields_=[("row",i32),("col",i32),("byte",i32)]
for name in ("fasta","clustal"):
    getattr(L, f"msa_{name}_scan").argtypes=[vp,i64,ctypes.POINTER(i32),ctypes.POINTER(i32)]
    getattr(L, f"msa_{name}_fill").argtypes=[vp,i64,i32,i32,vp,vp,vp,vp,ctypes.POINTER(Err)]
valid = np.zeros(256, dtype=np.uint8)
for c in range(256):
    if bytes([c]).isalpha() or c in b"-.?*": valid[c]=1
rng = np.random.default_rng(1)
def ingest(fmt, data):
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    m, n = i32(0), i32(0)
    rc = getattr(L, f"msa_{fmt}_scan")(buf.ctypes.data, len(data), ctypes.byref(m), ctypes.byref(n))
    if rc: return rc, None
    mat = np.zeros((max(m.value,1), max(n.value,1)), dtype=np.uint8); off=np.zeros(max(m.value,1),dtype=np.int64); ln=np.zeros(max(m.value,1),dtype=np.int32)
    d = Err()
    rc = getattr(L, f"msa_{fmt}_fill")(buf.ctypes.data, len(data), m.value, n.value, mat.ctypes.data, off.ctypes.data, ln.ctypes.data, valid.ctypes.data, ctypes.byref(d))
    return rc, (m.value, n.value)
files = ["ENOG411BWBU.seq40.res60.fasta","example.001.gt90.w3.clw","halorhodopsin.afa","PF12574.full.afa"]
cases = 0
for f in files:
    data = open(os.path.join(ROOT, "tests", "golden", "data", f),"rb").read()
    fmt = "clustal" if f.endswith(".clw") else "fasta"
    rc, shape = ingest(fmt, data); assert rc == 0, (f, rc)
    # every prefix / random corruption: must return a code, never touch memory it does not own
    for k in range(300):
        cut = int(rng.integers(0, len(data)))
        d2 = bytearray(data[:cut]) if k % 3 == 0 else bytearray(data)
        for _ in range(int(rng.integers(0, 6))):
            if len(d2): d2[int(rng.integers(0, len(d2)))] = int(rng.integers(0, 256))
        for fm in ("fasta","clustal"):
            ingest(fm, bytes(d2)); cases += 1
# text in another format passed as clustal: refused (header check)
assert ingest("clustal", b">a\nACGT\n>b\nACGA\n")[0] != 0
assert ingest("clustal", b"\n\nCLUSTAL W\n\na ACGT\nb ACGA\n")[0] == 0
assert ingest("clustal", b"MUSCLE (3.8)\n\na ACGT\nb ACGA\n")[0] == 0
# windows / cleaners on random vectors against the oracle
L.msa_window_i32.argtypes=[vp,i32,i32,vp]; L.msa_clean_gaps.argtypes=[vp,i32,f64,f32,vp]
for _ in range(300):
    n = int(rng.integers(1, 400)); g = rng.integers(0, 50, n).astype(np.int32); hw = int(rng.integers(0, max(1, n//4)+1))
    out = np.zeros(n, dtype=np.int32)
    if L.msa_window_i32(g.ctypes.data, n, hw, out.ctypes.data) == 0:
        assert np.array_equal(out, oracle.gaps_window(g, hw))
    keep = np.zeros(n, dtype=np.uint8); L.msa_clean_gaps(g.ctypes.data, n, float(rng.integers(0,50)), float(rng.choice([0, 30, 60])), keep.ctypes.data)
print("asan host run ok:", cases, "corrupted ingests, no report")
