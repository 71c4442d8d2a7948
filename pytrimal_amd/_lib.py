"""ctypes binding of ``libmsastat_hip.so`` (C ABI: ``include/msastat.h``).

This is the only way the package computes anything: there is no CPU fallback.  If the shared
library is missing or no MI355X device is visible, the trimmers fail loudly with `RuntimeError`
(the reference does the same for a SIMD platform that is not available,
``/root/reference/src/pytrimal/_trimal.pyx:1205-1218``).

Note for processes that also use PyTorch-ROCm (bench.py, pytrimal_amd.batch): ``import torch``
BEFORE the first call into this module.  The torch wheel bundles its own ``libamdhip64.so``; if
the system runtime is mapped first (through this library), torch later finds no GPU.
"""
import collections
import ctypes
import os
import threading

import numpy as np

# Concurrent trims (one context = two HIP streams per thread) only overlap on the GPU when their streams
# land on different hardware queues; the runtime's default is 4 queues per process, which makes batches of
# 3+ threads serialise erratically.  Read by the HIP runtime when it initialises (first HIP call), so
# this has no effect in a process that has already touched the GPU -- export it yourself there.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsastat_hip.so")

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM, E_WINDOW_TOO_BIG = -1, -2, -3, -4, -5
E_INCORRECT_SYMBOL, E_UNDEFINED_SYMBOL, E_NOT_IMPLEMENTED, E_NON_ASCII = -6, -7, -8, -9
E_LENGTH_MISMATCH, E_BAD_RESIDUE = -10, -11

METHOD_CODES = {
    None: 0, "strict": 1, "strictplus": 2, "gappyout": 3, "nogaps": 4, "noallgaps": 5,
    "automated1": 6, "automated2": 7, "noduplicateseqs": 8,
}

# every symbol include/msastat.h declares (tests check the library exports all of them)
EXPORTS = [
    "msa_strerror", "msa_device_count", "msa_last_hip_error", "msa_ctx_create", "msa_ctx_destroy",
    "msa_ctx_stream", "msa_ctx_sync", "msa_upload_rows", "msa_upload_packed", "msa_upload_packed_async", "msa_host_register", "msa_host_unregister", "msa_attach_device",
    "msa_gaps", "msa_gaps_cached", "msa_pair_counts", "msa_identities", "msa_identity_stats", "msa_similarity",
    "msa_overlap", "msa_window_i32", "msa_window_f32", "msa_gaps_cutpoint",
    "msa_gaps_cutpoint_2nd_slope", "msa_similarity_cutpoint", "msa_clean_gaps",
    "msa_clean_similarity", "msa_clean_both", "msa_clean_strict", "msa_select_method",
    "msa_representatives", "msa_cutpoint_clusters", "msa_trim", "msa_trim_only_gaps_rows", "msa_batch_create", "msa_batch_destroy", "msa_batch_workers", "msa_trim_batch",
    "msa_batch_only_gaps_rows", "msa_batch_last_hip_error", "msa_prof_get", "msa_prof_reset",
    "msa_prof_enable", "msa_debug_sim_launches", "msa_debug_last_paths", "msa_debug_switches_enabled", "msa_fasta_scan", "msa_fasta_fill", "msa_clustal_scan", "msa_clustal_fill",
]


class ErrDetail(ctypes.Structure):
    _fields_ = [("row", ctypes.c_int32), ("col", ctypes.c_int32), ("byte", ctypes.c_int32)]


class TrimParams(ctypes.Structure):
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
        ("vhash", ctypes.c_void_p),
        ("dist", ctypes.c_void_p),
        ("npos", ctypes.c_int32),
    ]


class TrimInfo(ctypes.Structure):
    _fields_ = [
        ("selected_method", ctypes.c_int32),
        ("avg_seq", ctypes.c_float),
        ("max_seq", ctypes.c_float),
        ("gap_cut", ctypes.c_int32),
        ("sim_cut", ctypes.c_float),
        ("kept_residues", ctypes.c_int32),
        ("kept_sequences", ctypes.c_int32),
        ("err", ErrDetail),
        ("ms_device", ctypes.c_float),
        ("warnings", ctypes.c_uint32),
        ("warn_row", ctypes.c_int32),
    ]


W_ONLY_GAPS_SEQUENCES, W_NO_COLUMNS_LEFT, W_UNDEFINED_IDENTITY = 1, 2, 4


_lib = None
_lock = threading.Lock()


def load():
    """Load the HIP library, or raise `RuntimeError` (never falls back to anything else)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pytrimal_amd/csrc` (hipcc --offload-arch=gfx950)"
            )
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, f32, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_double
        L.msa_strerror.restype = ctypes.c_char_p
        L.msa_strerror.argtypes = [ctypes.c_int]
        L.msa_device_count.restype = ctypes.c_int
        L.msa_last_hip_error.restype = ctypes.c_char_p
        L.msa_last_hip_error.argtypes = [vp]
        L.msa_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.msa_ctx_destroy.argtypes = [vp]
        L.msa_ctx_destroy.restype = None
        L.msa_ctx_stream.argtypes = [vp]
        L.msa_ctx_stream.restype = vp
        L.msa_ctx_sync.argtypes = [vp]
        L.msa_upload_rows.argtypes = [vp, vp, i32, i32, ctypes.c_uint8]
        L.msa_upload_packed.argtypes = [vp, vp, i32, i32, ctypes.c_int64, ctypes.c_uint8]
        L.msa_host_register.argtypes = [vp, ctypes.c_size_t]
        L.msa_host_unregister.argtypes = [vp]
        L.msa_upload_packed_async.argtypes = L.msa_upload_packed.argtypes
        L.msa_attach_device.argtypes = [vp, vp, i32, i32, ctypes.c_int64, ctypes.c_uint8]
        L.msa_gaps.argtypes = [vp, vp, vp]
        L.msa_gaps_cached.argtypes = [vp, i32, vp]
        L.msa_pair_counts.argtypes = [vp, vp, vp]
        L.msa_identities.argtypes = [vp, vp, vp]
        L.msa_identity_stats.argtypes = [vp, ctypes.POINTER(f32), ctypes.POINTER(f32)]
        L.msa_similarity.argtypes = [vp, vp, vp, i32, vp, vp, vp, ctypes.POINTER(ErrDetail)]
        L.msa_overlap.argtypes = [vp, f32, vp]
        L.msa_fasta_scan.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.msa_fasta_fill.argtypes = [vp, ctypes.c_int64, i32, i32, vp, vp, vp, vp, ctypes.POINTER(ErrDetail)]
        L.msa_clustal_scan.argtypes = L.msa_fasta_scan.argtypes
        L.msa_clustal_fill.argtypes = L.msa_fasta_fill.argtypes
        L.msa_window_i32.argtypes = [vp, i32, i32, vp]
        L.msa_window_f32.argtypes = [vp, i32, i32, vp]
        L.msa_gaps_cutpoint.argtypes = [vp, i32, i32, f32, f32]
        L.msa_gaps_cutpoint.restype = f64
        L.msa_gaps_cutpoint_2nd_slope.argtypes = [vp, i32, i32]
        L.msa_gaps_cutpoint_2nd_slope.restype = i32
        L.msa_similarity_cutpoint.argtypes = [vp, i32, f32, f32]
        L.msa_similarity_cutpoint.restype = f64
        L.msa_clean_gaps.argtypes = [vp, i32, f64, f32, vp]
        L.msa_clean_similarity.argtypes = [vp, i32, f32, f32, vp]
        L.msa_clean_both.argtypes = [vp, vp, i32, f64, f32, f32, vp]
        L.msa_clean_strict.argtypes = [vp, vp, vp, i32, i32, i32, vp, ctypes.POINTER(i32), ctypes.POINTER(f32)]
        L.msa_select_method.argtypes = [f32, f32, i32]
        L.msa_select_method.restype = i32
        L.msa_representatives.argtypes = [vp, vp, i32, f32, vp, ctypes.POINTER(i32)]
        L.msa_cutpoint_clusters.argtypes = [vp, vp, i32, i32]
        L.msa_cutpoint_clusters.restype = f32
        L.msa_trim.argtypes = [vp, ctypes.POINTER(TrimParams), vp, vp, ctypes.POINTER(TrimInfo)]
        L.msa_trim_only_gaps_rows.argtypes = [vp, vp, i32]
        L.msa_batch_create.argtypes = [ctypes.c_int, i32, ctypes.POINTER(vp)]
        L.msa_batch_destroy.argtypes = [vp]
        L.msa_batch_destroy.restype = None
        L.msa_batch_workers.argtypes = [vp]
        L.msa_batch_workers.restype = i32
        L.msa_trim_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.msa_batch_only_gaps_rows.argtypes = [vp, i32, vp, i32]
        L.msa_batch_last_hip_error.argtypes = [vp, i32]
        L.msa_batch_last_hip_error.restype = ctypes.c_char_p
        L.msa_prof_get.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(f32), ctypes.POINTER(i32)]
        L.msa_prof_reset.argtypes = [vp]
        L.msa_prof_reset.restype = None
        L.msa_prof_enable.argtypes = [vp, ctypes.c_int]
        L.msa_prof_enable.restype = None
        _lib = L
        return L


def device_count():
    """Number of visible HIP devices; 0 when the library is missing or the host has no GPU."""
    try:
        return int(load().msa_device_count())
    except (RuntimeError, OSError):
        return 0


def ptr(arr):
    return None if arr is None else arr.ctypes.data_as(ctypes.c_void_p)


# Page-locked caller arrays (`pin_array`): id(array) -> [finalizer, bytes, address, pid], least recently used first.
# Bounded: PYTRIMAL_AMD_PIN_MB (default 1024; 0 = never page-lock) is the most this process keeps locked through this
# path; the least recently uploaded arrays are unregistered to make room.  Guarded by a lock (trims run from several
# threads); an entry made by another process (a fork inherits the table, not the registrations) is ignored, and its
# finalizer does not enter a HIP runtime that process never initialised.
_pinned = collections.OrderedDict()
_pin_lock = threading.RLock()  # (re-entrant: a finalizer of a pinned array may run on the thread that holds it, inside an allocation)


_pin_budget_cache = (None, 1024 << 20)


def pin_budget_bytes():
    global _pin_budget_cache
    text = os.environ.get("PYTRIMAL_AMD_PIN_MB")
    if text != _pin_budget_cache[0]:  # (parsed once per value: trim_batch asks for every alignment)
        try:
            value = max(0, int(text if text is not None else "1024")) << 20
        except ValueError:
            value = 1024 << 20
        _pin_budget_cache = (text, value)
    return _pin_budget_cache[1]


def pinned_bytes():
    """Bytes this process currently keeps page-locked through `pin_array`."""
    pid = os.getpid()
    with _pin_lock:
        return sum(e[1] for e in _pinned.values() if e[3] == pid)


def _unpin_entry(entry):
    fin, _, address, pid = entry
    fin.detach()
    if pid == os.getpid():
        load().msa_host_unregister(ctypes.c_void_p(address))


def pin_array(a):
    """Page-lock the memory of a C-contiguous array (`msa_host_register`) until the array object dies or the budget
    (`pin_budget_bytes`) needs its room: uploads of it are then one DMA copy straight from its rows.  Returns False
    (and changes nothing) when that is not possible or not allowed."""
    import weakref

    budget = pin_budget_bytes()
    if not isinstance(a, np.ndarray) or not a.flags.c_contiguous or a.nbytes == 0 or a.nbytes > budget:
        return False
    key, pid = id(a), os.getpid()
    with _pin_lock:
        entry = _pinned.get(key)
        if entry is not None and entry[3] == pid:
            _pinned.move_to_end(key)
            return True
        if entry is not None:  # inherited through a fork: not registered in this process
            entry[0].detach()
            del _pinned[key]
        lib, address, nbytes = load(), a.ctypes.data, a.nbytes
        # make room: the least recently uploaded arrays first
        used = sum(e[1] for e in _pinned.values() if e[3] == pid)
        for k in list(_pinned):
            if used + nbytes <= budget:
                break
            e = _pinned.pop(k, None)
            if e is None:  # (released by a finalizer that ran under this very loop)
                continue
            if e[3] == pid:
                used -= e[1]
            _unpin_entry(e)
        if lib.msa_host_register(ctypes.c_void_p(address), nbytes) != OK:
            return False

        def release(key=key, address=address, pid=pid):
            # (runs when the array is collected, before numpy frees the buffer -- from whatever thread, or process, does that)
            with _pin_lock:
                e = _pinned.pop(key, None)
            if e is not None and pid == os.getpid():
                lib.msa_host_unregister(ctypes.c_void_p(address))

        fin = weakref.finalize(a, release)
        fin.atexit = False  # (at interpreter exit the process's mappings go away by themselves: no calls into a runtime that is shutting down)
        _pinned[key] = [fin, nbytes, address, pid]
        return True


class MsaError(RuntimeError):
    def __init__(self, code, message, detail=None):
        super().__init__(message)
        self.code, self.detail = code, detail


def check(lib, ctx, rc, detail=None):
    """Turn a C-ABI return code into the exception type the reference raises for it
    (``/root/reference/src/trimal/source/reportsystem.cpp:44-53``)."""
    if rc == OK:
        return
    msg = lib.msa_strerror(rc).decode()
    if rc in (E_INCORRECT_SYMBOL, E_UNDEFINED_SYMBOL):
        ch = chr(detail.byte) if detail is not None else "?"
        if rc == E_INCORRECT_SYMBOL:
            raise ValueError(f"the symbol {ch!r} is incorrect")
        raise ValueError(f"the symbol {ch!r} accesing the matrix is not defined in this object")
    if rc == E_HIP and ctx:
        msg += ": " + lib.msa_last_hip_error(ctx).decode()
    if rc == E_NON_ASCII:
        raise ValueError(msg)
    raise MsaError(rc, msg, detail)


class Context:
    """One `msa_ctx`: device buffers + a HIP stream.  Not shared between threads."""

    def __init__(self, device=None):
        self.lib = load()
        if device is None:
            device = int(os.environ.get("PYTRIMAL_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = self.lib.msa_device_count()
            if n > 0:
                device %= n
        h = ctypes.c_void_p()
        rc = self.lib.msa_ctx_create(int(device), ctypes.byref(h))
        if rc != OK:
            raise RuntimeError(
                f"cannot create a HIP context on device {device}: {self.lib.msa_strerror(rc).decode()} "
                "(platform='hip' needs a visible gfx950 device; there is no CPU fallback)"
            )
        self.h = h
        self.device = int(device)
        self.shape = (0, 0)

    def close(self):
        if getattr(self, "h", None):
            self.lib.msa_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- uploads ---
    def upload(self, matrix, indet, pin=False, wait=True):
        """`pin`: page-lock the rows first (once per array object; kept until the array dies) -- for rows that are
        uploaded again and again.  `wait=False`: do not wait for the copy -- the caller keeps `matrix` alive and unchanged
        until the next call that returns results (`trim`, `gaps`, ...); the array is remembered here until then."""
        a = np.ascontiguousarray(matrix, dtype=np.uint8)
        if pin and a is matrix:
            pin_array(a)
        m, n = a.shape
        if wait:
            check(self.lib, self.h, self.lib.msa_upload_packed(self.h, ptr(a), m, n, n, indet))
            self._in_flight = None
        else:
            check(self.lib, self.h, self.lib.msa_upload_packed_async(self.h, ptr(a), m, n, n, indet))
            self._in_flight = a  # (a reference: the rows cannot be freed under the copy)
        self.shape = (m, n)

    def upload_rows(self, rows, indet):
        """`rows`: equally long bytes-like sequences, handed over as an array of row pointers (msa_upload_rows: what
        a binding that holds one buffer per sequence would call)"""
        bufs = [np.frombuffer(r, dtype=np.uint8) for r in rows]
        m = len(bufs)
        n = len(bufs[0]) if m else 0
        if any(len(b) != n for b in bufs):
            raise ValueError("sequences of different lengths")
        ptrs = (ctypes.c_void_p * max(m, 1))(*[b.ctypes.data for b in bufs])
        check(self.lib, self.h, self.lib.msa_upload_rows(self.h, ptrs, m, n, indet))
        self.shape = (m, n)

    def attach(self, dev_ptr, m, n, ld, indet):
        check(self.lib, self.h, self.lib.msa_attach_device(self.h, ctypes.c_void_p(dev_ptr), m, n, ld, indet))
        self.shape = (m, n)

    # --- statistics ---
    def gaps(self, with_indet=False):
        m, n = self.shape
        g = np.zeros(n, dtype=np.int32)
        x = np.zeros(n, dtype=np.int32) if with_indet else None
        check(self.lib, self.h, self.lib.msa_gaps(self.h, ptr(g), ptr(x)))
        return (g, x) if with_indet else g

    def gaps_cached(self, half_window=0):
        """The windowed gap vector if the context already holds the counts on the host (no device work), else None."""
        _, n = self.shape
        out = np.empty(n, dtype=np.int32)
        rc = self.lib.msa_gaps_cached(self.h, int(half_window), ptr(out))
        if rc == 1:
            return None
        check(self.lib, self.h, rc)
        return out

    def pair_counts(self):
        m, _ = self.shape
        hit = np.zeros((m, m), dtype=np.uint32)
        dst = np.zeros((m, m), dtype=np.uint32)
        check(self.lib, self.h, self.lib.msa_pair_counts(self.h, ptr(hit), ptr(dst)))
        return hit, dst

    def identities(self, want_ident=True, want_w=True):
        m, _ = self.shape
        ident = np.zeros((m, m), dtype=np.float32) if want_ident else None
        w = np.zeros((m, m), dtype=np.float32) if want_w else None
        check(self.lib, self.h, self.lib.msa_identities(self.h, ptr(ident), ptr(w)))
        return ident, w

    def identity_stats(self):
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        check(self.lib, self.h, self.lib.msa_identity_stats(self.h, ctypes.byref(a), ctypes.byref(b)))
        return np.float32(a.value), np.float32(b.value)

    def similarity(self, vhash, dist, gaps_windowed=None):
        _, n = self.shape
        vhash = np.ascontiguousarray(vhash, dtype=np.int32)
        dist = np.ascontiguousarray(dist, dtype=np.float32)
        gw = None if gaps_windowed is None else np.ascontiguousarray(gaps_windowed, dtype=np.int32)
        mdk = np.zeros(n, dtype=np.float32)
        q = np.zeros(n, dtype=np.float32)
        det = ErrDetail()
        rc = self.lib.msa_similarity(self.h, ptr(vhash), ptr(dist), dist.shape[0], ptr(gw), ptr(mdk), ptr(q),
                                     ctypes.byref(det))
        check(self.lib, self.h, rc, det)
        return mdk, q

    def overlap(self, residue_overlap):
        m, _ = self.shape
        out = np.zeros(m, dtype=np.float32)
        check(self.lib, self.h, self.lib.msa_overlap(self.h, residue_overlap, ptr(out)))
        return out

    def trim(self, params):
        m, n = self.shape
        keep_res = np.ones(n, dtype=np.uint8)
        keep_seq = np.ones(m, dtype=np.uint8)
        info = TrimInfo()
        rc = self.lib.msa_trim(self.h, ctypes.byref(params), ptr(keep_res), ptr(keep_seq), ctypes.byref(info))
        self._in_flight = None  # (msa_trim waited for the stream, error or not)
        check(self.lib, self.h, rc, info.err)
        return keep_res.astype(bool), keep_seq.astype(bool), info

    def only_gaps_rows(self):
        """The sequences the last `trim` removed because it left them with gaps only."""
        n = self.lib.msa_trim_only_gaps_rows(self.h, None, 0)
        if n <= 0:
            return []
        rows = np.empty(n, dtype=np.int32)
        self.lib.msa_trim_only_gaps_rows(self.h, ptr(rows), n)
        return [int(r) for r in rows]

    # --- instrumentation ---
    def prof_enable(self, on=True):
        # True / 1: every kernel group; 2: the similarity and pair passes only (cheaper: see msa_prof_enable); False / 0: off
        self.lib.msa_prof_enable(self.h, int(on))

    def prof_reset(self):
        self.lib.msa_prof_reset(self.h)

    def prof_get(self, name):
        ms, k = ctypes.c_float(0), ctypes.c_int32(0)
        check(self.lib, self.h, self.lib.msa_prof_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(k)))
        return ms.value, k.value

    PATH_KEYS = ("upload", "pipeline", "sim_kernel", "sim_waves_per_column", "sim_launches", "sim_writes_mdk", "pair_kernel", "pair_waves_per_tile")
    PATH_NAMES = {
        "upload": ("none", "in_place", "linear", "pitched", "packed", "attached", "repitched"),
        "pipeline": ("none", "serial", "one_stream", "two_streams", "compact", "compact_gaps", "compact_sorted"),
        "sim_kernel": ("none", "flat", "lg", "lg_big", "seq", "cols", "lg_pipe", "lg_big_pipe", "lg_xseg", "lg_big_xseg"),
        "pair_kernel": ("none", "pipe", "two_rows", "pipe16"),
    }

    def last_paths(self):
        """Which code path the last upload and the last statistic / trim call of this context took (`msa_debug_last_paths`):
        a dict with the names of include/msastat.h's MSA_PATH_* values (diagnostics; tests/test_gpu_dispatch.py)."""
        out = (ctypes.c_int32 * 8)()
        self.lib.msa_debug_last_paths.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
        check(self.lib, self.h, self.lib.msa_debug_last_paths(self.h, out))
        rec = {}
        for key, v in zip(self.PATH_KEYS, out):
            names = self.PATH_NAMES.get(key)
            rec[key] = names[v] if names and 0 <= v < len(names) else int(v)
        return rec

    def sync(self):
        check(self.lib, self.h, self.lib.msa_ctx_sync(self.h))

    @property
    def stream(self):
        return self.lib.msa_ctx_stream(self.h)


class BatchClosed(RuntimeError):
    """`Batch.trim` on an object another thread has closed in the meantime (`pytrimal_amd.batch` then takes the new one)."""


class _BatchResults(list):
    """`Batch.trim`'s list of results; `packed` is the one uint8 vector every mask of the call is a view of."""
    packed = None


class Batch:
    """One `msa_batch`: native worker threads, each with its own device context, that trim the alignments of a call side
    by side (`msa_trim_batch`: the reference's `ThreadPool.map(trimmer.trim, ...)` without the interpreter)."""

    def __init__(self, device=None, workers=6):
        self.lib = load()
        if device is None:
            device = int(os.environ.get("PYTRIMAL_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = self.lib.msa_device_count()
            if n > 0:
                device %= n
        h = ctypes.c_void_p()
        rc = self.lib.msa_batch_create(int(device), int(workers), ctypes.byref(h))
        if rc != OK:
            raise RuntimeError(f"cannot create a batch of {workers} HIP contexts on device {device}: "
                               f"{self.lib.msa_strerror(rc).decode()} (there is no CPU fallback)")
        self.h, self.device, self.workers = h, int(device), int(workers)
        # one call at a time per batch object (msa_trim_batch refuses a second one): calls from several threads queue up here
        self._lock = threading.Lock()

    def close(self):
        lock = getattr(self, "_lock", None)
        if lock is None:
            return
        with lock:
            if getattr(self, "h", None):
                self.lib.msa_batch_destroy(self.h)
                self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim(self, items):
        """`items`: [(matrix uint8[m, n] (C-contiguous rows, any row stride), indet, TrimParams), ...] ->
        [(keep_res bool[n], keep_seq bool[m], TrimInfo, rc, only_gaps_rows), ...].  The interpreter lock is released for
        the whole call."""
        count = len(items)
        if count == 0:
            return []
        # (one pass over the items, plain Python ints; the arrays of the ABI are filled from the lists at once)
        addr, ms, ns, lds, indets = [], [], [], [], []
        params = (TrimParams * count)()
        u8 = np.dtype(np.uint8)
        for k, (a, indet, p) in enumerate(items):
            shape, strides = a.shape, a.strides
            if a.dtype != u8 or len(shape) != 2 or (shape[1] > 1 and strides[1] != 1):
                raise ValueError("batch items must be uint8 matrices with contiguous rows")
            addr.append(a.__array_interface__["data"][0])
            ms.append(shape[0])
            ns.append(shape[1])
            lds.append(strides[0] if shape[0] > 1 else max(shape[1], 1))
            indets.append(indet)
            params[k] = p
        ms = np.array(ms, dtype=np.int32)
        ns = np.array(ns, dtype=np.int32)
        lds = np.array(lds, dtype=np.int64)
        indets = np.array(indets, dtype=np.uint8)
        data = np.array(addr, dtype=np.uint64)
        sizes = ns.astype(np.int64) + ms
        ends = np.cumsum(sizes)
        starts = ends - sizes
        masks = np.ones(max(int(ends[-1]), 1), dtype=np.uint8)
        base = masks.ctypes.data
        kres = (starts + base).astype(np.uint64)
        kseq = (starts + ns + base).astype(np.uint64)
        infos = (TrimInfo * count)()
        rcs = np.zeros(count, dtype=np.int32)
        out = _BatchResults()
        out.packed = masks  # [residues mask, sequences mask] of every item, side by side (what trim_batch's gather sends)
        with self._lock:
            if not self.h:
                raise BatchClosed("the batch object is closed")
            rc_all = self.lib.msa_trim_batch(self.h, count, ptr(data), ptr(ms), ptr(ns), ptr(lds), ptr(indets), params, ptr(kres), ptr(kseq),
                                             infos, ptr(rcs))
            if rc_all != OK and not rcs.any():  # the call itself was refused: no alignment was looked at
                raise MsaError(rc_all, self.lib.msa_strerror(rc_all).decode())
            flags = masks.view(np.bool_)  # (the library writes 0 / 1: the same bytes as booleans, no copy per alignment)
            for k, pos, n, m, rc in zip(range(count), starts.tolist(), ns.tolist(), ms.tolist(), rcs.tolist()):
                rows = []
                if infos[k].warnings & W_ONLY_GAPS_SEQUENCES:
                    cnt = self.lib.msa_batch_only_gaps_rows(self.h, k, None, 0)
                    if cnt > 0:
                        buf = np.empty(cnt, dtype=np.int32)
                        self.lib.msa_batch_only_gaps_rows(self.h, k, ptr(buf), cnt)
                        rows = [int(r) for r in buf]
                out.append((flags[pos:pos + n], flags[pos + n:pos + n + m], infos[k], rc, rows))
        return out

    def check(self, rc, info):
        """Raise what `Context.trim` would raise for this return code."""
        if rc == E_HIP:
            msgs = [self.lib.msa_batch_last_hip_error(self.h, w).decode() for w in range(self.workers)]
            raise MsaError(rc, self.lib.msa_strerror(rc).decode() + ": " + "; ".join(x for x in msgs if x), info.err)
        check(self.lib, None, rc, info.err)


_tls = threading.local()


def reset_thread_context():
    """Drop this thread's context (the next `trim` creates a new one, which re-reads the MSA_* diagnostic
    switches: the library reads them once per context)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is not None:
        ctx.close()
    _tls.ctx = None


def thread_context():
    """A per-thread context, so that `trim` is re-entrant across threads like the reference
    (``_trimal.pyx:1305-1316``)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None or ctx.h is None:
        ctx = _tls.ctx = Context()
    return ctx
