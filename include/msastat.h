/*
 * msastat.h -- C ABI of libmsastat_hip.so, the MI355X (gfx950) implementation of the MSA
 * statistics hot path behind pytrimal's Trimmer classes.
 *
 * There is no C ABI in the reference: the statistics are C++ virtuals of trimAl's
 * statistics::Manager selected by `ComputePlatform` (reference include/trimal/statistics.pxd:67-96)
 * and reached from `BaseTrimmer.trim` (reference src/pytrimal/_trimal.pyx:1334-1359).  Each entry
 * point below names the reference interface it replaces; INTEGRATION.md shows the binding a
 * pytrimal maintainer would add (a `HIP` ComputePlatform / a Cython `cdef extern` block).
 *
 * Conventions
 *   - plain pointers and sizes only; every matrix is row-major and dense;
 *   - every call returns MSA_OK (0) or a negative error code; msa_strerror() names it;
 *   - outputs are caller-allocated HOST buffers unless the name ends in `_dev`;
 *   - a context owns its device buffers and one HIP stream; contexts are independent, so
 *     `trim` stays re-entrant across threads exactly as in the reference (_trimal.pyx:1305-1316):
 *     use one context per thread / per call;
 *   - nothing here touches Python.  Symbol errors are reported as codes + (row, column, byte)
 *     and turned into ValueError by the host, mirroring reference
 *     src/trimal/source/reportsystem.cpp:44-53.
 */
#ifndef MSASTAT_H
#define MSASTAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msa_ctx msa_ctx;

enum {
    MSA_OK = 0,
    MSA_E_INVALID = -1,          /* bad argument / call order */
    MSA_E_NO_DEVICE = -2,        /* no gfx950 device visible */
    MSA_E_HIP = -3,              /* HIP runtime failure (msa_last_hip_error) */
    MSA_E_NOMEM = -4,
    MSA_E_WINDOW_TOO_BIG = -5,   /* half-window > n/4  (trimAl ErrorCode::WindowTooBig) */
    MSA_E_INCORRECT_SYMBOL = -6, /* residue outside A-Z  -> ValueError (reportsystem.cpp:46-49) */
    MSA_E_UNDEFINED_SYMBOL = -7, /* residue not in the similarity matrix alphabet -> ValueError */
    MSA_E_NOT_IMPLEMENTED = -8,  /* automated2: no surviving pin in the reference checkout */
    MSA_E_NON_ASCII = -9,        /* byte >= 0x80 in the alignment */
    MSA_E_LENGTH_MISMATCH = -10, /* FASTA records of different lengths: not an alignment */
    MSA_E_BAD_RESIDUE = -11      /* FASTA ingest: byte outside the accepted residue set */
};

/* first offending residue of MSA_E_INCORRECT_SYMBOL / MSA_E_UNDEFINED_SYMBOL */
typedef struct {
    int32_t row, col, byte;
} msa_err_detail;

const char *msa_strerror(int code);
int msa_device_count(void);              /* number of visible HIP devices (0 on a CPU-only host) */
const char *msa_last_hip_error(const msa_ctx *ctx);

/* ---- context ------------------------------------------------------------------------------ */
int msa_ctx_create(int device, msa_ctx **out);
void msa_ctx_destroy(msa_ctx *ctx);
/* the context's hipStream_t, so that a caller can time / order work on it */
void *msa_ctx_stream(msa_ctx *ctx);
int msa_ctx_sync(msa_ctx *ctx);

/* ---- alignment upload: replaces the `std::string *sequences` rows of trimAl's Alignment
 *      (reference include/trimal/alignment.pxd:22) as the statistics' input ------------------ */
/* m row pointers of n raw residue bytes each; indet = 'X' (protein) or 'N' (nucleotides),
 * i.e. what Alignment::getAlignmentType selects (_trimal.pyx:759-763,891). */
int msa_upload_rows(msa_ctx *ctx, const uint8_t *const *rows, int32_t m, int32_t n, uint8_t indet);
/* one packed row-major host buffer, leading dimension ld >= n */
int msa_upload_packed(msa_ctx *ctx, const uint8_t *rowmajor, int32_t m, int32_t n, int64_t ld, uint8_t indet);
/* The same without waiting for the copy: the rows must stay valid and unchanged until the next call on this context that
 * returns results (msa_trim, msa_gaps, ...) or uploads again: each of them returns with the copy complete, msa_trim also
 * when the trim itself had nothing to read.  What a binding that holds the alignment anyway (pytrimal's Alignment
 * object) calls right in front of msa_trim.  (Rows of less than 96 KB are packed into the context's own page-locked memory
 * before either upload call returns, and the kernels read them there: nothing is left in flight.) */
int msa_upload_packed_async(msa_ctx *ctx, const uint8_t *rowmajor, int32_t m, int32_t n, int64_t ld, uint8_t indet);
/* Page-lock the caller's rows so that uploads of them are one DMA copy at the link's rate, straight from where they lie
 * (no staging, no packing).  Worth it for rows that are uploaded more than once; the range must be unregistered before
 * its memory is released.  msa_upload_packed recognises page-locked memory by itself. */
int msa_host_register(const void *rows, size_t bytes);
int msa_host_unregister(const void *rows);
/* residue matrix already resident in device memory (not copied, must outlive its use) */
int msa_attach_device(msa_ctx *ctx, const void *rowmajor_dev, int32_t m, int32_t n, int64_t ld, uint8_t indet);

/* ---- a1  statistics::Gaps::CalculateVectors (statistics.pxd:18-21) via
 *          Manager::calculateGapStats (statistics.pxd:90) ------------------------------------ */
/* gaps_out[n]: '-' count per column (gapsInColumn).  indet_out[n] (nullable): count of the
 * indetermination symbol per column (used by the overlap pass). */
int msa_gaps(msa_ctx *ctx, int32_t *gaps_out, int32_t *indet_out);

/* The (windowed) gap vector of the current alignment if its counts are already on the host -- i.e. a trim or msa_gaps
 * has fetched them -- without any device work: returns MSA_OK and fills out[n] (half_window as in msa_window_i32, 0 =
 * the plain counts), or 1 when the context holds no host copy (nothing is written).  This is what lets
 * TrimmedAlignment.terminal_only (Cleaner::removeOnlyTerminal, cleaner.pxd:38) read the gap statistics the trim
 * computed, as trimAl's trimmed alignment shares the statistics object of its source (statistics.pxd:47-51). */
int msa_gaps_cached(msa_ctx *ctx, int32_t half_window, int32_t *out);

/* ---- a2/a5  pairwise identity counts: Cleaner::calculateSeqIdentity (cleaner.pxd:42) and
 *             Similarity::calculateMatrixIdentity (statistics.pxd:56) share them -------------- */
/* hit[m*m], dst[m*m] (either nullable): symmetric, diagonal 0. */
int msa_pair_counts(msa_ctx *ctx, uint32_t *hit, uint32_t *dst);
/* ident[m*m] = (float)hit/dst  -> Alignment::identities (alignment.pxd:27);
 * w[m*m] = 1 - (float)hit/dst  -> Similarity::matrixIdentity (statistics.pxd:49).
 * Either may be NULL: the matrix then stays on the device for the later passes. */
int msa_identities(msa_ctx *ctx, float *ident, float *w);
/* the two means of Cleaner::selectMethod (cleaner.pxd:16), accumulated in the reference's
 * order on the device; returns them without moving the m*m matrix to the host. */
int msa_identity_stats(msa_ctx *ctx, float *avg_seq, float *max_seq);

/* ---- a3  statistics::Similarity::calculateVectors (statistics.pxd:55) ----------------------- */
/* vhash[26]: 'A'..'Z' -> matrix index or -1; dist[npos*npos]: Euclidean distance matrix
 * (similarity_matrix.pxd:4-7, built as _trimal.pyx:1987-1997); gaps_windowed[n] (nullable):
 * the vector used for the ">= 80 % gaps -> 0" cut, default = this context's gap counts.
 * mdk_out[n]: MDK; q_out[n] (nullable): num/den before the exp.  Bit-exact with the reference's
 * sequential float32 accumulation order. */
int msa_similarity(msa_ctx *ctx, const int32_t *vhash, const float *dist, int32_t npos,
                   const int32_t *gaps_windowed, float *mdk_out, float *q_out, msa_err_detail *detail);

/* ---- a6  Cleaner::calculateSpuriousVector (cleaner.pxd:27) --------------------------------- */
int msa_overlap(msa_ctx *ctx, float residue_overlap, float *spurious_out);

/* ---- a4/a9/a10  host selection logic (trimAl Cleaner / Gaps / Similarity cut points).
 *      Pure host functions: usable without a device. ------------------------------------------ */
int msa_window_i32(const int32_t *v, int32_t n, int32_t half_window, int32_t *out);
int msa_window_f32(const float *v, int32_t n, int32_t half_window, float *out);
/* Gaps::calcCutPoint / calcCutPoint2ndSlope */
double msa_gaps_cutpoint(const int32_t *gaps, int32_t m, int32_t n, float base_line, float gap_threshold);
int32_t msa_gaps_cutpoint_2nd_slope(const int32_t *gaps, int32_t m, int32_t n);
/* Similarity::calcCutPoint (statistics.pxd:61) */
double msa_similarity_cutpoint(const float *mdk_windowed, int32_t n, float base_line, float sim_threshold);
/* Cleaner::cleanByCutValueOverpass / FallBehind / OverpassOrEquals (cleaner.pxd:17-19);
 * keep[n]: 1 kept, 0 dropped */
int msa_clean_gaps(const int32_t *gaps_w, int32_t n, double cut, float base_line, uint8_t *keep);
int msa_clean_similarity(const float *mdk_w, int32_t n, float cut, float base_line, uint8_t *keep);
int msa_clean_both(const int32_t *gaps_w, const float *mdk_w, int32_t n, double cut_gaps, float cut_sim,
                   float base_line, uint8_t *keep);
/* Cleaner::cleanCombMethods + cleanStrict (cleaner.pxd:20,30); variable = strictplus.
 * gaps[n] feeds the 2nd-slope histogram, gaps_w[n] / mdk_w[n] are the windowed vectors. */
int msa_clean_strict(const int32_t *gaps, const int32_t *gaps_w, const float *mdk_w, int32_t m, int32_t n,
                     int32_t variable, uint8_t *keep, int32_t *gap_cut_out, float *sim_cut_out);
/* Cleaner::selectMethod decision from the two means: 1 = gappyout, 2 = strict */
int32_t msa_select_method(float avg_seq, float max_seq, int32_t m);
/* Cleaner::calculateRepresentativeSeq / getCutPointClusters (cleaner.pxd:35,44);
 * lengths[m] = ungapped sequence lengths; keep_seq[m] */
int msa_representatives(const float *ident, const int32_t *lengths, int32_t m, float max_identity,
                        uint8_t *keep_seq, int32_t *n_clusters);
float msa_cutpoint_clusters(const float *ident, const int32_t *lengths, int32_t m, int32_t clusters);

/* ---- alignment ingest: what `Alignment.load(file, "fasta")` needs from trimAl's FormatManager
 *      (format_handling.pxd:11-32, driven by _trimal.pyx:517-601): FASTA text -> dense row-major
 *      residue matrix, with no per-sequence host objects.  Pure host functions.
 *      msa_fasta_scan: number of records and the residue count of the first one (whitespace removed).
 *      msa_fasta_fill: writes matrix[m][n], the offset/length of every record's name (first field
 *      after '>') inside `data`; `valid` is an optional 256-entry table of acceptable residue bytes.
 *      Returns MSA_E_LENGTH_MISMATCH (detail->row = record, detail->col = its length) or
 *      MSA_E_BAD_RESIDUE (detail->row, ->col, ->byte). */
int msa_fasta_scan(const uint8_t *data, int64_t len, int32_t *m_out, int32_t *n_out);
int msa_fasta_fill(const uint8_t *data, int64_t len, int32_t m, int32_t n, uint8_t *matrix, int64_t *name_off,
                   int32_t *name_len, const uint8_t *valid, msa_err_detail *detail);
/* The same pair for Clustal text (`Alignment.load(file, "clustal")`, reference _trimal.pyx:517-601,834-850 and
 * trimAl's clustal_state, format_handling.pxd:11-32): interleaved blocks of `name residues [count]` lines behind
 * a header line; blank lines and conservation lines (leading blank) are skipped; the sequences are the names of
 * the first block, in that order.  Same outputs and error codes as the FASTA pair. */
int msa_clustal_scan(const uint8_t *data, int64_t len, int32_t *m_out, int32_t *n_out);
int msa_clustal_fill(const uint8_t *data, int64_t len, int32_t m, int32_t n, uint8_t *matrix, int64_t *name_off,
                     int32_t *name_len, const uint8_t *valid, msa_err_detail *detail);

/* ---- whole trim: trimAlManager::clean_alignment (manager.pxd:88) as configured by the four
 *      `_configure_manager` methods (_trimal.pyx:1479-1497,1651-1659,1766-1769,1859-1862) ------ */
enum {
    MSA_METHOD_NONE = 0, MSA_METHOD_STRICT = 1, MSA_METHOD_STRICTPLUS = 2, MSA_METHOD_GAPPYOUT = 3,
    MSA_METHOD_NOGAPS = 4, MSA_METHOD_NOALLGAPS = 5, MSA_METHOD_AUTOMATED1 = 6,
    MSA_METHOD_AUTOMATED2 = 7, MSA_METHOD_NODUPLICATESEQS = 8
};

typedef struct {
    int32_t method;                  /* MSA_METHOD_* (AutomaticTrimmer) */
    float gap_threshold;             /* trimAlManager::gapThreshold = 1 - kwarg, or -1 */
    int32_t gap_absolute_threshold;  /* or -1 */
    float similarity_threshold;      /* or -1 */
    float conservation_percentage;   /* trimAlManager::conservationThreshold, or -1 */
    int32_t window, gap_window, similarity_window; /* or -1 */
    float residue_overlap, sequence_overlap;       /* OverlapTrimmer, or -1 */
    int32_t clusters;                /* RepresentativeTrimmer, or -1 */
    float max_identity;              /* RepresentativeTrimmer, or -1 */
    /* similarity matrix (Manager::setSimilarityMatrix, statistics.pxd:89) */
    const int32_t *vhash;            /* [26] */
    const float *dist;               /* [npos*npos] */
    int32_t npos;
} msa_trim_params;

typedef struct {
    int32_t selected_method;  /* automated1: 1 gappyout / 2 strict; else 0 */
    float avg_seq, max_seq;   /* selectMethod means (automated1) */
    int32_t gap_cut;          /* calcCutPoint2ndSlope (gappyout/strict/strictplus) */
    float sim_cut;            /* cleanCombMethods */
    int32_t kept_residues, kept_sequences;
    msa_err_detail err;
    float ms_device;          /* device time of the statistics kernels of this call (HIP events) */
    /* Warnings of the call, MSA_W_* bits: what trimAl would hand to its report manager as a WarningCode and the
     * reference turns into a Python RuntimeWarning (reference src/trimal/source/reportsystem.cpp:132-173).  The
     * host binding raises one RuntimeWarning per bit; warn_row is the first sequence concerned (or -1). */
    uint32_t warnings;
    int32_t warn_row;
} msa_trim_info;

enum {
    /* the trimming left sequences composed only of gaps, which were removed (Cleaner::removeAllGapsSeqsAndCols,
     * cleaner.pxd:40; trimAl WarningCode RemovingOnlyGapsSequence [R]) */
    MSA_W_ONLY_GAPS_SEQUENCES = 1u << 0,
    /* every column was removed: the trimmed alignment is empty */
    MSA_W_NO_COLUMNS_LEFT = 1u << 1,
    /* a pair of sequences shares no column in which either holds a residue: its identity is undefined and taken
     * as 0 (hit / dst with dst = 0, Cleaner::calculateSeqIdentity, cleaner.pxd:42) */
    MSA_W_UNDEFINED_IDENTITY = 1u << 2
};

/* keep_res[n], keep_seq[m]: the saveResidues / saveSequences masks (alignment.pxd:29-30) as
 * 0/1 bytes, i.e. TrimmedAlignment.residues_mask / sequences_mask (_trimal.pyx:1085-1121). */
int msa_trim(msa_ctx *ctx, const msa_trim_params *params, uint8_t *keep_res, uint8_t *keep_seq,
             msa_trim_info *info);

/* ---- batches of independent alignments: the reference's `ThreadPool.map(trimmer.trim, alignments)` (README.md:136-152;
 *      possible there because trim releases the interpreter lock, _trimal.pyx:1334-1359) as one native call --------------
 * A batch object owns `workers` threads, each with its own context (device buffers + streams) on `device`.
 * msa_trim_batch trims `count` independent alignments: alignment k = m[k] rows of n[k] residue bytes, row-major at data[k]
 * with leading dimension ld[k] >= n[k], indetermination symbol indet[k], parameters params[k] (one entry per alignment:
 * the similarity matrix follows the alignment's type); keep_res[k] (n[k] bytes), keep_seq[k] (m[k] bytes), info[k]
 * (nullable array) and rc[k] receive what msa_trim would return for it.  The workers take the alignments largest first;
 * each uploads its alignment without waiting for the copy (the rows must stay valid until the call returns) and trims it,
 * so uploads, kernels and the host selection logic of different alignments overlap on the device.  Returns MSA_OK or the
 * first non-zero rc[k].  One call at a time per batch object; different batch objects are independent. */
typedef struct msa_batch msa_batch;
int msa_batch_create(int device, int32_t workers, msa_batch **out);
void msa_batch_destroy(msa_batch *b);
int32_t msa_batch_workers(const msa_batch *b);
int msa_trim_batch(msa_batch *b, int32_t count, const uint8_t *const *data, const int32_t *m, const int32_t *n,
                   const int64_t *ld, const uint8_t *indet, const msa_trim_params *params, uint8_t *const *keep_res,
                   uint8_t *const *keep_seq, msa_trim_info *info, int32_t *rc);
/* the rows behind MSA_W_ONLY_GAPS_SEQUENCES of alignment k of the last call (as msa_trim_only_gaps_rows) */
int msa_batch_only_gaps_rows(msa_batch *b, int32_t k, int32_t *rows, int32_t cap);
const char *msa_batch_last_hip_error(const msa_batch *b, int32_t worker);

/* The sequences the last msa_trim of this context removed because the trimming left them with gaps only (the rows behind
 * MSA_W_ONLY_GAPS_SEQUENCES; trimAl reports each of them): writes at most cap indices, returns their number (>= 0)
 * or a negative error code. */
int msa_trim_only_gaps_rows(msa_ctx *ctx, int32_t *rows, int32_t cap);

/* ---- instrumentation ---------------------------------------------------------------------- */
/* device time (ms, HIP events on the context stream) and launch count of the named kernel
 * family since the last msa_prof_reset: "gaps", "prep", "pairs", "idstats", "encode", "sim",
 * "overlap".  Returns MSA_E_INVALID for an unknown name. */
int msa_prof_get(msa_ctx *ctx, const char *kernel, float *ms_total, int32_t *launches);
void msa_prof_reset(msa_ctx *ctx);
/* enable: 0 off; 1 an event pair around every kernel group; 2 around the similarity and the pair pass only (each pair
 * costs a few microseconds of queue time per trim) */
void msa_prof_enable(msa_ctx *ctx, int enable);
/* Kernel launches the context's last similarity pass issued (one, or one per six rounds from 1800 rows on): what a
 * profiler lists per pass.  Diagnostics; bench.py reports it beside the pass's time. */
int msa_debug_sim_launches(msa_ctx *ctx);
/* Which code path the context's last upload and its last statistic / trim call took under the settings in force
 * (diagnostics, like msa_debug_sim_launches: the default dispatch by shape is what ships, and
 * tests/test_gpu_dispatch.py asserts it shape by shape).  out[0] how the rows reached the kernels (MSA_PATH_UPLOAD_*),
 * out[1] the launch sequence (MSA_PATH_PIPE_*), out[2] the similarity kernel (MSA_PATH_SIM_*), out[3] its waves per column,
 * out[4] its launches per pass, out[5] 1 when it wrote MDK and Q itself, out[6] the pair kernel (MSA_PATH_PAIRS_*), out[7] its
 * waves per tile.  Entries of a stage the call did not run are 0. */
enum {
    MSA_PATH_UPLOAD_NONE = 0,
    MSA_PATH_UPLOAD_IN_PLACE = 1, /* packed into the context's pinned staging area, read there by the kernels (no copy) */
    MSA_PATH_UPLOAD_LINEAR = 2,   /* rows already at the device pitch: one linear copy */
    MSA_PATH_UPLOAD_PITCHED = 3,  /* one pitched copy straight from the caller's rows */
    MSA_PATH_UPLOAD_PACKED = 4,   /* packed into pinned pieces, each sent as soon as it is packed */
    MSA_PATH_UPLOAD_ATTACHED = 5, /* msa_attach_device */
    MSA_PATH_UPLOAD_REPITCHED = 6 /* a contiguous pageable matrix of odd-sized rows: one linear copy, then a kernel lays the rows out at the device pitch */
};
enum {
    MSA_PATH_PIPE_NONE = 0,
    MSA_PATH_PIPE_SERIAL = 1,       /* msa_similarity's own launch sequence (also msa_trim with MSA_PIPELINE=0) */
    MSA_PATH_PIPE_ONE_STREAM = 2,   /* msa_trim's pipeline, everything on the context's stream */
    MSA_PATH_PIPE_TWO_STREAMS = 3,  /* ... codes, lists and row totals on the side stream beside the pair pass */
    MSA_PATH_PIPE_COMPACT = 4,      /* the compact pipeline of a small alignment: front, pairs, [identity], similarity */
    MSA_PATH_PIPE_COMPACT_GAPS = 5, /* its front kernel alone (a trim that needs the gap statistics only) */
    MSA_PATH_PIPE_COMPACT_SORTED = 6 /* the compact pipeline with the columns dealt to the similarity kernel by weight (from 513
                                        sequences on): the host sorts them behind the front kernel's event, while the pair pass runs */
};
enum {
    MSA_PATH_SIM_NONE = 0,
    MSA_PATH_SIM_FLAT = 1,       /* the column's pairs as one sequence, two waves per column */
    MSA_PATH_SIM_LG = 2,         /* a wave (or out[3] waves of a workgroup) per column, byte offsets in the lists */
    MSA_PATH_SIM_LG_BIG = 3,     /* ... row indices in the lists (beyond 32768 rows) */
    MSA_PATH_SIM_SEQ = 4,        /* the plain sequential kernel (MSA_SIM_KERNEL=seq) */
    MSA_PATH_SIM_COLS = 5,       /* a lane per column (groups of small alignments in msa_trim_batch) */
    MSA_PATH_SIM_LG_PIPE = 6,    /* out[3] loop waves and a service wave per column, no barrier (few columns, up to 9000 rows), byte offsets */
    MSA_PATH_SIM_LG_BIG_PIPE = 7, /* ... row indices in the lists */
    MSA_PATH_SIM_LG_XSEG = 8,    /* tall alignments of few columns: wave w of every column on XCD w, increments and sums through memory */
    MSA_PATH_SIM_LG_BIG_XSEG = 9 /* ... row indices in the lists */
};
enum { MSA_PATH_PAIRS_NONE = 0, MSA_PATH_PAIRS_PIPE = 1 /* one row j per lane */, MSA_PATH_PAIRS_TWO_ROWS = 2 /* two rows j per lane */,
       MSA_PATH_PAIRS_PIPE16 = 3 /* one row j per lane, sixteen rows i per tile (up to 1024 sequences) */ };
int msa_debug_last_paths(msa_ctx *ctx, int32_t out[8]);
/* 1 when the library honours its MSA_* diagnostic switches in this process: the build has them (not -DMSA_NO_DIAGNOSTICS) and
 * the environment variable MSA_DIAGNOSTICS is set.  Without it every context and batch object takes the default dispatch
 * whatever else the environment holds (MSA_TRACE, which only prints, is always read).  Needs no device. */
int msa_debug_switches_enabled(void);

#ifdef __cplusplus
}
#endif
#endif /* MSASTAT_H */
