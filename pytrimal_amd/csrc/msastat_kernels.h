// Launch wrappers of msastat_kernels.hip (internal; the public surface is include/msastat.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msak {

// Diagnostic switches (environment variables MSA_*), read ONCE when a context is created and handed to the
// launch wrappers through a thread-local pointer for the duration of an API call (contexts are per thread).
struct Tuning {
    int sim_kernel = 0;        // MSA_SIM_KERNEL=seq: the plain sequential similarity kernel (1) instead of the binade-exact one (tests)
    int sim_mode = 0;          // MSA_SIM_MODE=64: cycle stamps of the similarity kernel (tools/bx_stamps.py)
    int lg_r0 = -1;            // MSA_LG_R0: rows evaluated in order before the first round (-1 = default)
    int lg_big = 0;            // MSA_LG_BIG=1: row indices instead of byte offsets in the lists at any size (tests; default: beyond 32768 rows)
    int device_clusters = -1;  // MSA_DEVICE_CLUSTERS: -1 unset (size heuristic), 0 host, 1 device
    int trace = 0;             // MSA_TRACE
    int upload_piece_kb = 1024;  // MSA_UPLOAD_PIECE_KB: rows are packed and sent in pieces of this size (0: one copy after packing everything)
    int upload_direct = 1;     // MSA_UPLOAD_DIRECT=0: every upload through the packed pinned staging pieces (diagnostics)
    int pipeline = 1;          // MSA_PIPELINE: 0 msa_trim waits for the gap counts / identity statistics before it enqueues the similarity
                               // pass; 1 pipelined (side stream for large alignments); 2 pipelined, never a side stream; 3 always
    int lg_rounds = -1;        // MSA_LG_ROUNDS: rounds of the similarity kernel per launch (-1: by size, 0: one launch; tests: any)
    int lg_split = 0;          // MSA_LG_SPLIT: waves of a workgroup that share one column of the similarity kernel (0: by shape; tests: 1, 2, 4, 8, 16)
    int mdk_host = 0;          // MSA_MDK_HOST=1: the device hands every exponential of the MDK values to the host (tests: both paths agree bit for bit)
};
Tuning tuning_from_env();
void set_tuning(const Tuning *t);  // thread-local; nullptr = defaults
const Tuning *current_tuning();    // what set_tuning last received on this thread
const Tuning &tuning();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of on every launch
int set_max_lds_once(const void *kernel, int bytes);

constexpr int PAIR_TI = 8;   // rows "i" per wave in pair_counts (wave-uniform, SGPR operands)

void launch_prep_planes(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, uint32_t *planes,
                        int nchunk, int m_pad, int *err_flag);
void launch_gap_counts(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, int32_t *gaps,
                       int32_t *indets);
void launch_pair_counts(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int ldw, uint32_t *hit,
                        uint32_t *dst, float *ident, float *wmat, float *wlow, int *undef_flag);
int planes_total();  // planes in the plane array (seven symbol bits + validity)
// the similarity kernel and its layouts (msastat_simx.hip)
int64_t bx_ldk(int m);
int bx_cols_pad(int n);
size_t bx_wlow_rows(int m);
bool lg_big(int m, int ldw);  // the lists hold row indices instead of 32-bit byte offsets (beyond 32768 rows)
void launch_sim_encode_cm(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut,
                          const int32_t *gaps_w, uint8_t *codeT, unsigned long long *err_key);
void launch_bx_compact(hipStream_t s, const uint8_t *codeT, int m, int n, int ldw, int npos, uint32_t *voff, uint16_t *vtrow,
                       int32_t *nvalid);
void launch_identity_stats(hipStream_t s, const float *ident, int m, int ldw, float *row_avg, float *row_max,
                           float *out2, float *row_min = nullptr, int *gate = nullptr);
// One alignment as the similarity kernel sees it (device pointers).  By value for a single alignment, as a table in
// device memory for a batch (msa_trim_batch: every column of every alignment of a shard in one grid).
struct LgAlign {
    const uint32_t *voff;    // compacted lists of every column's valid rows: W row offset (or index) ...
    const uint16_t *vtrow;   // ... and table row offset
    const int32_t *nvalid;   // entries per column
    const uint8_t *codeT;    // column-major codes
    const float *wlow, *wup, *wbar;
    float *num_out, *den_out;
    float *state;            // lg_state_floats(n) floats: per-column state between launches (null: always one launch)
    const int *gate;         // device word that, when non-zero, turns the alignment's columns into no-ops (or null)
    const int32_t *cols;     // single alignment: the columns to evaluate (ncols entries)
    int64_t ldk;
    int32_t m, n, ldw, ncols;
};
int launch_similarity_lg(hipStream_t s, const LgAlign &one, int npos, const void *tab, int cus, int *launches_out);
int launch_similarity_lg_batch(hipStream_t s, const LgAlign *table, const int32_t *colprefix, int K, int ncols_total, int max_m, int npos,
                               const void *tab, bool with_state, int *launches_out);
int lg_split(int m, int ncols, int cus);  // waves per column the launcher picks
int lg_rounds_per_launch(int m);          // rounds per launch (0: everything in one launch)
size_t lg_state_floats(int n);
// One alignment of a batch as the batched kernels see it (device pointers): msa_trim_batch uploads a table of these and,
// per kernel family, the prefix sums of its blocks per alignment; ONE launch per family serves the whole shard.
struct BAlign {
    const uint8_t *raw;       // [m][ld] residues
    const uint8_t *fetch_src; // page-locked caller rows the device reads by itself (fetch_rows_batch_kernel), or null: a copy brings them
    int64_t fetch_ld;         // their row stride
    int64_t ld, ldk;
    uint32_t *planes;
    int32_t *gaps, *indets, *rowtot;
    int32_t *flags;           // 16 words, the layout of a context's state block (ST_*): non-ASCII flag, undefined-identity flag,
                              // first-bad-residue key (2), the two selectMethod means (2), the gate
    float *ident, *w, *wlow, *wbar, *row_avg, *row_max;
    uint8_t *codeT;
    uint8_t *codeR;           // row-major codes [m][ld] (groups of small alignments: similarity_cols, a lane per column)
    uint32_t *off;
    uint16_t *trow;
    int32_t *nvalid;
    float *simnum, *simden, *mdk;  // mdk [n] followed by Q [n]
    int32_t m, n, nchunk, m_pad, ldw, ncols_pad;
    uint32_t indet4;
    int32_t gated;            // automated1: the identity statistics decide on the device whether the similarity values are needed
};
int pair_tiles_pipe(int m, int m_pad);    // tiles of the pair pass in its one-row-per-lane regime
bool pair_pipe_regime(int m, int m_pad);  // ... which launch_pair_counts picks for this shape (a batch holds no other)
void launch_fetch_rows_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_gap_counts_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_row_nongap_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_prep_planes_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_pair_counts_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks, int min_nchunk);  // min_nchunk: over the group
void launch_sim_finish_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_w_row_means_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_identity_stats_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_sim_encode_rm_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks, const uint8_t *lut);
void launch_similarity_cols_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int total, const void *tab);
void launch_sim_lists_batch(hipStream_t s, const BAlign *table, const int32_t *prefix_encode, int blocks_encode, const int32_t *prefix_compact,
                            int blocks_compact, int K, const uint8_t *lut, int npos);
void launch_w_row_means(hipStream_t s, const float *wup, int m, int ldw, float *wbar);
int launch_similarity_seq(hipStream_t s, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, const float *wup, int ldw,
                          const void *tab, float *num_out, float *den_out);
void launch_sim_finish(hipStream_t s, const float *num, const float *den, const int32_t *gaps_w, int m, int n,
                       float *q_out, float *mdk_out);
void launch_overlap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                    const int32_t *indets, int need, uint32_t *col_ok, int nchunk, int32_t *good);
void launch_row_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_res,
                       int32_t *row_nongap);
void launch_col_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_seq,
                       int32_t *col_nongap);
void launch_row_digest(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, int32_t *lengths,
                       unsigned long long *hashes);
size_t cluster_adj_words(int m);
size_t cluster_adj_buffer_words(int m);  // what launch_cluster's `adj` must hold
int launch_cluster(hipStream_t s, const float *ident, int ldw, const int32_t *seq_at, int m, float thr, uint32_t *adj,
                   uint8_t *keep_seq, int32_t *count);
void launch_rows_equal(hipStream_t s, const uint8_t *raw, int n, int64_t ld, const int32_t *pairs, int npairs,
                       int32_t *equal);

}  // namespace msak
