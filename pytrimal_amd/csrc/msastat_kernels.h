// Launch wrappers of msastat_kernels.hip (internal; the public surface is include/msastat.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msak {

// Diagnostic switches (environment variables MSA_*; honoured only when MSA_DIAGNOSTICS is set: tuning_from_env), read ONCE when a context is created and handed to the
// launch wrappers through a thread-local pointer for the duration of an API call (contexts are per thread).
struct Tuning {
    int sim_kernel = 0;        // MSA_SIM_KERNEL=seq: the plain sequential similarity kernel (1) instead of the binade-exact one (tests)
    int sim_mode = 0;          // MSA_SIM_MODE=64: cycle stamps of the similarity kernel (tools/bx_stamps.py)
    int lg_r0 = -1;            // MSA_LG_R0: rows evaluated in order before the first round (-1 = default)
    int lg_big = 0;            // MSA_LG_BIG=1: row indices instead of byte offsets in the lists at any size (tests; default: beyond 32768 rows)
    int device_clusters = -1;  // MSA_DEVICE_CLUSTERS: -1 unset (size heuristic), 0 host, 1 device
    int trace = 0;             // MSA_TRACE
    int upload_direct = 1;     // MSA_UPLOAD_DIRECT=0: every upload through the packed pinned staging pieces (diagnostics)
    int pipeline = 1;          // MSA_PIPELINE: 0 msa_trim waits for the gap counts / identity statistics before it enqueues the similarity
                               // pass; 1 pipelined (side stream for large alignments); 2 pipelined, never a side stream; 3 always
    int lg_rounds = -1;        // MSA_LG_ROUNDS: rounds of the similarity kernel per launch (-1: by size, 0: one launch; tests: any)
    int lg_split = 0;          // MSA_LG_SPLIT: waves of a workgroup that share one column of the similarity kernel (0: by shape; tests: 1, 2, 4, 8, 16)
    int compact = 1;           // MSA_COMPACT=0: small alignments through the ordinary launch sequence (tests, A/B); 1: the compact pipeline
    int flat_max_m = 128;      // MSA_FLAT_MAX_M: sequences up to which the compact pipeline runs the flat similarity kernel (0: never)
    int flat_u = 0;            // MSA_FLAT_U: terms per lane and scan of the flat similarity kernel (0: by size; A/B: 4, 8, 16)
    int zerocopy_kb = 96;      // MSA_ZEROCOPY_KB: rows up to this size stay in pinned host memory and the kernels read them over the link
    int mdk_host = 0;          // MSA_MDK_HOST=1: the device hands every exponential of the MDK values to the host (tests: both paths agree bit for bit)
    int front_cw = 0;          // MSA_FRONT_CW: columns per column block of the compact pipeline's front kernel (0: by size; 64: the round-4 kernel; 16, 32)
    int front_nt = 0;          // MSA_FRONT_NT: threads per block of the narrow front kernel (0: by size; 256, 512, 1024)
    int front_from_m = 0;      // MSA_FRONT_FROM_M: sequences from which on the narrow front kernel runs (0: default)
    int front_xcd = 1;         // MSA_FRONT_XCD=0: the narrow column blocks in their own order
    int pair_ti = 0;           // MSA_PAIR_TI: rows i per tile of the pair pass (0: by size; 8, 16)
    int pair_k = 0;            // MSA_PAIR_K: waves per tile of the pair pass (0: by size)
    int lists_fused = 1;       // MSA_LISTS_FUSED=0: codes and lists by sim_encode_cm + bx_compact at every size (tests, A/B)
    int lg_parts = 2;          // MSA_LG_PARTS: staggered parts of such a pass (2, 3, 4)
    int lg_pipe = 1;           // MSA_LG_PIPE: 1 the waves of a split column run ahead of the stitching where it pays (lg_pipe), 0 never, 2 whenever a column is split (tests)
    int lg_xseg = 1;           // MSA_LG_XSEG: 1 tall alignments of few columns with an XCD per segment (lg_xseg), 0 never, 2 whenever the columns are few enough (tests)
    int lg_xseg_kx = 0;        // MSA_LG_XSEG_KX: loop waves of a column per XCD of that kernel (0: by the number of columns; 1, 2, 4)
    int lg_pipe_k = 1;         // MSA_LG_PIPE_K: sub-rounds per round of that kernel (1 .. 8)
    int lg_halves = 1;         // MSA_LG_HALVES: 1 the columns of a multi-launch similarity pass as two staggered halves where it pays (lg_halves), 0 never, 2 always (tests)
};
Tuning tuning_from_env();
bool diagnostics_enabled();  // MSA_DIAGNOSTICS is set (and the build has the switches): the other MSA_* variables are read
void set_tuning(const Tuning *t);  // thread-local; nullptr = defaults
const Tuning *current_tuning();    // what set_tuning last received on this thread
const Tuning &tuning();
// What the launch wrappers last chose on this thread (diagnostics: msa_debug_last_paths copies it into the context).
struct LaunchNote {
    int sim_kind = 0;     // 0 none, 1 flat, 2 wave-per-column (byte offsets), 3 wave-per-column (row indices), 4 sequential, 5 lane-per-column (batches), 6 / 7: 2 / 3 with the waves of a split column running ahead of the stitching, 8 / 9: 2 / 3 with an XCD per segment
    int lg_split = 0;     // waves of a workgroup per column (1: a wave per column)
    int lg_launches = 0;  // launches of the pass
    int lg_fin = 0;       // the kernel writes MDK and Q itself (the compact pipeline)
    int pair_kind = 0;    // 0 none, 1 one row j per lane (software pipeline), 2 two rows j per lane, 3 sixteen rows i per tile
    int pair_waves = 0;   // waves per tile
};
LaunchNote &launch_note();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of on every launch
int set_max_lds_once(const void *kernel, int bytes);

constexpr int PAIR_TI = 8;   // rows "i" per wave in pair_counts (wave-uniform, SGPR operands)

void launch_prep_planes(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, uint32_t *planes,
                        int nchunk, int m_pad, int *err_flag);
void launch_gap_counts(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, int32_t *gaps,
                       int32_t *indets);
void launch_pair_counts(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int ldw, uint32_t *hit,
                        uint32_t *dst, float *ident, float *wmat, float *wlow, int *undef_flag, uint32_t *wsum = nullptr);
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
// the two above in one pass (up to ~10 000 rows; false: not applicable, run the two)
bool launch_sim_lists_fused(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, const int32_t *gaps_w, uint8_t *codeT,
                            int ldw, int npos, uint32_t *voff, uint16_t *vtrow, int32_t *nvalid, unsigned long long *err_key);
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
    const uint32_t *wsum;    // (with mdk_out) the pair pass's fixed-point row sums of W instead of wbar
    float *num_out, *den_out;
    float *state;            // lg_state_floats(n) floats: per-column state between launches (null: always one launch)
    const int *gate;         // device word that, when non-zero, turns the alignment's columns into no-ops (or null)
    const int32_t *cols;     // single alignment: the columns to evaluate (ncols entries); null: the columns 0 .. ncols-1 themselves
    float *mdk_out, *q_out;  // the compact pipeline: the wave that finishes a column also writes its MDK and Q (else null: sim_finish)
    int64_t ldk;
    int32_t m, n, ldw, ncols;
    int32_t mdk_host;        // (with mdk_out) MSA_MDK_HOST: every exponential goes to the host as a NaN
};
struct LgSide {  // further streams and events of the caller's: the columns of a multi-launch pass as `parts` staggered parts (lg_halves)
    int parts;           // 2 (3, 4: MSA_LG_PARTS); part 0 runs on the launcher's own stream
    hipStream_t s[3];    // the streams of the parts 1 ..
    hipEvent_t fork, join[3];
};
struct LgXsegBufs {  // device memory of the XCD-per-segment kernel of tall alignments (similarity_lg_xseg_kernel: lg_xseg, lg_xseg_bytes, lg_xseg_bufs)
    float *dep;     // [columns][4 rounds][8 waves][4][64]: the increments on their way to the column's service wave
    float *eslots;  // [columns][4][4]: the exact sums and ratios in front of a round
    int *flags;     // [columns][8]
    int *err;       // [0] a wait timed out: the pass is void  [1] the gate of the launches that redo it
};
bool lg_xseg(int m, int ncols, int cus);
size_t lg_xseg_bytes(int ncols);
LgXsegBufs lg_xseg_bufs(void *base, int ncols);
int lg_parts();  // parts of a pass that lg_halves says is to be staggered (2; MSA_LG_PARTS: 3, 4)
int launch_similarity_lg(hipStream_t s, const LgAlign &one, int npos, const void *tab, int cus, int *launches_out, const LgSide *side = nullptr,
                         const LgXsegBufs *xseg = nullptr);
bool lg_halves(int m, int ncols, int cus, bool with_state);
int lg_halves_rounds(int m);  // rounds per launch of the two halves
bool lg_finishes(const LgAlign &one, int cus);  // launch_similarity_lg will honour one.mdk_out (else the caller runs sim_finish)
int launch_similarity_lg_batch(hipStream_t s, const LgAlign *table, const int32_t *colprefix, int K, int ncols_total, int max_m, int npos,
                               const void *tab, bool with_state, int *launches_out);
int lg_split(int m, int ncols, int cus);  // waves per column the launcher picks
bool lg_pipe(int m, int ncols, int cus);  // ... as loop waves of the pipelined kernel (+ a service wave), not of the barrier scheme
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
    // what the alignment's trim needs of the device (msa_trim_batch's engine: engine_needs): 1 the gap statistics alone, 2 the
    // similarity pipeline, 3 OverlapTrimmer's counts as well (ov_need, -> extra[m]), 4 the identities (ident points into the
    // result region: RepresentativeTrimmer clusters on the host-only view), 5 row digests (noduplicateseqs: extra = lengths
    // [round_up(m + 64, 64)], then two 64-bit hashes per row)
    int32_t kind;
    int32_t ov_need;          // kind 3: ceil(residue_overlap x (m - 1)) -- sequences that must agree with a residue
    int32_t *extra;
};
// The compact pipeline of one small alignment (msa_trim): THREE launches -- everything that reads the rows (gap counts, row
// totals, planes, codes + lists), the pair pass (which also sums the rows' weights), the similarity kernel with the MDK values
// (automated1: a fourth, the identity statistics) -- whose results the kernels store into pinned host memory themselves, instead of
// a dozen launches, two memsets and four copies: at 46 x 1181 such a trim was 0.15 ms of queue operations around 0.08 ms of kernels
// (profiles/r04_small_latency_ordinary_launch_sequence.jsonl).  No memset either: every word a kernel accumulates into is zeroed
// by the kernel in front of it (msastat_simx.hip: compact_front_kernel).
struct CompactArgs {
    const uint8_t *raw;       // [m][ld] residues (device memory, or pinned host memory read over the link: MSA_ZEROCOPY_KB)
    int64_t ld;
    int32_t m, n;
    uint32_t indet4;
    int32_t *flags;           // the context's 32 state words (ST_*) on the device: zeroed by the front kernel
    int32_t *gaps, *indets;   // the device's copies of the two count vectors
    int32_t *hres;            // the state block's mirror in pinned host memory, written by the kernels themselves (no copy back):
    int32_t h_gaps, h_indets, h_rowtot, h_slots;  // ... word offsets of the count vectors, the residues per sequence and the slots
                              // (per column block a first-bad-residue key, two words; then per column block a non-ASCII word)
    int32_t *scratch;         // compact_scratch_words(): [0] the ticket of the identity statistics; from [2]: wsum
    uint32_t *wsum;           // [m_pad + 64] fixed-point sums of W[i][j] over j > i (pair_epilogue), zeroed by the front kernel
    int32_t sim;              // 0: the gap statistics only (gaps, indets, rowtot; one launch)
    const uint8_t *lut;
    uint32_t *planes;
    int32_t nchunk, m_pad;
    uint8_t *codeT;
    int64_t ldk;
    int32_t ncols_pad;
    uint32_t *voff;
    uint16_t *vtrow;
    int32_t *nvalid;
    int32_t ldw, skiprow, big;
    int32_t lists;            // 0: codes only (the flat similarity kernel follows), 1: the compacted lists as well
    float *ident, *row_avg, *row_max;
    int32_t gated;            // automated1: the identity statistics + the selectMethod gate (compact_identity_kernel)
    int32_t cw;               // columns per column block (compact_front_cw): the slots are per column block
    int32_t xcd;              // the narrow column blocks that share the lines of a row dealt to one XCD
    int32_t nt;               // threads per block of the narrow front kernel (compact_front_nt)
};
// columns per column block of the front kernel for this shape (64: compact_front_kernel; 16 / 32: compact_front2_kernel)
int compact_front_cw(int m, int n, bool sim, bool rows_in_host_memory);
int compact_front_nt(int m, int n, int cus);
size_t compact_scratch_words(int m, int n);
size_t compact_slot_words(int n);
void launch_compact_front(hipStream_t s, const CompactArgs &a);
void launch_compact_identity(hipStream_t s, const CompactArgs &a);
// rows of n bytes, `ld_src` apart, into rows `ld_dst` apart (a multiple of 16, >= n), the padding zeroed: a contiguous host matrix
// that came up in one linear copy, laid out at the device pitch
void launch_repitch_rows(hipStream_t s, const uint8_t *src, int64_t ld_src, uint8_t *dst, int64_t ld_dst, int m, int n);
// OverlapTrimmer behind the front kernel (m <= 1024): the sequences' overlap counts, the device's decision, the residues per column
// over the sequences that stay -- two launches, every result stored into device AND pinned host memory (h_*)
void launch_overlap_small(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                          const int32_t *indets, int need, float min_ov, int32_t *good, int32_t *h_good, uint8_t *keep, uint8_t *h_keep,
                          int32_t *col_nongap, int32_t *h_col_nongap);
// ... the second of the two by itself (m <= 1024): the residues per column over keep_seq, plain stores into both vectors (no memset, no copy)
void launch_col_nongap_small(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_seq, int32_t *col_nongap,
                             int32_t *h_col_nongap);
int flat_rows_max();
void launch_similarity_flat(hipStream_t s, const LgAlign &one, const void *tab);  // two waves per column (one per sum): the column's pairs as one sequence
int pair_tiles_pipe(int m, int m_pad);    // tiles of the pair pass in its one-row-per-lane regime
bool pair_pipe_regime(int m, int m_pad);  // ... which launch_pair_counts picks for this shape (a batch holds no other)
void launch_fetch_rows_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_gap_counts_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_row_nongap_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_prep_planes_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_pair_counts_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks, int min_nchunk);  // min_nchunk: over the group
void launch_sim_finish_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_w_row_means_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
// OverlapTrimmer's counts (kind 3: a wave per sequence, the closed form over the group's gap / indetermination counts) and
// noduplicateseqs' row digests (kind 5), for every such alignment of a group
void launch_overlap_rows_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_row_digest_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_identity_stats_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks);
void launch_sim_encode_rm_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks, const uint8_t *lut);
void launch_similarity_cols_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int total, const void *tab);
// per alignment of a group: its columns by valid rows, most first, into lg[a].cols (a counting sort in LDS: max_m + 2 bins)
void launch_sort_columns_batch(hipStream_t s, const BAlign *table, const LgAlign *lg, int K, int max_m);
void launch_sim_lists_batch(hipStream_t s, const BAlign *table, const int32_t *prefix_encode, int blocks_encode, const int32_t *prefix_compact,
                            int blocks_compact, int K, const uint8_t *lut, int npos);
void launch_w_row_means(hipStream_t s, const float *wup, int m, int ldw, float *wbar);
int launch_similarity_seq(hipStream_t s, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, const float *wup, int ldw,
                          const void *tab, float *num_out, float *den_out);
void launch_sim_finish(hipStream_t s, const float *num, const float *den, const int32_t *gaps_w, int m, int n,
                       float *q_out, float *mdk_out);
void launch_overlap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                    const int32_t *indets, int need, uint32_t *col_ok, int nchunk, int32_t *good);
void launch_overlap_keep(hipStream_t s, const int32_t *good, int m, int n, float min_ov, uint8_t *keep);
void launch_row_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_res,
                       int32_t *row_nongap);
void launch_col_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_seq,
                       int32_t *col_nongap);
void launch_row_digest(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, int32_t *lengths,
                       unsigned long long *hashes);
size_t cluster_adj_words(int m);
size_t cluster_adj_buffer_words(int m);  // what launch_cluster's `adj` must hold
// count: the representatives are added to *count (zeroed by the caller), or null; h_keep: the mask in pinned host memory as well, or null
int launch_cluster(hipStream_t s, const float *ident, int ldw, const int32_t *seq_at, int m, float thr, uint32_t *adj,
                   uint8_t *keep_seq, int32_t *count, uint8_t *h_keep);
void launch_rows_equal(hipStream_t s, const uint8_t *raw, int n, int64_t ld, const int32_t *pairs, int npairs,
                       int32_t *equal);

}  // namespace msak
