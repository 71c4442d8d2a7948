// Launch wrappers of msastat_kernels.hip (internal; the public surface is include/msastat.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msak {

// Diagnostic switches (environment variables MSA_*), read ONCE when a context is created and handed to the
// launch wrappers through a thread-local pointer for the duration of an API call (contexts are per thread).
struct Tuning {
    int sim_kernel = 0;        // MSA_SIM_KERNEL: 0 / 4 "lg" binade-exact with per-lane grids, one column per wave (default),
                               // 5 "q2" two columns per wave, 3 "bx" its one-grid-per-round predecessor,
                               // 1 "chain" numerator + denominator kernels, 2 "pc"
    int sim_tcols = 0;         // MSA_SIM_TCOLS: column-tile width of the chain kernels (0 = 64)
    int sim_mode = 0;          // MSA_SIM_MODE: in-kernel stamps / ablations of the chain kernels
    int sim_tp = 1;            // MSA_SIM_TP=0: column-per-lane producers
    int den_waves = 0;         // MSA_DEN_WAVES
    int den_exec = 0;          // MSA_DEN_KERNEL=exec
    int sim_serial = 0;        // MSA_SIM_SERIAL: numerator and denominator kernel on one stream
    int device_clusters = -1;  // MSA_DEVICE_CLUSTERS: -1 unset (size heuristic), 0 host, 1 device
    int trace = 0;             // MSA_TRACE
    int upload_piece_kb = 1024;  // MSA_UPLOAD_PIECE_KB: rows are packed and sent in pieces of this size (0: one copy after packing everything)
    int pipeline = 1;          // MSA_PIPELINE: 0 msa_trim waits for the gap counts / identity statistics before it enqueues the similarity
                               // pass; 1 pipelined (side stream for large alignments); 2 pipelined, never a side stream; 3 always
    int bx_cols = 0;           // MSA_BX_COLS: columns per wave of the binade-exact kernel (0 = default)
    int bx_r0 = -1;            // MSA_BX_R0: rows evaluated in order before the first round (-1 = default)
    int bx_waves = 0;          // MSA_BX_WAVES: waves per workgroup of that kernel (0 = default)
    int bx_compact = 0;        // MSA_BX_COMPACT=1: the rows of a round are consecutive valid rows (gather loads of W)
    int bx_asm = 0;            // MSA_BX_ASM=1: the round loop with the table read folded into the multiply (inline asm; experimental)
    int lg_regs = 0;           // MSA_LG_REGS=1: the per-lane-grid kernel keeps the lane's table column in registers (not LDS)
    int lg_dbg = 0;            // MSA_LG_DBG: diagnostics of that kernel (1: no W loads, 64: W rows by buffer loads, 128: by compiler-addressed global loads -- all with MSA_SIM_MODE=64 only; 2: eight waves per workgroup; 16: two columns per wave without wave priorities)
    int mdk_host = 0;          // MSA_MDK_HOST=1: the device hands every exponential of the MDK values to the host (tests: both paths agree bit for bit)
    int pair_ti = 0;           // MSA_PAIR_TI: rows i per wave of the pair-count kernel (8, 16, 32; 0 = default)
    int pair_dense = 1;        // MSA_PAIR_DENSE: 0 the pair pass always on the seven raw symbol planes, 1 dense codes from 1500 sequences on, 2 always
    int pair_pipe = 1;         // MSA_PAIR_PIPE=0: the pair-count loop as the compiler schedules it instead of the software-pipelined one
    int pair_xcd = 1;          // MSA_PAIR_XCD=0: two-dimensional grid (half of its tiles return at once) instead of the triangle's tiles only
};
Tuning tuning_from_env();
void set_tuning(const Tuning *t);  // thread-local; nullptr = defaults
const Tuning &tuning();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of on every launch
int set_max_lds_once(const void *kernel, int bytes);

constexpr int PAIR_TI = 8;   // rows "i" per wave in pair_counts (wave-uniform, SGPR operands)
constexpr int PAIR_TJ = 2;   // rows "j" per lane at most (1 below ~3000 rows: twice the waves, 0.77 -> 0.52 ms at 2000 x 10000); m_pad % 128 == 0

void launch_prep_planes(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, uint32_t *planes,
                        int nchunk, int m_pad, int *err_flag, const uint32_t *used_slots, uint32_t *used_out);
int used_slot_words();  // words of the copies of the byte-value set that gap_counts fills and prep_planes folds into used_out[4]
// used: the copies (used_slot_words() zeroed words), nullptr: not collected
void launch_gap_counts(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, int32_t *gaps,
                       int32_t *indets, uint32_t *used);
void launch_pair_counts(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int ldw, uint32_t *hit,
                        uint32_t *dst, float *ident, float *wmat, float *wlow, int *undef_flag, const uint32_t *used);
bool pair_dense(int m);  // the pair pass of an m-sequence alignment runs on dense residue codes (MSA_PAIR_DENSE=0: never, 2: always)
int planes_total();  // planes in the plane array (raw + validity + dense codes as "j" and as "i")
// binade-exact similarity kernel (msastat_simx.hip)
int64_t bx_ldk(int m);
int bx_cols_pad(int n);
size_t bx_wlow_rows(int m);
void launch_sim_encode_cm(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut,
                          const int32_t *gaps_w, uint8_t *codeT, unsigned long long *err_key);
int bx_cols_per_wave();
void launch_bx_compact(hipStream_t s, const uint8_t *codeT, int m, int n, int ldw, int npos, uint32_t *voff, uint16_t *vrow,
                       uint8_t *vcode, uint16_t *vtrow, int32_t *nvalid);
int launch_similarity_bx(hipStream_t s, const uint32_t *voff, const uint16_t *vrow, const uint8_t *vcode,
                         const int32_t *nvalid, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, const float *wlow,
                         const float *wup, int ldw, const void *tab, float *num_out, float *den_out);
void launch_identity_stats(hipStream_t s, const float *ident, int m, int ldw, float *row_avg, float *row_max,
                           float *out2, float *row_min = nullptr, int *gate = nullptr);
int launch_similarity_lg(hipStream_t s, const uint32_t *voff, const uint16_t *vrow, const uint8_t *vcode, const uint16_t *vtrow,
                         int npos, const int32_t *nvalid, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols,
                         const float *wlow, const float *wup, int ldw, const void *tab, float *num_out, float *den_out,
                         const int *gate = nullptr);
bool lg2_fits(int npos);
int lg2_max_waves();
int launch_similarity_lg2(hipStream_t s, const uint32_t *voff, const uint16_t *vrow, const uint8_t *vcode, int npos,
                          const int32_t *nvalid, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, int waves,
                          uint32_t *uoff, uint32_t *utt, float *uee, int32_t *nunion, const float *wlow, const float *wup, int ldw,
                          const void *tab, float *num_out, float *den_out);
bool sim_num_transposed(int tcols);
void launch_sim_encode8(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, int npos,
                        const int32_t *gaps_w, void *codes8, unsigned long long *err_key, int tcols);
int launch_similarity_num(hipStream_t s, const void *codes8, int m, int n, int64_t ld, const float *wmat, int ldw,
                          const void *tab, int npos, float *num_out, int tcols);
void launch_sim_finish(hipStream_t s, const float *num, const float *den, const int32_t *gaps_w, int m, int n,
                       float *q_out, float *mdk_out);
int sim_den_workgroups(int nchunk, int m);
int den2_pm_ld(int m);
int launch_sim_den(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int n, const float *wmat,
                   int ldw, float *den_out, unsigned long long *pairmasks);
bool similarity_rc_fits(int m);
int sim_tile_cols(int n, int cus, int min_cols);
int sim_num_min_cols();
void launch_sim_encode32(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, int npos,
                         const int32_t *gaps_w, void *codes32, unsigned long long *err_key, int tcols);
int launch_similarity_pc(hipStream_t s, const void *codes32, int m, int n, int64_t ld, const float *wmat, int ldw,
                         const void *tab, int npos, const int32_t *gaps_w, float *q_out, float *mdk_out, int tcols);
void launch_overlap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                    const int32_t *indets, int need, uint32_t *col_ok, int nchunk, int32_t *good);
// used != nullptr: also collects the byte values of the alignment (as launch_gap_counts does)
void launch_row_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_res,
                       int32_t *row_nongap, uint32_t *used = nullptr);
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
