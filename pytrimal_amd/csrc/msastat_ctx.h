// msastat_ctx.h -- internal: the context behind include/msastat.h's opaque msa_ctx, its helpers, and the functions the
// translation units of the C-ABI shim share (msastat_ctx.hip: context, uploads, instrumentation; msastat_stats.hip: one
// entry point per statistic; msastat_trim.hip: msa_trim and its two pipelines; msastat_batch.hip: msa_trim_batch).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "msastat.h"
#include "msastat_host.h"
#include "msastat_kernels.h"

namespace msai {


constexpr int MSA_E_FALLBACK = -100;  // internal: a device path does not apply, take the host path (never returned by the ABI)

template <typename T>
struct DevBuf {  // grow-only device allocation, reused across uploads
    T *p = nullptr;
    size_t cap = 0;
    uint64_t tag = 0;  // caller's note about the contents (e.g. "padding zeroed for this shape"); a reallocation clears it
    hipError_t reserve(size_t count) {
        if (count <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        tag = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e == hipSuccess) cap = count;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct DevView {  // a window of another device allocation (the state block, the table block)
    T *p = nullptr;
};

template <typename T>
struct PinBuf {  // pinned host staging
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t count) {
        if (count <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = count;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ProfEntry {
    double ms = 0;
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

}  // namespace msai
using msai::DevBuf;
using msai::DevView;
using msai::PinBuf;
using msai::ProfEntry;

struct SimOrder {  // the similarity kernel's column list: entries (padded)
    int npad = 0;
};

struct msa_ctx {
    int device = 0;
    int cus = 256;  // compute units of the device
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // the side stream of msa_trim's pipeline (codes, lists, row totals beside the pair pass)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_digest = nullptr;  // behind the copies of row_digest_begin
    hipEvent_t ev_rowtot = nullptr;  // behind the copy of stage_row_totals
    hipEvent_t ev_front = nullptr;   // behind the compact pipeline's front kernel (compact_begin sorts the columns from its counts)
    PinBuf<int32_t> h_len;           // ungapped lengths on their way to the host
    hipEvent_t ev_gaps = nullptr;  // behind the staged copy of the gap counts: waiting for it does not wait for later work
    hipEvent_t ev_upload = nullptr;  // behind the copies of msa_upload_packed_async
    bool upload_pending = false;     // ... which no wait on the stream has covered yet: msa_trim waits for the event before it returns
    char hip_err[256] = {0};

    // alignment
    int m = 0, n = 0;
    int64_t ld = 0;
    uint8_t indet = 'X';
    const uint8_t *raw = nullptr;  // device
    DevBuf<uint8_t> raw_lin;            // a contiguous matrix as it lies on the host, on its way into raw_own's pitch (upload_packed: MSA_PATH_UPLOAD_REPITCHED)
    DevBuf<uint8_t> raw_own;

    // derived device data + validity flags
    int nchunk = 0, m_pad = 0, ldw = 0;
    DevBuf<uint32_t> planes;
    bool have_planes = false;
    // The small per-alignment state lives in ONE allocation so that one memset prepares it and one copy fetches its
    // flags: 16 words of flags (ST_*), then the gap counts and the indetermination counts (state_npad words each).
    DevBuf<int32_t> state;
    int state_npad = 0;
    int state_rpad = 0;
    DevBuf<int32_t> cscratch;    // the compact pipeline of small alignments (compact_begin): scratch words (msak::compact_scratch_words)
    PinBuf<int32_t> h_cres;      // ... and its results, written by the kernels themselves into pinned host memory: the state block's
                                 // layout, then the residues per sequence (state_rpad words), MDK and Q (state_npad floats each) and
                                 // the verdict slots of the front kernel's blocks
    int compact_cw = 64;         // columns per column block of the front kernel last launched (the slots are per block)
    bool sim_halves = false;     // the column list of the similarity kernel is laid out part by part (staggered parts: msak::lg_halves)
    hipStream_t part_stream[2] = {nullptr, nullptr};  // streams of a third and fourth part (MSA_LG_PARTS; the second part runs on stream2)
    hipEvent_t part_join[2] = {nullptr, nullptr};
    bool state_zeroed = false;   // for the current alignment
    bool errkey_dirty = false;   // an encode kernel may have written the first-bad-residue key since the state was zeroed
    bool flags_dirty = false;    // a kernel that may raise a flag was enqueued since the flags were last fetched
    PinBuf<int32_t> h_flags;     // the 16 flag words on the host (valid after the synchronisation that follows a fetch)
    DevView<int32_t> gaps, indets;
    bool have_gaps = false;
    DevBuf<float> ident, wmat;
    DevBuf<float> wlow;        // strictly lower triangular mirror of wmat (binade-exact similarity kernel)
    DevBuf<float> wbar;        // mean weight of every row over its later partners (that kernel's predictor)
    DevBuf<uint8_t> codeT;     // column-major similarity codes of that kernel
    DevBuf<uint32_t> bx_off;   // ... and the compacted lists of every column's valid rows: W row offset (or index),
    DevBuf<uint16_t> bx_trow;  //     byte offset of the residue's row in a [row][64 lanes] float table
    DevBuf<int32_t> bx_nvalid;
    DevBuf<int32_t> simcols;   // the columns that kernel evaluates (those the 80 % gap rule does not zero), sorted by gap count
    PinBuf<int32_t> h_simcols;
    std::vector<int32_t> sort_bins;  // scratch of the column ordering
    SimOrder order;                 // the column list in h_simcols when order_ready (built ahead of similarity() by msa_trim)
    bool order_ready = false;
    msak::Tuning tuning;       // the MSA_* diagnostic switches, read once in msa_ctx_create
    bool have_ident = false, have_w = false;
    bool have_wbar = false;    // wbar holds the mean weights of the current W (the compact pipeline leaves them to its kernel: wsum)
    DevBuf<uint32_t> hit, dst;
    DevBuf<float> row_avg, row_max, row_min;
    DevView<float> stats2;  // {mean, max} of the identity rows (state block)
    DevBuf<float> tables;  // {distance, both-valid} table followed by the byte -> code LUT: one upload, cached by content
    DevView<uint8_t> lut;
    DevView<float> tab;
    std::vector<int32_t> tab_vhash;  // what `tables` was built from
    std::vector<float> tab_dist;
    int tab_npos = -1, tab_indet = -1;
    DevBuf<int32_t> gaps_w;
    DevBuf<float> mdk, simnum, simden;  // mdk: MDK [n] followed by Q [n]
    bool compact_sorted = false;        // the last compact pipeline dealt the columns to its similarity kernel by weight (paths[1])
    DevBuf<char> xsegbuf;               // the XCD-per-segment kernel's rings and flags (msak::lg_xseg_bytes)
    DevBuf<float> simstate;             // the similarity kernel's per-column state between its launches (many rows: a few rounds per launch)
    DevView<unsigned long long> errkey;  // first bad residue of the similarity pass, complemented (0 = none): state block
    DevView<int> errflag;                // prep_planes' non-ASCII flag (state block)
    DevView<int> pairflag;               // set by the pair pass when some pair has dst = 0 (undefined identity; state block)
    int pairflag_state = 0;  // 0 no pair pass since the flag was last read, 1 enqueued and its flag word not fetched yet, 2 fetched (h_flags holds it)
    DevBuf<uint32_t> col_ok;
    DevBuf<int32_t> good, row_cnt, col_cnt, lengths, pairs, equal;
    DevBuf<uint8_t> keep_res_d, keep_seq_d;
    DevBuf<unsigned long long> hashes;

    // pinned staging
    PinBuf<int32_t> h_i32;
    PinBuf<float> h_f32;
    PinBuf<unsigned long long> h_u64;
    PinBuf<uint8_t> h_u8, h_raw;
    // results that are fetched asynchronously and validated at the next synchronisation of the stream
    bool planes_pending = false;   // prep_planes' non-ASCII flag has not been looked at yet
    PinBuf<int32_t> h_gapstage;    // gap / indetermination counts on their way to h_gaps / h_indets
    int gaps_staged = 0;           // 0 none, 1 copy enqueued, 2 copy complete (a synchronisation followed)
    PinBuf<int32_t> h_colcnt;      // residues per column over the sequences the device clustering kept (stage_kept_column_counts)
    bool colcnt_staged = false;
    PinBuf<int32_t> h_rowtot;      // residues (non-gap symbols) per sequence over all columns, fetched asynchronously
    int rowtot_staged = 0;         // 0 none, 1 copy enqueued, 2 copy complete, 3 enqueued on the side stream (not joined yet)
    bool pipe_active = false, pipe_gated = false;  // msa_trim's similarity pipeline is in flight (see sim_pipeline_begin)
    // A host-only view of one alignment of a batch (msa_trim_batch's engine): every statistic the selection logic reads is
    // already on the host (h_gaps, h_indets, h_flags, h_f32 = MDK + Q, h_rowtot point into the engine's result buffer);
    // trim_impl enqueues nothing, and a path that would need the device returns MSA_E_FALLBACK (the engine then trims that
    // alignment through an ordinary context).
    bool prefetched = false;
    const uint8_t *host_rows = nullptr;  // ... and the caller's rows (host memory), for the rare selection that looks at residues again
    int64_t host_ld = 0;
    const float *pref_ident = nullptr;               // (RepresentativeTrimmer in the engine) the identities [m][ldw] in the result buffer
    const int32_t *pref_lengths = nullptr;           // (noduplicateseqs in the engine) the rows' ungapped lengths ...
    const unsigned long long *pref_hashes = nullptr;  // ... and their two hashes each

    // host copies valid for the current alignment
    std::vector<int32_t> h_gaps, h_indets;
    std::vector<int32_t> only_gaps_rows;  // the sequences the last msa_trim removed because the trimming left them with gaps only

    // Cleaner::calculateSpuriousVector's values, staged by compact_overlap (a small alignment's front kernel and the overlap kernels
    // behind ONE wait) for the overlap() call that follows in the same msa_trim
    std::vector<float> ov_vals;
    float ov_key = 0.0f;
    bool ov_valid = false;
    std::vector<uint8_t> ov_keep;  // ... and the sequences its device-side decision kept (the mask h_colcnt was counted over)
    bool ov_colcnt = false;        // h_colcnt holds the residues per column over ov_keep
    int sim_launches = 0;  // kernel launches of the last similarity pass (msa_debug_sim_launches)
    // which path the last upload and the last statistic / trim call took (msa_debug_last_paths; MSA_PATH_* of msastat.h):
    // [0] upload, [1] pipeline, [2] similarity kernel, [3] its waves per column, [4] its launches, [5] it wrote MDK itself,
    // [6] pair kernel, [7] its waves per tile
    int32_t paths[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // profiling
    int prof_on = 0;  // 0 off, 1 every kernel group, 2 the similarity and pair passes only
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
};

namespace msai {


struct TuneScope {  // the context's diagnostic switches, visible to the launch wrappers for the duration of a call
    const msak::Tuning *prev;
    explicit TuneScope(msa_ctx *c) : prev(msak::current_tuning()) { msak::set_tuning(&c->tuning); }
    ~TuneScope() { msak::set_tuning(prev); }  // (calls nest: the batch engine runs trim_impl on its host-only view)
};

struct PathScope {  // msa_debug_last_paths: the call's entries start at "none", the launch wrappers' note lands in them at its end
    msa_ctx *c;
    explicit PathScope(msa_ctx *ctx) : c(ctx) {
        std::fill(c->paths + 1, c->paths + 8, 0);
        msak::launch_note() = msak::LaunchNote();
    }
    ~PathScope() {
        const msak::LaunchNote &k = msak::launch_note();
        c->paths[2] = k.sim_kind, c->paths[3] = k.lg_split, c->paths[4] = k.lg_launches, c->paths[5] = k.lg_fin;
        c->paths[6] = k.pair_kind, c->paths[7] = k.pair_waves;
    }
};

int fail_hip(msa_ctx *c, hipError_t e, const char *what);
#define HIPCHK(ctx, expr)                                    \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return fail_hip(ctx, _e, #expr); \
    } while (0)

// flag words of the state block
enum {
    ST_ERRFLAG = 0, ST_PAIRFLAG = 1, ST_ERRKEY = 2 /* 2 words */, ST_STATS = 4 /* 2 floats */, ST_GATE = 6,
    ST_FLAGS = 16 /* the words fetched with every wait */,
    ST_WORDS = 32
};

// Every wait on the context's stream goes through here: it fetches the flag words when a kernel that may have raised
// one ran since the last fetch (one 64-byte copy in front of the wait) and settles the asynchronous fetches above.
int sync_stream(msa_ctx *c);
#define SYNC(c)                          \
    do {                                 \
        const int rc_sync_ = sync_stream(c); \
        if (rc_sync_) return rc_sync_;   \
    } while (0)

struct ProfScope {  // records an event pair around a launch sequence when profiling is on
    msa_ctx *c;
    const char *name;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    bool on_ = false;
    ProfScope(msa_ctx *ctx, const char *nm, hipStream_t on = nullptr) : c(ctx), name(nm), st(on ? on : ctx->stream) {
        // level 2: the two pairwise passes only (an event pair costs a few microseconds of queue time: seven pairs per
        // trim are ~10 % of a 0.33 ms trim)
        on_ = c->prof_on == 1 || (c->prof_on == 2 && (std::strcmp(nm, "sim") == 0 || std::strcmp(nm, "pairs") == 0));
        if (!on_) return;
        a = take();
        b = take();
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!on_) return;
        (void)hipEventRecord(b, st);
        ProfEntry &e = c->prof[name];
        e.pending.emplace_back(a, b);
        e.launches++;
    }
    hipEvent_t take() {
        if (!c->event_pool.empty()) {
            hipEvent_t ev = c->event_pool.back();
            c->event_pool.pop_back();
            return ev;
        }
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        return ev;
    }
};

// ---- shared between the translation units (defined where the comment says)
// msastat_ctx.hip
int fail_hip(msa_ctx *c, hipError_t e, const char *what);
int sync_stream(msa_ctx *c);
void prof_collect(msa_ctx *c);
void invalidate(msa_ctx *c);
size_t state_words(const msa_ctx *c);
size_t mirror_words(const msa_ctx *c);
int layout_state(msa_ctx *c);
int ensure_state(msa_ctx *c);
int set_shape(msa_ctx *c, int m, int n, uint8_t indet);
bool zero_copy_rows(const msa_ctx *c, int m, int n);
// msastat_stats.hip
int ensure_planes(msa_ctx *c);
int copy_gap_counts(msa_ctx *c);
int ensure_gaps(msa_ctx *c, bool to_host);
int stage_gaps(msa_ctx *c);
int pair_buffers(msa_ctx *c, bool need_ident, bool need_w);
int run_pairs(msa_ctx *c, bool want_ident, bool want_w, bool want_counts);
int identity_stats(msa_ctx *c, float *avg_seq, float *max_seq);
void build_tables(const int32_t *vhash, const float *dist, int npos, uint8_t indet, uint8_t lut[256], float *tab);
int ensure_tables(msa_ctx *c, const int32_t *vhash, const float *dist, int npos);
int reset_errkey(msa_ctx *c);
int fetch_similarity_enqueue(msa_ctx *c, int n);
int fetch_similarity_finish(msa_ctx *c, int n, float *mdk_out, float *q_out, msa_err_detail *detail);
int fetch_similarity(msa_ctx *c, int n, float *mdk_out, float *q_out, msa_err_detail *detail);
int build_sim_order(msa_ctx *c, const int32_t *gaps_windowed, SimOrder *out);
int sim_lists_enqueue(msa_ctx *c, int npos, const int32_t *gw_dev, hipStream_t st);
int sim_order_enqueue(msa_ctx *c, const SimOrder &ord, hipStream_t st);
int sim_kernel_enqueue(msa_ctx *c, int npos, const SimOrder &ord, const int32_t *gw_dev, const int *gate);
int similarity(msa_ctx *c, const int32_t *vhash, const float *dist, int npos, const int32_t *gaps_windowed, float *mdk_out, float *q_out, msa_err_detail *detail);
int overlap_enqueue(msa_ctx *c, float residue_overlap);
int overlap(msa_ctx *c, float residue_overlap, float *out);
int stage_row_totals(msa_ctx *c, hipStream_t st = nullptr);
int remove_all_gaps(msa_ctx *c, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info);
int stage_kept_column_counts(msa_ctx *c, const std::vector<int32_t> &lengths);
int row_digest_begin(msa_ctx *c);
int row_digest_end(msa_ctx *c, std::vector<int32_t> &lengths, std::vector<unsigned long long> *hashes);
int row_digest(msa_ctx *c, std::vector<int32_t> &lengths, std::vector<unsigned long long> *hashes);
int ungapped_lengths(msa_ctx *c, std::vector<int32_t> &lengths);
int remove_duplicates(msa_ctx *c, uint8_t *keep_seq);
int device_representatives(msa_ctx *c, float max_identity, uint8_t *keep_seq);
int device_cluster_count(msa_ctx *c, int clusters, uint8_t *keep_seq);
int fetch_ident(msa_ctx *c, std::vector<float> &host);
// msastat_trim.hip
bool sim_pipeline_applies(const msa_ctx *c, const msa_trim_params *p, int sim_hw);
int sim_pipeline_begin(msa_ctx *c, const msa_trim_params *p, int gap_hw, bool gated, std::vector<int32_t> &gaps_w);
bool compact_sim_applies(const msa_ctx *c, int gap_hw);
bool compact_gaps_applies(const msa_ctx *c);
msak::CompactArgs compact_args(msa_ctx *c);
int compact_prepare(msa_ctx *c);
int compact_fetch(msa_ctx *c, bool sim);
int compact_gaps(msa_ctx *c);
int compact_overlap(msa_ctx *c, float residue_overlap, float sequence_overlap);
int compact_begin(msa_ctx *c, const int32_t *vhash, const float *dist, int npos, bool gated);
int trim_impl(msa_ctx *c, const msa_trim_params *p, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info);
// msastat_batch.hip
int engine_needs(const msa_trim_params *p);
}  // namespace msai
