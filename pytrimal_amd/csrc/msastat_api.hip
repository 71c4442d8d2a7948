// msastat_api.hip -- the C-ABI shim of include/msastat.h: context, uploads, one entry point per
// statistic, and msa_trim (the trimAlManager::clean_alignment equivalent that keeps the
// alignment and the m*m matrices on the device and moves only O(m + n) vectors to the host).
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

namespace {

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

}  // namespace

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

    // host copies valid for the current alignment
    std::vector<int32_t> h_gaps, h_indets;
    std::vector<int32_t> only_gaps_rows;  // the sequences the last msa_trim removed because the trimming left them with gaps only

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

namespace {

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

int fail_hip(msa_ctx *c, hipError_t e, const char *what) {
    std::snprintf(c->hip_err, sizeof(c->hip_err), "%s: %s", what, hipGetErrorString(e));
    return MSA_E_HIP;
}
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
static int sync_stream(msa_ctx *c) {
    const bool fetching = c->flags_dirty && c->state.p;
    if (c->flags_dirty && c->state.p) {
        hipError_t e = c->h_flags.reserve(ST_FLAGS);
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->h_flags.p, c->state.p, ST_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return fail_hip(c, e, "flag fetch");
        c->flags_dirty = false;
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail_hip(c, e, "hipStreamSynchronize");
    c->upload_pending = false;  // (whatever an asynchronous upload enqueued is through)
    if (fetching && c->pairflag_state == 1) c->pairflag_state = 2;  // (the pair pass's flag word came with this fetch)
    if (c->gaps_staged == 1) c->gaps_staged = 2;
    if (c->rowtot_staged == 1) c->rowtot_staged = 2;
    if (c->planes_pending) {
        c->planes_pending = false;
        if (c->h_flags.p[ST_ERRFLAG]) {
            c->have_planes = false;
            return MSA_E_NON_ASCII;
        }
    }
    return MSA_OK;
}
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

void prof_collect(msa_ctx *c) {
    for (auto &kv : c->prof) {
        for (auto &pr : kv.second.pending) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
                kv.second.ms += ms;
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

void invalidate(msa_ctx *c) {
    c->have_planes = c->have_gaps = c->have_ident = c->have_w = c->have_wbar = false;
    c->pairflag_state = 0;
    c->h_gaps.clear();
    c->gaps_staged = 0;
    c->rowtot_staged = 0;
    c->order_ready = false;
    c->planes_pending = false;
    c->h_indets.clear();
    c->state_zeroed = false;
    c->flags_dirty = false;
    c->colcnt_staged = false;
}

// the state block of the current alignment, zeroed once (one memset for the flags and both count vectors)
// the block and the views into it (no memset: ensure_state zeroes the flags and the counts, the compact pipeline writes every word)
size_t state_words(const msa_ctx *c) { return ST_WORDS + (size_t)2 * c->state_npad; }
// (the compact pipeline's mirror of the block in pinned host memory: behind the counts the residues per sequence, MDK, Q, the slots)
size_t mirror_words(const msa_ctx *c) { return ST_WORDS + (size_t)4 * c->state_npad + c->state_rpad + msak::compact_slot_words(c->n); }
int layout_state(msa_ctx *c) {
    c->state_npad = round_up(std::max(c->n, 1) + 64, 64);
    c->state_rpad = round_up(std::max(c->m, 1) + 64, 64);
    HIPCHK(c, c->state.reserve(state_words(c)));
    c->errflag.p = c->state.p + ST_ERRFLAG;
    c->pairflag.p = c->state.p + ST_PAIRFLAG;
    c->errkey.p = reinterpret_cast<unsigned long long *>(c->state.p + ST_ERRKEY);
    c->stats2.p = reinterpret_cast<float *>(c->state.p + ST_STATS);
    c->gaps.p = c->state.p + ST_WORDS;
    c->indets.p = c->gaps.p + c->state_npad;
    return MSA_OK;
}
int ensure_state(msa_ctx *c) {
    if (c->state_zeroed) return MSA_OK;
    int rc = layout_state(c);
    if (rc) return rc;
    const size_t words = ST_WORDS + (size_t)2 * c->state_npad;
    HIPCHK(c, hipMemsetAsync(c->state.p, 0, words * sizeof(int32_t), c->stream));
    c->state_zeroed = true;
    c->errkey_dirty = false;
    return MSA_OK;
}

int set_shape(msa_ctx *c, int m, int n, uint8_t indet) {
    if (m < 0 || n < 0) return MSA_E_INVALID;
    c->m = m;
    c->n = n;
    c->indet = indet;
    c->nchunk = (n + 31) / 32;
    c->m_pad = round_up(std::max(m, 1), 128);
    c->ldw = round_up(std::max(m, 1), 64);
    invalidate(c);
    return MSA_OK;
}

int ensure_gaps(msa_ctx *c, bool to_host);

int ensure_planes(msa_ctx *c) {
    if (c->have_planes) return MSA_OK;
    HIPCHK(c, c->planes.reserve((size_t)msak::planes_total() * c->nchunk * c->m_pad + 64));
    int rc = ensure_state(c);
    if (rc) return rc;
    {
        ProfScope ps(c, "prep");
        msak::launch_prep_planes(c->stream, c->raw, c->m, c->n, c->ld, c->indet, c->planes.p, c->nchunk, c->m_pad, c->errflag.p);
    }
    HIPCHK(c, hipGetLastError());
    // the non-ASCII verdict comes back with the next synchronisation of the stream (sync_stream): every caller
    // synchronises before it hands anything derived from the planes to the host
    c->flags_dirty = true;
    c->planes_pending = true;
    c->have_planes = true;
    return MSA_OK;
}

// both count vectors to the pinned staging area: neighbours in the state block, one copy
int copy_gap_counts(msa_ctx *c) {
    const size_t words = (size_t)c->state_npad + c->n;
    HIPCHK(c, c->h_gapstage.reserve(words));
    HIPCHK(c, hipMemcpyAsync(c->h_gapstage.p, c->gaps.p, sizeof(int32_t) * words, hipMemcpyDeviceToHost, c->stream));
    return MSA_OK;
}

int ensure_gaps(msa_ctx *c, bool to_host) {
    if (!c->have_gaps) {
        int rc = ensure_state(c);  // (zeroes both count vectors)
        if (rc) return rc;
        {
            ProfScope ps(c, "gaps");
            msak::launch_gap_counts(c->stream, c->raw, c->m, c->n, c->ld, c->indet, c->gaps.p, c->indets.p);
        }
        HIPCHK(c, hipGetLastError());
        c->have_gaps = true;
    }
    if (to_host && c->h_gaps.empty() && c->n > 0) {
        if (c->gaps_staged == 0) {
            int rc = copy_gap_counts(c);
            if (rc) return rc;
            c->gaps_staged = 1;
            SYNC(c);
        }
        if (c->gaps_staged == 1) {  // (2: a synchronisation has already covered the copies)
            // staged earlier: wait for the copies alone, not for what was enqueued behind them
            if (c->ev_gaps) HIPCHK(c, hipEventSynchronize(c->ev_gaps));
            else SYNC(c);
        }
        c->h_gaps.assign(c->h_gapstage.p, c->h_gapstage.p + c->n);
        c->h_indets.assign(c->h_gapstage.p + c->state_npad, c->h_gapstage.p + c->state_npad + c->n);
        c->gaps_staged = 0;
    }
    return MSA_OK;
}

// enqueue the gap counts and their copy to the host without waiting: the next synchronisation completes them
int stage_gaps(msa_ctx *c) {
    int rc = ensure_gaps(c, false);
    if (rc) return rc;
    if (!c->h_gaps.empty() || c->gaps_staged || c->n <= 0) return MSA_OK;
    rc = copy_gap_counts(c);
    if (rc) return rc;
    if (!c->ev_gaps) HIPCHK(c, hipEventCreateWithFlags(&c->ev_gaps, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_gaps, c->stream));
    c->gaps_staged = 1;
    return MSA_OK;
}

// the float matrices of the pair pass, sized, their padding zeroed
int pair_buffers(msa_ctx *c, bool need_ident, bool need_w) {
    int rc;
    const size_t fsz = (size_t)c->m * c->ldw + 512;  // slack: the similarity kernels read W a round past a row end
    // The pair pass writes every entry (i, j < m) of the float matrices and nothing else; the padding (columns m..ldw,
    // the rows and the slack behind row m, the diagonal and the unused triangle of W) must read as zero.  It is zeroed
    // when a buffer is new or the shape differs from the one it was last zeroed for -- not on every pass.
    const uint64_t shape_tag = ((uint64_t)(uint32_t)c->m << 32) | (uint32_t)c->ldw | (1ull << 63);
    auto zero_for_shape = [&](DevBuf<float> &buf, size_t count) -> int {
        HIPCHK(c, buf.reserve(count));
        if (buf.tag != shape_tag) {
            HIPCHK(c, hipMemsetAsync(buf.p, 0, count * sizeof(float), c->stream));
            buf.tag = shape_tag;
        }
        return MSA_OK;
    };
    if (need_ident && (rc = zero_for_shape(c->ident, fsz))) return rc;
    const size_t lsz = (msak::bx_wlow_rows(c->m) + 2) * (size_t)c->ldw;  // rows past m: zeros the kernel's prefetch may touch
    if (need_w) {
        if ((rc = zero_for_shape(c->wmat, fsz))) return rc;
        if ((rc = zero_for_shape(c->wlow, lsz))) return rc;
        HIPCHK(c, c->wbar.reserve((size_t)c->m + 128));
    }
    return MSA_OK;
}

// pair pass; want_* select which float matrices / integer matrices are produced
int run_pairs(msa_ctx *c, bool want_ident, bool want_w, bool want_counts) {
    int rc = ensure_planes(c);
    if (rc) return rc;
    const bool need_ident = want_ident && !c->have_ident, need_w = want_w && !c->have_w;
    if (!need_ident && !need_w && !want_counts) return MSA_OK;
    if ((rc = pair_buffers(c, need_ident, need_w))) return rc;
    if (want_counts) {
        HIPCHK(c, c->hit.reserve((size_t)c->m * c->m + 1));
        HIPCHK(c, c->dst.reserve((size_t)c->m * c->m + 1));
        HIPCHK(c, hipMemsetAsync(c->hit.p, 0, (size_t)c->m * c->m * sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->dst.p, 0, (size_t)c->m * c->m * sizeof(uint32_t), c->stream));
    }
    {
        ProfScope ps(c, "pairs");
        msak::launch_pair_counts(c->stream, c->planes.p, c->nchunk, c->m_pad, c->m, c->ldw,
                                 want_counts ? c->hit.p : nullptr, want_counts ? c->dst.p : nullptr,
                                 need_ident ? c->ident.p : nullptr, need_w ? c->wmat.p : nullptr, need_w ? c->wlow.p : nullptr,
                                 c->pairflag.p);
    }
    if (need_w) msak::launch_w_row_means(c->stream, c->wmat.p, c->m, c->ldw, c->wbar.p);
    HIPCHK(c, hipGetLastError());
    c->flags_dirty = true;
    c->pairflag_state = 1;  // (its flag word arrives with the next flag fetch: sync_stream)
    if (need_ident) c->have_ident = true;
    if (need_w) c->have_w = c->have_wbar = true;
    return MSA_OK;
}

int identity_stats(msa_ctx *c, float *avg_seq, float *max_seq) {
    int rc = run_pairs(c, true, false, false);
    if (rc) return rc;
    HIPCHK(c, c->row_avg.reserve(c->m + 64));
    HIPCHK(c, c->row_max.reserve(c->m + 64));
    {
        ProfScope ps(c, "idstats");
        msak::launch_identity_stats(c->stream, c->ident.p, c->m, c->ldw, c->row_avg.p, c->row_max.p, c->stats2.p);
    }
    HIPCHK(c, hipGetLastError());
    c->flags_dirty = true;  // (the two statistics are flag words: they come back with the synchronisation)
    SYNC(c);
    std::memcpy(avg_seq, c->h_flags.p + ST_STATS, sizeof(float));
    std::memcpy(max_seq, c->h_flags.p + ST_STATS + 1, sizeof(float));
    return MSA_OK;
}

// byte -> table offset LUT and {distance, both-valid} table (see msastat_kernels.hip)
void build_tables(const int32_t *vhash, const float *dist, int npos, uint8_t indet, uint8_t lut[256], float *tab) {
    for (int b = 0; b < 256; ++b) {
        uint8_t code;
        if (b == '-' || b == indet) code = 224;
        else {
            const int up = (b >= 'a' && b <= 'z') ? b - 32 : b;
            if (up < 'A' || up > 'Z') code = 0xFE;
            else if (vhash[up - 'A'] < 0 || vhash[up - 'A'] >= npos) code = 0xFF;
            else code = static_cast<uint8_t>(vhash[up - 'A'] * 8);
        }
        lut[b] = code;
    }
    std::memset(tab, 0, sizeof(float) * 2 * 29 * 32);
    for (int a = 0; a < npos; ++a)
        for (int b = 0; b < npos; ++b) {
            tab[(a * 32 + b) * 2 + 0] = dist[a * npos + b];
            tab[(a * 32 + b) * 2 + 1] = 1.0f;
        }
}

// The two tables of the similarity pass in one device block, uploaded when their inputs differ from the ones the block
// was built from (the same matrix serves call after call).
int ensure_tables(msa_ctx *c, const int32_t *vhash, const float *dist, int npos) {
    constexpr size_t TABF = 2 * 29 * 32;
    const bool same = c->tables.p && c->tab_npos == npos && c->tab_indet == (int)c->indet && c->tab_vhash.size() == 26 &&
                      std::memcmp(c->tab_vhash.data(), vhash, 26 * sizeof(int32_t)) == 0 &&
                      c->tab_dist.size() == (size_t)npos * npos &&
                      std::memcmp(c->tab_dist.data(), dist, sizeof(float) * npos * npos) == 0;
    if (same) return MSA_OK;
    HIPCHK(c, c->tables.reserve(TABF + 64));
    c->tab.p = c->tables.p;
    c->lut.p = reinterpret_cast<uint8_t *>(c->tables.p + TABF);
    HIPCHK(c, c->h_f32.reserve(std::max<size_t>(TABF + 64, (size_t)2 * c->n + 64)));
    // (the staging area may still be read by an earlier upload: wait for the stream before rewriting it)
    SYNC(c);
    build_tables(vhash, dist, npos, c->indet, reinterpret_cast<uint8_t *>(c->h_f32.p + TABF), c->h_f32.p);
    HIPCHK(c, hipMemcpyAsync(c->tables.p, c->h_f32.p, (TABF + 64) * sizeof(float), hipMemcpyHostToDevice, c->stream));
    SYNC(c);  // (h_f32 is reused for the results)
    c->tab_vhash.assign(vhash, vhash + 26);
    c->tab_dist.assign(dist, dist + (size_t)npos * npos);
    c->tab_npos = npos;
    c->tab_indet = c->indet;
    return MSA_OK;
}

// the first-bad-residue key is zero after ensure_state; a second similarity pass over the same alignment resets it
int reset_errkey(msa_ctx *c) {
    int rc = ensure_state(c);
    if (rc) return rc;
    if (c->errkey_dirty) HIPCHK(c, hipMemsetAsync(c->errkey.p, 0, sizeof(unsigned long long), c->stream));
    c->errkey_dirty = true;
    return MSA_OK;
}

// MDK / Q and the first-bad-residue key back to the host (one synchronisation)
int fetch_similarity_enqueue(msa_ctx *c, int n) {
    HIPCHK(c, c->h_f32.reserve(std::max<size_t>((size_t)2 * 29 * 32 + 64, (size_t)2 * n + 64)));
    // (MDK and Q are the two halves of one buffer: one copy; the first-bad-residue key is a flag word)
    HIPCHK(c, hipMemcpyAsync(c->h_f32.p, c->mdk.p, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    c->flags_dirty = true;
    return MSA_OK;
}
// (after the synchronisation that followed fetch_similarity_enqueue)
int fetch_similarity_finish(msa_ctx *c, int n, float *mdk_out, float *q_out, msa_err_detail *detail) {
    unsigned long long key;
    std::memcpy(&key, c->h_flags.p + ST_ERRKEY, sizeof(key));
    key = ~key;  // (kept complemented on the device, where 0 = none and the largest complement = the first residue)
    if (key != ~0ull) {
        if (detail) {
            detail->col = static_cast<int32_t>(key >> 40);
            detail->row = static_cast<int32_t>((key >> 16) & 0xFFFFFFull);
            detail->byte = static_cast<int32_t>(key & 0xFF);
        }
        return ((key >> 8) & 1ull) ? MSA_E_UNDEFINED_SYMBOL : MSA_E_INCORRECT_SYMBOL;
    }
    // (a NaN is a value whose exponential the device would not vouch for -- sim_finish_kernel: evaluated here, as the
    // reference does it, from the bit-exact Q)
    const float *dm = c->h_f32.p, *dq = c->h_f32.p + n;
    for (int i = 0; i < n; ++i) {
        float v = dm[i];
        if (v != v) {
            v = static_cast<float>(std::exp(-static_cast<double>(dq[i])));
            if (v > 1.0f) v = 1.0f;
        }
        mdk_out[i] = v;
    }
    if (q_out) std::memcpy(q_out, dq, sizeof(float) * n);
    return MSA_OK;
}
int fetch_similarity(msa_ctx *c, int n, float *mdk_out, float *q_out, msa_err_detail *detail) {
    int rc = fetch_similarity_enqueue(c, n);
    if (rc) return rc;
    SYNC(c);
    return fetch_similarity_finish(c, n, mdk_out, q_out, detail);
}

// The columns the binade-exact kernels evaluate (not zeroed by the ">= 80 % gaps" rule), the ones with the most valid
// rows first (their waves run longest), into the pinned staging list.  Host work only: msa_trim calls it while the pair
// pass runs, similarity() otherwise.
int build_sim_order(msa_ctx *c, const int32_t *gaps_windowed, SimOrder *out) {
    const int m = c->m, n = c->n;
    const int32_t *gw_host = gaps_windowed ? gaps_windowed : c->h_gaps.data();
    HIPCHK(c, c->h_simcols.reserve((size_t)n + 128));
    // (counting sort by the number of rows that take no part, stable, in ordinary memory: the pinned staging
    // buffer is only written once, front to back)
    int32_t *list = c->h_simcols.p;
    std::vector<int32_t> &bins = c->sort_bins;
    bins.assign((size_t)m + 2, 0);
    int nact = 0;
    for (int j = 0; j < n; ++j)
        if (!(((float)gw_host[j] / (float)m) >= 0.8f)) ++bins[std::min(c->h_gaps[j] + c->h_indets[j], m) + 1];
    for (int g = 0; g <= m; ++g) bins[g + 1] += bins[g];
    for (int j = 0; j < n; ++j)
        if (!(((float)gw_host[j] / (float)m) >= 0.8f)) {
            list[bins[std::min(c->h_gaps[j] + c->h_indets[j], m)]++] = j;
            ++nact;
        }
    out->npad = nact;
    return MSA_OK;
}

// ---- the binade-exact similarity pass in three enqueue steps (similarity() runs them back to back on the context's
// stream; msa_trim's pipeline puts the first two on the side stream, beside the pair pass) --------------------------
// 1. column-major codes and the compacted lists of every column's valid rows
int sim_lists_enqueue(msa_ctx *c, int npos, const int32_t *gw_dev, hipStream_t st) {
    const int m = c->m, n = c->n;
    const size_t lsz = (size_t)msak::bx_cols_pad(n) * msak::bx_ldk(m) + 64;
    HIPCHK(c, c->codeT.reserve(lsz));
    HIPCHK(c, c->bx_off.reserve(lsz));
    HIPCHK(c, c->bx_trow.reserve(lsz));
    HIPCHK(c, c->bx_nvalid.reserve((size_t)msak::bx_cols_pad(n) + 64));
    {
        ProfScope pe(c, "encode", st);
        msak::launch_sim_encode_cm(st, c->raw, m, n, c->ld, c->lut.p, gw_dev, c->codeT.p, c->errkey.p);
        msak::launch_bx_compact(st, c->codeT.p, m, n, c->ldw, npos, c->bx_off.p, c->bx_trow.p, c->bx_nvalid.p);
    }
    HIPCHK(c, hipGetLastError());
    return MSA_OK;
}
// 2. the column list (built on the host into h_simcols)
int sim_order_enqueue(msa_ctx *c, const SimOrder &ord, hipStream_t st) {
    HIPCHK(c, c->simcols.reserve((size_t)ord.npad + 8));
    if (ord.npad) HIPCHK(c, hipMemcpyAsync(c->simcols.p, c->h_simcols.p, sizeof(int32_t) * ord.npad, hipMemcpyHostToDevice, st));
    return MSA_OK;
}
// 3. the kernel and the MDK values (context's stream).  gate: device word that, when non-zero, turns the kernel into
//    a no-op (automated1: raised by the identity statistics when they select gappyout); one-column kernel only.
int sim_kernel_enqueue(msa_ctx *c, int npos, const SimOrder &ord, const int32_t *gw_dev, const int *gate) {
    const int m = c->m, n = c->n;
    HIPCHK(c, c->mdk.reserve((size_t)2 * n + 64));  // MDK [n], Q [n]
    HIPCHK(c, c->simnum.reserve((size_t)n + 64));
    HIPCHK(c, c->simden.reserve((size_t)n + 64));
    HIPCHK(c, c->simstate.reserve(msak::lg_state_floats(n)));
    if (!c->have_wbar) {
        // W is there from a compact pass (compact_begin), whose similarity kernel divides the pair pass's row sums itself:
        // the predictor's mean weights of the ordinary kernel have not been computed for this W yet
        HIPCHK(c, c->wbar.reserve((size_t)m + 128));
        msak::launch_w_row_means(c->stream, c->wmat.p, m, c->ldw, c->wbar.p);
        c->have_wbar = true;
    }
    // (no memset of the two sums: the kernel writes every evaluated column, sim_finish does not use the others)
    {
        ProfScope ps(c, "sim");
        const int e = c->tuning.sim_kernel == 1
                          ? msak::launch_similarity_seq(c->stream, c->codeT.p, m, n, c->simcols.p, ord.npad, c->wmat.p, c->ldw, c->tab.p,
                                                        c->simnum.p, c->simden.p)
                          : [&]() {
                                msak::LgAlign a = {};
                                a.voff = c->bx_off.p, a.vtrow = c->bx_trow.p, a.nvalid = c->bx_nvalid.p, a.codeT = c->codeT.p;
                                a.wlow = c->wlow.p, a.wup = c->wmat.p, a.wbar = c->wbar.p;
                                a.num_out = c->simnum.p, a.den_out = c->simden.p, a.state = c->simstate.p;
                                a.gate = gate, a.cols = c->simcols.p;
                                a.ldk = msak::bx_ldk(m), a.m = m, a.n = n, a.ldw = c->ldw, a.ncols = ord.npad;
                                return msak::launch_similarity_lg(c->stream, a, npos, c->tab.p, c->cus, &c->sim_launches);
                            }();
        if (e) return fail_hip(c, (hipError_t)e, "launch_similarity");
    }
    msak::launch_sim_finish(c->stream, c->simnum.p, c->simden.p, gw_dev, m, n, c->mdk.p + n, c->mdk.p);
    HIPCHK(c, hipGetLastError());
    return MSA_OK;
}

bool compact_sim_applies(const msa_ctx *c, int gap_hw);
int compact_begin(msa_ctx *c, const int32_t *vhash, const float *dist, int npos, bool gated);

int similarity(msa_ctx *c, const int32_t *vhash, const float *dist, int npos, const int32_t *gaps_windowed,
               float *mdk_out, float *q_out, msa_err_detail *detail) {
    if (npos < 1 || npos > 28) return MSA_E_INVALID;
    if (!gaps_windowed && !c->order_ready && compact_sim_applies(c, 0)) {  // a small alignment: three launches (compact_begin)
        int rc = compact_begin(c, vhash, dist, npos, false);
        c->pipe_active = false;
        if (rc != MSA_E_FALLBACK) {
            if (rc) {
                (void)hipStreamSynchronize(c->stream);
                return rc;
            }
            c->paths[1] = MSA_PATH_PIPE_COMPACT;
            return fetch_similarity_finish(c, c->n, mdk_out, q_out, detail);
        }
    }
    c->paths[1] = MSA_PATH_PIPE_SERIAL;
    const auto t_begin = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {  // MSA_TRACE=1: host time since the call began
        if (c->tuning.trace)
            std::fprintf(stderr, "[similarity] %-20s at %8.1f us\n", what,
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
    };
    int rc = run_pairs(c, false, true, false);
    if (rc) return rc;
    rc = ensure_gaps(c, true);  // (the kernel's column list is built on the host)
    if (rc) return rc;
    const int m = c->m, n = c->n;
    rc = ensure_tables(c, vhash, dist, npos);
    if (rc) return rc;
    rc = reset_errkey(c);
    if (rc) return rc;
    HIPCHK(c, c->h_u8.reserve(256 + (size_t)std::max(m, n)));
    HIPCHK(c, c->h_f32.reserve(std::max<size_t>((size_t)2 * 29 * 32, (size_t)2 * n + 64)));
    // the vector used for the ">= 80 % gaps" cut
    const int32_t *gw_dev = c->gaps.p;
    if (gaps_windowed) {
        HIPCHK(c, c->gaps_w.reserve((size_t)n + 64));
        HIPCHK(c, c->h_i32.reserve((size_t)2 * n + 4));
        std::memcpy(c->h_i32.p, gaps_windowed, sizeof(int32_t) * n);
        HIPCHK(c, hipMemcpyAsync(c->gaps_w.p, c->h_i32.p, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
        gw_dev = c->gaps_w.p;
    }
    rc = sim_lists_enqueue(c, npos, gw_dev, c->stream);
    if (rc) return rc;
    mark("lists enqueued");
    SimOrder ord;
    if (c->order_ready) {
        ord = c->order;  // built by msa_trim while the pair pass ran
    } else {
        rc = build_sim_order(c, gaps_windowed, &ord);
        if (rc) return rc;
    }
    c->order_ready = false;
    mark("columns sorted");
    rc = sim_order_enqueue(c, ord, c->stream);
    if (rc) return rc;
    rc = sim_kernel_enqueue(c, npos, ord, gw_dev, nullptr);
    if (rc) return rc;
    mark("kernel enqueued");
    rc = fetch_similarity(c, n, mdk_out, q_out, detail);
    mark("results fetched");
    return rc;
}

int overlap(msa_ctx *c, float residue_overlap, float *out) {
    int rc = ensure_gaps(c, false);
    if (rc) return rc;
    const int m = c->m, n = c->n;
    const float fo = residue_overlap * static_cast<float>(m - 1);
    const int need = static_cast<int>(std::ceil(fo));
    HIPCHK(c, c->col_ok.reserve((size_t)3 * c->nchunk + 64));
    HIPCHK(c, c->good.reserve((size_t)m + 64));
    {
        ProfScope ps(c, "overlap");
        msak::launch_overlap(c->stream, c->raw, m, n, c->ld, c->indet, c->gaps.p, c->indets.p, need, c->col_ok.p,
                             c->nchunk, c->good.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->h_i32.reserve((size_t)std::max(m, 2 * n) + 4));
    HIPCHK(c, hipMemcpyAsync(c->h_i32.p, c->good.p, sizeof(int32_t) * m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    for (int i = 0; i < m; ++i) out[i] = static_cast<float>(c->h_i32.p[i]) / n;
    return MSA_OK;
}

// Cleaner::removeAllGapsSeqsAndCols: first sequences (over kept columns), then columns (over
// the updated sequences).
// Residues per sequence over ALL columns, enqueued without waiting (the next synchronisation completes the copy):
// remove_all_gaps can then tell from the host that no sequence can have lost all its residues.
int stage_row_totals(msa_ctx *c, hipStream_t st = nullptr) {
    if (c->rowtot_staged || c->m <= 0 || c->n <= 0) return MSA_OK;
    const bool side = st && st != c->stream;
    if (!st) st = c->stream;
    const int m = c->m, n = c->n;
    HIPCHK(c, c->row_cnt.reserve((size_t)m + 64));
    HIPCHK(c, c->h_rowtot.reserve((size_t)m + 4));
    // (no mask: every column counts)
    msak::launch_row_nongap(st, c->raw, m, n, c->ld, nullptr, c->row_cnt.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_rowtot.p, c->row_cnt.p, sizeof(int32_t) * m, hipMemcpyDeviceToHost, st));
    if (!c->ev_rowtot) HIPCHK(c, hipEventCreateWithFlags(&c->ev_rowtot, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_rowtot, st));
    // (3: on the side stream -- a wait on the context's stream does not cover it until the streams have joined)
    c->rowtot_staged = side ? 3 : 1;
    return MSA_OK;
}

int remove_all_gaps(msa_ctx *c, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info) {
    const int m = c->m, n = c->n;
    // A kept sequence with more residues than there are removed columns keeps at least one of them: when that holds
    // for every kept sequence (residues per sequence staged by stage_row_totals), none is left with gaps only and the
    // pass over the rows is not needed.
    bool rows_safe = false, every_row_kept = true;
    if (c->rowtot_staged == 2) {
        const int removed = n - (int)std::count(keep_res, keep_res + n, 1);
        rows_safe = true;
        for (int i = 0; i < m; ++i) {
            if (keep_seq[i]) rows_safe = rows_safe && c->h_rowtot.p[i] > removed;
            else every_row_kept = false;
        }
    }
    if (rows_safe && every_row_kept && (int)c->h_gaps.size() == n) {
        // all sequences kept: a column is all-gap exactly when its gap count is m -- no pass over the alignment at all
        for (int j = 0; j < n; ++j)
            if (keep_res[j] && c->h_gaps[j] == m) keep_res[j] = 0;
        return MSA_OK;
    }
    if (c->prefetched) {
        // a host-only view (the batch engine): the same two passes over the caller's rows, for the sequences and columns
        // in question only -- the sequences whose residue total does not exceed the number of removed columns, then the
        // columns' counts less what the dropped sequences held
        if (!c->host_rows) return MSA_E_FALLBACK;
        const uint8_t *H = c->host_rows;
        const int64_t hld = c->host_ld;
        const int removed = n - (int)std::count(keep_res, keep_res + n, 1);
        bool all_rows = true;
        for (int i = 0; i < m; ++i) {
            if (keep_seq[i] && !(c->rowtot_staged == 2 && c->h_rowtot.p[i] > removed)) {
                const uint8_t *row = H + (size_t)i * hld;
                bool any = false;
                for (int j = 0; j < n && !any; ++j) any = keep_res[j] && row[j] != '-';
                if (!any) {
                    keep_seq[i] = 0;
                    c->only_gaps_rows.push_back(i);
                    if (info) {
                        if (!(info->warnings & MSA_W_ONLY_GAPS_SEQUENCES)) info->warn_row = i;
                        info->warnings |= MSA_W_ONLY_GAPS_SEQUENCES;
                    }
                }
            }
            all_rows &= keep_seq[i] != 0;
        }
        if ((int)c->h_gaps.size() != n) return MSA_E_FALLBACK;
        if (all_rows) {
            for (int j = 0; j < n; ++j)
                if (keep_res[j] && c->h_gaps[j] == m) keep_res[j] = 0;
            return MSA_OK;
        }
        std::vector<int32_t> left(n);
        for (int j = 0; j < n; ++j) left[j] = m - c->h_gaps[j];  // residues of the column over every sequence
        for (int i = 0; i < m; ++i) {
            if (keep_seq[i]) continue;
            const uint8_t *row = H + (size_t)i * hld;
            for (int j = 0; j < n; ++j) left[j] -= row[j] != '-';
        }
        for (int j = 0; j < n; ++j)
            if (keep_res[j] && left[j] == 0) keep_res[j] = 0;
        return MSA_OK;
    }
    HIPCHK(c, c->keep_res_d.reserve((size_t)n + 64));
    HIPCHK(c, c->keep_seq_d.reserve((size_t)m + 64));
    HIPCHK(c, c->row_cnt.reserve((size_t)m + 64));
    HIPCHK(c, c->col_cnt.reserve((size_t)n + 64));
    HIPCHK(c, c->h_u8.reserve(256 + (size_t)std::max(m, n)));
    HIPCHK(c, c->h_i32.reserve((size_t)std::max(m, 2 * n) + 4));
    if (rows_safe) {
        for (int i = 0; i < m; ++i) c->h_i32.p[i] = 1;  // (no sequence can be empty: skip the pass over the rows)
    } else {
        std::memcpy(c->h_u8.p, keep_res, n);
        HIPCHK(c, hipMemcpyAsync(c->keep_res_d.p, c->h_u8.p, n, hipMemcpyHostToDevice, c->stream));
        msak::launch_row_nongap(c->stream, c->raw, m, n, c->ld, c->keep_res_d.p, c->row_cnt.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_i32.p, c->row_cnt.p, sizeof(int32_t) * m, hipMemcpyDeviceToHost, c->stream));
        SYNC(c);
    }
    bool all_rows = true;
    for (int i = 0; i < m; ++i) {
        if (keep_seq[i] && c->h_i32.p[i] == 0) {
            keep_seq[i] = 0;
            c->only_gaps_rows.push_back(i);
            if (info) {
                if (!(info->warnings & MSA_W_ONLY_GAPS_SEQUENCES)) info->warn_row = i;
                info->warnings |= MSA_W_ONLY_GAPS_SEQUENCES;
            }
        }
        all_rows &= keep_seq[i] != 0;
    }
    if (all_rows && (int)c->h_gaps.size() == n) {
        // every sequence stays: a column is all-gap exactly when its gap count is m (already on the host)
        for (int j = 0; j < n; ++j)
            if (keep_res[j] && c->h_gaps[j] == m) keep_res[j] = 0;
        return MSA_OK;
    }
    const int32_t *col_counts = c->h_i32.p;
    if (c->colcnt_staged && rows_safe) {
        // counted on the device behind the clustering, over the very mask that came back (no row was dropped since)
        col_counts = c->h_colcnt.p;
    } else {
        std::memcpy(c->h_u8.p, keep_seq, m);
        HIPCHK(c, hipMemcpyAsync(c->keep_seq_d.p, c->h_u8.p, m, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->col_cnt.p, 0, sizeof(int32_t) * n, c->stream));
        msak::launch_col_nongap(c->stream, c->raw, m, n, c->ld, c->keep_seq_d.p, c->col_cnt.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_i32.p, c->col_cnt.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
        SYNC(c);
    }
    c->colcnt_staged = false;
    for (int j = 0; j < n; ++j)
        if (keep_res[j] && col_counts[j] == 0) keep_res[j] = 0;
    return MSA_OK;
}

// Residues per column over the sequences the device clustering keeps, enqueued behind it (keep_seq_d is its output):
// remove_all_gaps needs them when sequences were dropped, and would otherwise send the mask back and wait once more.
// Only when no sequence is empty (every ungapped length > 0): then remove_all_gaps cannot drop a sequence itself and
// the mask it counts over is the one the clustering produced.
int stage_kept_column_counts(msa_ctx *c, const std::vector<int32_t> &lengths) {
    c->colcnt_staged = false;
    for (int32_t len : lengths)
        if (len <= 0) return MSA_OK;
    const int m = c->m, n = c->n;
    HIPCHK(c, c->col_cnt.reserve((size_t)n + 64));
    HIPCHK(c, c->h_colcnt.reserve((size_t)n + 4));
    HIPCHK(c, hipMemsetAsync(c->col_cnt.p, 0, sizeof(int32_t) * n, c->stream));
    msak::launch_col_nongap(c->stream, c->raw, m, n, c->ld, c->keep_seq_d.p, c->col_cnt.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_colcnt.p, c->col_cnt.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    c->colcnt_staged = true;
    return MSA_OK;
}

// ungapped lengths and row hashes: enqueued without waiting (row_digest_begin), fetched behind their own event
// (row_digest_end) so that work enqueued in between -- the pair pass -- does not sit in front of the host
int row_digest_begin(msa_ctx *c) {
    const int m = c->m;
    HIPCHK(c, c->lengths.reserve((size_t)m + 64));
    HIPCHK(c, c->hashes.reserve((size_t)2 * m + 64));
    msak::launch_row_digest(c->stream, c->raw, m, c->n, c->ld, c->lengths.p, c->hashes.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->h_len.reserve((size_t)m + 4));
    HIPCHK(c, c->h_u64.reserve((size_t)2 * m + 1));
    HIPCHK(c, hipMemcpyAsync(c->h_len.p, c->lengths.p, sizeof(int32_t) * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_u64.p, c->hashes.p, sizeof(unsigned long long) * 2 * m, hipMemcpyDeviceToHost,
                             c->stream));
    if (!c->ev_digest) HIPCHK(c, hipEventCreateWithFlags(&c->ev_digest, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_digest, c->stream));
    return MSA_OK;
}
int row_digest_end(msa_ctx *c, std::vector<int32_t> &lengths, std::vector<unsigned long long> *hashes) {
    const int m = c->m;
    HIPCHK(c, hipEventSynchronize(c->ev_digest));
    lengths.assign(c->h_len.p, c->h_len.p + m);
    if (hashes) hashes->assign(c->h_u64.p, c->h_u64.p + 2 * m);
    return MSA_OK;
}
int row_digest(msa_ctx *c, std::vector<int32_t> &lengths, std::vector<unsigned long long> *hashes) {
    const int rc = row_digest_begin(c);
    return rc ? rc : row_digest_end(c, lengths, hashes);
}

// Ungapped lengths = the residues per sequence over all columns that stage_row_totals fetches: waits for that copy
// alone (its event), not for what was enqueued behind it.
int ungapped_lengths(msa_ctx *c, std::vector<int32_t> &lengths) {
    int rc = stage_row_totals(c);
    if (rc) return rc;
    if (c->rowtot_staged == 1 || c->rowtot_staged == 3) {
        HIPCHK(c, hipEventSynchronize(c->ev_rowtot));
        c->rowtot_staged = 2;
    }
    lengths.assign(c->h_rowtot.p, c->h_rowtot.p + c->m);
    return MSA_OK;
}

// Cleaner::removeDuplicates (as patched by the reference): the earlier of two identical rows goes.
int remove_duplicates(msa_ctx *c, uint8_t *keep_seq) {
    const int m = c->m;
    std::vector<int32_t> lengths;
    std::vector<unsigned long long> hashes;
    int rc = row_digest(c, lengths, &hashes);
    if (rc) return rc;
    // candidate pairs: for each row i, the first later row x with equal digest
    struct Key {
        unsigned long long a, b;
        bool operator==(const Key &o) const { return a == o.a && b == o.b; }
    };
    struct KeyHash {
        size_t operator()(const Key &k) const { return static_cast<size_t>(k.a ^ (k.b * 0x9E3779B97F4A7C15ull)); }
    };
    std::unordered_map<Key, std::vector<int>, KeyHash> groups;
    for (int i = 0; i < m; ++i) groups[Key{hashes[2 * i], hashes[2 * i + 1]}].push_back(i);
    std::vector<int32_t> pairs;
    for (auto &kv : groups) {
        const std::vector<int> &g = kv.second;
        for (size_t a = 0; a < g.size(); ++a)
            for (size_t b = a + 1; b < g.size(); ++b) {
                pairs.push_back(g[a]);
                pairs.push_back(g[b]);
            }
    }
    const int npairs = static_cast<int>(pairs.size() / 2);
    if (npairs == 0) return MSA_OK;
    HIPCHK(c, c->pairs.reserve(pairs.size()));
    HIPCHK(c, c->equal.reserve(npairs));
    HIPCHK(c, hipMemcpyAsync(c->pairs.p, pairs.data(), pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    msak::launch_rows_equal(c->stream, c->raw, c->n, c->ld, c->pairs.p, npairs, c->equal.p);
    HIPCHK(c, hipGetLastError());
    std::vector<int32_t> equal(npairs);
    HIPCHK(c, hipMemcpyAsync(equal.data(), c->equal.p, npairs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    for (int p = 0; p < npairs; ++p)
        if (equal[p]) keep_seq[std::min(pairs[2 * p], pairs[2 * p + 1])] = 0;  // a later identical row exists
    return MSA_OK;
}

// Cleaner::calculateRepresentativeSeq with a fixed threshold, without moving the m*m matrix:
// lengths to the host (m ints), trimAl's processing order back (m ints), adjacency bits + the
// round-based independent-set kernel on the device, mask to the host (m bytes).
int device_representatives(msa_ctx *c, float max_identity, uint8_t *keep_seq) {
    const int m = c->m;
    if (m < 2) return MSA_E_FALLBACK;
    // the ungapped lengths first (msa_trim staged them), then the pair pass: the processing order is sorted on the
    // host while it runs
    int rc = stage_row_totals(c);
    if (rc) return rc;
    rc = run_pairs(c, true, false, false);
    if (rc) return rc;
    std::vector<int32_t> lengths;
    rc = ungapped_lengths(c, lengths);
    if (rc) return rc;
    const std::vector<int32_t> seq_at = msah::processing_order(lengths.data(), m);
    HIPCHK(c, c->pairs.reserve((size_t)2 * m + 64));
    HIPCHK(c, c->col_ok.reserve(msak::cluster_adj_buffer_words(m) + 64));
    HIPCHK(c, c->keep_seq_d.reserve((size_t)m + 64));
    HIPCHK(c, c->equal.reserve(4));
    HIPCHK(c, c->h_i32.reserve((size_t)std::max(2 * m, 2 * c->n) + 4));
    std::memcpy(c->h_i32.p, seq_at.data(), sizeof(int32_t) * m);
    for (int t = 0; t < m; ++t) c->h_i32.p[m + seq_at[t]] = t;  // the inverse: where each sequence stands in the order
    HIPCHK(c, hipMemcpyAsync(c->pairs.p, c->h_i32.p, sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->equal.p, 0, sizeof(int32_t), c->stream));
    {
        ProfScope ps(c, "cluster");
        if (msak::launch_cluster(c->stream, c->ident.p, c->ldw, c->pairs.p, m, max_identity, c->col_ok.p,
                                 c->keep_seq_d.p, c->equal.p) != 0)
            return MSA_E_FALLBACK;  // too many sequences for the LDS bit sets: host path
    }
    HIPCHK(c, hipGetLastError());
    if ((rc = stage_kept_column_counts(c, lengths))) return rc;
    HIPCHK(c, c->h_u8.reserve(256 + (size_t)std::max(m, c->n)));
    HIPCHK(c, hipMemcpyAsync(c->h_u8.p, c->keep_seq_d.p, m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    std::memcpy(keep_seq, c->h_u8.p, m);
    return MSA_OK;
}

// RepresentativeTrimmer(clusters=K): Cleaner::getCutPointClusters bisects the identity threshold until the greedy
// clustering yields K clusters.  Start value and bounds come from the row statistics (the start value is the
// selectMethod mean, same order of operations), every probe is one run of the device clustering: no m*m transfer.
int device_cluster_count(msa_ctx *c, int clusters, uint8_t *keep_seq) {
    const int m = c->m;
    if (m < 2 || clusters < 1) return MSA_E_FALLBACK;
    // below ~2000 sequences the m*m copy (< 16 MB) is cheaper than a synchronisation per probe: host path
    if (c->tuning.device_clusters == 0 || (m < 2000 && c->tuning.device_clusters < 0)) return MSA_E_FALLBACK;
    int rc = run_pairs(c, true, false, false);
    if (rc) return rc;
    const size_t words = msak::cluster_adj_words(m);
    if ((size_t)4 * words * sizeof(uint32_t) > 60 * 1024) return MSA_E_FALLBACK;  // host path
    HIPCHK(c, c->row_avg.reserve(m + 64));
    HIPCHK(c, c->row_max.reserve(m + 64));
    HIPCHK(c, c->row_min.reserve(m + 64));
    {
        ProfScope ps(c, "idstats");
        msak::launch_identity_stats(c->stream, c->ident.p, m, c->ldw, c->row_avg.p, c->row_max.p, c->stats2.p, c->row_min.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->h_f32.reserve((size_t)2 * m + 8));
    HIPCHK(c, hipMemcpyAsync(c->h_f32.p, c->stats2.p, 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_f32.p + 2, c->row_max.p, sizeof(float) * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_f32.p + 2 + m, c->row_min.p, sizeof(float) * m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    std::vector<int32_t> lengths;
    rc = ungapped_lengths(c, lengths);
    if (rc) return rc;
    float threshold = c->h_f32.p[0], hi = 0, lo = 1;
    for (int i = 0; i < m; ++i) {
        hi = std::max(hi, c->h_f32.p[2 + i]);
        lo = std::min(lo, c->h_f32.p[2 + m + i]);
    }
    if (clusters == m) threshold = 1;
    if (clusters == 1) threshold = 0;
    const std::vector<int32_t> seq_at = msah::processing_order(lengths.data(), m);
    HIPCHK(c, c->pairs.reserve((size_t)2 * m + 64));
    HIPCHK(c, c->col_ok.reserve(msak::cluster_adj_buffer_words(m) + 64));
    HIPCHK(c, c->keep_seq_d.reserve((size_t)m + 64));
    HIPCHK(c, c->equal.reserve(4));
    HIPCHK(c, c->h_i32.reserve((size_t)std::max(2 * m, 2 * c->n) + 4));
    std::memcpy(c->h_i32.p, seq_at.data(), sizeof(int32_t) * m);
    for (int t = 0; t < m; ++t) c->h_i32.p[m + seq_at[t]] = t;  // the inverse: where each sequence stands in the order
    HIPCHK(c, hipMemcpyAsync(c->pairs.p, c->h_i32.p, sizeof(int32_t) * 2 * m, hipMemcpyHostToDevice, c->stream));
    float previous = 0, stalled = 0;
    for (;;) {
        HIPCHK(c, hipMemsetAsync(c->equal.p, 0, sizeof(int32_t), c->stream));
        {
            ProfScope ps(c, "cluster");
            if (msak::launch_cluster(c->stream, c->ident.p, c->ldw, c->pairs.p, m, threshold, c->col_ok.p,
                                     c->keep_seq_d.p, c->equal.p) != 0)
                return MSA_E_FALLBACK;
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_i32.p, c->equal.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        SYNC(c);
        const int count = c->h_i32.p[0];
        if (clusters == m || clusters == 1) break;  // the bounds of getCutPointClusters: no search
        if (count == clusters || stalled > 10) break;
        if (count > clusters) hi = threshold;
        else lo = threshold;
        threshold = (hi + lo) / 2;
        if (previous != count) {
            stalled = 0;
            previous = static_cast<float>(count);
        } else {
            ++stalled;
        }
    }
    if ((rc = stage_kept_column_counts(c, lengths))) return rc;
    HIPCHK(c, c->h_u8.reserve(256 + (size_t)std::max(m, c->n)));
    HIPCHK(c, hipMemcpyAsync(c->h_u8.p, c->keep_seq_d.p, m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    std::memcpy(keep_seq, c->h_u8.p, m);
    return MSA_OK;
}

int fetch_ident(msa_ctx *c, std::vector<float> &host) {  // dense m*m copy of the identity matrix
    int rc = run_pairs(c, true, false, false);
    if (rc) return rc;
    host.resize((size_t)c->m * c->m);
    HIPCHK(c, hipMemcpy2DAsync(host.data(), (size_t)c->m * sizeof(float), c->ident.p, (size_t)c->ldw * sizeof(float),
                               (size_t)c->m * sizeof(float), c->m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    return MSA_OK;
}

// ---- msa_trim's similarity pipeline ----------------------------------------------------------------------------
// A trim that needs (or, for automated1, may need) the similarity values enqueues everything before its first wait:
//
//   context's stream:  state memset, gap counts + copy (ev_gaps) | fork | planes, pair pass, [identity statistics ->
//                      gate] | join | similarity kernel (skipped on the device when the gate is up), MDK, result copy
//   side stream:       row totals + copy (ev_rowtot), column-major codes, compacted lists, column list
//
// and waits once, for everything.  The host work that depends on the gap counts alone (their window, the gap cut, the
// column order) runs while the pair pass does.  gated: automated1 -- Cleaner::selectMethod's decision is also taken on
// the device (identity_final_kernel), so that the similarity kernel can sit in the queue behind the statistics
// without a round trip to the host; the host takes the same decision from the same two floats after the wait.
bool sim_pipeline_applies(const msa_ctx *c, const msa_trim_params *p, int sim_hw) {
    return p->vhash && p->dist && p->npos >= 1 && p->npos <= 28 && c->m >= 2 && sim_hw <= c->n / 4 && c->tuning.pipeline != 0;
}

int sim_pipeline_begin(msa_ctx *c, const msa_trim_params *p, int gap_hw, bool gated, std::vector<int32_t> &gaps_w) {
    const int n = c->n;
    int rc = ensure_tables(c, p->vhash, p->dist, p->npos);
    if (rc) return rc;
    if ((rc = reset_errkey(c))) return rc;
    if ((rc = stage_gaps(c))) return rc;
    // A pair pass of a few hundred microseconds hides the side stream's work and what it costs the host to enqueue it
    // (~80 us: events, a second queue); below that everything goes on the context's stream, the small kernels in
    // front of the pair pass (the host sorts the column order while they run).
    const bool forked = c->tuning.pipeline == 3 || ((double)c->m * c->m * n >= 2e9 && c->tuning.pipeline != 2);
    hipStream_t side = c->stream;
    if (forked) {
        if (!c->stream2) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        side = c->stream2;
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        // the context's stream first: the pair pass starts while the side stream is being filled
        if ((rc = run_pairs(c, gated, true, false))) return rc;
        HIPCHK(c, hipStreamWaitEvent(side, c->ev_fork, 0));
    }
    if ((rc = stage_row_totals(c, side))) return rc;
    const int32_t *gw_dev = c->gaps.p;
    if (gap_hw == 0 && (rc = sim_lists_enqueue(c, p->npos, gw_dev, side))) return rc;  // (needs nothing from the host)
    if (!forked && (rc = run_pairs(c, gated, true, false))) return rc;
    // host: the gap counts arrive behind their own event while the pair pass runs
    if ((rc = ensure_gaps(c, true))) return rc;
    gaps_w.resize(n);
    if ((rc = msah::window_i32(c->h_gaps.data(), n, gap_hw, gaps_w.data()))) return rc;
    if (gap_hw > 0) {
        HIPCHK(c, c->gaps_w.reserve((size_t)n + 64));
        HIPCHK(c, c->h_i32.reserve((size_t)std::max(c->m, 2 * n) + 4));
        std::memcpy(c->h_i32.p, gaps_w.data(), sizeof(int32_t) * n);
        HIPCHK(c, hipMemcpyAsync(c->gaps_w.p, c->h_i32.p, sizeof(int32_t) * n, hipMemcpyHostToDevice, side));
        gw_dev = c->gaps_w.p;
        if ((rc = sim_lists_enqueue(c, p->npos, gw_dev, side))) return rc;
    }
    SimOrder ord;
    if ((rc = build_sim_order(c, gap_hw > 0 ? gaps_w.data() : nullptr, &ord))) return rc;
    if ((rc = sim_order_enqueue(c, ord, side))) return rc;
    if (forked) HIPCHK(c, hipEventRecord(c->ev_join, side));
    int *gate = nullptr;
    if (gated) {
        HIPCHK(c, c->row_avg.reserve(c->m + 64));
        HIPCHK(c, c->row_max.reserve(c->m + 64));
        gate = c->state.p + ST_GATE;
        ProfScope ps(c, "idstats");
        msak::launch_identity_stats(c->stream, c->ident.p, c->m, c->ldw, c->row_avg.p, c->row_max.p, c->stats2.p, nullptr, gate);
    }
    HIPCHK(c, hipGetLastError());
    if (forked) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if (c->rowtot_staged == 3) c->rowtot_staged = 1;  // (joined: a wait on the context's stream now covers the copy)
    if ((rc = sim_kernel_enqueue(c, p->npos, ord, gw_dev, gate))) return rc;
    if ((rc = fetch_similarity_enqueue(c, n))) return rc;
    c->pipe_active = true;
    c->pipe_gated = gated;
    c->paths[1] = forked ? MSA_PATH_PIPE_TWO_STREAMS : MSA_PATH_PIPE_ONE_STREAM;
    return MSA_OK;
}


// ---- the compact pipeline of a small alignment -------------------------------------------------------------------------
// A trim of 46 x 1181 residues spent 0.15 ms on ~18 queue operations (memsets, a dozen launches, four copies, events) around
// 0.08 ms of kernels (profiles/r04_small_latency_ordinary_launch_sequence.jsonl).  Up to `compact_max_m` sequences, with every
// column's wave resident at once (no column order needed) and no gap window, the same statistics take THREE launches and no copy
// (msak::CompactArgs):
//   front  -- gap / indetermination counts, residues per sequence, planes, column-major codes (+ lists; the ">= 80 % gaps" cut
//             from the block's own counts), all from one pass over the rows by independent blocks; no memset in front of it;
//   pairs  -- the ordinary pair pass, whose tiles also add up the rows' weight sums (the predictor's input);
//   [automated1: the identity statistics with the selectMethod gate, one launch]
//   sim    -- up to `flat_max_m` sequences the flat kernel, else the wave-per-column kernel over the columns in their own order;
//             the wave that finishes a column writes its MDK and Q;
//   every result is stored by the kernels into a mirror of the state block in pinned host memory (h_cres): one wait, then the
//   host folds the front kernel's per-block verdicts into the two flag words.
// The kernels' arithmetic is the ordinary path's (the same device functions); MSA_COMPACT=0 switches the pipeline off.
bool compact_sim_applies(const msa_ctx *c, int gap_hw) {
    const msak::Tuning &t = c->tuning;
    return t.compact != 0 && gap_hw == 0 && t.sim_kernel == 0 && (t.sim_mode & 64) == 0 && t.lg_rounds < 0 && t.lg_split == 0 &&
           t.lg_big == 0 && c->m >= 2 && c->m <= std::min(t.compact_max_m, 512) && c->n <= c->cus * 20 &&
           msak::pair_pipe_regime(c->m, c->m_pad);
}
bool compact_gaps_applies(const msa_ctx *c) {
    // (up to 1024 sequences: a column block walks its 64-row tiles four at a time, and beyond sixteen tiles that chain is longer
    // than the launches it saves)
    return c->tuning.compact != 0 && !c->have_gaps && c->m >= 1 && c->m <= 1024 && (size_t)c->m * (size_t)c->ld <= ((size_t)4 << 20);
}
msak::CompactArgs compact_args(msa_ctx *c) {
    msak::CompactArgs a = {};
    a.raw = c->raw, a.ld = c->ld, a.m = c->m, a.n = c->n;
    a.indet4 = 0x01010101u * c->indet;
    a.flags = c->state.p;
    a.gaps = c->gaps.p, a.indets = c->indets.p;
    a.hres = c->h_cres.p;
    a.h_gaps = ST_WORDS, a.h_indets = ST_WORDS + c->state_npad, a.h_rowtot = ST_WORDS + 2 * c->state_npad;
    a.h_slots = ST_WORDS + 4 * c->state_npad + c->state_rpad;
    a.scratch = c->cscratch.p;
    a.wsum = reinterpret_cast<uint32_t *>(c->cscratch.p + 2);
    a.ncols_pad = msak::bx_cols_pad(c->n);
    return a;
}
// sizes the state block and its host mirror; the mirror's flag words start at zero (the kernels only ever raise them)
int compact_prepare(msa_ctx *c) {
    int rc = layout_state(c);
    if (rc) return rc;
    HIPCHK(c, c->cscratch.reserve(msak::compact_scratch_words(c->m, c->n)));
    HIPCHK(c, c->h_cres.reserve(mirror_words(c)));
    std::memset(c->h_cres.p, 0, ST_WORDS * sizeof(int32_t));
    return MSA_OK;
}
// the one wait, and every host-side cache a pipelined trim reads filled from the mirror
int compact_fetch(msa_ctx *c, bool sim) {
    const int m = c->m, n = c->n;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->upload_pending = false;
    int32_t *H = c->h_cres.p;
    if (sim) {
        // the verdicts of the front kernel's blocks -> the two flag words (what atomicOr / atomicMax leave there in the ordinary path)
        const int32_t *S = H + ST_WORDS + 4 * c->state_npad + c->state_rpad;
        const int ncb = msak::bx_cols_pad(n) / 64;
        int bad = 0;
        unsigned long long key = 0;
        for (int i = 0; i < ncb; ++i) {
            unsigned long long k;
            std::memcpy(&k, S + 2 * i, sizeof(k));
            key = std::max(key, k);
            bad |= S[2 * ncb + i];
        }
        H[ST_ERRFLAG] = bad ? 1 : 0;
        std::memcpy(H + ST_ERRKEY, &key, sizeof(key));
    }
    HIPCHK(c, c->h_flags.reserve(ST_FLAGS));
    std::memcpy(c->h_flags.p, H, ST_FLAGS * sizeof(int32_t));
    c->h_gaps.assign(H + ST_WORDS, H + ST_WORDS + n);
    c->h_indets.assign(H + ST_WORDS + c->state_npad, H + ST_WORDS + c->state_npad + n);
    HIPCHK(c, c->h_rowtot.reserve((size_t)m + 4));
    std::memcpy(c->h_rowtot.p, H + ST_WORDS + 2 * c->state_npad, sizeof(int32_t) * m);
    c->rowtot_staged = 2;
    c->gaps_staged = 0;
    c->have_gaps = true;
    c->state_zeroed = true;  // (the front kernel wrote the device's flag words and both count vectors)
    c->flags_dirty = false;
    if (sim) {
        HIPCHK(c, c->h_f32.reserve(std::max<size_t>((size_t)2 * 29 * 32 + 64, (size_t)2 * n + 64)));
        const int32_t *F = H + ST_WORDS + 2 * c->state_npad + c->state_rpad;
        std::memcpy(c->h_f32.p, F, sizeof(float) * n);
        std::memcpy(c->h_f32.p + n, F + c->state_npad, sizeof(float) * n);
    }
    return MSA_OK;
}
// the gap statistics alone: one launch
int compact_gaps(msa_ctx *c) {
    int rc = compact_prepare(c);
    if (rc) return rc;
    msak::CompactArgs a = compact_args(c);
    {
        ProfScope ps(c, "gaps");
        msak::launch_compact_front(c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    c->errkey_dirty = false;
    return compact_fetch(c, false);
}
int compact_begin(msa_ctx *c, const int32_t *vhash, const float *dist, int npos, bool gated) {
    const int m = c->m, n = c->n;
    int rc = ensure_tables(c, vhash, dist, npos);
    if (rc) return rc;
    if ((rc = compact_prepare(c))) return rc;
    if ((rc = pair_buffers(c, gated, true))) return rc;
    const size_t lsz = (size_t)msak::bx_cols_pad(n) * msak::bx_ldk(m) + 64;
    HIPCHK(c, c->planes.reserve((size_t)msak::planes_total() * c->nchunk * c->m_pad + 64));
    HIPCHK(c, c->codeT.reserve(lsz));
    HIPCHK(c, c->bx_off.reserve(lsz));
    HIPCHK(c, c->bx_trow.reserve(lsz));
    HIPCHK(c, c->bx_nvalid.reserve((size_t)msak::bx_cols_pad(n) + 64));
    HIPCHK(c, c->simnum.reserve((size_t)n + 64));
    HIPCHK(c, c->simden.reserve((size_t)n + 64));
    if (gated) {
        HIPCHK(c, c->row_avg.reserve(m + 64));
        HIPCHK(c, c->row_max.reserve(m + 64));
    }
    msak::CompactArgs a = compact_args(c);
    a.sim = 1;
    a.lut = c->lut.p;
    a.planes = c->planes.p, a.nchunk = c->nchunk, a.m_pad = c->m_pad;
    a.codeT = c->codeT.p, a.ldk = msak::bx_ldk(m);
    a.voff = c->bx_off.p, a.vtrow = c->bx_trow.p, a.nvalid = c->bx_nvalid.p;
    a.ldw = c->ldw, a.skiprow = npos, a.big = 0;
    a.ident = c->ident.p, a.row_avg = c->row_avg.p, a.row_max = c->row_max.p;
    a.gated = gated ? 1 : 0;
    float *mdk = reinterpret_cast<float *>(c->h_cres.p + ST_WORDS + 2 * c->state_npad + c->state_rpad);  // (the host mirror)
    msak::LgAlign L = {};
    L.voff = a.voff, L.vtrow = a.vtrow, L.nvalid = a.nvalid, L.codeT = a.codeT;
    L.wlow = c->wlow.p, L.wup = c->wmat.p, L.wbar = c->wbar.p, L.wsum = a.wsum;
    L.num_out = c->simnum.p, L.den_out = c->simden.p;
    L.gate = gated ? c->state.p + ST_GATE : nullptr;
    L.mdk_out = mdk, L.q_out = mdk + c->state_npad, L.mdk_host = c->tuning.mdk_host;
    L.ldk = a.ldk, L.m = m, L.n = n, L.ldw = c->ldw, L.ncols = n;  // (cols = null: every column, in its own order)
    // the flat kernel up to flat_max_m sequences -- with two waves per column, and from half that size on, only while both waves of
    // every column are resident at once (128 x 5000: 106 us against 70 for the wave-per-column kernel; 64 x 5000: 46 against 58)
    const int flat_m = std::min(c->tuning.flat_max_m, msak::flat_rows_max());
    const bool flat = m <= flat_m && (n <= c->cus * 10 || m <= flat_m / 2);
    a.lists = flat ? 0 : 1;
    // (compact_sim_applies keeps the shapes and switches out for which the wave-per-column kernel would not finish the columns
    // itself; should the two ever disagree, nothing has been enqueued yet and the callers take the ordinary pipeline)
    if (!flat && !msak::lg_finishes(L, c->cus)) return MSA_E_FALLBACK;
    {
        ProfScope ps(c, "front");
        msak::launch_compact_front(c->stream, a);
    }
    {
        ProfScope ps(c, "pairs");
        msak::launch_pair_counts(c->stream, c->planes.p, c->nchunk, c->m_pad, m, c->ldw, nullptr, nullptr, gated ? c->ident.p : nullptr,
                                 c->wmat.p, c->wlow.p, c->h_cres.p + ST_PAIRFLAG, a.wsum);
    }
    if (gated) {
        ProfScope ps(c, "idstats");
        msak::launch_compact_identity(c->stream, a);
    }
    {
        ProfScope ps(c, "sim");
        if (flat) {
            msak::launch_similarity_flat(c->stream, L, c->tab.p);
            c->sim_launches = 1;
        } else {
            const int e = msak::launch_similarity_lg(c->stream, L, npos, c->tab.p, c->cus, &c->sim_launches);
            if (e) return fail_hip(c, (hipError_t)e, "launch_similarity");
        }
    }
    HIPCHK(c, hipGetLastError());
    c->errkey_dirty = true;
    if ((rc = compact_fetch(c, true))) return rc;
    c->have_planes = true;
    c->planes_pending = false;
    c->have_w = true;
    if (gated) c->have_ident = true;
    c->pairflag_state = 2;  // (the flag words are on the host)
    if (c->h_flags.p[ST_ERRFLAG]) {
        c->have_planes = false;
        return MSA_E_NON_ASCII;
    }
    c->pipe_active = true;
    c->pipe_gated = gated;
    return MSA_OK;
}

}  // namespace

extern "C" {

const char *msa_strerror(int code) {
    switch (code) {
        case MSA_OK: return "ok";
        case MSA_E_INVALID: return "invalid argument or call order";
        case MSA_E_NO_DEVICE: return "no HIP device available";
        case MSA_E_HIP: return "HIP runtime error";
        case MSA_E_NOMEM: return "out of memory";
        case MSA_E_WINDOW_TOO_BIG: return "window size is too big for this alignment";
        case MSA_E_INCORRECT_SYMBOL: return "incorrect symbol in the alignment";
        case MSA_E_UNDEFINED_SYMBOL: return "symbol not defined in the similarity matrix";
        case MSA_E_NOT_IMPLEMENTED: return "method not implemented";
        case MSA_E_NON_ASCII: return "non-ASCII byte in the alignment";
        case MSA_E_LENGTH_MISMATCH: return "sequences of different lengths";
        case MSA_E_BAD_RESIDUE: return "unknown character in a sequence";
        default: return "unknown error";
    }
}

int msa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *msa_last_hip_error(const msa_ctx *ctx) { return ctx ? ctx->hip_err : ""; }

int msa_ctx_create(int device, msa_ctx **out) {
    if (!out) return MSA_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MSA_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MSA_E_NO_DEVICE;
    msa_ctx *c = new (std::nothrow) msa_ctx();
    if (!c) return MSA_E_NOMEM;
    c->device = device;
    c->tuning = msak::tuning_from_env();
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return MSA_E_HIP;
    }
    *out = c;
    return MSA_OK;
}

void msa_ctx_destroy(msa_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (hipEvent_t ev : c->event_pool) (void)hipEventDestroy(ev);
    c->raw_own.release(); c->planes.release(); c->state.release(); c->h_flags.release(); c->tables.release(); c->ident.release();
    c->wmat.release(); c->wlow.release(); c->wbar.release(); c->codeT.release(); c->simcols.release(); c->h_simcols.release(); c->bx_off.release(); c->bx_trow.release(); c->bx_nvalid.release(); c->hit.release(); c->dst.release(); c->row_avg.release(); c->row_max.release(); c->row_min.release();
    c->gaps_w.release(); c->cscratch.release(); c->h_cres.release();
    c->mdk.release(); c->simnum.release(); c->simden.release(); c->simstate.release(); c->col_ok.release();
    c->good.release(); c->row_cnt.release(); c->col_cnt.release(); c->lengths.release(); c->pairs.release();
    c->equal.release(); c->keep_res_d.release(); c->keep_seq_d.release(); c->hashes.release();
    c->h_i32.release(); c->h_f32.release(); c->h_u64.release(); c->h_u8.release(); c->h_raw.release();
    c->h_gapstage.release(); c->h_rowtot.release(); c->h_len.release(); c->h_colcnt.release();
    if (c->ev_gaps) (void)hipEventDestroy(c->ev_gaps);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
    if (c->ev_digest) (void)hipEventDestroy(c->ev_digest);
    if (c->ev_rowtot) (void)hipEventDestroy(c->ev_rowtot);
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipEventDestroy(c->ev_fork);
        (void)hipEventDestroy(c->ev_join);
        (void)hipStreamDestroy(c->stream2);
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

void *msa_ctx_stream(msa_ctx *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

int msa_ctx_sync(msa_ctx *c) {
    if (!c) return MSA_E_INVALID;
    SYNC(c);
    return MSA_OK;
}

}  // extern "C"

namespace {
// A few helper threads that pack upload pieces (memcpy into pinned staging) beside the calling thread: one core copies
// ~12 - 35 GB/s from pageable memory, the link takes > 50 GB/s.  Process-wide, created on first use, never destroyed
// (the threads sleep on a condition variable; a leaked singleton has no destruction-order problems at exit).
struct PackJob {
    std::atomic<int> next{0};
    int npieces = 0;
    std::function<void(int)> pack;
    std::unique_ptr<std::atomic<unsigned char>[]> done;
};
class PackPool {
  public:
    static PackPool &get() {
        static PackPool *pool = new PackPool();
        return *pool;
    }
    void submit(const std::shared_ptr<PackJob> &job) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
    }
    // the job is packed: drop it if no helper has yet (a helper pops an exhausted job only when it next looks at the queue;
    // in a forked child there are no helpers at all)
    void retire(const std::shared_ptr<PackJob> &job) {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
            if (it->get() == job.get()) {
                jobs_.erase(it);
                break;
            }
    }
    int helpers() const { return (int)threads_.size(); }

  private:
    PackPool() {
        const char *e = std::getenv("MSA_PACK_THREADS");
        int k = e ? std::atoi(e) : 3;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0) k = std::min(k, std::max(0, hw - 1));
        for (int i = 0; i < k; ++i) {
            threads_.emplace_back([this] { run(); });
            threads_.back().detach();
        }
    }
    void run() {
        for (;;) {
            std::shared_ptr<PackJob> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !jobs_.empty(); });
                job = jobs_.front();
                if (job->next.load(std::memory_order_relaxed) >= job->npieces) {
                    jobs_.pop_front();
                    continue;
                }
            }
            for (;;) {
                const int p = job->next.fetch_add(1, std::memory_order_relaxed);
                if (p >= job->npieces) break;
                job->pack(p);
                job->done[p].store(1, std::memory_order_release);
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<PackJob>> jobs_;
    std::vector<std::thread> threads_;
};

// Rows -> the device's pitched layout.  The rows are re-pitched on the host into pinned staging (a pitched copy from
// pageable memory degenerates into a transfer per row when the rows are not 16-byte multiples: 1.5 ms for 209 x 1227
// instead of 0.05 ms) in pieces of about 1 MB, each sent as soon as it is packed: the copy engine moves piece k while
// the host packs the pieces behind it -- the calling thread and the helper threads of PackPool, pieces taken in order
// (20 MB: 1.0 ms packed by one thread before the copy, 0.56 ms piece by piece, see DESIGN.md for the helpers).
// row(i) -> pointer to the n bytes of row i.
template <typename RowAt>
int upload_rows_pitched(msa_ctx *c, int m, int n, RowAt row) {
    const size_t bytes = (size_t)m * c->ld;
    HIPCHK(c, c->h_raw.reserve(bytes));
    const int rows_per_piece = c->tuning.upload_piece_kb > 0
                                   ? std::max<int>(1, (int)(((size_t)c->tuning.upload_piece_kb << 10) / (size_t)c->ld))
                                   : std::max(m, 1);
    const int npieces = (m + rows_per_piece - 1) / rows_per_piece;
    auto pack_piece = [c, m, n, rows_per_piece, row](int p) {
        const int i0 = p * rows_per_piece, i1 = std::min(m, i0 + rows_per_piece);
        for (int i = i0; i < i1; ++i) {
            uint8_t *dst = c->h_raw.p + (size_t)i * c->ld;
            std::memcpy(dst, row(i), (size_t)n);
            std::memset(dst + n, 0, (size_t)(c->ld - n));
        }
    };
    auto send_piece = [&](int p) -> int {
        const int i0 = p * rows_per_piece, i1 = std::min(m, i0 + rows_per_piece);
        HIPCHK(c, hipMemcpyAsync(c->raw_own.p + (size_t)i0 * c->ld, c->h_raw.p + (size_t)i0 * c->ld, (size_t)(i1 - i0) * c->ld,
                                 hipMemcpyHostToDevice, c->stream));
        return MSA_OK;
    };
    PackPool &pool = PackPool::get();
    if (npieces < 4 || pool.helpers() == 0) {  // small: the calling thread alone
        for (int p = 0; p < npieces; ++p) {
            pack_piece(p);
            const int rc = send_piece(p);
            if (rc) return rc;
        }
        return MSA_OK;
    }
    auto job = std::make_shared<PackJob>();
    job->npieces = npieces;
    job->pack = pack_piece;
    job->done.reset(new std::atomic<unsigned char>[npieces]);
    for (int p = 0; p < npieces; ++p) job->done[p].store(0, std::memory_order_relaxed);
    pool.submit(job);
    int rc = MSA_OK;
    for (int p = 0; p < npieces; ++p) {
        // help until piece p is packed (by whoever took it), then send it: the copies go out in order
        while (!job->done[p].load(std::memory_order_acquire)) {
            const int q = job->next.fetch_add(1, std::memory_order_relaxed);
            if (q < npieces) {
                pack_piece(q);
                job->done[q].store(1, std::memory_order_release);
            } else {
                std::this_thread::yield();
            }
        }
        if (rc == MSA_OK) rc = send_piece(p);  // (after an error: keep draining, the helpers still write into h_raw)
    }
    pool.retire(job);
    return rc;
}

// Small alignments are not copied to the device at all: the rows are packed into the context's pinned staging area and the
// kernels read them there, over the link (a copy costs the queue ~8 us whatever its size; the compact pipeline reads the rows
// three times, 54 KB each at 46 x 1181).  Nothing is enqueued: no event, nothing for msa_trim to wait for.
bool zero_copy_rows(const msa_ctx *c, int m, int n) {
    return m > 0 && n > 0 && c->tuning.zerocopy_kb > 0 && (size_t)m * (size_t)c->ld <= ((size_t)c->tuning.zerocopy_kb << 10);
}
template <typename RowAt>
int stage_rows_zero_copy(msa_ctx *c, int m, int n, RowAt row) {
    // (every entry point returns with nothing in flight on the context's stream: no kernel still reads the staging area)
    HIPCHK(c, c->h_raw.reserve((size_t)m * c->ld + 256));
    for (int i = 0; i < m; ++i) {
        uint8_t *dst = c->h_raw.p + (size_t)i * c->ld;
        std::memcpy(dst, row(i), (size_t)n);
        std::memset(dst + n, 0, (size_t)(c->ld - n));
    }
    c->raw = c->h_raw.p;  // (hipHostMalloc memory: the device reads it at the same address)
    return MSA_OK;
}
}  // namespace

extern "C" {

// Host rows -> the device's pitched layout, by where the rows live (measured on an MI355X, 2000 x 10000:
// tools/ubench_register.hip, tools/upload_sweep.sh -> profiles/r03_upload.txt):
//   * page-locked memory (hipHostMalloc, or registered by the caller: msa_host_register): ONE pitched copy straight from the
//     caller's rows, no staging, no packing -- 0.37 ms, 54 GB/s, the link's rate;
//   * pageable rows of a multiple of 16 bytes at a 16-byte aligned address: the runtime's own pitched copy (it stages
//     internally): 0.52 ms;
//   * anything else: packed into pinned staging piece by piece by the calling thread and PackPool's helpers, each piece
//     sent as soon as it is packed: 0.61 - 0.69 ms (a pitched copy of odd-sized pageable rows degenerates into one
//     transfer per row: 1.5 ms for 209 x 1227).
static int zero_padding_for_shape(msa_ctx *c, int m, int n) {
    const uint64_t pad_tag = ((uint64_t)(uint32_t)m << 32) | (uint32_t)n | (1ull << 63);
    if (c->raw_own.tag != pad_tag) {
        HIPCHK(c, hipMemsetAsync(c->raw_own.p, 0, (size_t)m * c->ld, c->stream));
        c->raw_own.tag = pad_tag;
    }
    return MSA_OK;
}
static int upload_packed(msa_ctx *c, const uint8_t *rowmajor, int32_t m, int32_t n, int64_t ld, uint8_t indet, bool wait) {
    if (!c || (!rowmajor && m > 0 && n > 0) || ld < n) return MSA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->upload_pending) {  // an asynchronous upload nobody waited for: its caller's rows are released by this call
        HIPCHK(c, hipEventSynchronize(c->ev_upload));
        c->upload_pending = false;
    }
    int rc = set_shape(c, m, n, indet);
    if (rc) return rc;
    c->ld = round_up(std::max(n, 1), 64);
    c->paths[0] = MSA_PATH_UPLOAD_NONE;
    if (zero_copy_rows(c, m, n)) {
        c->paths[0] = MSA_PATH_UPLOAD_IN_PLACE;
        return stage_rows_zero_copy(c, m, n, [&](int i) { return rowmajor + (size_t)i * ld; });
    }
    HIPCHK(c, c->raw_own.reserve((size_t)std::max(m, 1) * c->ld + 256));
    c->raw = c->raw_own.p;
    if (m > 0 && n > 0) {
        bool locked = false;
        if (c->tuning.upload_direct) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, rowmajor) == hipSuccess) locked = at.type == hipMemoryTypeHost;
            else (void)hipGetLastError();  // (ordinary memory: "invalid value", not an error of ours)
        }
        const bool aligned16 = ld % 16 == 0 && reinterpret_cast<uintptr_t>(rowmajor) % 16 == 0;
        if (ld == c->ld) {  // already pitched: one linear copy
            c->paths[0] = MSA_PATH_UPLOAD_LINEAR;
            c->raw_own.tag = 0;
            HIPCHK(c, hipMemcpyAsync(c->raw_own.p, rowmajor, (size_t)m * c->ld, hipMemcpyHostToDevice, c->stream));
        } else if ((locked && ld % 8 == 0) || (aligned16 && c->tuning.upload_direct)) {
            // (page-locked rows of an odd size degenerate as well: 209 x 1227 took 1.5 ms; 5000-byte rows are fine)
            // The copy writes n bytes per row; the padding columns n .. ld must read as zero (what the staged path
            // writes).  They are zeroed when the buffer is new or was last used for another shape, and stay zero under
            // pitched copies of the same shape.
            if ((rc = zero_padding_for_shape(c, m, n))) return rc;
            c->paths[0] = MSA_PATH_UPLOAD_PITCHED;
            HIPCHK(c, hipMemcpy2DAsync(c->raw_own.p, (size_t)c->ld, rowmajor, (size_t)ld, (size_t)n, (size_t)m, hipMemcpyHostToDevice,
                                       c->stream));
        } else {
            c->paths[0] = MSA_PATH_UPLOAD_PACKED;
            c->raw_own.tag = 0;
            rc = upload_rows_pitched(c, m, n, [&](int i) { return rowmajor + (size_t)i * ld; });
            if (rc) return rc;
        }
    }
    if (wait) {
        SYNC(c);  // the caller may free `rowmajor` on return
    } else if (m > 0 && n > 0) {
        // the caller keeps the rows until the next call that returns results: msa_trim waits for this event before it
        // returns, whatever the trim itself waited for (a trim that removes nothing may not read the rows at all)
        if (!c->ev_upload) HIPCHK(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_upload, c->stream));
        c->upload_pending = true;
    }
    return MSA_OK;
}

int msa_host_register(const void *rows, size_t bytes) {
    if (!rows || bytes == 0) return MSA_E_INVALID;
    return hipHostRegister(const_cast<void *>(rows), bytes, hipHostRegisterDefault) == hipSuccess ? MSA_OK : MSA_E_HIP;
}

int msa_host_unregister(const void *rows) {
    if (!rows) return MSA_E_INVALID;
    return hipHostUnregister(const_cast<void *>(rows)) == hipSuccess ? MSA_OK : MSA_E_HIP;
}

int msa_upload_packed(msa_ctx *c, const uint8_t *rowmajor, int32_t m, int32_t n, int64_t ld, uint8_t indet) {
    return upload_packed(c, rowmajor, m, n, ld, indet, true);
}

int msa_upload_packed_async(msa_ctx *c, const uint8_t *rowmajor, int32_t m, int32_t n, int64_t ld, uint8_t indet) {
    return upload_packed(c, rowmajor, m, n, ld, indet, false);
}

int msa_upload_rows(msa_ctx *c, const uint8_t *const *rows, int32_t m, int32_t n, uint8_t indet) {
    if (!c || (!rows && m > 0) || m < 0 || n < 0) return MSA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = set_shape(c, m, n, indet);
    if (rc) return rc;
    c->ld = round_up(std::max(n, 1), 64);
    c->paths[0] = MSA_PATH_UPLOAD_NONE;
    if (zero_copy_rows(c, m, n)) {
        c->paths[0] = MSA_PATH_UPLOAD_IN_PLACE;
        return stage_rows_zero_copy(c, m, n, [&](int i) { return rows[i]; });
    }
    HIPCHK(c, c->raw_own.reserve((size_t)std::max(m, 1) * c->ld + 256));
    c->raw = c->raw_own.p;
    if (m > 0 && n > 0) {  // (straight from the row pointers into the pinned pieces: no packed copy in between)
        c->paths[0] = MSA_PATH_UPLOAD_PACKED;
        c->raw_own.tag = 0;
        rc = upload_rows_pitched(c, m, n, [&](int i) { return rows[i]; });
        if (rc) return rc;
    }
    SYNC(c);  // the caller may free the rows on return
    return MSA_OK;
}

int msa_attach_device(msa_ctx *c, const void *rowmajor_dev, int32_t m, int32_t n, int64_t ld, uint8_t indet) {
    if (!c || !rowmajor_dev || ld < n) return MSA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = set_shape(c, m, n, indet);
    if (rc) return rc;
    const bool usable = (ld % 64 == 0) && (ld >= round_up(std::max(n, 1), 64)) &&
                        (reinterpret_cast<uintptr_t>(rowmajor_dev) % 16 == 0);
    c->paths[0] = MSA_PATH_UPLOAD_ATTACHED;
    if (usable) {
        c->raw = static_cast<const uint8_t *>(rowmajor_dev);
        c->ld = ld;
    } else {  // re-pitch into an owned buffer (device-to-device)
        c->ld = round_up(std::max(n, 1), 64);
        HIPCHK(c, c->raw_own.reserve((size_t)std::max(m, 1) * c->ld + 256));
        if ((rc = zero_padding_for_shape(c, m, n))) return rc;
        HIPCHK(c, hipMemcpy2DAsync(c->raw_own.p, (size_t)c->ld, rowmajor_dev, (size_t)ld, (size_t)n, (size_t)m,
                                   hipMemcpyDeviceToDevice, c->stream));
        c->raw = c->raw_own.p;
    }
    return MSA_OK;
}

int msa_gaps(msa_ctx *c, int32_t *gaps_out, int32_t *indet_out) {
    if (!c || !c->raw || c->m <= 0 || c->n <= 0) return MSA_E_INVALID;  // empty alignments never reach the device
    HIPCHK(c, hipSetDevice(c->device));
    TuneScope tune(c);
    PathScope path(c);
    int rc = ensure_gaps(c, true);
    if (rc) return rc;
    if (gaps_out) std::copy(c->h_gaps.begin(), c->h_gaps.end(), gaps_out);
    if (indet_out) std::copy(c->h_indets.begin(), c->h_indets.end(), indet_out);
    return MSA_OK;
}

int msa_gaps_cached(msa_ctx *c, int32_t half_window, int32_t *out) {
    if (!c || !out || half_window < 0) return MSA_E_INVALID;
    if (c->n <= 0 || (int)c->h_gaps.size() != c->n) return 1;  // no host copy for the current alignment
    return msah::window_i32(c->h_gaps.data(), c->n, half_window, out);
}

int msa_pair_counts(msa_ctx *c, uint32_t *hit, uint32_t *dst) {
    if (!c || !c->raw || c->m <= 0 || c->n <= 0) return MSA_E_INVALID;  // empty alignments never reach the device
    HIPCHK(c, hipSetDevice(c->device));
    TuneScope tune(c);
    PathScope path(c);
    int rc = run_pairs(c, false, false, true);
    if (rc) return rc;
    const size_t bytes = (size_t)c->m * c->m * sizeof(uint32_t);
    if (hit) HIPCHK(c, hipMemcpyAsync(hit, c->hit.p, bytes, hipMemcpyDeviceToHost, c->stream));
    if (dst) HIPCHK(c, hipMemcpyAsync(dst, c->dst.p, bytes, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    return MSA_OK;
}

int msa_identities(msa_ctx *c, float *ident, float *w) {
    if (!c || !c->raw || c->m <= 0 || c->n <= 0) return MSA_E_INVALID;  // empty alignments never reach the device
    HIPCHK(c, hipSetDevice(c->device));
    TuneScope tune(c);
    PathScope path(c);
    int rc = run_pairs(c, true, true, false);
    if (rc) return rc;
    const size_t row = (size_t)c->m * sizeof(float);
    if (ident)
        HIPCHK(c, hipMemcpy2DAsync(ident, row, c->ident.p, (size_t)c->ldw * sizeof(float), row, c->m,
                                   hipMemcpyDeviceToHost, c->stream));
    if (w)
        HIPCHK(c, hipMemcpy2DAsync(w, row, c->wmat.p, (size_t)c->ldw * sizeof(float), row, c->m, hipMemcpyDeviceToHost,
                                   c->stream));
    SYNC(c);
    if (w) {  // the device keeps W strictly upper triangular; the ABI returns the symmetric matrix
        const size_t m = c->m;
        for (size_t i = 0; i < m; ++i)
            for (size_t j = i + 1; j < m; ++j) w[j * m + i] = w[i * m + j];
    }
    return MSA_OK;
}

int msa_identity_stats(msa_ctx *c, float *avg_seq, float *max_seq) {
    if (!c || !c->raw || !avg_seq || !max_seq || c->m < 2 || c->n <= 0) return MSA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TuneScope tune(c);
    PathScope path(c);
    return identity_stats(c, avg_seq, max_seq);
}

int msa_similarity(msa_ctx *c, const int32_t *vhash, const float *dist, int32_t npos, const int32_t *gaps_windowed,
                   float *mdk_out, float *q_out, msa_err_detail *detail) {
    if (!c || !c->raw || !vhash || !dist || !mdk_out || c->m <= 0 || c->n <= 0) return MSA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TuneScope tune(c);
    PathScope path(c);
    c->order_ready = false;  // (only msa_trim builds the column order ahead, for its own call)
    return similarity(c, vhash, dist, npos, gaps_windowed, mdk_out, q_out, detail);
}

int msa_overlap(msa_ctx *c, float residue_overlap, float *spurious_out) {
    if (!c || !c->raw || !spurious_out || c->m <= 0 || c->n <= 0) return MSA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TuneScope tune(c);
    PathScope path(c);
    return overlap(c, residue_overlap, spurious_out);
}

namespace {
// MSA_TRACE=1: host-side wall-clock marks of msa_trim on stderr (diagnostics)
struct TrimTrace {
    bool on = false;
    explicit TrimTrace(bool enabled) : on(enabled) {}
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[msa_trim] %-22s +%8.1f us  (at %8.1f us)\n", what,
                     std::chrono::duration<double, std::micro>(now - last).count(),
                     std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};
}  // namespace

static int trim_impl(msa_ctx *c, const msa_trim_params *p, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info);
namespace {
int engine_needs(const msa_trim_params *p);
}

int msa_trim(msa_ctx *c, const msa_trim_params *p, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info) {
    if (!c || !c->raw || !p || !keep_res || !keep_seq) return MSA_E_INVALID;
    int rc;
    try {  // (no exception crosses the C boundary: trim_impl sizes std::vectors by m, n and m * m)
        rc = trim_impl(c, p, keep_res, keep_seq, info);
    } catch (const std::bad_alloc &) {
        rc = MSA_E_NOMEM;
    } catch (...) {
        rc = MSA_E_INVALID;
    }
    if (rc != MSA_OK) {
        // an error return is a return with nothing in flight as well (an upload enqueued by msa_upload_packed_async included)
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamSynchronize(c->stream);
        c->upload_pending = false;
    }
    if (c->upload_pending) {  // the rows of msa_upload_packed_async are the caller's again
        (void)hipEventSynchronize(c->ev_upload);
        c->upload_pending = false;
    }
    return rc;
}

static int trim_impl(msa_ctx *c, const msa_trim_params *p, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info) {
    if (!c->prefetched) HIPCHK(c, hipSetDevice(c->device));
    TrimTrace trace(c->tuning.trace != 0);
    TuneScope tune(c);
    PathScope path(c);
    c->order_ready = false;
    c->pipe_active = false;
    c->colcnt_staged = false;
    c->only_gaps_rows.clear();
    msa_trim_info local;
    if (!info) info = &local;
    std::memset(info, 0, sizeof(*info));
    info->warn_row = -1;
    const int m = c->m, n = c->n;
    std::fill(keep_res, keep_res + n, 1);
    std::fill(keep_seq, keep_seq + m, 1);
    if (m == 0 || n == 0) return MSA_OK;
    if (p->method == MSA_METHOD_AUTOMATED2) return MSA_E_NOT_IMPLEMENTED;  // (no body in the reference tree, no surviving pin)
    int rc = MSA_OK;

    // trimAlManager::set_window_size
    int gap_hw = p->gap_window, sim_hw = p->similarity_window;
    if (p->window != -1) gap_hw = sim_hw = p->window;
    if (gap_hw == -1) gap_hw = 0;
    if (sim_hw == -1) sim_hw = 0;

    std::vector<int32_t> gaps_w;
    std::vector<float> mdk, mdk_w;
    auto need_gaps = [&]() -> int {
        if (!gaps_w.empty()) return MSA_OK;
        int r = ensure_gaps(c, true);
        if (r) return r;
        gaps_w.resize(n);
        return msah::window_i32(c->h_gaps.data(), n, gap_hw, gaps_w.data());
    };
    bool pipe_waited = false;
    struct PipeGuard {  // an error exit must not leave the pipeline's copies in flight over the staging buffers
        msa_ctx *c;
        bool &waited;
        ~PipeGuard() {
            if (c->pipe_active && !waited && !c->prefetched) (void)hipStreamSynchronize(c->stream);
            c->pipe_active = false;
        }
    } pipe_guard{c, pipe_waited};
    auto pipe_wait = [&]() -> int {  // the one wait of a pipelined trim
        if (pipe_waited) return MSA_OK;
        pipe_waited = true;
        if (c->prefetched) return c->h_flags.p[ST_ERRFLAG] ? MSA_E_NON_ASCII : MSA_OK;  // (the engine has waited)
        return sync_stream(c);
    };
    auto need_sim = [&]() -> int {
        if (!mdk.empty()) return MSA_OK;
        int r = need_gaps();
        if (r) return r;
        if (sim_hw > n / 4) return MSA_E_WINDOW_TOO_BIG;
        if (!p->vhash || !p->dist) return MSA_E_INVALID;
        mdk.resize(n);
        if (c->pipe_active && !(c->pipe_gated && c->h_flags.p && pipe_waited && c->h_flags.p[ST_GATE])) {
            // enqueued by sim_pipeline_begin: wait (once) and take the values
            if ((r = pipe_wait())) return r;
            r = fetch_similarity_finish(c, n, mdk.data(), nullptr, &info->err);
        } else if (c->prefetched) {
            return MSA_E_FALLBACK;
        } else {
            r = similarity(c, p->vhash, p->dist, p->npos, gap_hw > 0 ? gaps_w.data() : nullptr, mdk.data(), nullptr,
                           &info->err);
        }
        if (r) return r;
        mdk_w.resize(n);
        return msah::window_f32(mdk.data(), n, sim_hw, mdk_w.data());
    };

    int method = p->method;
    // Which trims enqueue the similarity pass up front (sim_pipeline_begin): the methods that always need its values,
    // and automated1, which may (gated on the device by the identity statistics).
    const bool column_mode = method != MSA_METHOD_NODUPLICATESEQS && p->clusters == -1 && p->max_identity == -1 &&
                             !(p->residue_overlap != -1 && p->sequence_overlap != -1);
    const bool sim_always = method == MSA_METHOD_STRICT || method == MSA_METHOD_STRICTPLUS ||
                            (method == MSA_METHOD_NONE && p->similarity_threshold != -1);
    const bool pipelined = column_mode && (sim_always || method == MSA_METHOD_AUTOMATED1) && sim_pipeline_applies(c, p, sim_hw);
    if (c->prefetched) {
        // (the engine only takes these: the similarity pipeline, or a trim that needs the gap statistics alone; no windows)
        if (gap_hw != 0 || (int)c->h_gaps.size() != n || !(pipelined || engine_needs(p) == 1)) return MSA_E_FALLBACK;
        gaps_w = c->h_gaps;
        c->pipe_active = pipelined;
        c->pipe_gated = pipelined && method == MSA_METHOD_AUTOMATED1;
    } else if (pipelined && compact_sim_applies(c, gap_hw) &&
               (rc = compact_begin(c, p->vhash, p->dist, p->npos, method == MSA_METHOD_AUTOMATED1)) != MSA_E_FALLBACK) {
        if (rc) {
            (void)hipStreamSynchronize(c->stream);
            return rc;
        }
        gaps_w = c->h_gaps;  // (no window)
        pipe_waited = true;  // (compact_begin waits itself: there is nothing for the host to do in between)
        c->paths[1] = MSA_PATH_PIPE_COMPACT;
        trace.mark("compact pipeline");
    } else if (pipelined) {
        rc = sim_pipeline_begin(c, p, gap_hw, method == MSA_METHOD_AUTOMATED1, gaps_w);
        if (rc) {  // (nothing of a half-built pipeline may stay in flight over the staging buffers)
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);
            (void)hipStreamSynchronize(c->stream);
            return rc;
        }
        trace.mark("pipeline enqueued");
    } else {
        // residues per sequence: fetched by whatever synchronisation comes first, used by remove_all_gaps -- a small alignment
        // gets them together with its gap counts in one launch and one copy
        if (compact_gaps_applies(c)) rc = compact_gaps(c), c->paths[1] = MSA_PATH_PIPE_COMPACT_GAPS;
        else rc = stage_row_totals(c);
        if (rc) return rc;
    }
    bool seq_mode = false, have_gap_cut = false;
    if (method == MSA_METHOD_NODUPLICATESEQS) {
        rc = remove_duplicates(c, keep_seq);
        if (rc) return rc;
        seq_mode = true;
    } else if (p->clusters != -1 || p->max_identity != -1) {
        // RepresentativeTrimmer: clustered on the device, only the mask comes back (identity_threshold: one run of
        // the clustering kernels; clusters=K: the threshold search probes them).  MSA_E_FALLBACK = "not applicable
        // here" (bit sets larger than the LDS, or a small m where the m*m copy is cheaper than a synchronisation
        // per probe); every other code is a real failure and propagates.
        rc = p->clusters == -1 ? device_representatives(c, p->max_identity, keep_seq)
                               : device_cluster_count(c, p->clusters, keep_seq);
        if (rc != MSA_OK && rc != MSA_E_FALLBACK) return rc;
        seq_mode = true;
    }
    if (seq_mode && rc == MSA_E_FALLBACK) {
        // the m*m identities come to the host for the greedy clustering
        std::vector<float> ident;
        rc = fetch_ident(c, ident);
        if (rc) return rc;
        std::vector<int32_t> lengths;
        rc = ungapped_lengths(c, lengths);
        if (rc) return rc;
        float thr = p->max_identity;
        if (p->clusters != -1) thr = msah::cutpoint_clusters(ident.data(), m, lengths.data(), m, p->clusters);
        msah::representatives(ident.data(), m, lengths.data(), m, thr, keep_seq);
    } else if (!seq_mode && p->residue_overlap != -1 && p->sequence_overlap != -1) {
        std::vector<float> ov(m);
        rc = overlap(c, p->residue_overlap, ov.data());
        if (rc) return rc;
        const float min_ov = p->sequence_overlap / 100.0F;
        for (int i = 0; i < m; ++i)
            if (ov[i] < min_ov) keep_seq[i] = 0;
        seq_mode = true;
    }

    if (!seq_mode) {
        if (method == MSA_METHOD_AUTOMATED1 && pipelined) {
            // everything is in the queue; the gap cut while it runs, then the one wait, then Cleaner::selectMethod
            info->gap_cut = msah::GapHistogram(c->h_gaps.data(), m, n).cut_point_2nd_slope();
            have_gap_cut = true;
            trace.mark("gap cut");
            if ((rc = pipe_wait())) return rc;
            trace.mark("waited");
            std::memcpy(&info->avg_seq, c->h_flags.p + ST_STATS, sizeof(float));
            std::memcpy(&info->max_seq, c->h_flags.p + ST_STATS + 1, sizeof(float));
            info->selected_method = msah::select_method(info->avg_seq, info->max_seq, m);
            method = info->selected_method == 1 ? MSA_METHOD_GAPPYOUT : MSA_METHOD_STRICT;
        } else if (method == MSA_METHOD_AUTOMATED1) {
            // The gap counts first (both methods need them), then one pair pass that produces both float matrices
            // (strict is likely to follow).  While the pair pass runs, the host does everything that depends on the
            // gap counts alone: their window, the gap cut, the similarity kernel's column order.
            rc = stage_gaps(c);
            if (rc) return rc;
            rc = run_pairs(c, true, true, false);
            if (rc) return rc;
            trace.mark("pairs enqueued");
            if ((rc = need_gaps())) return rc;  // (waits for the staged copy only)
            info->gap_cut = msah::GapHistogram(c->h_gaps.data(), m, n).cut_point_2nd_slope();
            have_gap_cut = true;
            if (p->vhash && p->dist) {
                rc = build_sim_order(c, gap_hw > 0 ? gaps_w.data() : nullptr, &c->order);
                if (rc) return rc;
                c->order_ready = true;
            }
            trace.mark("gap cut + column order");
            rc = identity_stats(c, &info->avg_seq, &info->max_seq);
            if (rc) return rc;
            trace.mark("identity stats");
            info->selected_method = msah::select_method(info->avg_seq, info->max_seq, m);
            method = info->selected_method == 1 ? MSA_METHOD_GAPPYOUT : MSA_METHOD_STRICT;
        }
        if (method == MSA_METHOD_GAPPYOUT) {
            if ((rc = need_gaps())) return rc;
            if (!have_gap_cut) info->gap_cut = msah::GapHistogram(c->h_gaps.data(), m, n).cut_point_2nd_slope();
            msah::clean_gaps(gaps_w.data(), n, info->gap_cut, 0, keep_res);
        } else if (method == MSA_METHOD_STRICT || method == MSA_METHOD_STRICTPLUS) {
            if ((rc = need_gaps())) return rc;
            trace.mark("gaps");
            if (!have_gap_cut) info->gap_cut = msah::GapHistogram(c->h_gaps.data(), m, n).cut_point_2nd_slope();
            trace.mark("gap cut");
            if ((rc = need_sim())) return rc;
            trace.mark("similarity");
            info->sim_cut = msah::comb_similarity_cut(gaps_w.data(), mdk_w.data(), n, info->gap_cut);
            msah::clean_strict(gaps_w.data(), mdk_w.data(), n, info->gap_cut, info->sim_cut,
                               method == MSA_METHOD_STRICTPLUS, keep_res);
            trace.mark("strict selection");
        } else if (method == MSA_METHOD_NOGAPS) {
            if ((rc = need_gaps())) return rc;
            msah::clean_gaps(gaps_w.data(), n, msah::GapHistogram(c->h_gaps.data(), m, n).cut_point(0, 0), 0, keep_res);
        } else if (method == MSA_METHOD_NOALLGAPS) {
            if ((rc = need_gaps())) return rc;
            msah::clean_gaps(gaps_w.data(), n, m - 1, 0, keep_res);
        } else if (method == MSA_METHOD_NONE) {
            float gap_thr = p->gap_threshold;
            if (p->gap_absolute_threshold != -1) gap_thr = static_cast<float>(p->gap_absolute_threshold) / m;
            const bool has_g = gap_thr != -1, has_s = p->similarity_threshold != -1;
            const float base = p->conservation_percentage;
            if (has_g && has_s) {
                if ((rc = need_sim())) return rc;
                const double cg = msah::GapHistogram(c->h_gaps.data(), m, n).cut_point(base, gap_thr);
                const float cs = static_cast<float>(msah::similarity_cut_point(mdk_w.data(), n, base, p->similarity_threshold));
                msah::clean_both(gaps_w.data(), mdk_w.data(), n, cg, cs, base, keep_res);
            } else if (has_g) {
                if ((rc = need_gaps())) return rc;
                msah::clean_gaps(gaps_w.data(), n, msah::GapHistogram(c->h_gaps.data(), m, n).cut_point(base, gap_thr), base,
                                 keep_res);
            } else if (has_s) {
                if ((rc = need_sim())) return rc;
                const float cs = static_cast<float>(msah::similarity_cut_point(mdk_w.data(), n, base, p->similarity_threshold));
                msah::clean_similarity(mdk_w.data(), n, cs, base, keep_res);
            }
        } else {
            return MSA_E_INVALID;
        }
    }
    rc = remove_all_gaps(c, keep_res, keep_seq, info);
    if (rc) return rc;
    trace.mark("remove all-gap");
    info->kept_residues = static_cast<int32_t>(std::count(keep_res, keep_res + n, 1));
    info->kept_sequences = static_cast<int32_t>(std::count(keep_seq, keep_seq + m, 1));
    if (info->kept_residues == 0) info->warnings |= MSA_W_NO_COLUMNS_LEFT;
    if (c->pairflag_state) {
        if (c->pairflag_state == 1) SYNC(c);  // no wait since the pair pass fetched the flag words: fetch them now
        c->pairflag_state = 0;
        if (c->h_flags.p && c->h_flags.p[ST_PAIRFLAG]) info->warnings |= MSA_W_UNDEFINED_IDENTITY;
    }
    return MSA_OK;
}

int msa_trim_only_gaps_rows(msa_ctx *c, int32_t *rows, int32_t cap) {
    if (!c || cap < 0 || (!rows && cap > 0)) return MSA_E_INVALID;
    const int count = (int)c->only_gaps_rows.size();
    std::copy_n(c->only_gaps_rows.begin(), std::min(count, (int)cap), rows);
    return count;
}

int msa_prof_get(msa_ctx *c, const char *kernel, float *ms_total, int32_t *launches) {
    if (!c || !kernel) return MSA_E_INVALID;
    static const char *names[] = {"gaps", "prep", "pairs", "idstats", "encode", "sim", "overlap", "cluster", "front"};
    bool known = false;
    for (const char *nm : names) known |= (std::strcmp(nm, kernel) == 0);
    if (!known) return MSA_E_INVALID;
    (void)hipSetDevice(c->device);
    prof_collect(c);
    auto it = c->prof.find(kernel);
    if (ms_total) *ms_total = it == c->prof.end() ? 0.0f : static_cast<float>(it->second.ms);
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.launches;
    return MSA_OK;
}

void msa_prof_reset(msa_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    prof_collect(c);
    c->prof.clear();
}

// kernel launches of the context's last similarity pass (bench.py reports it beside the pass's time: a profiler lists launches)
int msa_debug_sim_launches(msa_ctx *c) { return c ? c->sim_launches : 0; }

int msa_debug_last_paths(msa_ctx *c, int32_t out[8]) {
    if (!c || !out) return MSA_E_INVALID;
    std::copy_n(c->paths, 8, out);
    return MSA_OK;
}

void msa_prof_enable(msa_ctx *c, int enable) {
    if (c) c->prof_on = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
}

}  // extern "C"

// ---- batches of independent alignments ----------------------------------------------------------------------------
// The reference's batch idiom is a thread pool over `trimmer.trim` (README.md:136-152), possible because its `trim`
// releases the interpreter lock for the whole computation (_trimal.pyx:1334-1359).  Here the pool is native: worker
// threads, each with its own context (device buffers, streams), take the alignments of a call largest first; a worker
// uploads its alignment without waiting (the caller's rows outlive the call) and trims it, so that the upload of one
// alignment, the kernels of others and the host selection logic of yet others overlap on one GPU, with nothing of the
// interpreter in between.
struct msa_batch {
    int device = 0;
    std::vector<msa_ctx *> ctxs;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    uint64_t generation = 0;
    bool stop = false;
    int running = 0;
    // the call in flight
    int32_t count = 0;
    const uint8_t *const *data = nullptr;
    const int32_t *m = nullptr, *n = nullptr;
    const int64_t *ld = nullptr;
    const uint8_t *indet = nullptr;
    const msa_trim_params *params = nullptr;
    uint8_t *const *keep_res = nullptr, *const *keep_seq = nullptr;
    msa_trim_info *info = nullptr;
    int32_t *rc = nullptr;
    std::vector<int32_t> order;
    std::atomic<int32_t> next{0};
    std::vector<std::vector<int32_t>> only_gaps;  // per alignment: the rows behind MSA_W_ONLY_GAPS_SEQUENCES
    struct Engine *engine = nullptr;              // the batched-kernel path (below), created on first use
    // the engine's host-side loops (packing rows for the upload, the selection step), shared with the workers that have
    // nothing else to do: engine_parallel_for
    uint64_t sel_generation = 0;
    bool sel_open = false;   // (under `mu`) helpers may still join the job
    int sel_active = 0;      // (under `mu`) helpers inside it
    std::function<void(int32_t, int)> sel_fn;  // (item, thread: a worker's index, or workers.size() for the calling thread)
    int32_t sel_total = 0;
    std::atomic<int32_t> sel_next{0};
    std::vector<int32_t> sel_redo;
    std::vector<uint8_t> sel_finished;  // per alignment of the call: the engine has delivered its result
    bool use_engine = true;                       // MSA_BATCH_ENGINE=0: every alignment through the workers (diagnostics, tests)
    double engine_max_work = 3e8;                 // m * m * n up to which the engine takes an alignment (MSA_BATCH_ENGINE_MAX)
    int engine_min_count = 40;                    // fewer eligible alignments than this go to the workers instead (MSA_BATCH_ENGINE_MIN)
    bool in_call = false;
};

// ---- the batch engine: one launch per kernel family for a whole group of alignments -------------------------------------
// The workers above give every alignment its own ~14 launches, and with four of them in flight the device runs kernels of
// different alignments against each other (profiles/r03_c5_timeline.txt: the kernel durations add up to 3.4 x the window, the
// small layout kernels stretch 4 x).  The engine takes the alignments whose trim is the similarity pipeline of
// sim_pipeline_begin (strict, strictplus, automated1, a manual similarity threshold; no windows; fewer than ~4100 sequences,
// where the pair pass has one regime) in groups: ONE device arena per group laid out alignment after alignment, ONE table of
// descriptors, ONE launch per kernel family with blockIdx -> (alignment, block) through prefix sums, the similarity grid over
// every column of every alignment, ONE copy of every result vector back, and nothing in between that needs the host.  Two
// groups are in flight: the host takes the selection decisions of group g (trim_impl on a host-only view per alignment)
// while the device works on group g + 1.  Alignments the engine does not take, and the rare alignment whose selection needs
// another pass over the rows, go through the workers / an ordinary context as before.
struct Engine {
    struct Item {
        int32_t k;            // index in the call
        size_t res_word;      // offset (words) of its block in the result region: flags[16] gaps[npad] indets[npad] rowtot[mpad] mdk[n] q[n]
        int npad, mpad;
    };
    struct Lane {
        // two queues per group: uploads and the short VALU-bound kernels (counts, planes, pair pass, lists) at HIGH priority,
        // the similarity kernel behind them at normal priority -- the next group's preparation then gets its workgroups
        // dispatched while this group's similarity grid (tens of thousands of waves, bound by the vector-memory pipeline)
        // is still draining; with one priority the queues take turns and nothing overlaps
        hipStream_t pre = nullptr, stream = nullptr;
        hipEvent_t prepared = nullptr, done = nullptr;
        DevBuf<uint8_t> arena, meta;
        PinBuf<uint8_t> h_meta, h_res, h_stage;
        uint64_t sig = 0;
        std::vector<Item> items;
        size_t res_words = 0;
        bool busy = false;
    };
    static constexpr int MAX_LANES = 4;
    Lane lanes[MAX_LANES];
    int nlanes = 2;             // groups in flight (MSA_BATCH_LANES)
    bool trace = false;         // MSA_BATCH_TRACE=1: host-side timing of every group on stderr
    long fetch_max_bytes = 1 << 20;  // page-locked alignments up to this size are fetched by a kernel instead of a copy each (MSA_BATCH_FETCH_KB)
    int cols_max_m = 128;       // groups whose alignments have at most this many sequences: a lane per column (MSA_BATCH_COLS_MAX <= 128)
    msa_ctx *tables = nullptr;  // owns the similarity tables (and trims the alignments that fall back)
    std::vector<msa_ctx *> views;  // the host-only views handed to trim_impl: one per worker, the last one the calling thread's
};

namespace {

inline size_t align_up(size_t x, size_t q) { return (x + q - 1) / q * q; }

// Can the engine take alignment k?  (the similarity pipeline's conditions, one pair-pass regime, 32-bit list offsets, rows the
// copy engine takes in one piece or that are small enough to pack on the way)
// what a column-mode trim needs of the device: 0 not a column-mode trim the engine knows, 1 the gap statistics alone (gappyout,
// nogaps, noallgaps, a manual gap threshold), 2 the similarity pipeline as well
int engine_needs(const msa_trim_params *p) {
    const int method = p->method;
    const bool column_mode = method != MSA_METHOD_NODUPLICATESEQS && p->clusters == -1 && p->max_identity == -1 &&
                             !(p->residue_overlap != -1 && p->sequence_overlap != -1);
    if (!column_mode) return 0;
    if (method == MSA_METHOD_STRICT || method == MSA_METHOD_STRICTPLUS || method == MSA_METHOD_AUTOMATED1 ||
        (method == MSA_METHOD_NONE && p->similarity_threshold != -1))
        return 2;
    if (method == MSA_METHOD_GAPPYOUT || method == MSA_METHOD_NOGAPS || method == MSA_METHOD_NOALLGAPS ||
        (method == MSA_METHOD_NONE && (p->gap_threshold != -1 || p->gap_absolute_threshold != -1)))
        return 1;
    return 0;
}

bool engine_takes(const msa_batch *b, int32_t k, const msa_trim_params *ref) {
    const msa_trim_params *p = b->params + k;
    const int m = b->m[k], n = b->n[k];
    if (m < 2 || n < 1 || m > 32768 || !b->data[k] || b->ld[k] < n) return false;
    const int needs = engine_needs(p);
    if (!needs) return false;
    int gap_hw = p->gap_window, sim_hw = p->similarity_window;
    if (p->window != -1) gap_hw = sim_hw = p->window;
    if (gap_hw > 0) return false;
    if (needs == 1) return (double)m * n <= 4e6;  // (the gap statistics alone: small alignments, where launches are the cost)
    if (sim_hw > n / 4) return false;
    if (!p->vhash || !p->dist || p->npos < 1 || p->npos > 28) return false;
    // one set of tables per call: the first taken alignment's
    if (ref && (ref->npos != p->npos || b->indet[k] != b->indet[ref - b->params] ||
                (ref->vhash != p->vhash && std::memcmp(ref->vhash, p->vhash, 26 * sizeof(int32_t)) != 0) ||
                (ref->dist != p->dist && std::memcmp(ref->dist, p->dist, sizeof(float) * p->npos * p->npos) != 0)))
        return false;
    const int m_pad = round_up(m, 128);
    if (!msak::pair_pipe_regime(m, m_pad)) return false;
    // Where the batched kernels pay: alignments that do not fill the chip by themselves.  From ~600 x 2500 on a context per
    // alignment (four workers) is as fast or faster -- the similarity kernel bounds both (64 x 1000 x 4000: 23.8 ms of it in
    // either scheme), and four alignments in flight overlap the VALU-bound pair pass of one with the similarity kernel of
    // another, which one launch per family cannot (measured: 26.8 ms against 25.4; 96 x 700 x 3000: 19.7 against 17.5; 1024 x 100 x 1000:
    // 10.2 against 30 through trim_batch).  Since a worker's trim of a small alignment is the compact pipeline the line lies lower:
    // 128 x 500 x 2000 12.8 against 11.7 for the workers, 256 x 300 x 1200 11.0 against 15.2 for the engine.
    // MSA_BATCH_ENGINE_MAX: the m * m * n up to which the engine takes an alignment.
    if ((double)m * m * n > b->engine_max_work) return false;
    if ((double)m * m * 12 + (double)msak::bx_cols_pad(n) * msak::bx_ldk(m) * 7 > 6e9) return false;  // (a few GB per alignment: one at a time)
    return true;
}

int engine_parallel_for(msa_batch *b, int32_t total, std::function<void(int32_t, int)> fn);

struct EngineLayout {  // byte offsets of one alignment's arrays in the arena
    size_t raw, planes, ident, w, wlow, wbar, row_avg, row_max, codeT, codeR, off, trow, nvalid, simnum, simden, simstate, end;
};

int engine_enqueue(msa_batch *b, Engine *e, Engine::Lane &L, const std::vector<int32_t> &ks) {
    msa_ctx *tc = e->tables;
    const int K = (int)ks.size();
    if (!L.stream) {
        int least = 0, greatest = 0;
        HIPCHK(tc, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(tc, hipStreamCreateWithPriority(&L.pre, hipStreamNonBlocking, greatest));
        HIPCHK(tc, hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, least));
        HIPCHK(tc, hipEventCreateWithFlags(&L.prepared, hipEventDisableTiming));
        HIPCHK(tc, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    }
    // layout: the result region first (one memset, one copy), then the alignments' arrays
    std::vector<EngineLayout> lay(K);
    L.items.resize(K);
    size_t res_words = 0, stage_bytes = 0;
    int max_m = 0, any_sim = -1;  // (over the alignments that run the similarity pipeline)
    uint64_t sig = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
    for (int i = 0; i < K; ++i) {
        const int k = ks[i], m = b->m[k], n = b->n[k];
        Engine::Item &it = L.items[i];
        it.k = k;
        it.npad = round_up(n + 64, 64);
        it.mpad = round_up(m + 64, 64);
        it.res_word = res_words;
        res_words += 16 + (size_t)2 * it.npad + it.mpad + (size_t)2 * it.npad;
        if (engine_needs(b->params + k) == 2) max_m = std::max(max_m, m), any_sim = i;
        mix(((uint64_t)(uint32_t)m << 32) | (uint32_t)n);
        mix(b->params[k].method == MSA_METHOD_AUTOMATED1);
        mix(engine_needs(b->params + k));
    }
    const bool multi = msak::lg_rounds_per_launch(max_m) > 0;  // (the similarity kernel in several launches: per-column state)
    // groups of small alignments: the similarity statistic with a lane per column (similarity_cols_batch_kernel) -- no
    // column-major codes, no lists
    const bool cols_mode = max_m <= e->cols_max_m;
    mix(cols_mode);
    size_t off = align_up(res_words * 4, 4096);
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    // the rows of every alignment side by side, then the derived arrays.  Rows the copy engine takes as they lie (page-locked,
    // or 16-byte aligned rows of a multiple of 16 bytes) go up in a copy each -- 8 us of the copy queue per small alignment,
    // beside the kernels of the group before; the others are packed into pinned staging at the same offsets (the calling
    // thread and the idle workers) and go up in one copy per run.  (Packing everything small, measured: 1.7 ms of five
    // threads per 25 MB of cache-cold rows against 2.1 ms of copy queue that nobody waits for.)
    const size_t raw_base = off;
    std::vector<uint8_t> packed(K, 0);
    std::vector<const uint8_t *> fetch(K, nullptr);  // page-locked rows as the device sees them: read by fetch_rows_batch_kernel
    for (int i = 0; i < K; ++i) {
        const int k = ks[i], m = b->m[k], n = b->n[k];
        const size_t ld = round_up(n, 64);
        lay[i].raw = take((size_t)m * ld + 256);
        const uint8_t *rows = b->data[k];
        const int64_t hld = b->ld[k];
        bool locked = false;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, rows) == hipSuccess) locked = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
        const bool direct = hld == (int64_t)ld || (locked && hld % 8 == 0) || (hld % 16 == 0 && reinterpret_cast<uintptr_t>(rows) % 16 == 0);
        packed[i] = !direct;
        // small page-locked alignments: the device fetches the rows itself (one launch per group); big ones keep their copy
        // (one DMA transfer at the link's rate needs no help)
        if (locked && e->fetch_max_bytes > 0 && (size_t)m * ld <= (size_t)e->fetch_max_bytes) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, const_cast<uint8_t *>(rows), 0) == hipSuccess && dp) {
                fetch[i] = static_cast<const uint8_t *>(dp);
                packed[i] = 0;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    const size_t raw_bytes = off - raw_base;
    for (int i = 0; i < K; ++i) {
        const int k = ks[i], m = b->m[k], n = b->n[k];
        const size_t ld = round_up(n, 64), nchunk = (n + 31) / 32, m_pad = round_up(m, 128), ldw = round_up(m, 64);
        const size_t ncp = msak::bx_cols_pad(n), ldk = msak::bx_ldk(m);
        EngineLayout &y = lay[i];
        if (engine_needs(b->params + k) < 2) {  // the gap statistics alone: the rows are all it needs on the device
            y.planes = y.ident = y.w = y.wlow = y.wbar = y.row_avg = y.row_max = y.codeT = y.codeR = y.off = y.trow = y.nvalid = y.simnum =
                y.simden = y.simstate = 0;
            y.end = off;
            continue;
        }
        y.planes = take(((size_t)msak::planes_total() * nchunk * m_pad + 64) * 4);
        y.ident = take(((size_t)m * ldw + 512) * 4);
        y.w = take(((size_t)m * ldw + 512) * 4);
        y.wlow = take((msak::bx_wlow_rows(m) + 2) * ldw * 4);
        y.wbar = take(((size_t)m + 128) * 4);
        y.row_avg = take(((size_t)m + 64) * 4);
        y.row_max = take(((size_t)m + 64) * 4);
        y.codeT = y.codeR = y.off = y.trow = y.nvalid = 0;
        if (cols_mode) {
            y.codeR = take((size_t)m * ld + 256);
        } else {
            y.codeT = take(ncp * ldk + 64);
            y.off = take((ncp * ldk + 64) * 4);
            y.trow = take((ncp * ldk + 64) * 2);
            y.nvalid = take((ncp + 64) * 4);
        }
        y.simnum = take(((size_t)n + 64) * 4);
        y.simden = take(((size_t)n + 64) * 4);
        y.simstate = multi ? take(msak::lg_state_floats(n) * 4) : 0;
        y.end = off;
    }
    if (std::find(packed.begin(), packed.end(), 1) != packed.end()) stage_bytes = raw_bytes;  // (the staging mirrors the raw region)
    mix(multi);
    const size_t arena_bytes = off;
    const uint8_t *old_base = L.arena.p;
    HIPCHK(tc, L.arena.reserve(arena_bytes));
    mix((uint64_t)(uintptr_t)L.arena.p);
    // tables: [BAlign K][LgAlign K][prefix arrays: F x (K + 1)]
    enum { F_FETCH, F_GAPS, F_ROWTOT, F_PLANES, F_PAIRS, F_WMEANS, F_IDROWS, F_ENCODE, F_COMPACT, F_FINISH, F_COLS, F_COUNT };  // (F_ENCODE / F_COLS: by mode)
    const size_t meta_bytes = align_up((size_t)K * sizeof(msak::BAlign), 256) + align_up((size_t)K * sizeof(msak::LgAlign), 256) +
                              align_up((size_t)F_COUNT * (K + 1) * sizeof(int32_t), 256);
    HIPCHK(tc, L.meta.reserve(meta_bytes));
    HIPCHK(tc, L.h_meta.reserve(meta_bytes));
    HIPCHK(tc, L.h_res.reserve(res_words * 4 + 64));
    if (stage_bytes) HIPCHK(tc, L.h_stage.reserve(stage_bytes));
    (void)old_base;
    uint8_t *A = L.arena.p;
    msak::BAlign *bt = reinterpret_cast<msak::BAlign *>(L.h_meta.p);
    msak::LgAlign *lt = reinterpret_cast<msak::LgAlign *>(L.h_meta.p + align_up((size_t)K * sizeof(msak::BAlign), 256));
    int32_t *pf = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(lt) + align_up((size_t)K * sizeof(msak::LgAlign), 256));
    uint8_t *meta_d = L.meta.p;
    const msak::BAlign *bt_d = reinterpret_cast<const msak::BAlign *>(meta_d);
    const msak::LgAlign *lt_d = reinterpret_cast<const msak::LgAlign *>(meta_d + align_up((size_t)K * sizeof(msak::BAlign), 256));
    const int32_t *pf_d = reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(lt_d) + align_up((size_t)K * sizeof(msak::LgAlign), 256));
    for (int f = 0; f < F_COUNT; ++f) pf[(size_t)f * (K + 1)] = 0;
    int32_t *res_d = reinterpret_cast<int32_t *>(A);
    for (int i = 0; i < K; ++i) {
        const int k = ks[i], m = b->m[k], n = b->n[k];
        const Engine::Item &it = L.items[i];
        const EngineLayout &y = lay[i];
        msak::BAlign d = {};
        d.raw = A + y.raw;
        d.fetch_src = fetch[i];
        d.fetch_ld = b->ld[k];
        d.ld = round_up(n, 64);
        d.ldk = msak::bx_ldk(m);
        d.planes = reinterpret_cast<uint32_t *>(A + y.planes);
        d.flags = res_d + it.res_word;
        d.gaps = d.flags + 16;
        d.indets = d.gaps + it.npad;
        d.rowtot = d.indets + it.npad;
        d.mdk = reinterpret_cast<float *>(d.rowtot + it.mpad);
        d.gated = b->params[k].method == MSA_METHOD_AUTOMATED1;
        d.ident = d.gated ? reinterpret_cast<float *>(A + y.ident) : nullptr;
        d.w = reinterpret_cast<float *>(A + y.w);
        d.wlow = reinterpret_cast<float *>(A + y.wlow);
        d.wbar = reinterpret_cast<float *>(A + y.wbar);
        d.row_avg = reinterpret_cast<float *>(A + y.row_avg);
        d.row_max = reinterpret_cast<float *>(A + y.row_max);
        d.codeT = A + y.codeT;
        d.codeR = A + y.codeR;
        d.off = reinterpret_cast<uint32_t *>(A + y.off);
        d.trow = reinterpret_cast<uint16_t *>(A + y.trow);
        d.nvalid = reinterpret_cast<int32_t *>(A + y.nvalid);
        d.simnum = reinterpret_cast<float *>(A + y.simnum);
        d.simden = reinterpret_cast<float *>(A + y.simden);
        d.m = m, d.n = n, d.nchunk = (n + 31) / 32, d.m_pad = round_up(m, 128), d.ldw = round_up(m, 64);
        d.ncols_pad = msak::bx_cols_pad(n);
        d.indet4 = 0x01010101u * b->indet[k];
        bt[i] = d;
        msak::LgAlign g = {};
        g.voff = d.off, g.vtrow = d.trow, g.nvalid = d.nvalid, g.codeT = d.codeT;
        g.wlow = d.wlow, g.wup = d.w, g.wbar = d.wbar, g.num_out = d.simnum, g.den_out = d.simden;
        g.state = multi ? reinterpret_cast<float *>(A + y.simstate) : nullptr;
        g.gate = d.gated ? d.flags + ST_GATE : nullptr;
        g.cols = nullptr;
        g.ldk = d.ldk, g.m = m, g.n = n, g.ldw = d.ldw, g.ncols = n;
        lt[i] = g;
        const bool sim = engine_needs(b->params + k) == 2;
        auto add = [&](int f, int blocks) { pf[(size_t)f * (K + 1) + i + 1] = pf[(size_t)f * (K + 1) + i] + ((sim || f <= F_ROWTOT) ? blocks : 0); };
        add(F_FETCH, fetch[i] ? (int)(((int64_t)m * (d.ld / 16) + 255) / 256) : 0);
        add(F_GAPS, (int)((d.ld / 4 + 255) / 256) * ((m + 63) / 64));
        add(F_ROWTOT, (m + 3) / 4);
        add(F_PLANES, ((d.nchunk + 1) / 2) * ((d.m_pad + 255) / 256));
        add(F_PAIRS, msak::pair_tiles_pipe(m, d.m_pad));
        add(F_WMEANS, cols_mode ? 0 : (m + 64 + 3) / 4);  // (the predictor's input: the wave-per-column kernel only)
        add(F_IDROWS, d.gated ? (m + 3) / 4 : 0);
        add(F_ENCODE, cols_mode ? (int)((d.ld + 255) / 256) * ((m + 15) / 16) : (d.ncols_pad / 64) * (int)(d.ldk / 64));
        add(F_COMPACT, cols_mode ? 0 : (d.ncols_pad + 3) / 4);
        add(F_FINISH, (n + 255) / 256);
        add(F_COLS, cols_mode ? (n + 63) / 64 : n);
    }
    auto PF = [&](int f) { return pf_d + (size_t)f * (K + 1); };
    auto NB = [&](int f) { return pf[(size_t)f * (K + 1) + K]; };
    hipStream_t st = L.pre;
    // zeroes: everything when the layout differs from the one the arena was last zeroed for (padding of W, of the rows:
    // the kernels write the same entries for the same layout), else the result region alone (counts, flags)
    if (sig != L.sig) {
        HIPCHK(tc, hipMemsetAsync(A, 0, arena_bytes, st));
        L.sig = sig;
    } else {
        HIPCHK(tc, hipMemsetAsync(A, 0, res_words * 4, st));
    }
    HIPCHK(tc, hipMemcpyAsync(meta_d, L.h_meta.p, meta_bytes, hipMemcpyHostToDevice, st));
    if (stage_bytes) {
        // pack (the calling thread and the idle workers), then one copy per run of packed alignments
        uint8_t *stage = L.h_stage.p;
        engine_parallel_for(b, K, [&](int32_t i, int) {
            if (!packed[i]) return;
            const int k = ks[i], m = b->m[k], n = b->n[k];
            const size_t ld = round_up(n, 64);
            const uint8_t *rows = b->data[k];
            const int64_t hld = b->ld[k];
            uint8_t *dst = stage + (lay[i].raw - raw_base);
            for (int r = 0; r < m; ++r) {
                std::memcpy(dst + (size_t)r * ld, rows + (size_t)r * hld, (size_t)n);
                std::memset(dst + (size_t)r * ld + n, 0, ld - n);
            }
        });
    }
    for (int i = 0; i < K; ++i) {
        const int k = ks[i], m = b->m[k], n = b->n[k];
        const size_t ld = round_up(n, 64);
        if (fetch[i]) continue;
        if (packed[i]) {
            int j = i;
            while (j + 1 < K && packed[j + 1]) ++j;
            const size_t from = lay[i].raw, to = j + 1 < K ? lay[j + 1].raw : raw_base + raw_bytes;
            HIPCHK(tc, hipMemcpyAsync(A + from, L.h_stage.p + (from - raw_base), to - from, hipMemcpyHostToDevice, st));
            i = j;
            continue;
        }
        const uint8_t *rows = b->data[k];
        const int64_t hld = b->ld[k];
        uint8_t *dst = A + lay[i].raw;
        if (hld == (int64_t)ld) HIPCHK(tc, hipMemcpyAsync(dst, rows, (size_t)m * ld, hipMemcpyHostToDevice, st));
        else HIPCHK(tc, hipMemcpy2DAsync(dst, ld, rows, (size_t)hld, (size_t)n, (size_t)m, hipMemcpyHostToDevice, st));
    }
    msak::launch_fetch_rows_batch(st, bt_d, PF(F_FETCH), K, NB(F_FETCH));
    msak::launch_gap_counts_batch(st, bt_d, PF(F_GAPS), K, NB(F_GAPS));
    msak::launch_row_nongap_batch(st, bt_d, PF(F_ROWTOT), K, NB(F_ROWTOT));
    msak::launch_prep_planes_batch(st, bt_d, PF(F_PLANES), K, NB(F_PLANES));
    int min_nchunk = 1 << 30;
    for (int i = 0; i < K; ++i)
        if (engine_needs(b->params + ks[i]) == 2) min_nchunk = std::min(min_nchunk, (b->n[ks[i]] + 31) / 32);
    msak::launch_pair_counts_batch(st, bt_d, PF(F_PAIRS), K, NB(F_PAIRS), min_nchunk);
    msak::launch_w_row_means_batch(st, bt_d, PF(F_WMEANS), K, NB(F_WMEANS));
    msak::launch_identity_stats_batch(st, bt_d, PF(F_IDROWS), K, NB(F_IDROWS));
    const int npos = any_sim >= 0 ? b->params[ks[any_sim]].npos : 0;
    if (cols_mode) msak::launch_sim_encode_rm_batch(st, bt_d, PF(F_ENCODE), K, NB(F_ENCODE), tc->lut.p);
    else msak::launch_sim_lists_batch(st, bt_d, PF(F_ENCODE), NB(F_ENCODE), PF(F_COMPACT), NB(F_COMPACT), K, tc->lut.p, npos);
    HIPCHK(tc, hipEventRecord(L.prepared, st));
    st = L.stream;
    HIPCHK(tc, hipStreamWaitEvent(st, L.prepared, 0));
    if (cols_mode) {
        msak::launch_similarity_cols_batch(st, bt_d, PF(F_COLS), K, NB(F_COLS), tc->tab.p);
    } else {
        int launches = 0;
        const int er = msak::launch_similarity_lg_batch(st, lt_d, PF(F_COLS), K, NB(F_COLS), max_m, npos, tc->tab.p, multi, &launches);
        if (er) return fail_hip(tc, (hipError_t)er, "launch_similarity (batch)");
    }
    msak::launch_sim_finish_batch(st, bt_d, PF(F_FINISH), K, NB(F_FINISH));
    HIPCHK(tc, hipGetLastError());
    HIPCHK(tc, hipMemcpyAsync(L.h_res.p, A, res_words * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(tc, hipEventRecord(L.done, st));
    L.res_words = res_words;
    L.busy = true;
    return MSA_OK;
}

// The selection decisions of one alignment of a finished group, on view `v` (trim_impl reads the statistics the group's
// result copy brought back).
void engine_select_item(msa_batch *b, Engine::Lane &L, const Engine::Item &it, msa_ctx *v) {
    int32_t *res = reinterpret_cast<int32_t *>(L.h_res.p);
    const int k = it.k, m = b->m[k], n = b->n[k];
    int32_t *flags = res + it.res_word;
    v->m = m, v->n = n, v->indet = b->indet[k], v->ld = round_up(n, 64);
    v->raw = L.arena.p;  // (never read through the view)
    v->host_rows = b->data[k], v->host_ld = b->ld[k];
    v->h_flags.p = flags;
    v->h_gaps.assign(flags + 16, flags + 16 + n);
    v->h_indets.assign(flags + 16 + it.npad, flags + 16 + it.npad + n);
    v->h_rowtot.p = flags + 16 + 2 * it.npad;
    v->rowtot_staged = 2;
    v->h_f32.p = reinterpret_cast<float *>(flags + 16 + 2 * it.npad + it.mpad);
    v->pairflag_state = 2;
    v->have_gaps = true;
    msa_trim_info local;
    msa_trim_info *info = b->info ? b->info + k : &local;
    int rc;
    try {
        rc = trim_impl(v, b->params + k, b->keep_res[k], b->keep_seq[k], info);
    } catch (const std::bad_alloc &) {
        rc = MSA_E_NOMEM;
    } catch (...) {
        rc = MSA_E_INVALID;
    }
    if (rc == MSA_E_FALLBACK) {
        if (v->tuning.trace) std::fprintf(stderr, "[engine] alignment %d (%d x %d) needs the device again: ordinary context\n", k, m, n);
        std::lock_guard<std::mutex> lk(b->mu);
        b->sel_redo.push_back(k);
        return;
    }
    b->only_gaps[k] = v->only_gaps_rows;
    b->rc[k] = rc;
    b->sel_finished[k] = 1;
}

// items of the job in flight until none is left (the calling thread and every idle worker)
int engine_job_some(msa_batch *b, int thread) {
    int mine = 0;
    for (;;) {
        const int32_t i = b->sel_next.fetch_add(1, std::memory_order_relaxed);
        if (i >= b->sel_total) break;
        b->sel_fn(i, thread);
        ++mine;
    }
    return mine;
}

// fn(item, thread) for item = 0 .. total-1, by the calling thread and the workers that are idle; returns the calling thread's share
int engine_parallel_for(msa_batch *b, int32_t total, std::function<void(int32_t, int)> fn) {
    const bool share = total >= 8 && !b->workers.empty();
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->sel_fn = std::move(fn);
        b->sel_total = total;
        b->sel_next.store(0);
        if (share) {
            b->sel_open = true;
            ++b->sel_generation;
        }
    }
    if (share) b->cv_work.notify_all();
    const int mine = engine_job_some(b, (int)b->workers.size());
    if (share) {  // no helper joins from here on; wait for those that hold items
        std::unique_lock<std::mutex> lk(b->mu);
        b->sel_open = false;
        b->cv_done.wait(lk, [&] { return b->sel_active == 0; });
    }
    return mine;
}

// wait for a lane's group and take its selection decisions (with the workers that are idle: the host side of a trim is
// 10 - 20 us of cut points and masks per alignment, serial work that would otherwise leave the device waiting on batches of
// small alignments); alignments that need the device again are collected in b->sel_redo
int engine_finish(msa_batch *b, Engine *e, Engine::Lane &L) {
    if (!L.busy) return MSA_OK;
    L.busy = false;
    msa_ctx *tc = e->tables;
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(tc, hipEventSynchronize(L.done));
    const auto t1 = std::chrono::steady_clock::now();
    const int mine = engine_parallel_for(b, (int32_t)L.items.size(), [&](int32_t i, int thread) { engine_select_item(b, L, L.items[i], e->views[thread]); });
    if (e->trace)
        std::fprintf(stderr, "[engine] group of %zu: waited %.0f us, selection %.0f us (%d of them by the calling thread)\n", L.items.size(),
                     std::chrono::duration<double, std::micro>(t1 - t0).count(),
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count(), mine);
    return MSA_OK;
}

void engine_destroy(Engine *e) {
    if (!e) return;
    if (e->tables) (void)hipSetDevice(e->tables->device);
    for (Engine::Lane &L : e->lanes) {
        if (L.pre) (void)hipStreamSynchronize(L.pre);
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        L.arena.release(), L.meta.release(), L.h_meta.release(), L.h_res.release(), L.h_stage.release();
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.prepared) (void)hipEventDestroy(L.prepared);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.pre) (void)hipStreamDestroy(L.pre);
    }
    for (msa_ctx *v : e->views) {  // (their pinned pointers are the lanes': nothing of their own to release)
        v->h_flags.p = nullptr, v->h_rowtot.p = nullptr, v->h_f32.p = nullptr;
        delete v;
    }
    if (e->tables) msa_ctx_destroy(e->tables);
    delete e;
}

// the alignments of the call in flight that the engine takes (`ks`, largest first), in groups, two groups in flight
int engine_run(msa_batch *b, const std::vector<int32_t> &ks) {
    if (ks.empty()) return MSA_OK;
    if (!b->engine) {
        Engine *e = new (std::nothrow) Engine();
        if (!e) return MSA_E_NOMEM;
        int rc = msa_ctx_create(b->device, &e->tables);
        if (rc != MSA_OK) {
            delete e;
            return rc;
        }
        for (size_t w = 0; w <= b->workers.size(); ++w) {
            msa_ctx *v = new msa_ctx();
            v->device = b->device;
            v->tuning = e->tables->tuning;
            v->prefetched = true;
            e->views.push_back(v);
        }
        e->trace = std::getenv("MSA_BATCH_TRACE") != nullptr;
        if (const char *ev = std::getenv("MSA_BATCH_FETCH_KB")) e->fetch_max_bytes = (long)std::atol(ev) << 10;
        if (const char *ev = std::getenv("MSA_BATCH_COLS_MAX")) e->cols_max_m = std::min(128, std::atoi(ev));  // (the kernel's LDS tile)
        if (const char *ev = std::getenv("MSA_BATCH_LANES")) e->nlanes = std::max(1, std::min((int)Engine::MAX_LANES, std::atoi(ev)));
        b->engine = e;
    }
    Engine *e = b->engine;
    msa_ctx *tc = e->tables;
    HIPCHK(tc, hipSetDevice(b->device));
    TuneScope tune(tc);
    int rc = MSA_OK;
    for (int32_t k : ks)
        if (engine_needs(b->params + k) == 2) {  // the one set of tables of the call (engine_takes: every such alignment shares it)
            const msa_trim_params *p0 = b->params + k;
            tc->indet = b->indet[k];
            if ((rc = ensure_tables(tc, p0->vhash, p0->dist, p0->npos))) return rc;
            break;
        }
    // groups: about a quarter of the call each (at least two groups in flight whenever there are two alignments), bounded
    // by the arena (~8 GB) and by 256 alignments
    const int total = (int)ks.size();
    int parts = 4;
    if (const char *ev = std::getenv("MSA_BATCH_GROUPS")) parts = std::max(1, std::atoi(ev));  // (diagnostics)
    const int target = std::max(1, std::min(256, (total + parts - 1) / parts));
    std::vector<std::vector<int32_t>> groups;
    {
        std::vector<int32_t> cur;
        double bytes = 0;
        for (int32_t k : ks) {
            const double need = (double)b->m[k] * b->m[k] * 12 + (double)msak::bx_cols_pad(b->n[k]) * msak::bx_ldk(b->m[k]) * 7 + (double)b->m[k] * b->n[k] * 2;
            // (a group is small alignments -- a lane per column -- or not: the kernels differ)
            const bool turn = !cur.empty() && (b->m[cur.front()] <= e->cols_max_m) != (b->m[k] <= e->cols_max_m);
            if (!cur.empty() && ((int)cur.size() >= target || bytes + need > 8e9 || turn)) {
                groups.push_back(cur);
                cur.clear();
                bytes = 0;
            }
            cur.push_back(k);
            bytes += need;
        }
        if (!cur.empty()) groups.push_back(cur);
    }
    b->sel_redo.clear();
    b->sel_finished.assign((size_t)b->count, 0);
    const int G = (int)groups.size();
    int first_error = MSA_OK;
    for (int g = 0; g < G && first_error == MSA_OK; ++g) {
        Engine::Lane &L = e->lanes[g % e->nlanes];
        rc = engine_finish(b, e, L);  // (the group that used this lane `nlanes` steps ago)
        const auto te = std::chrono::steady_clock::now();
        if (rc == MSA_OK) rc = engine_enqueue(b, e, L, groups[g]);
        if (e->trace)
            std::fprintf(stderr, "[engine] group %d enqueued in %.0f us\n", g,
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - te).count());
        if (rc != MSA_OK) first_error = rc;
    }
    for (int g = 0; g < e->nlanes; ++g) {
        rc = engine_finish(b, e, e->lanes[(G + g) % e->nlanes]);
        if (rc != MSA_OK && first_error == MSA_OK) first_error = rc;
    }
    if (first_error != MSA_OK) {
        for (Engine::Lane &L : e->lanes) {
            if (L.pre) (void)hipStreamSynchronize(L.pre);
            if (L.stream) (void)hipStreamSynchronize(L.stream);
            L.busy = false;
        }
        for (int32_t k : ks)
            if (!b->sel_finished[k]) b->rc[k] = first_error;  // (the groups that were through keep their results)
        return first_error;
    }
    // the selection of these needs another pass over the rows: an ordinary context
    for (int32_t k : b->sel_redo) {
        msa_trim_info local;
        msa_trim_info *info = b->info ? b->info + k : &local;
        rc = upload_packed(tc, b->data[k], b->m[k], b->n[k], b->ld[k], b->indet[k], false);
        if (rc == MSA_OK) rc = msa_trim(tc, b->params + k, b->keep_res[k], b->keep_seq[k], info);
        else (void)hipStreamSynchronize(tc->stream);
        b->only_gaps[k] = tc->only_gaps_rows;
        b->rc[k] = rc;
    }
    return MSA_OK;
}

}  // namespace

namespace {
void batch_worker(msa_batch *b, int w) {
    (void)hipSetDevice(b->device);
    uint64_t seen = 0, seen_sel = 0;
    for (;;) {
        bool select = false;
        {
            std::unique_lock<std::mutex> lk(b->mu);
            b->cv_work.wait(lk, [&] { return b->stop || b->generation != seen || b->sel_generation != seen_sel; });
            if (b->stop) return;
            if (b->generation != seen) {
                seen = b->generation;
            } else {
                seen_sel = b->sel_generation;
                if (!b->sel_open) continue;  // (the job was over before this thread woke up)
                select = true;
                ++b->sel_active;
            }
        }
        if (select) {  // the engine's selection step: a helper beside the calling thread
            engine_job_some(b, w);
            std::lock_guard<std::mutex> lk(b->mu);
            if (--b->sel_active == 0) b->cv_done.notify_all();
            continue;
        }
        msa_ctx *c = b->ctxs[w];
        for (;;) {
            const int32_t slot = b->next.fetch_add(1, std::memory_order_relaxed);
            if (slot >= (int32_t)b->order.size()) break;
            const int32_t k = b->order[slot];
            msa_trim_info local;
            msa_trim_info *info = b->info ? b->info + k : &local;
            int rc;
            try {  // (an exception must not leave the thread: std::terminate would take the caller's process with it)
                c->only_gaps_rows.clear();
                rc = upload_packed(c, b->data[k], b->m[k], b->n[k], b->ld[k], b->indet[k], false);
                if (rc == MSA_OK) rc = msa_trim(c, b->params + k, b->keep_res[k], b->keep_seq[k], info);
                else {
                    std::memset(info, 0, sizeof(*info));
                    (void)hipStreamSynchronize(c->stream);  // (nothing of a failed upload may stay in flight over the caller's rows)
                }
                b->only_gaps[k] = c->only_gaps_rows;
            } catch (const std::bad_alloc &) {
                rc = MSA_E_NOMEM;
            } catch (...) {
                rc = MSA_E_INVALID;
            }
            if (rc == MSA_E_NOMEM || rc == MSA_E_INVALID) {
                if (c->stream2) (void)hipStreamSynchronize(c->stream2);
                (void)hipStreamSynchronize(c->stream);
                c->upload_pending = false;
            }
            b->rc[k] = rc;
        }
        {
            std::lock_guard<std::mutex> lk(b->mu);
            if (--b->running == 0) b->cv_done.notify_all();
        }
    }
}
}  // namespace

extern "C" {

int msa_batch_create(int device, int32_t workers, msa_batch **out) {
    if (!out || workers < 1 || workers > 64) return MSA_E_INVALID;
    *out = nullptr;
    msa_batch *b = new (std::nothrow) msa_batch();
    if (!b) return MSA_E_NOMEM;
    b->device = device;
    if (const char *e = std::getenv("MSA_BATCH_ENGINE")) b->use_engine = std::atoi(e) != 0;
    if (const char *e = std::getenv("MSA_BATCH_ENGINE_MAX")) b->engine_max_work = std::atof(e);
    if (const char *e = std::getenv("MSA_BATCH_ENGINE_MIN")) b->engine_min_count = std::atoi(e);
    for (int w = 0; w < workers; ++w) {
        msa_ctx *c = nullptr;
        const int rc = msa_ctx_create(device, &c);
        if (rc != MSA_OK) {
            for (msa_ctx *x : b->ctxs) msa_ctx_destroy(x);
            delete b;
            return rc;
        }
        b->ctxs.push_back(c);
    }
    for (int w = 0; w < workers; ++w) b->workers.emplace_back(batch_worker, b, w);
    *out = b;
    return MSA_OK;
}

void msa_batch_destroy(msa_batch *b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;
    }
    b->cv_work.notify_all();
    for (std::thread &t : b->workers) t.join();
    for (msa_ctx *c : b->ctxs) msa_ctx_destroy(c);
    engine_destroy(b->engine);
    delete b;
}

int32_t msa_batch_workers(const msa_batch *b) { return b ? (int32_t)b->workers.size() : 0; }

int msa_trim_batch(msa_batch *b, int32_t count, const uint8_t *const *data, const int32_t *m, const int32_t *n, const int64_t *ld,
                   const uint8_t *indet, const msa_trim_params *params, uint8_t *const *keep_res, uint8_t *const *keep_seq,
                   msa_trim_info *info, int32_t *rc) {
    if (!b || count < 0 || (count > 0 && (!data || !m || !n || !ld || !indet || !params || !keep_res || !keep_seq || !rc)))
        return MSA_E_INVALID;
    if (count == 0) return MSA_OK;
    std::vector<int32_t> engine_ks;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        if (b->running || b->in_call) return MSA_E_INVALID;  // one call at a time per batch object
        b->in_call = true;
        b->count = count;
        b->data = data, b->m = m, b->n = n, b->ld = ld, b->indet = indet, b->params = params;
        b->keep_res = keep_res, b->keep_seq = keep_seq, b->info = info, b->rc = rc;
        // largest first (cost ~ m^2 n): the last alignments to finish are the small ones.  What the engine takes (batched
        // kernels, this thread) and what the workers take (a context per alignment)
        std::vector<int32_t> all(count);
        for (int32_t k = 0; k < count; ++k) all[k] = k;
        std::stable_sort(all.begin(), all.end(), [&](int32_t x, int32_t y) {
            return (double)m[x] * m[x] * n[x] > (double)m[y] * m[y] * n[y];
        });
        b->order.clear();
        const msa_trim_params *ref = nullptr;
        for (int32_t k : all) {
            if (b->use_engine && engine_takes(b, k, ref)) {
                if (!ref && engine_needs(params + k) == 2) ref = params + k;
                engine_ks.push_back(k);
            } else {
                b->order.push_back(k);
            }
        }
        // A handful of small alignments is faster through the worker contexts (each a compact pipeline of three launches,
        // compact_begin) than as a group of the batched kernels with its arena, tables and ten launches: 8 x (100 x 1000) 0.31
        // against 0.96 ms, 16: 0.57 / 1.11, 32: 1.0 / 1.2, 64: 2.0 / 1.45 (tools/small_batch.py).
        if ((int)engine_ks.size() < b->engine_min_count) {
            b->order.insert(b->order.end(), engine_ks.begin(), engine_ks.end());
            std::stable_sort(b->order.begin(), b->order.end(), [&](int32_t x, int32_t y) {
                return (double)m[x] * m[x] * n[x] > (double)m[y] * m[y] * n[y];
            });
            engine_ks.clear();
        }
        b->only_gaps.assign(count, {});
        for (int32_t k = 0; k < count; ++k) rc[k] = MSA_OK;
        b->next.store(0);
        b->running = b->order.empty() ? 0 : (int)b->workers.size();
        if (b->running) ++b->generation;
    }
    if (!b->order.empty()) b->cv_work.notify_all();
    const auto t_call = std::chrono::steady_clock::now();
    int engine_rc = MSA_OK;
    try {
        engine_rc = engine_run(b, engine_ks);
    } catch (const std::bad_alloc &) {
        engine_rc = MSA_E_NOMEM;
    } catch (...) {
        engine_rc = MSA_E_INVALID;
    }
    if (engine_rc != MSA_OK)
        for (int32_t k : engine_ks)
            if (rc[k] == MSA_OK && !(k < (int32_t)b->sel_finished.size() && b->sel_finished[k])) rc[k] = engine_rc;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv_done.wait(lk, [&] { return b->running == 0; });
    }
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->in_call = false;
    }
    if (std::getenv("MSA_BATCH_TRACE"))
        std::fprintf(stderr, "[msa_trim_batch] %d alignments: %zu through the batched kernels, %zu through the workers, %.2f ms\n", (int)count,
                     engine_ks.size(), b->order.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    for (int32_t k = 0; k < count; ++k)
        if (rc[k] != MSA_OK) return rc[k];
    return MSA_OK;
}

int msa_batch_only_gaps_rows(msa_batch *b, int32_t k, int32_t *rows, int32_t cap) {
    if (!b || k < 0 || k >= (int32_t)b->only_gaps.size() || cap < 0 || (!rows && cap > 0)) return MSA_E_INVALID;
    const std::vector<int32_t> &v = b->only_gaps[k];
    std::copy_n(v.begin(), std::min((int)v.size(), (int)cap), rows);
    return (int)v.size();
}

const char *msa_batch_last_hip_error(const msa_batch *b, int32_t worker) {
    return (b && worker >= 0 && worker < (int32_t)b->ctxs.size()) ? b->ctxs[worker]->hip_err : "";
}

}  // extern "C"
