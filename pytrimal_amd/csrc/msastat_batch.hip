// msastat_batch.hip -- msa_trim_batch: native worker threads with a context each, and the batch engine (one launch per kernel
// family over whole groups of small alignments).
#include "msastat_ctx.h"

using namespace msai;

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
// every column of every alignment (each alignment's columns by weight: a counting sort per alignment on the device, behind the
// gap counts), ONE copy of every result vector back, and nothing in between that needs the host.  Two
// groups are in flight: the host takes the selection decisions of group g (trim_impl on a host-only view per alignment)
// while the device works on group g + 1.  Alignments the engine does not take, and the rare alignment whose selection needs
// another pass over the rows, go through the workers / an ordinary context as before.
struct Engine {
    struct Item {
        int32_t k;            // index in the call
        size_t res_word;      // offset (words) of its block in the result region: flags[16] gaps[npad] indets[npad] rowtot[mpad] mdk[n] q[n]
        int npad, mpad;
        size_t extra_word;    // ... and of what its kind adds behind them (engine_needs 3 - 5)
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
    int nlanes = 2;             // groups in flight
    bool trace = false;         // MSA_TRACE=1: host-side timing of every group on stderr
    long fetch_max_bytes = 1 << 20;  // page-locked alignments up to this size are fetched by a kernel instead of a copy each
    int cols_max_m = 128;       // groups whose alignments have at most this many sequences: a lane per column (MSA_BATCH_COLS_MAX <= 128)
    msa_ctx *tables = nullptr;  // owns the similarity tables (and trims the alignments that fall back)
    std::vector<msa_ctx *> views;  // the host-only views handed to trim_impl: one per worker, the last one the calling thread's
};

namespace msai {

inline size_t align_up(size_t x, size_t q) { return (x + q - 1) / q * q; }

// Can the engine take alignment k?  (the similarity pipeline's conditions, one pair-pass regime, 32-bit list offsets, rows the
// copy engine takes in one piece or that are small enough to pack on the way)
// what a trim needs of the device: 0 not a trim the engine knows, 1 the gap statistics alone (gappyout, nogaps, noallgaps, a
// manual gap threshold), 2 the similarity pipeline as well, 3 - 5 the trimmers that remove sequences (below)
int engine_needs(const msa_trim_params *p) {
    const int method = p->method;
    // the trimmers that remove sequences (round 6): their statistics come back with the group's one copy as well, and the
    // selection runs on the host-only view -- 3 OverlapTrimmer (the overlap counts of every sequence), 4 RepresentativeTrimmer
    // (the identities: clustered on the host, threshold mode and clusters=K alike), 5 noduplicateseqs (row digests)
    if (method == MSA_METHOD_NODUPLICATESEQS) return 5;
    if (p->clusters != -1 || p->max_identity != -1) return 4;
    if (p->residue_overlap != -1 && p->sequence_overlap != -1) return 3;
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
    // (a gap window is host work on the counts -- unless the similarity pipeline follows: its ">= 80 % gaps" cut reads the windowed
    // counts on the device)
    if (gap_hw > 0 && needs == 2) return false;
    if (gap_hw > n / 4) return false;  // (an error return: the ordinary path reports it)
    if (needs == 1 || needs == 3 || needs == 5) return (double)m * n <= 4e6;  // (one pass over the rows: small alignments, where launches are the cost)
    if (needs == 4) {  // the pair pass, and m x m identities in the group's copy back
        const int m_pad4 = round_up(m, 128);
        return m <= 1024 && msak::pair_pipe_regime(m, m_pad4) && (double)m * m * n <= b->engine_max_work;
    }
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
    size_t raw, planes, ident, w, wlow, wbar, row_avg, row_max, codeT, codeR, off, trow, nvalid, simnum, simden, simstate, cols, end;
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
        const int kind = engine_needs(b->params + k);
        if (kind == 2) max_m = std::max(max_m, m), any_sim = i;
        // what the kind brings back beside the common vectors: overlap counts [mpad]; identities [m][ldw]; lengths [mpad] + hashes
        it.extra_word = res_words;
        res_words += kind == 3 ? (size_t)it.mpad : kind == 4 ? (size_t)m * round_up(m, 64) : kind == 5 ? (size_t)5 * it.mpad : 0;
        mix(((uint64_t)(uint32_t)m << 32) | (uint32_t)n);
        mix(b->params[k].method == MSA_METHOD_AUTOMATED1);
        mix(engine_needs(b->params + k));
    }
    const bool multi = msak::lg_rounds_per_launch(max_m) > 0;  // (the similarity kernel in several launches: per-column state)
    // groups of small alignments: the similarity statistic with a lane per column (similarity_cols_batch_kernel) -- no
    // column-major codes, no lists
    const bool cols_mode = max_m <= e->cols_max_m;
    mix(cols_mode);
    // the wave-per-column kernel's columns dealt by weight, alignment by alignment (a counting sort per alignment on the device,
    // behind the gap counts: workgroups of four columns of like weight, the heaviest first -- over the columns as they lie a
    // workgroup's slots are held until its heaviest column is done: profiles/r05_engine_sort_ab.txt)
    const bool sort_cols = !cols_mode && any_sim >= 0 && max_m <= 15000;  // (the sort's bins live in LDS)
    mix(sort_cols);
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
        const int kind = engine_needs(b->params + k);
        if (kind == 4) {  // RepresentativeTrimmer: planes and the pair pass; the identities land in the result region
            y.ident = y.w = y.wlow = y.wbar = y.row_avg = y.row_max = y.codeT = y.codeR = y.off = y.trow = y.nvalid = y.simnum = y.simden =
                y.simstate = y.cols = 0;
            y.planes = take(((size_t)msak::planes_total() * nchunk * m_pad + 64) * 4);
            y.end = off;
            continue;
        }
        if (kind != 2) {  // the gap statistics (and what one more pass over the rows computes): the rows are all it needs on the device
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
        y.cols = sort_cols ? take(((size_t)n + 64) * 4) : 0;  // (the wave-per-column kernel's columns by weight: sort_columns_batch)
        y.end = off;
    }
    if (std::find(packed.begin(), packed.end(), 1) != packed.end()) stage_bytes = raw_bytes;  // (the staging mirrors the raw region)
    mix(multi);
    const size_t arena_bytes = off;
    const uint8_t *old_base = L.arena.p;
    HIPCHK(tc, L.arena.reserve(arena_bytes));
    mix((uint64_t)(uintptr_t)L.arena.p);
    // tables: [BAlign K][LgAlign K][prefix arrays: F x (K + 1)]
    enum { F_FETCH, F_GAPS, F_ROWTOT, F_PLANES, F_PAIRS, F_WMEANS, F_IDROWS, F_ENCODE, F_COMPACT, F_FINISH, F_COLS, F_OVERLAP, F_DIGEST, F_COUNT };  // (F_ENCODE / F_COLS: by mode)
    const size_t meta_bytes = align_up((size_t)K * sizeof(msak::BAlign), 256) + align_up((size_t)K * sizeof(msak::LgAlign), 256) +
                              align_up((size_t)F_COUNT * (K + 1) * sizeof(int32_t), 256);
    HIPCHK(tc, L.meta.reserve(meta_bytes));
    HIPCHK(tc, L.h_meta.reserve(meta_bytes));
    HIPCHK(tc, L.h_res.reserve(res_words * 4 + 64));
    if (stage_bytes) {
        // (a grown staging buffer starts as zeros: the runs copied out of it span the slack and the alignment gaps between the packed
        // alignments, which only ever hold what the buffer held when it was allocated -- the arena's padding stays zero whatever
        // way the rows arrive)
        const size_t had = L.h_stage.cap;
        HIPCHK(tc, L.h_stage.reserve(stage_bytes));
        if (L.h_stage.cap != had) std::memset(L.h_stage.p, 0, L.h_stage.cap);
    }
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
        d.kind = engine_needs(b->params + k);
        d.extra = res_d + it.extra_word;
        d.gated = d.kind == 2 && b->params[k].method == MSA_METHOD_AUTOMATED1;
        d.ident = d.gated ? reinterpret_cast<float *>(A + y.ident) : nullptr;
        d.w = reinterpret_cast<float *>(A + y.w);
        d.wlow = reinterpret_cast<float *>(A + y.wlow);
        if (d.kind == 4) d.ident = reinterpret_cast<float *>(d.extra), d.w = nullptr, d.wlow = nullptr;
        if (d.kind == 3) d.ov_need = static_cast<int>(std::ceil(b->params[k].residue_overlap * static_cast<float>(m - 1)));
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
        const bool sim = d.kind == 2;
        // (only the alignments whose trim runs the similarity kernel: max_m -- the sort's LDS bins -- is taken over those)
        g.cols = sort_cols && sim ? reinterpret_cast<const int32_t *>(A + y.cols) : nullptr;
        g.ldk = d.ldk, g.m = m, g.n = n, g.ldw = d.ldw, g.ncols = n;
        lt[i] = g;
        auto add = [&](int f, int blocks) {
            const bool runs = sim || f <= F_ROWTOT || (d.kind == 4 && (f == F_PLANES || f == F_PAIRS)) || (d.kind == 3 && f == F_OVERLAP) ||
                              (d.kind == 5 && f == F_DIGEST);
            pf[(size_t)f * (K + 1) + i + 1] = pf[(size_t)f * (K + 1) + i] + (runs ? blocks : 0);
        };
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
        add(F_OVERLAP, d.kind == 3 ? (m + 3) / 4 : 0);
        add(F_DIGEST, d.kind == 5 ? (m + 3) / 4 : 0);
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
    if (sort_cols) msak::launch_sort_columns_batch(st, bt_d, lt_d, K, max_m);
    msak::launch_row_nongap_batch(st, bt_d, PF(F_ROWTOT), K, NB(F_ROWTOT));
    msak::launch_overlap_rows_batch(st, bt_d, PF(F_OVERLAP), K, NB(F_OVERLAP));
    msak::launch_row_digest_batch(st, bt_d, PF(F_DIGEST), K, NB(F_DIGEST));
    msak::launch_prep_planes_batch(st, bt_d, PF(F_PLANES), K, NB(F_PLANES));
    int min_nchunk = 1 << 30;
    for (int i = 0; i < K; ++i) {
        const int kind = engine_needs(b->params + ks[i]);
        if (kind == 2 || kind == 4) min_nchunk = std::min(min_nchunk, (b->n[ks[i]] + 31) / 32);
    }
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
    // what the trimmers that remove sequences read (engine_needs 3 - 5)
    const int kind = engine_needs(b->params + k);
    const int32_t *extra = res + it.extra_word;
    v->ov_valid = false;
    v->pref_ident = nullptr, v->pref_lengths = nullptr, v->pref_hashes = nullptr;
    if (kind == 3) {  // Cleaner::calculateSpuriousVector's values, as overlap() derives them from the counts
        v->ov_vals.resize(m);
        for (int i = 0; i < m; ++i) v->ov_vals[i] = static_cast<float>(extra[i]) / n;
        v->ov_key = b->params[k].residue_overlap;
        v->ov_valid = true;
        v->ov_colcnt = false;
    } else if (kind == 4) {
        v->pref_ident = reinterpret_cast<const float *>(extra);
        v->ldw = round_up(m, 64);
    } else if (kind == 5) {
        v->pref_lengths = extra;
        v->pref_hashes = reinterpret_cast<const unsigned long long *>(extra + it.mpad);
    }
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
        e->trace = std::getenv("MSA_TRACE") != nullptr;
        if (msak::diagnostics_enabled())
            if (const char *ev = std::getenv("MSA_BATCH_COLS_MAX")) e->cols_max_m = std::min(128, std::atoi(ev));  // (the kernel's LDS tile)
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
        rc = msa_upload_packed_async(tc, b->data[k], b->m[k], b->n[k], b->ld[k], b->indet[k]);
        if (rc == MSA_OK) rc = msa_trim(tc, b->params + k, b->keep_res[k], b->keep_seq[k], info);
        else (void)hipStreamSynchronize(tc->stream);
        b->only_gaps[k] = tc->only_gaps_rows;
        b->rc[k] = rc;
    }
    return MSA_OK;
}

}  // namespace msai

namespace msai {
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
                rc = msa_upload_packed_async(c, b->data[k], b->m[k], b->n[k], b->ld[k], b->indet[k]);
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
}  // namespace msai

extern "C" {

int msa_batch_create(int device, int32_t workers, msa_batch **out) {
    if (!out || workers < 1 || workers > 64) return MSA_E_INVALID;
    *out = nullptr;
    msa_batch *b = new (std::nothrow) msa_batch();
    if (!b) return MSA_E_NOMEM;
    b->device = device;
    if (msak::diagnostics_enabled()) {  // (MSA_DIAGNOSTICS: msastat_kernels.hip)
        if (const char *e = std::getenv("MSA_BATCH_ENGINE")) b->use_engine = std::atoi(e) != 0;
        if (const char *e = std::getenv("MSA_BATCH_ENGINE_MAX")) b->engine_max_work = std::atof(e);
        if (const char *e = std::getenv("MSA_BATCH_ENGINE_MIN")) b->engine_min_count = std::atoi(e);
    }
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
    if (std::getenv("MSA_TRACE"))
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

