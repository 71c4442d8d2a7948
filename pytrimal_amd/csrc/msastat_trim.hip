// msastat_trim.hip -- msa_trim (the trimAlManager::clean_alignment equivalent: the alignment and the m x m matrices stay on the
// device, O(m + n) vectors move) and its two launch sequences: the similarity pipeline and the compact pipeline of small alignments.
#include "msastat_ctx.h"

namespace msai {

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
// 0.08 ms of kernels (profiles/r04_small_latency_ordinary_launch_sequence.jsonl).  Up to 1024 sequences (512 until late in round 5),
// and no gap window, the same statistics take THREE launches and no copy
// (msak::CompactArgs):
//   front  -- gap / indetermination counts, residues per sequence, planes, column-major codes (+ lists; the ">= 80 % gaps" cut
//             from the block's own counts), all from one pass over the rows by independent blocks; no memset in front of it;
//   pairs  -- the ordinary pair pass, whose tiles also add up the rows' weight sums (the predictor's input);
//   [automated1: the identity statistics with the selectMethod gate, one launch]
//   sim    -- up to `flat_max_m` sequences the flat kernel, else the wave-per-column kernel over the columns in their own order --
//             from 513 sequences on dealt by weight: the host sorts them behind the front kernel's event while the pair pass runs;
//             the wave that finishes a column writes its MDK and Q;
//   every result is stored by the kernels into a mirror of the state block in pinned host memory (h_cres): one wait, then the
//   host folds the front kernel's per-block verdicts into the two flag words.
// The kernels' arithmetic is the ordinary path's (the same device functions); MSA_COMPACT=0 switches the pipeline off.
#ifndef MSA_COMPACT_SIM_MAX_M  // (A/B builds: tools/build_variant.sh)
#define MSA_COMPACT_SIM_MAX_M 1024
#endif
#ifndef MSA_COMPACT_ANY_N  // 1: any number of columns (beyond 20 per compute unit -- more waves than slots -- always dealt by weight)
#define MSA_COMPACT_ANY_N 1
#endif
#ifndef MSA_COMPACT_SORT_FROM_M  // sequences from which on the similarity kernel of the pipeline gets its columns dealt by weight
#define MSA_COMPACT_SORT_FROM_M 513
#endif
bool compact_sim_applies(const msa_ctx *c, int gap_hw) {
    const msak::Tuning &t = c->tuning;
    return t.compact != 0 && gap_hw == 0 && t.sim_kernel == 0 && (t.sim_mode & 64) == 0 && t.lg_rounds < 0 && t.lg_split == 0 &&
           t.lg_big == 0 && c->m >= 2 && c->m <= MSA_COMPACT_SIM_MAX_M && (MSA_COMPACT_ANY_N || c->n <= c->cus * 20) &&
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
    a.cw = 64;
    c->compact_cw = 64;
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
        const int ncb = msak::bx_cols_pad(n) / c->compact_cw;
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
// ... and, for OverlapTrimmer, the overlap kernels behind it (they read the counts the front kernel leaves on the device): one wait
// where the front kernel's and overlap()'s were two (the reference's ENOG411BWBU fixture, OverlapTrimmer(80, 0.8): 0.094 ms in round 4)
int compact_overlap(msa_ctx *c, float residue_overlap, float sequence_overlap) {
    int rc = compact_prepare(c);
    if (rc) return rc;
    msak::CompactArgs a = compact_args(c);
    {
        ProfScope ps(c, "gaps");
        msak::launch_compact_front(c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    c->errkey_dirty = false;
    // the overlap counts, and what remove_all_gaps will ask for if sequences go: the residues per column over the sequences that
    // stay -- the mask decided on the device as the host will decide it (the same float division and comparison), counted over at
    // once; two launches that store into pinned host memory themselves (round 5, late: they were two overlap kernels, the
    // decision, a memset, the counts and three copies -- nine queue operations behind the front kernel)
    const int m = c->m, n = c->n;
    const float fo = residue_overlap * static_cast<float>(m - 1);
    const int need = static_cast<int>(std::ceil(fo));
    HIPCHK(c, c->good.reserve((size_t)m + 64));
    HIPCHK(c, c->h_i32.reserve((size_t)std::max(m, 2 * n) + 4));
    HIPCHK(c, c->keep_seq_d.reserve((size_t)m + 64));
    HIPCHK(c, c->col_cnt.reserve((size_t)n + 64));
    HIPCHK(c, c->h_colcnt.reserve((size_t)n + 4));
    HIPCHK(c, c->h_u8.reserve(256 + (size_t)std::max(m, n)));
    {
        ProfScope ps(c, "overlap");
        msak::launch_overlap_small(c->stream, c->raw, m, n, c->ld, c->indet, c->gaps.p, c->indets.p, need, sequence_overlap / 100.0F,
                                   c->good.p, c->h_i32.p, c->keep_seq_d.p, c->h_u8.p, c->col_cnt.p, c->h_colcnt.p);
    }
    HIPCHK(c, hipGetLastError());
    if ((rc = compact_fetch(c, false))) return rc;
    c->ov_vals.resize(m);
    for (int i = 0; i < m; ++i) c->ov_vals[i] = static_cast<float>(c->h_i32.p[i]) / n;
    c->ov_key = residue_overlap;
    c->ov_valid = true;
    c->ov_keep.assign(c->h_u8.p, c->h_u8.p + m);
    c->ov_colcnt = true;
    return MSA_OK;
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
    a.cw = c->compact_cw = msak::compact_front_cw(m, n, true, c->raw == c->h_raw.p);
    a.xcd = c->tuning.front_xcd;
    a.nt = msak::compact_front_nt(m, n, c->cus);
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
    // the flat kernel up to flat_max_m sequences -- with two waves per column, and from three quarters of that size on, only while
    // both waves of every column are resident at once (profiles/r05_flat_sweep.jsonl, one strict trim, flat / wave per column, ms:
    // 128 x 1200 0.098 / 0.114, 160 x 1200 0.118 / 0.112; 100 x 4000 0.132 / 0.143, 128 x 4000 0.161 / 0.157)
    const int flat_m = std::min(c->tuning.flat_max_m, msak::flat_rows_max());
    const bool flat = m <= flat_m && (n <= c->cus * 10 || m <= flat_m * 3 / 4);
    a.lists = flat ? 0 : 1;
    // (compact_sim_applies keeps the shapes and switches out for which the wave-per-column kernel would not finish the columns
    // itself; should the two ever disagree, nothing has been enqueued yet and the callers take the ordinary pipeline)
    if (!flat && !msak::lg_finishes(L, c->cus)) return MSA_E_FALLBACK;
    // From 513 sequences on the columns are dealt to the similarity kernel's waves BY WEIGHT, as in the ordinary pipeline: a launch
    // of ~1000 rows takes what one wave takes for its heaviest column plus what the deal leaves uneven, and over the columns as they
    // lie that is 0.40 instead of 0.33 ms at 1000 x 4000 (profiles/r05_compact_1024_ab.txt).  The host sorts while the pair pass
    // runs: it waits for the front kernel alone (an event), takes the counts from the mirror, and the list goes up in front of
    // the similarity kernel -- every column is in it, the ones the ">= 80 % gaps" rule cuts last (their waves write the zeros).
    const bool sorted = !flat && (m >= MSA_COMPACT_SORT_FROM_M || n > c->cus * 20);
    c->compact_sorted = sorted;
    {
        ProfScope ps(c, "front");
        msak::launch_compact_front(c->stream, a);
    }
    if (sorted) {
        if (!c->ev_front) HIPCHK(c, hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_front, c->stream));
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
    if (sorted) {
        HIPCHK(c, hipEventSynchronize(c->ev_front));
        const int32_t *H = c->h_cres.p;
        c->h_gaps.assign(H + ST_WORDS, H + ST_WORDS + n);
        c->h_indets.assign(H + ST_WORDS + c->state_npad, H + ST_WORDS + c->state_npad + n);
        SimOrder ord;
        if ((rc = build_sim_order(c, nullptr, &ord))) return rc;
        int32_t *list = c->h_simcols.p;
        int k = ord.npad;
        for (int j = 0; j < n; ++j)
            if (((float)c->h_gaps[j] / (float)m) >= 0.8f) list[k++] = j;
        ord.npad = k;  // (= n)
        if ((rc = sim_order_enqueue(c, ord, c->stream))) return rc;
        L.cols = c->simcols.p;
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
int trim_impl(msa_ctx *c, const msa_trim_params *p, uint8_t *keep_res, uint8_t *keep_seq, msa_trim_info *info) {
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
        // (what the engine takes: the similarity pipeline without a gap window, a trim that needs the gap statistics alone, the
        // trimmers that remove sequences -- their statistics are in the view: engine_needs)
        const int kind = engine_needs(p);
        if ((int)c->h_gaps.size() != n || !((pipelined && gap_hw == 0) || kind == 1 || kind >= 3)) return MSA_E_FALLBACK;
        gaps_w.resize(n);
        if ((rc = msah::window_i32(c->h_gaps.data(), n, gap_hw, gaps_w.data()))) return rc;
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
        c->paths[1] = c->compact_sorted ? MSA_PATH_PIPE_COMPACT_SORTED : MSA_PATH_PIPE_COMPACT;
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
        const bool overlap_mode = method != MSA_METHOD_NODUPLICATESEQS && p->clusters == -1 && p->max_identity == -1 &&
                                  p->residue_overlap != -1 && p->sequence_overlap != -1;
        if (compact_gaps_applies(c)) {
            rc = overlap_mode ? compact_overlap(c, p->residue_overlap, p->sequence_overlap) : compact_gaps(c);
            c->paths[1] = MSA_PATH_PIPE_COMPACT_GAPS;
        } else rc = stage_row_totals(c);
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
        // (staged by compact_overlap: the column counts over the device's mask serve remove_all_gaps if that mask is this one)
        c->colcnt_staged = c->ov_colcnt && (int)c->ov_keep.size() == m && std::memcmp(c->ov_keep.data(), keep_seq, (size_t)m) == 0;
        c->ov_colcnt = false;
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

}  // namespace msai
using namespace msai;

extern "C" {

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

int msa_trim_only_gaps_rows(msa_ctx *c, int32_t *rows, int32_t cap) {
    if (!c || cap < 0 || (!rows && cap > 0)) return MSA_E_INVALID;
    const int count = (int)c->only_gaps_rows.size();
    std::copy_n(c->only_gaps_rows.begin(), std::min(count, (int)cap), rows);
    return count;
}

}  // extern "C"
