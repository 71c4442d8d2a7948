// msastat_stats.hip -- one entry point per statistic of include/msastat.h (gap counts, pair counts, identities, similarity,
// overlap) and the device-side pieces msa_trim builds on (all-gap removal, duplicates, clustering).
#include "msastat_ctx.h"

namespace msai {


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
        // (Q is a quotient of two sums of terms >= 0 or 0: never a NaN -- unless the pipelined split-column kernel gave up on an internal
        // wait and said so in its sums: similarity_lg_pipe_body)
        if (dq[i] != dq[i]) return fail_hip(c, hipErrorUnknown, "similarity kernel: an internal wait between the waves of a column timed out");
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
        if (!msak::launch_sim_lists_fused(st, c->raw, m, n, c->ld, c->lut.p, gw_dev, c->codeT.p, c->ldw, npos, c->bx_off.p, c->bx_trow.p,
                                          c->bx_nvalid.p, c->errkey.p)) {
            msak::launch_sim_encode_cm(st, c->raw, m, n, c->ld, c->lut.p, gw_dev, c->codeT.p, c->errkey.p);
            msak::launch_bx_compact(st, c->codeT.p, m, n, c->ldw, npos, c->bx_off.p, c->bx_trow.p, c->bx_nvalid.p);
        }
    }
    HIPCHK(c, hipGetLastError());
    return MSA_OK;
}
// 2. the column list (built on the host into h_simcols)
int sim_order_enqueue(msa_ctx *c, const SimOrder &ord, hipStream_t st) {
    HIPCHK(c, c->simcols.reserve((size_t)ord.npad + 8));
    // a pass whose columns run as staggered parts on as many streams (msak::lg_halves; two parts, MSA_LG_PARTS: up to four): the list
    // part by part -- part p = its entries p, p + P, ...: every part by weight, all of the same weight
    c->sim_halves = msak::lg_halves(c->m, ord.npad, c->cus, true);
    if (c->sim_halves) {
        const int P = msak::lg_parts();
        if (!c->stream2) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        for (int p = 0; p < P - 2; ++p)
            if (!c->part_stream[p]) {
                HIPCHK(c, hipStreamCreateWithFlags(&c->part_stream[p], hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->part_join[p], hipEventDisableTiming));
            }
        std::vector<int32_t> tmp(c->h_simcols.p, c->h_simcols.p + ord.npad);
        int k = 0;
        for (int p = 0; p < P; ++p)
            for (int i = p; i < ord.npad; i += P) c->h_simcols.p[k++] = tmp[i];
    }
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
                                msak::LgSide side = {msak::lg_parts(), {c->stream2, c->part_stream[0], c->part_stream[1]}, c->ev_fork,
                                                     {c->ev_join, c->part_join[0], c->part_join[1]}};
                                msak::LgXsegBufs xb = {};
                                if (msak::lg_xseg(m, ord.npad, c->cus) && c->xsegbuf.reserve(msak::lg_xseg_bytes(ord.npad)) == hipSuccess)
                                    xb = msak::lg_xseg_bufs(c->xsegbuf.p, ord.npad);
                                return msak::launch_similarity_lg(c->stream, a, npos, c->tab.p, c->cus, &c->sim_launches,
                                                                  c->sim_halves && c->stream2 ? &side : nullptr, xb.dep ? &xb : nullptr);
                            }();
        if (e) return fail_hip(c, (hipError_t)e, "launch_similarity");
    }
    msak::launch_sim_finish(c->stream, c->simnum.p, c->simden.p, gw_dev, m, n, c->mdk.p + n, c->mdk.p);
    HIPCHK(c, hipGetLastError());
    return MSA_OK;
}


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
            c->paths[1] = c->compact_sorted ? MSA_PATH_PIPE_COMPACT_SORTED : MSA_PATH_PIPE_COMPACT;
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

// the overlap kernels behind whatever the stream holds, their counts on the way to h_i32 (no wait)
int overlap_enqueue(msa_ctx *c, float residue_overlap) {
    const int m = c->m, n = c->n;
    const float fo = residue_overlap * static_cast<float>(m - 1);
    const int need = static_cast<int>(std::ceil(fo));
    HIPCHK(c, c->col_ok.reserve((size_t)3 * c->nchunk + 64));
    HIPCHK(c, c->good.reserve((size_t)m + 64));
    HIPCHK(c, c->h_i32.reserve((size_t)std::max(m, 2 * n) + 4));
    {
        ProfScope ps(c, "overlap");
        msak::launch_overlap(c->stream, c->raw, m, n, c->ld, c->indet, c->gaps.p, c->indets.p, need, c->col_ok.p,
                             c->nchunk, c->good.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_i32.p, c->good.p, sizeof(int32_t) * m, hipMemcpyDeviceToHost, c->stream));
    return MSA_OK;
}

int overlap(msa_ctx *c, float residue_overlap, float *out) {
    if (c->ov_valid && c->ov_key == residue_overlap && (int)c->ov_vals.size() == c->m) {  // (staged by compact_overlap)
        std::copy(c->ov_vals.begin(), c->ov_vals.end(), out);
        return MSA_OK;
    }
    int rc = ensure_gaps(c, false);
    if (rc) return rc;
    const int m = c->m, n = c->n;
    if ((rc = overlap_enqueue(c, residue_overlap))) return rc;
    SYNC(c);
    for (int i = 0; i < m; ++i) out[i] = static_cast<float>(c->h_i32.p[i]) / n;
    return MSA_OK;
}

// Cleaner::removeAllGapsSeqsAndCols: first sequences (over kept columns), then columns (over
// the updated sequences).
// Residues per sequence over ALL columns, enqueued without waiting (the next synchronisation completes the copy):
// remove_all_gaps can then tell from the host that no sequence can have lost all its residues.
int stage_row_totals(msa_ctx *c, hipStream_t st) {
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
    if (m <= 1024) {  // a small alignment: one launch that stores into the pinned vector itself (no memset, no copy)
        msak::launch_col_nongap_small(c->stream, c->raw, m, n, c->ld, c->keep_seq_d.p, c->col_cnt.p, c->h_colcnt.p);
        HIPCHK(c, hipGetLastError());
        c->colcnt_staged = true;
        return MSA_OK;
    }
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
    if (c->prefetched) {  // (the batch engine's host-only view: the digests came back with the group's copy)
        if (!c->pref_hashes || !c->host_rows) return MSA_E_FALLBACK;
        hashes.assign(c->pref_hashes, c->pref_hashes + 2 * (size_t)m);
    } else {
        int rc = row_digest(c, lengths, &hashes);
        if (rc) return rc;
    }
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
    if (c->prefetched) {  // rows of equal digests compared where they lie, on the host (rare, and the rows are the caller's)
        for (int p = 0; p < npairs; ++p) {
            const uint8_t *x = c->host_rows + (size_t)pairs[2 * p] * c->host_ld, *y = c->host_rows + (size_t)pairs[2 * p + 1] * c->host_ld;
            if (std::memcmp(x, y, (size_t)c->n) == 0) keep_seq[std::min(pairs[2 * p], pairs[2 * p + 1])] = 0;
        }
        return MSA_OK;
    }
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
    if (m < 2 || c->prefetched) return MSA_E_FALLBACK;  // (a host-only view clusters on the host, from the identities it holds)
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
    HIPCHK(c, c->h_u8.reserve(256 + (size_t)std::max(m, c->n)));
    {
        // (the number of representatives is not asked for here: no counter to zero; up to 4096 sequences the mask lands in pinned
        // host memory by the kernel's own byte stores: no copy behind it)
        ProfScope ps(c, "cluster");
        if (msak::launch_cluster(c->stream, c->ident.p, c->ldw, c->pairs.p, m, max_identity, c->col_ok.p,
                                 c->keep_seq_d.p, nullptr, m <= 4096 ? c->h_u8.p : nullptr) != 0)
            return MSA_E_FALLBACK;  // too many sequences for the LDS bit sets: host path
    }
    HIPCHK(c, hipGetLastError());
    if ((rc = stage_kept_column_counts(c, lengths))) return rc;
    if (m > 4096) HIPCHK(c, hipMemcpyAsync(c->h_u8.p, c->keep_seq_d.p, m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    std::memcpy(keep_seq, c->h_u8.p, m);
    return MSA_OK;
}

// RepresentativeTrimmer(clusters=K): Cleaner::getCutPointClusters bisects the identity threshold until the greedy
// clustering yields K clusters.  Start value and bounds come from the row statistics (the start value is the
// selectMethod mean, same order of operations), every probe is one run of the device clustering: no m*m transfer.
int device_cluster_count(msa_ctx *c, int clusters, uint8_t *keep_seq) {
    const int m = c->m;
    if (m < 2 || clusters < 1 || c->prefetched) return MSA_E_FALLBACK;
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
                                     c->keep_seq_d.p, c->equal.p, nullptr) != 0)
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
    if (c->prefetched) {
        if (!c->pref_ident) return MSA_E_FALLBACK;
        host.resize((size_t)c->m * c->m);
        for (int i = 0; i < c->m; ++i) std::memcpy(host.data() + (size_t)i * c->m, c->pref_ident + (size_t)i * c->ldw, sizeof(float) * c->m);
        return MSA_OK;
    }
    int rc = run_pairs(c, true, false, false);
    if (rc) return rc;
    host.resize((size_t)c->m * c->m);
    HIPCHK(c, hipMemcpy2DAsync(host.data(), (size_t)c->m * sizeof(float), c->ident.p, (size_t)c->ldw * sizeof(float),
                               (size_t)c->m * sizeof(float), c->m, hipMemcpyDeviceToHost, c->stream));
    SYNC(c);
    return MSA_OK;
}

}  // namespace msai
using namespace msai;

extern "C" {

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

}  // extern "C"
