// msastat_ctx.hip -- the context of the C-ABI shim (include/msastat.h): creation, the one wait every call goes through, the state
// block, uploads (host rows -> the device's pitched layout, or read in place), instrumentation.
#include "msastat_ctx.h"

namespace msai {

int fail_hip(msa_ctx *c, hipError_t e, const char *what) {
    std::snprintf(c->hip_err, sizeof(c->hip_err), "%s: %s", what, hipGetErrorString(e));
    return MSA_E_HIP;
}
int sync_stream(msa_ctx *c) {
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
    c->ov_valid = c->ov_colcnt = false;
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

}  // namespace msai
using namespace msai;

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
    c->raw_lin.release(); c->wmat.release(); c->wlow.release(); c->wbar.release(); c->codeT.release(); c->simcols.release(); c->h_simcols.release(); c->bx_off.release(); c->bx_trow.release(); c->bx_nvalid.release(); c->hit.release(); c->dst.release(); c->row_avg.release(); c->row_max.release(); c->row_min.release();
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
    if (c->ev_front) (void)hipEventDestroy(c->ev_front);
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipEventDestroy(c->ev_fork);
        (void)hipEventDestroy(c->ev_join);
        (void)hipStreamDestroy(c->stream2);
    }
    for (int p = 0; p < 2; ++p)
        if (c->part_stream[p]) {
            (void)hipStreamSynchronize(c->part_stream[p]);
            (void)hipEventDestroy(c->part_join[p]);
            (void)hipStreamDestroy(c->part_stream[p]);
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

namespace msai {
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
        int k = 3;  // (round 3's sweep, profiles/r03_upload.txt: three helpers beside the calling thread saturate the link)
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
    const int rows_per_piece = std::max<int>(1, (int)(((size_t)1 << 20) / (size_t)c->ld));  // pieces of 1 MB (round 3's sweep: profiles/r03_upload.txt)
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
}  // namespace msai
using namespace msai;

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
        } else if (c->tuning.upload_direct && (size_t)m * (size_t)ld >= ((size_t)1 << 20) && (size_t)(ld - n) * 8 <= (size_t)ld) {
            // a contiguous matrix of odd-sized rows (5000 x 5000 from pageable memory: the BASELINE's C4), a megabyte or more and
            // not much wider than its rows: ONE linear copy as it lies -- the runtime stages pageable memory at the link's rate,
            // which the packed pieces below do not reach (they are packed by host threads first) -- and a kernel lays the rows out
            // at the device pitch, padding zeroed (round 5, late: profiles/r05_upload.txt)
            c->paths[0] = MSA_PATH_UPLOAD_REPITCHED;
            c->raw_own.tag = 0;
            const size_t bytes = (size_t)(m - 1) * (size_t)ld + (size_t)n;
            HIPCHK(c, c->raw_lin.reserve(bytes + 256));
            HIPCHK(c, hipMemcpyAsync(c->raw_lin.p, rowmajor, bytes, hipMemcpyHostToDevice, c->stream));
            msak::launch_repitch_rows(c->stream, c->raw_lin.p, ld, c->raw_own.p, c->ld, m, n);
            HIPCHK(c, hipGetLastError());
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

int msa_debug_switches_enabled(void) { return msak::diagnostics_enabled() ? 1 : 0; }

int msa_debug_last_paths(msa_ctx *c, int32_t out[8]) {
    if (!c || !out) return MSA_E_INVALID;
    std::copy_n(c->paths, 8, out);
    return MSA_OK;
}

void msa_prof_enable(msa_ctx *c, int enable) {
    if (c) c->prof_on = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
}

}  // extern "C"
