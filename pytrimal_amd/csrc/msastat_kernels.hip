// msastat_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the MSA statistics path.
//
// Data layout in HBM (all produced on the device from the row-major residue bytes):
//   raw     [m][ld]              u8   residues as given (ld % 64 == 0, columns >= n undefined)
//   planes  [8][nchunk][m_pad]   u32  bit-sliced rows: plane p<7 = bit p of the ASCII byte,
//                                      plane 7 = validity (not '-' and not indet); bit b of word
//                                      (chunk c, row r) is column 32c+b.  Row index is fastest, so a
//                                      wave whose lanes own 64 consecutive rows loads 256 B per plane.
//   gaps / indet [n]             i32  per-column '-' / indetermination counts
//   ident [m][ldw]               f32  pairwise identity, symmetric (ldw % 64 == 0, pad = 0)
//   w     [m][ldw]               f32  1 - identity, STRICTLY UPPER triangular (0 elsewhere)
//   wlow  [m + 2][ldw]           f32  the same weights mirrored (strictly LOWER triangular): msastat_simx.hip
//   tab     [29][32]             f32x2 {distance, both-valid} indexed by (row index, column index)
// The similarity kernels and their layouts (column-major codes, compacted lists) are in msastat_simx.hip.
//
// No MFMA anywhere: this is integer / lookup / ordered-fp32 work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "msastat_kernels.h"
#include "msastat_device.h"

namespace msak {

// (the bodies shared with the compact pipeline of msastat_simx.hip -- planes, pair tiles, MDK, row totals -- are in msastat_device.h)
constexpr int PLANES_TOTAL = 8;

__global__ __launch_bounds__(256) void prep_planes_kernel(const uint8_t *__restrict__ raw, int m, int n,
                                                          int64_t ld, uint32_t indet4, uint32_t *__restrict__ planes,
                                                          int nchunk, int m_pad, int *__restrict__ err_flag) {
    prep_planes_body(raw, m, n, ld, indet4, planes, nchunk, m_pad, err_flag, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- batches: one launch per kernel family for every alignment of a shard (msa_trim_batch) -------------------------------
// A table of BAlign descriptors in device memory and, per family, the prefix sums of its blocks per alignment: a block
// finds its alignment by bisection (wave-uniform: scalar loads) and runs the single-alignment body on its descriptor.
typedef const __attribute__((address_space(4))) int32_t *ci32p;
__device__ __forceinline__ int batch_find(const int32_t *prefix_, int K, int idx, int &local) {
    ci32p prefix = (ci32p)(uint64_t)prefix_;
    int lo = 0, hi = K;  // prefix[lo] <= idx < prefix[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= idx) lo = mid;
        else hi = mid;
    }
    local = idx - prefix[lo];
    return lo;
}
__device__ __forceinline__ BAlign batch_desc(const BAlign *table, int a) {  // wave-uniform copy: scalar loads
    typedef const __attribute__((address_space(4))) uint32_t *cw;
    cw src = (cw)(uint64_t)(table + a);
    uint32_t words[sizeof(BAlign) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(BAlign) / 4); ++i) words[i] = src[i];
    BAlign d;
    __builtin_memcpy(&d, words, sizeof(BAlign));
    return d;
}
// Page-locked caller rows -> the device's pitched rows, read over the link by the kernel itself: ONE launch for every
// alignment of a group instead of a copy each (a copy per 100 KB alignment costs the copy queue 8 us: 12 GB/s; with a few
// thousand loads in flight the link runs at its rate).  A thread moves 16 bytes of a destination row; columns n .. ld are
// written as zeros.  Alignments whose rows arrive by copy have fetch_src == nullptr: their blocks return at once.
__global__ __launch_bounds__(256) void fetch_rows_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    if (!d.fetch_src) return;
    const int64_t chunk = (int64_t)local * 256 + threadIdx.x;  // 16-byte chunk of the destination
    const int per_row = (int)(d.ld / 16);
    const int r = (int)(chunk / per_row), c = (int)(chunk % per_row) * 16;
    if (r >= d.m) return;
    const uint8_t *src = d.fetch_src + (size_t)r * d.fetch_ld + c;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (c + 16 <= d.n && (reinterpret_cast<uintptr_t>(src) & 3u) == 0) {
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
        w[0] = s4[0], w[1] = s4[1], w[2] = s4[2], w[3] = s4[3];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (c + i < d.n) w[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
    }
    *reinterpret_cast<uint4 *>(const_cast<uint8_t *>(d.raw) + (size_t)r * d.ld + c) = make_uint4(w[0], w[1], w[2], w[3]);
}
__global__ __launch_bounds__(256) void prep_planes_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    const int nbx = (d.nchunk + 1) / 2;
    prep_planes_body(d.raw, d.m, d.n, d.ld, d.indet4, d.planes, d.nchunk, d.m_pad, d.flags + 0, local % nbx, local / nbx);
}

// ------------------------------------------------------------------------------------------
// gap_counts: per-column '-' and indetermination counts (statistics::Gaps::CalculateVectors).
// One thread = 4 adjacent columns (one dword per row, a wave reads 256 contiguous bytes per
// row), SWAR byte counters over a slab of <= 128 rows, then one integer atomic per column.
// HBM-bound: reads m*n bytes once.
// ------------------------------------------------------------------------------------------
constexpr int GAP_SLAB = 64;
__device__ __forceinline__ void gap_counts_body(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                uint32_t indet4, int32_t *__restrict__ gaps,
                                                int32_t *__restrict__ indets, int bx, int by) {
    const int c4 = bx * 256 + threadIdx.x;  // dword column
    const bool active = (int64_t)c4 * 4 < ld;
    if (active) {
        const int r0 = by * GAP_SLAB;
        const int r1 = min(m, r0 + GAP_SLAB);
        const uint32_t *p = reinterpret_cast<const uint32_t *>(raw + (size_t)r0 * ld) + c4;
        const size_t stride = (size_t)(ld >> 2);
        uint32_t accg = 0, acci = 0;
        auto take = [&](uint32_t x) {
            accg += zero_bytes(x ^ 0x2d2d2d2du) >> 7;
            acci += zero_bytes(x ^ indet4) >> 7;
        };
        int r = r0;
        for (; r + 8 <= r1; r += 8) {  // eight rows requested before the first is looked at
            uint32_t xs[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xs[i] = p[(size_t)i * stride];
            p += 8 * stride;
#pragma unroll
            for (int i = 0; i < 8; ++i) take(xs[i]);
        }
        for (; r < r1; ++r) {
            take(*p);
            p += stride;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c4 * 4 + k;
            if (c < n) {
                uint32_t g = (accg >> (8 * k)) & 0xFFu, x = (acci >> (8 * k)) & 0xFFu;
                if (g) atomicAdd(&gaps[c], (int)g);
                if (x) atomicAdd(&indets[c], (int)x);
            }
        }
    }
}
__global__ __launch_bounds__(256) void gap_counts_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                         uint32_t indet4, int32_t *__restrict__ gaps,
                                                         int32_t *__restrict__ indets) {
    gap_counts_body(raw, m, n, ld, indet4, gaps, indets, (int)blockIdx.x, (int)blockIdx.y);
}
__global__ __launch_bounds__(256) void gap_counts_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    const int nbx = (int)((d.ld / 4 + 255) / 256);
    gap_counts_body(d.raw, d.m, d.n, d.ld, d.indet4, d.gaps, d.indets, local % nbx, local / nbx);
}

// ------------------------------------------------------------------------------------------
// pair_counts: hit/dst of every sequence pair (Cleaner::calculateSeqIdentity ==
// Similarity::calculateMatrixIdentity integers).  One wave = TI rows "i" (wave-uniform, read
// through the scalar cache, used as SGPR operands) x 64*TJ rows "j" (TJ per lane; TJ = 1 in production:
// the kernel is short of waves, not of reuse); per 32
// columns and pair: 7 xor/or (v_or3 / v_bitop3 fuse most of them) + and-not + 2 bcnt + or.
// Integer work, any order is exact.
// ------------------------------------------------------------------------------------------

// the loop over the chunks of a tile on the raw planes, as the compiler schedules it
template <int TI, int TJ>
__device__ __forceinline__ void pair_loop_plain(const uint32_t *__restrict__ planes, int nchunk, int m_pad, int i0, int j0, int lane,
                                                uint32_t (&hit)[TJ][TI], uint32_t (&dst)[TJ][TI]) {
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
        for (int t = 0; t < TI; ++t) hit[u][t] = dst[u][t] = 0;
    const size_t ps = (size_t)nchunk * m_pad;
    const uint32_t *pj = planes + j0 + lane;
    const uint32_t *pi = planes + i0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int c = 0; c < nchunk; ++c) {
        const size_t off = (size_t)c * m_pad;
        uint32_t b[TJ][8];
#pragma unroll
        for (int u = 0; u < TJ; ++u)
#pragma unroll
            for (int p = 0; p < 8; ++p) b[u][p] = pj[off + p * ps + 64 * u];
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const uint32_t *q = pi + off + t;  // wave-uniform address -> scalar loads
            const uint32_t a0 = q[0], a1 = q[ps], a2 = q[2 * ps], a3 = q[3 * ps], a4 = q[4 * ps], a5 = q[5 * ps],
                           a6 = q[6 * ps], vi = q[7 * ps];
            const uint32_t nvi = ~vi;  // (scalar) columns in which row i holds no residue never count as hits
#pragma unroll
            for (int u = 0; u < TJ; ++u) {
                // d = "differs or not counted": one three-input boolean per plane, and hits are what is left --
                // counted as misses (32 per chunk minus the hits), which saves the and-not in front of the popcount
                // (v_bitop3_b32, truth table 0xF6 = x | (y ^ z): spelled out, the compiler pairs the planes into
                // two xors and an or3 -- 10 instead of 7 instructions)
                uint32_t d = __builtin_amdgcn_bitop3_b32(nvi, a0, b[u][0], 0xF6);
                d = __builtin_amdgcn_bitop3_b32(d, a1, b[u][1], 0xF6);
                d = __builtin_amdgcn_bitop3_b32(d, a2, b[u][2], 0xF6);
                d = __builtin_amdgcn_bitop3_b32(d, a3, b[u][3], 0xF6);
                d = __builtin_amdgcn_bitop3_b32(d, a4, b[u][4], 0xF6);
                d = __builtin_amdgcn_bitop3_b32(d, a5, b[u][5], 0xF6);
                d = __builtin_amdgcn_bitop3_b32(d, a6, b[u][6], 0xF6);
                hit[u][t] += __builtin_popcount(d);
                dst[u][t] += __builtin_popcount(vi | b[u][7]);
            }
        }
    }
}

template <int TI, int TJ>
__global__ __launch_bounds__(64) void pair_counts_kernel(const uint32_t *__restrict__ planes, int nchunk, int m_pad,
                                                         int m, int ldw, uint32_t *__restrict__ hit_out,
                                                         uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                                         float *__restrict__ wmat, float *__restrict__ wlow,
                                                         int *__restrict__ undef_flag, int n_iblocks) {
    const int lane = threadIdx.x;
    int ib, jb;
    pair_tile<TI, TJ>(n_iblocks, (int)blockIdx.x, ib, jb);
    const int i0 = ib * TI;  // uniform
    const int j0 = jb * (64 * TJ);
    if (j0 >= m_pad) return;
    if (j0 + 64 * TJ - 1 <= i0) return;  // tile holds no pair with j > i: its mirror tile writes both halves
    uint32_t hit[TJ][TI], dst[TJ][TI];
    pair_loop_plain<TI, TJ>(planes, nchunk, m_pad, i0, j0, lane, hit, dst);
    pair_epilogue<TI, TJ>(hit, dst, i0, j0, lane, nchunk, m, ldw, hit_out, dst_out, ident, wmat, wlow, undef_flag);
}

template <int TJ>
__global__ __launch_bounds__(64 * PAIR_KMAX) void pair_counts_pipe_kernel(const uint32_t *__restrict__ planes, int nchunk, int m_pad,
                                                              int m, int ldw, uint32_t *__restrict__ hit_out,
                                                              uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                                              float *__restrict__ wmat, float *__restrict__ wlow,
                                                              int *__restrict__ undef_flag, int n_iblocks, uint32_t *__restrict__ wsum) {
    int ib, jb;
    pair_tile<PAIR_TI, TJ>(n_iblocks, (int)blockIdx.x, ib, jb);
    pair_counts_pipe_body<TJ>(planes, nchunk, m_pad, m, ldw, hit_out, dst_out, ident, wmat, wlow, undef_flag, ib, jb, wsum);
}
__global__ __launch_bounds__(64 * PAIR_KMAX) void pair_counts_pipe16_kernel(const uint32_t *__restrict__ planes, int nchunk, int m_pad, int m, int ldw,
                                                                            float *__restrict__ ident, float *__restrict__ wmat,
                                                                            float *__restrict__ wlow, int *__restrict__ undef_flag, int n_iblocks,
                                                                            uint32_t *__restrict__ wsum) {
    int ib, jb;
    pair_tile<16, 1>(n_iblocks, (int)blockIdx.x, ib, jb);
    pair_counts_pipe16_body(planes, nchunk, m_pad, m, ldw, ident, wmat, wlow, undef_flag, ib, jb, wsum);
}
// (a batch holds alignments of the one-row-per-lane regime only -- below ~4100 sequences: pair_tiles_pipe)
__global__ __launch_bounds__(64 * PAIR_KMAX) void pair_counts_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    int ib, jb;
    pair_tile<PAIR_TI, 1>((d.m + PAIR_TI - 1) / PAIR_TI, local, ib, jb);
    pair_counts_pipe_body<1>(d.planes, d.nchunk, d.m_pad, d.m, d.ldw, nullptr, nullptr, d.ident, d.w, d.wlow, d.flags + 1, ib, jb);
}

// (the identity row statistics -- selectMethod's sequential float32 sums -- live in msastat_simx.hip: they are
// evaluated with the same binade-exact block test as the ordered rows of the similarity kernel)

__global__ __launch_bounds__(256) void sim_finish_kernel(const float *__restrict__ num, const float *__restrict__ den,
                                                         const int32_t *__restrict__ gaps_w, int m, int n,
                                                         float *__restrict__ q_out, float *__restrict__ mdk_out, int all_on_host) {
    sim_finish_body(num, den, gaps_w, m, n, q_out, mdk_out, all_on_host, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void sim_finish_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K,
                                                               int all_on_host) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    if (d.gated && d.flags[6]) return;  // (selectMethod took gappyout on the device: the similarity values are not used)
    sim_finish_body(d.simnum, d.simden, d.gaps, d.m, d.n, d.mdk + d.n, d.mdk, all_on_host, local);
}

// ------------------------------------------------------------------------------------------
// overlap: Cleaner::calculateSpuriousVector.  For residue x of row i in column c the hit count
// over the other rows is   valid(x) ? nvalid_c - 1 : (count of x in c) - 1,   so the O(n m^2)
// loop collapses to per-column counts (gap_counts) + one masked popcount pass over the planes.
// col_ok[3][nchunk] bit masks: column is "good" for a valid / gap / indetermination residue.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void overlap_colmask_kernel(const int32_t *__restrict__ gaps,
                                                              const int32_t *__restrict__ indets, int m, int n,
                                                              int need, uint32_t *__restrict__ col_ok, int nchunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    bool okv = false, okg = false, oki = false;
    if (c < n) {
        const int g = gaps[c], x = indets[c], v = m - g - x;
        okv = (v - 1) >= need;
        okg = (g - 1) >= need;
        oki = (x - 1) >= need;
    }
    // one 64-bit ballot per wave -> two chunk words
    const unsigned long long bv = __ballot(okv), bg = __ballot(okg), bi = __ballot(oki);
    const int lane = threadIdx.x & 63;
    const int chunk0 = (blockIdx.x * 256 + (threadIdx.x & ~63)) >> 5;
    if (lane < 2 && chunk0 + lane < nchunk) {
        col_ok[chunk0 + lane] = (uint32_t)(bv >> (32 * lane));
        col_ok[nchunk + chunk0 + lane] = (uint32_t)(bg >> (32 * lane));
        col_ok[2 * nchunk + chunk0 + lane] = (uint32_t)(bi >> (32 * lane));
    }
}

__global__ __launch_bounds__(256) void overlap_rows_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                           uint32_t indet4, const uint32_t *__restrict__ col_ok,
                                                           int nchunk, int32_t *__restrict__ good) {
    // one wave per row, lanes sweep the row 4 bytes at a time; shuffle-reduce the count
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m) return;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(raw + (size_t)row * ld);
    int cnt = 0;
    for (int c4 = lane; c4 * 4 < n; c4 += 64) {
        const uint32_t x = p[c4];
        const uint32_t isg = zero_bytes(x ^ 0x2d2d2d2du), isi = zero_bytes(x ^ indet4);
        const int chunk = c4 >> 3, sh = (c4 & 7) * 4;
        const uint32_t okv = (col_ok[chunk] >> sh) & 0xFu, okg = (col_ok[nchunk + chunk] >> sh) & 0xFu,
                       oki = (col_ok[2 * nchunk + chunk] >> sh) & 0xFu;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c4 * 4 + k < n) {
                const bool g = (isg >> (8 * k + 7)) & 1u, i = (isi >> (8 * k + 7)) & 1u;
                const uint32_t ok = g ? okg : (i ? oki : okv);
                cnt += (ok >> k) & 1u;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 0) good[row] = cnt;
}

// ------------------------------------------------------------------------------------------
// masked non-gap counts for Cleaner::removeAllGapsSeqsAndCols
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_nongap_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                         const uint8_t *__restrict__ keep_res,
                                                         int32_t *__restrict__ row_nongap) {
    row_nongap_body(raw, m, n, ld, keep_res, row_nongap, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void row_nongap_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    row_nongap_body(d.raw, d.m, d.n, d.ld, nullptr, d.rowtot, local);
}

__global__ __launch_bounds__(256) void col_nongap_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                         const uint8_t *__restrict__ keep_seq,
                                                         int32_t *__restrict__ col_nongap) {
    // as gap_counts: one thread = 4 adjacent columns (a dword per row), byte counters over a slab of 64 rows, one
    // atomic per column and slab
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if ((int64_t)c4 * 4 >= ld) return;
    const int r0 = blockIdx.y * 64, r1 = min(m, r0 + 64);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(raw + (size_t)r0 * ld) + c4;
    const size_t stride = (size_t)(ld >> 2);
    uint32_t acc = 0;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        const uint32_t x = *p;
        p += stride;
        const uint32_t nongap = (~zero_bytes(x ^ 0x2d2d2d2du) & 0x80808080u) >> 7;
        acc += keep_seq[r] ? nongap : 0u;  // (wave-uniform)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
        const uint32_t v = (acc >> (8 * k)) & 0xFFu;
        if (c < n && v) atomicAdd(&col_nongap[c], (int)v);
    }
}

// per-row ungapped length + 2x64-bit row hash (duplicate detection, representative ordering)
__device__ __forceinline__ void row_digest_body(const uint8_t *__restrict__ raw, int m, int n, int64_t ld, int32_t *__restrict__ lengths,
                                                unsigned long long *__restrict__ hashes, int bx) {
    const int row = bx * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m) return;
    const uint8_t *p = raw + (size_t)row * ld;
    int cnt = 0;
    unsigned long long h1 = 0, h2 = 0;
    for (int c = lane; c < n; c += 64) {
        const unsigned long long x = p[c];
        cnt += (x != '-') ? 1 : 0;
        // position-keyed mixing (splitmix-style), summed => order independent across lanes
        unsigned long long z = (x + 1) * 0x9E3779B97F4A7C15ull + (unsigned long long)c * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        h1 += z;
        unsigned long long y = (x + 7) * 0xD6E8FEB86659FD93ull ^ ((unsigned long long)c + 1) * 0xCA5A826395121157ull;
        y ^= y >> 32; y *= 0xFF51AFD7ED558CCDull; y ^= y >> 33;
        h2 += y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        h1 += __shfl_down(h1, off, 64);
        h2 += __shfl_down(h2, off, 64);
    }
    if (lane == 0) {
        lengths[row] = cnt;
        hashes[2 * row] = h1;
        hashes[2 * row + 1] = h2;
    }
}
__global__ __launch_bounds__(256) void row_digest_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                         int32_t *__restrict__ lengths,
                                                         unsigned long long *__restrict__ hashes) {
    row_digest_body(raw, m, n, ld, lengths, hashes, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void row_digest_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    const int mpad = (d.m + 64 + 63) / 64 * 64;  // (the engine's row padding: lengths [mpad], then the hashes, 8-byte aligned)
    row_digest_body(d.raw, d.m, d.n, d.ld, d.extra, reinterpret_cast<unsigned long long *>(d.extra + mpad), local);
}
// OverlapTrimmer in a batch: Cleaner::calculateSpuriousVector's closed form, a wave per sequence, over the group's own gap and
// indetermination counts (overlap_small_kernel's walk: four columns per lane and load)
__global__ __launch_bounds__(256) void overlap_rows_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    const int row = local * 4 + (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int m = d.m, n = d.n, need = d.ov_need;
    if (row >= m) return;
    typedef int i4 __attribute__((ext_vector_type(4)));
    const uint32_t *p = reinterpret_cast<const uint32_t *>(d.raw + (size_t)row * d.ld);
    int cnt = 0;
    for (int c4 = lane; c4 * 4 < n; c4 += 64) {
        const uint32_t x = p[c4];
        const i4 g4 = *reinterpret_cast<const i4 *>(d.gaps + 4 * c4), x4 = *reinterpret_cast<const i4 *>(d.indets + 4 * c4);
        const uint32_t isg = zero_bytes(x ^ 0x2d2d2d2du), isi = zero_bytes(x ^ d.indet4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c4 * 4 + k < n) {
                const int g = g4[k], xi = x4[k];
                const bool gap = (isg >> (8 * k + 7)) & 1u, ind = (isi >> (8 * k + 7)) & 1u;
                const int agree = gap ? g : (ind ? xi : m - g - xi);  // sequences that hold the same kind of symbol, this one included
                cnt += (agree - 1) >= need ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 0) d.extra[row] = cnt;
}

// exact row equality for candidate pairs (hash-equal rows)
__global__ __launch_bounds__(64) void rows_equal_kernel(const uint8_t *__restrict__ raw, int n, int64_t ld,
                                                        const int32_t *__restrict__ pairs, int npairs,
                                                        int32_t *__restrict__ equal) {
    const int pidx = blockIdx.x;
    if (pidx >= npairs) return;
    const uint8_t *a = raw + (size_t)pairs[2 * pidx] * ld, *b = raw + (size_t)pairs[2 * pidx + 1] * ld;
    int diff = 0;
    for (int c = threadIdx.x; c < n; c += 64) diff |= (a[c] != b[c]);
    const unsigned long long any = __ballot(diff);
    if (threadIdx.x == 0) equal[pidx] = any ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Representative sequences on the device (Cleaner::calculateRepresentativeSeq with a fixed
// identity threshold).  The reference walks the sequences from the longest down and makes one a
// new representative iff no EARLIER representative has identity > thr with it: that is the
// lexicographically-first maximal independent set of the graph {identity > thr} in processing
// order, which is unique, so it can be computed by rounds instead of one-by-one:
//   a vertex becomes REP once all its earlier neighbours are decided non-representatives,
//   and NON-REP as soon as one earlier neighbour is a REP.
// Each round reads a snapshot of the two bit sets and writes the next one (no torn reads).
// adjacency bits: bit u of row t  <=>  ident[seq_at[t]][seq_at[u]] > thr, only u < t kept.
// ------------------------------------------------------------------------------------------
// adjacency words are stored WORD-major (adj[w * m + t]: word w of vertex t), so that the threads of the clustering
// kernel -- one vertex each, all walking the words in step -- read consecutive addresses.
__global__ __launch_bounds__(256) void cluster_adjacency_kernel(const float *__restrict__ ident, int ldw,
                                                                const int32_t *__restrict__ seq_at,
                                                                const int32_t *__restrict__ pos_of, int m, float thr,
                                                                uint32_t *__restrict__ adj, int words,
                                                                uint32_t *__restrict__ nz) {
    // A thread owns the SEQUENCE col (its vertex is t = pos_of[col], the inverse of seq_at): the 256 threads of a block
    // then read 256 consecutive floats of row seq_at[u] (ident is symmetric) -- with a thread per processing index
    // the same reads were a gather inside the row, 16 times the bytes (99 -> 23 us at m = 5000).  The word is stored
    // at the vertex's place, a scattered 4-byte store per thread.
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int w = blockIdx.y;  // word of earlier vertices
    if (col >= m) return;
    const int t = pos_of[col];
    uint32_t bits = 0;
    if (w * 32 < t) {  // (only earlier vertices u < t are kept)
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            const int u = w * 32 + b;
            if (u < t) bits |= (ident[(size_t)seq_at[u] * ldw + col] > thr ? 1u : 0u) << b;
        }
    }
    adj[(size_t)w * m + t] = bits;
    // which words of vertex t hold a bit at all (bit w % 32 of nz[(w / 32) * m + t], zeroed by the launcher): the
    // clustering kernel walks those instead of every word -- its loads depend on one another, and with few or no
    // pairs above the threshold nearly every word is empty
    if (bits) atomicOr(&nz[(size_t)(w >> 5) * m + t], 1u << (w & 31));
}

__global__ __launch_bounds__(1024) void cluster_mis_kernel(const uint32_t *__restrict__ adj, const uint32_t *__restrict__ nz,
                                                           int m, int words, const int32_t *__restrict__ seq_at,
                                                           uint8_t *__restrict__ keep_seq, int32_t *__restrict__ count,
                                                           uint8_t *__restrict__ h_keep) {
    extern __shared__ uint32_t sets[];  // rep[2][words], undec[2][words]
    uint32_t *rep = sets, *undec = sets + 2 * words;
    __shared__ int remaining;
    for (int w = threadIdx.x; w < 2 * words; w += 1024) rep[w] = 0;
    for (int w = threadIdx.x; w < words; w += 1024) {
        const int base = w * 32;
        const uint32_t u = (base + 32 <= m) ? 0xFFFFFFFFu : (base < m ? ((1u << (m - base)) - 1u) : 0u);
        undec[w] = u;
        undec[words + w] = u;
    }
    __syncthreads();
    int cur = 0;
    for (int round = 0; round <= m; ++round) {
        const uint32_t *rc = rep + cur * words, *uc = undec + cur * words;
        uint32_t *rn = rep + (cur ^ 1) * words, *un = undec + (cur ^ 1) * words;
        if (threadIdx.x == 0) remaining = 0;
        for (int w = threadIdx.x; w < words; w += 1024) {  // next = current, then decisions are OR/AND-ed in
            rn[w] = rc[w];
            un[w] = uc[w];
        }
        __syncthreads();
        int mine = 0;
        for (int t = threadIdx.x; t < m; t += 1024) {
            if (!((uc[t >> 5] >> (t & 31)) & 1u)) continue;
            int verdict = 1;  // 1 = REP, 0 = NON-REP, -1 = blocked by an undecided earlier neighbour
            // (the adjacency kernel keeps bits of earlier vertices only: every marked word is <= t / 32)
            for (int g = 0; g <= (t >> 10) && verdict; ++g) {
                uint32_t marked = nz[(size_t)g * m + t];
                while (marked) {
                    const int w = g * 32 + __builtin_ctz(marked);
                    marked &= marked - 1;
                    const uint32_t a = adj[(size_t)w * m + t];
                    if (a & rc[w]) { verdict = 0; break; }
                    if (a & uc[w]) verdict = -1;
                }
            }
            if (verdict == 1) atomicOr(&rn[t >> 5], 1u << (t & 31));
            if (verdict >= 0) atomicAnd(&un[t >> 5], ~(1u << (t & 31)));
            else mine = 1;
        }
        if (mine) remaining = 1;
        __syncthreads();
        cur ^= 1;
        if (!remaining) break;
        __syncthreads();
    }
    const uint32_t *rf = rep + cur * words;
    int local = 0;
    for (int t = threadIdx.x; t < m; t += 1024) {
        const int r = (rf[t >> 5] >> (t & 31)) & 1u;
        keep_seq[seq_at[t]] = (uint8_t)r;
        if (h_keep) h_keep[seq_at[t]] = (uint8_t)r;  // (the mask in pinned host memory as well: no copy behind the kernel)
        local += r;
    }
    if (local && count) atomicAdd(count, local);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline uint32_t rep4(uint8_t b) { return 0x01010101u * b; }

// ---- diagnostic switches ---------------------------------------------------------------------
static thread_local const Tuning *tl_tuning = nullptr;
void set_tuning(const Tuning *t) { tl_tuning = t; }
const Tuning *current_tuning() { return tl_tuning; }
const Tuning &tuning() {
    static const Tuning defaults;
    return tl_tuning ? *tl_tuning : defaults;
}
LaunchNote &launch_note() {
    static thread_local LaunchNote note;
    return note;
}
// MSA_DIAGNOSTICS gates every switch but MSA_TRACE: a process that does not set it gets the default dispatch whatever else its
// environment holds (the switches exist for the parity suite -- which runs every path of THIS binary, the one that ships -- and for the
// A/B tools; none is needed in production, and a stray MSA_SIM_KERNEL=seq would cost a factor of ten).  -DMSA_NO_DIAGNOSTICS (make
// DIAGNOSTICS=0) compiles the reading out altogether.
bool diagnostics_enabled() {
#ifdef MSA_NO_DIAGNOSTICS
    return false;
#else
    return getenv("MSA_DIAGNOSTICS") != nullptr;
#endif
}
Tuning tuning_from_env() {
    Tuning t;
    t.trace = getenv("MSA_TRACE") != nullptr;
    if (!diagnostics_enabled()) return t;
    auto num = [](const char *name, int dflt) {
        const char *e = getenv(name);
        return e ? atoi(e) : dflt;
    };
    if (const char *k = getenv("MSA_SIM_KERNEL")) t.sim_kernel = k[0] == 's' ? 1 : 0;  // "seq": the plain sequential cross-check kernel
    t.sim_mode = num("MSA_SIM_MODE", 0);
    t.device_clusters = num("MSA_DEVICE_CLUSTERS", -1);
    t.pipeline = num("MSA_PIPELINE", 1);
    t.upload_direct = num("MSA_UPLOAD_DIRECT", 1);
    t.lg_r0 = num("MSA_LG_R0", -1);
    t.lg_big = num("MSA_LG_BIG", 0);
    t.mdk_host = num("MSA_MDK_HOST", 0);
    t.compact = num("MSA_COMPACT", 1);
    t.zerocopy_kb = num("MSA_ZEROCOPY_KB", 96);
    t.flat_max_m = num("MSA_FLAT_MAX_M", 128);
    t.flat_u = num("MSA_FLAT_U", 0);
    t.lg_rounds = num("MSA_LG_ROUNDS", -1);
    t.lg_split = num("MSA_LG_SPLIT", 0);
    t.front_cw = num("MSA_FRONT_CW", 0);
    t.front_nt = num("MSA_FRONT_NT", 0);
    t.front_from_m = num("MSA_FRONT_FROM_M", 0);
    t.front_xcd = num("MSA_FRONT_XCD", 1);
    t.pair_ti = num("MSA_PAIR_TI", 0);
    t.pair_k = num("MSA_PAIR_K", 0);
    t.lists_fused = num("MSA_LISTS_FUSED", 1);
    t.lg_halves = num("MSA_LG_HALVES", 1);
    t.lg_pipe = num("MSA_LG_PIPE", 1);
    t.lg_pipe_k = num("MSA_LG_PIPE_K", 1);
    t.lg_xseg = num("MSA_LG_XSEG", 1);
    t.lg_xseg_kx = num("MSA_LG_XSEG_KX", 0);
    t.lg_parts = num("MSA_LG_PARTS", 2);
    return t;
}
int set_max_lds_once(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> done;  // (kernel, device) -> bytes granted
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({kernel, dev});
    if (it != done.end() && it->second >= bytes) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done[{kernel, dev}] = bytes;
    return 0;
}

void launch_prep_planes(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, uint32_t *planes,
                        int nchunk, int m_pad, int *err_flag) {
    dim3 grid((nchunk + 1) / 2, (m_pad + 255) / 256);
    prep_planes_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, rep4(indet), planes, nchunk, m_pad, err_flag);
}
int planes_total() { return PLANES_TOTAL; }

void launch_gap_counts(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, int32_t *gaps,
                       int32_t *indets) {
    dim3 grid((unsigned)((ld / 4 + 255) / 256), (m + GAP_SLAB - 1) / GAP_SLAB);
    gap_counts_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, rep4(indet), gaps, indets);
}

// One loop per row-count regime (profiles/r02_pairs_time.jsonl, r02_ab_switches.txt):
//   * below ~4100 sequences the kernel is short of waves (three to four per SIMD at m = 2000): one row j per lane, the
//     software-pipelined loop (0.46 -> 0.37 ms at 2000 x 10000, 0.088 -> 0.074 ms at 1000 x 4000);
//   * from there on two rows j per lane reuse the wave-uniform words of the rows i twice, and the plain loop is 4 %
//     faster than the pipelined one (twice the registers).
// Both on a one-dimensional grid over the tiles that hold pairs (j > i).  (A third loop on dense residue codes -- 8
// instead of 11 VALU instructions per pair and word -- was built in round 2 and removed in round 3: collecting the byte
// values and writing two more sets of planes ate what it gained, 3.22 -> 3.19 ms per trim at best.)
// waves per tile (pair_counts_pipe_body): from ~8 waves per SIMD on (8192 tiles) a wave per tile is best -- the grid refills the
// slots as tiles finish; below, up to eight waves per tile, at least eight chunks each (measured at 2000 x 10000, 4000 tiles:
// 0.350 / 0.349 / 0.320 / 0.313 ms with 1 / 2 / 4 / 8 waves; 1000 x 4000: 0.074 -> 0.040; 500 x 2000: 0.033 -> 0.016; 3000 x 8000,
// 8800 tiles: 0.514 with one wave, 0.530 with two)
#ifndef MSA_PAIR16_FROM_M  // sequences from which on the pair pass runs sixteen rows i per tile (up to the two-rows regime)
#define MSA_PAIR16_FROM_M 513
#endif
static int pair_split(long tiles, int nchunk) {
    int k = 1;
    if (tiles >= 8192) return k;
    while (2 * k <= PAIR_KMAX && tiles * (2 * k) <= 32768 && nchunk >= 16 * k) k *= 2;
    return k;
}

void launch_pair_counts(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int ldw, uint32_t *hit,
                        uint32_t *dst, float *ident, float *wmat, float *wlow, int *undef_flag, uint32_t *wsum) {
    const long waves2 = (long)((m + PAIR_TI - 1) / PAIR_TI) * (m_pad / 128) / 2;
    const bool two = waves2 >= 8192;  // m_pad is a multiple of 128
    const int nib = (m + PAIR_TI - 1) / PAIR_TI, njb = two ? m_pad / 128 : m_pad / 64;
    const int R = 64 * (two ? 2 : 1) / PAIR_TI, jc = (nib + R - 1) / R - 1;
    const unsigned tiles = (unsigned)(R * jc * (jc + 1) / 2 + (njb - jc) * nib);
    launch_note().pair_kind = two ? 2 : 1;
    launch_note().pair_waves = 1;
    // Sixteen rows i per tile (two halves of eight on the same j planes: pair_counts_pipe16_body) from 513 sequences up to the
    // two-rows regime: the j planes -- seven eighths of what a tile reads -- are loaded half as often per pair.  As fast as eight rows
    // alone (profiles/r06_front_pairs_ab.txt: 1000 x 4000 43.3 / 42.4 us, 2000 x 10000 0.343 / 0.349 ms), half the traffic beyond the L2
    // (r06_pmc_hbm_traffic.txt), and 0.25 ms off the C5 batch, where the pair pass runs beside other contexts' similarity kernels
    // (r06_c5_front_pairs_ab.txt).  Below 513 sequences it loses (150 ... 500 rows: 13 -> 18, 19 -> 22 us: too few tiles).  K waves per
    // tile while the launch stays within twice the chip's wave slots (2000 x 10000: K = 8 0.388 ms, K = 4 0.343).  MSA_PAIR_TI / MSA_PAIR_K: A/B.
    const int ti_forced = tuning().pair_ti;
    if (!two && !hit && !dst && (ti_forced == 16 || (ti_forced == 0 && m >= MSA_PAIR16_FROM_M)) &&
        (uint64_t)planes_total() * (uint64_t)nchunk * (uint64_t)m_pad * 4u < (1ull << 32)) {  // (its plane addresses are 32-bit offsets)
        const int nib16 = (m + 15) / 16, jc16 = (nib16 + 3) / 4 - 1;
        const unsigned tiles16 = (unsigned)(4 * jc16 * (jc16 + 1) / 2 + (m_pad / 64 - jc16) * nib16);
        int K = 1;
        while (2 * K <= PAIR_KMAX && (long)tiles16 * (2 * K) <= 10240 && nchunk >= 8 * (2 * K)) K *= 2;
        if (tuning().pair_k > 0) K = std::min(PAIR_KMAX, tuning().pair_k);
        launch_note().pair_kind = 3;
        launch_note().pair_waves = K;
        pair_counts_pipe16_kernel<<<tiles16, 64 * K, 0, s>>>(planes, nchunk, m_pad, m, ldw, ident, wmat, wlow, undef_flag, nib16, wsum);
        return;
    }
    if (two) pair_counts_kernel<PAIR_TI, 2><<<tiles, 64, 0, s>>>(planes, nchunk, m_pad, m, ldw, hit, dst, ident, wmat, wlow, undef_flag, nib);
    else {
        int K = pair_split(tiles, nchunk);
        if (tuning().pair_k > 0) K = std::min(PAIR_KMAX, tuning().pair_k);
        launch_note().pair_waves = K;
        pair_counts_pipe_kernel<1><<<tiles, 64 * K, (size_t)(K - 1) * 2 * PAIR_TI * 64 * sizeof(uint32_t), s>>>(planes, nchunk, m_pad, m, ldw, hit, dst, ident, wmat,
                                                                                                                wlow, undef_flag, nib, wsum);
    }
}

// tiles of the pair pass in its one-row-per-lane regime (launch_pair_counts; what a batch uses for every alignment)
int pair_tiles_pipe(int m, int m_pad) {
    const int nib = (m + PAIR_TI - 1) / PAIR_TI, njb = m_pad / 64;
    const int R = 64 / PAIR_TI, jc = (nib + R - 1) / R - 1;
    return R * jc * (jc + 1) / 2 + (njb - jc) * nib;
}
bool pair_pipe_regime(int m, int m_pad) { return (long)((m + PAIR_TI - 1) / PAIR_TI) * (m_pad / 128) / 2 < 8192; }

void launch_fetch_rows_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) fetch_rows_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_gap_counts_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) gap_counts_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_row_nongap_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) row_nongap_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_prep_planes_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) prep_planes_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_pair_counts_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks, int min_nchunk) {
    const int W = pair_split(blocks, min_nchunk);  // (one value for the launch: by the tiles of the whole group)
    if (blocks > 0) pair_counts_batch_kernel<<<blocks, 64 * W, (size_t)(W - 1) * 2 * PAIR_TI * 64 * sizeof(uint32_t), s>>>(table, prefix, K);
}
void launch_overlap_rows_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) overlap_rows_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_row_digest_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) row_digest_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_sim_finish_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) sim_finish_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K, tuning().mdk_host);
}

void launch_sim_finish(hipStream_t s, const float *num, const float *den, const int32_t *gaps_w, int m, int n,
                       float *q_out, float *mdk_out) {
    sim_finish_kernel<<<(n + 255) / 256, 256, 0, s>>>(num, den, gaps_w, m, n, q_out, mdk_out, tuning().mdk_host);
}

// OverlapTrimmer's decision on the device as well (the host takes it from the same counts after the wait and compares): a sequence
// stays when (float)good / n is not below the threshold -- the mask the all-gap column counts behind it are taken over
__global__ __launch_bounds__(256) void overlap_keep_kernel(const int32_t *__restrict__ good, int m, int n, float min_ov, uint8_t *__restrict__ keep) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < m) keep[i] = (static_cast<float>(good[i]) / n) < min_ov ? 0 : 1;
}
void launch_overlap_keep(hipStream_t s, const int32_t *good, int m, int n, float min_ov, uint8_t *keep) {
    overlap_keep_kernel<<<(m + 255) / 256, 256, 0, s>>>(good, m, n, min_ov, keep);
}

void launch_overlap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                    const int32_t *indets, int need, uint32_t *col_ok, int nchunk, int32_t *good) {
    overlap_colmask_kernel<<<(nchunk * 32 + 255) / 256, 256, 0, s>>>(gaps, indets, m, n, need, col_ok, nchunk);
    overlap_rows_kernel<<<(m + 3) / 4, 256, 0, s>>>(raw, m, n, ld, rep4(indet), col_ok, nchunk, good);
}

void launch_row_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_res,
                       int32_t *row_nongap) {
    row_nongap_kernel<<<(m + 3) / 4, 256, 0, s>>>(raw, m, n, ld, keep_res, row_nongap);
}

void launch_col_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_seq,
                       int32_t *col_nongap) {
    dim3 grid((int)((ld / 4 + 255) / 256), (m + 63) / 64);
    col_nongap_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, keep_seq, col_nongap);
}

void launch_row_digest(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, int32_t *lengths,
                       unsigned long long *hashes) {
    row_digest_kernel<<<(m + 3) / 4, 256, 0, s>>>(raw, m, n, ld, lengths, hashes);
}

size_t cluster_adj_words(int m) { return (size_t)((m + 31) / 32); }
// words of the adjacency buffer: the bit sets and, behind them, the map of their non-empty words
size_t cluster_adj_buffer_words(int m) { return (size_t)m * (cluster_adj_words(m) + (cluster_adj_words(m) + 31) / 32); }

// seq_at: the processing order [m] followed by its inverse [m] (position of every sequence in that order)
int launch_cluster(hipStream_t s, const float *ident, int ldw, const int32_t *seq_at, int m, float thr, uint32_t *adj,
                   uint8_t *keep_seq, int32_t *count, uint8_t *h_keep) {
    const int words = (int)cluster_adj_words(m);
    const size_t lds = (size_t)4 * words * sizeof(uint32_t);
    if (lds > 60 * 1024) return -1;  // caller falls back to the host path
    dim3 grid((m + 255) / 256, words);
    uint32_t *nz = adj + (size_t)m * words;
    if (hipMemsetAsync(nz, 0, (size_t)m * ((words + 31) / 32) * sizeof(uint32_t), s) != hipSuccess) return -2;
    cluster_adjacency_kernel<<<grid, 256, 0, s>>>(ident, ldw, seq_at, seq_at + m, m, thr, adj, words, nz);
    cluster_mis_kernel<<<1, 1024, lds, s>>>(adj, nz, m, words, seq_at, keep_seq, count, h_keep);
    return 0;
}

void launch_rows_equal(hipStream_t s, const uint8_t *raw, int n, int64_t ld, const int32_t *pairs, int npairs,
                       int32_t *equal) {
    if (npairs > 0) rows_equal_kernel<<<npairs, 64, 0, s>>>(raw, n, ld, pairs, npairs, equal);
}

}  // namespace msak
