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
//   codes8  [ceil(m/8)+1][ld]    8xu8 per column: table entries of 8 consecutive rows (numerator kernel);
//                                the single-chain kernel uses codes32, one byte offset per dword
//   tab     [29][32]             f32x2 {distance, both-valid} indexed by (row index, column index)
//
// No MFMA anywhere: this is integer / lookup / ordered-fp32 work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "msastat_kernels.h"

namespace msak {

// ------------------------------------------------------------------------------------------
// prep_planes: raw bytes -> bit-sliced planes.  One thread = one row x 64 columns (a full 64-B
// line of that row); lanes of a wave own consecutive rows so the plane stores coalesce.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gather_bit4(uint32_t x, int b) {
    // bit b of each of the 4 bytes of x -> 4 adjacent bits (byte 0 -> bit 0)
    return ((((x >> b) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {
    // 0x80 in every byte of v that is zero (exact, bytes < 0x80 or not)
    uint32_t t = (v & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | v | 0x7f7f7f7fu);
}

// Dense residue codes.  The pair pass compares symbols plane by plane; raw bytes have seven planes, but an alignment
// uses a few dozen symbols at most.  gap_counts records which byte values occur (`used`: 128 bits); the rank of a
// byte among the used values is its dense code, K = their number, and K itself is the code the rows "i" of the pair
// pass carry wherever they hold no residue -- a code no row "j" has, so such a column is a mismatch without the
// validity plane entering the comparison.  NP = 5 planes serve K <= 31 symbols (proteins with gaps, X, B, Z, ...),
// 6 planes K <= 63; beyond that (NP = 0) the pair pass works on the raw planes.
// Plane array: [0..6] raw symbol planes (written only when NP = 0), [7] validity, [8..13] dense codes of the rows as
// "j", [14..19] dense codes of the rows as "i".
constexpr int DENSE_J0 = 8, DENSE_I0 = 14, PLANES_TOTAL = 20;
__device__ __forceinline__ void dense_set(const uint32_t *__restrict__ used, uint32_t (&w)[4], int &K, int &NP, int force_raw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = used[i];
    w[1] |= 1u << ('-' - 32);  // (the columns behind n are stored as gaps)
    K = __builtin_popcount(w[0]) + __builtin_popcount(w[1]) + __builtin_popcount(w[2]) + __builtin_popcount(w[3]);
    NP = force_raw ? 0 : (K + 1 <= 32 ? 5 : (K + 1 <= 64 ? 6 : 0));
}

__global__ __launch_bounds__(256) void prep_planes_kernel(const uint8_t *__restrict__ raw, int m, int n,
                                                          int64_t ld, uint32_t indet4, uint32_t *__restrict__ planes,
                                                          int nchunk, int m_pad, int *__restrict__ err_flag,
                                                          const uint32_t *__restrict__ used_slots,
                                                          uint32_t *__restrict__ used_out, int force_raw) {
    __shared__ uint8_t lut[128];  // byte -> dense code
    const int row = blockIdx.x * 256 + threadIdx.x;  // < m_pad
    const int cpair = blockIdx.y;                    // 64-column group
    int K = 0, NP = 0;
    if (used_slots) {
        // fold gap_counts' copies of the set: word q of copy s sits at 4 s + q
        __shared__ uint32_t folded[2][4];
        if (threadIdx.x < 128) {
            uint32_t v = used_slots[threadIdx.x];
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) v |= __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) < 4) folded[threadIdx.x >> 6][threadIdx.x & 3] = v;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            folded[0][threadIdx.x] |= folded[1][threadIdx.x];
            if (blockIdx.x == 0 && blockIdx.y == 0) used_out[threadIdx.x] = folded[0][threadIdx.x];  // for the pair pass
        }
        __syncthreads();
        const uint32_t *used = folded[0];
        uint32_t w[4];
        dense_set(used, w, K, NP, force_raw);
        if (NP && threadIdx.x < 128) {
            const int k = threadIdx.x, q = k >> 5;
            int rank = __builtin_popcount(w[q] & ((1u << (k & 31)) - 1u));
            for (int i = 0; i < q; ++i) rank += __builtin_popcount(w[i]);
            lut[k] = (uint8_t)rank;
        }
        __syncthreads();
    }
    if (row >= m_pad) return;
    uint32_t out[2][8], outj[2][6], outi[2][6];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int p = 0; p < 8; ++p) out[h][p] = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) outj[h][p] = outi[h][p] = 0;
    }
    const uint32_t k4 = (uint32_t)K * 0x01010101u;
    if (row < m) {  // (the rows behind m stay zero in every plane: the pair pass computes them and writes nothing)
        const uint4 *src = reinterpret_cast<const uint4 *>(raw + (size_t)row * ld + (size_t)cpair * 64);
        uint32_t bad = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 v4 = src[q];
            uint32_t w[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = cpair * 64 + q * 16 + e * 4;  // first column of this dword
                uint32_t x = w[e];
                // mask columns >= n (undefined bytes) to '-' so they are invalid everywhere
                uint32_t keep = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) keep |= (col + k < n) ? (0xFFu << (8 * k)) : 0u;
                x = (x & keep) | (0x2d2d2d2du & ~keep);
                bad |= x & 0x80808080u;
                uint32_t inval = zero_bytes(x ^ 0x2d2d2d2du) | zero_bytes(x ^ indet4);  // 0x80 flags
                uint32_t vbits = gather_bit4(~inval, 7);
                const int h = q >> 1, sh = ((q & 1) * 4 + e) * 4;
                if (NP) {
                    const uint32_t cj = (uint32_t)lut[x & 127u] | ((uint32_t)lut[(x >> 8) & 127u] << 8) |
                                        ((uint32_t)lut[(x >> 16) & 127u] << 16) | ((uint32_t)lut[(x >> 24) & 127u] << 24);
                    const uint32_t im = (inval >> 7) * 0xFFu;  // 0xFF in the bytes that hold no residue
                    const uint32_t ci = (cj & ~im) | (k4 & im);
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        outj[h][p] |= gather_bit4(cj, p) << sh;
                        outi[h][p] |= gather_bit4(ci, p) << sh;
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 7; ++p) out[h][p] |= gather_bit4(x, p) << sh;
                }
                out[h][7] |= vbits << sh;
            }
        }
        if (bad) atomicOr(err_flag, 1);
    }
    const size_t pstride = (size_t)nchunk * m_pad;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int chunk = cpair * 2 + h;
        if (chunk < nchunk) {
            const size_t at = (size_t)chunk * m_pad + row;
            if (NP) {
                planes[7 * pstride + at] = out[h][7];
                for (int p = 0; p < NP; ++p) {
                    planes[(DENSE_J0 + p) * pstride + at] = outj[h][p];
                    planes[(DENSE_I0 + p) * pstride + at] = outi[h][p];
                }
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) planes[p * pstride + at] = out[h][p];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// gap_counts: per-column '-' and indetermination counts (statistics::Gaps::CalculateVectors).
// One thread = 4 adjacent columns (one dword per row, a wave reads 256 contiguous bytes per
// row), SWAR byte counters over a slab of <= 128 rows, then one integer atomic per column.
// HBM-bound: reads m*n bytes once.
// ------------------------------------------------------------------------------------------
constexpr int GAP_SLAB = 64;
constexpr int USED_SLOTS = 32;  // copies of the 128-bit set of byte values that gap_counts / row_nongap fill (see gap_counts)

// the set of byte values a thread has seen (see gap_counts)
struct ByteSet {
    uint32_t low = 0, w1 = 0, w2 = 0, w3 = 0;  // low: some byte below 0x20; w_s: bit (b & 31) for the bytes with bits 6..5 = s
};
// imask: 0xFF in the bytes of x that count
__device__ __forceinline__ void byteset_add(ByteSet &b, uint32_t x, uint32_t imask) {
    const uint32_t h = x >> 1;
    // bit 5 of a byte of y_s: bits 6..5 of that byte of x are s (binary 11 / 10 / 01)
    const uint32_t y3 = x & h & imask, y2 = ~x & h & imask, y1 = x & ~h & imask;
    b.low |= zero_bytes(x & 0x60606060u) & imask;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t bit = 1u << ((x >> (8 * k)) & 31u);
        // truth table 0xF8 = a | (b & c)
        b.w1 = __builtin_amdgcn_bitop3_b32(b.w1, bit, (uint32_t)__builtin_amdgcn_sbfe(y1, 8 * k + 5, 1), 0xF8);
        b.w2 = __builtin_amdgcn_bitop3_b32(b.w2, bit, (uint32_t)__builtin_amdgcn_sbfe(y2, 8 * k + 5, 1), 0xF8);
        b.w3 = __builtin_amdgcn_bitop3_b32(b.w3, bit, (uint32_t)__builtin_amdgcn_sbfe(y3, 8 * k + 5, 1), 0xF8);
    }
}
// Block union (256 threads, all of them call), then one atomic per word into one of USED_SLOTS copies of the set: all
// waves of a launch are resident at once, and with a single copy their atomics queue up on one cache line for tens of
// microseconds.  prep_planes folds the copies.
__device__ __forceinline__ void byteset_publish(const ByteSet &b, uint32_t *__restrict__ used, unsigned block) {
    __shared__ uint32_t red[4][4];
    uint32_t v[4] = {b.low ? 0xFFFFFFFFu : 0u, b.w1, b.w2, b.w3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] |= __shfl_xor(v[q], off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t u = red[0][threadIdx.x] | red[1][threadIdx.x] | red[2][threadIdx.x] | red[3][threadIdx.x];
        uint32_t *slot = used + 4 * (block % USED_SLOTS) + threadIdx.x;
        if (u & ~__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(slot, u);
    }
}

// `used` != nullptr: the set of byte values that occur in columns < n is OR-ed into used[0..3] (bit b of the 128 = byte
// value b & 127; bytes >= 0x80 are an error that prep_planes reports).  Every thread keeps the three set words of the
// printable values in registers: per byte one shift (1 << (b & 31)), one sign-extending bit-field extract per word
// (all ones when bits 6..5 of the byte name that word) and one three-input boolean; a byte below 0x20 marks all of
// word 0 (a superset is as good as the set: codes stay distinct).  Measured alternatives: one LDS byte flag per
// value (same-address stores serialise: 12 -> 42 us at 2000 x 10000), per-thread words in LDS updated with ds_or
// (64 -> 72 us).
__global__ __launch_bounds__(256) void gap_counts_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                         uint32_t indet4, int32_t *__restrict__ gaps,
                                                         int32_t *__restrict__ indets, uint32_t *__restrict__ used) {
    ByteSet seen;
    const int c4 = blockIdx.x * 256 + threadIdx.x;  // dword column
    const bool active = (int64_t)c4 * 4 < ld;
    if (active) {
        const int r0 = blockIdx.y * GAP_SLAB;
        const int r1 = min(m, r0 + GAP_SLAB);
        const uint32_t *p = reinterpret_cast<const uint32_t *>(raw + (size_t)r0 * ld) + c4;
        const size_t stride = (size_t)(ld >> 2);
        const int inside = used ? max(0, min(4, n - c4 * 4)) : 0;  // bytes of this dword column in front of column n
        uint32_t accg = 0, acci = 0;
        // (0xFF in the bytes in front of column n: the others must not enter the set)
        const uint32_t imask = inside >= 4 ? 0xFFFFFFFFu : (inside <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - inside))));
        auto take = [&](uint32_t x) {
            accg += zero_bytes(x ^ 0x2d2d2d2du) >> 7;
            acci += zero_bytes(x ^ indet4) >> 7;
            if (used) byteset_add(seen, x, imask);  // (uniform)
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
    if (used) byteset_publish(seen, used, blockIdx.x + blockIdx.y * gridDim.x);
}

// ------------------------------------------------------------------------------------------
// pair_counts: hit/dst of every sequence pair (Cleaner::calculateSeqIdentity ==
// Similarity::calculateMatrixIdentity integers).  One wave = TI rows "i" (wave-uniform, read
// through the scalar cache, used as SGPR operands) x 64*TJ rows "j" (TJ per lane; TJ = 1 in production:
// the kernel is short of waves, not of reuse); per 32
// columns and pair: 7 xor/or (v_or3 / v_bitop3 fuse most of them) + and-not + 2 bcnt + or.
// Integer work, any order is exact.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }

// tile of a block: (i-block, j-block).  n_iblocks > 0: one-dimensional grid over the tiles that hold pairs (j > i)
// only, j-block by j-block -- a two-dimensional grid launches as many tiles that return at once, and the waves that
// do the work end up unevenly spread over the SIMDs (every active wave is resident from the start: the fullest SIMD
// sets the time).  j-block y holds min(n_iblocks, (y + 1) R) tiles, R = 64 TJ / TI.
template <int TI, int TJ>
__device__ __forceinline__ void pair_tile(int n_iblocks, int &ib, int &jb) {
    ib = blockIdx.x;
    jb = blockIdx.y;
    if (n_iblocks > 0) {
        constexpr int R = 64 * TJ / TI;
        const int t = blockIdx.x;
        const int jc = (n_iblocks + R - 1) / R - 1;  // first j-block whose row of tiles is cut off at n_iblocks
        const int pc = R * jc * (jc + 1) / 2;
        if (t < pc) {
            jb = (int)((sqrtf(8.0f * (float)t / (float)R + 1.0f) - 1.0f) * 0.5f);
            while (R * jb * (jb + 1) / 2 > t) --jb;
            while (R * (jb + 1) * (jb + 2) / 2 <= t) ++jb;
            ib = t - R * jb * (jb + 1) / 2;
        } else {
            jb = jc + (t - pc) / n_iblocks;
            ib = (t - pc) % n_iblocks;
        }
    }
}

// epilogue of a tile: row-wise (coalesced along j) and mirrored (TI contiguous values per lane).  miss[][] counted the
// misses of all 32 * nchunk columns (the columns behind n are gaps in every row).
template <int TI, int TJ>
__device__ __forceinline__ void pair_epilogue(const uint32_t (&miss)[TJ][TI], const uint32_t (&dst)[TJ][TI], int i0, int j0, int lane,
                                              int nchunk, int m, int ldw, uint32_t *__restrict__ hit_out,
                                              uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                              float *__restrict__ wmat, float *__restrict__ wlow, int *__restrict__ undef_flag) {
#pragma unroll
    for (int u = 0; u < TJ; ++u) {
        const int j = j0 + 64 * u + lane;
        if (j >= m) continue;
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const int i = i0 + t;
            if (i >= m) break;
            const bool diag = (i == j);
            const uint32_t h = diag ? 0u : 32u * (uint32_t)nchunk - miss[u][t], d = diag ? 0u : dst[u][t];
            if (hit_out) {
                hit_out[(size_t)i * m + j] = h;
                hit_out[(size_t)j * m + i] = h;
            }
            if (dst_out) {
                dst_out[(size_t)i * m + j] = d;
                dst_out[(size_t)j * m + i] = d;
            }
            if (!diag && d == 0u && undef_flag) atomicOr(undef_flag, 1);  // no column holds a residue of either row
            if (ident || wmat) {
                const float r = d ? (float)h / (float)d : 0.0f;
                if (ident) {
                    const float v = diag ? 0.0f : r;
                    ident[(size_t)i * ldw + j] = v;
                    ident[(size_t)j * ldw + i] = v;
                }
                if (wmat && i != j) {  // strictly upper triangular: the similarity pass reads W[j][k], k > j
                    const float v = 1.0f - r;
                    if (i < j) wmat[(size_t)i * ldw + j] = v;
                    else wmat[(size_t)j * ldw + i] = v;
                    // the mirror image (strictly lower triangular) for the kernel whose lanes are the rows j
                    if (wlow) wlow[(size_t)(i < j ? j : i) * ldw + (i < j ? i : j)] = v;
                }
            }
        }
    }
}

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
    pair_tile<TI, TJ>(n_iblocks, ib, jb);
    const int i0 = ib * TI;  // uniform
    const int j0 = jb * (64 * TJ);
    if (j0 >= m_pad) return;
    if (j0 + 64 * TJ - 1 <= i0) return;  // tile holds no pair with j > i: its mirror tile writes both halves
    uint32_t hit[TJ][TI], dst[TJ][TI];
    pair_loop_plain<TI, TJ>(planes, nchunk, m_pad, i0, j0, lane, hit, dst);
    pair_epilogue<TI, TJ>(hit, dst, i0, j0, lane, nchunk, m, ldw, hit_out, dst_out, ident, wmat, wlow, undef_flag);
}

// The same tile with the loads software-pipelined (TI = 8).  The loop above leaves the schedule to the compiler: the
// eight plane words of the rows i arrive through scalar loads in four batches per chunk, each followed by a full
// wait (scalar loads return out of order: only lgkmcnt(0) is safe), and the j planes are requested at the top of the
// chunk that uses them -- at three to four waves per SIMD (m = 2000) the SIMDs idle half of the time.  Here:
//   * the j planes of chunk c + 1 are requested while chunk c is computed (two register sets, the loop unrolled by two);
//   * the i planes come in two groups of four planes (32 SGPRs each, as many as the plain loop uses): the validity
//     plane + planes 0..2, then planes 3..6; a chunk is computed in two phases of 5 VALU instructions per pair, and
//     each group is requested at the start of the phase BEFORE the one that uses it, right behind the wait for the
//     other group -- one phase of the wave (and of the SIMD's other waves) covers its latency.
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
template <int TJ>
__global__ __launch_bounds__(64) void pair_counts_pipe_kernel(const uint32_t *__restrict__ planes, int nchunk, int m_pad,
                                                              int m, int ldw, uint32_t *__restrict__ hit_out,
                                                              uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                                              float *__restrict__ wmat, float *__restrict__ wlow,
                                                              int *__restrict__ undef_flag, int n_iblocks) {
    constexpr int TI = 8;
    typedef const __attribute__((address_space(4))) u32x8 *c8;
    const int lane = threadIdx.x;
    int ib, jb;
    pair_tile<TI, TJ>(n_iblocks, ib, jb);
    const int i0 = ib * TI;  // uniform
    const int j0 = jb * (64 * TJ);
    if (j0 >= m_pad) return;
    if (j0 + 64 * TJ - 1 <= i0) return;
    uint32_t miss[TJ][TI], dst[TJ][TI], d[TJ][TI];
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
        for (int t = 0; t < TI; ++t) miss[u][t] = dst[u][t] = 0;
    const size_t ps = (size_t)nchunk * m_pad;
    const uint32_t *pj = planes + j0 + lane;
    const uint32_t *pi = planes + i0;  // 32-byte aligned (i0 % 8 == 0, m_pad % 128 == 0)
    struct Group {
        u32x8 p[4];
    };
    auto request_a = [&](Group &g, int c) {  // validity plane, planes 0..2
        const uint32_t *q = pi + (size_t)c * m_pad;
        g.p[0] = *(c8)(uint64_t)(q + 7 * ps);
        g.p[1] = *(c8)(uint64_t)(q);
        g.p[2] = *(c8)(uint64_t)(q + ps);
        g.p[3] = *(c8)(uint64_t)(q + 2 * ps);
    };
    auto request_b = [&](Group &g, int c) {  // planes 3..6
        const uint32_t *q = pi + (size_t)c * m_pad + 3 * ps;
#pragma unroll
        for (int p = 0; p < 4; ++p) g.p[p] = *(c8)(uint64_t)(q + p * ps);
    };
    // everything requested so far has arrived (nothing younger is in flight here).  `pin`: the results of the phase in
    // front of the wait pass through it, so that the optimiser cannot sink that phase behind the wait (which would
    // then follow its request at once)
    auto arrived = [&](Group &g, uint32_t (&pin)[TJ][TI]) {
        if constexpr (TJ == 1)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+s"(g.p[0]), "+s"(g.p[1]), "+s"(g.p[2]), "+s"(g.p[3]), "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]),
                           "+v"(pin[0][3]), "+v"(pin[0][4]), "+v"(pin[0][5]), "+v"(pin[0][6]), "+v"(pin[0][7]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+s"(g.p[0]), "+s"(g.p[1]), "+s"(g.p[2]), "+s"(g.p[3]), "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]),
                           "+v"(pin[0][3]), "+v"(pin[0][4]), "+v"(pin[0][5]), "+v"(pin[0][6]), "+v"(pin[0][7]), "+v"(pin[TJ - 1][0]),
                           "+v"(pin[TJ - 1][1]), "+v"(pin[TJ - 1][2]), "+v"(pin[TJ - 1][3]), "+v"(pin[TJ - 1][4]), "+v"(pin[TJ - 1][5]),
                           "+v"(pin[TJ - 1][6]), "+v"(pin[TJ - 1][7]));
    };
    auto request_j = [&](uint32_t (&b)[TJ][8], int c) {
        const size_t off = (size_t)c * m_pad;
#pragma unroll
        for (int u = 0; u < TJ; ++u)
#pragma unroll
            for (int p = 0; p < 8; ++p) b[u][p] = pj[off + p * ps + 64 * u];
    };
    Group ga, gb;
    uint32_t b0[TJ][8], b1[TJ][8];
    auto step = [&](int c, uint32_t (&b)[TJ][8], uint32_t (&bn)[TJ][8]) {
        arrived(ga, miss);  // group A of chunk c
        request_b(gb, c);
        request_j(bn, c + 1 < nchunk ? c + 1 : c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const uint32_t vi = ga.p[0][t], nvi = ~vi;  // columns in which row i holds no residue never count as hits
#pragma unroll
            for (int u = 0; u < TJ; ++u) {
                uint32_t x = __builtin_amdgcn_bitop3_b32(nvi, ga.p[1][t], b[u][0], 0xF6);  // x | (y ^ z)
                x = __builtin_amdgcn_bitop3_b32(x, ga.p[2][t], b[u][1], 0xF6);
                d[u][t] = __builtin_amdgcn_bitop3_b32(x, ga.p[3][t], b[u][2], 0xF6);
                dst[u][t] += __builtin_popcount(vi | b[u][7]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (the first phase must not sink below the wait: the wait would then follow its request at once)
        arrived(gb, d);  // group B of chunk c
        request_a(ga, c + 1 < nchunk ? c + 1 : c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TI; ++t)
#pragma unroll
            for (int u = 0; u < TJ; ++u) {
                uint32_t x = __builtin_amdgcn_bitop3_b32(d[u][t], gb.p[0][t], b[u][3], 0xF6);
                x = __builtin_amdgcn_bitop3_b32(x, gb.p[1][t], b[u][4], 0xF6);
                x = __builtin_amdgcn_bitop3_b32(x, gb.p[2][t], b[u][5], 0xF6);
                x = __builtin_amdgcn_bitop3_b32(x, gb.p[3][t], b[u][6], 0xF6);
                miss[u][t] += __builtin_popcount(x);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    request_a(ga, 0);
    request_j(b0, 0);
    int c = 0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (; c + 1 < nchunk; c += 2) {  // (no branch inside the body: one basic block, the order above is the order issued)
        step(c, b0, b1);
        step(c + 1, b1, b0);
    }
    if (c < nchunk) step(c, b0, b1);
    arrived(ga, miss);  // (the last request, a repeat of the last chunk, is not used)
    pair_epilogue<TI, TJ>(miss, dst, i0, j0, lane, nchunk, m, ldw, hit_out, dst_out, ident, wmat, wlow, undef_flag);
}

// The pair pass on dense codes (see prep_planes): NP symbol planes instead of seven and no validity term in the
// comparison -- per pair and 32 columns  xor + (NP - 1) bitop3 + bcnt  for the misses and  or + bcnt  for the columns
// that count: 8 VALU instructions at NP = 5 against 11 on the raw planes.  Same software pipeline as above: group A =
// validity + code planes 0, 1 of the rows i, group B = the remaining code planes; TI = 8 rows i x 64 rows j per wave,
// one-dimensional grid over the triangle's tiles.  The number of planes is only known on the device (`used`, filled by
// gap_counts): the kernel branches once; with more than 62 symbols in the alignment it runs the plain loop on the
// raw planes (prep_planes has then written those).
template <int NP>
__device__ __forceinline__ void pair_loop_dense(const uint32_t *__restrict__ planes, int nchunk, int m_pad, int i0, int j0, int lane,
                                                uint32_t (&miss)[1][8], uint32_t (&dst)[1][8]) {
    constexpr int TI = 8, NB = NP - 2;  // NB planes in group B
    typedef const __attribute__((address_space(4))) u32x8 *c8;
    uint32_t d[1][TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) miss[0][t] = dst[0][t] = 0;
    const size_t ps = (size_t)nchunk * m_pad;
    // (per-lane pointers: six 64-bit additions per chunk on the vector unit.  Wave-uniform bases with the lane as a
    // 32-bit offset would move them to the scalar unit, but the loop already uses every SGPR: 28 instead of 12
    // instructions per two chunks once the spills are counted)
    const uint32_t *pjv = planes + 7 * ps + j0 + lane;          // validity of the rows j
    const uint32_t *pj = planes + DENSE_J0 * ps + j0 + lane;    // their codes
    const uint32_t *piv = planes + 7 * ps + i0;                 // 32-byte aligned (i0 % 8 == 0, m_pad % 128 == 0)
    const uint32_t *pi = planes + DENSE_I0 * ps + i0;
    struct GroupA {
        u32x8 p[3];
    };
    struct GroupB {
        u32x8 p[NB];
    };
    auto request_a = [&](GroupA &g, int c) {
        const size_t off = (size_t)c * m_pad;
        g.p[0] = *(c8)(uint64_t)(piv + off);
        g.p[1] = *(c8)(uint64_t)(pi + off);
        g.p[2] = *(c8)(uint64_t)(pi + off + ps);
    };
    auto request_b = [&](GroupB &g, int c) {
        const uint32_t *q = pi + (size_t)c * m_pad + 2 * ps;
#pragma unroll
        for (int p = 0; p < NB; ++p) g.p[p] = *(c8)(uint64_t)(q + p * ps);
    };
    // (the results of the phase in front of a wait pass through it: see pair_counts_pipe_kernel)
    auto arrived_a = [&](GroupA &g, uint32_t (&pin)[1][TI]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+s"(g.p[0]), "+s"(g.p[1]), "+s"(g.p[2]), "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]), "+v"(pin[0][3]),
                       "+v"(pin[0][4]), "+v"(pin[0][5]), "+v"(pin[0][6]), "+v"(pin[0][7]));
    };
    auto arrived_b = [&](GroupB &g, uint32_t (&pin)[1][TI]) {
        if constexpr (NB == 3)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+s"(g.p[0]), "+s"(g.p[1]), "+s"(g.p[2]), "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]), "+v"(pin[0][3]),
                           "+v"(pin[0][4]), "+v"(pin[0][5]), "+v"(pin[0][6]), "+v"(pin[0][7]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+s"(g.p[0]), "+s"(g.p[1]), "+s"(g.p[2]), "+s"(g.p[NB - 1]), "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]),
                           "+v"(pin[0][3]), "+v"(pin[0][4]), "+v"(pin[0][5]), "+v"(pin[0][6]), "+v"(pin[0][7]));
    };
    auto request_j = [&](uint32_t (&b)[NP + 1], int c) {
        const size_t off = (size_t)c * m_pad;
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = pj[off + p * ps];
        b[NP] = pjv[off];
    };
    GroupA ga;
    GroupB gb;
    uint32_t b0[NP + 1], b1[NP + 1];
    auto step = [&](int c, uint32_t (&b)[NP + 1], uint32_t (&bn)[NP + 1]) {
        arrived_a(ga, miss);  // group A of chunk c
        request_b(gb, c);
        request_j(bn, c + 1 < nchunk ? c + 1 : c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const uint32_t x = ga.p[1][t] ^ b[0];
            d[0][t] = __builtin_amdgcn_bitop3_b32(x, ga.p[2][t], b[1], 0xF6);  // x | (y ^ z)
            dst[0][t] += __builtin_popcount(ga.p[0][t] | b[NP]);
        }
        __builtin_amdgcn_sched_barrier(0);
        arrived_b(gb, d);  // group B of chunk c
        request_a(ga, c + 1 < nchunk ? c + 1 : c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            uint32_t x = d[0][t];
#pragma unroll
            for (int p = 0; p < NB; ++p) x = __builtin_amdgcn_bitop3_b32(x, gb.p[p][t], b[2 + p], 0xF6);
            miss[0][t] += __builtin_popcount(x);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    request_a(ga, 0);
    request_j(b0, 0);
    int c = 0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (; c + 1 < nchunk; c += 2) {
        step(c, b0, b1);
        step(c + 1, b1, b0);
    }
    if (c < nchunk) step(c, b0, b1);
    arrived_a(ga, miss);  // (the last request, a repeat of the last chunk, is not used)
}

__global__ __launch_bounds__(64) void pair_counts_dense_kernel(const uint32_t *__restrict__ planes, int nchunk, int m_pad, int m,
                                                               int ldw, uint32_t *__restrict__ hit_out,
                                                               uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                                               float *__restrict__ wmat, float *__restrict__ wlow,
                                                               int *__restrict__ undef_flag, int n_iblocks,
                                                               const uint32_t *__restrict__ used) {
    constexpr int TI = 8;
    const int lane = threadIdx.x;
    int ib, jb;
    pair_tile<TI, 1>(n_iblocks, ib, jb);
    const int i0 = ib * TI, j0 = jb * 64;
    if (j0 >= m_pad) return;
    if (j0 + 63 <= i0) return;
    uint32_t w[4];
    int K, NP;
    dense_set(used, w, K, NP, 0);
    NP = __builtin_amdgcn_readfirstlane(NP);
    uint32_t miss[1][TI], dst[1][TI];
    if (NP == 5) pair_loop_dense<5>(planes, nchunk, m_pad, i0, j0, lane, miss, dst);
    else if (NP == 6) pair_loop_dense<6>(planes, nchunk, m_pad, i0, j0, lane, miss, dst);
    else pair_loop_plain<TI, 1>(planes, nchunk, m_pad, i0, j0, lane, miss, dst);
    pair_epilogue<TI, 1>(miss, dst, i0, j0, lane, nchunk, m, ldw, hit_out, dst_out, ident, wmat, wlow, undef_flag);
}

// (the identity row statistics -- selectMethod's sequential float32 sums -- live in msastat_simx.hip: they are
// evaluated with the same binade-exact block test as the ordered rows of the similarity kernel)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// similarity_mdk, single-chain producer/consumer form: both sums in one packed chain.  msa_similarity
// launches the numerator + denominator kernels further down; this one remains as the alternative
// selected by MSA_SIM_KERNEL=pc and is parity-tested at every size.
//
// The float32 sums of one column are a strictly sequential chain, so the only way to go faster
// than one-wave-does-everything is to strip the chain-carrying wave down to the chain itself.
// One workgroup = 64 columns = 1 consumer wave + NP producer waves (two waves per SIMD: a lone
// wave issues one instruction per ~4.75 cycles whatever its type, so instruction count per wave,
// not ALU width, is the currency here):
//   producers  walk the (j,k) pair sequence in units of 8 consecutive k ("octs"), look up
//              {D[a_j][a_k], valid} for their lane's column, multiply by the wave-uniform
//              W[j][k] (separate rounding), and store {x, w_eff} pairs into an LDS ring;
//   consumer   reads the ring in pair order and does ONE v_pk_add_f32 per step:
//              {num, den} += {x, w_eff}   -- bit-identical to the reference's two scalar adds
//              (a skipped pair contributes {+0, +0}, which leaves both sums unchanged).
// Each producer keeps a private, lane-interleaved copy of the table row of its current j
// ([entry][lane] x 8 B: every lane owns its bank pair, so the per-step gather is conflict-free).
// codes32 [ceil(m/8) + 1][2][ld] x 4 u32: entry * 512 + (column % tile) * 8, i.e. the byte offset
// into such a slice; entry `npos` is the all-zero entry of skipped residues; the extra last
// row is all-skipped and is what producers read once they run past the end of the sequence.
// W is strictly upper triangular, so the rows k <= j of a row's first oct need no masking.
// ------------------------------------------------------------------------------------------
// Rounds are row-aligned: a round (ROUND_OCTS octs) never spans two rows j -- the tail of a row
// is padded with null octs that read the all-skipped codes row -- so every producer is always on
// the same row, positions are a function of the round alone, and ONE double-buffered table slice
// (row parity) serves the whole workgroup.  The padding costs ~ROUND_OCTS/2 octs per row.
#ifndef NK_OCTS
#define NK_OCTS 2  // numerator kernel: octs per producer per round
#endif
constexpr int SIM_NP = 7;    // producer waves (+1 consumer = 8 waves = 2 per SIMD)
constexpr int SIM_OCTS = 2;  // octs per producer per round
constexpr int SIM_ROUND_OCTS = SIM_NP * SIM_OCTS;       // 14 octs = 112 steps per round
constexpr int SIM_PAIRS = SIM_ROUND_OCTS * 4;           // float4 {x0,w0,x1,w1} per lane per round
constexpr int SIM_MASTER_BYTES = 29 * 32 * 8;           // {D, valid} table, [29][32] x 8 B
constexpr int SIM_RING_BYTES = 2 * SIM_PAIRS * 64 * 16; // 114688
constexpr int SIM_SLICE_STRIDE = 29 * 512;              // table slice [entry][lane] x 8 B, fixed size
constexpr int SIM_WSTAGE_BYTES = 256;                   // per producer: the 16 W values of its round, [lane] x 4 B
__host__ __device__ constexpr int sim_lds_bytes(int) {
    return SIM_MASTER_BYTES + 2 * SIM_SLICE_STRIDE + SIM_RING_BYTES + SIM_NP * SIM_WSTAGE_BYTES;  // 153600
}

// numerator-kernel codes: one byte per residue, NK_K + table row (row `npos` = skipped), 8 rows per uint2
__global__ __launch_bounds__(256) void sim_encode8_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                          const uint8_t *__restrict__ lut_g, int npos,
                                                          const int32_t *__restrict__ gaps_w,
                                                          uint2 *__restrict__ codes8,
                                                          unsigned long long *__restrict__ err_key, int kbase) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ld) return;
    const int g = blockIdx.y;  // 0 .. G8: the extra row G8 lies past row m-1 => all skipped
    bool skipcol = true;
    if (c < n) skipcol = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = g * 8 + r;
        uint32_t idx = (uint32_t)npos;
        if (row < m && c < n) {
            const uint32_t byte = raw[(size_t)row * ld + c];
            const uint32_t code = lut[byte];  // idx * 8, 224 = skipped, 0xFE / 0xFF = bad symbol
            if (code >= 0xFEu) {
                if (!skipcol) {
                    const unsigned long long key = ((unsigned long long)c << 40) | ((unsigned long long)row << 16) |
                                                   ((unsigned long long)(code & 1u) << 8) | byte;
                    atomicMax(err_key, ~key);  // (kept complemented: 0 = none, the largest complement = the first residue)
                }
            } else if (code != 224u) {
                idx = code >> 3;
            }
        }
        w[r >> 2] |= ((uint32_t)kbase + idx) << (8 * (r & 3));
    }
    codes8[(size_t)g * ld + c] = make_uint2(w[0], w[1]);
}

// numerator-kernel codes in the producers' transposed order (tiles of 64 columns):
// [group of 16 rows g][tile][lane p] x 16 B; word w, byte b of lane p = 4 x table row of
// (row 16 g + 4 b + p % 4, column 64 tile + 16 w + p / 4); row `npos` = skipped.  Group (m + 15) / 16 lies past
// the last row: all skipped.
__global__ __launch_bounds__(256) void sim_encodeT_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                          const uint8_t *__restrict__ lut_g, int npos,
                                                          const int32_t *__restrict__ gaps_w,
                                                          uint4 *__restrict__ codesT, int ntiles,
                                                          unsigned long long *__restrict__ err_key) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), p = threadIdx.x & 63, g = blockIdx.y;
    if (tile >= ntiles) return;
    uint32_t word[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int c = tile * 64 + 16 * w + (p >> 2);
        bool skipcol = true;
        if (c < n) skipcol = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
        uint32_t x = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = g * 16 + 4 * b + (p & 3);
            uint32_t idx = (uint32_t)npos;
            if (row < m && c < n) {
                const uint32_t byte = raw[(size_t)row * ld + c];
                const uint32_t code = lut[byte];  // idx * 8, 224 = skipped, 0xFE / 0xFF = bad symbol
                if (code >= 0xFEu) {
                    if (!skipcol) {
                        const unsigned long long key = ((unsigned long long)c << 40) | ((unsigned long long)row << 16) |
                                                       ((unsigned long long)(code & 1u) << 8) | byte;
                        atomicMax(err_key, ~key);  // (kept complemented: 0 = none, the largest complement = the first residue)
                    }
                } else if (code != 224u) {
                    idx = code >> 3;
                }
            }
            x |= (idx * 4u) << (8 * b);
        }
        word[w] = x;
    }
    codesT[((size_t)g * ntiles + tile) * 64 + p] = make_uint4(word[0], word[1], word[2], word[3]);
}

__global__ __launch_bounds__(256) void sim_encode32_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                           const uint8_t *__restrict__ lut_g, int npos,
                                                           const int32_t *__restrict__ gaps_w,
                                                           uint4 *__restrict__ codes32,
                                                           unsigned long long *__restrict__ err_key, int tcols) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ld) return;
    const int g = blockIdx.y;  // 0 .. G8: the extra row G8 lies past row m-1 => all skipped
    bool skipcol = true;
    if (c < n) skipcol = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
    uint32_t half[8];
    const uint32_t lane8 = (uint32_t)(c % tcols) * 8u;  // the column's lane in its similarity workgroup
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = g * 8 + r;
        uint32_t idx = (uint32_t)npos;
        if (row < m && c < n) {
            const uint32_t byte = raw[(size_t)row * ld + c];
            const uint32_t code = lut[byte];  // idx * 8, 224 = skipped, 0xFE / 0xFF = bad symbol
            if (code >= 0xFEu) {
                if (!skipcol) {
                    const unsigned long long key = ((unsigned long long)c << 40) | ((unsigned long long)row << 16) |
                                                   ((unsigned long long)(code & 1u) << 8) | byte;
                    atomicMax(err_key, ~key);  // (kept complemented: 0 = none, the largest complement = the first residue)
                }
            } else if (code != 224u) {
                idx = code >> 3;
            }
        }
        half[r] = idx * 512u + lane8;
    }
    // [oct][half][column]: each half is one coalesced 16-B load per lane
    codes32[((size_t)g * 2 + 0) * ld + c] = make_uint4(half[0], half[1], half[2], half[3]);
    codes32[((size_t)g * 2 + 1) * ld + c] = make_uint4(half[4], half[5], half[6], half[7]);
}

__device__ __forceinline__ void sim_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The chain itself: 28 dependent v_pk_add_f32 {num, den} += {x, w_eff} in pair order, as ONE asm block.
// Left to the compiler, every dependent pair of packed-fp32 adds gets an s_nop between them (its
// dst-sel forwarding hazard rule, which does not apply to full-dword packed adds; parity is bit-exact
// without it) and a lone wave pays a whole issue slot for each: 8.3 instead of 4.7 cycles per step
// (profiles/r01_ubench_chain_forms.txt).
__device__ __forceinline__ void sim_chain(f32x2 &acc, const float4 (&v)[SIM_PAIRS / 4]) {
    static_assert(SIM_PAIRS / 4 == 14, "operand list below");
#define SIM_XY(p) "v"(f32x2{v[p].x, v[p].y}), "v"(f32x2{v[p].z, v[p].w})
    asm volatile(
        "v_pk_add_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %0, %3, %0\n\tv_pk_add_f32 %0, %4, %0\n\t"
        "v_pk_add_f32 %0, %5, %0\n\tv_pk_add_f32 %0, %6, %0\n\tv_pk_add_f32 %0, %7, %0\n\tv_pk_add_f32 %0, %8, %0\n\t"
        "v_pk_add_f32 %0, %9, %0\n\tv_pk_add_f32 %0, %10, %0\n\tv_pk_add_f32 %0, %11, %0\n\tv_pk_add_f32 %0, %12, %0\n\t"
        "v_pk_add_f32 %0, %13, %0\n\tv_pk_add_f32 %0, %14, %0\n\tv_pk_add_f32 %0, %15, %0\n\tv_pk_add_f32 %0, %16, %0\n\t"
        "v_pk_add_f32 %0, %17, %0\n\tv_pk_add_f32 %0, %18, %0\n\tv_pk_add_f32 %0, %19, %0\n\tv_pk_add_f32 %0, %20, %0\n\t"
        "v_pk_add_f32 %0, %21, %0\n\tv_pk_add_f32 %0, %22, %0\n\tv_pk_add_f32 %0, %23, %0\n\tv_pk_add_f32 %0, %24, %0\n\t"
        "v_pk_add_f32 %0, %25, %0\n\tv_pk_add_f32 %0, %26, %0\n\tv_pk_add_f32 %0, %27, %0\n\tv_pk_add_f32 %0, %28, %0"
        : "+v"(acc)
        : SIM_XY(0), SIM_XY(1), SIM_XY(2), SIM_XY(3), SIM_XY(4), SIM_XY(5), SIM_XY(6), SIM_XY(7), SIM_XY(8), SIM_XY(9),
          SIM_XY(10), SIM_XY(11), SIM_XY(12), SIM_XY(13));
#undef SIM_XY
}

// Diagnostics (MSA_SIM_MODE bit 6): per-phase cycle sums of workgroup 0, [wave][phase].
__device__ unsigned long long g_sim_stamps[8 * 8];
__device__ unsigned long long g_den_ticks[1024];  // diagnostics: cycles of every denominator wave (first 1024 chunks)
__device__ __forceinline__ unsigned long long sim_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// everything a producer needs for one oct, fetched two rounds ahead
struct SimOct {
    uint4 c0, c1;  // codes32 of rows 8g..8g+7 (the all-skipped row G8 for a null oct): byte offsets
                   // into a table slice, one dword per step so that a gather needs no address op
};
typedef float f32x4 __attribute__((ext_vector_type(4)));

// round position: row j and first oct gb of the round; rows own the octs (j+1)>>3 .. G8-1
struct SimPos {
    int j, gb;
};
__device__ __forceinline__ SimPos sim_next(SimPos p, int G8) {
    p.gb += SIM_ROUND_OCTS;
    if (p.gb >= G8) {
        ++p.j;
        p.gb = (p.j + 1) >> 3;
    }
    return p;
}

// DIAG = true compiles the diagnostics in (MSA_SIM_MODE: bit0 producers skip gather/emit,
// bit1 consumer skips the chain, bit5 codes from a few cache lines, bit6 phase stamps).  The
// production instantiation has none of these branches: the compiler's s_waitcnt placement in
// this loop is sensitive to every extra control-flow edge.
template <bool DIAG>
__device__ __forceinline__ void sim_producer(const int P, unsigned char *smem, const uint4 *__restrict__ codes32,
                                             int m, int64_t ld, const float *__restrict__ wmat, int ldw, int npos,
                                             int lane, int c, int rounds, int mode_arg) {
    const int mode = DIAG ? mode_arg : 0;
    const f32x2 *master = reinterpret_cast<const f32x2 *>(smem);
    unsigned char *slices = smem + SIM_MASTER_BYTES;
    constexpr int slice_bytes = SIM_SLICE_STRIDE;
    float4 *ring = reinterpret_cast<float4 *>(smem + SIM_MASTER_BYTES + 2 * slice_bytes);
    const int G8 = (m + 7) >> 3;
    const uint4 *col = codes32 + c;
    // W[j][k..k+15] of a round is wave-uniform.  Lanes load it as 16 consecutive floats with ONE vector
    // load (a 64-lane broadcast load of 16 B costs the vector memory pipe as much as a full 1-KiB load,
    // and that pipe is what bounds this kernel), park them in a private LDS line and read them back
    // as four broadcast ds_read_b128 (same address in every lane) right behind the gathers.  Scalar
    // loads would share lgkmcnt with the LDS traffic and return out of order.
    const float *wlane = wmat + (lane & 15);
    float *wstage = reinterpret_cast<float *>(smem + SIM_MASTER_BYTES + 2 * slice_bytes + SIM_RING_BYTES +
                                              P * SIM_WSTAGE_BYTES);
    const uint32_t wstage_addr = (uint32_t)(SIM_MASTER_BYTES + 2 * slice_bytes + SIM_RING_BYTES + P * SIM_WSTAGE_BYTES);

    // 32-bit offsets (the launcher checks the arrays are < 4 GiB): 64-bit scalar multiplies would
    // dominate the fetch, and every instruction of a lone wave costs ~4.75 cycles.
    const unsigned char *codes_bytes = reinterpret_cast<const unsigned char *>(codes32);
    const uint32_t ld16 = (uint32_t)ld * 16u, c16 = (uint32_t)c * 16u;
    auto fetch = [&](SimOct (&u)[SIM_OCTS], float &wv, SimPos p) {  // branch-free: null octs read the skipped row
        const bool past = p.j >= m - 1;
        const uint32_t wrow = (uint32_t)(past ? 0 : p.j) * (uint32_t)ldw;
#pragma unroll
        for (int t = 0; t < SIM_OCTS; ++t) {
            const int g = p.gb + P * SIM_OCTS + t;
            int gc = (past || g >= G8) ? G8 : g;
            if (DIAG && (mode & 32)) gc = P;  // diagnostics: always the same few cache lines
            const uint32_t off = (uint32_t)gc * 2u * ld16 + c16;
            u[t].c0 = *reinterpret_cast<const uint4 *>(codes_bytes + off);
            u[t].c1 = *reinterpret_cast<const uint4 *>(codes_bytes + (off + ld16));
        }
        // a null oct multiplies zero table entries: any finite W does (padding columns of W are zero,
        // a read that runs past the row end lands in the next row)
        const int g0 = p.gb + P * SIM_OCTS;
        const int gw = g0 >= G8 ? G8 - 1 : g0;
        wv = wlane[wrow + 8u * (uint32_t)gw];
    };
    // this lane's table row index for row jn (its residue in that row), npos when skipped
    auto load_cj = [&](int jn) -> uint32_t {
        if (jn >= m - 1) return (uint32_t)npos << 9;
        const uint32_t *cj = reinterpret_cast<const uint32_t *>(col + ((size_t)(jn >> 3) * 2 + ((jn & 7) >> 2)) * ld);
        return cj[jn & 3];
    };
    // producers share the copy of table row idx into slice[jn & 1]: entry e -> [e][lane]
    auto refresh = [&](int jn, uint32_t cjcode) {
        const uint32_t idx = cjcode >> 9;
        f32x2 *sl = reinterpret_cast<f32x2 *>(slices + (jn & 1) * slice_bytes) + lane;
        const f32x2 *mrow = master + idx * 32;
        for (int e = P; e <= npos; e += SIM_NP) sl[e * 64] = mrow[e];
    };
    // The gather address is the code itself: the slice base is a literal LDS address (this kernel
    // has no static __shared__, so the dynamic segment starts at LDS address 0 -- checked at kernel
    // entry) carried in the ds_read offset field, so a gather costs no address instruction.  Inline
    // asm because the compiler would merge the two parities into one path with a selected base and
    // an add per gather; its lgkmcnt bookkeeping does not see these reads, hence the explicit
    // s_waitcnt at the top of emit_all.
    auto gather = [&](const SimOct &o, auto base, f32x2 (&tv)[8]) {
        constexpr uint32_t BASE = decltype(base)::value;
        const uint32_t cw[8] = {o.c0.x, o.c0.y, o.c0.z, o.c0.w, o.c1.x, o.c1.y, o.c1.z, o.c1.w};
#pragma unroll
        for (int s = 0; s < 8; ++s)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tv[s]) : "v"(cw[s]), "i"(BASE));
    };
    auto emit = [&](const f32x4 &w03, const f32x4 &w47, const f32x2 (&tv)[8], float4 *out) {
        // {x, w_eff} = {D, valid} * {W, W}: one v_pk_mul_f32 per step, W broadcast from the low or
        // the high half of an aligned register pair through op_sel (no moves, no scratch)
        const f32x2 wp[4] = {{w03.x, w03.y}, {w03.z, w03.w}, {w47.x, w47.y}, {w47.z, w47.w}};
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            f32x2 xa, xb;
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(xa) : "v"(tv[2 * pp]), "v"(wp[pp]));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(xb) : "v"(tv[2 * pp + 1]), "v"(wp[pp]));
            out[pp * 64] = make_float4(xa.x, xa.y, xb.x, xb.y);
        }
    };
    // one round of this producer = gather (16 LDS reads in flight) ... emit (16 multiplies, 8 ring
    // stores); the next-but-one round's fetch is slotted between the two so that its scalar
    // arithmetic and memory requests issue while the gathers wait for the LDS
    f32x2 tvs[SIM_OCTS][8];
    f32x4 wq[2 * SIM_OCTS];
    auto gather_all = [&](const SimOct (&u)[SIM_OCTS], float wv, int j) {
        wstage[lane] = wv;  // lanes 16.. hold copies; the LDS executes a wave's operations in order
        asm volatile("" ::: "memory");
        // constant slice bases, so that the base folds into the ds_read offset field
        if (j & 1) {
#pragma unroll
            for (int t = 0; t < SIM_OCTS; ++t)
                gather(u[t], std::integral_constant<uint32_t, SIM_MASTER_BYTES + SIM_SLICE_STRIDE>{}, tvs[t]);
        } else {
#pragma unroll
            for (int t = 0; t < SIM_OCTS; ++t) gather(u[t], std::integral_constant<uint32_t, SIM_MASTER_BYTES>{}, tvs[t]);
        }
#pragma unroll
        for (int i = 0; i < 2 * SIM_OCTS; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[i]) : "v"(wstage_addr), "i"(16 * i));
    };
    auto emit_all = [&](int r) {
        float4 *out = ring + ((r & 1) * SIM_PAIRS + P * SIM_OCTS * 4) * 64 + lane;
        // the asm reads above: the registers are tied to the wait so that no use (not even a copy)
        // can be scheduled ahead of it
        static_assert(SIM_OCTS == 2, "operand list below");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(tvs[0][0]), "+v"(tvs[0][1]),
                       "+v"(tvs[0][2]), "+v"(tvs[0][3]), "+v"(tvs[0][4]), "+v"(tvs[0][5]), "+v"(tvs[0][6]),
                       "+v"(tvs[0][7]), "+v"(tvs[1][0]), "+v"(tvs[1][1]), "+v"(tvs[1][2]), "+v"(tvs[1][3]),
                       "+v"(tvs[1][4]), "+v"(tvs[1][5]), "+v"(tvs[1][6]), "+v"(tvs[1][7])
                     :
                     : "memory");
#pragma unroll
        for (int t = 0; t < SIM_OCTS; ++t) emit(wq[2 * t], wq[2 * t + 1], tvs[t], out + t * 4 * 64);
    };

    SimPos pos = {0, 0};
    refresh(0, load_cj(0));
    uint32_t cj_next = load_cj(1);  // code of the next row, loaded a whole row ahead of its use
    sim_barrier();                  // slice[0] complete
    // Loads run TWO rounds ahead of their use (three register sets): under load an L2 hit takes
    // about as long as a whole round, so one round of distance leaves the latency exposed.
    SimOct a[SIM_OCTS], b[SIM_OCTS], d[SIM_OCTS];
    float wa, wb, wd;
    SimPos pos1 = sim_next(pos, G8);
    fetch(a, wa, pos);
    fetch(b, wb, pos1);
    const bool stamp = DIAG && (mode & 64) && blockIdx.x == 0;
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
    auto round = [&](SimOct (&cur)[SIM_OCTS], float wcur, SimOct (&far)[SIM_OCTS], float &wfar, int r) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (stamp) t0 = sim_now();
        if (stamp) t1 = sim_now();
        const SimPos pos2 = sim_next(pos1, G8);
        if (DIAG && (mode & 1)) {
            fetch(far, wfar, pos2);
        } else {
            gather_all(cur, wcur, pos.j);
            fetch(far, wfar, pos2);
            if (stamp) t2 = sim_now();
            emit_all(r);
            if (stamp) t3 = sim_now();
        }
        if (pos1.j != pos.j) {  // last round of row j (wave- and workgroup-uniform): stage row j+1
            refresh(pos1.j, cj_next);
            cj_next = load_cj(pos1.j + 1);
        }
        if (stamp) t4 = sim_now();
        sim_barrier();
        if (stamp) {
            const unsigned long long t5 = sim_now();
            acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4;
        }
        pos = pos1;
        pos1 = pos2;
    };
    for (int r = 0; r < rounds; r += 3) {
        round(a, wa, d, wd, r);
        if (r + 1 < rounds) round(b, wb, a, wa, r + 1);
        if (r + 2 < rounds) round(d, wd, b, wb, r + 2);
    }
    if (stamp && lane == 0)
        for (int k = 0; k < 5; ++k) g_sim_stamps[(P + 1) * 8 + k] = acc_t[k];
    sim_barrier();  // the consumer's drain round
}

template <bool DIAG>
__global__ __launch_bounds__(512) void similarity_pc_kernel(
    const uint4 *__restrict__ codes32, int m, int n, int64_t ld, const float *__restrict__ wmat, int ldw,
    const f32x2 *__restrict__ tab_g, int npos, const int32_t *__restrict__ gaps_w, int rounds, int mode_arg,
    float *__restrict__ q_out, float *__restrict__ mdk_out, int tcols) {
    const int mode = DIAG ? mode_arg : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the producers address the table slices by literal LDS addresses (see `gather`)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * tcols + lane;
    // Only the first tcols lanes of every wave work (EXEC stays masked for the whole kernel): LDS time
    // is per active lane, and the columns are spread over all CUs (sim_tile_cols).
    const bool active = lane < tcols && c < ld;
    {
        f32x2 *master = reinterpret_cast<f32x2 *>(smem);
        for (int t = threadIdx.x; t < 29 * 32; t += 512) master[t] = tab_g[t];
    }
    __syncthreads();
    if (wave != 0) {
        if (active) sim_producer<DIAG>(wave - 1, smem, codes32, m, ld, wmat, ldw, npos, lane, c, rounds, mode);
    } else if (active) {
        __builtin_amdgcn_s_setprio(3);  // the chain wave wins every issue arbitration on its SIMD
        const float4 *ring = reinterpret_cast<const float4 *>(smem + SIM_MASTER_BYTES + 2 * SIM_SLICE_STRIDE);
        f32x2 acc = {0.0f, 0.0f};  // {num, den}
        sim_barrier();             // slice[0] staged
        sim_barrier();             // round 0 produced
        const bool stamp = DIAG && (mode & 64) && blockIdx.x == 0;
        unsigned long long tw = 0, tb = 0;
        // The chain runs half a round behind the reads.  Right after a barrier the producers flood
        // the LDS queue and reads issued then take long to return; so the consumer always keeps two
        // quarter-rounds of ring data pending in registers across the barrier and adds those while
        // the new round's first reads crawl through the queue.  Three register sets of one quarter
        // (SIM_PAIRS / 4 float4) rotate; with four quarters per round the rotation repeats every
        // three rounds, hence the unrolling.  (+0 entries before round 1 leave the sums unchanged.)
        constexpr int QP = SIM_PAIRS / 4;
        float4 s0[QP], s1[QP], s2[QP];
#pragma unroll
        for (int p = 0; p < QP; ++p) s0[p] = s1[p] = s2[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto rd = [&](float4 (&v)[QP], const float4 *in, int quarter) {
#pragma unroll
            for (int p = 0; p < QP; ++p) v[p] = in[(quarter * QP + p) * 64];
            __builtin_amdgcn_sched_barrier(0);
        };
        auto add = [&](const float4 (&v)[QP]) { sim_chain(acc, v); };
        // one round: pending on entry = (x, y) holding quarters 2, 3 of the previous round, z free;
        // pending on exit = (y, z) holding quarters 2, 3 of this round, x free
        auto one_round = [&](float4 (&x)[QP], float4 (&y)[QP], float4 (&z)[QP], int r) {
            unsigned long long t0 = 0, t1 = 0;
            if (stamp) t0 = sim_now();
            const float4 *in = ring + ((r - 1) & 1) * SIM_PAIRS * 64 + lane;
            if (!(DIAG && (mode & 2))) {
                rd(z, in, 0);
                add(x);
                rd(x, in, 1);
                add(y);
                rd(y, in, 2);
                add(z);
                rd(z, in, 3);
                add(x);
            }
            if (stamp) t1 = sim_now();
            sim_barrier();  // also waits for the y / z reads: their buffer is rewritten two rounds on
            if (stamp) {
                tw += t1 - t0;
                tb += sim_now() - t1;
            }
        };
        for (int r = 1; r + 2 <= rounds; r += 3) {  // the launcher makes `rounds` a multiple of 3
            one_round(s0, s1, s2, r);
            one_round(s1, s2, s0, r + 1);
            one_round(s2, s0, s1, r + 2);
        }
        add(s0);  // the two quarters still pending
        add(s1);
        if (stamp && lane == 0) {
            g_sim_stamps[0] = tw;
            g_sim_stamps[1] = tb;
            g_sim_stamps[2] = (unsigned long long)rounds;
        }
        if (c < n) {
            const bool skip = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
            float q = 0.0f, v = 0.0f;
            if (!skip && acc.y != 0.0f) {
                q = acc.x / acc.y;
                v = __uint_as_float(0x7FC00000u);  // (the host evaluates the exponential: see sim_finish_kernel)
            }
            if (q_out) q_out[c] = q;
            mdk_out[c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// similarity denominators as their own kernel.  den[c] = sum over valid pairs (j < k) of W[j][k] in
// pair order (float32, sequential): it depends only on W and on which residues are valid, so it needs
// no table, no codes and no LDS.  One wave = one 32-column chunk of the validity plane (plane 7 of
// `planes`, [chunk][row] u32): the pair's lane mask is Vj & Vk, a scalar AND into EXEC, and the add
// is one v_add_f32 with the wave-uniform W[j][k] under that mask -- lanes whose pair is not valid
// keep their sum, which is what adding +0 would do.  Two instructions per pair step for a lone wave
// (10.8 cycles measured, profiles/r01_ubench_den_wave.txt); it runs beside the numerator kernel on
// CUs that one leaves idle.  Masks and W stream through the scalar cache (16 steps per buffer, two buffers,
// lgkmcnt(0) discipline: scalar loads return out of order) and feed the adds as SGPR operands; 16-B
// broadcast vector loads for W were measured slower (17.7 vs 12.9 cycles per step: each costs the wave
// ~24 cycles of issue).  Rows start at the 16-aligned k below j+1 and end at the 16-aligned k above m:
// W is strictly upper triangular and zero-padded, so the extra steps add +0.
// ------------------------------------------------------------------------------------------
constexpr int DEN_WAVES = 8;             // chunks per workgroup at most (default 4: one per SIMD)
constexpr int DEN_GROUP = 16;            // pair steps per SGPR buffer of the denominator loop
constexpr int DEN_LDS_BYTES = 96 * 1024;  // never touched: keeps a numerator workgroup off this CU (see below)
__global__ __launch_bounds__(64 * DEN_WAVES) void sim_den_kernel(const uint32_t *__restrict__ planes, int nchunk,
                                                                 int m_pad, int m, int n,
                                                                 const float *__restrict__ wmat, int ldw,
                                                                 float *__restrict__ den_out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (chunk >= nchunk) return;
    const uint32_t *masks = planes + ((size_t)7 * nchunk + chunk) * (size_t)m_pad;
    const int lane = threadIdx.x & 63;
    constexpr int G = DEN_GROUP;  // pair steps per buffer
    const int mend = (m + G - 1) / G * G;
    float den = 0.0f;
    const unsigned long long t_start = sim_now();
    unsigned long long n_steps = 0;
    uint32_t vnext = __builtin_amdgcn_readfirstlane(masks[0]);
    for (int j = 0; j + 1 < m; ++j) {
        const uint32_t vj = vnext;
        const uint32_t *mnext = masks + j + 1;
        if (vj == 0u) {  // no column of this chunk has a residue in row j
            vnext = __builtin_amdgcn_readfirstlane(*mnext);
            continue;
        }
        const int k0 = (j + 1) / G * G;
        const int ng = (mend - k0) / G;  // >= 1 groups
        n_steps += (unsigned long long)ng * G;
        const uint32_t *mp = masks + k0;
        const float *wp = wmat + ((size_t)j * (size_t)ldw + (size_t)k0);
        // W is streamed once per XCD (16 MB at m = 2000: it does not stay in the 4 MB L2) and the loop below
        // covers only one group of latency: a W line that has to come from HBM stalls every wave that needs it
        // (alone on the GPU this kernel runs 2.4x slower than beside the numerator kernel, whose W traffic
        // happens to warm the L2).  So the workgroup touches the row after next with vector loads that nobody
        // waits for: LDS-DMA into a scratch line, no register in flight.
        if (j + 2 < m) {
            const float *pre = wmat + (size_t)(j + 2) * (size_t)ldw;
            for (int off = (j + 2) / 8 * 8 + (threadIdx.x * 8); off < ldw; off += (int)blockDim.x * 8) {
                const uint32_t voff = (uint32_t)off * 4u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(1024u), "v"(voff), "s"(pre) : "m0", "memory");
            }
        }
        // Two SGPR buffers of 16 steps: A = masks s[36:51], W s[52:67]; B = masks s[68:83], W s[84:99].
        // The loop is ISSUE-bound (a lone wave issues one instruction per ~4.3 cycles whatever its type: 2
        // instructions per step + the loop's own), so the bookkeeping is pared down: one byte offset (s30)
        // serves both streams, the group counter exits on the borrow of its decrement.
        asm volatile(
            "s_mov_b64 s[10:11], exec\n\t"
            "s_mov_b32 exec_hi, 0\n\t"
            "s_mov_b32 s8, %2\n\ts_mov_b64 s[12:13], %4\n\ts_mov_b64 s[14:15], %5\n\ts_mov_b32 s30, 0\n\t"
            "s_load_dword %1, %6, 0x0\n\t"  // V(j+1): a vector load + readfirstlane here costs a memory round trip per row
            "s_load_dwordx16 s[36:51], s[12:13], s30\n\ts_load_dwordx16 s[52:67], s[14:15], s30\n\t"
            // %3 = ng: the loop runs ng / 2 pairs of groups with ONE exit test per pair, an odd group follows
            "s_lshr_b32 s9, %3, 1\n\t"
            "s_sub_u32 s9, s9, 1\n\t"
            "s_cbranch_scc1 2f\n\t"  // no pair at all
            "1:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_load_dwordx16 s[68:83], s[12:13], s30 offset:0x40\n\ts_load_dwordx16 s[84:99], s[14:15], s30 offset:0x40\n\t"
            "s_and_b32 exec_lo, s8, s36\n\tv_add_f32 %0, s52, %0\n\t"
            "s_and_b32 exec_lo, s8, s37\n\tv_add_f32 %0, s53, %0\n\t"
            "s_and_b32 exec_lo, s8, s38\n\tv_add_f32 %0, s54, %0\n\t"
            "s_and_b32 exec_lo, s8, s39\n\tv_add_f32 %0, s55, %0\n\t"
            "s_and_b32 exec_lo, s8, s40\n\tv_add_f32 %0, s56, %0\n\t"
            "s_and_b32 exec_lo, s8, s41\n\tv_add_f32 %0, s57, %0\n\t"
            "s_and_b32 exec_lo, s8, s42\n\tv_add_f32 %0, s58, %0\n\t"
            "s_and_b32 exec_lo, s8, s43\n\tv_add_f32 %0, s59, %0\n\t"
            "s_and_b32 exec_lo, s8, s44\n\tv_add_f32 %0, s60, %0\n\t"
            "s_and_b32 exec_lo, s8, s45\n\tv_add_f32 %0, s61, %0\n\t"
            "s_and_b32 exec_lo, s8, s46\n\tv_add_f32 %0, s62, %0\n\t"
            "s_and_b32 exec_lo, s8, s47\n\tv_add_f32 %0, s63, %0\n\t"
            "s_and_b32 exec_lo, s8, s48\n\tv_add_f32 %0, s64, %0\n\t"
            "s_and_b32 exec_lo, s8, s49\n\tv_add_f32 %0, s65, %0\n\t"
            "s_and_b32 exec_lo, s8, s50\n\tv_add_f32 %0, s66, %0\n\t"
            "s_and_b32 exec_lo, s8, s51\n\tv_add_f32 %0, s67, %0\n\t"
            "s_add_u32 s30, s30, 0x80\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_load_dwordx16 s[36:51], s[12:13], s30\n\ts_load_dwordx16 s[52:67], s[14:15], s30\n\t"
            "s_and_b32 exec_lo, s8, s68\n\tv_add_f32 %0, s84, %0\n\t"
            "s_and_b32 exec_lo, s8, s69\n\tv_add_f32 %0, s85, %0\n\t"
            "s_and_b32 exec_lo, s8, s70\n\tv_add_f32 %0, s86, %0\n\t"
            "s_and_b32 exec_lo, s8, s71\n\tv_add_f32 %0, s87, %0\n\t"
            "s_and_b32 exec_lo, s8, s72\n\tv_add_f32 %0, s88, %0\n\t"
            "s_and_b32 exec_lo, s8, s73\n\tv_add_f32 %0, s89, %0\n\t"
            "s_and_b32 exec_lo, s8, s74\n\tv_add_f32 %0, s90, %0\n\t"
            "s_and_b32 exec_lo, s8, s75\n\tv_add_f32 %0, s91, %0\n\t"
            "s_and_b32 exec_lo, s8, s76\n\tv_add_f32 %0, s92, %0\n\t"
            "s_and_b32 exec_lo, s8, s77\n\tv_add_f32 %0, s93, %0\n\t"
            "s_and_b32 exec_lo, s8, s78\n\tv_add_f32 %0, s94, %0\n\t"
            "s_and_b32 exec_lo, s8, s79\n\tv_add_f32 %0, s95, %0\n\t"
            "s_and_b32 exec_lo, s8, s80\n\tv_add_f32 %0, s96, %0\n\t"
            "s_and_b32 exec_lo, s8, s81\n\tv_add_f32 %0, s97, %0\n\t"
            "s_and_b32 exec_lo, s8, s82\n\tv_add_f32 %0, s98, %0\n\t"
            "s_and_b32 exec_lo, s8, s83\n\tv_add_f32 %0, s99, %0\n\t"
            "s_sub_u32 s9, s9, 1\n\ts_cbranch_scc0 1b\n\t"
            "2:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_bitcmp0_b32 %3, 0\n\t"
            "s_cbranch_scc1 3f\n\t"  // even group count: done (buffer A holds a group past the row end)
            "s_and_b32 exec_lo, s8, s36\n\tv_add_f32 %0, s52, %0\n\t"
            "s_and_b32 exec_lo, s8, s37\n\tv_add_f32 %0, s53, %0\n\t"
            "s_and_b32 exec_lo, s8, s38\n\tv_add_f32 %0, s54, %0\n\t"
            "s_and_b32 exec_lo, s8, s39\n\tv_add_f32 %0, s55, %0\n\t"
            "s_and_b32 exec_lo, s8, s40\n\tv_add_f32 %0, s56, %0\n\t"
            "s_and_b32 exec_lo, s8, s41\n\tv_add_f32 %0, s57, %0\n\t"
            "s_and_b32 exec_lo, s8, s42\n\tv_add_f32 %0, s58, %0\n\t"
            "s_and_b32 exec_lo, s8, s43\n\tv_add_f32 %0, s59, %0\n\t"
            "s_and_b32 exec_lo, s8, s44\n\tv_add_f32 %0, s60, %0\n\t"
            "s_and_b32 exec_lo, s8, s45\n\tv_add_f32 %0, s61, %0\n\t"
            "s_and_b32 exec_lo, s8, s46\n\tv_add_f32 %0, s62, %0\n\t"
            "s_and_b32 exec_lo, s8, s47\n\tv_add_f32 %0, s63, %0\n\t"
            "s_and_b32 exec_lo, s8, s48\n\tv_add_f32 %0, s64, %0\n\t"
            "s_and_b32 exec_lo, s8, s49\n\tv_add_f32 %0, s65, %0\n\t"
            "s_and_b32 exec_lo, s8, s50\n\tv_add_f32 %0, s66, %0\n\t"
            "s_and_b32 exec_lo, s8, s51\n\tv_add_f32 %0, s67, %0\n\t"
            "3:\n\t"
            "s_mov_b64 exec, s[10:11]"
            : "+v"(den), "=&s"(vnext)
            : "s"(vj), "s"(ng), "s"(mp), "s"(wp), "s"(mnext)
            : "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s30", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "scc", "memory");
    }
    if (lane == 0) {  // diagnostics (tools/sim_modes.py)
        const unsigned long long dt = sim_now() - t_start;
        if (chunk < 1024) g_den_ticks[chunk] = dt;
        if (chunk == 0) {
            g_sim_stamps[56] = dt;
            g_sim_stamps[57] = n_steps;
        }
    }
    const int c = chunk * 32 + lane;
    if (lane < 32 && c < n) den_out[c] = den;
}

// ------------------------------------------------------------------------------------------
// Denominators, two lanes per column.  The EXEC-masked loop above spends a scalar AND and an add on every pair
// step and a lone wave issues one instruction per ~4 cycles.  Here lane 2 c + i of a wave works on column c
// (32 columns per wave, as above) at the steps of parity i: one v_cndmask_b32 selects W or +0 for TWO steps
// (the 64-bit SGPR mask holds valid(c, k + i) at bit 2 c + i: the validity words of rows k and k + 1,
// bit-interleaved by den_pairmask_kernel), and the chain add of step k + i reads the term from lane i of the
// pair through DPP (quad_perm [0,0,2,2] / [1,1,3,3]), so both lanes of a pair carry the same sum.  Per 16 steps:
// 8 selects, 16 adds, one scalar load (8 masks) and two ds_read_b128 (this lane's 8 W values) -- 1.8
// instructions per step instead of 2.26, 9.8 cycles instead of 12.3 (a dependent DPP add costs 5.75 cycles;
// tools/ubench9.hip).  EXEC holds the pairs whose column has a residue in row j.
// W rows reach the LDS by DMA one ROW ahead (wave-private double buffer; the DMA de-interleaves: dword
// [g][i][t] of the line = W[16 g + 2 t + i]), which also hides the HBM latency that the loop above needs its L2
// warm-up loads for.  Rows run from the 16-aligned k below j + 1 over whole blocks of 64 steps: W is strictly
// upper triangular, and past row m the masks are zero, so the extra steps add +0.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long den_spread_bits(uint32_t x) {
    unsigned long long v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__global__ __launch_bounds__(256) void den_pairmask_kernel(const uint32_t *__restrict__ planes, int nchunk, int m_pad,
                                                           int m, unsigned long long *__restrict__ pm, int pm_ld) {
    const int kk = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y;
    if (kk >= pm_ld) return;
    const uint32_t *masks = planes + ((size_t)7 * nchunk + chunk) * (size_t)m_pad;
    const int r0 = 2 * kk;
    const uint32_t a = r0 < m ? masks[r0] : 0u, b = r0 + 1 < m ? masks[r0 + 1] : 0u;
    pm[(size_t)chunk * pm_ld + kk] = den_spread_bits(a) | (den_spread_bits(b) << 1);
}
int den2_pm_ld(int m) { return (((m + 16 + 63) / 64) * 64 + 128) / 2; }
__host__ __device__ inline int den2_row_bytes(int m) { return ((m + 16 + 63) / 64 + 2) * 256; }

__global__ __launch_bounds__(64 * DEN_WAVES) void sim_den2_kernel(const unsigned long long *__restrict__ pm, int pm_ld,
                                                                  int nchunk, int m, int n,
                                                                  const float *__restrict__ wmat, int ldw,
                                                                  float *__restrict__ den_out, int row_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (chunk >= nchunk) return;
    const unsigned long long *pmc = pm + (size_t)chunk * (size_t)pm_ld;
    const int lane = threadIdx.x & 63;
    const int mend = (m + 15) / 16 * 16;
    const uint32_t buf0 = (uint32_t)(wave * 2 * row_bytes);
    // DMA lane L fills dword L of a 256-byte line = [group L / 16][parity (L / 8) % 2][t = L % 8] <- element 16 g + 2 t + i
    const uint32_t dma_pat = (uint32_t)(((lane >> 4) * 16 + 2 * (lane & 7) + ((lane >> 3) & 1)) * 4);
    auto issue_row_dma = [&](int jr) {
        if (jr + 1 >= m) return;
        const int k0 = (jr + 1) / 16 * 16;
        const int ng4 = ((mend - k0) / 16 + 3) / 4;
        const float *src = wmat + ((size_t)jr * (size_t)ldw + (size_t)k0);
        const uint32_t dst = buf0 + (uint32_t)(jr & 1) * (uint32_t)row_bytes;
        int q = 0;
        for (; q + 8 <= ng4; q += 8)  // the immediate offset moves the LDS and the global address alike
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "global_load_lds_dword %1, %2\n\tglobal_load_lds_dword %1, %2 offset:256\n\t"
                         "global_load_lds_dword %1, %2 offset:512\n\tglobal_load_lds_dword %1, %2 offset:768\n\t"
                         "global_load_lds_dword %1, %2 offset:1024\n\tglobal_load_lds_dword %1, %2 offset:1280\n\t"
                         "global_load_lds_dword %1, %2 offset:1536\n\tglobal_load_lds_dword %1, %2 offset:1792"
                         ::"s"(dst + (uint32_t)q * 256u), "v"(dma_pat), "s"(src + q * 64) : "m0", "memory");
        for (; q < ng4; ++q)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                         ::"s"(dst + (uint32_t)q * 256u), "v"(dma_pat), "s"(src + q * 64) : "m0", "memory");
    };
    float den = 0.0f;
    const unsigned long long t_start = sim_now();
    unsigned long long n_steps = 0;
    issue_row_dma(0);
    unsigned long long pw = pmc[0];  // the pair word of rows (j, j ^ 1); the row loop fetches the next one by scalar load
    for (int j = 0; j + 1 < m; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // row j's W has landed
        issue_row_dma(j + 1);
        const unsigned long long *pnext = pmc + ((j + 1) >> 1);
        const unsigned long long half = (j & 1) ? ((pw >> 1) & 0x5555555555555555ull) : (pw & 0x5555555555555555ull);
        const unsigned long long vj = half | (half << 1);  // both lanes of every column with a residue in row j
        if (vj == 0ull) {  // no column of this chunk has a residue in row j
            pw = *pnext;
            continue;
        }
        const int k0 = (j + 1) / 16 * 16;
        const int ng4 = ((mend - k0) / 16 + 3) / 4;
        n_steps += (unsigned long long)ng4 * 64;
        unsigned long long pw_next;
        const unsigned long long *pmrow = pmc + (k0 >> 1);
        const uint32_t waddr = buf0 + (uint32_t)(j & 1) * (uint32_t)row_bytes + (uint32_t)(lane & 1) * 32u;
        // One loop pass = 64 steps: v1 = LDS address, v[2:33] = this lane's 32 W values, v[34:65] = the 32 terms,
        // s[36:99] = 32 pair masks.  Scalar loads return out of order, so lgkmcnt(0) is the only usable wait and the
        // latency a pass can hide is one pass: the next pass's masks and W are requested behind the selects and
        // land under the 64 adds (~370 cycles; with 16-step passes every group stalled on its mask load).
        asm volatile(
            "s_mov_b64 s[10:11], exec\n\t"
            "s_mov_b64 exec, %2\n\t"
            "s_mov_b64 s[12:13], %4\n\t"
            "s_sub_u32 s9, %3, 1\n\t"
            "v_mov_b32 v1, %5\n\t"
            "s_load_dwordx2 %1, %6, 0x0\n\t"  // the next row's pair word (a vector load would cost a round trip per row)
            "s_load_dwordx16 s[36:51], s[12:13], 0x0\n\t"
            "s_load_dwordx16 s[52:67], s[12:13], 0x40\n\t"
            "s_load_dwordx16 s[68:83], s[12:13], 0x80\n\t"
            "s_load_dwordx16 s[84:99], s[12:13], 0xc0\n\t"
            "ds_read_b128 v[2:5], v1 offset:0\n\tds_read_b128 v[6:9], v1 offset:16\n\t"
            "ds_read_b128 v[10:13], v1 offset:64\n\tds_read_b128 v[14:17], v1 offset:80\n\t"
            "ds_read_b128 v[18:21], v1 offset:128\n\tds_read_b128 v[22:25], v1 offset:144\n\t"
            "ds_read_b128 v[26:29], v1 offset:192\n\tds_read_b128 v[30:33], v1 offset:208\n\t"
            "1:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cndmask_b32_e64 v34, 0, v2, s[36:37]\n\t"
            "v_cndmask_b32_e64 v35, 0, v3, s[38:39]\n\t"
            "v_cndmask_b32_e64 v36, 0, v4, s[40:41]\n\t"
            "v_cndmask_b32_e64 v37, 0, v5, s[42:43]\n\t"
            "v_cndmask_b32_e64 v38, 0, v6, s[44:45]\n\t"
            "v_cndmask_b32_e64 v39, 0, v7, s[46:47]\n\t"
            "v_cndmask_b32_e64 v40, 0, v8, s[48:49]\n\t"
            "v_cndmask_b32_e64 v41, 0, v9, s[50:51]\n\t"
            "v_cndmask_b32_e64 v42, 0, v10, s[52:53]\n\t"
            "v_cndmask_b32_e64 v43, 0, v11, s[54:55]\n\t"
            "v_cndmask_b32_e64 v44, 0, v12, s[56:57]\n\t"
            "v_cndmask_b32_e64 v45, 0, v13, s[58:59]\n\t"
            "v_cndmask_b32_e64 v46, 0, v14, s[60:61]\n\t"
            "v_cndmask_b32_e64 v47, 0, v15, s[62:63]\n\t"
            "v_cndmask_b32_e64 v48, 0, v16, s[64:65]\n\t"
            "v_cndmask_b32_e64 v49, 0, v17, s[66:67]\n\t"
            "v_cndmask_b32_e64 v50, 0, v18, s[68:69]\n\t"
            "v_cndmask_b32_e64 v51, 0, v19, s[70:71]\n\t"
            "v_cndmask_b32_e64 v52, 0, v20, s[72:73]\n\t"
            "v_cndmask_b32_e64 v53, 0, v21, s[74:75]\n\t"
            "v_cndmask_b32_e64 v54, 0, v22, s[76:77]\n\t"
            "v_cndmask_b32_e64 v55, 0, v23, s[78:79]\n\t"
            "v_cndmask_b32_e64 v56, 0, v24, s[80:81]\n\t"
            "v_cndmask_b32_e64 v57, 0, v25, s[82:83]\n\t"
            "v_cndmask_b32_e64 v58, 0, v26, s[84:85]\n\t"
            "v_cndmask_b32_e64 v59, 0, v27, s[86:87]\n\t"
            "v_cndmask_b32_e64 v60, 0, v28, s[88:89]\n\t"
            "v_cndmask_b32_e64 v61, 0, v29, s[90:91]\n\t"
            "v_cndmask_b32_e64 v62, 0, v30, s[92:93]\n\t"
            "v_cndmask_b32_e64 v63, 0, v31, s[94:95]\n\t"
            "v_cndmask_b32_e64 v64, 0, v32, s[96:97]\n\t"
            "v_cndmask_b32_e64 v65, 0, v33, s[98:99]\n\t"
            "s_add_u32 s12, s12, 0x100\n\ts_addc_u32 s13, s13, 0\n\tv_add_u32_e32 v1, 0x100, v1\n\t"
            "s_load_dwordx16 s[36:51], s[12:13], 0x0\n\t"
            "s_load_dwordx16 s[52:67], s[12:13], 0x40\n\t"
            "s_load_dwordx16 s[68:83], s[12:13], 0x80\n\t"
            "s_load_dwordx16 s[84:99], s[12:13], 0xc0\n\t"
            "ds_read_b128 v[2:5], v1 offset:0\n\tds_read_b128 v[6:9], v1 offset:16\n\t"
            "ds_read_b128 v[10:13], v1 offset:64\n\tds_read_b128 v[14:17], v1 offset:80\n\t"
            "ds_read_b128 v[18:21], v1 offset:128\n\tds_read_b128 v[22:25], v1 offset:144\n\t"
            "ds_read_b128 v[26:29], v1 offset:192\n\tds_read_b128 v[30:33], v1 offset:208\n\t"
            "v_add_f32_dpp %0, v34, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v34, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v35, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v35, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v36, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v36, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v37, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v37, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v38, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v38, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v39, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v39, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v40, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v40, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v41, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v41, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v42, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v42, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v43, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v43, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v44, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v44, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v45, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v45, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v46, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v46, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v47, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v47, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v48, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v48, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v49, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v49, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v50, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v50, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v51, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v51, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v52, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v52, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v53, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v53, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v54, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v54, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v55, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v55, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v56, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v56, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v57, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v57, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v58, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v58, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v59, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v59, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v60, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v60, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v61, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v61, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v62, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v62, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v63, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v63, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v64, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v64, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v65, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, v65, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "s_sub_u32 s9, s9, 1\n\ts_cbranch_scc0 1b\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 exec, s[10:11]"
            : "+v"(den), "=&s"(pw_next)
            : "s"(vj), "s"(ng4), "s"(pmrow), "v"(waddr), "s"(pnext)
            : "s9", "s10", "s11", "s12", "s13", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "scc", "memory");
        pw = pw_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {  // diagnostics (tools/sim_modes.py)
        const unsigned long long dt = sim_now() - t_start;
        if (chunk < 1024) g_den_ticks[chunk] = dt;
        if (chunk == 0) {
            g_sim_stamps[56] = dt;
            g_sim_stamps[57] = n_steps;
        }
    }
    const int c = chunk * 32 + (lane >> 1);
    if ((lane & 1) == 0 && c < n) den_out[c] = den;
}

// Waves (= 32-column chunks) per denominator workgroup.  Each wave cycles through its m validity words once per
// row; they are served by the scalar cache while the waves of a CU fit it together (4 x 8 KB at m = 2000 run at
// 12.4 cycles per step; 4 x 14 KB at m = 3583 ran at 32: every group load exposed the L2 latency).
int sim_den_waves(int m) {
    if (tuning().den_waves >= 1 && tuning().den_waves <= DEN_WAVES) return tuning().den_waves;
    const long mask_bytes = 4L * (m + 64);
    const long w = (36 * 1024) / mask_bytes;
    return (int)(w < 1 ? 1 : (w > 4 ? 4 : w));
}
int sim_den_workgroups(int nchunk, int m) {
    const int w = sim_den_waves(m);
    return (nchunk + w - 1) / w;
}

// pairmasks: [nchunk][den2_pm_ld(m)] u64 of scratch for the two-lanes-per-column kernel (nullptr: EXEC-masked kernel)
int launch_sim_den(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int n, const float *wmat,
                   int ldw, float *den_out, unsigned long long *pairmasks) {
    // The dynamic LDS request is a placement device: with it a CU cannot hold this workgroup and a numerator
    // workgroup at once -- sharing a SIMD with the chain waves of the other kernel slows both by ~1.7x.
    const bool pair_kernel = !tuning().den_exec;  // MSA_DEN_KERNEL=exec: the EXEC-masked kernel (diagnostics, parity tests)
    const int waves = sim_den_waves(m);
    const int row_bytes = den2_row_bytes(m);
    const int lds2 = waves * 2 * row_bytes;
    if (pair_kernel && pairmasks && lds2 <= 150 * 1024) {
        const int pm_ld = den2_pm_ld(m);
        den_pairmask_kernel<<<dim3((pm_ld + 255) / 256, nchunk), 256, 0, s>>>(planes, nchunk, m_pad, m, pairmasks, pm_ld);
        const int lds = lds2 > DEN_LDS_BYTES ? lds2 : DEN_LDS_BYTES;
        if (int e = set_max_lds_once(reinterpret_cast<const void *>(sim_den2_kernel), lds)) return e;
        sim_den2_kernel<<<sim_den_workgroups(nchunk, m), 64 * waves, lds, s>>>(pairmasks, pm_ld, nchunk, m, n, wmat, ldw,
                                                                              den_out, row_bytes);
        return 0;
    }
    if (int e = set_max_lds_once(reinterpret_cast<const void *>(sim_den_kernel), DEN_LDS_BYTES)) return e;
    sim_den_kernel<<<sim_den_workgroups(nchunk, m), 64 * waves, DEN_LDS_BYTES, s>>>(planes, nchunk, m_pad, m, n, wmat, ldw,
                                                                                  den_out);
    return 0;
}

// ------------------------------------------------------------------------------------------
// similarity NUMERATORS (the denominators come from sim_den_kernel): producer/consumer like
// similarity_pc_kernel, with REGISTER-RESIDENT codes (m <= 4032 rows) and half the LDS traffic.
//
// In similarity_pc_kernel every workgroup re-reads its column tile of codes once per row j: at
// 2000 x 10000 that is 85 GB through the vector memory pipe per launch, and that kernel's skeleton
// (fetch + barriers, no LDS work, no chain) already takes 10.7 of its 13.7 ms.  The codes of a
// producer do not depend on j, so here they live in its registers for the whole kernel:
//   * rounds are aligned to ABSOLUTE oct positions (round q = octs 14q .. 14q+13), so producer P
//     always works on octs 14q + 2P, 14q + 2P + 1 and the register holding them is a compile-time
//     function of q: the row loop is unrolled over q.  Row j starts at round q0 = ((j+1)>>3) / 14;
//     the octs of that round that lie at or before j multiply W = 0 (W is strictly upper
//     triangular) -- exact no-ops, like the null octs past the last row;
//   * 8-bit codes (table entry + NK_K), four per dword: 4 VGPRs per round, 144 for NK_RMAX = 36 rounds
//     (m <= 4032); a gather costs one v_perm_b32 (address from the code byte) and one ds_read_b32;
//   * the table slices hold D alone ([entry][lane] x 4 B), the ring carries x = W * D (4 B per lane
//     and step, one float4 = 4 steps), the chain is one v_add_f32 per step; a skipped pair
//     contributes W * 0 = +0;
//   * the only global traffic left in the loop is W (see nk_producer).
// Barrier protocol as in similarity_pc_kernel (2 + rounds barriers, ring buffer = round parity);
// rounds are padded to a multiple of 3 with pseudo-rows j = m-1, whose W row is all zero.
// LDS: master D [29][32] f32 | slice 0 | ring 2 x [28][64] float4 | slice 1 (64 KB above slice 0) | W stage.
// ------------------------------------------------------------------------------------------
constexpr int NK_MASTER_LD = 33;                // row pitch of the master table: lanes read different ROWS at the same
                                               // column when a slice is staged -- with a pitch of 32 that is one bank
constexpr int NK_MASTER_BYTES = 29 * NK_MASTER_LD * 4;   // 3828, at LDS address 0
constexpr int NK_SLICE_STRIDE = 29 * 256;      // 7424: [entry][lane] x 4 B
constexpr int NK_ROUND_OCTS = SIM_NP * NK_OCTS;  // 14 octs = 112 steps per round
constexpr int NK_QUADS = NK_ROUND_OCTS * 2;      // float4 (4 steps) per lane per round
constexpr int NK_RING_BYTES = 2 * NK_QUADS * 64 * 16;  // 57344
// A code is ONE BYTE: NK_K + table entry.  The gather address (slice base + entry * 256 + lane * 4) is built by a
// single v_perm_b32: byte 0 = lane * 4, byte 1 = the code, byte 2 = the row parity (the two slices lie 64 KB apart).
constexpr int NK_K = 16;                                   // slice 0 starts at LDS byte NK_K * 256
constexpr int NK_SLICE0_OFF = NK_K * 256;                  // 4096
constexpr int NK_SLICE1_OFF = NK_SLICE0_OFF + 65536;       // 69632
constexpr int NK_RING_OFF = NK_SLICE0_OFF + NK_SLICE_STRIDE;  // 11520 .. 68864: between the slices
constexpr int NK_WSTAGE_OFF = NK_SLICE1_OFF + NK_SLICE_STRIDE;  // 77056; per producer 2 x 256 B: W of this and the next round
static_assert(NK_MASTER_BYTES <= NK_SLICE0_OFF && NK_RING_OFF + NK_RING_BYTES <= NK_SLICE1_OFF && NK_K + 29 <= 256, "LDS layout");
// Rounds per row with resident codes (one uint4 = 16 codes per round): two instantiations, 18 rounds (m <= 2016)
// and 36 (m <= 4032) -- the row loop is unrolled over the rounds, and the 36-round body (70 KB of code) overflows
// the instruction cache enough to cost 13 % at m = 2000, and the denominator kernel next door as much.
constexpr int NK_RMAX = 36;
__host__ __device__ constexpr int nk_lds_bytes() { return NK_WSTAGE_OFF + SIM_NP * 512; }  // 80640

template <int B>
__device__ __forceinline__ void nk_chain(float &acc, const float4 (&v)[NK_QUADS / 4]) {  // quads B .. B+6: 28 steps
#define NK_Q(p) "v"(v[B + p].x), "v"(v[B + p].y), "v"(v[B + p].z), "v"(v[B + p].w)
    asm volatile(
        "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0\n\t"
        "v_add_f32 %0, %5, %0\n\tv_add_f32 %0, %6, %0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %0, %8, %0\n\t"
        "v_add_f32 %0, %9, %0\n\tv_add_f32 %0, %10, %0\n\tv_add_f32 %0, %11, %0\n\tv_add_f32 %0, %12, %0\n\t"
        "v_add_f32 %0, %13, %0\n\tv_add_f32 %0, %14, %0\n\tv_add_f32 %0, %15, %0\n\tv_add_f32 %0, %16, %0\n\t"
        "v_add_f32 %0, %17, %0\n\tv_add_f32 %0, %18, %0\n\tv_add_f32 %0, %19, %0\n\tv_add_f32 %0, %20, %0\n\t"
        "v_add_f32 %0, %21, %0\n\tv_add_f32 %0, %22, %0\n\tv_add_f32 %0, %23, %0\n\tv_add_f32 %0, %24, %0\n\t"
        "v_add_f32 %0, %25, %0\n\tv_add_f32 %0, %26, %0\n\tv_add_f32 %0, %27, %0\n\tv_add_f32 %0, %28, %0"
        : "+v"(acc)
        : NK_Q(0), NK_Q(1), NK_Q(2), NK_Q(3), NK_Q(4), NK_Q(5), NK_Q(6));
#undef NK_Q
}

template <int Q, int RM, class F>
__device__ __forceinline__ void nk_unroll(F &&f) {
    if constexpr (Q < RM) {
        f(std::integral_constant<int, Q>{});
        nk_unroll<Q + 1, RM>(f);
    }
}

template <bool DIAG, int RING_OFF>
__device__ __forceinline__ void nk_consumer(unsigned char *smem, int rounds, int lane, int c, int n,
                                            float *__restrict__ num_out) {
    const float4 *ring = reinterpret_cast<const float4 *>(smem + RING_OFF);
    float acc = 0.0f;
    sim_barrier();  // slice[0] staged
    sim_barrier();  // round 0 produced
    constexpr int QP = NK_QUADS / 4;
    // Six quarter-round register sets.  A round's 28 ring reads are issued together right after the barrier;
    // the chain first adds the two quarters left pending by the previous round (their reads landed long
    // ago), then this round's first two, and leaves the last two pending across the next barrier: every read
    // has at least two quarters of adds (~240 cycles) plus a barrier to land before it is needed.
    float4 sa[QP], sb[QP], sc[QP], sd[QP], se[QP], sf[QP];
#pragma unroll
    for (int p = 0; p < QP; ++p) sa[p] = sb[p] = sc[p] = sd[p] = se[p] = sf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto rd = [&](float4 (&v)[QP], const float4 *in, int quarter) {
#pragma unroll
        for (int p = 0; p < QP; ++p) v[p] = in[(quarter * QP + p) * 64];
    };
    static_assert(QP % 7 == 0, "the chain is issued in blocks of 7 quads");
    auto add = [&](const float4 (&v)[QP]) {
        nk_chain<0>(acc, v);
        if constexpr (QP > 7) nk_chain<7>(acc, v);
    };
    const bool stamp = DIAG && blockIdx.x == 0;
    unsigned long long tw = 0, tb = 0;
    // (p0, p1) pending on entry; (q0 .. q3) receive this round; (q2, q3) are pending on exit
    auto one_round = [&](float4 (&p0)[QP], float4 (&p1)[QP], float4 (&q0)[QP], float4 (&q1)[QP], float4 (&q2)[QP],
                         float4 (&q3)[QP], int r) {
        unsigned long long t0 = 0, t1 = 0;
        if (stamp) t0 = sim_now();
        const float4 *in = ring + ((r - 1) & 1) * NK_QUADS * 64 + lane;
        rd(q0, in, 0);
        rd(q1, in, 1);
        rd(q2, in, 2);
        rd(q3, in, 3);
        __builtin_amdgcn_sched_barrier(0);
        add(p0);
        add(p1);
        add(q0);
        add(q1);
        if (stamp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t1 = sim_now();
        }
        sim_barrier();  // also waits for the q2 / q3 reads: their buffer is rewritten two rounds on
        if (stamp) {
            tw += t1 - t0;
            tb += sim_now() - t1;
        }
    };
    for (int r = 1; r + 2 <= rounds; r += 3) {  // the launcher makes `rounds` a multiple of 3
        one_round(sa, sb, sc, sd, se, sf, r);
        one_round(se, sf, sa, sb, sc, sd, r + 1);
        one_round(sc, sd, se, sf, sa, sb, r + 2);
    }
    add(sa);  // the two quarters still pending
    add(sb);
    if (stamp && lane == 0) {
        g_sim_stamps[0] = tw;
        g_sim_stamps[1] = tb;
        g_sim_stamps[2] = (unsigned long long)rounds;
    }
    if (c < n) num_out[c] = acc;
}

// RESIDENT = true: the codes stay in registers (m <= 4032).  RESIDENT = false: any m; the codes are fetched two
// rounds ahead into three rotating register sets (16 B per lane and round: a quarter of the bytes of the streaming
// similarity_pc_kernel), everything else is shared.
template <bool DIAG, int RM>  // RM = resident rounds (18 or 36), 0 = streaming codes
__device__ __forceinline__ void nk_producer(const int P, unsigned char *smem, const uint2 *__restrict__ codes8, int m,
                                            int64_t ld, const float *__restrict__ wmat, int ldw, int npos, int lane,
                                            int c, int R, int pad, int rounds) {
    const float *master = reinterpret_cast<const float *>(smem);
    float4 *ring = reinterpret_cast<float4 *>(smem + NK_RING_OFF);
    const int G8 = (m + 7) >> 3;
    const uint2 *col = codes8 + c;
    constexpr bool RESIDENT = RM > 0;

    auto fetch_codes = [&](uint4 &u, int q) {  // one uint4 = the 16 codes of this producer's round; row G8 is all-skipped
        const int g0 = q * NK_ROUND_OCTS + P * NK_OCTS, g1 = g0 + 1;
        const uint2 lo = col[(size_t)(g0 >= G8 ? G8 : g0) * ld], hi = col[(size_t)(g1 >= G8 ? G8 : g1) * ld];
        u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    uint4 cod[RESIDENT ? RM : 3];  // this producer's codes: all of them, or three rounds' worth
    if (RESIDENT) {
#pragma unroll
        for (int q = 0; q < (RESIDENT ? RM : 3); ++q) fetch_codes(cod[q], q);
    }
    auto load_cj = [&](int jn) -> uint32_t {  // table row of this lane's residue in row jn
        if (jn >= m - 1) return (uint32_t)npos;
        const uint8_t *cj = reinterpret_cast<const uint8_t *>(col + (size_t)(jn >> 3) * ld);
        return (uint32_t)cj[jn & 7] - (uint32_t)NK_K;
    };
    auto refresh = [&](int jn, uint32_t idx) {
        float *sl = reinterpret_cast<float *>(smem + ((jn & 1) ? NK_SLICE1_OFF : NK_SLICE0_OFF)) + lane;
        const float *mrow = master + idx * NK_MASTER_LD;
        for (int e = P; e <= npos; e += SIM_NP) sl[e * 64] = mrow[e];
    };
    auto q0_of = [&](int j) { return j < m - 1 ? ((j + 1) >> 3) / NK_ROUND_OCTS : R - 1; };
    int r = 0;
    refresh(0, load_cj(0));
    uint32_t cj_next = load_cj(1);
    sim_barrier();  // slice[0] complete

    float tv[NK_OCTS][8];
    static_assert(NK_OCTS == 2, "16 W values per producer and round");
    f32x4 wq[4];
    const bool stamp = DIAG && blockIdx.x == 0;
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0}, tg = 0;
    if (stamp) tg = sim_now();
    // W[j][8 g .. 8 g + 15] of this producer's round is wave-uniform.  It is DMA'd into a private LDS line
    // (global_load_lds_dword: lane i carries element i & 15, no VGPR in flight) ONE round ahead -- within a
    // row the source is the row base plus a compile-time offset -- and read back as four broadcast
    // ds_read_b128 behind the gathers.  (Scalar loads would be cheaper here, but they thrash the scalar
    // cache that the denominator kernel on the neighbouring CUs lives on: measured 10.5 -> 15 ms there.)
    // Reads past the row end (null octs) land in the zero padding or the next row: finite, and multiplied
    // by zero table entries.
    const uint32_t wstage_base = (uint32_t)(NK_WSTAGE_OFF + P * 512);
    const uint32_t lane15x4 = (uint32_t)(lane & 15) * 4u;
    auto wrow_of = [&](int jj) {
        const int jr = jj < m - 1 ? jj : m - 1;  // the pseudo-rows use the all-zero W row m-1
        return wmat + ((size_t)jr * (size_t)ldw + (size_t)(P * NK_OCTS * 8));
    };
    {
        const float *src = wrow_of(0) + q0_of(0) * (NK_ROUND_OCTS * 8);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(wstage_base), "v"(lane15x4), "s"(src) : "m0", "memory");
    }
    // One round: cq = this round's codes; wsrc_next = W source of the next round; last_of_row: the table slice of
    // row j+1 is staged at the end; prefetch(): further loads for the next round, issued after the wait.
    // v_perm_b32 selectors {lane.b3, lane.b2, code.b[k], lane.b0}, kept in VGPRs (as SGPR constants they
    // get spilled to VGPR lanes and cost a v_readlane each per round)
    uint32_t selv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(selv[k]) : "s"(0x03020400u + ((uint32_t)k << 8)));
    auto round_work = [&](const uint4 &cq, int j, const float *wsrc_next, bool last_of_row,
                          auto &&prefetch) __attribute__((always_inline)) {
        // This round's W (DMA'd a round ago) has landed.  When streaming, the two code loads of the NEXT round,
        // issued after that DMA, may still be in flight (vmcnt counts in order).
        if (RESIDENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        const uint32_t wdst = wstage_base + (uint32_t)((r + 1) & 1) * 256u;
        // (the instruction's immediate offset would also move the LDS address: the source is a full pointer)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(wdst), "v"(lane15x4), "s"(wsrc_next) : "m0", "memory");
        prefetch();
        // address bytes: [lane * 4][code][row parity][0]
        const uint32_t vlane = (uint32_t)lane * 4u + ((uint32_t)(j & 1) << 16);
        const uint32_t cw[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t addr = __builtin_amdgcn_perm(cw[d], vlane, selv[k]);
                asm volatile("ds_read_b32 %0, %1" : "=v"(tv[d >> 1][(d & 1) * 4 + k]) : "v"(addr));
            }
        const uint32_t waddr = wstage_base + (uint32_t)(r & 1) * 256u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[i]) : "v"(waddr), "i"(16 * i));
        float4 *out = ring + ((r & 1) * NK_QUADS + P * NK_OCTS * 2) * 64 + lane;
        // the asm reads above are tied to the wait so that no use (not even a copy) is scheduled ahead of it
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(tv[0][0]), "+v"(tv[0][1]),
                       "+v"(tv[0][2]), "+v"(tv[0][3]), "+v"(tv[0][4]), "+v"(tv[0][5]), "+v"(tv[0][6]),
                       "+v"(tv[0][7]), "+v"(tv[1][0]), "+v"(tv[1][1]), "+v"(tv[1][2]), "+v"(tv[1][3]),
                       "+v"(tv[1][4]), "+v"(tv[1][5]), "+v"(tv[1][6]), "+v"(tv[1][7])
                     :
                     : "memory");
        unsigned long long td = 0;
        if (stamp) td = sim_now();
#pragma unroll
        for (int t = 0; t < NK_OCTS; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // 4 steps: two packed multiplies (separate rounding), one 16-B ring store
                const f32x4 w = wq[2 * t + h];
                const f32x2 xa = f32x2{tv[t][4 * h], tv[t][4 * h + 1]} * f32x2{w.x, w.y};
                const f32x2 xb = f32x2{tv[t][4 * h + 2], tv[t][4 * h + 3]} * f32x2{w.z, w.w};
                out[(t * 2 + h) * 64] = make_float4(xa.x, xa.y, xb.x, xb.y);
            }
        if (last_of_row) {  // stage the table slice of row j+1
            refresh(j + 1, cj_next);
            cj_next = load_cj(j + 2);
        }
        if (stamp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long tf = sim_now();
            sim_barrier();
            const unsigned long long tn = sim_now();
            acc_t[0] += td - tg;  // barrier exit -> gathers and W landed
            acc_t[1] += tf - td;  // multiplies, ring stores (drained), slice refresh
            acc_t[2] += tn - tf;  // barrier wait
            tg = tn;
        } else {
            sim_barrier();
        }
        ++r;
    };
    const int nrows = m - 1 + pad;
    if constexpr (RESIDENT) {
        (void)rounds;
        for (int jj = 0; jj < nrows; ++jj) {
            const int q0 = q0_of(jj);
            const float *wrow = wrow_of(jj);
            nk_unroll<0, RM>([&](auto qc) __attribute__((always_inline)) {
                constexpr int Q = decltype(qc)::value;
                if (Q >= q0 && Q < R) {
                    const bool last = Q == R - 1;
                    const float *wsrc_next = last ? wrow_of(jj + 1) + q0_of(jj + 1) * (NK_ROUND_OCTS * 8)
                                                  : wrow + (Q + 1) * (NK_ROUND_OCTS * 8);
                    round_work(cod[Q], jj, wsrc_next, last, [] {});
                }
            });
        }
    } else {
        // (row, round) sequence: row jj runs rounds q0(jj) .. R-1.  Codes are fetched TWO rounds ahead (an L2
        // miss takes longer than a round) into three rotating register sets.
        static_assert(NK_OCTS == 2, "the vmcnt(2) above counts the two 8-byte code loads of a round");
        struct Pos {
            int j, q;
        };
        auto next = [&](Pos p) { return p.q == R - 1 ? Pos{p.j + 1, q0_of(p.j + 1)} : Pos{p.j, p.q + 1}; };
        Pos pos = {0, q0_of(0)};
        Pos pos1 = next(pos);
        fetch_codes(cod[0], pos.q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // order: [codes(0)] [W(0) DMA above] -> then codes(1) is the young pair
        fetch_codes(cod[1], pos1.q);
        auto step = [&](const uint4 &cur, uint4 &far) __attribute__((always_inline)) {
            const Pos pos2 = next(pos1);
            const float *wsrc_next = wrow_of(pos1.j) + pos1.q * (NK_ROUND_OCTS * 8);
            round_work(cur, pos.j, wsrc_next, pos.q == R - 1, [&] { fetch_codes(far, pos2.q); });
            pos = pos1;
            pos1 = pos2;
        };
        for (int rr = 0; rr < rounds; rr += 3) {
            step(cod[0], cod[2]);
            if (rr + 1 < rounds) step(cod[1], cod[0]);
            if (rr + 2 < rounds) step(cod[2], cod[1]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the W prefetch past the end
    if (stamp && lane == 0)
        for (int k = 0; k < 5; ++k) g_sim_stamps[(P + 1) * 8 + k] = acc_t[k];
    sim_barrier();  // the consumer's drain round
}


// ------------------------------------------------------------------------------------------
// Transposed producers (tiles of 64 columns, resident codes).  The numerator kernel is bound by LDS cycles, and
// in the layout above a producer lane is a column: 4 steps of a column leave as one ds_write_b128 (13 LDS
// cycles: a store moves its 5 dwords per lane to the LDS at 2-3 cycles each) and W arrives as four broadcast
// ds_read_b128.  Here lane p of a producer works, in register (w, b), on column 16 w + p / 4 at step
// 4 b + p % 4 of the producer's 16 steps: a quad of the ring ([column][4 steps] x 4 B = 1 KB) is then four
// 256-byte pieces whose dword index is the lane number -- ds_write_addtid_b32 (address = M0 + offset + 4 lane, no
// address VGPR, 2 LDS cycles) stores each -- and the 4 W values a lane needs are one ds_read_b128 of a line the
// DMA fills already permuted.  Per 112-step round: 112 gathers x 2 + 7 W reads x 4 + 112 stores x 2 + 28 ring
// reads x 4 = 588 LDS cycles instead of 812.  The price: the table slices hold every entry four times
// ([entry][column][step % 4], 1 KB per entry, so that the four lanes of a column hit four banks), and the
// ring must lie below 64 KB (M0 carries 16 address bits).  The consumer is unchanged.
// LDS: master | ring 2 x 28 KB | slice 0 (29 KB) | W stage | ... | slice 1 (64 KB above slice 0).
// A code byte is 4 x the table entry: byte 1 of the gather address = entry x 1 KB.
// ------------------------------------------------------------------------------------------
constexpr int TP_RING_OFF = 4096;
constexpr int TP_SLICE_OFF = TP_RING_OFF + NK_RING_BYTES;  // 61440
constexpr int TP_SLICE_BYTES = 29 * 1024;
constexpr int TP_WSTAGE_OFF = TP_SLICE_OFF + TP_SLICE_BYTES;  // 91136; per producer 2 x 256 B
constexpr int TP_SLICE1_OFF = TP_SLICE_OFF + 65536;           // 126976
__host__ __device__ constexpr int tp_lds_bytes() { return TP_SLICE1_OFF + TP_SLICE_BYTES; }  // 156672
static_assert(NK_MASTER_BYTES <= TP_RING_OFF && TP_RING_OFF + NK_RING_BYTES <= 65536 - 4096 &&
                  TP_WSTAGE_OFF + SIM_NP * 512 <= TP_SLICE1_OFF && TP_SLICE_OFF + 768 < 65536 && tp_lds_bytes() <= 160 * 1024,
              "LDS layout of the transposed producers");

template <bool DIAG, int RM>  // RM = resident rounds (18 or 36), 0 = codes streamed two rounds ahead
__device__ __forceinline__ void nk_producer_tp(const int P, unsigned char *smem, const uint4 *__restrict__ codesT, int ntiles,
                                               int m, const float *__restrict__ wmat, int ldw, int npos, int lane, int R,
                                               int pad, int rounds) {
    const float *master = reinterpret_cast<const float *>(smem);
    const int GG = (m + 15) >> 4;
    constexpr bool RESIDENT = RM > 0;
    const uint4 *mine = codesT + (size_t)blockIdx.x * 64 + lane;  // + group * ntiles * 64
    auto fetch_codes = [&](uint4 &u, int q) {  // round q of this producer = row group 7 q + P
        const int g = q * SIM_NP + P;
        u = mine[(size_t)(g >= GG ? GG : g) * (size_t)ntiles * 64u];
    };
    uint4 cod[RESIDENT ? RM : 3];  // this producer's codes: all of them, or three rounds' worth
    if (RESIDENT) {
#pragma unroll
        for (int q = 0; q < (RESIDENT ? RM : 3); ++q) fetch_codes(cod[q], q);
    }
    // for the slice refresh a lane is a column: its code of row jn sits in lane 4 (lane % 16) + jn % 4 of the tile's
    // group jn / 16, word lane / 16, byte (jn % 16) / 4
    const uint8_t *tile_bytes = reinterpret_cast<const uint8_t *>(codesT + (size_t)blockIdx.x * 64);
    auto load_cj = [&](int jn) -> uint32_t {
        if (jn >= m - 1) return (uint32_t)npos;
        const int r16 = jn & 15;
        const size_t off = ((size_t)(jn >> 4) * (size_t)ntiles * 64u + (size_t)(4 * (lane & 15) + (r16 & 3))) * 16u +
                           (size_t)((lane >> 4) * 4 + (r16 >> 2));
        return (uint32_t)tile_bytes[off] >> 2;
    };
    auto refresh = [&](int jn, uint32_t idx) {
        float4 *sl = reinterpret_cast<float4 *>(smem + ((jn & 1) ? TP_SLICE1_OFF : TP_SLICE_OFF)) + lane;
        const float *mrow = master + idx * NK_MASTER_LD;
        for (int e = P; e <= npos; e += SIM_NP) {
            const float v = mrow[e];
            sl[e * 64] = make_float4(v, v, v, v);
        }
    };
    auto q0_of = [&](int j) { return j < m - 1 ? ((j + 1) >> 3) / NK_ROUND_OCTS : R - 1; };
    int r = 0;
    refresh(0, load_cj(0));
    uint32_t cj_next = load_cj(1);
    sim_barrier();  // slice[0] complete

    const bool stamp = DIAG && blockIdx.x == 0;
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0}, tg = 0;
    if (stamp) tg = sim_now();
    // W[j][k0 .. k0 + 15] of this producer's round, DMA'd one round ahead: DMA lane 4 i + b carries element
    // 4 b + i, so that the line reads back as [i][b] and lane p takes its four values W[4 b + p % 4] in one read
    const uint32_t wstage_base = (uint32_t)(TP_WSTAGE_OFF + P * 512);
    const uint32_t dma_off = (uint32_t)(((lane & 3) * 4 + ((lane >> 2) & 3)) * 4);
    const uint32_t wread_off = (uint32_t)(lane & 3) * 16u;
    const uint32_t ring_m0 = (uint32_t)(TP_RING_OFF + P * NK_OCTS * 2 * 1024);
    auto wrow_of = [&](int jj) {
        const int jr = jj < m - 1 ? jj : m - 1;
        return wmat + ((size_t)jr * (size_t)ldw + (size_t)(P * NK_OCTS * 8));
    };
    {
        const float *src = wrow_of(0) + q0_of(0) * (NK_ROUND_OCTS * 8);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(wstage_base), "v"(dma_off), "s"(src) : "m0", "memory");
    }
    uint32_t selv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(selv[k]) : "s"(0x03020400u + ((uint32_t)k << 8)));
    auto round_work = [&](const uint4 &cq, int j, const float *wsrc_next, bool last_of_row,
                          auto &&prefetch) __attribute__((always_inline)) {
        // This round's W (DMA'd a round ago) has landed.  When streaming, the code load of the round after next,
        // issued behind that DMA, may still be in flight (vmcnt counts in order).
        if (RESIDENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        const uint32_t wdst = wstage_base + (uint32_t)((r + 1) & 1) * 256u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(wdst), "v"(dma_off), "s"(wsrc_next) : "m0", "memory");
        prefetch();
        const uint32_t vlane = (uint32_t)lane * 4u + ((uint32_t)(j & 1) << 16);  // [lane * 4][code][row parity][0]
        const uint32_t cw[4] = {cq.x, cq.y, cq.z, cq.w};
        float tv[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t addr = __builtin_amdgcn_perm(cw[w], vlane, selv[b]);
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tv[w][b]) : "v"(addr), "i"(TP_SLICE_OFF + 256 * w));
            }
        f32x4 wq;
        const uint32_t waddr = wstage_base + (uint32_t)(r & 1) * 256u + wread_off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wq) : "v"(waddr));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(wq), "+v"(tv[0][0]), "+v"(tv[0][1]), "+v"(tv[0][2]), "+v"(tv[0][3]), "+v"(tv[1][0]),
                       "+v"(tv[1][1]), "+v"(tv[1][2]), "+v"(tv[1][3]), "+v"(tv[2][0]), "+v"(tv[2][1]), "+v"(tv[2][2]),
                       "+v"(tv[2][3]), "+v"(tv[3][0]), "+v"(tv[3][1]), "+v"(tv[3][2]), "+v"(tv[3][3])
                     :
                     : "memory");
        unsigned long long td = 0;
        if (stamp) td = sim_now();
        f32x2 xa[4], xb[4];  // two packed multiplies per register set (separate rounding of every product)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            xa[w] = f32x2{tv[w][0], tv[w][1]} * f32x2{wq.x, wq.y};
            xb[w] = f32x2{tv[w][2], tv[w][3]} * f32x2{wq.z, wq.w};
        }
        const uint32_t m0v = ring_m0 + (uint32_t)(r & 1) * (uint32_t)(NK_QUADS * 1024);
#define TP_ST(w, b, reg) "ds_write_addtid_b32 " reg " offset:" #b "*1024+" #w "*256\n\t"
        // (an SALU write of M0 needs one wait state before an add-TID LDS instruction reads it)
        asm volatile("s_mov_b32 m0, %16\n\ts_nop 0\n\t"
                     TP_ST(0, 0, "%0") TP_ST(0, 1, "%1") TP_ST(0, 2, "%2") TP_ST(0, 3, "%3")
                     TP_ST(1, 0, "%4") TP_ST(1, 1, "%5") TP_ST(1, 2, "%6") TP_ST(1, 3, "%7")
                     TP_ST(2, 0, "%8") TP_ST(2, 1, "%9") TP_ST(2, 2, "%10") TP_ST(2, 3, "%11")
                     TP_ST(3, 0, "%12") TP_ST(3, 1, "%13") TP_ST(3, 2, "%14") TP_ST(3, 3, "%15")
                     :
                     : "v"(xa[0].x), "v"(xa[0].y), "v"(xb[0].x), "v"(xb[0].y), "v"(xa[1].x), "v"(xa[1].y), "v"(xb[1].x),
                       "v"(xb[1].y), "v"(xa[2].x), "v"(xa[2].y), "v"(xb[2].x), "v"(xb[2].y), "v"(xa[3].x), "v"(xa[3].y),
                       "v"(xb[3].x), "v"(xb[3].y), "s"(m0v)
                     : "m0", "memory");
#undef TP_ST
        if (last_of_row) {  // stage the table slice of row j+1
            refresh(j + 1, cj_next);
            cj_next = load_cj(j + 2);
        }
        if (stamp) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long tf = sim_now();
            sim_barrier();
            const unsigned long long tn = sim_now();
            acc_t[0] += td - tg;
            acc_t[1] += tf - td;
            acc_t[2] += tn - tf;
            tg = tn;
        } else {
            sim_barrier();
        }
        ++r;
    };
    const int nrows = m - 1 + pad;
    if constexpr (RESIDENT) {
        (void)rounds;
        for (int jj = 0; jj < nrows; ++jj) {
            const int q0 = q0_of(jj);
            const float *wrow = wrow_of(jj);
            nk_unroll<0, RM>([&](auto qc) __attribute__((always_inline)) {
                constexpr int Q = decltype(qc)::value;
                if (Q >= q0 && Q < R) {
                    const bool last = Q == R - 1;
                    const float *wsrc_next = last ? wrow_of(jj + 1) + q0_of(jj + 1) * (NK_ROUND_OCTS * 8)
                                                  : wrow + (Q + 1) * (NK_ROUND_OCTS * 8);
                    round_work(cod[Q], jj, wsrc_next, last, [] {});
                }
            });
        }
    } else {
        // (row, round) sequence: row jj runs rounds q0(jj) .. R-1.  Codes are fetched TWO rounds ahead (an L2 miss
        // takes longer than a round) into three rotating register sets.
        struct Pos {
            int j, q;
        };
        auto next = [&](Pos p) { return p.q == R - 1 ? Pos{p.j + 1, q0_of(p.j + 1)} : Pos{p.j, p.q + 1}; };
        Pos pos = {0, q0_of(0)};
        Pos pos1 = next(pos);
        fetch_codes(cod[0], pos.q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // order: [codes(0)] [W(0) DMA above] -> codes(1) is the young one
        fetch_codes(cod[1], pos1.q);
        auto step = [&](const uint4 &cur, uint4 &far) __attribute__((always_inline)) {
            const Pos pos2 = next(pos1);
            const float *wsrc_next = wrow_of(pos1.j) + pos1.q * (NK_ROUND_OCTS * 8);
            round_work(cur, pos.j, wsrc_next, pos.q == R - 1, [&] { fetch_codes(far, pos2.q); });
            pos = pos1;
            pos1 = pos2;
        };
        for (int rr = 0; rr < rounds; rr += 3) {
            step(cod[0], cod[2]);
            if (rr + 1 < rounds) step(cod[1], cod[0]);
            if (rr + 2 < rounds) step(cod[2], cod[1]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the W prefetch past the end
    if (stamp && lane == 0)
        for (int k = 0; k < 5; ++k) g_sim_stamps[(P + 1) * 8 + k] = acc_t[k];
    sim_barrier();  // the consumer's drain round
}

template <bool DIAG, int RM, bool TP = false>
__global__ __launch_bounds__(512) void similarity_num_kernel(
    const uint2 *__restrict__ codes8, int m, int n, int64_t ld, const float *__restrict__ wmat, int ldw,
    const f32x2 *__restrict__ tab_g, int npos, int R, int pad, int rounds, float *__restrict__ num_out, int tcols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * tcols + lane;
    const bool active = lane < tcols && c < ld;
    {
        float *master = reinterpret_cast<float *>(smem);
        for (int t = threadIdx.x; t < 29 * 32; t += 512) master[(t >> 5) * NK_MASTER_LD + (t & 31)] = tab_g[t].x;
    }
    __syncthreads();
    if constexpr (TP) {  // 64-column tiles, codes in the transposed layout: every lane works (a producer lane is not a column)
        if (wave != 0) {
            __builtin_amdgcn_s_setprio(2);
            nk_producer_tp<DIAG, RM>(wave - 1, smem, reinterpret_cast<const uint4 *>(codes8), (int)gridDim.x, m, wmat, ldw,
                                     npos, lane, R, pad, rounds);
        } else {
            nk_consumer<DIAG, TP_RING_OFF>(smem, rounds, lane, c, n, num_out);
        }
    } else if (wave != 0) {
        // the chain wave has slack every round; the producer that shares its SIMD does not
        __builtin_amdgcn_s_setprio(2);
        if (active) nk_producer<DIAG, RM>(wave - 1, smem, codes8, m, ld, wmat, ldw, npos, lane, c, R, pad, rounds);
    } else if (active) {
        nk_consumer<DIAG, NK_RING_OFF>(smem, rounds, lane, c, n, num_out);
    }
}

// MDK from the two sums (Similarity::calculateVectors tail): 0 for >= 80 % gaps or an empty denominator, else
// min(1, (float)exp(-(double)Q)).  Q = num / den is bit-exact; the exponential is the device library's, which may differ
// from the host's in the last place of the DOUBLE -- and then in the float only when the double lies within a few of
// its own ulps of a point where the conversion to float changes its result.  Such a value (and one in the float
// denormal range, where flush modes could differ) is not trusted: it goes out as a NaN and the host evaluates
// (float)exp(-(double)Q) itself (fetch_similarity_finish).  Both libraries are accurate to an ulp, so every value that
// passes the test rounds to the same float on both sides: MDK is bit-identical to the host computation by construction.
__global__ __launch_bounds__(256) void sim_finish_kernel(const float *__restrict__ num, const float *__restrict__ den,
                                                         const int32_t *__restrict__ gaps_w, int m, int n,
                                                         float *__restrict__ q_out, float *__restrict__ mdk_out, int all_on_host) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const bool skip = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
    float q = 0.0f, v = 0.0f;
    const float d = den[c];
    if (!skip && d != 0.0f) {
        q = num[c] / d;
        const double e = exp(-(double)q);
        v = (float)e;
        bool safe = e >= 1e-37 && !all_on_host;
        if (safe) {
            const double up = (double)__uint_as_float(__float_as_uint(v) + 1u), dn = (double)__uint_as_float(__float_as_uint(v) - 1u);
            const double tol = e * 0x1p-49;  // eight ulps of the double
            safe = (0.5 * (up + (double)v) - e) > tol && (e - 0.5 * (dn + (double)v)) > tol;
        }
        v = safe ? (v > 1.0f ? 1.0f : v) : __uint_as_float(0x7FC00000u);
    }
    if (q_out) q_out[c] = q;
    mdk_out[c] = v;
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
                                                         int32_t *__restrict__ row_nongap, uint32_t *__restrict__ used) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    ByteSet seen;  // used != nullptr: this pass collects the byte values instead of gap_counts (see there)
    if (row < m) {
        // 16 bytes per lane and load (rows are 64-byte aligned, ld % 64 == 0; keep_res has 64 bytes of slack); bytes at
        // or past n are masked off.  keep_res == nullptr: every column counts.
        const uint4 *p = reinterpret_cast<const uint4 *>(raw + (size_t)row * ld);
        const uint4 *k = reinterpret_cast<const uint4 *>(keep_res);
        int cnt = 0;
        for (int q = lane; q * 16 < n; q += 64) {
            const uint4 x = p[q], kk = k ? k[q] : make_uint4(~0u, ~0u, ~0u, ~0u);
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ks[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int left = n - (q * 16 + w * 4);  // bytes of this word inside the row
                const uint32_t inside = left >= 4 ? 0x80808080u : (left <= 0 ? 0u : (0x80808080u >> (8 * (4 - left))));
                const uint32_t kept = ~zero_bytes(ks[w]) & 0x80808080u, gap = zero_bytes(xs[w] ^ 0x2d2d2d2du);
                cnt += __popc(kept & ~gap & inside);
                if (used) byteset_add(seen, xs[w], (inside >> 7) * 0xFFu);  // (uniform)
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
        if (lane == 0) row_nongap[row] = cnt;
    }
    if (used) byteset_publish(seen, used, blockIdx.x);
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
__global__ __launch_bounds__(256) void row_digest_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                         int32_t *__restrict__ lengths,
                                                         unsigned long long *__restrict__ hashes) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
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
                                                           uint8_t *__restrict__ keep_seq, int32_t *__restrict__ count) {
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
        local += r;
    }
    if (local) atomicAdd(count, local);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline uint32_t rep4(uint8_t b) { return 0x01010101u * b; }

// ---- diagnostic switches ---------------------------------------------------------------------
static thread_local const Tuning *tl_tuning = nullptr;
void set_tuning(const Tuning *t) { tl_tuning = t; }
const Tuning &tuning() {
    static const Tuning defaults;
    return tl_tuning ? *tl_tuning : defaults;
}
Tuning tuning_from_env() {
    Tuning t;
    auto num = [](const char *name, int dflt) {
        const char *e = getenv(name);
        return e ? atoi(e) : dflt;
    };
    if (const char *k = getenv("MSA_SIM_KERNEL")) t.sim_kernel = k[0] == 'c' ? 1 : (k[0] == 'p' ? 2 : (k[0] == 'b' ? 3 : (k[0] == 'l' ? 4 : (k[0] == 'q' ? 5 : 0))));
    t.sim_tcols = num("MSA_SIM_TCOLS", 0);
    t.sim_mode = num("MSA_SIM_MODE", 0);
    t.sim_tp = num("MSA_SIM_TP", 1);
    t.den_waves = num("MSA_DEN_WAVES", 0);
    if (const char *k = getenv("MSA_DEN_KERNEL")) t.den_exec = k[0] == 'e';
    t.sim_serial = getenv("MSA_SIM_SERIAL") != nullptr;
    t.device_clusters = num("MSA_DEVICE_CLUSTERS", -1);
    t.trace = getenv("MSA_TRACE") != nullptr;
    t.pipeline = num("MSA_PIPELINE", 1);
    t.upload_piece_kb = num("MSA_UPLOAD_PIECE_KB", 1024);
    t.bx_cols = num("MSA_BX_COLS", 0);
    t.bx_r0 = num("MSA_BX_R0", -1);
    t.bx_waves = num("MSA_BX_WAVES", 0);
    t.pair_ti = num("MSA_PAIR_TI", 0);
    t.pair_xcd = num("MSA_PAIR_XCD", 1);
    t.pair_pipe = num("MSA_PAIR_PIPE", 1);
    t.pair_dense = num("MSA_PAIR_DENSE", 1);
    t.bx_compact = num("MSA_BX_COMPACT", 0);
    t.bx_asm = num("MSA_BX_ASM", 0);
    t.lg_regs = num("MSA_LG_REGS", 0);
    t.lg_dbg = num("MSA_LG_DBG", 0);
    t.mdk_host = num("MSA_MDK_HOST", 0);
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
                        int nchunk, int m_pad, int *err_flag, const uint32_t *used_slots, uint32_t *used_out) {
    dim3 grid((m_pad + 255) / 256, (nchunk + 1) / 2);
    prep_planes_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, rep4(indet), planes, nchunk, m_pad, err_flag, used_slots, used_out, 0);
}
int used_slot_words() { return 4 * USED_SLOTS; }
// Dense codes pay from about 1500 sequences on: collecting the byte values (in gap_counts, or in the row-totals pass of
// a trim that needs no gap counts) and writing two sets of code planes cost ~20 us, more than the pair pass gains
// below that (0.344 -> 0.355 ms per trim at 500 x 2000; 3.22 -> 3.19 ms at 2000 x 10000).
// MSA_PAIR_DENSE=2 forces dense codes at any size, 0 never (profiles/r02_ab_switches.txt, r02_pairs_time.jsonl).
bool pair_dense(int m) {
    if (tuning().pair_ti == 16 || tuning().pair_ti == 32) return false;
    return tuning().pair_dense == 2 || (tuning().pair_dense == 1 && m >= 1500);
}
int planes_total() { return PLANES_TOTAL; }

void launch_gap_counts(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, int32_t *gaps,
                       int32_t *indets, uint32_t *used) {
    dim3 grid((unsigned)((ld / 4 + 255) / 256), (m + GAP_SLAB - 1) / GAP_SLAB);
    gap_counts_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, rep4(indet), gaps, indets, used);
}

void launch_pair_counts(hipStream_t s, const uint32_t *planes, int nchunk, int m_pad, int m, int ldw, uint32_t *hit,
                        uint32_t *dst, float *ident, float *wmat, float *wlow, int *undef_flag, const uint32_t *used) {
    if (used) {
        // dense codes (prep_planes was given the same `used`): 8 rows i x 64 rows j per wave, the triangle's tiles only
        const int nib = (m + PAIR_TI - 1) / PAIR_TI, njb = m_pad / 64;
        const int R = 64 / PAIR_TI, jc = (nib + R - 1) / R - 1;
        const unsigned tiles = (unsigned)(R * jc * (jc + 1) / 2 + (njb - jc) * nib);
        pair_counts_dense_kernel<<<tiles, 64, 0, s>>>(planes, nchunk, m_pad, m, ldw, hit, dst, ident, wmat, wlow, undef_flag, nib, used);
        return;
    }
    // Two rows "j" per lane reuse the wave-uniform "i" words twice, but halve the number of waves: worth it only
    // once the upper triangle still holds several waves per SIMD (m >= ~3000); m_pad is a multiple of 128.
    const int ti = tuning().pair_ti == 16 || tuning().pair_ti == 32 ? tuning().pair_ti : PAIR_TI;
    const long waves2 = (long)((m + ti - 1) / ti) * (m_pad / 128) / 2;
    const bool two = waves2 >= 8192;
    const int nib = (m + ti - 1) / ti, njb = two ? m_pad / 128 : m_pad / 64;
    const bool xcd = tuning().pair_xcd != 0;
    dim3 grid(nib, njb);
    // Below the two-rows-per-lane size the kernel is short of waves (three to four per SIMD at m = 2000): there the
    // triangle-only grid and the software-pipelined loop pay (0.46 -> 0.37 ms at 2000 x 10000, 0.088 -> 0.074 ms at
    // 1000 x 4000); with TJ = 2 the plain loop is 4 % faster (tools/pairs_time.py).
    const bool lean = !two && ti == PAIR_TI;
    if (xcd && lean) {
        const int R = 64 * (two ? 2 : 1) / ti, jc = (nib + R - 1) / R - 1;
        grid = dim3((unsigned)(R * jc * (jc + 1) / 2 + (njb - jc) * nib), 1);
    }
    const int nib_arg = xcd && lean ? nib : 0;
    if ((lean && tuning().pair_pipe != 0) || (ti == PAIR_TI && tuning().pair_pipe == 2)) {  // (2: also with TJ = 2 -- diagnostics)
        if (two) pair_counts_pipe_kernel<2><<<grid, 64, 0, s>>>(planes, nchunk, m_pad, m, ldw, hit, dst, ident, wmat, wlow, undef_flag, nib_arg);
        else pair_counts_pipe_kernel<1><<<grid, 64, 0, s>>>(planes, nchunk, m_pad, m, ldw, hit, dst, ident, wmat, wlow, undef_flag, nib_arg);
        return;
    }
#define PAIR_LAUNCH(TI_, TJ_) \
    pair_counts_kernel<TI_, TJ_><<<grid, 64, 0, s>>>(planes, nchunk, m_pad, m, ldw, hit, dst, ident, wmat, wlow, undef_flag, nib_arg)
    if (ti == 32) {
        if (two) PAIR_LAUNCH(32, 2);
        else PAIR_LAUNCH(32, 1);
    } else if (ti == 16) {
        if (two) PAIR_LAUNCH(16, 2);
        else PAIR_LAUNCH(16, 1);
    } else {
        if (two) PAIR_LAUNCH(8, 2);
        else PAIR_LAUNCH(8, 1);
    }
#undef PAIR_LAUNCH
}

void launch_sim_encode8(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, int npos,
                        const int32_t *gaps_w, void *codes8, unsigned long long *err_key, int tcols);
bool sim_num_transposed(int tcols);

// Columns per similarity workgroup: a full wave.  The kernel's time is (pair steps) x (cycles per step)
// whatever the column count, and the LDS time per instruction does not depend on the active lanes, so
// narrower tiles spread over more CUs buy nothing (13.8 vs 14.0 ms at 2000 x 10000) and cost a batch of
// concurrent alignments its parallelism.  MSA_SIM_TCOLS overrides (tests exercise ragged tiles with it).
int sim_tile_cols(int n, int cus, int min_cols) {
    (void)n;
    (void)cus;
    if (tuning().sim_tcols >= min_cols && tuning().sim_tcols <= 64) return tuning().sim_tcols;
    return 64;
}
int sim_num_min_cols() { return 16; }

void launch_sim_encode32(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, int npos,
                         const int32_t *gaps_w, void *codes32, unsigned long long *err_key, int tcols) {
    dim3 grid((unsigned)((ld + 255) / 256), (m + 7) / 8 + 1);
    sim_encode32_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, lut, npos, gaps_w, reinterpret_cast<uint4 *>(codes32),
                                             err_key, tcols);
}

// MSA_SIM_MODE (diagnostics only, never set in production): bit0 producers skip gather/emit,
// bit1 consumer skips the chain
static int sim_debug_mode() {
    return tuning().sim_mode;
}

extern "C" int msa_debug_den_ticks(unsigned long long *out1024) {
    return (int)hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_den_ticks), sizeof(unsigned long long) * 1024);
}

extern "C" int msa_debug_sim_stamps(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_sim_stamps), sizeof(unsigned long long) * 64);
}

int launch_similarity_pc(hipStream_t s, const void *codes32, int m, int n, int64_t ld, const float *wmat, int ldw,
                         const void *tab, int npos, const int32_t *gaps_w, float *q_out, float *mdk_out, int tcols) {
    const int G8 = (m + 7) / 8;
    long long rounds = 0;  // row-aligned: every row j takes ceil(octs_j / ROUND_OCTS) rounds
    for (int j = 0; j + 1 < m; ++j) rounds += (G8 - ((j + 1) >> 3) + SIM_ROUND_OCTS - 1) / SIM_ROUND_OCTS;
    rounds = (rounds + 2) / 3 * 3;  // the consumer's register sets rotate with period 3; extra rounds are null
    const int lds = sim_lds_bytes(npos);
    const int mode = sim_debug_mode();
    auto kern = mode ? similarity_pc_kernel<true> : similarity_pc_kernel<false>;
    if (int e = set_max_lds_once(reinterpret_cast<const void *>(kern), lds)) return e;
    kern<<<(n + tcols - 1) / tcols, 512, lds, s>>>(reinterpret_cast<const uint4 *>(codes32), m, n, ld, wmat, ldw,
                                                   reinterpret_cast<const f32x2 *>(tab), npos, gaps_w, (int)rounds, mode,
                                                   q_out, mdk_out, tcols);
    return 0;
}

// the numerator kernel keeps every producer's codes in its registers: that bounds the row count
bool similarity_rc_fits(int m) { return (m + 7) / 8 <= NK_RMAX * NK_ROUND_OCTS; }

// The numerator kernel's producers read the transposed layout on 64-column tiles (MSA_SIM_TP=0: the [oct][column]
// layout, which narrower tiles -- a diagnostic -- always use).
bool sim_num_transposed(int tcols) {
    return tuning().sim_tp != 0 && tcols == 64;
}

void launch_sim_encode8(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, int npos,
                        const int32_t *gaps_w, void *codes8, unsigned long long *err_key, int tcols) {
    if (sim_num_transposed(tcols)) {
        const int ntiles = (n + 63) / 64;
        dim3 grid((unsigned)((ntiles + 3) / 4), (m + 15) / 16 + 1);
        sim_encodeT_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, lut, npos, gaps_w, reinterpret_cast<uint4 *>(codes8), ntiles,
                                                err_key);
        return;
    }
    dim3 grid((unsigned)((ld + 255) / 256), (m + 7) / 8 + 1);
    sim_encode8_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, lut, npos, gaps_w, reinterpret_cast<uint2 *>(codes8), err_key,
                                            NK_K);
}

int launch_similarity_num(hipStream_t s, const void *codes8, int m, int n, int64_t ld, const float *wmat, int ldw,
                          const void *tab, int npos, float *num_out, int tcols) {
    const int G8 = (m + 7) / 8;
    const int R = (G8 + NK_ROUND_OCTS - 1) / NK_ROUND_OCTS;
    long long rounds = 0;
    for (int j = 0; j + 1 < m; ++j) rounds += R - ((j + 1) >> 3) / NK_ROUND_OCTS;
    const int pad = (int)((3 - rounds % 3) % 3);
    rounds += pad;
    // the request is rounded up to the denominator kernel's: no two chain workgroups -- of this launch, of
    // the denominator kernel or of another context's launches -- ever share a CU (and so a SIMD)
    const int lds = nk_lds_bytes() > DEN_LDS_BYTES ? nk_lds_bytes() : DEN_LDS_BYTES;
    const int octs = (m + 7) / 8;
    const int rm = octs <= 18 * NK_ROUND_OCTS ? 18 : (octs <= NK_RMAX * NK_ROUND_OCTS ? NK_RMAX : 0);
    const bool diag = (sim_debug_mode() & 64) != 0;
    if (sim_num_transposed(tcols)) {
        auto tk = rm == 18 ? (diag ? similarity_num_kernel<true, 18, true> : similarity_num_kernel<false, 18, true>)
                : rm == 0 ? (diag ? similarity_num_kernel<true, 0, true> : similarity_num_kernel<false, 0, true>)
                          : (diag ? similarity_num_kernel<true, NK_RMAX, true> : similarity_num_kernel<false, NK_RMAX, true>);
        if (int e = set_max_lds_once(reinterpret_cast<const void *>(tk), tp_lds_bytes())) return e;
        tk<<<(n + 63) / 64, 512, tp_lds_bytes(), s>>>(reinterpret_cast<const uint2 *>(codes8), m, n, ld, wmat, ldw,
                                                      reinterpret_cast<const f32x2 *>(tab), npos, R, pad, (int)rounds,
                                                      num_out, 64);
        return 0;
    }
    auto kern = rm == 18 ? (diag ? similarity_num_kernel<true, 18> : similarity_num_kernel<false, 18>)
              : rm == 0 ? (diag ? similarity_num_kernel<true, 0> : similarity_num_kernel<false, 0>)
                        : (diag ? similarity_num_kernel<true, NK_RMAX> : similarity_num_kernel<false, NK_RMAX>);
    if (int e = set_max_lds_once(reinterpret_cast<const void *>(kern), lds)) return e;
    kern<<<(n + tcols - 1) / tcols, 512, lds, s>>>(
        reinterpret_cast<const uint2 *>(codes8), m, n, ld, wmat, ldw, reinterpret_cast<const f32x2 *>(tab), npos, R, pad,
        (int)rounds, num_out, tcols);
    return 0;
}

void launch_sim_finish(hipStream_t s, const float *num, const float *den, const int32_t *gaps_w, int m, int n,
                       float *q_out, float *mdk_out) {
    sim_finish_kernel<<<(n + 255) / 256, 256, 0, s>>>(num, den, gaps_w, m, n, q_out, mdk_out, tuning().mdk_host);
}

void launch_overlap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                    const int32_t *indets, int need, uint32_t *col_ok, int nchunk, int32_t *good) {
    overlap_colmask_kernel<<<(nchunk * 32 + 255) / 256, 256, 0, s>>>(gaps, indets, m, n, need, col_ok, nchunk);
    overlap_rows_kernel<<<(m + 3) / 4, 256, 0, s>>>(raw, m, n, ld, rep4(indet), col_ok, nchunk, good);
}

void launch_row_nongap(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_res,
                       int32_t *row_nongap, uint32_t *used) {
    row_nongap_kernel<<<(m + 3) / 4, 256, 0, s>>>(raw, m, n, ld, keep_res, row_nongap, used);
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
                   uint8_t *keep_seq, int32_t *count) {
    const int words = (int)cluster_adj_words(m);
    const size_t lds = (size_t)4 * words * sizeof(uint32_t);
    if (lds > 60 * 1024) return -1;  // caller falls back to the host path
    dim3 grid((m + 255) / 256, words);
    uint32_t *nz = adj + (size_t)m * words;
    if (hipMemsetAsync(nz, 0, (size_t)m * ((words + 31) / 32) * sizeof(uint32_t), s) != hipSuccess) return -2;
    cluster_adjacency_kernel<<<grid, 256, 0, s>>>(ident, ldw, seq_at, seq_at + m, m, thr, adj, words, nz);
    cluster_mis_kernel<<<1, 1024, lds, s>>>(adj, nz, m, words, seq_at, keep_seq, count);
    return 0;
}

void launch_rows_equal(hipStream_t s, const uint8_t *raw, int n, int64_t ld, const int32_t *pairs, int npairs,
                       int32_t *equal) {
    if (npairs > 0) rows_equal_kernel<<<npairs, 64, 0, s>>>(raw, n, ld, pairs, npairs, equal);
}

}  // namespace msak
