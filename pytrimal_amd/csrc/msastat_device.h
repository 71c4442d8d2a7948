// msastat_device.h -- device-side bodies shared by the two kernel files (msastat_kernels.hip, msastat_simx.hip): the
// single-alignment kernels, their batched wrappers and the compact pipeline of small alignments (msastat_simx.hip) all run
// the same code.  Internal; see msastat_kernels.hip for the data layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msastat_kernels.h"

namespace msak {


__device__ __forceinline__ uint32_t gather_bit4(uint32_t x, int b) {
    // bit b of each of the 4 bytes of x -> 4 adjacent bits (byte 0 -> bit 0)
    return ((((x >> b) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {
    // 0x80 in every byte of v that is zero (exact, bytes < 0x80 or not)
    uint32_t t = (v & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | v | 0x7f7f7f7fu);
}

// prep_planes: raw bytes -> bit-sliced planes.  One thread = one row x 64 columns (a full 64-B line of that row); lanes of a
// wave own consecutive rows so the plane stores coalesce.
// Four dwords (16 columns from col0 on) of a row -> 16 bits of every plane (bit = column - col0); returns the non-ASCII bits seen.
__device__ __forceinline__ uint32_t planes_of_quarter(const uint32_t *w, int col0, int n, uint32_t indet4, uint32_t (&out)[8]) {
    uint32_t bad = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = col0 + e * 4;  // first column of this dword
        uint32_t x = w[e];
        // mask columns >= n (undefined bytes) to '-' so they are invalid everywhere
        uint32_t keep = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) keep |= (col + k < n) ? (0xFFu << (8 * k)) : 0u;
        x = (x & keep) | (0x2d2d2d2du & ~keep);
        bad |= x & 0x80808080u;
        uint32_t inval = zero_bytes(x ^ 0x2d2d2d2du) | zero_bytes(x ^ indet4);  // 0x80 flags
        uint32_t vbits = gather_bit4(~inval, 7);
#pragma unroll
        for (int p = 0; p < 7; ++p) out[p] |= gather_bit4(x, p) << (e * 4);
        out[7] |= vbits << (e * 4);
    }
    return bad;
}
// The sixteen dwords of a row's 64-column group -> its two chunk words of every plane.
__device__ __forceinline__ uint32_t planes_of_row(const uint32_t (&w)[16], int col0, int n, uint32_t indet4, uint32_t (&out)[2][8]) {
    uint32_t bad = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t o[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) o[p] = 0;
        bad |= planes_of_quarter(w + q * 4, col0 + q * 16, n, indet4, o);
#pragma unroll
        for (int p = 0; p < 8; ++p) out[q >> 1][p] |= o[p] << ((q & 1) * 16);
    }
    return bad;
}
__device__ __forceinline__ void planes_store(uint32_t *__restrict__ planes, int nchunk, int m_pad, int cpair, int row, const uint32_t (&out)[2][8]) {
    const size_t pstride = (size_t)nchunk * m_pad;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int chunk = cpair * 2 + h;
        if (chunk < nchunk) {
            const size_t at = (size_t)chunk * m_pad + row;
#pragma unroll
            for (int p = 0; p < 8; ++p) planes[p * pstride + at] = out[h][p];
        }
    }
}
// Returns the thread's non-ASCII bits (0: none).
__device__ __forceinline__ uint32_t prep_planes_core(const uint8_t *__restrict__ raw, int m, int n,
                                                     int64_t ld, uint32_t indet4, uint32_t *__restrict__ planes,
                                                     int nchunk, int m_pad, int bx, int by) {
    const int row = by * 256 + threadIdx.x;  // < m_pad
    const int cpair = bx;                    // 64-column group (the x dimension: no 65 535 limit on the columns)
    if (row >= m_pad) return 0u;
    uint32_t bad = 0;
    uint32_t out[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 8; ++p) out[h][p] = 0;
    if (row < m) {  // (the rows behind m stay zero in every plane: the pair pass computes them and writes nothing)
        const uint4 *src = reinterpret_cast<const uint4 *>(raw + (size_t)row * ld + (size_t)cpair * 64);
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v4 = src[q];
            w[q * 4 + 0] = v4.x, w[q * 4 + 1] = v4.y, w[q * 4 + 2] = v4.z, w[q * 4 + 3] = v4.w;
        }
        bad = planes_of_row(w, cpair * 64, n, indet4, out);
    }
    planes_store(planes, nchunk, m_pad, cpair, row, out);
    return bad;
}
__device__ __forceinline__ void prep_planes_body(const uint8_t *__restrict__ raw, int m, int n,
                                                 int64_t ld, uint32_t indet4, uint32_t *__restrict__ planes,
                                                 int nchunk, int m_pad, int *__restrict__ err_flag, int bx, int by) {
    if (prep_planes_core(raw, m, n, ld, indet4, planes, nchunk, m_pad, bx, by)) atomicOr(err_flag, 1);
}

// ---- pair pass: tiles, epilogue, the software-pipelined loop (see msastat_kernels.hip: pair_counts) ----
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }

// tile of a block: (i-block, j-block).  One-dimensional grid over the tiles that hold pairs (j > i) only, j-block by
// j-block -- a two-dimensional grid launches as many tiles that return at once, and the waves that do the work end up
// unevenly spread over the SIMDs (every active wave is resident from the start: the fullest SIMD sets the time).
// j-block y holds min(n_iblocks, (y + 1) R) tiles, R = 64 TJ / TI.
template <int TI, int TJ>
__device__ __forceinline__ void pair_tile(int n_iblocks, int t, int &ib, int &jb) {
    {
        constexpr int R = 64 * TJ / TI;
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

// (Measured twice and not shipped -- round 2, and round 5 with the waves of a workgroup sharing a tile: the same tiles dealt to the
// XCDs j-block by j-block, XCD x owning the j-blocks x, 15 - x, 16 + x, ... so that its L2 holds the j planes its waves stream
// instead of a fifth of them.  The re-reads from beyond the L2 go, the time does not: 0.31 -> 0.37 ms at 2000 x 10000, 0.041 -> 0.053
// at 1000 x 4000 alone -- hundreds of tiles of one XCD ask its L2 for the same lines at the same moment -- and 24.0 against 24.2 ms
// for the C5 batch: gpurun_out/ab6, profiles/r05_pairs_xcd.txt.)
// epilogue of a tile: row-wise (coalesced along j) and mirrored (TI contiguous values per lane).  miss[][] counted the
// misses of all 32 * nchunk columns (the columns behind n are gaps in every row).
template <int TI, int TJ>
__device__ __forceinline__ void pair_epilogue(const uint32_t (&miss)[TJ][TI], const uint32_t (&dst)[TJ][TI], int i0, int j0, int lane,
                                              int nchunk, int m, int ldw, uint32_t *__restrict__ hit_out,
                                              uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                              float *__restrict__ wmat, float *__restrict__ wlow, int *__restrict__ undef_flag,
                                              uint32_t *__restrict__ wsum = nullptr) {
    // wsum (the compact pipeline): row i's sum of W[i][j] over j > i in 16.16 fixed point, accumulated by integer atomics --
    // any order gives the same sum; the similarity kernel's predictor takes its mean weights from it.  Every pair i < j
    // is seen by exactly one tile as (i, j) (pairs inside a diagonal block are seen a second time as (j, i): not counted).
    uint32_t rowacc[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) rowacc[t] = 0;
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
            // no column holds a residue of either row (every writer stores the same value: no atomic -- the word may live in
            // pinned host memory)
            if (!diag && d == 0u && undef_flag) *reinterpret_cast<volatile int *>(undef_flag) = 1;
            if (ident || wmat) {
                const float r = d ? (float)h / (float)d : 0.0f;
                if (ident) {
                    const float v = diag ? 0.0f : r;
                    ident[(size_t)i * ldw + j] = v;
                    ident[(size_t)j * ldw + i] = v;
                }
                if (wmat && i != j) {  // strictly upper triangular: the similarity pass reads W[j][k], k > j
                    const float v = 1.0f - r;
                    if (i < j) {
                        wmat[(size_t)i * ldw + j] = v;
                        rowacc[t] += (uint32_t)(v * 65536.0f + 0.5f);
                    } else wmat[(size_t)j * ldw + i] = v;
                    // the mirror image (strictly lower triangular) for the kernel whose lanes are the rows j
                    if (wlow) wlow[(size_t)(i < j ? j : i) * ldw + (i < j ? i : j)] = v;
                }
            }
        }
    }
    if (wsum) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            uint32_t v = rowacc[t];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
            if (lane == 0 && i0 + t < m && v) atomicAdd(&wsum[i0 + t], v);
        }
    }
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
//
// The waves of a workgroup share a tile (round 4): wave w of K walks the chunks [nchunk w / K, nchunk (w + 1) / K), the partial
// counts meet in LDS and wave 0 writes the tile.  A tile's wave used to walk ALL chunks alone, and below ~4000 sequences there are
// fewer tiles than the chip has room for waves (2000 sequences: 4000 tiles, all resident from the start, the fullest SIMD sets
// the time; 1000 sequences: 1000 tiles for 1024 SIMDs; 500: 252): K = 2 at 2000 rows, 8 at 1000 and below (launch_pair_counts).
constexpr int PAIR_KMAX = 8;
template <int TJ>
__device__ __forceinline__ void pair_counts_pipe_body(const uint32_t *__restrict__ planes, int nchunk, int m_pad,
                                                      int m, int ldw, uint32_t *__restrict__ hit_out,
                                                      uint32_t *__restrict__ dst_out, float *__restrict__ ident,
                                                      float *__restrict__ wmat, float *__restrict__ wlow,
                                                      int *__restrict__ undef_flag, int ib, int jb,
                                                      uint32_t *__restrict__ wsum = nullptr) {
    constexpr int TI = 8;
    typedef const __attribute__((address_space(4))) u32x8 *c8;
    extern __shared__ uint32_t pair_part[];  // [K - 1][2][TI][64]: the partial counts of the waves 1 .. K-1
    const int lane = threadIdx.x & 63;
    const int K = (int)(blockDim.x >> 6), kw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cbeg = (int)((long)nchunk * kw / K), cend = (int)((long)nchunk * (kw + 1) / K);
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
        request_j(bn, c + 1 < cend ? c + 1 : c);
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
        request_a(ga, c + 1 < cend ? c + 1 : c);
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
    if (cbeg < cend) {
        request_a(ga, cbeg);
        request_j(b0, cbeg);
        int c = cbeg;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (; c + 1 < cend; c += 2) {  // (no branch inside the body: one basic block, the order above is the order issued)
            step(c, b0, b1);
            step(c + 1, b1, b0);
        }
        if (c < cend) step(c, b0, b1);
        arrived(ga, miss);  // (the last request, a repeat of the last chunk, is not used)
    }
    if (K > 1) {
        static_assert(TJ == 1, "the waves of a workgroup share a tile in the one-row-per-lane regime only");
        if (kw > 0) {
            uint32_t *mine = pair_part + (size_t)(kw - 1) * 2 * TI * 64 + lane;
#pragma unroll
            for (int t = 0; t < TI; ++t) mine[t * 64] = miss[0][t], mine[(TI + t) * 64] = dst[0][t];
        }
        __syncthreads();
        if (kw > 0) return;
        for (int w = 0; w < K - 1; ++w) {
            const uint32_t *theirs = pair_part + (size_t)w * 2 * TI * 64 + lane;
#pragma unroll
            for (int t = 0; t < TI; ++t) miss[0][t] += theirs[t * 64], dst[0][t] += theirs[(TI + t) * 64];
        }
    }
    pair_epilogue<TI, TJ>(miss, dst, i0, j0, lane, nchunk, m, ldw, hit_out, dst_out, ident, wmat, wlow, undef_flag, wsum);
}

// The same loop on SIXTEEN rows i per tile (round 6), two halves of eight one after the other on the SAME j planes: a chunk is
// four phases (A and B of rows 0-7, A and B of rows 8-15), each on a group of 32 SGPRs requested one phase earlier -- the scalar
// registers in use are what the eight-row loop holds, while the j planes (seven eighths of what a tile reads) are loaded half as
// often per pair and there are half as many tiles, epilogues and partial-count exchanges.  (Round 2's sixteen-row tile held all
// sixteen rows' plane words at once and serialised on its scalar loads: DESIGN A.2.)  The K waves of the workgroup add their
// partial counts into ONE LDS tile with integer atomics (8 KB whatever K is; the eight-row loop keeps K - 1 tiles: 28 KB at K = 8).
__device__ __forceinline__ void pair_counts_pipe16_body(const uint32_t *__restrict__ planes, int nchunk, int m_pad, int m, int ldw,
                                                        float *__restrict__ ident, float *__restrict__ wmat, float *__restrict__ wlow,
                                                        int *__restrict__ undef_flag, int ib, int jb, uint32_t *__restrict__ wsum) {
    constexpr int TI = 8, H = 2;
    typedef const __attribute__((address_space(4))) u32x8 *c8;
    __shared__ uint32_t part[2][H][TI][64];
    const int lane = threadIdx.x & 63;
    const int K = (int)(blockDim.x >> 6), kw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cbeg = (int)((long)nchunk * kw / K), cend = (int)((long)nchunk * (kw + 1) / K);
    const int i0 = ib * (TI * H);  // uniform
    const int j0 = jb * 64;
    if (j0 >= m_pad) return;
    if (j0 + 63 <= i0) return;
    if (K > 1) {
        for (int i = threadIdx.x; i < 2 * H * TI * 64; i += (int)blockDim.x) (&part[0][0][0][0])[i] = 0u;
        __syncthreads();
    }
    uint32_t miss[H][1][TI], dst[H][1][TI], d[1][TI];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int t = 0; t < TI; ++t) miss[h][0][t] = dst[h][0][t] = 0;
    // Every address is the planes' base (a scalar pair) plus a 32-bit byte offset (the launcher keeps this kernel to plane arrays
    // below 4 GB): `s_load_dwordx8 dst, base, offset` for the rows i, `global_load_dword dst, voffset, base` for the rows j, and
    // the eight plane offsets are eight scalars shared by both -- sixteen 64-bit plane addresses kept alive across the four
    // phases spilled a hundred scalar registers into the loop.
    typedef const __attribute__((address_space(4))) char *cbytep;
    typedef const __attribute__((address_space(1))) char *gbytep;
    const uint32_t ps4 = (uint32_t)nchunk * (uint32_t)m_pad * 4u, mp4 = (uint32_t)m_pad * 4u;
    uint32_t poff[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) poff[p] = (uint32_t)p * ps4;
    const cbytep pib = (cbytep)(uint64_t)planes;
    const gbytep pjb = (gbytep)(uint64_t)planes;
    const uint32_t i04 = (uint32_t)i0 * 4u, jl4 = (uint32_t)(j0 + lane) * 4u;
    struct Group {
        u32x8 p[4];
    };
    auto request_a = [&](Group &g, int c, int h) {  // validity plane, planes 0..2
        const uint32_t q = (uint32_t)c * mp4 + i04 + 32u * (uint32_t)h;
        g.p[0] = *(c8)(pib + (q + poff[7]));
        g.p[1] = *(c8)(pib + (q + poff[0]));
        g.p[2] = *(c8)(pib + (q + poff[1]));
        g.p[3] = *(c8)(pib + (q + poff[2]));
    };
    auto request_b = [&](Group &g, int c, int h) {  // planes 3..6
        const uint32_t q = (uint32_t)c * mp4 + i04 + 32u * (uint32_t)h;
#pragma unroll
        for (int p = 0; p < 4; ++p) g.p[p] = *(c8)(pib + (q + poff[3 + p]));
    };
    auto arrived = [&](Group &g, uint32_t (&pin)[1][TI]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+s"(g.p[0]), "+s"(g.p[1]), "+s"(g.p[2]), "+s"(g.p[3]), "+v"(pin[0][0]), "+v"(pin[0][1]), "+v"(pin[0][2]),
                       "+v"(pin[0][3]), "+v"(pin[0][4]), "+v"(pin[0][5]), "+v"(pin[0][6]), "+v"(pin[0][7]));
    };
    auto request_j = [&](uint32_t (&b)[8], int c) {
        const uint32_t q = (uint32_t)c * mp4 + jl4;
#pragma unroll
        for (int p = 0; p < 8; ++p) b[p] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(pjb + (q + poff[p]));
    };
    Group ga, gb;
    uint32_t b0[8], b1[8];
    auto phase_a = [&](int h, const uint32_t (&b)[8]) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const uint32_t vi = ga.p[0][t], nvi = ~vi;  // columns in which row i holds no residue never count as hits
            uint32_t x = __builtin_amdgcn_bitop3_b32(nvi, ga.p[1][t], b[0], 0xF6);  // x | (y ^ z)
            x = __builtin_amdgcn_bitop3_b32(x, ga.p[2][t], b[1], 0xF6);
            d[0][t] = __builtin_amdgcn_bitop3_b32(x, ga.p[3][t], b[2], 0xF6);
            dst[h][0][t] += __builtin_popcount(vi | b[7]);
        }
    };
    auto phase_b = [&](int h, const uint32_t (&b)[8]) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            uint32_t x = __builtin_amdgcn_bitop3_b32(d[0][t], gb.p[0][t], b[3], 0xF6);
            x = __builtin_amdgcn_bitop3_b32(x, gb.p[1][t], b[4], 0xF6);
            x = __builtin_amdgcn_bitop3_b32(x, gb.p[2][t], b[5], 0xF6);
            x = __builtin_amdgcn_bitop3_b32(x, gb.p[3][t], b[6], 0xF6);
            miss[h][0][t] += __builtin_popcount(x);
        }
    };
    auto step = [&](int c, uint32_t (&b)[8], uint32_t (&bn)[8]) {
        const int cn = c + 1 < cend ? c + 1 : c;
        arrived(ga, miss[1]);  // group A of (chunk c, rows 0-7)
        request_b(gb, c, 0);
        request_j(bn, cn);
        __builtin_amdgcn_sched_barrier(0);
        phase_a(0, b);
        __builtin_amdgcn_sched_barrier(0);
        arrived(gb, d);  // group B of (c, rows 0-7)
        request_a(ga, c, 1);
        __builtin_amdgcn_sched_barrier(0);
        phase_b(0, b);
        __builtin_amdgcn_sched_barrier(0);
        arrived(ga, miss[0]);  // group A of (c, rows 8-15)
        request_b(gb, c, 1);
        __builtin_amdgcn_sched_barrier(0);
        phase_a(1, b);
        __builtin_amdgcn_sched_barrier(0);
        arrived(gb, d);  // group B of (c, rows 8-15)
        request_a(ga, cn, 0);
        __builtin_amdgcn_sched_barrier(0);
        phase_b(1, b);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (cbeg < cend) {
        request_a(ga, cbeg, 0);
        request_j(b0, cbeg);
        int c = cbeg;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (; c + 1 < cend; c += 2) {
            step(c, b0, b1);
            step(c + 1, b1, b0);
        }
        if (c < cend) step(c, b0, b1);
        arrived(ga, miss[1]);  // (the last request, a repeat of the last chunk, is not used)
    }
    if (K > 1) {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                atomicAdd(&part[0][h][t][lane], miss[h][0][t]);
                atomicAdd(&part[1][h][t][lane], dst[h][0][t]);
            }
        __syncthreads();
        // the epilogue's two halves on the workgroup's first two waves
        if (kw >= H) return;
#pragma unroll
        for (int h = 0; h < H; ++h)
            if (h == kw) {
#pragma unroll
                for (int t = 0; t < TI; ++t) miss[h][0][t] = part[0][h][t][lane], dst[h][0][t] = part[1][h][t][lane];
            }
        if (kw == 0) pair_epilogue<TI, 1>(miss[0], dst[0], i0, j0, lane, nchunk, m, ldw, nullptr, nullptr, ident, wmat, wlow, undef_flag, wsum);
        else pair_epilogue<TI, 1>(miss[1], dst[1], i0 + TI, j0, lane, nchunk, m, ldw, nullptr, nullptr, ident, wmat, wlow, undef_flag, wsum);
        return;
    }
#pragma unroll
    for (int h = 0; h < H; ++h)
        pair_epilogue<TI, 1>(miss[h], dst[h], i0 + TI * h, j0, lane, nchunk, m, ldw, nullptr, nullptr, ident, wmat, wlow, undef_flag, wsum);
}

// MDK from the two sums (Similarity::calculateVectors tail): 0 for >= 80 % gaps or an empty denominator, else
// min(1, (float)exp(-(double)Q)).  Q = num / den is bit-exact; the exponential is the device library's, which may differ
// from the host's in the last place of the DOUBLE -- and then in the float only when the double lies within a few of
// its own ulps of a point where the conversion to float changes its result.  Such a value (and one in the float
// denormal range, where flush modes could differ) is not trusted: it goes out as a NaN and the host evaluates
// (float)exp(-(double)Q) itself (fetch_similarity_finish).  Both libraries are accurate to an ulp, so every value that
// passes the test rounds to the same float on both sides: MDK is bit-identical to the host computation by construction.
__device__ __forceinline__ float mdk_value(float num, float d, bool skip, int all_on_host, float &q) {
    float v = 0.0f;
    q = 0.0f;
    if (!skip && d != 0.0f) {
        q = num / d;
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
    return v;
}
__device__ __forceinline__ void sim_finish_body(const float *__restrict__ num, const float *__restrict__ den,
                                                const int32_t *__restrict__ gaps_w, int m, int n,
                                                float *__restrict__ q_out, float *__restrict__ mdk_out, int all_on_host, int bx) {
    const int c = bx * 256 + threadIdx.x;
    if (c >= n) return;
    const bool skip = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
    float q;
    const float v = mdk_value(num[c], den[c], skip, all_on_host, q);
    if (q_out) q_out[c] = q;
    mdk_out[c] = v;
}

// residues (non-gap symbols) of a row over the kept columns: Cleaner::removeAllGapsSeqsAndCols, the row totals
__device__ __forceinline__ void row_nongap_body(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                const uint8_t *__restrict__ keep_res,
                                                int32_t *__restrict__ row_nongap, int bx) {
    const int row = bx * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
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
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
        if (lane == 0) row_nongap[row] = cnt;
    }
}

}  // namespace msak
