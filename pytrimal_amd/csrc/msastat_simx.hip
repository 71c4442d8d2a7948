// msastat_simx.hip -- the binade-exact similarity kernel (statistics::Similarity::calculateVectors,
// reference include/trimal/statistics.pxd:55): both sequential float32 sums of every column, evaluated
// in parallel and still bit-identical to the reference's one-add-after-the-other order.
//
// Why a sequential float32 sum can be evaluated out of order (DESIGN.md section 5b has the proof):
//   s' = fl(s + x) with s in the binade [B, 2B), ulp u = B * 2^-23, and x >= 0 rounds x to the grid u:
//   s' = s + r_u(x), where r_u(x) depends on s only through the PARITY of s/u, and only when x is an exact
//   tie ((k + 1/2) u).  All terms of this statistic are >= 0, so s never leaves a binade downwards.  Hence,
//   as long as the running sum stays inside [B, 2B):
//     * an accumulator started at B (even) and one started at B + u (odd) that add the same terms in the same
//       order reproduce the increments the true sum would receive for either parity -- (inc_even, inc_odd);
//     * segments compose: after a segment the sum is s + inc_{parity(s)}, every quantity an exact multiple of u.
//   One lane owns one row j of the pair sequence (its terms k > j are contiguous in the reference's order), 64
//   rows make a round, and a scan over the lanes stitches the rows together: prefix sums (exact), parity picks at
//   the few tie rows, and the first row whose sum would reach 2B.  That row is evaluated in the reference's
//   order (blocks of 64 terms with the same test, then term by term); the rows behind it use a second pair of
//   accumulators kept on the next grid (2B, 2u).  Anything else ends the round early.  Every commit is checked
//   (sum < 2B, grids as assumed), so speculation can only cost time, never exactness.
//
// Work mapping: one wave = C columns; lane = row j of the round; the wave walks k = j0+1 .. m-1 once per round:
// W[k][j] (lower-triangular copy, one coalesced 256-B load per k shared by the C columns), the residue code of
// row k per column (scalar loads), one ds_read_b64 gather of {D, valid} per column, one packed multiply and four
// packed adds.  No chain: the kernel is bound by VALU / LDS issue, not by add latency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "msastat_kernels.h"

namespace msak {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) uint32_t *cu32p;  // constant address space: wave-uniform loads go through the scalar cache

constexpr uint32_t BX_SKIP = 224;  // code (8 x table row) of a residue that takes no part: row / column 28 of the table is zero
constexpr int BX_R0 = 8;           // rows evaluated in the reference's order before the first round
constexpr int BX_WAVES = 4;        // waves per workgroup (they only share the table in LDS)

// explicit address spaces: global loads (not flat) everywhere, scalar loads for wave-uniform addresses
typedef const __attribute__((address_space(1))) float *gf32p;
typedef const __attribute__((address_space(1))) uint8_t *gu8p;
typedef const __attribute__((address_space(3))) char *ldsp;  // the {distance, valid} table in LDS

__device__ __forceinline__ uint2 ld_codes8(gu8p p) {
    cu32p q = (cu32p)(uint64_t)p;
    return make_uint2(q[0], q[1]);
}

template <typename P>
__device__ __forceinline__ P uniform_ptr(P p) {  // a pointer every lane agrees on, moved to SGPRs
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (P)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ float rl(float v, int lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Inclusive prefix sum over the wave with DPP row shifts (the pattern LLVM's atomic optimizer uses on gfx9):
// Hillis-Steele inside each row of 16 lanes, then the row totals are carried across rows.  Lanes that a shift
// has nothing to bring to receive the identity (`old` = 0).  All adds are exact where the callers use the
// result (multiples of one ulp below 2^24 ulps), so the association order does not matter.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_prefix(float v) {
    v += dpp0<0x111, 0xF>(v);  // row_shr:1
    v += dpp0<0x112, 0xF>(v);  // row_shr:2
    v += dpp0<0x114, 0xF>(v);  // row_shr:4
    v += dpp0<0x118, 0xF>(v);  // row_shr:8
    v += dpp0<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp0<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return rl(wave_prefix(v), 63); }

// binade of s: B = 2^e, u = ulp; false for zero / tiny sums (those go through the ordered path)
__device__ __forceinline__ bool grid_of(float s, float &B, float &u) {
    const uint32_t b = __float_as_uint(s);
    if ((b >> 23) < 30u) {
        B = 0.0f;
        u = 0.0f;
        return false;
    }
    B = __uint_as_float(b & 0xFF800000u);
    u = __uint_as_float((b & 0xFF800000u) - (23u << 23));
    return true;
}

// 64 consecutive terms of one row (lane = term) added to s in order.
__device__ __forceinline__ float block_step(float s, float x) {
    if (__ballot(x != 0.0f) == 0ull) return s;
    float B, u;
    if (grid_of(s, B, u)) {
        const float Bo = B + u;
        const float re = (B + x) - B;
        const float ro = (Bo + x) - Bo;
        if (__ballot(re != ro) == 0ull) {  // no tie: the increments do not depend on the order
            const float sn = s + wave_sum(re);
            if (sn < 2.0f * B) return sn;
        }
    }
    for (int l = 0; l < 64; ++l) s = s + rl(x, l);
    return s;
}

// Row j of one column in the reference's order: k = j+1 .. m-1, lane = k within a block of 64.
// s = {numerator sum, denominator sum}; `which` selects the sums to advance (bit 0 / bit 1).
// (Every argument by value: a struct passed by reference would live in scratch memory and make the
// caller's loop counters look divergent to the compiler.)
__device__ __noinline__ f2 exact_row(gu8p cp, gf32p wup, int ldw, int m, ldsp tab, int j, int which, f2 s) {
    const int lane = threadIdx.x & 63;
    const uint32_t cj = cp[j];
    if (uni((int)cj) == (int)BX_SKIP) return s;
    gf32p wr = wup + (size_t)j * ldw + lane;
    gu8p cr = cp + lane;
    float s0 = s.x, s1 = s.y;
    int kb = ((j + 1) >> 6) << 6;
    float w = wr[kb];         // zero for k <= j and for the padding columns k >= m
    uint32_t ck = cr[kb];     // BX_SKIP for k >= m
    for (; kb < m; kb += 64) {
        const float wn = wr[kb + 64];  // (one block past the end: still inside the padded row / the next row)
        const uint32_t cn = cr[kb + 64];
        const f2 de = *reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tab + ((cj << 5) + ck));
        if (which & 1) s0 = block_step(s0, w * de.x);
        if (which & 2) s1 = block_step(s1, w * de.y);
        w = wn;
        ck = cn;
    }
    return f2{s0, s1};
}

// Stitch the rows [lo, hi) of a round: s before row lo, per-lane increments for an even / odd sum on the grid
// whose binade ends at `top`.  Returns the sum after each row (valid for the lanes before `cross`) and the
// first row whose sum would reach `top` (hi if none).
__device__ __forceinline__ float scan_rows(float s, float top, float ie, float io, int lo, int hi, int lane, int &cross) {
    const bool in = lane >= lo && lane < hi;
    const float a = in ? ie : 0.0f;
    const float P = wave_prefix(a);
    unsigned long long ties = __ballot(in && ie != io);
    float corr = 0.0f;
    while (ties) {
        const int t = __builtin_ctzll(ties);
        ties &= ties - 1;
        const float at = rl(a, t);
        const float st = s + ((rl(P, t) - at) + rl(corr, t));
        if (!(st < top)) break;
        const float chosen = (__float_as_uint(st) & 1u) ? rl(io, t) : at;
        const float delta = chosen - at;
        if (lane >= t) corr += delta;
    }
    const float sp = s + (P + corr);
    const unsigned long long x = __ballot(in && !(sp < top));
    cross = x ? __builtin_ctzll(x) : hi;
    return sp;
}

// One chain (numerator or denominator of one column) at the end of a round of `limit` rows starting at j0.
// Returns {the sum after every row (per lane), the new limit}: the limit shrinks when the rows behind some
// point cannot be committed.
struct Resolved {
    float sp;
    int limit;
};
__device__ __noinline__ Resolved resolve_chain(gu8p cp, gf32p wup, int ldw, int m, ldsp tab, int j0, int kind, float s,
                                               float ie, float io, float ie2, float io2, int limit, bool dual) {
    const int lane = threadIdx.x & 63;
    float B, u;
    if (!grid_of(s, B, u)) {
        // no binade yet (sum still zero): the accumulators are plain sums; the first row that contributes is
        // evaluated in order and ends the round
        const unsigned long long nz = __ballot(lane < limit && ie != 0.0f);
        if (!nz) return Resolved{s, limit};
        const int x = __builtin_ctzll(nz);
        const f2 r = exact_row(cp, wup, ldw, m, tab, j0 + x, kind ? 2 : 1, f2{s, s});
        const float sx = kind ? r.y : r.x;
        return Resolved{lane < x ? s : sx, x + 1};
    }
    int x;
    const float sp = scan_rows(s, 2.0f * B, ie, io, 0, limit, lane, x);
    if (x >= limit) return Resolved{sp, limit};
    // row x would leave the binade: evaluate it in order
    const float before = x > 0 ? rl(sp, x - 1) : s;
    const f2 r = exact_row(cp, wup, ldw, m, tab, j0 + x, kind ? 2 : 1, f2{before, before});
    const float sx = kind ? r.y : r.x;
    const float B2 = 2.0f * B;
    // without the second grid (the round did not expect this chain to cross), or after a row that spans two
    // binades, the round ends behind row x
    if (!dual || !(sx >= B2 && sx < 2.0f * B2)) return Resolved{lane < x ? sp : sx, x + 1};
    // the rows behind it were also accumulated on the next grid
    int y;
    const float sp2 = scan_rows(sx, 2.0f * B2, ie2, io2, x + 1, limit, lane, y);
    // (y < limit: a second crossing in the same round; the next round starts at that row)
    return Resolved{lane < x ? sp : (lane == x ? sx : sp2), y};
}

// cycle stamps of MSA_SIM_MODE=64 (diagnostics): [0] prologue, [1] round loops, [2] stitching, [3] waves, [4] rounds,
// [5] column slots that carried the second grid, summed over rounds, [6] wave lifetimes in 100 MHz ticks, [7] longest wave (cycles)
__device__ unsigned long long g_bx_stamps[16];
__device__ unsigned int g_bx_rec[16384 * 8];  // per wave (diagnostics): first column, cycles / 64 of the three phases, rounds, shortened rounds  // [8] most rounds of a wave, [9] shortened rounds, [10] rounds of the longest-running wave

// The k loop of one round.  Slot cc < ND also accumulates on the next grid (an2 / ad2): only the columns with a
// chain that may leave its binade in this round pay for that (they are sorted to the front, see the kernel).
template <int C, int ND>
__device__ __forceinline__ void round_loop(ldsp tabp, gf32p wrow, size_t ldw, const gu8p (&cp)[C], int kstart, int kend,
                                           int lane, const uint32_t (&cj8)[C], f2 (&an)[C], f2 (&an2)[C], f2 (&ad)[C],
                                           f2 (&ad2)[C]) {
    // uniform row pointer + lane: global loads with an SGPR base; two register sets (A, B) of 8 rows each
    // alternate between "being loaded" and "being consumed"
    float wA[8], wB[8];
    uint2 cA[C], cB[C];
    auto load8 = [&](float(&w)[8], uint2(&c)[C], int kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = (wrow + (size_t)(kb - kstart + i) * ldw)[lane];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) c[cc] = ld_codes8(cp[cc] + kb);
    };
    auto consume8 = [&](const float(&w)[8], const uint2(&c)[C]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f2 de[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const uint32_t word = i < 4 ? c[cc].x : c[cc].y;
                const uint32_t ck = (word >> (8 * (i & 3))) & 0xFFu;
                de[cc] = *reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tabp + ((ck << 5) + cj8[cc]));
            }
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const f2 x = de[cc] * w[i];
                const f2 xn = {x.x, x.x}, xd = {x.y, x.y};
                an[cc] += xn;
                ad[cc] += xd;
                if (cc < ND) {
                    an2[cc] += xn;
                    ad2[cc] += xd;
                }
            }
        }
    };
    load8(wA, cA, kstart);
#pragma unroll 1
    for (int kb = kstart; kb < kend; kb += 16) {  // (rows up to kend + 23 exist: zero padding, skipped codes)
        load8(wB, cB, kb + 8);
        consume8(wA, cA);
        load8(wA, cA, kb + 16);
        consume8(wB, cB);
    }
}

template <typename T, int C>
__device__ __forceinline__ T pick(const T (&v)[C], int idx) {  // v[idx] for a wave-uniform idx, without indexing registers
    T r = v[0];
#pragma unroll
    for (int i = 1; i < C; ++i) r = idx == i ? v[i] : r;
    return r;
}

template <int C, bool STAMP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void similarity_bx_kernel(const uint8_t *__restrict__ codeT_, int64_t ldk, int m,
                                                                       int n, const int32_t *__restrict__ cols,
                                                                       int ncols, const float *__restrict__ wlow_,
                                                                       const float *__restrict__ wup_, int ldw_, int r0_,
                                                                       const float *__restrict__ tab_g,
                                                                       float *__restrict__ num_out,
                                                                       float *__restrict__ den_out) {
    const gu8p codeT = (gu8p)(uint64_t)codeT_;
    const gf32p wlow = (gf32p)(uint64_t)wlow_, wup = (gf32p)(uint64_t)wup_;
    __shared__ f2 tab[32 * 32];                       // {distance, both valid}[row code][column code], rows 28.. zero
    __shared__ float spbuf[WAVES][2 * C][64];         // per chain: the sum after every row of the round
    for (int i = threadIdx.x; i < 32 * 32; i += 64 * WAVES) {
        f2 v = {0.0f, 0.0f};
        if (i < 29 * 32) v = reinterpret_cast<const f2 *>(tab_g)[i];
        tab[i] = v;
    }
    __syncthreads();
    const ldsp tabp = (ldsp)(const __attribute__((address_space(3))) void *)tab;
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    const int c0 = (blockIdx.x * WAVES + wave) * C;  // position in the list of active columns
    if (c0 >= ncols) return;
    int colidx[C];   // the wave's columns (the list is padded with column n: an all-skipped column)
    gu8p cp[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        colidx[cc] = uni(cols[c0 + cc]);
        cp[cc] = uniform_ptr(codeT + (size_t)colidx[cc] * ldk);
    }

    // lane i < 2C holds the running sum of chain i (numerators 0..C-1, denominators C..2C-1) and the increment
    // its last full round brought (the estimate behind the second-grid decision; < 0: unknown)
    float sall = 0.0f, pinc = -1.0f;
    unsigned long long t_pro = 0, t_loop = 0, t_res = 0, n_rounds = 0, n_dual = 0, n_short = 0, t0 = 0, rt0 = 0;
    if (STAMP) {
        t0 = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    {
        const int r0 = min(r0_, m - 1);
#pragma unroll 1
        for (int cc = 0; cc < C; ++cc) {
            f2 s2 = {0.0f, 0.0f};
            const gu8p p = pick(cp, cc);
            for (int j = 0; j < r0; ++j) s2 = exact_row(p, wup, ldw_, m, tabp, j, 3, s2);
            if (lane == cc) sall = s2.x;
            if (lane == C + cc) sall = s2.y;
        }
    }
    if (STAMP) {
        const unsigned long long t1 = __builtin_readcyclecounter();
        t_pro = t1 - t0;
        t0 = t1;
    }
    int j0 = min(r0_, m - 1);
    const int kend = (m + 7) & ~7;
    const size_t ldw = (size_t)ldw_;
    int guard = 0;  // every round commits at least one row; a round that does not would loop forever
    while (j0 < m - 1) {
        if (++guard > m + 64) {
            sall = __uint_as_float(0x7FC00000u);  // (never reached; NaN results fail every parity test)
            break;
        }
        const int nrows = min(64 - ((j0 - r0_) & 63), m - 1 - j0);
        // Which chains may leave their binade in this round?  Their columns go to the front slots, which also
        // accumulate on the next grid.  A wrong "no" only shortens the round (resolve_chain), never the result.
        int perm[C], nd = 0;
        {
            float Bl, ul;
            const bool grid = grid_of(sall, Bl, ul);
            const bool risky = !grid || pinc < 0.0f || !(sall + 1.3f * pinc * ((float)nrows * (1.0f / 64.0f)) < 2.0f * Bl);
            const unsigned long long rb = __ballot(lane < 2 * C && risky);
            const uint32_t colrisk = (uint32_t)(rb | (rb >> C)) & ((1u << C) - 1u);
#pragma unroll
            for (int cc = 0; cc < C; ++cc)
                if (colrisk >> cc & 1u) {
#pragma unroll
                    for (int q = 0; q < C; ++q)
                        if (q == nd) perm[q] = cc;
                    ++nd;
                }
            int pos = nd;
#pragma unroll
            for (int cc = 0; cc < C; ++cc)
                if (!(colrisk >> cc & 1u)) {
#pragma unroll
                    for (int q = 0; q < C; ++q)
                        if (q == pos) perm[q] = cc;
                    ++pos;
                }
        }
        gu8p cps[C];
        uint32_t cj8[C];
        f2 an[C], an2[C], ad[C], ad2[C];
        float Bn[C], un[C], Bd[C], ud[C];
#pragma unroll
        for (int q = 0; q < C; ++q) {
            cps[q] = pick(cp, perm[q]);
            cj8[q] = lane < nrows ? (uint32_t)cps[q][j0 + lane] : BX_SKIP;
            grid_of(rl(sall, perm[q]), Bn[q], un[q]);
            grid_of(rl(sall, C + perm[q]), Bd[q], ud[q]);
            an[q] = f2{Bn[q], Bn[q] + un[q]};
            an2[q] = f2{2.0f * Bn[q], 2.0f * Bn[q] + 2.0f * un[q]};
            ad[q] = f2{Bd[q], Bd[q] + ud[q]};
            ad2[q] = f2{2.0f * Bd[q], 2.0f * Bd[q] + 2.0f * ud[q]};
        }
        const int kstart = (j0 + 1) & ~7;
        const gf32p wrow = uniform_ptr(wlow + (size_t)kstart * ldw + j0);
        switch (nd) {
            case 0: round_loop<C, 0>(tabp, wrow, ldw, cps, kstart, kend, lane, cj8, an, an2, ad, ad2); break;
            case 1: round_loop<C, 1>(tabp, wrow, ldw, cps, kstart, kend, lane, cj8, an, an2, ad, ad2); break;
            case 2: round_loop<C, (C > 2 ? 2 : C)>(tabp, wrow, ldw, cps, kstart, kend, lane, cj8, an, an2, ad, ad2); break;
            case 3: round_loop<C, (C > 3 ? 3 : C)>(tabp, wrow, ldw, cps, kstart, kend, lane, cj8, an, an2, ad, ad2); break;
            default: round_loop<C, C>(tabp, wrow, ldw, cps, kstart, kend, lane, cj8, an, an2, ad, ad2); break;
        }
        if (STAMP) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_loop += t1 - t0;
            t0 = t1;
            ++n_rounds;
            n_dual += nd;
        }
        // per-row increments of every chain (slot order), then the chains one after the other
        float ie[2 * C], io[2 * C], ie2[2 * C], io2[2 * C];
#pragma unroll
        for (int q = 0; q < C; ++q) {
            ie[q] = an[q].x - Bn[q];
            io[q] = an[q].y - (Bn[q] + un[q]);
            ie2[q] = an2[q].x - 2.0f * Bn[q];
            io2[q] = an2[q].y - (2.0f * Bn[q] + 2.0f * un[q]);
            ie[C + q] = ad[q].x - Bd[q];
            io[C + q] = ad[q].y - (Bd[q] + ud[q]);
            ie2[C + q] = ad2[q].x - 2.0f * Bd[q];
            io2[C + q] = ad2[q].y - (2.0f * Bd[q] + 2.0f * ud[q]);
        }
        int limit = nrows;
#pragma unroll 1
        for (int ch = 0; ch < 2 * C; ++ch) {
            float e = 0.0f, o = 0.0f, e2 = 0.0f, o2 = 0.0f;
#pragma unroll
            for (int q = 0; q < 2 * C; ++q)
                if (ch == q) {
                    e = ie[q];
                    o = io[q];
                    e2 = ie2[q];
                    o2 = io2[q];
                }
            const int slot = ch < C ? ch : ch - C;
            const int col = pick(perm, slot);
            const int chain = ch < C ? col : C + col;  // the chain's lane in `sall`
            const Resolved r = resolve_chain(pick(cps, slot), wup, ldw_, m, tabp, j0, ch < C ? 0 : 1, rl(sall, chain), e, o,
                                             e2, o2, limit, slot < nd);
            spbuf[wave][chain][lane] = r.sp;
            limit = uni(r.limit);
        }
        limit = max(uni(limit), 1);
        if (lane < 2 * C) {
            const float snew = spbuf[wave][lane][limit - 1];  // (limit >= 1: every chain commits at least one row)
            float Bl, ul;
            // (a short round is a poor sample of the increment per row: keep the previous estimate)
            if (!grid_of(sall, Bl, ul)) pinc = -1.0f;
            else if (limit >= 16) pinc = (snew - sall) * (64.0f / (float)limit);
            sall = snew;
        }
        j0 += limit;
        if (STAMP) {
            n_short += limit < nrows;
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_res += t1 - t0;
            t0 = t1;
        }
    }
    if (STAMP && lane == 0) {
        atomicAdd(&g_bx_stamps[0], t_pro);
        atomicAdd(&g_bx_stamps[1], t_loop);
        atomicAdd(&g_bx_stamps[2], t_res);
        atomicAdd(&g_bx_stamps[3], 1ull);
        atomicAdd(&g_bx_stamps[4], n_rounds);
        atomicAdd(&g_bx_stamps[5], n_dual);
        atomicAdd(&g_bx_stamps[6], __builtin_amdgcn_s_memrealtime() - rt0);  // 100 MHz ticks
        atomicMax(&g_bx_stamps[7], t_pro + t_loop + t_res);
        atomicMax(&g_bx_stamps[8], n_rounds);
        const unsigned wid = (unsigned)(c0 / C);
        if (wid < 16384u) {
            unsigned int *r = g_bx_rec + 8 * wid;
            r[0] = (unsigned)colidx[0];
            r[1] = (unsigned)(t_pro >> 6);
            r[2] = (unsigned)(t_loop >> 6);
            r[3] = (unsigned)(t_res >> 6);
            r[4] = (unsigned)n_rounds;
            r[5] = (unsigned)n_short;
            r[6] = (unsigned)n_dual;
        }
        atomicAdd(&g_bx_stamps[9], n_short);
    }
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        if (colidx[cc] < n) {
            const float sn = rl(sall, cc), sd = rl(sall, C + cc);
            if (lane == 0) {
                num_out[colidx[cc]] = sn;
                den_out[colidx[cc]] = sd;
            }
        }
    }
}

// raw bytes -> column-major codes (64 x 64 tiles through LDS); first bad residue through atomicMin as in the
// other encode kernels.  Columns cut by the ">= 80 % gaps" rule and all padding hold BX_SKIP.
__global__ __launch_bounds__(256) void sim_encode_cm_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                            const uint8_t *__restrict__ lut_g,
                                                            const int32_t *__restrict__ gaps_w, uint8_t *__restrict__ codeT,
                                                            int64_t ldk, int ncols_pad,
                                                            unsigned long long *__restrict__ err_key) {
    __shared__ uint8_t lut[256];
    __shared__ uint8_t tile[64][68];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    bool skipcol = true;
    if (c < n) skipcol = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
    for (int r = ty; r < 64; r += 4) {
        const int row = blockIdx.y * 64 + r;
        uint32_t code = BX_SKIP;
        if (row < m && c < n && !skipcol) {
            const uint32_t byte = raw[(size_t)row * ld + c];
            code = lut[byte];  // 8 x table row, 224 = skipped, 0xFE / 0xFF = bad symbol
            if (code >= 0xFEu) {
                const unsigned long long key = ((unsigned long long)c << 40) | ((unsigned long long)row << 16) |
                                               ((unsigned long long)(code & 1u) << 8) | byte;
                atomicMin(err_key, key);
                code = BX_SKIP;
            }
        }
        tile[r][tx] = (uint8_t)code;
    }
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.y * 64 + tx;
    for (int q = ty; q < 64; q += 4) {
        const int col = blockIdx.x * 64 + q;
        if (col < ncols_pad && k < ldk) codeT[(size_t)col * ldk + k] = tile[tx][q];
    }
}

}  // namespace

int64_t bx_ldk(int m) { return ((int64_t)m + 63) / 64 * 64 + 64; }
int bx_cols_pad(int n) { return (n + 1 + 63) / 64 * 64; }  // at least one all-skipped column behind the last one
size_t bx_wlow_rows(int m) { return (size_t)((m + 7) / 8 * 8 + 32); }

void launch_sim_encode_cm(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut,
                          const int32_t *gaps_w, uint8_t *codeT, unsigned long long *err_key) {
    const int64_t ldk = bx_ldk(m);
    const int ncp = bx_cols_pad(n);
    dim3 grid((unsigned)(ncp / 64), (unsigned)(ldk / 64));
    sim_encode_cm_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, lut, gaps_w, codeT, ldk, ncp, err_key);
}

int bx_cols_per_wave() {
    const int c = tuning().bx_cols;
    return (c == 1 || c == 2 || c == 4) ? c : 2;
}

// cols: the columns to evaluate (device, ncols entries rounded up to a multiple of bx_cols_per_wave() with the
// index n, the all-skipped column); the sums of every other column must have been zeroed by the caller.
int launch_similarity_bx(hipStream_t s, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols,
                         const float *wlow, const float *wup, int ldw, const void *tab, float *num_out, float *den_out) {
    const int64_t ldk = bx_ldk(m);
    const int C = bx_cols_per_wave();
    const int r0 = tuning().bx_r0 >= 0 ? tuning().bx_r0 : BX_R0;
    const int waves = tuning().bx_waves == 1 ? 1 : BX_WAVES;
    const int per_wg = waves * C;
    const unsigned grid = (unsigned)((ncols + per_wg - 1) / per_wg);
    if (grid == 0) return 0;
    const float *t = static_cast<const float *>(tab);
    const bool stamp = (tuning().sim_mode & 64) != 0;
#define BX_LAUNCH2(CC, ST, WV)                                                                                        \
    similarity_bx_kernel<CC, ST, WV><<<grid, 64 * WV, 0, s>>>(codeT, ldk, m, n, cols, ncols, wlow, wup, ldw, r0, t, num_out, den_out)
#define BX_LAUNCH(CC)                                                   \
    do {                                                                \
        if (stamp && waves == 1) BX_LAUNCH2(CC, true, 1);               \
        else if (stamp) BX_LAUNCH2(CC, true, BX_WAVES);                 \
        else if (waves == 1) BX_LAUNCH2(CC, false, 1);                  \
        else BX_LAUNCH2(CC, false, BX_WAVES);                           \
    } while (0)
    if (C == 1) BX_LAUNCH(1);
    else if (C == 2) BX_LAUNCH(2);
    else BX_LAUNCH(4);
#undef BX_LAUNCH
#undef BX_LAUNCH2
    return 0;
}

extern "C" int msa_debug_bx_stamps(unsigned long long *out16, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bx_stamps), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bx_stamps), z, sizeof(z));
    }
    return rc;
}

extern "C" int msa_debug_bx_records(unsigned int *out, int nwaves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bx_rec), sizeof(unsigned int) * 8 * (size_t)(nwaves < 16384 ? nwaves : 16384));
}

}  // namespace msak
