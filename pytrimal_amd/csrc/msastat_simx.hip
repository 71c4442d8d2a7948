// msastat_simx.hip -- the binade-exact similarity kernels (statistics::Similarity::calculateVectors,
// reference include/trimal/statistics.pxd:55): both sequential float32 sums of every column, evaluated
// in parallel and still bit-identical to the reference's one-add-after-the-other order.
//
// Why a sequential float32 sum can be evaluated out of order (DESIGN.md section 5.1 has the proof):
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
//   order (chunks of 256 terms with the same test, then blocks of 64, then term by term).  Every commit is checked
//   (sum < 2B, grids as assumed), so speculation can only cost time, never exactness.
//
// What is in this file, in order:
//   * helpers (scans, exact_row: one row in the reference's order);
//   * `similarity_lg`, THE similarity kernel: every lane on the grid of its own predicted sum, one loop version, the
//     lane's table column in a per-wave LDS table read by ds_read_addtid_b32, W rows by hand-issued global loads.  Work
//     mapping: one wave = one column (the columns the ">= 80 % gaps" rule zeroes never get a wave), lane = one of 64
//     consecutive rows j of the round, one step per VALID partner row k behind the round's first row (compacted
//     list): 3 VALU + 1 LDS + 1 VMEM instruction per 64 terms.  Bound by the vector L1's bandwidth (the W stream:
//     texture addresser 92 % busy), not by VALU issue or add latency.  Two instantiations: 32-bit byte offsets of the
//     W rows in the lists (m <= 32768), or row indices multiplied out on the scalar unit (any m);
//   * `similarity_seq` (MSA_SIM_KERNEL=seq): the statistic as the reference writes it -- one lane per column, two
//     nested loops, one add after the other.  Slow by construction; the cross-check of the kernel above at sizes the
//     CPU oracle cannot reach;
//   * the identity row statistics (sequential float32 sums through the same chunk test);
//   * the layout kernels (column-major codes, compacted lists) and the launchers.
// History: the round-1 chain kernels, the one-grid-per-round `bx` kernel, the table-in-registers
// and two-columns-per-wave (`q2`) variants were measured against this kernel in round 2 and removed in round 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "msastat_kernels.h"
#include "msastat_device.h"
#include "msastat_lgloop.inc"

namespace msak {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) uint32_t *cu32p;  // constant address space: wave-uniform loads go through the scalar cache

constexpr uint32_t BX_SKIP = 224;  // code (8 x table row) of a residue that takes no part: row / column 28 of the table is zero

// explicit address spaces: global loads (not flat) everywhere, scalar loads for wave-uniform addresses
typedef const __attribute__((address_space(1))) float *gf32p;
typedef const __attribute__((address_space(1))) uint8_t *gu8p;
typedef const __attribute__((address_space(3))) char *ldsp;  // the {distance, valid} table in LDS


template <typename P>
__device__ __forceinline__ P uniform_ptr(P p) {  // a pointer every lane agrees on, moved to SGPRs
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (P)(((uint64_t)hi << 32) | lo);
}

// (batches) the alignment a block belongs to: bisection over the prefix sums of blocks per alignment, wave-uniform
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
__device__ __forceinline__ float rl(float v, int lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a wave-uniform 64-bit mask as a per-lane predicate / the set bits below the lane: the mask stays in scalar registers (the
// shifts by the lane index the plain C spelling implies keep 64-bit per-lane masks alive in two vector registers each, across the
// whole round loop)
__device__ __forceinline__ bool lane_in(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ int bits_below_lane(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ float unif(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}

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

// One column as the kernels see it.
struct ColView {
    const __attribute__((address_space(1))) uint32_t *off;  // compacted list of its valid rows (bx_compact_kernel): byte offset of the row
                                                             // in W (row * ldw * 4), or the row index (`big` lists); padding: the zero row m
    int nvalid;                                              // entries of that list
    gu8p colcode;  // the column's codes by row (codeT): 8 x table row, BX_SKIP for a row that takes no part and behind row m
    int ldw;
    int m;
};

// 256 consecutive terms (x[i]: term 64 i + lane) added to s in order: one test for all of them, else block by block
__device__ __forceinline__ float chunk_step(float s, const float (&x)[4]) {
    float B, u;
    if (grid_of(s, B, u)) {
        const float Bo = B + u;
        float tot = 0.0f;
        bool tie = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float re = (B + x[i]) - B, ro = (Bo + x[i]) - Bo;
            tie |= re != ro;
            tot += re;  // (multiples of u; exact while the sum stays in the binade, and a sum that does not fails the test)
        }
        if (__ballot(tie) == 0ull) {
            const float sn = s + wave_sum(tot);
            if (sn < 2.0f * B) return sn;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s = block_step(s, x[i]);
    return s;
}

// scan_rows over an arbitrary set of lanes
__device__ __forceinline__ float scan_lanes(float s, float top, float ie, float io, unsigned long long segm, int lane, int &cross) {
    const bool in = lane_in(segm);
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
    cross = x ? __builtin_ctzll(x) : 64;
    return sp;
}


// 256 consecutive terms, FOUR CONSECUTIVE ones per lane (x[i]: term 4 lane + i), added to s in order.  A lane is to its four
// terms what a lane of the kernel above is to its row: accumulators started at B and at B + u give its increments for an
// even and an odd sum in front of it, scan_lanes composes the lanes (ties by parity, exact prefix sums) and names the first
// lane whose sum would leave the binade; that lane's four terms are then added one by one, as the reference does, and the
// lanes behind it start over on the new grid.  Every commit passes scan_lanes' test (sum < 2B), which also vouches for the
// accumulators of the lanes it commits (terms >= 0: an increment below B means the accumulator never left [B, 2B)).
__device__ __forceinline__ float flat_add_chunk(float s, const float (&x)[4], int lane) {
    if (__ballot((x[0] != 0.0f) | (x[1] != 0.0f) | (x[2] != 0.0f) | (x[3] != 0.0f)) == 0ull) return s;
    unsigned long long live = ~0ull;
    while (live) {
        float B, u;
        int f;
        if (grid_of(s, B, u)) {
            const float Bo = B + u;
            float ae = B, ao = Bo;
#pragma unroll
            for (int i = 0; i < 4; ++i) ae = ae + x[i], ao = ao + x[i];
            const float sp = scan_lanes(s, 2.0f * B, ae - B, ao - Bo, live, lane, f);
            if (f >= 64) return rl(sp, 63);  // (live always ends at lane 63)
            const unsigned long long before = live & ((1ull << f) - 1ull);
            if (before) s = rl(sp, 63 - __builtin_clzll(before));
        } else {
            f = __builtin_ctzll(live);  // zero / tiny sum: the next lane's terms as the reference adds them
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s = s + rl(x[i], f);
        live &= ~((2ull << f) - 1ull);
    }
    return s;
}

// cycle stamps of MSA_SIM_MODE=64 (diagnostics): [0] prologue, [1] round loops, [2] stitching, [3] waves, [4] rounds,
// [6] wave lifetimes in 100 MHz ticks, [7] longest wave (cycles), [8] most rounds of a wave, [10] ordered rows,
// [11] their cycles; [13] [14] [15] [5] the prologue's parts (histogram, G, rows skipped, ordered first rows)
__device__ unsigned long long g_bx_stamps[16];
__device__ unsigned int g_bx_rec[16384 * 8];  // per wave (diagnostics): column, cycles / 64 of the three phases, rounds, shortened rounds, [6] ordered rows

// U consecutive terms per lane (x[i]: term U lane + i) added to s in order: flat_add_chunk for any run length.  A row evaluated in
// the reference's order is there BECAUSE it holds a binade crossing: with the row's terms in ONE run per 64 U partners the scan
// names the crossing lane at once -- the lanes before it commit on the old grid, its U terms are added one by one, the lanes
// behind it start over on the new grid: two scans and U dependent adds per crossing, however long the row (with U = 4 a row of
// 1000 partners was four chunks, each with a scan of its own, the crossing chunk with two: 6 000 cycles per ordered row at 1000
// rows, a fifth of a wave's life outside the round loop).
template <int U>
__device__ __forceinline__ float flat_add_run(float s, const float (&x)[U], int lane) {
    float top = x[0];  // (terms >= 0: nothing to add if the largest is zero)
#pragma unroll
    for (int i = 1; i < U; ++i) top = fmaxf(top, x[i]);
    if (__ballot(top != 0.0f) == 0ull) return s;
    unsigned long long live = ~0ull;
    while (live) {
        float B, u;
        int f;
        if (grid_of(s, B, u)) {
            const float Bo = B + u;
            float ae = B, ao = Bo;
#pragma unroll
            for (int i = 0; i < U; ++i) ae = ae + x[i], ao = ao + x[i];
            const float sp = scan_lanes(s, 2.0f * B, ae - B, ao - Bo, live, lane, f);
            if (f >= 64) return rl(sp, 63);  // (live always ends at lane 63)
            const unsigned long long before = live & ((1ull << f) - 1ull);
            if (before) s = rl(sp, 63 - __builtin_clzll(before));
        } else {
            f = __builtin_ctzll(live);  // zero / tiny sum: the next lane's terms as the reference adds them
        }
#pragma unroll
        for (int i = 0; i < U; ++i) s = s + rl(x[i], f);
        live &= ~((2ull << f) - 1ull);
    }
    return s;
}

// Row j of one column in the reference's order: its partners are the rows k > j, U CONSECUTIVE ones per lane (k = kb + U lane +
// i), 64 U per chunk.  Dense: EVERY row behind j is read -- W[j][k] from the row-major upper triangle (16-byte loads), the
// column's code of row k (U bytes per lane) -- and a row that takes no part contributes W x 0 = 0, which changes no float32 sum;
// the rows up to j inside the first chunk read the zeros of the lower triangle, the codes behind row m are BX_SKIP; a lane whose
// partners lie behind the row's padding (k >= ldw, a multiple of 64) loads nothing.  No list, hence no load that depends on
// another (with the compacted lists of round 2 every chunk waited for a gather through entries it had to load first, under a
// vector-memory pipeline that the round loops of the other waves keep saturated: 20 000 cycles per ordered row).
// s = {numerator sum, denominator sum}; `which` selects the sums to advance.
// (Every argument by value: a struct passed by reference would live in scratch memory and make the caller's
// loop counters look divergent to the compiler.)
// cj: the column's code of row j (the callers hold it: loading it here put one more memory latency in front of the row's
// own loads, a fifth of an ordered row's time on short rows).
// (measured in round 5, every variant bit-exact: 4 / 8 / 16 partners per lane -- 99 / 76 / 85 kcycles of ordered rows per wave at
// 1000 x 4000 once the arguments are scalars again; sixteen spill, four cost a scan per 256 partners; the next chunk's loads in
// flight while this one is added up: no gain, twelve more live registers)
#ifndef MSA_ROW_U  // (A/B builds: tools/build_variant.sh)
#define MSA_ROW_U 8
#endif
constexpr int ROW_U = MSA_ROW_U;  // partners per lane and chunk of an ordered row
__device__ __noinline__ f2 exact_row(ColView cv, gf32p wup, ldsp tab, int j, uint32_t cj, int which, f2 s) {
    constexpr int U = ROW_U, Q = U / 4;
    const int lane = threadIdx.x & 63;
    // Everything the callers pass is wave-uniform, but the arguments of a function that is not inlined arrive in vector registers
    // and the compiler must take them for divergent: the sums' loop below (grid, crossing lane, the set of lanes still to commit)
    // then runs as a divergent loop -- per-lane 64-bit masks, exec-mask bookkeeping around every branch, twice the instructions.
    // Through readfirstlane they are scalars again.
    j = uni(j), cj = (uint32_t)uni((int)cj), which = uni(which);
    cv.m = uni(cv.m), cv.ldw = uni(cv.ldw);
    cv.colcode = uniform_ptr(cv.colcode), wup = uniform_ptr(wup);
    tab = (ldsp)(__attribute__((address_space(3))) void *)(uintptr_t)(uint32_t)uni((int)(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)tab);
    if (cj == BX_SKIP) return s;
    gf32p wr = wup + (size_t)j * cv.ldw;
    float s0 = unif(s.x), s1 = unif(s.y);
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const int lim = cv.ldw;  // (W rows and code columns are zeros / BX_SKIP from row m up to here; U divides 64)
    auto request = [&](int kb, f4 (&w)[Q], uint32_t (&c)[Q]) {
        const int k0 = kb + U * lane;
        if (k0 < lim) {
#pragma unroll
            for (int q = 0; q < Q; ++q) w[q] = *reinterpret_cast<const __attribute__((address_space(1))) f4 *>(wr + k0 + 4 * q);
            if constexpr (Q == 4) {
                const u4 cc = *reinterpret_cast<const __attribute__((address_space(1))) u4 *>(cv.colcode + k0);
                c[0] = cc[0], c[1] = cc[1], c[2] = cc[2], c[3] = cc[3];
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) c[q] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(cv.colcode + k0 + 4 * q);
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) w[q] = f4{0.0f, 0.0f, 0.0f, 0.0f}, c[q] = 0x01010101u * BX_SKIP;
        }
    };
    f4 w[Q];
    uint32_t c[Q];
    int kb = (j + 1) & ~(U - 1);
    request(kb, w, c);
    for (; kb < cv.m; kb += 64 * U) {
        const uint32_t rowoff = cj << 5;
        if (which & 1) {
            float x[U];
#pragma unroll
            for (int i = 0; i < U; ++i)
                x[i] = w[i >> 2][i & 3] * *reinterpret_cast<const __attribute__((address_space(3))) float *>(tab + (rowoff + ((c[i >> 2] >> (8 * (i & 3))) & 0xFFu)));
            s0 = flat_add_run<U>(s0, x, lane);
        }
        if (which & 2) {
            float x[U];
#pragma unroll
            for (int i = 0; i < U; ++i)
                x[i] = w[i >> 2][i & 3] * *reinterpret_cast<const __attribute__((address_space(3))) float *>(tab + (rowoff + 4 + ((c[i >> 2] >> (8 * (i & 3))) & 0xFFu)));
            s1 = flat_add_run<U>(s1, x, lane);
        }
        if (kb + 64 * U < cv.m) request(kb + 64 * U, w, c);
    }
    return f2{s0, s1};
}


// ---- per-lane grids ------------------------------------------------------------------------------------------
// A kernel that fixes ONE grid per chain and round (the binade of the sum at the round's first row: round 2's `bx`)
// pays for every binade crossing: a second pair of accumulators for predicted crossings, a round cut short for the
// others (the early rounds, where the sum doubles every few rows, are walked three to four times).  Nothing in the
// scheme needs the lanes of a round to agree on a grid: a lane's accumulators only have to start on the grid of the sum
// IN FRONT OF ITS OWN ROW.  That sum is not known before the round, but it is predictable to a fraction of a row:
//     increment of row j  ~  c * (valid rows behind j) * wbar_j * G[a_j],   G[a] = sum_b h_b D[b][a]
// (h = the column's residue frequencies; G = 1 for the denominator; wbar_j = the mean of W[j][k > j], one number per
// sequence computed behind the pair pass -- on alignments with families of close sequences the rows differ by a factor of
// several in what they add, and a predictor blind to that misjudges the binade of every row near a crossing: 52.9
// instead of 19.1 ordered rows per column on the ENOG-like alignment of tools/sim_by_data.py; c = the ratio measured
// on the previous round, on the ordered first row for the first round).  Every lane starts on the grid of its predicted sum, the round
// loop has a single version (two packed adds per step, no second grid), every round covers its 64 rows, and the
// stitching commits segment by segment: lanes whose grid is the binade the sum really is in are scanned as before;
// a row whose sum leaves the binade (the crossing row, ~9 per chain at m = 2000) or whose prediction was wrong
// (measured: none on the synthetic alignments) is evaluated in the reference's order.  A wrong prediction costs
// one ordered row, never exactness: every commit is checked against the true sum.
constexpr int LG_R0 = 1;  // rows evaluated in order before the first round (at least up to the first valid row)


// One chain at the end of a round: s before the round's first row; per lane the grid it accumulated on (Bl; 0 =
// plain sums from zero) and its increments for an even / odd sum.  Commits every row of vmask; returns the sum
// behind the round and the number of rows that went through the ordered path.
struct ResolvedLg {
    float s;
    int ordered;
    unsigned long long t_ordered;  // STAMP: cycles spent in the ordered rows
    unsigned long long rows;       // STAMP: the rows that went through the ordered path (bit = lane)
};
template <bool STAMP>
__device__ __forceinline__ ResolvedLg resolve_lg(ColView cv, gf32p wup, ldsp tab, int j0, uint32_t code, unsigned long long vmask, int kind,
                                                 float s, float Bl, float ie, float io) {
    const int lane = threadIdx.x & 63;
    unsigned long long t_ordered = 0, rows_ordered = 0;
    auto ordered_row = [&](int x, float from) {
        unsigned long long t0 = 0;
        if (STAMP) t0 = __builtin_readcyclecounter(), rows_ordered |= 1ull << x;
        const f2 r = exact_row(cv, wup, tab, j0 + x, (uint32_t)__builtin_amdgcn_readlane((int)code, x), kind ? 2 : 1, f2{from, from});
        const float v = unif(kind ? r.y : r.x);
        if (STAMP) t_ordered += __builtin_readcyclecounter() - t0;
        return v;
    };
    // a lane on plain sums that added nothing has only zero terms: it commits whatever the sum is
    const unsigned long long plain = __ballot(Bl == 0.0f), nonzero = __ballot(ie != 0.0f);
    unsigned long long todo = vmask & ~(plain & ~nonzero);
    int ordered = 0;
    while (todo) {
        const int f = __builtin_ctzll(todo);
        const float Bf = rl(Bl, f);
        float B, u;
        const bool g = grid_of(s, B, u);
        if (Bf == 0.0f || !g || B != Bf) {  // not the grid the sum is on: this row in order
            s = ordered_row(f, s);
            ++ordered;
            todo &= todo - 1;
            continue;
        }
        const unsigned long long segm = __ballot(Bl == Bf) & todo;
        int x;
        const float sp = scan_lanes(s, 2.0f * B, ie, io, segm, lane, x);
        if (x >= 64) {
            s = rl(sp, 63 - __builtin_clzll(segm));
            todo &= ~segm;
            continue;
        }
        // row x would leave the binade: commit the rows before it, evaluate it in order
        const unsigned long long before = segm & ((1ull << x) - 1ull);
        s = ordered_row(x, before ? rl(sp, 63 - __builtin_clzll(before)) : s);
        ++ordered;
        todo &= ~(segm & ((2ull << x) - 1ull));
    }
    return ResolvedLg{s, ordered, t_ordered, rows_ordered};
}

// The round loop with the lane's table column in LDS instead of 32 registers.  Per wave and round a [row][lane]
// float table (row a, lane l: D[a][a_j(l)]; the row behind the alphabet is zero) is written once; a step reads
// row a_k with ds_read_addtid_b32 (address = M0 + 4 lane: no address register, no VALU), M0 = the row's byte offset
// straight from the compacted list (16-bit entries) + the wave's table base.  A step is then
//     SALU  s_bfe (the entry), s_add (M0)            VALU  v_mul (W x D), two packed adds
//     LDS   one 256-byte row                          VMEM  one 256-byte row of W
// against 4 VALU + 3 SALU with the table in registers (s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off in front of
// the multiply), and 32 registers are free: 6 waves per SIMD, bound by the LDS allocation.  The denominator term
// is W itself (a lane whose row takes no part is ignored when the round is stitched).
// LDS reads and the scalar list loads share the LGKM counter and scalar loads return out of order, so the loop
// waits with lgkmcnt(0) once per 16 steps, at a point where everything outstanding was issued 16 steps earlier.
struct LgEntries {
    uint32_t o[16];  // row offsets in W
    uint32_t c[8];   // 16 table-row offsets (u16 each)
};
struct LgD {
    float d[16];
};

__device__ __forceinline__ void lg_issue_rows(LgD &D, const LgEntries &en, uint32_t base) {
    uint32_t t0, t1;
    // (an instruction between the M0 write and the LDS instruction that reads it: the hazard needs one wait state)
    asm volatile(
        "s_bfe_u32 %[t0], %[c0], 0x100000\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c0], 0x100010\n\tds_read_addtid_b32 %[d0]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c1], 0x100000\n\tds_read_addtid_b32 %[d1]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c1], 0x100010\n\tds_read_addtid_b32 %[d2]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c2], 0x100000\n\tds_read_addtid_b32 %[d3]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c2], 0x100010\n\tds_read_addtid_b32 %[d4]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c3], 0x100000\n\tds_read_addtid_b32 %[d5]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c3], 0x100010\n\tds_read_addtid_b32 %[d6]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c4], 0x100000\n\tds_read_addtid_b32 %[d7]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c4], 0x100010\n\tds_read_addtid_b32 %[d8]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c5], 0x100000\n\tds_read_addtid_b32 %[d9]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c5], 0x100010\n\tds_read_addtid_b32 %[d10]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c6], 0x100000\n\tds_read_addtid_b32 %[d11]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c6], 0x100010\n\tds_read_addtid_b32 %[d12]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_bfe_u32 %[t0], %[c7], 0x100000\n\tds_read_addtid_b32 %[d13]\n\t"
        "s_add_u32 m0, %[t0], %[base]\n\ts_bfe_u32 %[t1], %[c7], 0x100010\n\tds_read_addtid_b32 %[d14]\n\t"
        "s_add_u32 m0, %[t1], %[base]\n\ts_nop 0\n\tds_read_addtid_b32 %[d15]"
        : [d0] "=&v"(D.d[0]), [d1] "=&v"(D.d[1]), [d2] "=&v"(D.d[2]), [d3] "=&v"(D.d[3]), [d4] "=&v"(D.d[4]), [d5] "=&v"(D.d[5]),
          [d6] "=&v"(D.d[6]), [d7] "=&v"(D.d[7]), [d8] "=&v"(D.d[8]), [d9] "=&v"(D.d[9]), [d10] "=&v"(D.d[10]),
          [d11] "=&v"(D.d[11]), [d12] "=&v"(D.d[12]), [d13] "=&v"(D.d[13]), [d14] "=&v"(D.d[14]), [d15] "=&v"(D.d[15]),
          [t0] "=&s"(t0), [t1] "=&s"(t1)
        : [c0] "s"(en.c[0]), [c1] "s"(en.c[1]), [c2] "s"(en.c[2]), [c3] "s"(en.c[3]), [c4] "s"(en.c[4]), [c5] "s"(en.c[5]),
          [c6] "s"(en.c[6]), [c7] "s"(en.c[7]), [base] "s"(base)
        : "m0", "scc", "memory");
}
// everything on the LGKM counter has arrived; the rows read above may be used behind this point
__device__ __forceinline__ void lg_wait_rows(LgD &D) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(D.d[0]), "+v"(D.d[1]), "+v"(D.d[2]), "+v"(D.d[3]), "+v"(D.d[4]), "+v"(D.d[5]), "+v"(D.d[6]), "+v"(D.d[7]),
                   "+v"(D.d[8]), "+v"(D.d[9]), "+v"(D.d[10]), "+v"(D.d[11]), "+v"(D.d[12]), "+v"(D.d[13]), "+v"(D.d[14]),
                   "+v"(D.d[15])
                 :
                 : "memory");
}

// W rows by global_load_dword with the matrix base in an SGPR pair and list offset + lane offset in a VGPR (one v_add_u32;
// until round 4 the row address in an SGPR pair: two SALU), hand-issued, so the loop counts VMCNT itself: 16 loads are in flight at every step
// and nothing else of the loop is a vector-memory instruction.  (Measured against it in round 2, tools/ubench_wstream.hip:
// buffer_load_dword with the list offset as SGPR offset costs the texture addresser 8.6 instead of 5.3 CU-cycles per
// 256-byte wave-load -- 4.03 instead of 3.13 ms at C3; a 64-bit per-lane address left to the compiler one more VALU
// per step -- 3.38 ms.)
// BIG: the list holds row indices instead of byte offsets (any number of rows: 32-bit offsets end at m = 32768); the
// row address is multiplied out on the scalar unit, two more SALU instructions per step.
template <bool BIG>
__device__ __forceinline__ void round_loop_lds(const float *wlow_g, uint32_t rowbytes,
                                               const __attribute__((address_space(1))) uint32_t *off,
                                               const __attribute__((address_space(1))) uint16_t *trow, int tstart, int tend,
                                               uint32_t joff, uint32_t base, f2 &an, f2 &ad) {
    if constexpr (!BIG) {
        // THE loop: one asm statement with hand-allocated registers (msastat_lgloop.inc, generated by tools/gen_lg_loop.py, where
        // its design is described): blocks of 16 steps, 16 W rows in flight, VMCNT counted by hand, two steps' products per
        // v_pk_mul_f32, nothing requested behind the last block.  (The C++ loop below -- small asm statements around compiler-
        // scheduled code, what rounds 2 - 4 shipped for every size -- stays for the row-index lists beyond 32768 rows.)
        const uint64_t wbase = (uint64_t)uniform_ptr((const __attribute__((address_space(1))) char *)(uint64_t)wlow_g);
        const uint64_t offp = (uint64_t)uniform_ptr(off + tstart), trowp = (uint64_t)uniform_ptr(trow + tstart);
        uint32_t nblk = (uint32_t)uni((tend - tstart + 15) >> 4);
        asm volatile(LG_LOOP_ASM
                     : [an] "+v"(an), [ad] "+v"(ad), [nblk] "+s"(nblk)
                     : [joff] "v"(joff), [wuni] "s"(wbase), [base] "s"(base), [offp] "s"(offp), [trowp] "s"(trowp)
                     : LG_LOOP_CLOBBERS);
        return;
    }
    typedef const __attribute__((address_space(4))) uint32_t *c32;
    auto sload = [&](LgEntries &en, int t) {  // 16 entries = 64 + 32 bytes (t % 16 == 0)
        c32 po = (c32)(uint64_t)(off + t);
        c32 pc = (c32)(uint64_t)(trow + t);
#pragma unroll
        for (int i = 0; i < 16; ++i) en.o[i] = po[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) en.c[i] = pc[i];
    };
    // (every list offset + lane offset lies inside wlow: rows 0 .. m, columns below ldw)
    const uint64_t wuni = (uint64_t)uniform_ptr((const __attribute__((address_space(1))) char *)(uint64_t)wlow_g);
    auto wrow = [&](float &dst, uint32_t o) {
        if constexpr (!BIG) {
            // the list offset added to the lane offset on the VALU, the matrix base in an SGPR pair: one VALU instruction per
            // step instead of two scalar ones.  The addresser charges every address form the same (profiles/r04_ubench_wform.txt);
            // what differs is who computes the address, and the scalar unit -- one instruction per CU-cycle for all of its
            // waves -- was 75 % busy with 4.4 instructions per step while the VALU had room: 2.67 -> 2.57 ms at C3, 0.443 ->
            // 0.412 at 1000 x 4000, 8.14 -> 7.70 at 5000 x 5000 (the table row's address moved over as well -- v_add_u32_sdwa +
            // ds_read_b32 instead of M0 + ds_read_addtid_b32 -- tips it the other way: 2.80 ms, the VALU becomes the wall; a
            // quarter or half of the steps back on the scalar unit: no difference.  Now: texture addresser 90 %, VALU 86 %,
            // scalar unit 45 % busy)
            uint32_t vo;
            asm volatile("v_add_u32 %1, %2, %3\n\tglobal_load_dword %0, %1, %4" : "=v"(dst), "=&v"(vo) : "s"(o), "v"(joff), "s"(wuni) : "memory");
            return;
        }
        const uint64_t row = wuni + (uint64_t)o * (uint64_t)rowbytes;  // (BIG: the row index multiplied out on the scalar unit)
        asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(joff), "s"(row) : "memory");
    };
    auto bload = [&](float(&w)[16], const LgEntries &en) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wrow(w[i], en.o[i]);
    };
    // 16 steps; every W row is requested 16 steps before its use
    auto consume_reload = [&](float(&w)[16], const LgD &D, const LgEntries &next) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("s_waitcnt vmcnt(15)" : "+v"(w[i])::"memory");  // the oldest of the 16 loads in flight
            const float wi = w[i];
            const float x = wi * D.d[i];
            const f2 xn = {x, x}, xd = {wi, wi};
            wrow(w[i], next.o[i]);
            an += xn;
            ad += xd;
        }
    };
    float w[16];
    LgEntries eA, eB;
    LgD dA, dB;
    sload(eA, tstart);
    sload(eB, tstart + 16);
    bload(w, eA);
    lg_issue_rows(dA, eA, base);
#pragma unroll 1
    for (int t = tstart; t < tend; t += 32) {  // (the lists are padded: zero row of W, zero row of the table)
        lg_wait_rows(dA);  // rows t .. t+15 and the entries t+16 .. t+31
        lg_issue_rows(dB, eB, base);
        sload(eA, t + 32);
        __builtin_amdgcn_sched_barrier(0);  // (the scalar loads must not sink towards their use: the next wait would stall on them)
        consume_reload(w, dA, eB);
        lg_wait_rows(dB);
        lg_issue_rows(dA, eA, base);
        sload(eB, t + 48);
        __builtin_amdgcn_sched_barrier(0);
        consume_reload(w, dB, eA);
    }
    lg_wait_rows(dA);  // (nothing may stay in flight into the caller's LDS traffic)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int LG_WAVES_MAX = 8;   // waves per workgroup with a column each (they share the distance table; the launcher picks 4)
constexpr int LG_SPLIT_MAX = 16;  // waves of a workgroup that share ONE column (S > 1 below)
constexpr int LG_STATE = 40;      // floats of per-column state between launches (similarity_lg_body)

// The lane's table column lives in LDS (round_loop_lds; `nr` rows per wave behind the static arrays, within the first
// 64 KB: M0 holds 16 bits).
//
// S > 1 -- the waves of a workgroup share ONE column (tall alignments: fewer columns than the chip has wave slots, and a
// column of 20 000 rows is two million steps of one wave).  Every round's partner list is cut into S contiguous segments,
// wave w accumulates segment w on the SAME predicted grids (every wave computes them from the same numbers), and the
// increment pairs compose exactly as the rows of a round do (the header's proof, point 3): a lane whose sum in front has
// parity p receives inc_p of segment 0, which decides the parity in front of segment 1, and so on -- a few selects per
// lane and segment.  Wave 0 composes, stitches and publishes the new sums through LDS; three workgroup barriers per round.
// The accumulators of every segment start at {B, B + u}: if the composed increments stay below B (the commit test of the
// stitching), no segment's accumulator left the binade either, and a composed sum that reaches 2B is seen exactly as
// before (all terms >= 0: a partial sum >= B stays >= B).
template <bool STAMP, bool BIG, bool SPLIT, bool FIN>
__device__ __forceinline__ void similarity_lg_body(const LgAlign &A, int col, int ci, int nr, int r0_,
                                                   const float *__restrict__ tab_g, int jbegin, int jend) {
    const int S = SPLIT ? (int)(blockDim.x >> 6) : 1;  // waves that share the column
    const int m = A.m, n = A.n;
    const gf32p wup = (gf32p)(uint64_t)A.wup;
    __shared__ f2 tab[32 * 32];                  // {distance, both valid}[row code][column code], rows 28.. zero
    __shared__ uint32_t hist[LG_WAVES_MAX][32];  // residue counts of the wave's column
    __shared__ float gtab[LG_WAVES_MAX][32];     // G[a] = mean over the column's valid rows of D[.][a]
    __shared__ float seg[SPLIT ? LG_SPLIT_MAX - 1 : 1][4][64];  // SPLIT: the increment pairs of the segments 1 .. S-1
    __shared__ float hdr[8];                     // SPLIT: sums, ratios and position, published by wave 0
    extern __shared__ float ltab[];              // [wave][nr][64 lanes]  (S > 1: one table, shared)
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        f2 v = {0.0f, 0.0f};
        if (i < 29 * 32) v = reinterpret_cast<const f2 *>(tab_g)[i];
        tab[i] = v;
    }
    if (threadIdx.x < LG_WAVES_MAX * 32) (&hist[0][0])[threadIdx.x] = 0u;
    __syncthreads();
    const ldsp tabp = (ldsp)(const __attribute__((address_space(3))) void *)tab;
    const int lane = threadIdx.x & 63;
    const int wv = uni(threadIdx.x >> 6);
    const int wave = SPLIT ? 0 : wv;     // slot of hist / gtab / ltab
    const bool lead = !SPLIT || wv == 0;  // the wave that owns the column's sums
    if (col < 0) return;  // (a wave behind the last column; uniform per workgroup when the waves share a column)
    ColView cv;
    cv.off = uniform_ptr((const __attribute__((address_space(1))) uint32_t *)(uint64_t)A.voff + (size_t)col * A.ldk);
    cv.nvalid = uni(A.nvalid[col]);
    cv.colcode = uniform_ptr((gu8p)(uint64_t)A.codeT + (size_t)col * A.ldk);
    cv.ldw = A.ldw;
    cv.m = m;
    const __attribute__((address_space(1))) uint16_t *vtrow =
        uniform_ptr((const __attribute__((address_space(1))) uint16_t *)(uint64_t)A.vtrow + (size_t)col * A.ldk);
    // the mean weight of row j over its later partners: w_row_means' vector, or (FIN: the compact pipeline) the pair pass's
    // fixed-point row sums divided here
    auto wmean = [&](int j) -> float {
        if constexpr (FIN) return j < m - 1 ? (float)A.wsum[j] * (1.0f / 65536.0f) / (float)(m - 1 - j) : 0.0f;
        else return A.wbar[j];
    };
    // ... in two halves: what is loaded (requested a round ahead: behind the round loop, in front of the stitching) and what is
    // made of it (rows m .. of wbar / wsum are zero padding up to m + 64; a round may look further)
    auto wmean_raw = [&](int j) -> uint32_t {
        if (j >= m) return 0u;
        if constexpr (FIN) return A.wsum[j];
        else return __float_as_uint(A.wbar[j]);
    };
    auto wmean_of = [&](uint32_t raw, int j) -> float {
        if constexpr (FIN) return j < m - 1 ? (float)raw * (1.0f / 65536.0f) / (float)(m - 1 - j) : 0.0f;
        else return __uint_as_float(raw);
    };
    const int nv = cv.nvalid;
    unsigned long long t_pro = 0, t_loop = 0, t_res = 0, n_rounds = 0, n_ordered = 0, t_ord = 0, t0c = 0, rt0 = 0, n_both = 0;
    if (STAMP) {
        t0c = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    // A launch covers the rounds whose first row lies in [jbegin, jend) (launch_similarity_lg: one launch for everything, or
    // a few rounds per launch from ~1800 rows on).  Between launches a column's state lives in `state`: the two sums, the two
    // ratios of the predictor, its position (rows in front of the next round, valid rows in front of it, lanes of that
    // round the ordered prologue has done) and G.
    const bool resume = jbegin > 0;
    float *st = A.state ? A.state + (size_t)col * LG_STATE : nullptr;
    f2 s2 = {0.0f, 0.0f};
    float sn = 0.0f, sd = 0.0f, cn = 0.0f, cd = 0.0f;
    int j0 = 0, first = 0, tbase = 0;
    bool alive = true;
    if (lead) {
        if (!resume) {
            __builtin_amdgcn_s_setprio(3);  // (the ordered first row: as the stitching below)
            // the column's residue frequencies -> G
            // (four codes per lane and load, four loads in flight: a byte per lane and pass was sixteen memory latencies one
            // after the other at 1000 rows -- a third of the prologue; the column is padded with BX_SKIP up to a multiple of 256 rows)
            {
                const int mup = (m + 255) & ~255;
                for (int kb = 0; kb < mup; kb += 1024) {
                    uint32_t cw[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k0 = kb + 256 * q + 4 * lane;
                        cw[q] = k0 < mup ? *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(cv.colcode + k0) : 0x01010101u * BX_SKIP;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t ck = (cw[q] >> (8 * i)) & 0xFFu;
                            if (ck != BX_SKIP) atomicAdd(&hist[wave][ck >> 3], 1u);
                        }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            unsigned long long tp1 = 0, tp2 = 0, tp3 = 0;
            if (STAMP) tp1 = __builtin_readcyclecounter();
            if (lane < 32) {
                float g = 0.0f;
                for (int b = 0; b < 29; ++b) g += (float)hist[wave][b] * tab[b * 32 + lane].x;
                gtab[wave][lane] = nv > 0 ? g / (float)nv : 0.0f;
                if (st) st[8 + lane] = gtab[wave][lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

            if (STAMP) tp2 = __builtin_readcyclecounter();
            // the first rows in the reference's order: at least up to the first row that takes part
            float qn0 = 0.0f, qd0 = 0.0f;
            int jstart = 0, tb = 0;
            {
                // (the rows in front of the first one that takes part need nothing: skipped 64 at a time -- a column whose first
                // 570 rows are gaps spent 350 000 cycles walking them one dependent load at a time)
                for (; jstart + 64 < m - 1; jstart += 64)
                    if (__ballot(cv.colcode[jstart + lane] != BX_SKIP)) break;
                if (STAMP) tp3 = __builtin_readcyclecounter();
                bool seen = false;
                while (jstart < m - 1 && tb < nv && (jstart < (r0_ & 0xFFFF) || !seen)) {
                    const uint32_t cj = (uint32_t)uni((int)cv.colcode[jstart]);
                    if (cj != BX_SKIP) {
                        ++tb;
                        seen = true;
                        const float rem = (float)(nv - tb) * unif(wmean(jstart));
                        qd0 += rem;
                        qn0 += rem * gtab[wave][cj >> 3];
                        s2 = exact_row(cv, wup, tabp, jstart, cj, 3, s2);
                    }
                    ++jstart;
                }
            }
            sn = unif(s2.x), sd = unif(s2.y);
            qn0 = unif(qn0);
            qd0 = unif(qd0);
            // increment per unit of the estimate (any positive value is correct; a poor one costs ordered rows)
            cn = unif((sn > 0.0f && qn0 > 0.0f) ? sn / qn0 : 0.8f);
            cd = unif((sd > 0.0f && qd0 > 0.0f) ? sd / qd0 : 0.8f);
            if (STAMP) {
                const unsigned long long t1 = __builtin_readcyclecounter();
                t_pro = t1 - t0c;
                if (lane == 0) {  // the prologue's parts: histogram, G, rows skipped, the ordered first row(s)
                    atomicAdd(&g_bx_stamps[13], tp1 - t0c);
                    atomicAdd(&g_bx_stamps[14], tp2 - tp1);
                    atomicAdd(&g_bx_stamps[15], tp3 - tp2);
                    atomicAdd(&g_bx_stamps[5], t1 - tp3);
                }
                t0c = t1;
            }
            __builtin_amdgcn_s_setprio(0);
            j0 = jstart & ~63;
            first = jstart - j0;  // the round's lanes before it were evaluated above
            {
                const int r = j0 + lane;
                const bool v = r < jstart && cv.colcode[r] != BX_SKIP;
                tbase = tb - __builtin_popcountll(__ballot(v));  // valid rows before j0
            }
        } else {
            sn = unif(st[0]), sd = unif(st[1]), cn = unif(st[2]), cd = unif(st[3]);
            tbase = uni(reinterpret_cast<const int *>(st)[4]);
            j0 = uni(reinterpret_cast<const int *>(st)[5]);
            first = uni(reinterpret_cast<const int *>(st)[6]);
            alive = j0 < m - 1 && tbase < nv;  // else: the column was finished by an earlier launch (its sums are written)
            if (alive) {
                if (lane < 32) gtab[wave][lane] = st[8 + lane];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
        if (SPLIT && lane == 0) {
            hdr[0] = sn, hdr[1] = sd, hdr[2] = cn, hdr[3] = cd;
            reinterpret_cast<int *>(hdr)[4] = tbase;
            reinterpret_cast<int *>(hdr)[5] = j0;
            reinterpret_cast<int *>(hdr)[6] = first;
            reinterpret_cast<int *>(hdr)[7] = alive ? 1 : 0;
        }
    }
    if (SPLIT) {
        __syncthreads();
        sn = unif(hdr[0]), sd = unif(hdr[1]), cn = unif(hdr[2]), cd = unif(hdr[3]);
        tbase = uni(reinterpret_cast<const int *>(hdr)[4]);
        j0 = uni(reinterpret_cast<const int *>(hdr)[5]);
        first = uni(reinterpret_cast<const int *>(hdr)[6]);
        alive = uni(reinterpret_cast<const int *>(hdr)[7]) != 0;
    }
    if (!alive) return;  // (uniform per workgroup when the waves share a column)
    // the round's codes and mean weights, requested a round ahead (two memory latencies in front of every round loop otherwise:
    // with three to four waves per SIMD -- one alignment of 1000 x 4000 -- nothing hides them)
    uint32_t code_next = (uint32_t)cv.colcode[j0 + lane], wm_next = wmean_raw(j0 + lane);  // (j0 + 63 < ldk: the column's padding)
    for (; j0 < m - 1 && tbase < nv && j0 < jend; j0 += 64) {
        const int nrows = min(64, m - 1 - j0);
        const uint32_t craw = lane < nrows ? code_next : BX_SKIP;
        const uint32_t wm_raw = wm_next;
        const unsigned long long vall = __ballot(craw != BX_SKIP);
        const uint32_t cj8 = lane >= first ? craw : BX_SKIP;
        const unsigned long long vmask = __ballot(cj8 != BX_SKIP);
        first = 0;
        // predicted sum in front of every row -> the lane's grid
        const int behind = nv - (tbase + bits_below_lane(vall) + (lane_in(vall) ? 1 : 0));
        const bool takes = cj8 != BX_SKIP;
        // (the row's mean weight over its partners, a property of the alignment: rows of a tight family add less)
        const float qd = takes ? (float)behind * wmean_of(wm_raw, j0 + lane) : 0.0f;
        const float qn = takes ? qd * gtab[wave][cj8 >> 3] : 0.0f;
        float Bn, Bd, Qn, Qd;
        {
            const float Pd = wave_prefix(qd), Pn = wave_prefix(qn);
            float un, ud;
            grid_of(sn + cn * (Pn - qn), Bn, un);
            grid_of(sd + cd * (Pd - qd), Bd, ud);
            Qn = rl(Pn, 63);
            Qd = rl(Pd, 63);
        }
        // (the ulp of a grid is B * 2^-23, exact: grid_of only accepts exponents >= 30)
        const float Bno = Bn + Bn * 0x1p-23f, Bdo = Bd + Bd * 0x1p-23f;
        f2 an = {Bn, Bno}, ad = {Bd, Bdo};
        const uint32_t joff = 4u * (uint32_t)(j0 + lane);
        {
            // the lane's table column into the wave's [row][lane] table (zeros for a row that takes no part: column 28)
            float *lt = ltab + (size_t)wave * nr * 64 + lane;
            for (int a = SPLIT ? wv : 0; a < nr; a += S)
                lt[a * 64] = (*reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tabp + ((a << 8) + cj8))).x;
            const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)(ltab + (size_t)wave * nr * 64);
            int ts = tbase & ~15, te = nv;
            if (SPLIT) {
                __syncthreads();  // the shared table is complete
                const int per = ((nv - ts + 31) / 32 + S - 1) / S * 32;  // list entries per wave
                ts += per * wv;
                te = min(nv, ts + per);
            }
            if (ts < te) round_loop_lds<BIG>(A.wlow, 4u * (uint32_t)A.ldw, cv.off, vtrow, ts, te, joff, (uint32_t)uni((int)base), an, ad);
        }
        // (the next round's: in flight during the stitching)
        code_next = (uint32_t)cv.colcode[j0 + 64 + lane], wm_next = wmean_raw(j0 + 64 + lane);
        if (STAMP) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_loop += t1 - t0c;
            t0c = t1;
            ++n_rounds;
        }
        float ien = an.x - Bn, ion = an.y - Bno, ied = ad.x - Bd, iod = ad.y - Bdo;
        if (SPLIT) {
            if (wv > 0) {
                seg[wv - 1][0][lane] = ien, seg[wv - 1][1][lane] = ion, seg[wv - 1][2][lane] = ied, seg[wv - 1][3][lane] = iod;
            }
            __syncthreads();
            if (lead) {
                // segments in the order of their partners; the parity of the sum in front of a segment is the mantissa's last
                // bit of (grid base + what the segments before it added)
#pragma unroll 1
                for (int w = 0; w < S - 1; ++w) {
                    const float en = seg[w][0][lane], on = seg[w][1][lane], ed = seg[w][2][lane], od = seg[w][3][lane];
                    ien += (__float_as_uint(Bn + ien) & 1u) ? on : en;
                    ion += (__float_as_uint(Bno + ion) & 1u) ? on : en;
                    ied += (__float_as_uint(Bd + ied) & 1u) ? od : ed;
                    iod += (__float_as_uint(Bdo + iod) & 1u) ? od : ed;
                }
            }
        }
        if (lead) {
            // The stitching is a latency-bound instruction stream that competes for issue slots with the round loops of
            // the SIMD's other waves (which are bound by the W stream, not by issue): it runs at the highest wave priority.
            __builtin_amdgcn_s_setprio(3);
            // (a lane whose row takes no part: the LDS loop added W to its denominator accumulators -- ignored)
            const ResolvedLg rn = resolve_lg<STAMP>(cv, wup, tabp, j0, cj8, vmask, 0, sn, Bn, ien, ion);
            const ResolvedLg rd = resolve_lg<STAMP>(cv, wup, tabp, j0, cj8, vmask, 1, sd, Bd, takes ? ied : 0.0f, takes ? iod : 0.0f);
            const float sn1 = unif(rn.s), sd1 = unif(rd.s);
            if (sn1 > sn && Qn > 0.0f) cn = unif((sn1 - sn) / Qn);
            if (sd1 > sd && Qd > 0.0f) cd = unif((sd1 - sd) / Qd);
            sn = sn1;
            sd = sd1;
            __builtin_amdgcn_s_setprio(0);
            if (STAMP) {
                const unsigned long long t1 = __builtin_readcyclecounter();
                t_res += t1 - t0c;
                t0c = t1;
                n_ordered += (unsigned)(rn.ordered + rd.ordered);
                t_ord += rn.t_ordered + rd.t_ordered;
                n_both += (unsigned)__builtin_popcountll(rn.rows & rd.rows);
            }
            if (SPLIT && lane == 0) hdr[0] = sn, hdr[1] = sd, hdr[2] = cn, hdr[3] = cd;
        }
        tbase += __builtin_popcountll(vall);
        if (SPLIT) {
            __syncthreads();
            sn = unif(hdr[0]), sd = unif(hdr[1]), cn = unif(hdr[2]), cd = unif(hdr[3]);
        }
    }
    if (!lead) return;
    if (STAMP && lane == 0) {
        atomicAdd(&g_bx_stamps[0], t_pro);
        atomicAdd(&g_bx_stamps[1], t_loop);
        atomicAdd(&g_bx_stamps[2], t_res);
        atomicAdd(&g_bx_stamps[3], 1ull);
        atomicAdd(&g_bx_stamps[4], n_rounds);
        atomicAdd(&g_bx_stamps[6], __builtin_amdgcn_s_memrealtime() - rt0);  // 100 MHz ticks
        atomicMax(&g_bx_stamps[7], t_pro + t_loop + t_res);
        atomicMax(&g_bx_stamps[8], n_rounds);
        atomicAdd(&g_bx_stamps[10], n_ordered);
        atomicAdd(&g_bx_stamps[11], t_ord);
        atomicAdd(&g_bx_stamps[12], n_both);  // rows that were ordered in BOTH chains (one pass could serve both)
        if (ci < 16384) {
            unsigned int *r = g_bx_rec + 8 * ci;
            r[0] = (unsigned)col;
            r[1] = (unsigned)(t_pro >> 6);
            r[2] = (unsigned)(t_loop >> 6);
            r[3] = (unsigned)(t_res >> 6);
            r[4] = (unsigned)n_rounds;
            r[5] = 0;
            r[6] = (unsigned)n_ordered;
        }
    }
    if (st && lane == 0) {
        st[0] = sn, st[1] = sd, st[2] = cn, st[3] = cd;
        reinterpret_cast<int *>(st)[4] = tbase;
        reinterpret_cast<int *>(st)[5] = j0;
        reinterpret_cast<int *>(st)[6] = first;
    }
    if (col < n && lane == 0) {
        A.num_out[col] = sn;
        A.den_out[col] = sd;
        if constexpr (FIN) {  // (a column the ">= 80 % gaps" rule cuts has no valid row: den = 0, MDK = Q = 0 as sim_finish writes them)
            float q;
            A.mdk_out[col] = mdk_value(sn, sd, false, A.mdk_host, q);
            A.q_out[col] = q;
        }
    }
}

// Compiled for five waves per SIMD (the LDS allocation admits 20 waves per CU for a 20-letter alphabet).  Instantiations:
// 32-bit byte offsets in the lists or row indices (BIG, m > 32768); a wave per column, four per workgroup, or (SPLIT) a
// workgroup of blockDim / 64 waves per column.  One alignment (`one`, by value: no table to upload) or a batch: `table` + `items`
// ({alignment, column} per work item -- msa_trim_batch's launch over every column of every alignment of a shard).
template <bool STAMP, bool BIG, bool SPLIT, bool FIN>
__global__ __launch_bounds__(SPLIT ? 64 * LG_SPLIT_MAX : 64 * LG_WAVES_MAX) __attribute__((amdgpu_waves_per_eu(5, 5)))
void similarity_lg_kernel(LgAlign one, const LgAlign *__restrict__ table, const int32_t *__restrict__ items, int nitems, int nr, int r0,
                          const float *__restrict__ tab_g, int jbegin, int jend) {
    const int ci = SPLIT ? (int)blockIdx.x : blockIdx.x * (int)(blockDim.x >> 6) + uni(threadIdx.x >> 6);
    int col = -1;
    if (table) {
        if (ci < nitems) {
            // (a batch: `items` = the prefix sums of the alignments' column counts, `one.ncols` alignments)
            const int a = batch_find(reinterpret_cast<const int32_t *>(items), one.ncols, ci, col);
            // (wave-uniform copy of the descriptor: scalar loads)
            cu32p src = (cu32p)(uint64_t)(table + a);
            uint32_t words[sizeof(LgAlign) / 4];
#pragma unroll
            for (int i = 0; i < (int)(sizeof(LgAlign) / 4); ++i) words[i] = src[i];
            __builtin_memcpy(&one, words, sizeof(LgAlign));
        }
    } else if (ci < nitems) {
        col = one.cols ? uni(one.cols[ci]) : ci;
    }
    // (automated1 enqueues this kernel before the host knows which method the identity statistics select: the
    // kernel that computes them raises the gate when the similarity values will not be used)
    if (col >= 0 && one.gate && *one.gate) col = -1;
    similarity_lg_body<STAMP, BIG, SPLIT, FIN>(one, col, ci, nr, r0, tab_g, jbegin, jend);
}

// ---- the statistic as the reference writes it ------------------------------------------------------------------------
// Similarity::calculateVectors (statistics.pxd:55): for every column, rows j ascending, partners
// k > j ascending, both valid:  num += W[j][k] * D[a_j][a_k];  den += W[j][k]  -- float32, one add after the other, the
// product rounded before the add (-ffp-contract=off).  One lane per column, no cleverness: m^2 / 2 dependent adds per
// lane.  The cross-check of similarity_lg where the CPU oracle is out of reach (MSA_SIM_KERNEL=seq; tests only).
__global__ __launch_bounds__(64) void similarity_seq_kernel(const uint8_t *__restrict__ codeT, int64_t ldk, int m, int n,
                                                            const int32_t *__restrict__ cols, int ncols, const float *__restrict__ wup,
                                                            int ldw, const float *__restrict__ tab_g, float *__restrict__ num_out,
                                                            float *__restrict__ den_out) {
    __shared__ float dtab[29 * 32];
    for (int i = threadIdx.x; i < 29 * 32; i += 64) dtab[i] = tab_g[2 * i];
    __syncthreads();
    const int ci = blockIdx.x * 64 + threadIdx.x;
    if (ci >= ncols) return;
    const int col = cols[ci];
    if (col >= n) return;
    const uint8_t *code = codeT + (size_t)col * ldk;
    float num = 0.0f, den = 0.0f;
    for (int j = 0; j + 1 < m; ++j) {
        const uint32_t cj = code[j];
        if (cj == BX_SKIP) continue;
        const float *wr = wup + (size_t)j * ldw;
        const float *drow = dtab + (cj >> 3) * 32;
        for (int k = j + 1; k < m; ++k) {
            const uint32_t ck = code[k];
            if (ck == BX_SKIP) continue;
            const float w = wr[k];
            const float x = w * drow[ck >> 3];
            num = num + x;
            den = den + w;
        }
    }
    num_out[col] = num;
    den_out[col] = den;
}

// wbar[j] = mean of W[j][k] over k > j (the upper triangle of a row; 0 for the last row): the similarity kernel's
// predictor scales its per-row estimates with it.  Any order of summation: it is an estimate, nothing exact hangs on it.
__device__ __forceinline__ void w_row_means_body(const float *__restrict__ wup, int m, int ldw, float *__restrict__ wbar, int bx) {
    const int lane = threadIdx.x & 63;
    const int j = bx * 4 + (threadIdx.x >> 6);
    if (j >= m + 64) return;
    float s = 0.0f;
    if (j < m) {
        const float *r = wup + (size_t)j * ldw;
        for (int k = j + 1 + lane; k < m; k += 64) s += r[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    }
    if (lane == 0) wbar[j] = (j < m - 1) ? s / (float)(m - 1 - j) : 0.0f;  // (rows m .. m + 63: zeros, a round reads past the end)
}
__global__ __launch_bounds__(256) void w_row_means_kernel(const float *__restrict__ wup, int m, int ldw, float *__restrict__ wbar) {
    w_row_means_body(wup, m, ldw, wbar, (int)blockIdx.x);
}

// ---- identity row statistics (Cleaner::calculateSeqIdentity's consumers: selectMethod, getCutPointClusters) --------
// Per sequence: the float32 sum of its identities with every other sequence IN INDEX ORDER (/ (m - 1)), their
// maximum and minimum; then the sums of the row averages and of the row maxima in index order (/ m).  The terms are
// >= 0, so the sequential sums are evaluated a chunk of 256 terms at a time with the binade test of the similarity
// kernel's ordered rows (chunk_step): one wave per sequence instead of one dependent add chain per lane (83 + 24 us
// -> a few us at m = 2000), bit-identical.
__device__ __forceinline__ void identity_rows_body(const float *__restrict__ ident, int m, int ldw,
                                                   float *__restrict__ row_avg, float *__restrict__ row_max,
                                                   float *__restrict__ row_min, int bx) {
    const int lane = threadIdx.x & 63;
    const int i = uni((int)(bx * 4 + (threadIdx.x >> 6)));
    if (i >= m) return;
    const float *r = ident + (size_t)i * ldw;  // ident[i][j] == ident[j][i]
    float s = 0.0f, mx = 0.0f, mn = 1.0f;      // (getCutPointClusters starts its minimum at 1)
    for (int base = 0; base < m; base += 256) {
        float x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = base + 64 * c + lane;
            const bool in = t < m && t != i;
            const float v = in ? r[t] : 0.0f;
            x[c] = v;
            if (in) {
                mx = mx < v ? v : mx;
                mn = mn > v ? v : mn;
            }
        }
        s = chunk_step(s, x);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float a = __shfl_xor(mx, off, 64), b = __shfl_xor(mn, off, 64);
        mx = mx < a ? a : mx;
        mn = mn > b ? b : mn;
    }
    if (lane == 0) {
        row_avg[i] = s / (float)(m - 1);
        row_max[i] = mx;
        if (row_min) row_min[i] = mn;
    }
}
__global__ __launch_bounds__(256) void identity_rows_kernel(const float *__restrict__ ident, int m, int ldw,
                                                            float *__restrict__ row_avg, float *__restrict__ row_max,
                                                            float *__restrict__ row_min) {
    identity_rows_body(ident, m, ldw, row_avg, row_max, row_min, (int)blockIdx.x);
}

// (two waves: one per sum.  gate != nullptr: Cleaner::selectMethod's decision is taken here as well -- *gate = 1 when it
// selects gappyout, i.e. the similarity kernel enqueued behind this one has nothing to do; the host takes the same
// decision from the same two numbers when they arrive)
__device__ __forceinline__ void identity_final_body(const float *__restrict__ row_avg, const float *__restrict__ row_max,
                                                    int m, float *__restrict__ out2, int *__restrict__ gate, int *__restrict__ gate_host = nullptr) {
    __shared__ float res[2];
    const int lane = threadIdx.x & 63;
    const int which = uni((int)(threadIdx.x >> 6));
    if (which < 2) {  // (the compact pipeline calls this from a workgroup of eight waves)
        const float *src = which ? row_max : row_avg;
        float a = 0.0f;
        for (int base = 0; base < m; base += 256) {
            float xa[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = base + 64 * c + lane;
                xa[c] = t < m ? src[t] : 0.0f;
            }
            a = chunk_step(a, xa);
        }
        if (lane == 0) {
            a = a / (float)m;
            out2[which] = a;
            res[which] = a;
        }
    }
    __syncthreads();
    if (gate && threadIdx.x == 0) {
        const float avg = res[0], mx = res[1];
        int sel;  // msah::select_method, literally
        if (avg >= 0.55) sel = 1;
        else if (avg <= 0.38) sel = 2;
        else if (m <= 20) sel = 1;
        else if (mx >= 0.5 && mx <= 0.65) sel = 1;
        else sel = 2;
        *gate = sel == 1 ? 1 : 0;
        if (gate_host) *gate_host = sel == 1 ? 1 : 0;
    }
}
__global__ __launch_bounds__(128) void identity_final_kernel(const float *__restrict__ row_avg, const float *__restrict__ row_max,
                                                             int m, float *__restrict__ out2, int *__restrict__ gate) {
    identity_final_body(row_avg, row_max, m, out2, gate);
}

// codeT -> the compacted lists of one column's valid rows (one wave per column): byte offset of the row in W (or its
// index: `big`) and byte offset of its residue's row in the per-wave table; padded behind the last valid row with
// {zero row m, the table's zero row}.  6 bytes per residue (round 2 also kept the row index and the code for the
// ordered rows, which now read the column densely: 9 bytes).
__device__ __forceinline__ void bx_compact_body(const uint8_t *__restrict__ codeT, int64_t ldk, int m, int ncols_pad,
                                                uint32_t ldw4, uint32_t *__restrict__ voff, uint16_t *__restrict__ vtrow,
                                                int skiprow, int32_t *__restrict__ nvalid, int big, int bx) {
    const int lane = threadIdx.x & 63;
    const int col = bx * 4 + (threadIdx.x >> 6);
    if (col >= ncols_pad) return;
    const uint8_t *src = codeT + (size_t)col * ldk;
    uint32_t *po = voff + (size_t)col * ldk;
    uint16_t *pt = vtrow + (size_t)col * ldk;  // byte offset of the residue's row in a [row][64 lanes] float table
    int count = 0;
    for (int kb = 0; kb < m; kb += 64) {
        const int k = kb + lane;
        const uint32_t code = k < m ? src[k] : BX_SKIP;
        const unsigned long long mask = __ballot(code != BX_SKIP);
        if (code != BX_SKIP) {
            const int pos = count + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            po[pos] = big ? (uint32_t)k : (uint32_t)k * ldw4;  // (big: the row index; the kernel multiplies it out)
            pt[pos] = (uint16_t)((code >> 3) * 256u);
        }
        count += __builtin_popcountll(mask);
    }
    for (int64_t t = count + lane; t < ldk; t += 64) {
        po[t] = big ? (uint32_t)m : (uint32_t)m * ldw4;  // row m of W: zeros
        pt[t] = (uint16_t)(skiprow * 256);  // the table's zero row
    }
    if (lane == 0) nvalid[col] = count;
}
__global__ __launch_bounds__(256) void bx_compact_kernel(const uint8_t *__restrict__ codeT, int64_t ldk, int m, int ncols_pad,
                                                         uint32_t ldw4, uint32_t *__restrict__ voff, uint16_t *__restrict__ vtrow,
                                                         int skiprow, int32_t *__restrict__ nvalid, int big) {
    bx_compact_body(codeT, ldk, m, ncols_pad, ldw4, voff, vtrow, skiprow, nvalid, big, (int)blockIdx.x);
}

// raw bytes -> column-major codes (64 x 64 tiles through LDS); first bad residue through atomicMin as in the
// other encode kernels.  Columns cut by the ">= 80 % gaps" rule and all padding hold BX_SKIP.
__device__ __forceinline__ void sim_encode_cm_body(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                   const uint8_t *__restrict__ lut_g,
                                                   const int32_t *__restrict__ gaps_w, uint8_t *__restrict__ codeT,
                                                   int64_t ldk, int ncols_pad,
                                                   unsigned long long *__restrict__ err_key, int bx, int by) {
    __shared__ uint8_t lut[256];
    __shared__ uint8_t tile[64][68];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = bx * 64 + tx;
    bool skipcol = true;
    if (c < n) skipcol = gaps_w ? (((float)gaps_w[c] / (float)m) >= 0.8f) : false;
    for (int r = ty; r < 64; r += 4) {
        const int row = by * 64 + r;
        uint32_t code = BX_SKIP;
        if (row < m && c < n && !skipcol) {
            const uint32_t byte = raw[(size_t)row * ld + c];
            code = lut[byte];  // 8 x table row, 224 = skipped, 0xFE / 0xFF = bad symbol
            if (code >= 0xFEu) {
                const unsigned long long key = ((unsigned long long)c << 40) | ((unsigned long long)row << 16) |
                                               ((unsigned long long)(code & 1u) << 8) | byte;
                atomicMax(err_key, ~key);  // (kept complemented: 0 = none, the largest complement = the first residue)
                code = BX_SKIP;
            }
        }
        tile[r][tx] = (uint8_t)code;
    }
    __syncthreads();
    const int64_t k = (int64_t)by * 64 + tx;
    for (int q = ty; q < 64; q += 4) {
        const int col = bx * 64 + q;
        if (col < ncols_pad && k < ldk) codeT[(size_t)col * ldk + k] = tile[tx][q];
    }
}

__global__ __launch_bounds__(256) void sim_encode_cm_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                            const uint8_t *__restrict__ lut_g,
                                                            const int32_t *__restrict__ gaps_w, uint8_t *__restrict__ codeT,
                                                            int64_t ldk, int ncols_pad,
                                                            unsigned long long *__restrict__ err_key) {
    sim_encode_cm_body(raw, m, n, ld, lut_g, gaps_w, codeT, ldk, ncols_pad, err_key, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- batches (msa_trim_batch): the kernels above for every alignment of a shard in one launch each -- a table of BAlign
// descriptors, a block finds its alignment by bisection over the family's prefix sums of blocks (msastat_kernels.hip) ----
__device__ __forceinline__ BAlign batch_desc(const BAlign *table, int a) {  // wave-uniform copy: scalar loads
    cu32p src = (cu32p)(uint64_t)(table + a);
    uint32_t words[sizeof(BAlign) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(BAlign) / 4); ++i) words[i] = src[i];
    BAlign d;
    __builtin_memcpy(&d, words, sizeof(BAlign));
    return d;
}
__global__ __launch_bounds__(256) void w_row_means_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    w_row_means_body(d.w, d.m, d.ldw, d.wbar, local);
}
__global__ __launch_bounds__(256) void identity_rows_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    if (!d.gated) return;  // (only automated1 needs the identity statistics)
    identity_rows_body(d.ident, d.m, d.ldw, d.row_avg, d.row_max, nullptr, local);
}
__global__ __launch_bounds__(128) void identity_final_batch_kernel(const BAlign *__restrict__ table) {  // a block per alignment
    const BAlign d = batch_desc(table, (int)blockIdx.x);
    if (!d.gated) return;
    identity_final_body(d.row_avg, d.row_max, d.m, reinterpret_cast<float *>(d.flags + 4), d.flags + 6);
}
__global__ __launch_bounds__(256) void sim_encode_cm_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K,
                                                                  const uint8_t *__restrict__ lut_g) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    const int nbx = d.ncols_pad / 64;
    sim_encode_cm_body(d.raw, d.m, d.n, d.ld, lut_g, d.gaps, d.codeT, d.ldk, d.ncols_pad,
                       reinterpret_cast<unsigned long long *>(d.flags + 2), local % nbx, local / nbx);
}
// ---- small alignments in a batch: one LANE per column ---------------------------------------------------------------------
// A column of 100 rows is a dozen binade crossings and a prologue around two rounds of 64 rows: the wave-per-column kernel
// spends 50 us on it, nearly all of it in ordered rows.  With thousands of such columns in one launch (a batch of small
// alignments: 10^5 - 10^6 columns) the statistic as the reference writes it is the better kernel: a lane per column, the two
// nested loops, one add after the other -- W[j][k] is the same for the 64 columns of a wave (a scalar load), their codes of
// row k are 64 consecutive bytes of a ROW-major code array, the {distance, valid} pair comes from the LDS table; 5 VALU
// instructions per step for 64 pairs, no prologue, no stitching, and enough waves to hide the add latency.  m * m / 2 steps per
// wave against ~170 000 cycles of fixed cost per column.  The wave's codes are staged in LDS (64 columns x m bytes, four codes
// to a dword), the weights of a row arrive sixteen per scalar load: up to 128 rows (msa_trim_batch picks per group).
__global__ __launch_bounds__(256) void sim_encode_rm_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K,
                                                                  const uint8_t *__restrict__ lut_g) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    // a block = 256 columns x 16 rows
    const int nbx = (int)((d.ld + 255) / 256);
    const int c = (local % nbx) * 256 + (int)threadIdx.x, r0 = (local / nbx) * 16;
    if (c >= d.ld) return;
    const bool skipcol = c >= d.n || (((float)d.gaps[c] / (float)d.m) >= 0.8f);
    for (int r = r0; r < min(d.m, r0 + 16); ++r) {
        uint32_t code = BX_SKIP;
        if (!skipcol) {
            const uint32_t byte = d.raw[(size_t)r * d.ld + c];
            code = lut[byte];
            if (code >= 0xFEu) {
                const unsigned long long key = ((unsigned long long)c << 40) | ((unsigned long long)r << 16) |
                                               ((unsigned long long)(code & 1u) << 8) | byte;
                atomicMax(reinterpret_cast<unsigned long long *>(d.flags + 2), ~key);
                code = BX_SKIP;
            }
        }
        d.codeR[(size_t)r * d.ld + c] = (uint8_t)code;
    }
}

constexpr int COLS_MAX_M = 128;  // rows of an alignment the lane-per-column kernel takes (its code tiles live in LDS)
__global__ __launch_bounds__(256) void similarity_cols_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K,
                                                                    int total, const float *__restrict__ tab_g) {
    __shared__ f2 tab[32 * 32];  // {distance, both valid}[row code][column code], rows 28.. zero
    // the wave's 64 columns x m codes, [lane][k] packed four to a dword; 33 dwords per lane: lanes on different banks
    __shared__ uint32_t tile[4][64][COLS_MAX_M / 4 + 1];
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        f2 v = {0.0f, 0.0f};
        if (i < 29 * 32) v = reinterpret_cast<const f2 *>(tab_g)[i];
        tab[i] = v;
    }
    __syncthreads();
    const ldsp tabp = (ldsp)(const __attribute__((address_space(3))) void *)tab;
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
    const int item = (int)blockIdx.x * 4 + wave;  // a wave = 64 columns of one alignment
    if (item >= total) return;
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, item, local));
    if (d.gated && d.flags[6]) return;  // (selectMethod took gappyout on the device)
    const int c = local * 64 + lane, m = d.m;
    const bool in = c < d.n;
    const int64_t ld = d.ld;
    const gu8p code = (gu8p)(uint64_t)(d.codeR + (in ? c : 0));
    uint32_t *mine = tile[wave][lane];
    for (int k4 = 0; k4 < COLS_MAX_M / 4 + 1; ++k4) {  // (rows behind m: codes that take no part)
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * k4 + i;
            const uint32_t ck = (in && k < m) ? (uint32_t)code[(size_t)k * ld] : BX_SKIP;
            packed |= ck << (8 * i);
        }
        mine[k4] = packed;
    }
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef const __attribute__((address_space(4))) f16v *c16;
    float num = 0.0f, den = 0.0f;
    for (int j = 0; j + 1 < m; ++j) {
        const uint32_t cj = (mine[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (cj == BX_SKIP) continue;
        const float *wr = d.w + (size_t)j * d.ldw;  // the row's weights: the same for every column (scalar loads, 16 at a time)
        const ldsp row = tabp + (cj << 5);
        // (chunks of 16 partners from the one that holds row j + 1: W[j][k <= j] = 0, the upper triangle is strict, and a
        // product with it adds +0 to either sum)
        for (int kb = (j + 1) & ~15; kb < m; kb += 16) {
            const f16v w = *(c16)(uint64_t)(wr + kb);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c4 = mine[(kb >> 2) + q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t ck = (c4 >> (8 * i)) & 0xFFu;
                    const f2 de = *reinterpret_cast<const __attribute__((address_space(3))) f2 *>(row + ck);
                    const float wk = w[4 * q + i];
                    const float x = wk * de.x, y = wk * de.y;
                    num = num + x;
                    den = den + y;
                }
            }
        }
    }
    if (in) {
        d.simnum[c] = num;
        d.simden[c] = den;
    }
}

__global__ __launch_bounds__(256) void bx_compact_batch_kernel(const BAlign *__restrict__ table, const int32_t *__restrict__ prefix, int K,
                                                               int skiprow, int big) {
    int local;
    const BAlign d = batch_desc(table, batch_find(prefix, K, (int)blockIdx.x, local));
    bx_compact_body(d.codeT, d.ldk, d.m, d.ncols_pad, (uint32_t)d.ldw * 4u, d.off, d.trow, skiprow, d.nvalid, big, local);
}


// ---- small alignments, one at a time: the FLAT similarity kernel ---------------------------------------------------------
// The binade-exact kernel above spends ~35 us on a column of 46 rows and ~100 us on one of 200: a prologue, a predictor,
// ordered rows at every binade crossing -- fixed costs that m^2 / 2 terms per column repay only from a few hundred rows on.
// Below that the terms of a column are few enough to be taken as ONE sequence: the pairs (j, k > j) of the column's valid
// rows in the reference's order, 256 at a time through scan_lanes (flat_add_chunk below).  Its valid rows (index, code) are compacted into LDS from the
// column-major codes; W[j][k] a gather from the (L2-resident) upper triangle; the distance from the LDS table.  No lists, no
// predictor, no per-row state.  Two waves per column (one per sum); the numerator's wave writes MDK and Q itself (mdk_value).
constexpr int FLAT_ROWS_MAX = 512;
__global__ __launch_bounds__(256) void similarity_flat_kernel(LgAlign A, const float *__restrict__ tab_g) {
    __shared__ f2 tab[32 * 32];                    // {distance, both valid}[row code][column code], rows 28.. zero
    __shared__ uint32_t rows[4][FLAT_ROWS_MAX];    // per wave: the column's valid rows, index | table row << 16
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        f2 v = {0.0f, 0.0f};
        if (i < 29 * 32) v = reinterpret_cast<const f2 *>(tab_g)[i];
        tab[i] = v;
    }
    __syncthreads();
    // TWO waves per column, one per sum: the chains are independent, and a lone wave on its SIMD issues an instruction of a
    // dependent chain every ten cycles or so -- a second wave costs the walk over the pairs twice and still halves the time
    __shared__ float sums[4];
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
    const int col = (int)blockIdx.x * 2 + (wave >> 1);
    const bool denominator = (wave & 1) != 0;
    // (automated1: selectMethod may have taken gappyout on the device)
    const bool active = col < A.n && !(A.gate && *A.gate);
    float sum = 0.0f;
    if (active) {
    const int m = A.m;
    uint32_t *mine = rows[wave];
    const gu8p code = (gu8p)(uint64_t)(A.codeT + (size_t)col * A.ldk);
    int nv = 0;
    for (int kb = 0; kb < m; kb += 64) {
        const int k = kb + lane;
        const uint32_t ck = k < m ? (uint32_t)code[k] : BX_SKIP;
        const unsigned long long mask = __ballot(ck != BX_SKIP);
        if (ck != BX_SKIP) mine[nv + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint32_t)k | ((ck >> 3) << 16);
        nv += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const int T = nv * (nv - 1) / 2;  // pairs (a < b) of list positions, lexicographic: row a holds the nv - 1 - a pairs (a, b > a)
    const gf32p wup = (gf32p)(uint64_t)A.wup;
    const int ldw = A.ldw;
    // Eight terms per lane and pass (two chunks of 256, four consecutive terms of each), requested a pass ahead: with two
    // waves per SIMD (two per column, ~1000 columns) nothing else hides the gather's latency.
    constexpr int U = 8;
    int rt = 4 * lane;  // the lane's first term of the next pass
    const float fn = (float)(2 * nv - 1);
    auto request = [&](float (&w)[U], uint32_t (&ti)[U]) {
        uint32_t ea[U], eb[U];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the lane's first term of the chunk -> (row a, place p): the largest a with a nv - a (a + 1) / 2 <= t, from the root
            // of an exact integer below 2^24 (two corrections cover a root that is off by an ulp); its three neighbours by a carry
            // into the next row.  Straight-line code: the eight gathers below leave back to back.  Terms behind the last one
            // read the last one's operands and count as zero (the caller masks them).
            const int t0 = rt + 256 * h;
            const int t = t0 < T ? t0 : T - 1;
            int a = (int)((fn - sqrtf(fn * fn - 8.0f * (float)t)) * 0.5f);
            a = a > nv - 2 ? nv - 2 : a;
            a -= (a * nv - a * (a + 1) / 2 > t) ? 1 : 0;
            a += (a + 1 <= nv - 2 && (a + 1) * nv - (a + 1) * (a + 2) / 2 <= t) ? 1 : 0;
            int pl = t - (a * nv - a * (a + 1) / 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ea[4 * h + i] = mine[a], eb[4 * h + i] = mine[a + 1 + pl];
                const bool adv = t0 + i + 1 < T, wrap = pl + 1 >= nv - 1 - a;
                a += (adv && wrap) ? 1 : 0;
                pl = adv ? (wrap ? 0 : pl + 1) : pl;
            }
        }
        rt += 64 * U;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            w[i] = wup[(size_t)(ea[i] & 0xFFFFu) * ldw + (eb[i] & 0xFFFFu)];
            ti[i] = (ea[i] >> 16) * 32 + (eb[i] >> 16);
        }
    };
    if (T > 0) {
        float w[U];
        uint32_t ti[U];
        request(w, ti);
        for (int t0 = 0; t0 < T; t0 += 64 * U) {
            float wn[U];
            uint32_t tn[U];
            request(wn, tn);
            float x[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                if (t0 + 256 * (i >> 2) + 4 * lane + (i & 3) >= T) w[i] = 0.0f;
                const float d = tab[ti[i]].x;  // (both waves read it: no branch around eight LDS loads)
                x[i] = denominator ? w[i] : w[i] * d;
            }
#pragma unroll
            for (int h = 0; h < U; h += 4) {
                if (t0 + 64 * h >= T) break;  // (the pass's second chunk starts at term t0 + 256)
                const float xs[4] = {x[h], x[h + 1], x[h + 2], x[h + 3]};
                sum = flat_add_chunk(sum, xs, lane);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) w[i] = wn[i], ti[i] = tn[i];
        }
    }
    }
    if (lane == 0) sums[wave] = sum;
    __syncthreads();
    if (active && !denominator && lane == 0) {
        const float sn = sums[wave], sd = sums[wave + 1];
        if (A.num_out) A.num_out[col] = sn, A.den_out[col] = sd;
        float q;
        A.mdk_out[col] = mdk_value(sn, sd, false, A.mdk_host, q);
        A.q_out[col] = q;
    }
}

// ---- the compact pipeline of a small alignment (CompactArgs, msastat_kernels.h) -------------------------------------------
// Front kernel, one launch for everything that reads the rows.  Blocks by role:
//   * a COLUMN block owns 64 columns over all rows, read once (64 consecutive bytes of a row per wave and load): their gap and
//     indetermination counts (plain stores: no atomics, hence no memset), the bit planes of the pair pass (a thread per row on
//     the 64 x 64 tile in LDS: planes_of_row), and -- the ">= 80 % gaps" cut follows from the block's own counts -- the
//     column-major codes and the compacted lists.  The block's codes live in LDS ([column][row] bytes) between the pass over
//     the rows and the pass that writes them out: nothing is read back from memory.  A bad residue counts only in a column
//     that is not cut: the first bad row of every column by an LDS minimum, the block's first bad residue and its non-ASCII
//     verdict into the block's own slots;
//   * a ROW block: the residues of four sequences (row_nongap_body).
// Block 0 zeroes the device's flag words, the identity statistics' ticket and the pair pass's row sums (nothing of this launch
// touches them).  Results go to the state block's mirror in pinned host memory as well (`hres`: same offsets; the rows' totals,
// the slots, MDK and Q only there): no copy back, the host folds the slots into the two flag words after the wait.
constexpr int COMPACT_ROWS_MAX = 512;  // rows of the LDS code array of a column block
constexpr int COMPACT_TEAMS_MAX = 4;  // 64-row tiles a column block works on at once (a team of four waves each)
template <bool SIM>
__device__ __forceinline__ void compact_column_block(const CompactArgs &a, int b) {
    constexpr int LDC = COMPACT_ROWS_MAX + 4;  // bytes per column (4 past a multiple of 128: consecutive columns on different banks)
    constexpr int TM = COMPACT_TEAMS_MAX;
    __shared__ uint8_t lut[256];
    __shared__ uint8_t codes[SIM ? 64 * LDC : 4];
    __shared__ uint32_t rawt[SIM ? TM * 64 * 17 : 1];  // per team a tile's bytes, [row][64 columns + 4]
    __shared__ uint32_t cnt[2][4 * TM][64];
    __shared__ uint32_t firstbad[64];
    __shared__ uint8_t skipc[64];
    __shared__ int anybad;
    // The workgroup is `teams` teams of four waves; team t takes the tiles t, t + teams, ...: a tile is a chain of small
    // latencies (loads, LDS, two barriers), and a column block of 500 rows that walked its eight tiles one after the other
    // took 58 us where the whole alignment's pair pass takes 18.
    const int teams = (int)(blockDim.x >> 8), team = (int)(threadIdx.x >> 8), tid = (int)(threadIdx.x & 255);
    const int tx = threadIdx.x & 63, ty = tid >> 6, wave = (int)(threadIdx.x >> 6), nwaves = 4 * teams;
    const int m = a.m, n = a.n;
    const int64_t ld = a.ld, ldk = a.ldk;
    const int c = b * 64 + tx;
    const bool inb = c < n;
    uint8_t lutbyte = 0;
    if (SIM) {
        if (threadIdx.x < 256) lutbyte = a.lut[threadIdx.x];  // (requested with the first tile's rows; stored behind them)
        if (threadIdx.x < 64) firstbad[threadIdx.x] = 0xFFFFFFFFu;
        if (threadIdx.x == 0) anybad = 0;
    }
    const uint32_t indet = a.indet4 & 0xFFu;
    const uint8_t *col0 = a.raw + (inb ? c : 0);
    const int mtiles = (m + 63) / 64, passes = (mtiles + teams - 1) / teams;
    uint32_t g = 0, x = 0;
    uint32_t next[16];  // (sixteen rows requested before the first is looked at, and a pass ahead of the one being worked on)
    auto request = [&](int by) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = by * 64 + ty + 4 * i;
            next[i] = (row < m && inb) ? (uint32_t)col0[(size_t)row * ld] : 0x100u;  // 0x100: outside, counts as nothing
        }
    };
    request(team);
    uint32_t *myraw = rawt + (SIM ? team * 64 * 17 : 0);
    for (int ps = 0; ps < passes; ++ps) {
        const int by = ps * teams + team;  // (a team without a tile in the last pass walks rows behind m: nothing counts)
        uint32_t bytes[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) bytes[i] = next[i];
        if (ps + 1 < passes) request(by + teams);
        if (SIM && ps == 0) {
            if (threadIdx.x < 256) lut[threadIdx.x] = lutbyte;
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t byte = bytes[i];
            g += byte == (uint32_t)'-';
            x += byte == indet;
            if (SIM) {
                const int row = by * 64 + ty + 4 * i;
                uint32_t code = BX_SKIP;
                if (byte < 0x100u) {
                    code = lut[byte];  // 8 x table row, 224 = skipped, 0xFE / 0xFF = bad symbol
                    if (code >= 0xFEu) {
                        atomicMin(&firstbad[tx], ((uint32_t)row << 16) | ((code & 1u) << 8) | byte);
                        code = BX_SKIP;
                    }
                }
                if (row < mtiles * 64) codes[tx * LDC + row] = (uint8_t)code;
                reinterpret_cast<uint8_t *>(myraw)[(ty + 4 * i) * 68 + tx] = (uint8_t)(byte < 0x100u ? byte : (uint32_t)'-');
            }
        }
        if (SIM) {
            __syncthreads();
            {  // four threads per row of the team's tile, 16 columns each: the row's two chunk words of every plane
                const int r = tid >> 2, q = tid & 3;
                const int row = by * 64 + r;
                uint32_t o[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) o[p] = 0;
                uint32_t bad = 0;
                if (row < m) bad = planes_of_quarter(myraw + r * 17 + q * 4, b * 64 + q * 16, n, a.indet4, o);
                const int chunk = b * 2 + (q >> 1);
                const size_t pstride = (size_t)a.nchunk * a.m_pad;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    uint32_t v = o[p] << ((q & 1) * 16);
                    v |= (uint32_t)__shfl_xor((int)v, 1, 64);  // (the other half of the chunk word: the neighbouring lane)
                    if ((q & 1) == 0 && row < a.m_pad && chunk < a.nchunk) a.planes[p * pstride + (size_t)chunk * a.m_pad + row] = v;
                }
                if (bad) anybad = 1;
            }
            __syncthreads();
        }
    }
    if (SIM && wave == 1) {  // the rows between the last pass and m_pad: zero in every plane
        const uint32_t zero[2][8] = {};
        for (int row = passes * teams * 64 + tx; row < a.m_pad; row += 64) planes_store(a.planes, a.nchunk, a.m_pad, b, row, zero);
    }
    cnt[0][wave][tx] = g;
    cnt[1][wave][tx] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t G = 0, X = 0;
        for (int w = 0; w < nwaves; ++w) G += cnt[0][w][tx], X += cnt[1][w][tx];
        if (inb) {
            a.gaps[c] = (int32_t)G;
            a.indets[c] = (int32_t)X;
            a.hres[a.h_gaps + c] = (int32_t)G;
            a.hres[a.h_indets + c] = (int32_t)X;
        }
        if (SIM) {
            const bool skip = !inb || (((float)(int32_t)G / (float)m) >= 0.8f);
            skipc[tx] = skip ? 1 : 0;
            // the block's first bad residue: smallest column, then smallest row -- the key of sim_encode_cm, complemented
            unsigned long long key = ~0ull;
            if (!skip && firstbad[tx] != 0xFFFFFFFFu) key = ((unsigned long long)c << 40) | firstbad[tx];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), off, 64);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = other < key ? other : key;
            }
            if (tx == 0) {
                key = key == ~0ull ? 0ull : ~key;
                a.hres[a.h_slots + 2 * b] = (int32_t)(uint32_t)key;
                a.hres[a.h_slots + 2 * b + 1] = (int32_t)(uint32_t)(key >> 32);
                a.hres[a.h_slots + 2 * (a.ncols_pad / 64) + b] = anybad;
            }
        }
    }
    if (SIM) {
        __syncthreads();
        // a wave per column, lane = row: the codes (coalesced) and the compacted lists, as sim_encode_cm and bx_compact write them
        const int lane = tx;
        const uint32_t ldw4 = (uint32_t)a.ldw * 4u;
        for (int q = wave; q < 64; q += nwaves) {
            const size_t col = (size_t)b * 64 + q;
            const bool skip = skipc[q] != 0;
            uint8_t *ct = a.codeT + col * ldk;
            uint32_t *po = a.voff + col * ldk;
            uint16_t *pt = a.vtrow + col * ldk;
            int count = 0;
            for (int kb = 0; kb < mtiles * 64; kb += 64) {
                const int k = kb + lane;
                const uint32_t code = (k < m && !skip) ? (uint32_t)codes[q * LDC + k] : BX_SKIP;
                ct[k] = (uint8_t)code;
                if (!a.lists) continue;  // (the flat similarity kernel reads the codes alone)
                const unsigned long long mask = __ballot(code != BX_SKIP);
                if (code != BX_SKIP) {
                    const int pos = count + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    po[pos] = a.big ? (uint32_t)k : (uint32_t)k * ldw4;
                    pt[pos] = (uint16_t)((code >> 3) * 256u);
                }
                count += __builtin_popcountll(mask);
            }
            if (!a.lists) continue;
            for (int64_t k = (int64_t)mtiles * 64 + lane; k < ldk; k += 64) ct[k] = (uint8_t)BX_SKIP;
            for (int64_t t = count + lane; t < ldk; t += 64) {
                po[t] = a.big ? (uint32_t)m : (uint32_t)m * ldw4;  // row m of W: zeros
                pt[t] = (uint16_t)(a.skiprow * 256);               // the table's zero row
            }
            if (lane == 0) a.nvalid[col] = count;
        }
    }
}
template <bool SIM>
__global__ __launch_bounds__(256 * COMPACT_TEAMS_MAX) void compact_front_kernel(CompactArgs a) {
    const int b = (int)blockIdx.x;
    const int ncb = a.ncols_pad / 64;
    if (b == 0) {
        if (threadIdx.x < 32) a.flags[threadIdx.x] = 0;
        if (SIM) {
            if (threadIdx.x == 0) a.scratch[0] = 0;  // the ticket of the identity statistics
            for (int i = threadIdx.x; i < a.m_pad + 64; i += (int)blockDim.x) a.wsum[i] = 0u;
        }
    }
    if (b < ncb) compact_column_block<SIM>(a, b);
    else if (threadIdx.x < 256) row_nongap_body(a.raw, a.m, a.n, a.ld, nullptr, a.hres + a.h_rowtot, b - ncb);
}

// automated1: the identity statistics -- a wave per sequence (identity_rows_body), and in the workgroup that finishes last (a
// ticket) the two means and Cleaner::selectMethod's decision (identity_final_body): one launch for the ordinary path's two.
__global__ __launch_bounds__(256) void compact_identity_kernel(CompactArgs a) {
    identity_rows_body(a.ident, a.m, a.ldw, a.row_avg, a.row_max, nullptr, (int)blockIdx.x);
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // (release: the workgroup's four rows, device-wide -- its waves' stores are complete behind the barrier)
        last = atomicAdd(a.scratch, 1) == (int)gridDim.x - 1;
        __threadfence();  // (acquire)
    }
    __syncthreads();
    if (!last) return;
    identity_final_body(a.row_avg, a.row_max, a.m, reinterpret_cast<float *>(a.hres + 4), a.flags + 6, a.hres + 6);
}

}  // namespace

// leading dimension of the per-column lists: the valid rows, then >= 192 padding entries (a block of the ordered
// path, two prefetched groups of the round loop)
int64_t bx_ldk(int m) { return ((int64_t)m + 63) / 64 * 64 + 256; }  // (the ordered path reads up to 191 entries past the last valid one, the round loops up to 63)
int bx_cols_pad(int n) { return (n + 1 + 63) / 64 * 64; }  // at least one all-skipped column behind the last one
size_t bx_wlow_rows(int m) { return (size_t)m + 2; }         // row m: zeros (the padding entries of the lists point there)

void launch_sim_encode_cm(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut,
                          const int32_t *gaps_w, uint8_t *codeT, unsigned long long *err_key) {
    const int64_t ldk = bx_ldk(m);
    const int ncp = bx_cols_pad(n);
    dim3 grid((unsigned)(ncp / 64), (unsigned)(ldk / 64));
    sim_encode_cm_kernel<<<grid, 256, 0, s>>>(raw, m, n, ld, lut, gaps_w, codeT, ldk, ncp, err_key);
}

// the lists hold 32-bit byte offsets of the W rows while those fit (row m included); beyond that, row indices
bool lg_big(int m, int ldw) { return tuning().lg_big != 0 || ((uint64_t)m + 2) * (uint64_t)ldw * 4u > 0xFFFFFFFFull; }

void launch_bx_compact(hipStream_t s, const uint8_t *codeT, int m, int n, int ldw, int npos, uint32_t *voff, uint16_t *vtrow,
                       int32_t *nvalid) {
    const int ncp = bx_cols_pad(n);
    bx_compact_kernel<<<(ncp + 3) / 4, 256, 0, s>>>(codeT, bx_ldk(m), m, ncp, (uint32_t)ldw * 4u, voff, vtrow, npos, nvalid,
                                                    lg_big(m, ldw) ? 1 : 0);
}

// the sums of the columns that are not in the list are left alone.  Any number of rows (see lg_big).
size_t lg_state_floats(int n) { return (size_t)bx_cols_pad(n) * LG_STATE; }

// rounds per launch: 0 = everything in one launch.  The waves of an XCD share a 64-column block of W (m x 256 bytes) in its
// L2 only while they walk the same round, and light columns run ahead of heavy ones: a launch boundary every few rounds
// brings the columns back together.  Measured (tools/sim_rounds_ab.py, profiles/r03_sim_rounds.jsonl): 2000 x 10000 2.87 ->
// 2.74 ms, 3583 x 7287 8.65 -> 6.0, 5000 x 5000 15.4 -> 8.2 (45 % L2 hits in one launch), 8000 x 3000 28.7 -> 13.9; six to ten
// rounds per launch are all about the same; below ~1800 rows (W within the L2's reach whatever the drift) one launch wins.
int lg_rounds_per_launch(int m) {
    if (tuning().lg_rounds >= 0) return tuning().lg_rounds;
    return m >= 1800 ? 6 : 0;
}

// Waves per column.  A wave per column fills the chip from ~5000 columns on (256 CUs x 4 SIMDs x 5 waves); below that the
// wave slots that would stay empty go to the columns' partner lists instead, as long as every wave still has a list worth
// walking (>= ~1000 rows per wave) -- up to eight: measured at 20000 x 500 and 40000 x 300, 4 / 6 / 8 / 12 / 16 waves per column
// are within 15 % of one another (the W stream then misses the L2 a fifth of the time: columns of different weight drift
// apart inside a launch, and that, not the number of waves, sets the time) and sixteen need 1024-thread workgroups that no
// longer all fit the chip at once.  MSA_LG_SPLIT forces a value (tests: any S up to 16 at any size).
int lg_split(int m, int ncols, int cus) {
    const int forced = tuning().lg_split;
    if (forced > 0) return std::min(forced, LG_SPLIT_MAX);
    const long slots = (long)cus * 20;
    int s = (int)std::min<long>(8, slots / std::max(ncols, 1));
    s = std::min(s, m / 1024);
    return std::max(s, 1);
}

// one alignment (`one.cols` lists the `one.ncols` columns to evaluate), or a batch (`table`, `items`: device memory).
// *launches_out: kernel launches issued.
static int launch_lg(hipStream_t s, const LgAlign &one, const LgAlign *table, const int32_t *items, int nitems, int max_m, int ldw_for_big,
                     int npos, const void *tab, bool with_state, int split, int *launches_out) {
    const int r0 = tuning().lg_r0 >= 0 ? tuning().lg_r0 : LG_R0;
    const int nr = npos + 1;  // table rows per wave in LDS: the alphabet + the zero row
    if (launches_out) *launches_out = 0;
    if (nitems <= 0) return 0;
    // a wave per column: four waves per workgroup -- five workgroups (20 waves) per CU for a 20-letter alphabet, and a
    // workgroup's slots are refilled as soon as its four columns are done (eight per workgroup: 4.0 instead of 3.8 ms at C3)
    const int waves = 4;
    const float *t = static_cast<const float *>(tab);
    const bool stamp = (tuning().sim_mode & 64) != 0, big = lg_big(max_m, ldw_for_big);
    const int rounds = (std::max(max_m, 2) - 1 + 63) / 64, per = with_state ? lg_rounds_per_launch(max_m) : 0;
    const int launches = per > 0 && per < rounds ? (rounds + per - 1) / per : 1;
    LgAlign a1 = one;
    if (launches == 1) a1.state = nullptr;  // (nothing to carry over)
    const bool fin = one.mdk_out && !table && !stamp && !big && split == 1 && launches == 1;
#define LG_LAUNCH(STAMP_, BIG_, SPLIT_)                                                                                \
    do {                                                                                                               \
        auto kernel = similarity_lg_kernel<STAMP_, BIG_, SPLIT_, false>;                                               \
        const size_t dyn = (size_t)(SPLIT_ ? 1 : waves) * nr * 256;                                                    \
        const unsigned grid = SPLIT_ ? (unsigned)nitems : (unsigned)((nitems + waves - 1) / waves);                    \
        const int e = set_max_lds_once((const void *)kernel, (int)dyn);                                                \
        if (e) return e;                                                                                               \
        for (int l = 0; l < launches; ++l)                                                                             \
            kernel<<<grid, SPLIT_ ? 64 * split : 64 * waves, dyn, s>>>(a1, table, items, nitems, nr, r0, t, l * per * 64, \
                                                                       l + 1 == launches ? 0x7FFFFFFF : (l + 1) * per * 64); \
    } while (0)
#define LG_BY_BIG(STAMP_, SPLIT_)                    \
    do {                                             \
        if (big) LG_LAUNCH(STAMP_, true, SPLIT_);    \
        else LG_LAUNCH(STAMP_, false, SPLIT_);       \
    } while (0)
    if (fin) {  // (lg_finishes: no stamps, 32-bit offsets, a wave per column, one launch)
        auto kernel = similarity_lg_kernel<false, false, false, true>;
        const size_t dyn = (size_t)waves * nr * 256;
        const int e = set_max_lds_once((const void *)kernel, (int)dyn);
        if (e) return e;
        kernel<<<(unsigned)((nitems + waves - 1) / waves), 64 * waves, dyn, s>>>(a1, table, items, nitems, nr, r0, t, 0, 0x7FFFFFFF);
    } else if (stamp && split > 1) LG_BY_BIG(true, true);
    else if (stamp) LG_BY_BIG(true, false);
    else if (split > 1) LG_BY_BIG(false, true);
    else LG_BY_BIG(false, false);
#undef LG_BY_BIG
#undef LG_LAUNCH
    LaunchNote &note = launch_note();
    note.sim_kind = big ? 3 : 2, note.lg_split = split, note.lg_launches = launches, note.lg_fin = fin ? 1 : 0;
    if (launches_out) *launches_out = launches;
    return 0;
}

bool lg_finishes(const LgAlign &one, int cus) {
    const int rounds = (std::max(one.m, 2) - 1 + 63) / 64, per = one.state ? lg_rounds_per_launch(one.m) : 0;
    return one.mdk_out && (tuning().sim_mode & 64) == 0 && !lg_big(one.m, one.ldw) && lg_split(one.m, one.ncols, cus) == 1 &&
           !(per > 0 && per < rounds);
}
int launch_similarity_lg(hipStream_t s, const LgAlign &one, int npos, const void *tab, int cus, int *launches_out) {
    const int split = lg_split(one.m, one.ncols, cus);
    return launch_lg(s, one, nullptr, nullptr, one.ncols, one.m, one.ldw, npos, tab, one.state != nullptr, split, launches_out);
}

// every column of every alignment of a shard in one grid (a wave per column, in the alignments' own column order: a
// column the ">= 80 % gaps" rule cuts has no valid row and costs its wave a histogram pass): `table` and `colprefix` (K + 1
// prefix sums of the alignments' column counts) in device memory; the alignments' m <= max_m, byte offsets in every list
// (the caller keeps alignments beyond 32768 rows out of a batch)
int launch_similarity_lg_batch(hipStream_t s, const LgAlign *table, const int32_t *colprefix, int K, int ncols_total, int max_m, int npos,
                               const void *tab, bool with_state, int *launches_out) {
    LgAlign none = {};
    none.ncols = K;
    return launch_lg(s, none, table, colprefix, ncols_total, max_m, 64, npos, tab, with_state, 1, launches_out);
}

void launch_w_row_means_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) w_row_means_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
}
void launch_identity_stats_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks) {
    if (blocks > 0) identity_rows_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K);
    if (K > 0) identity_final_batch_kernel<<<K, 128, 0, s>>>(table);
}
void launch_sim_encode_rm_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int blocks, const uint8_t *lut) {
    if (blocks > 0) sim_encode_rm_batch_kernel<<<blocks, 256, 0, s>>>(table, prefix, K, lut);
}
// prefix: 64-column groups per alignment; total = their number
void launch_similarity_cols_batch(hipStream_t s, const BAlign *table, const int32_t *prefix, int K, int total, const void *tab) {
    if (total > 0) similarity_cols_batch_kernel<<<(total + 3) / 4, 256, 0, s>>>(table, prefix, K, total, static_cast<const float *>(tab));
    launch_note() = LaunchNote{5, 0, 1, 0, launch_note().pair_kind, launch_note().pair_waves};
}
void launch_sim_lists_batch(hipStream_t s, const BAlign *table, const int32_t *prefix_encode, int blocks_encode, const int32_t *prefix_compact,
                            int blocks_compact, int K, const uint8_t *lut, int npos) {
    if (blocks_encode > 0) sim_encode_cm_batch_kernel<<<blocks_encode, 256, 0, s>>>(table, prefix_encode, K, lut);
    if (blocks_compact > 0) bx_compact_batch_kernel<<<blocks_compact, 256, 0, s>>>(table, prefix_compact, K, npos, tuning().lg_big != 0 ? 1 : 0);
}


// words of the slots behind the state block's mirror: a first-bad-residue key (two words) and a non-ASCII word per column block
size_t compact_slot_words(int n) { return (size_t)3 * (bx_cols_pad(n) / 64) + 2; }
size_t compact_scratch_words(int m, int n) { return (size_t)2 + (std::max(m, 1) + 127) / 128 * 128 + 64 + 8; }
void launch_compact_front(hipStream_t s, const CompactArgs &a) {
    const unsigned blocks = (unsigned)(a.ncols_pad / 64 + (a.m + 3) / 4);
    const int teams = std::min(COMPACT_TEAMS_MAX, std::max(1, (a.m + 63) / 64));  // a team of four waves per 64-row tile, up to four
    if (a.sim) compact_front_kernel<true><<<blocks, 256 * teams, 0, s>>>(a);
    else compact_front_kernel<false><<<blocks, 256 * teams, 0, s>>>(a);
}
// the flat similarity kernel: any alignment of up to FLAT_ROWS_MAX rows whose codes exist (A.codeT, A.wup, A.mdk_out, A.q_out)
int flat_rows_max() { return FLAT_ROWS_MAX; }
void launch_similarity_flat(hipStream_t s, const LgAlign &one, const void *tab) {
    if (one.n > 0) similarity_flat_kernel<<<(unsigned)((one.n + 1) / 2), 256, 0, s>>>(one, static_cast<const float *>(tab));
    LaunchNote &note = launch_note();
    note.sim_kind = 1, note.lg_split = 0, note.lg_launches = 1, note.lg_fin = 1;
}
void launch_compact_identity(hipStream_t s, const CompactArgs &a) {
    compact_identity_kernel<<<(unsigned)((a.m + 3) / 4), 256, 0, s>>>(a);
}

// mean weight of every row over its later partners (m + 64 floats): the similarity kernel's predictor reads it
void launch_w_row_means(hipStream_t s, const float *wup, int m, int ldw, float *wbar) {
    w_row_means_kernel<<<(m + 64 + 3) / 4, 256, 0, s>>>(wup, m, ldw, wbar);
}

// the plain sequential kernel (cross-check): same column list, same outputs
int launch_similarity_seq(hipStream_t s, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, const float *wup, int ldw,
                          const void *tab, float *num_out, float *den_out) {
    if (ncols <= 0) return 0;
    similarity_seq_kernel<<<(ncols + 63) / 64, 64, 0, s>>>(codeT, bx_ldk(m), m, n, cols, ncols, wup, ldw, static_cast<const float *>(tab),
                                                           num_out, den_out);
    LaunchNote &note = launch_note();
    note.sim_kind = 4, note.lg_split = 0, note.lg_launches = 1, note.lg_fin = 0;
    return 0;
}

void launch_identity_stats(hipStream_t s, const float *ident, int m, int ldw, float *row_avg, float *row_max, float *out2,
                           float *row_min, int *gate) {
    identity_rows_kernel<<<(m + 3) / 4, 256, 0, s>>>(ident, m, ldw, row_avg, row_max, row_min);
    identity_final_kernel<<<1, 128, 0, s>>>(row_avg, row_max, m, out2, gate);
}

extern "C" int msa_debug_bx_stamps(unsigned long long *out16, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bx_stamps), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bx_stamps), z, sizeof(z));
        void *rec = nullptr;
        if (hipGetSymbolAddress(&rec, HIP_SYMBOL(g_bx_rec)) == hipSuccess) rc |= (int)hipMemset(rec, 0, sizeof(unsigned int) * 16384 * 8);
    }
    return rc;
}

extern "C" int msa_debug_bx_records(unsigned int *out, int nwaves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bx_rec), sizeof(unsigned int) * 8 * (size_t)(nwaves < 16384 ? nwaves : 16384));
}

}  // namespace msak
