// msastat_sums.h -- device helpers of the order-preserving float32 sums (internal): wave scans, the binade test, the step that
// adds a run of consecutive terms to a sum in the reference's order.  Shared by msastat_simx.hip (the similarity kernel) and
// msastat_small.hip (layouts, identity statistics, the kernels of small alignments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msastat_kernels.h"

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


// 64 U consecutive terms, U CONSECUTIVE ones per lane (x[i]: term U lane + i), added to s in order.  A lane is to its U terms what a
// lane of the similarity kernel is to its row: accumulators started at B and at B + u give its increments for an even and an odd
// sum in front of it, scan_lanes composes the lanes (ties by parity, exact prefix sums) and names the first lane whose sum would
// leave the binade; that lane's terms are then added one by one, as the reference does, and the lanes behind it start over on
// the new grid.  Every commit passes scan_lanes' test (sum < 2B), which also vouches for the accumulators of the lanes it commits
// (terms >= 0: an increment below B means the accumulator never left [B, 2B)).  Two scans and U dependent adds per crossing,
// however long the run.
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

}  // namespace
}  // namespace msak
