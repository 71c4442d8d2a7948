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
//   * helpers (scans, exact_row: one row in the reference's order) and `similarity_bx`, the first kernel of the
//     family: ONE grid per chain and round, a second pair of accumulators for predicted crossings, rounds cut short
//     for the others, the lane's table column in 32 VGPRs (MSA_SIM_KERNEL=bx; section 5.7 of DESIGN.md);
//   * `similarity_lg` (the default): every lane on the grid of its own predicted sum, one loop version, the table
//     column in a per-wave LDS table read by ds_read_addtid_b32, W rows by hand-issued global loads.  Work mapping:
//     one wave = one column (the columns the ">= 80 % gaps" rule zeroes never get a wave), lane = one of 64
//     consecutive rows j of the round, one step per VALID partner row k behind the round's first row (compacted
//     list): 3 VALU + 1 LDS + 1 VMEM instruction per 64 terms.  Bound by the vector L1's bandwidth (the W stream:
//     texture addresser 92 % busy), not by VALU issue or add latency;
//   * `similarity_lg2` (MSA_SIM_KERNEL=q2): two columns per wave sharing the W loads -- measured, not faster;
//   * the identity row statistics (sequential float32 sums through the same chunk test);
//   * the layout kernels (column-major codes, compacted lists, union lists) and the launchers.
// Limits: m < 32000 (16-bit row indices, 32-bit W offsets); above that the chain kernels of msastat_kernels.hip run.
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
constexpr int BX_Q = 1;            // columns per wave, advanced in turn.  2 (the host then pairs a heavy column with a
                                   // light one, all waves resident from the start) measured 6.2 instead of 5.1 ms at C3

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

// The valid rows of one column, compacted (bx_compact_kernel): entry t is the t-th row whose residue takes part.
struct ColView {
    const __attribute__((address_space(1))) uint32_t *off;  // byte offset of that row in W (row * ldw * 4); padding: a zero row
    const __attribute__((address_space(1))) uint16_t *row;  // its row index; padding: m
    gu8p code;                                               // its code (8 x table row); padding: BX_SKIP
    int nvalid;
    gu8p colcode;                                            // the column's codes by row (codeT), BX_SKIP for a row that takes no part
    int ldw;
    int compact;  // lanes of a round are consecutive entries of the compacted list (else consecutive rows)
    int lastpad;  // the last entry of the lists (padding)
    int nr;       // 0: `tab` is the {distance, valid} table [32][32]; > 0: the replicated float table [nr][nr][32 copies]
};

// Row j of one column in the reference's order: its partners are the valid rows behind it, i.e. the entries
// tfirst .. nvalid-1 of the compacted list (tfirst = number of valid rows <= j), 64 per block (lane = partner).
// s = {numerator sum, denominator sum}; `which` selects the sums to advance.
// (Every argument by value: a struct passed by reference would live in scratch memory and make the caller's
// loop counters look divergent to the compiler.)
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

__device__ __noinline__ f2 exact_row(ColView cv, gf32p wup, ldsp tab, int j, int tfirst, int which, f2 s) {
    const int lane = threadIdx.x & 63;
    if (cv.compact) j = uni((int)cv.row[j]);  // (a round's rows are list entries there)
    const uint32_t cj = cv.colcode[j];
    if (uni((int)cj) == (int)BX_SKIP) return s;
    gf32p wr = wup + (size_t)j * cv.ldw;
    float s0 = s.x, s1 = s.y;
    // Chunks of four blocks (256 terms).  Entries behind the lists' padding are clamped onto its last entry.
    auto entries = [&](int t, uint32_t(&k)[4], uint32_t(&c)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = min(t + 64 * i + lane, cv.lastpad);
            k[i] = cv.row[e];
            c[i] = cv.code[e];
        }
    };
    // (replicated table: entry (a, b) has 32 copies, lane l reads copy l % 32 -- no bank conflict whatever the codes)
    const uint32_t rrow = cv.nr ? (((cj == BX_SKIP ? (uint32_t)cv.nr - 1u : cj >> 3) * (uint32_t)cv.nr) << 7) + ((uint32_t)(lane & 31) << 2) : 0u;
    auto values = [&](const uint32_t(&k)[4], const uint32_t(&c)[4], float(&w)[4], f2(&de)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w[i] = wr[k[i]];  // wup[j][k]; whatever lies at column m is multiplied by a skipped code's zeros
            if (cv.nr) {
                const bool skip = c[i] == BX_SKIP;
                const uint32_t b = skip ? (uint32_t)cv.nr - 1u : c[i] >> 3;
                de[i].x = *reinterpret_cast<const __attribute__((address_space(3))) float *>(tab + (rrow + (b << 7)));
                de[i].y = skip ? 0.0f : 1.0f;
            } else {
                de[i] = *reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tab + ((cj << 5) + c[i]));
            }
        }
    };
    // (list entries a chunk ahead; the W gather and the table gather of a chunk are issued together.  A deeper
    // pipeline -- values a chunk ahead as well -- measured no faster: what an ordered row waits for is its turn to
    // issue, which is why the callers raise the wave priority, and it costs 32 registers)
    uint32_t k[4], c[4];
    int tb = tfirst;
    entries(tb, k, c);
    for (; tb < cv.nvalid; tb += 256) {
        float w[4];
        f2 de[4];
        values(k, c, w, de);
        entries(tb + 256, k, c);
        if (which & 1) {
            const float x[4] = {w[0] * de[0].x, w[1] * de[1].x, w[2] * de[2].x, w[3] * de[3].x};
            s0 = chunk_step(s0, x);
        }
        if (which & 2) {
            const float x[4] = {w[0] * de[0].y, w[1] * de[1].y, w[2] * de[2].y, w[3] * de[3].y};
            s1 = chunk_step(s1, x);
        }
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

// One chain (numerator or denominator of the column) at the end of a round of `limit` rows starting at row j0
// (tbase valid rows lie before j0; vmask: the valid rows of the round).  Returns {the sum after every row (per lane), the new limit}: the limit shrinks when the rows behind
// some point cannot be committed.
struct Resolved {
    float sp;
    int limit;
};
__device__ __noinline__ Resolved resolve_chain(ColView cv, gf32p wup, ldsp tab, int j0, int tbase, unsigned long long vmask,
                                               int kind, float s, float ie, float io, float ie2, float io2, int limit,
                                               bool dual) {
    // (row j0 + x of the round; its partners start at entry tbase + (valid rows of the round up to and including x))
    auto tfirst = [&](int x) { return tbase + __builtin_popcountll(vmask & ((2ull << x) - 1ull)); };
    const int lane = threadIdx.x & 63;
    float B, u;
    if (!grid_of(s, B, u)) {
        // no binade yet (sum still zero): the accumulators are plain sums; the first row that contributes is
        // evaluated in order and ends the round
        const unsigned long long nz = __ballot(lane < limit && ie != 0.0f);
        if (!nz) return Resolved{s, limit};
        const int x = __builtin_ctzll(nz);
        const f2 r = exact_row(cv, wup, tab, j0 + x, tfirst(x), kind ? 2 : 1, f2{s, s});
        const float sx = kind ? r.y : r.x;
        return Resolved{lane < x ? s : sx, x + 1};
    }
    int x;
    const float sp = scan_rows(s, 2.0f * B, ie, io, 0, limit, lane, x);
    if (x >= limit) return Resolved{sp, limit};
    // row x would leave the binade: evaluate it in order
    const float before = x > 0 ? rl(sp, x - 1) : s;
    const f2 r = exact_row(cv, wup, tab, j0 + x, tfirst(x), kind ? 2 : 1, f2{before, before});
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
// [5] rounds x chains that carried the second grid, [6] wave lifetimes in 100 MHz ticks, [7] longest wave (cycles),
// [8] most rounds of a wave, [9] shortened rounds
__device__ unsigned long long g_bx_stamps[16];
__device__ unsigned int g_bx_rec[16384 * 8];  // per wave (diagnostics): column, cycles / 64 of the three phases, rounds, shortened rounds

// The partner loop of one round: entries tstart .. tend-1 of the compacted list (valid rows only), one step per
// partner row k:
//   W[k][j(lane)]  one coalesced buffer load (SGPR row offset straight from the list + per-lane column offset);
//   D[a_k][a_j]    from the lane's own copy of its table column, T[a] = D[a][a_j(lane)], held in 32 VGPRs and
//                  indexed by the wave-uniform a_k (relative VGPR addressing: no LDS access in the loop, so the
//                  scalar prefetch of the list is the only thing on the LGKM counter);
//   one packed multiply {W, W} x {D, e} (e = 1 for a lane whose row takes part) and two packed adds per grid.
// DN / DD: the numerator / denominator chain also accumulates on the next grid.
typedef float v32f __attribute__((ext_vector_type(32)));

template <bool DN, bool DD>
__device__ __forceinline__ void round_loop(__amdgpu_buffer_rsrc_t wrsrc, ColView cv, int tstart, int tend, uint32_t joff,
                                           const v32f &T, float e, f2 &an, f2 &an2, f2 &ad, f2 &ad2) {
    typedef const __attribute__((address_space(4))) uint32_t *c32;
    // the list entries of the group after next arrive by scalar loads while two groups of W rows are in flight
    struct Entries {
        uint32_t o[8];  // row offsets in W
        uint2 codes;    // 8 codes
    };
    auto sload = [&](Entries &en, int t) {  // 8 entries = 32 + 8 bytes (t % 8 == 0)
        c32 po = (c32)(uint64_t)(cv.off + t);
        c32 pc = (c32)(uint64_t)(cv.code + t);
#pragma unroll
        for (int i = 0; i < 8; ++i) en.o[i] = po[i];
        en.codes = make_uint2(pc[0], pc[1]);
    };
    auto bload = [&](float(&w)[8], const Entries &en) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, en.o[i], 0));
    };
    // consume the 8 rows of a group and, row by row, put the group after next into the registers just freed:
    // every W row is requested 16 steps before its use with 16 registers in all
    auto consume_reload = [&](float(&w)[8], const uint2 &codes, const Entries &next) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t word = i < 4 ? codes.x : codes.y;
            const uint32_t ak = (word >> (8 * (i & 3) + 3)) & 0x1Fu;  // code = 8 x table row
            const f2 x = f2{T[ak], e} * w[i];
            w[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, next.o[i], 0));
            const f2 xn = {x.x, x.x}, xd = {x.y, x.y};
            an += xn;
            ad += xd;
            if (DN) an2 += xn;
            if (DD) ad2 += xd;
        }
    };
    float wA[8], wB[8];
    Entries eA, eB;
    uint2 cA, cB;
    sload(eA, tstart);
    sload(eB, tstart + 8);
    bload(wA, eA);
    cA = eA.codes;
    bload(wB, eB);
    cB = eB.codes;
    sload(eA, tstart + 16);
    sload(eB, tstart + 24);
#pragma unroll 1
    for (int t = tstart; t < tend; t += 16) {  // (the lists are padded: zero row, skipped codes)
        consume_reload(wA, cA, eA);
        cA = eA.codes;
        sload(eA, t + 32);
        consume_reload(wB, cB, eB);
        cB = eB.codes;
        sload(eB, t + 40);
    }
}

// ---- the same loop with the table read folded into the multiply --------------------------------------------
// The compiler turns T[a_k] into s_set_gpr_idx_on / v_mov_b32 (relative source) / s_set_gpr_idx_off in front of the
// packed multiply: four VALU instructions per step.  Here the lane's table column is a 32-register value pinned to
// v[64:95] (physical-register constraints on every asm statement that touches it, so the compiler itself keeps it
// there and out of everybody's way), the multiply reads v[64 + a_k] through the relative-source mode, four steps
// under one s_set_gpr_idx_on, and EXEC holds only the lanes whose row takes part, so that the denominator term is
// W itself: v_mul_f32 + two packed adds = three VALU instructions per step.  EXPERIMENTAL (MSA_BX_ASM=1, parity-tested):
// per wave it is 18 % faster, but the pinned table fragments the register file -- at the 96 registers that five
// waves per SIMD allow the allocator spills inside one of the loop versions, at 117 registers the occupancy drops
// to four waves; either way the kernel as a whole does not gain (4.98 vs 5.27 ms at best, 6.5 ms at worst).

// the lane's table column D[0..28][a_j] (stride 256 B in the LDS table); entries 29..31 are zero
__device__ __forceinline__ v32f fill_table_regs(uint32_t lds_addr) {
    v32f T;
    asm volatile(
        "ds_read_b32 v64, %1 offset:0\n\tds_read_b32 v65, %1 offset:256\n\tds_read_b32 v66, %1 offset:512\n\t"
        "ds_read_b32 v67, %1 offset:768\n\tds_read_b32 v68, %1 offset:1024\n\tds_read_b32 v69, %1 offset:1280\n\t"
        "ds_read_b32 v70, %1 offset:1536\n\tds_read_b32 v71, %1 offset:1792\n\tds_read_b32 v72, %1 offset:2048\n\t"
        "ds_read_b32 v73, %1 offset:2304\n\tds_read_b32 v74, %1 offset:2560\n\tds_read_b32 v75, %1 offset:2816\n\t"
        "ds_read_b32 v76, %1 offset:3072\n\tds_read_b32 v77, %1 offset:3328\n\tds_read_b32 v78, %1 offset:3584\n\t"
        "ds_read_b32 v79, %1 offset:3840\n\tds_read_b32 v80, %1 offset:4096\n\tds_read_b32 v81, %1 offset:4352\n\t"
        "ds_read_b32 v82, %1 offset:4608\n\tds_read_b32 v83, %1 offset:4864\n\tds_read_b32 v84, %1 offset:5120\n\t"
        "ds_read_b32 v85, %1 offset:5376\n\tds_read_b32 v86, %1 offset:5632\n\tds_read_b32 v87, %1 offset:5888\n\t"
        "ds_read_b32 v88, %1 offset:6144\n\tds_read_b32 v89, %1 offset:6400\n\tds_read_b32 v90, %1 offset:6656\n\t"
        "ds_read_b32 v91, %1 offset:6912\n\tds_read_b32 v92, %1 offset:7168\n\t"
        "v_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\tv_mov_b32 v95, 0\n\ts_waitcnt lgkmcnt(0)"
        : "={v[64:95]}"(T)
        : "v"(lds_addr)
        : "memory");
    return T;
}

template <bool DN, bool DD>
__device__ __forceinline__ void round_loop_asm(__amdgpu_buffer_rsrc_t wrsrc, ColView cv, int tstart, int tend, uint32_t joff,
                                               const v32f &T, f2 &an, f2 &an2, f2 &ad, f2 &ad2) {
    typedef const __attribute__((address_space(4))) uint32_t *c32;
    struct Entries {
        uint32_t o[8];  // row offsets in W
        uint2 codes;    // 8 codes
    };
    auto sload = [&](Entries &en, int t) {
        c32 po = (c32)(uint64_t)(cv.off + t);
        c32 pc = (c32)(uint64_t)(cv.code + t);
#pragma unroll
        for (int i = 0; i < 8; ++i) en.o[i] = po[i];
        en.codes = make_uint2(pc[0], pc[1]);
    };
    auto bload = [&](float(&w)[8], const Entries &en) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, en.o[i], 0));
    };
    auto consume_reload = [&](float(&w)[8], const uint2 &codes, const Entries &next) {
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            const uint32_t word = h ? codes.y : codes.x;
            const uint32_t k0 = (word >> 3) & 0x1Fu, k1 = (word >> 11) & 0x1Fu, k2 = (word >> 19) & 0x1Fu, k3 = word >> 27;
            float x0, x1, x2, x3;
            asm volatile(
                "s_set_gpr_idx_on %[k0], gpr_idx(SRC1)\n\tv_mul_f32 %[x0], %[w0], v64\n\t"
                "s_set_gpr_idx_idx %[k1]\n\tv_mul_f32 %[x1], %[w1], v64\n\t"
                "s_set_gpr_idx_idx %[k2]\n\tv_mul_f32 %[x2], %[w2], v64\n\t"
                "s_set_gpr_idx_idx %[k3]\n\tv_mul_f32 %[x3], %[w3], v64\n\t"
                "s_set_gpr_idx_off"
                : [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3)
                : [w0] "v"(w[h]), [w1] "v"(w[h + 1]), [w2] "v"(w[h + 2]), [w3] "v"(w[h + 3]), [k0] "s"(k0), [k1] "s"(k1),
                  [k2] "s"(k2), [k3] "s"(k3), "{v[64:95]}"(T)
                : "m0");
            const float xs[4] = {x0, x1, x2, x3};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2 xn = {xs[i], xs[i]}, xd = {w[h + i], w[h + i]};
                w[h + i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, next.o[h + i], 0));
                an += xn;
                ad += xd;
                if (DN) an2 += xn;
                if (DD) ad2 += xd;
            }
        }
    };
    // one group of 8 W rows in flight (8 steps of prefetch: more does not pay here and the registers are needed)
    float w[8];
    Entries e0, e1;
    uint2 codes;
    sload(e0, tstart);
    sload(e1, tstart + 8);
    bload(w, e0);
    codes = e0.codes;
#pragma unroll 1
    for (int t = tstart; t < tend; t += 16) {
        consume_reload(w, codes, e1);
        codes = e1.codes;
        sload(e0, t + 16);
        consume_reload(w, codes, e0);
        codes = e0.codes;
        sload(e1, t + 24);
    }
}

template <bool STAMP, bool ASM>
__device__ __forceinline__ void similarity_bx_body(const uint32_t *__restrict__ voff_, const uint16_t *__restrict__ vrow_,
                                                                       const uint8_t *__restrict__ vcode_,
                                                                       const int32_t *__restrict__ nvalid,
                                                                       const uint8_t *__restrict__ codeT_, int64_t ldk, int m_,
                                                                       int n, const int32_t *__restrict__ cols, int ncols,
                                                                       const float *__restrict__ wlow_, uint32_t wbytes,
                                                                       const float *__restrict__ wup_, int ldw_, int r0_, int compact_,
                                                                       const float *__restrict__ tab_g,
                                                                       float *__restrict__ num_out,
                                                                       float *__restrict__ den_out) {
    const gf32p wup = (gf32p)(uint64_t)wup_;
    __shared__ f2 tab[32 * 32];                    // {distance, both valid}[row code][column code], rows 28.. zero
    __shared__ float spbuf[BX_WAVES][2][64];       // per chain: the sum after every row of the round
    for (int i = threadIdx.x; i < 32 * 32; i += 64 * BX_WAVES) {
        f2 v = {0.0f, 0.0f};
        if (i < 29 * 32) v = reinterpret_cast<const f2 *>(tab_g)[i];
        tab[i] = v;
    }
    __syncthreads();
    const ldsp tabp = (ldsp)(const __attribute__((address_space(3))) void *)tab;
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    const int ci = blockIdx.x * BX_WAVES + wave;  // the wave's group of BX_Q columns in the list
    if (ci * BX_Q >= ncols) return;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wlow_, 0, (int)wbytes, 0x00027000);
    // A wave owns BX_Q columns and advances them in turn, one round each (BX_Q = 1 in production).
    ColView cv[BX_Q];
    int colid[BX_Q], mrows[BX_Q], j0[BX_Q], tbase[BX_Q];
#pragma unroll
    for (int q = 0; q < BX_Q; ++q) {
        const int col = uni(cols[ci * BX_Q + q]);  // (the list is padded with column n: an all-skipped column)
        colid[q] = col;
        cv[q].off = uniform_ptr((const __attribute__((address_space(1))) uint32_t *)(uint64_t)voff_ + (size_t)col * ldk);
        cv[q].row = uniform_ptr((const __attribute__((address_space(1))) uint16_t *)(uint64_t)vrow_ + (size_t)col * ldk);
        cv[q].code = uniform_ptr((gu8p)(uint64_t)vcode_ + (size_t)col * ldk);
        cv[q].nvalid = uni(nvalid[col]);
        cv[q].colcode = uniform_ptr((gu8p)(uint64_t)codeT_ + (size_t)col * ldk);
        cv[q].ldw = ldw_;
        cv[q].compact = compact_;
        cv[q].lastpad = (int)ldk - 1;
        cv[q].nr = 0;
        // Lanes are CONSECUTIVE rows (one coalesced 256-byte load per partner row; a row that takes no part idles
        // its lane), partners come from the compacted list (only valid rows cost a step).  In compact mode the lanes
        // are consecutive ENTRIES of the list as well (no idle lanes, the W load becomes a 64-lane gather over ~90
        // consecutive floats): "row" j then means list entry j, m the number of valid rows, every row is valid.
        mrows[q] = compact_ ? cv[q].nvalid : m_;
        j0[q] = min(r0_, max(mrows[q] - 1, 0));
        tbase[q] = 0;  // valid rows before j0
    }

    // lanes 2q / 2q+1 hold the running numerator / denominator sum of column q and the increment their last full
    // round brought (the estimate behind the second-grid decision; < 0: unknown)
    float sall = 0.0f, pinc = -1.0f;
    unsigned long long t_pro = 0, t_loop = 0, t_res = 0, n_rounds = 0, n_dual = 0, n_short = 0, t0c = 0, rt0 = 0;
    if (STAMP) {
        t0c = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
#pragma unroll
    for (int q = 0; q < BX_Q; ++q) {
        f2 s2 = {0.0f, 0.0f};
        for (int j = 0; j < j0[q]; ++j) {
            tbase[q] += compact_ ? 1 : (uni((int)cv[q].colcode[j]) != (int)BX_SKIP);
            s2 = exact_row(cv[q], wup, tabp, j, tbase[q], 3, s2);
        }
        if (lane == 2 * q) sall = s2.x;
        if (lane == 2 * q + 1) sall = s2.y;
    }
    if (STAMP) {
        const unsigned long long t1 = __builtin_readcyclecounter();
        t_pro = t1 - t0c;
        t0c = t1;
    }
    int guard = 0;  // every round commits at least one row; a round that does not would loop forever
    bool more = true;
    while (more) {
        more = false;
        if (++guard > m_ + 64) {
            sall = __uint_as_float(0x7FC00000u);  // (never reached; NaN results fail every parity test)
            break;
        }
#pragma unroll
        for (int q = 0; q < BX_Q; ++q) {
            const int nv = cv[q].nvalid, m = mrows[q];
            if (!(j0[q] < m - 1 && tbase[q] < nv)) continue;
            more = true;
            const int nrows = min(64 - ((j0[q] - r0_) & 63), m - 1 - j0[q]);
            // Which chains may leave their binade in this round?  They also accumulate on the next grid.  A wrong
            // "no" only shortens the round (resolve_chain), never the result.
            float Bl, ul;
            const bool grid = grid_of(sall, Bl, ul);
            const bool risky = !grid || pinc < 0.0f || !(sall + 1.3f * pinc * ((float)nrows * (1.0f / 64.0f)) < 2.0f * Bl);
            const uint32_t rb = (uint32_t)(__ballot(risky) >> (2 * q)) & 3u;
            const float Bn = rl(Bl, 2 * q), un = rl(ul, 2 * q), Bd = rl(Bl, 2 * q + 1), ud = rl(ul, 2 * q + 1);
            f2 an = {Bn, Bn + un}, an2 = {2.0f * Bn, 2.0f * Bn + 2.0f * un};
            f2 ad = {Bd, Bd + ud}, ad2 = {2.0f * Bd, 2.0f * Bd + 2.0f * ud};
            const uint32_t joff = 4u * (compact_ ? (uint32_t)cv[q].row[j0[q] + lane] : (uint32_t)(j0[q] + lane));
            const uint32_t cj8 =
                lane < nrows ? (uint32_t)(compact_ ? cv[q].code[j0[q] + lane] : cv[q].colcode[j0[q] + lane]) : BX_SKIP;
            const unsigned long long vmask = __ballot(cj8 != BX_SKIP);
            // partners: the valid rows behind j0 (entries at or before a lane's own row read zeros: W is lower
            // triangular here); the group of 8 that holds the first of them
            const int tstart = tbase[q] & ~7, tend = (nv + 7) & ~7;
            if (ASM) {
                // EXEC = the lanes whose row takes part (the others keep their accumulators: increment 0)
                const v32f T = fill_table_regs((uint32_t)(uintptr_t)tabp + cj8);
                if (cj8 != BX_SKIP) {
                    switch (rb) {
                        case 0: round_loop_asm<false, false>(wrsrc, cv[q], tstart, tend, joff, T, an, an2, ad, ad2); break;
                        case 1: round_loop_asm<true, false>(wrsrc, cv[q], tstart, tend, joff, T, an, an2, ad, ad2); break;
                        case 2: round_loop_asm<false, true>(wrsrc, cv[q], tstart, tend, joff, T, an, an2, ad, ad2); break;
                        default: round_loop_asm<true, true>(wrsrc, cv[q], tstart, tend, joff, T, an, an2, ad, ad2); break;
                    }
                }
            } else {
                v32f T;  // the lane's table column (zeros for a row that takes no part: column 28 of the table)
#pragma unroll
                for (int a = 0; a < 32; ++a)
                    T[a] = a < 29 ? (*reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tabp + ((a << 8) + cj8))).x : 0.0f;
                const float e = cj8 != BX_SKIP ? 1.0f : 0.0f;
                switch (rb) {
                    case 0: round_loop<false, false>(wrsrc, cv[q], tstart, tend, joff, T, e, an, an2, ad, ad2); break;
                    case 1: round_loop<true, false>(wrsrc, cv[q], tstart, tend, joff, T, e, an, an2, ad, ad2); break;
                    case 2: round_loop<false, true>(wrsrc, cv[q], tstart, tend, joff, T, e, an, an2, ad, ad2); break;
                    default: round_loop<true, true>(wrsrc, cv[q], tstart, tend, joff, T, e, an, an2, ad, ad2); break;
                }
            }
            if (STAMP) {
                const unsigned long long t1 = __builtin_readcyclecounter();
                t_loop += t1 - t0c;
                t0c = t1;
                ++n_rounds;
                n_dual += __builtin_popcount(rb);
            }
            int limit = nrows;
            {
                const Resolved r = resolve_chain(cv[q], wup, tabp, j0[q], tbase[q], vmask, 0, rl(sall, 2 * q), an.x - Bn,
                                                 an.y - (Bn + un), an2.x - 2.0f * Bn, an2.y - (2.0f * Bn + 2.0f * un), limit,
                                                 (rb & 1u) != 0);
                spbuf[wave][0][lane] = r.sp;
                limit = uni(r.limit);
            }
            {
                const Resolved r = resolve_chain(cv[q], wup, tabp, j0[q], tbase[q], vmask, 1, rl(sall, 2 * q + 1), ad.x - Bd,
                                                 ad.y - (Bd + ud), ad2.x - 2.0f * Bd, ad2.y - (2.0f * Bd + 2.0f * ud), limit,
                                                 (rb & 2u) != 0);
                spbuf[wave][1][lane] = r.sp;
                limit = uni(r.limit);
            }
            limit = max(limit, 1);
            if ((lane >> 1) == q) {
                const float snew = spbuf[wave][lane & 1][limit - 1];  // (limit >= 1: every chain commits at least one row)
                // (a short round is a poor sample of the increment per row: keep the previous estimate)
                if (!grid) pinc = -1.0f;
                else if (limit >= 16) pinc = (snew - sall) * (64.0f / (float)limit);
                sall = snew;
            }
            if (STAMP) n_short += limit < nrows;
            tbase[q] += __builtin_popcountll(vmask & ((limit >= 64 ? 0ull : (1ull << limit)) - 1ull));
            j0[q] += limit;
            if (STAMP) {
                const unsigned long long t1 = __builtin_readcyclecounter();
                t_res += t1 - t0c;
                t0c = t1;
            }
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
        atomicAdd(&g_bx_stamps[9], n_short);
        if (ci < 16384) {
            unsigned int *r = g_bx_rec + 8 * ci;
            r[0] = (unsigned)colid[0];
            r[1] = (unsigned)(t_pro >> 6);
            r[2] = (unsigned)(t_loop >> 6);
            r[3] = (unsigned)(t_res >> 6);
            r[4] = (unsigned)n_rounds;
            r[5] = (unsigned)n_short;
            r[6] = (unsigned)n_dual;
        }
    }
#pragma unroll
    for (int q = 0; q < BX_Q; ++q) {
        if (colid[q] < n) {
            const float sn = rl(sall, 2 * q), sd = rl(sall, 2 * q + 1);
            if (lane == 0) {
                num_out[colid[q]] = sn;
                den_out[colid[q]] = sd;
            }
        }
    }
}

// Two entry points over the same body: the loop with the folded table read needs 96 VGPRs to keep five waves per
// SIMD and is compiled under that limit; the compiler-built loop must not be squeezed (its indexed table would spill).
template <bool STAMP>
__global__ __launch_bounds__(64 * BX_WAVES) void similarity_bx_kernel(const uint32_t *__restrict__ voff_, const uint16_t *__restrict__ vrow_, const uint8_t *__restrict__ vcode_,
                 const int32_t *__restrict__ nvalid, const uint8_t *__restrict__ codeT_, int64_t ldk, int m_, int n,
                 const int32_t *__restrict__ cols, int ncols, const float *__restrict__ wlow_, uint32_t wbytes,
                 const float *__restrict__ wup_, int ldw_, int r0_, int compact_, const float *__restrict__ tab_g,
                 float *__restrict__ num_out, float *__restrict__ den_out) {
    similarity_bx_body<STAMP, false>(voff_, vrow_, vcode_, nvalid, codeT_, ldk, m_, n, cols, ncols, wlow_, wbytes, wup_, ldw_, r0_, compact_, tab_g, num_out, den_out);
}
template <bool STAMP>
__global__ __launch_bounds__(64 * BX_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void similarity_bx_asm_kernel(
    const uint32_t *__restrict__ voff_, const uint16_t *__restrict__ vrow_, const uint8_t *__restrict__ vcode_,
                 const int32_t *__restrict__ nvalid, const uint8_t *__restrict__ codeT_, int64_t ldk, int m_, int n,
                 const int32_t *__restrict__ cols, int ncols, const float *__restrict__ wlow_, uint32_t wbytes,
                 const float *__restrict__ wup_, int ldw_, int r0_, int compact_, const float *__restrict__ tab_g,
                 float *__restrict__ num_out, float *__restrict__ den_out) {
    similarity_bx_body<STAMP, true>(voff_, vrow_, vcode_, nvalid, codeT_, ldk, m_, n, cols, ncols, wlow_, wbytes, wup_, ldw_, r0_, compact_, tab_g, num_out, den_out);
}

// ---- per-lane grids ------------------------------------------------------------------------------------------
// The kernel above fixes ONE grid per chain and round (the binade of the sum at the round's first row) and pays for
// every binade crossing: a second pair of accumulators for predicted crossings, a round cut short for the others
// (the early rounds, where the sum doubles every few rows, are walked three to four times).  Nothing in the scheme
// needs the lanes of a round to agree on a grid: a lane's accumulators only have to start on the grid of the sum
// IN FRONT OF ITS OWN ROW.  That sum is not known before the round, but it is predictable to a fraction of a row:
//     increment of row j  ~  c * (valid rows behind j) * G[a_j],   G[a] = sum_b h_b D[b][a]
// (h = the column's residue frequencies; G = 1 for the denominator; c = the ratio measured on the previous round,
// on the ordered first row for the first round).  Every lane starts on the grid of its predicted sum, the round
// loop has a single version (two packed adds per step, no second grid), every round covers its 64 rows, and the
// stitching commits segment by segment: lanes whose grid is the binade the sum really is in are scanned as before;
// a row whose sum leaves the binade (the crossing row, ~9 per chain at m = 2000) or whose prediction was wrong
// (measured: none on the synthetic alignments) is evaluated in the reference's order.  A wrong prediction costs
// one ordered row, never exactness: every commit is checked against the true sum.
constexpr int LG_R0 = 1;  // rows evaluated in order before the first round (at least up to the first valid row)

__device__ __forceinline__ float unif(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}

// scan_rows over an arbitrary set of lanes
__device__ __forceinline__ float scan_lanes(float s, float top, float ie, float io, unsigned long long segm, int lane, int &cross) {
    const bool in = ((segm >> lane) & 1ull) != 0ull;
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

// One chain at the end of a round: s before the round's first row; per lane the grid it accumulated on (Bl; 0 =
// plain sums from zero) and its increments for an even / odd sum.  Commits every row of vmask; returns the sum
// behind the round and the number of rows that went through the ordered path.
struct ResolvedLg {
    float s;
    int ordered;
    unsigned long long t_ordered;  // STAMP: cycles spent in the ordered rows
};
template <bool STAMP>
__device__ __forceinline__ ResolvedLg resolve_lg(ColView cv, gf32p wup, ldsp tab, int j0, int tbase, unsigned long long vall,
                                              unsigned long long vmask, int kind, float s, float Bl, float ie, float io) {
    const int lane = threadIdx.x & 63;
    // (row j0 + x: its partners start at entry tbase + (valid rows of the round up to and including x))
    auto tfirst = [&](int x) { return tbase + __builtin_popcountll(vall & ((2ull << x) - 1ull)); };
    unsigned long long t_ordered = 0;
    auto ordered_row = [&](int x, float from) {
        unsigned long long t0 = 0;
        if (STAMP) t0 = __builtin_readcyclecounter();
        const f2 r = exact_row(cv, wup, tab, j0 + x, tfirst(x), kind ? 2 : 1, f2{from, from});
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
    return ResolvedLg{s, ordered, t_ordered};
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

// W rows by global_load_dword with a 64-bit per-lane address instead of buffer_load_dword (SGPR row offset + per-lane
// offset, range-checked): tools/ubench_wstream.hip measures 5.3 instead of 8.6 CU-cycles per 256-byte wave-load for
// the same rows (30 against 18 TB/s chip-wide) -- the buffer path costs the texture addresser more per instruction.
// One more VALU instruction per step (the address).
// MODE 3: (production) global loads with the row address in an SGPR pair and the lane offset in a VGPR (s_add_u32 /
//         s_addc_u32: two SALU, no VALU); hand-issued, so the loop counts VMCNT itself: 16 loads are in flight at
//         every step and nothing else of the loop is a vector-memory instruction
//      0: global loads, 64-bit per-lane address left to the compiler (v_lshl_add_u64: one VALU + one SALU per step;
//         diagnostics, MSA_LG_DBG & 128: 3.38 instead of 3.13 ms at C3)
//      1: buffer loads (diagnostics, MSA_LG_DBG & 64: 4.03 ms)
//      2: no W reloads at all (diagnostics, MSA_LG_DBG & 1: what the W stream costs; results are wrong)
template <int MODE>
__device__ __forceinline__ void round_loop_lds(__amdgpu_buffer_rsrc_t wrsrc, const float *wlow_g,
                                               const __attribute__((address_space(1))) uint32_t *off,
                                               const __attribute__((address_space(1))) uint16_t *trow, int tstart, int tend,
                                               uint32_t joff, uint32_t base, f2 &an, f2 &ad) {
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
    const __attribute__((address_space(1))) char *wlane = (const __attribute__((address_space(1))) char *)(uint64_t)wlow_g + joff;
    const uint64_t wuni = (uint64_t)uniform_ptr((const __attribute__((address_space(1))) char *)(uint64_t)wlow_g);
    auto wrow = [&](float &dst, uint32_t o) {
        if (MODE == 1) dst = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, o, 0));
        else if (MODE == 3) {
            const uint64_t row = wuni + o;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(joff), "s"(row) : "memory");
        } else dst = *reinterpret_cast<const __attribute__((address_space(1))) float *>(wlane + o);
    };
    auto bload = [&](float(&w)[16], const LgEntries &en) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wrow(w[i], en.o[i]);
    };
    // 16 steps; every W row is requested 16 steps before its use
    auto consume_reload = [&](float(&w)[16], const LgD &D, const LgEntries &next) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 3) asm volatile("s_waitcnt vmcnt(15)" : "+v"(w[i])::"memory");  // the oldest of the 16 loads in flight
            const float wi = w[i];
            const float x = wi * D.d[i];
            const f2 xn = {x, x}, xd = {wi, wi};
            if (MODE != 2) wrow(w[i], next.o[i]);
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
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int LG_WAVES_MAX = 8;  // waves per workgroup (they share the distance table; the launcher picks 4 or 8)

// LDST: the lane's table column lives in LDS (round_loop_lds; `nr` rows per wave behind the static arrays, within
// the first 64 KB: M0 holds 16 bits) instead of 32 registers (round_loop).
template <bool STAMP, bool LDST>
__device__ __forceinline__ void similarity_lg_body(
    const uint32_t *__restrict__ voff_, const uint16_t *__restrict__ vrow_, const uint8_t *__restrict__ vcode_,
    const uint16_t *__restrict__ vtrow_, int nr, const int32_t *__restrict__ nvalid, const uint8_t *__restrict__ codeT_,
    int64_t ldk, int m, int n, const int32_t *__restrict__ cols, int ncols, const float *__restrict__ wlow_,
    uint32_t wbytes, const float *__restrict__ wup_, int ldw_, int r0_, const float *__restrict__ tab_g,
    float *__restrict__ num_out, float *__restrict__ den_out) {
    const gf32p wup = (gf32p)(uint64_t)wup_;
    __shared__ f2 tab[32 * 32];                  // {distance, both valid}[row code][column code], rows 28.. zero
    __shared__ uint32_t hist[LG_WAVES_MAX][32];  // residue counts of the wave's column
    __shared__ float gtab[LG_WAVES_MAX][32];     // G[a] = mean over the column's valid rows of D[.][a]
    extern __shared__ float ltab[];              // LDST: [wave][nr][64 lanes]
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        f2 v = {0.0f, 0.0f};
        if (i < 29 * 32) v = reinterpret_cast<const f2 *>(tab_g)[i];
        tab[i] = v;
    }
    if (threadIdx.x < LG_WAVES_MAX * 32) (&hist[0][0])[threadIdx.x] = 0u;
    __syncthreads();
    const ldsp tabp = (ldsp)(const __attribute__((address_space(3))) void *)tab;
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    const int ci = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (ci >= ncols) return;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wlow_, 0, (int)wbytes, 0x00027000);
    const int col = uni(cols[ci]);
    ColView cv;
    cv.off = uniform_ptr((const __attribute__((address_space(1))) uint32_t *)(uint64_t)voff_ + (size_t)col * ldk);
    cv.row = uniform_ptr((const __attribute__((address_space(1))) uint16_t *)(uint64_t)vrow_ + (size_t)col * ldk);
    cv.code = uniform_ptr((gu8p)(uint64_t)vcode_ + (size_t)col * ldk);
    cv.nvalid = uni(nvalid[col]);
    cv.colcode = uniform_ptr((gu8p)(uint64_t)codeT_ + (size_t)col * ldk);
    cv.ldw = ldw_;
    cv.compact = 0;
    cv.lastpad = (int)ldk - 1;
    cv.nr = 0;
    const __attribute__((address_space(1))) uint16_t *vtrow =
        uniform_ptr((const __attribute__((address_space(1))) uint16_t *)(uint64_t)vtrow_ + (size_t)col * ldk);
    const int nv = cv.nvalid;
    unsigned long long t_pro = 0, t_loop = 0, t_res = 0, n_rounds = 0, n_ordered = 0, t_ord = 0, t0c = 0, rt0 = 0;
    if (STAMP) {
        t0c = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    __builtin_amdgcn_s_setprio(3);  // (the ordered first row: as the stitching below)
    // the column's residue frequencies -> G
    for (int t = lane; t < nv; t += 64) atomicAdd(&hist[wave][cv.code[t] >> 3], 1u);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (lane < 32) {
        float g = 0.0f;
        for (int b = 0; b < 29; ++b) g += (float)hist[wave][b] * tab[b * 32 + lane].x;
        gtab[wave][lane] = nv > 0 ? g / (float)nv : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    // the first rows in the reference's order: at least up to the first row that takes part
    f2 s2 = {0.0f, 0.0f};
    float qn0 = 0.0f, qd0 = 0.0f;
    int jstart = 0, tb = 0;
    {
        bool seen = false;
        while (jstart < m - 1 && tb < nv && (jstart < (r0_ & 0xFFFF) || !seen)) {
            const uint32_t cj = (uint32_t)uni((int)cv.colcode[jstart]);
            if (cj != BX_SKIP) {
                ++tb;
                seen = true;
                const float rem = (float)(nv - tb);
                qd0 += rem;
                qn0 += rem * gtab[wave][cj >> 3];
                s2 = exact_row(cv, wup, tabp, jstart, tb, 3, s2);
            }
            ++jstart;
        }
    }
    float sn = unif(s2.x), sd = unif(s2.y);
    qn0 = unif(qn0);
    qd0 = unif(qd0);
    // increment per unit of the estimate (any positive value is correct; a poor one costs ordered rows)
    float cn = unif((sn > 0.0f && qn0 > 0.0f) ? sn / qn0 : 0.8f);
    float cd = unif((sd > 0.0f && qd0 > 0.0f) ? sd / qd0 : 0.8f);
    if (STAMP) {
        const unsigned long long t1 = __builtin_readcyclecounter();
        t_pro = t1 - t0c;
        t0c = t1;
    }
    __builtin_amdgcn_s_setprio(0);
    int j0 = jstart & ~63;
    int first = jstart - j0;  // the round's lanes before it were evaluated above
    int tbase;                // valid rows before j0
    {
        const int r = j0 + lane;
        const bool v = r < jstart && cv.colcode[r] != BX_SKIP;
        tbase = tb - __builtin_popcountll(__ballot(v));
    }
    for (; j0 < m - 1 && tbase < nv; j0 += 64) {
        const int nrows = min(64, m - 1 - j0);
        const uint32_t craw = lane < nrows ? (uint32_t)cv.colcode[j0 + lane] : BX_SKIP;
        const unsigned long long vall = __ballot(craw != BX_SKIP);
        const uint32_t cj8 = lane >= first ? craw : BX_SKIP;
        const unsigned long long vmask = __ballot(cj8 != BX_SKIP);
        first = 0;
        // predicted sum in front of every row -> the lane's grid
        const int behind = nv - (tbase + __builtin_popcountll(vall & ((2ull << lane) - 1ull)));
        const bool takes = cj8 != BX_SKIP;
        const float qd = takes ? (float)behind : 0.0f;
        const float qn = takes ? (float)behind * gtab[wave][cj8 >> 3] : 0.0f;
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
        f2 an = {Bn, Bn + Bn * 0x1p-23f}, ad = {Bd, Bd + Bd * 0x1p-23f}, an2 = {0.0f, 0.0f}, ad2 = {0.0f, 0.0f};
        const uint32_t joff = 4u * (uint32_t)(j0 + lane);
        if (LDST) {
            // the lane's table column into the wave's [row][lane] table (zeros for a row that takes no part: column 28)
            float *lt = ltab + (size_t)wave * nr * 64 + lane;
            for (int a = 0; a < nr; ++a)
                lt[a * 64] = (*reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tabp + ((a << 8) + cj8))).x;
            const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)(ltab + (size_t)wave * nr * 64);
            // (the diagnostic loop versions exist in the stamped kernel only)
            if (STAMP && (r0_ & 0x10000))
                round_loop_lds<2>(wrsrc, wlow_, cv.off, vtrow, tbase & ~15, nv, joff, (uint32_t)uni((int)base), an, ad);
            else if (STAMP && (r0_ & 0x40000))
                round_loop_lds<1>(wrsrc, wlow_, cv.off, vtrow, tbase & ~15, nv, joff, (uint32_t)uni((int)base), an, ad);
            else if (STAMP && (r0_ & 0x80000))
                round_loop_lds<0>(wrsrc, wlow_, cv.off, vtrow, tbase & ~15, nv, joff, (uint32_t)uni((int)base), an, ad);
            else
                round_loop_lds<3>(wrsrc, wlow_, cv.off, vtrow, tbase & ~15, nv, joff, (uint32_t)uni((int)base), an, ad);
        } else {
            const int tstart = tbase & ~7, tend = (nv + 7) & ~7;
            v32f T;  // the lane's table column
#pragma unroll
            for (int a = 0; a < 32; ++a)
                T[a] = a < 29 ? (*reinterpret_cast<const __attribute__((address_space(3))) f2 *>(tabp + ((a << 8) + cj8))).x : 0.0f;
            const float e = takes ? 1.0f : 0.0f;
            round_loop<false, false>(wrsrc, cv, tstart, tend, joff, T, e, an, an2, ad, ad2);
        }
        if (STAMP) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_loop += t1 - t0c;
            t0c = t1;
            ++n_rounds;
        }
        // The stitching is a latency-bound instruction stream that competes for issue slots with the round loops of
        // the SIMD's other waves (which are bound by the W stream, not by issue): it runs at the highest wave priority.
        __builtin_amdgcn_s_setprio(3);
        // (a lane whose row takes no part: the LDS loop added W to its denominator accumulators -- ignored)
        const ResolvedLg rn = resolve_lg<STAMP>(cv, wup, tabp, j0, tbase, vall, vmask, 0, sn, Bn, an.x - Bn, an.y - (Bn + Bn * 0x1p-23f));
        const ResolvedLg rd = resolve_lg<STAMP>(cv, wup, tabp, j0, tbase, vall, vmask, 1, sd, Bd, takes ? ad.x - Bd : 0.0f,
                                         takes ? ad.y - (Bd + Bd * 0x1p-23f) : 0.0f);
        const float sn1 = unif(rn.s), sd1 = unif(rd.s);
        if (sn1 > sn && Qn > 0.0f) cn = unif((sn1 - sn) / Qn);
        if (sd1 > sd && Qd > 0.0f) cd = unif((sd1 - sd) / Qd);
        sn = sn1;
        sd = sd1;
        tbase += __builtin_popcountll(vall);
        __builtin_amdgcn_s_setprio(0);
        if (STAMP) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_res += t1 - t0c;
            t0c = t1;
            n_ordered += (unsigned)(rn.ordered + rd.ordered);
            t_ord += rn.t_ordered + rd.t_ordered;
        }
    }
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
    if (col < n && lane == 0) {
        num_out[col] = sn;
        den_out[col] = sd;
    }
}

// Two entry points over the same body: with the table in LDS the kernel is compiled for six waves per SIMD (80
// registers; the LDS allocation admits 24 waves per CU for a 20-letter alphabet); the register-table version must
// not be squeezed (its indexed table would go to scratch memory).
#define LG_PARAMS                                                                                                      \
    const uint32_t *__restrict__ voff_, const uint16_t *__restrict__ vrow_, const uint8_t *__restrict__ vcode_,      \
        const uint16_t *__restrict__ vtrow_, int nr, const int32_t *__restrict__ nvalid, const uint8_t *__restrict__ codeT_, \
        int64_t ldk, int m, int n, const int32_t *__restrict__ cols, int ncols, const float *__restrict__ wlow_,    \
        uint32_t wbytes, const float *__restrict__ wup_, int ldw_, int r0_, const float *__restrict__ tab_g,         \
        float *__restrict__ num_out, float *__restrict__ den_out, const int *__restrict__ gate
#define LG_ARGS voff_, vrow_, vcode_, vtrow_, nr, nvalid, codeT_, ldk, m, n, cols, ncols, wlow_, wbytes, wup_, ldw_, r0_, tab_g, num_out, den_out
template <bool STAMP>
__global__ __launch_bounds__(64 * LG_WAVES_MAX) __attribute__((amdgpu_waves_per_eu(5, 5))) void similarity_lg_kernel(LG_PARAMS) {
    // (automated1 enqueues this kernel before the host knows which method the identity statistics select: the
    // kernel that computes them raises the gate when the similarity values will not be used)
    if (gate && *gate) return;
    similarity_lg_body<STAMP, true>(LG_ARGS);
}
template <bool STAMP>
__global__ __launch_bounds__(64 * LG_WAVES_MAX) void similarity_lg_regs_kernel(LG_PARAMS) {
    if (gate && *gate) return;
    similarity_lg_body<STAMP, false>(LG_ARGS);
}
#undef LG_PARAMS
#undef LG_ARGS

// ---- two columns per wave ------------------------------------------------------------------------------------
// The kernel above is bound by the vector-memory pipeline: one 256-byte buffer load of a W row per 64 terms costs
// the CU ~8.5 cycles whatever the occupancy (tools/ubench_wstream.hip: 18 TB/s chip-wide for this pattern, all
// L2 hits), and 2.2e8 of them are 3.6 of its 4.0 ms at 2000 x 10000.  Here a wave owns TWO columns (neighbours in
// the order by valid rows) and walks the UNION of their valid rows once: every W row loaded serves both columns.
//   * the lane's table lookup D[a_k][a_j] comes from ONE table per workgroup instead of one per wave and round:
//     every entry replicated 32 times ([a_k][a_j][copy], lane l reads copy l % 32: no bank conflict), address =
//     (a_k * nr * 128, from the list, SGPR) + (a_j * 128 + 4 (l % 32), per lane and round): one v_add + ds_read_b32;
//   * a row that takes part in only one of the two columns: the other column reads the table's zero row
//     (numerator term 0) and adds W * 0 to its denominator -- v_pk_fma_f32 with the 1/0 flag of the list as the
//     scalar operand; W * 1 is exact, so fl(W * 1 + s) is the reference's fl(s + W);
//   * no per-wave LDS, 5 waves per SIMD, all the waves of 10 000 columns resident at once.
struct PairView {  // the union of two columns' valid rows, in row order (lg2_union_kernel)
    const __attribute__((address_space(1))) uint32_t *off;  // W row offset; padding: the zero row
    const __attribute__((address_space(1))) uint32_t *tt;   // table-row offsets a_k * nr * 128 of the two columns, 16 bits each; padding: zero rows
    const __attribute__((address_space(1))) float *ee;      // {takes part in column 1, in column 2} as 1.0 / 0.0
    int n;
};

__device__ __forceinline__ void round_loop_q2(__amdgpu_buffer_rsrc_t wrsrc, PairView pv, int tstart, int tend, uint32_t joff,
                                              ldsp rtab, uint32_t vb1, uint32_t vb2, f2 &an1, f2 &ad1, f2 &an2, f2 &ad2) {
    typedef const __attribute__((address_space(4))) uint32_t *c32;
    typedef const __attribute__((address_space(4))) float *cf32;
    auto lo = [&](uint32_t(&o)[8], int t) {
        c32 p = (c32)(uint64_t)(pv.off + t);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = p[i];
    };
    auto lt = [&](uint32_t(&x)[8], int t) {
        c32 p = (c32)(uint64_t)(pv.tt + t);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = p[i];
    };
    auto le = [&](float(&e)[16], int t) {
        cf32 p = (cf32)(uint64_t)(pv.ee + 2 * t);
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = p[i];
    };
    auto bload = [&](float(&w)[8], const uint32_t(&o)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, o[i], 0));
    };
    auto dissue = [&](float(&d1)[8], float(&d2)[8], const uint32_t(&x)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            d1[i] = *reinterpret_cast<const __attribute__((address_space(3))) float *>(rtab + (vb1 + (x[i] & 0xFFFFu)));
            d2[i] = *reinterpret_cast<const __attribute__((address_space(3))) float *>(rtab + (vb2 + (x[i] >> 16)));
        }
    };
    auto consume_reload = [&](float(&w)[8], const float(&d1)[8], const float(&d2)[8], const float(&e)[16], const uint32_t(&next)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float wi = w[i];
            const float x1 = wi * d1[i], x2 = wi * d2[i];
            w[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, joff, next[i], 0));
            an1 += f2{x1, x1};
            an2 += f2{x2, x2};
            ad1 = __builtin_elementwise_fma(f2{wi, wi}, f2{e[2 * i], e[2 * i]}, ad1);
            ad2 = __builtin_elementwise_fma(f2{wi, wi}, f2{e[2 * i + 1], e[2 * i + 1]}, ad2);
        }
    };
    // Groups of 8 union entries, A and B in turn.  In front of a group: its table values and flags and the W row
    // offsets of the group after next have arrived (one lgkmcnt(0), everything outstanding is a group old); then
    // the table reads of the next group and the scalar loads of later groups are issued, then 8 steps of VALU work
    // whose W rows were requested two groups ago.
    uint32_t oA[8], oB[8], tA[8], tB[8];
    float eA[16], eB[16], wA[8], wB[8], d1A[8], d2A[8], d1B[8], d2B[8];
    lo(oA, tstart);
    lo(oB, tstart + 8);
    lt(tA, tstart);
    le(eA, tstart);
    bload(wA, oA);
    bload(wB, oB);
    dissue(d1A, d2A, tA);
    lt(tB, tstart + 8);
    lo(oA, tstart + 16);
#pragma unroll 1
    for (int t = tstart; t < tend; t += 16) {  // (the lists are padded: zero row of W, zero rows of the table, flags 0)
        dissue(d1B, d2B, tB);
        lt(tA, t + 16);
        le(eB, t + 8);
        lo(oB, t + 24);
        __builtin_amdgcn_sched_barrier(0);  // (the scalar loads must not sink towards their use: the next wait would stall on them)
        consume_reload(wA, d1A, d2A, eA, oA);
        dissue(d1A, d2A, tA);
        lt(tB, t + 24);
        le(eA, t + 16);
        lo(oA, t + 32);
        __builtin_amdgcn_sched_barrier(0);
        consume_reload(wB, d1B, d2B, eB, oB);
    }
}

constexpr int LG2_WAVES = 12;  // waves per workgroup (a multiple of 4: a workgroup fills the SIMDs evenly): two workgroups share a CU (the 56 KB table each), 6 waves per SIMD

template <bool STAMP>
__global__ __launch_bounds__(64 * LG2_WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) void similarity_lg2_kernel(
    const uint32_t *__restrict__ voff_, const uint16_t *__restrict__ vrow_, const uint8_t *__restrict__ vcode_,
    const int32_t *__restrict__ nvalid, const uint8_t *__restrict__ codeT_, int64_t ldk, int m, int n,
    const int32_t *__restrict__ cols, int npairs, const uint32_t *__restrict__ uoff_, const uint32_t *__restrict__ utt_,
    const float *__restrict__ uee_, const int32_t *__restrict__ nunion, int nr, const float *__restrict__ wlow_,
    uint32_t wbytes, const float *__restrict__ wup_, int ldw_, int r0_, const float *__restrict__ tab_g,
    float *__restrict__ num_out, float *__restrict__ den_out) {
    const gf32p wup = (gf32p)(uint64_t)wup_;
    __shared__ uint32_t hist[LG2_WAVES][2][32];  // residue counts of the wave's columns
    __shared__ float gtab[LG2_WAVES][2][32];     // G[a] = mean over the column's valid rows of D[.][a]
    extern __shared__ float rtab_[];             // [nr][nr][32]: D[a][b] (zero where a or b is the row behind the alphabet), 32 copies
    for (int i = threadIdx.x; i < nr * nr * 32; i += blockDim.x) {
        const int a = (i >> 5) / nr, b = (i >> 5) % nr;
        rtab_[i] = (a < nr - 1 && b < nr - 1) ? tab_g[(a * 32 + b) * 2] : 0.0f;
    }
    for (int i = threadIdx.x; i < LG2_WAVES * 2 * 32; i += blockDim.x) (&hist[0][0][0])[i] = 0u;
    __syncthreads();
    const ldsp rtab = (ldsp)(const __attribute__((address_space(3))) void *)rtab_;
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    const int pi = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (pi >= npairs) return;
    // Every wave is resident from the start and the kernel ends with the heaviest pair: the host deals the pairs so
    // that wave 0 of a workgroup holds the heaviest and the last wave the lightest -- the heavy ones issue first.
    if (!(r0_ & 0x20000)) {
        const int nw = (int)(blockDim.x >> 6);
        const int band = nw > 1 ? (wave * 4) / nw : 0;  // 0 (heaviest quarter) .. 3
        if (band == 0) __builtin_amdgcn_s_setprio(3);
        else if (band == 1) __builtin_amdgcn_s_setprio(2);
        else if (band == 2) __builtin_amdgcn_s_setprio(1);
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wlow_, 0, (int)wbytes, 0x00027000);
    PairView pv;
    pv.off = uniform_ptr((const __attribute__((address_space(1))) uint32_t *)(uint64_t)uoff_ + (size_t)pi * ldk);
    pv.tt = uniform_ptr((const __attribute__((address_space(1))) uint32_t *)(uint64_t)utt_ + (size_t)pi * ldk);
    pv.ee = uniform_ptr((const __attribute__((address_space(1))) float *)(uint64_t)uee_ + (size_t)pi * ldk * 2);
    pv.n = uni(nunion[pi]);
    ColView cv[2];
    int col[2], nv[2], jstart[2], tb[2];
    float sn[2], sd[2], cn[2], cd[2];
    unsigned long long t_pro = 0, t_loop = 0, t_res = 0, n_rounds = 0, n_ordered = 0, t_ord = 0, t0c = 0, rt0 = 0;
    if (STAMP) {
        t0c = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        col[q] = uni(cols[2 * pi + q]);  // (the list is padded with column n: an all-skipped column)
        const int c = col[q];
        cv[q].off = uniform_ptr((const __attribute__((address_space(1))) uint32_t *)(uint64_t)voff_ + (size_t)c * ldk);
        cv[q].row = uniform_ptr((const __attribute__((address_space(1))) uint16_t *)(uint64_t)vrow_ + (size_t)c * ldk);
        cv[q].code = uniform_ptr((gu8p)(uint64_t)vcode_ + (size_t)c * ldk);
        cv[q].nvalid = uni(nvalid[c]);
        cv[q].colcode = uniform_ptr((gu8p)(uint64_t)codeT_ + (size_t)c * ldk);
        cv[q].ldw = ldw_;
        cv[q].compact = 0;
        cv[q].lastpad = (int)ldk - 1;
        cv[q].nr = nr;
        nv[q] = cv[q].nvalid;
        // the column's residue frequencies -> G
        for (int t = lane; t < nv[q]; t += 64) atomicAdd(&hist[wave][q][cv[q].code[t] >> 3], 1u);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (lane < 32) {
            float g = 0.0f;
            if (lane < nr - 1)
                for (int b = 0; b < nr - 1; ++b) g += (float)hist[wave][q][b] * rtab_[(b * nr + lane) * 32];
            gtab[wave][q][lane] = nv[q] > 0 ? g / (float)nv[q] : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // the first rows in the reference's order: at least up to the first row that takes part
        f2 s2 = {0.0f, 0.0f};
        float qn0 = 0.0f, qd0 = 0.0f;
        jstart[q] = 0;
        tb[q] = 0;
        bool seen = false;
        while (jstart[q] < m - 1 && tb[q] < nv[q] && (jstart[q] < (r0_ & 0xFFFF) || !seen)) {
            const uint32_t cj = (uint32_t)uni((int)cv[q].colcode[jstart[q]]);
            if (cj != BX_SKIP) {
                ++tb[q];
                seen = true;
                const float rem = (float)(nv[q] - tb[q]);
                qd0 += rem;
                qn0 += rem * gtab[wave][q][cj >> 3];
                s2 = exact_row(cv[q], wup, rtab, jstart[q], tb[q], 3, s2);
            }
            ++jstart[q];
        }
        if (tb[q] >= nv[q]) jstart[q] = m;  // nothing behind the ordered rows: the column is done
        sn[q] = unif(s2.x);
        sd[q] = unif(s2.y);
        qn0 = unif(qn0);
        qd0 = unif(qd0);
        cn[q] = unif((sn[q] > 0.0f && qn0 > 0.0f) ? sn[q] / qn0 : 0.8f);
        cd[q] = unif((sd[q] > 0.0f && qd0 > 0.0f) ? sd[q] / qd0 : 0.8f);
    }
    if (STAMP) {
        const unsigned long long t1 = __builtin_readcyclecounter();
        t_pro = t1 - t0c;
        t0c = t1;
    }
    int j0 = min(jstart[0], jstart[1]) & ~63;
    // valid rows of either column / of their union before j0
    int tbase[2] = {0, 0}, tbaseU = 0;
    for (int r = lane; r < j0; r += 64) {  // (j0 > 0 only when both columns begin with 64 or more rows that take no part)
        const bool v0 = cv[0].colcode[r] != BX_SKIP, v1 = cv[1].colcode[r] != BX_SKIP;
        tbase[0] += __builtin_popcountll(__ballot(v0));
        tbase[1] += __builtin_popcountll(__ballot(v1));
        tbaseU += __builtin_popcountll(__ballot(v0 || v1));
    }
    tbase[0] = uni(tbase[0]);
    tbase[1] = uni(tbase[1]);
    tbaseU = uni(tbaseU);
    for (; j0 < m - 1 && (tbase[0] < nv[0] || tbase[1] < nv[1]); j0 += 64) {
        const int nrows = min(64, m - 1 - j0);
        unsigned long long vall[2], vmask[2];
        uint32_t vb[2];
        float Bn[2], Bd[2], Qn[2], Qd[2];
        bool takes[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t craw = lane < nrows ? (uint32_t)cv[q].colcode[j0 + lane] : BX_SKIP;
            vall[q] = __ballot(craw != BX_SKIP);
            const uint32_t cj8 = j0 + lane >= jstart[q] ? craw : BX_SKIP;  // (rows before jstart were evaluated in order)
            vmask[q] = __ballot(cj8 != BX_SKIP);
            takes[q] = cj8 != BX_SKIP;
            const uint32_t aj = takes[q] ? cj8 >> 3 : (uint32_t)nr - 1u;
            vb[q] = (aj << 7) + ((uint32_t)(lane & 31) << 2);
            // predicted sum in front of every row -> the lane's grid
            const int behind = nv[q] - (tbase[q] + __builtin_popcountll(vall[q] & ((2ull << lane) - 1ull)));
            const float qd = takes[q] ? (float)behind : 0.0f;
            const float qn = takes[q] ? (float)behind * gtab[wave][q][cj8 >> 3] : 0.0f;
            const float Pd = wave_prefix(qd), Pn = wave_prefix(qn);
            float un, ud;
            grid_of(sn[q] + cn[q] * (Pn - qn), Bn[q], un);
            grid_of(sd[q] + cd[q] * (Pd - qd), Bd[q], ud);
            Qn[q] = rl(Pn, 63);
            Qd[q] = rl(Pd, 63);
        }
        // (the ulp of a grid is B * 2^-23, exact: grid_of only accepts exponents >= 30)
        f2 an0 = {Bn[0], Bn[0] + Bn[0] * 0x1p-23f}, ad0 = {Bd[0], Bd[0] + Bd[0] * 0x1p-23f};
        f2 an1 = {Bn[1], Bn[1] + Bn[1] * 0x1p-23f}, ad1 = {Bd[1], Bd[1] + Bd[1] * 0x1p-23f};
        round_loop_q2(wrsrc, pv, tbaseU & ~7, pv.n, 4u * (uint32_t)(j0 + lane), rtab, vb[0], vb[1], an0, ad0, an1, ad1);
        if (STAMP) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_loop += t1 - t0c;
            t0c = t1;
            ++n_rounds;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f2 an = q ? an1 : an0, ad = q ? ad1 : ad0;
            // (a lane whose row takes no part added W x flag to its denominator accumulators: ignored)
            const ResolvedLg rn = resolve_lg<STAMP>(cv[q], wup, rtab, j0, tbase[q], vall[q], vmask[q], 0, sn[q], Bn[q], an.x - Bn[q],
                                                    an.y - (Bn[q] + Bn[q] * 0x1p-23f));
            const ResolvedLg rd = resolve_lg<STAMP>(cv[q], wup, rtab, j0, tbase[q], vall[q], vmask[q], 1, sd[q], Bd[q],
                                                    takes[q] ? ad.x - Bd[q] : 0.0f, takes[q] ? ad.y - (Bd[q] + Bd[q] * 0x1p-23f) : 0.0f);
            const float sn1 = unif(rn.s), sd1 = unif(rd.s);
            if (sn1 > sn[q] && Qn[q] > 0.0f) cn[q] = unif((sn1 - sn[q]) / Qn[q]);
            if (sd1 > sd[q] && Qd[q] > 0.0f) cd[q] = unif((sd1 - sd[q]) / Qd[q]);
            sn[q] = sn1;
            sd[q] = sd1;
            tbase[q] += __builtin_popcountll(vall[q]);
            if (STAMP) {
                n_ordered += (unsigned)(rn.ordered + rd.ordered);
                t_ord += rn.t_ordered + rd.t_ordered;
            }
        }
        tbaseU += __builtin_popcountll(vall[0] | vall[1]);
        if (STAMP) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            t_res += t1 - t0c;
            t0c = t1;
        }
    }
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
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (col[q] < n) {
                num_out[col[q]] = sn[q];
                den_out[col[q]] = sd[q];
            }
    }
}

// The union lists of the column pairs (one wave per pair): for every row that takes part in either column, in row
// order: W row offset, the two table-row offsets (row behind the alphabet = zero row where the row takes no part),
// the two 1/0 flags; padded behind the last entry like the single-column lists.
__global__ __launch_bounds__(256) void lg2_union_kernel(const uint8_t *__restrict__ codeT, int64_t ldk, int m, const int32_t *__restrict__ cols,
                                                        int npairs, uint32_t ldw4, int nr, uint32_t *__restrict__ uoff,
                                                        uint32_t *__restrict__ utt, float *__restrict__ uee, int32_t *__restrict__ nunion) {
    const int lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= npairs) return;
    const uint8_t *s0 = codeT + (size_t)cols[2 * pi] * ldk, *s1 = codeT + (size_t)cols[2 * pi + 1] * ldk;
    uint32_t *po = uoff + (size_t)pi * ldk, *pt = utt + (size_t)pi * ldk;
    float2 *pe = reinterpret_cast<float2 *>(uee) + (size_t)pi * ldk;
    const uint32_t stride = (uint32_t)nr * 128u, zrow = (uint32_t)(nr - 1) * stride;
    int count = 0;
    for (int kb = 0; kb < m; kb += 64) {
        const int k = kb + lane;
        const uint32_t c0 = k < m ? s0[k] : BX_SKIP, c1 = k < m ? s1[k] : BX_SKIP;
        const bool v0 = c0 != BX_SKIP, v1 = c1 != BX_SKIP;
        const unsigned long long mask = __ballot(v0 || v1);
        if (v0 || v1) {
            const int pos = count + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            po[pos] = (uint32_t)k * ldw4;
            pt[pos] = (v0 ? (c0 >> 3) * stride : zrow) | ((v1 ? (c1 >> 3) * stride : zrow) << 16);
            pe[pos] = make_float2(v0 ? 1.0f : 0.0f, v1 ? 1.0f : 0.0f);
        }
        count += __builtin_popcountll(mask);
    }
    for (int64_t t = count + lane; t < ldk; t += 64) {
        po[t] = (uint32_t)m * ldw4;  // row m of W: zeros
        pt[t] = zrow | (zrow << 16);
        pe[t] = make_float2(0.0f, 0.0f);
    }
    if (lane == 0) nunion[pi] = count;
}

// ---- identity row statistics (Cleaner::calculateSeqIdentity's consumers: selectMethod, getCutPointClusters) --------
// Per sequence: the float32 sum of its identities with every other sequence IN INDEX ORDER (/ (m - 1)), their
// maximum and minimum; then the sums of the row averages and of the row maxima in index order (/ m).  The terms are
// >= 0, so the sequential sums are evaluated a chunk of 256 terms at a time with the binade test of the similarity
// kernel's ordered rows (chunk_step): one wave per sequence instead of one dependent add chain per lane (83 + 24 us
// -> a few us at m = 2000), bit-identical.
__global__ __launch_bounds__(256) void identity_rows_kernel(const float *__restrict__ ident, int m, int ldw,
                                                            float *__restrict__ row_avg, float *__restrict__ row_max,
                                                            float *__restrict__ row_min) {
    const int lane = threadIdx.x & 63;
    const int i = uni((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
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

// (two waves: one per sum.  gate != nullptr: Cleaner::selectMethod's decision is taken here as well -- *gate = 1 when it
// selects gappyout, i.e. the similarity kernel enqueued behind this one has nothing to do; the host takes the same
// decision from the same two numbers when they arrive)
__global__ __launch_bounds__(128) void identity_final_kernel(const float *__restrict__ row_avg, const float *__restrict__ row_max,
                                                             int m, float *__restrict__ out2, int *__restrict__ gate) {
    __shared__ float res[2];
    const int lane = threadIdx.x & 63;
    const int which = uni((int)(threadIdx.x >> 6));
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
    }
}

// codeT -> the compacted lists of one column's valid rows (one wave per column): byte offset of the row in W,
// row index, code; padded behind the last valid row by >= 192 entries of {zero row m, row m, skipped}.
__global__ __launch_bounds__(256) void bx_compact_kernel(const uint8_t *__restrict__ codeT, int64_t ldk, int m, int ncols_pad,
                                                         uint32_t ldw4, uint32_t *__restrict__ voff, uint16_t *__restrict__ vrow,
                                                         uint8_t *__restrict__ vcode, uint16_t *__restrict__ vtrow, int skiprow,
                                                         int32_t *__restrict__ nvalid) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= ncols_pad) return;
    const uint8_t *src = codeT + (size_t)col * ldk;
    uint32_t *po = voff + (size_t)col * ldk;
    uint16_t *pr = vrow + (size_t)col * ldk;
    uint8_t *pc = vcode + (size_t)col * ldk;
    uint16_t *pt = vtrow + (size_t)col * ldk;  // byte offset of the residue's row in a [row][64 lanes] float table
    int count = 0;
    for (int kb = 0; kb < m; kb += 64) {
        const int k = kb + lane;
        const uint32_t code = k < m ? src[k] : BX_SKIP;
        const unsigned long long mask = __ballot(code != BX_SKIP);
        if (code != BX_SKIP) {
            const int pos = count + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            po[pos] = (uint32_t)k * ldw4;
            pr[pos] = (uint16_t)k;
            pc[pos] = (uint8_t)code;
            pt[pos] = (uint16_t)((code >> 3) * 256u);
        }
        count += __builtin_popcountll(mask);
    }
    for (int64_t t = count + lane; t < ldk; t += 64) {
        po[t] = (uint32_t)m * ldw4;  // row m of W: zeros
        pr[t] = (uint16_t)m;         // column m of W: zeros
        pc[t] = (uint8_t)BX_SKIP;
        pt[t] = (uint16_t)(skiprow * 256);  // the table's zero row
    }
    if (lane == 0) nvalid[col] = count;
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
                atomicMax(err_key, ~key);  // (kept complemented: 0 = none, the largest complement = the first residue)
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

void launch_bx_compact(hipStream_t s, const uint8_t *codeT, int m, int n, int ldw, int npos, uint32_t *voff, uint16_t *vrow,
                       uint8_t *vcode, uint16_t *vtrow, int32_t *nvalid) {
    const int ncp = bx_cols_pad(n);
    bx_compact_kernel<<<(ncp + 3) / 4, 256, 0, s>>>(codeT, bx_ldk(m), m, ncp, (uint32_t)ldw * 4u, voff, vrow, vcode, vtrow, npos, nvalid);
}

int bx_cols_per_wave() { return BX_Q; }

// cols: the columns to evaluate (device, ncols entries, a multiple of bx_cols_per_wave(): consecutive entries share a
// wave; pad with the index n, the all-skipped column); the sums of every other column must have been zeroed by the
// caller.  W (both triangles) must be smaller than 4 GB and m < 65535 (checked by the caller).
int launch_similarity_bx(hipStream_t s, const uint32_t *voff, const uint16_t *vrow, const uint8_t *vcode,
                         const int32_t *nvalid, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, const float *wlow,
                         const float *wup, int ldw, const void *tab, float *num_out, float *den_out) {
    const int64_t ldk = bx_ldk(m);
    const int r0 = tuning().bx_r0 >= 0 ? tuning().bx_r0 : BX_R0;
    const int compact = tuning().bx_compact > 0 ? 1 : 0;
    const unsigned grid = (unsigned)((ncols / BX_Q + BX_WAVES - 1) / BX_WAVES);  // ncols is a multiple of BX_Q
    if (grid == 0) return 0;
    const float *t = static_cast<const float *>(tab);
    const uint32_t wbytes = (uint32_t)(bx_wlow_rows(m) * (size_t)ldw * 4);
    const bool stamp = (tuning().sim_mode & 64) != 0, use_asm = tuning().bx_asm != 0;
#define BX_LAUNCH(KERNEL)                                                                                              \
    KERNEL<<<grid, 64 * BX_WAVES, 0, s>>>(voff, vrow, vcode, nvalid, codeT, ldk, m, n, cols, ncols, wlow, wbytes, wup, ldw, r0, \
                                          compact, t, num_out, den_out)
    if (stamp && use_asm) BX_LAUNCH(similarity_bx_asm_kernel<true>);
    else if (stamp) BX_LAUNCH(similarity_bx_kernel<true>);
    else if (use_asm) BX_LAUNCH(similarity_bx_asm_kernel<false>);
    else BX_LAUNCH(similarity_bx_kernel<false>);
#undef BX_LAUNCH
    return 0;
}

// the same contract as launch_similarity_bx plus the 16-bit table-row list; the kernel with per-lane grids (the default)
int launch_similarity_lg(hipStream_t s, const uint32_t *voff, const uint16_t *vrow, const uint8_t *vcode, const uint16_t *vtrow,
                         int npos, const int32_t *nvalid, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols,
                         const float *wlow, const float *wup, int ldw, const void *tab, float *num_out, float *den_out,
                         const int *gate) {
    const int64_t ldk = bx_ldk(m);
    // (diagnostics ride in the high bits of r0: MSA_LG_DBG & 1 -> no W loads, stamped kernel only)
    const int r0 = (tuning().bx_r0 >= 0 ? tuning().bx_r0 : LG_R0) | ((tuning().lg_dbg & 1) << 16) | ((tuning().lg_dbg & 64) ? 0x40000 : 0) |
                   ((tuning().lg_dbg & 128) ? 0x80000 : 0);
    const bool ldst = tuning().lg_regs == 0;
    const int nr = npos + 1;  // table rows per wave in LDS: the alphabet + the zero row
    // eight waves per workgroup while their tables stay within the 64 KB M0 can address (static arrays: 10.5 KB)
    // four waves per workgroup: five workgroups (20 waves) per CU for a 20-letter alphabet, and a workgroup's slots
    // are refilled as soon as its four columns are done (eight per workgroup, MSA_LG_DBG & 2: 4.0 instead of 3.8 ms at C3)
    const int waves = ldst ? ((tuning().lg_dbg & 2) && (size_t)8 * nr * 256 + 10752 <= 65536 ? 8 : 4) : BX_WAVES;
    const size_t dyn = ldst ? (size_t)waves * nr * 256 : 0;
    const unsigned grid = (unsigned)((ncols + waves - 1) / waves);
    if (grid == 0) return 0;
    const float *t = static_cast<const float *>(tab);
    const uint32_t wbytes = (uint32_t)(bx_wlow_rows(m) * (size_t)ldw * 4);
    const bool stamp = (tuning().sim_mode & 64) != 0;
#define LG_LAUNCH(KERNEL)                                                                                              \
    KERNEL<<<grid, 64 * waves, dyn, s>>>(voff, vrow, vcode, vtrow, nr, nvalid, codeT, ldk, m, n, cols, ncols, wlow, wbytes, wup, ldw, r0, \
                                         t, num_out, den_out, gate)
    if (ldst) {
        const void *k = stamp ? (const void *)similarity_lg_kernel<true> : (const void *)similarity_lg_kernel<false>;
        const int e = set_max_lds_once(k, (int)dyn);
        if (e) return e;
        if (stamp) LG_LAUNCH(similarity_lg_kernel<true>);
        else LG_LAUNCH(similarity_lg_kernel<false>);
    } else {
        if (stamp) LG_LAUNCH(similarity_lg_regs_kernel<true>);
        else LG_LAUNCH(similarity_lg_regs_kernel<false>);
    }
#undef LG_LAUNCH
    return 0;
}

// Two columns per wave (consecutive entries of `cols`, ncols even, padded with the all-skipped column n; `waves`
// consecutive pairs share a workgroup): the union lists are built here from the column order.  npos + 1 <= 23
// (16-bit table-row offsets).
bool lg2_fits(int npos) { return npos + 1 <= 23; }
int lg2_max_waves() { return LG2_WAVES; }
int launch_similarity_lg2(hipStream_t s, const uint32_t *voff, const uint16_t *vrow, const uint8_t *vcode, int npos,
                          const int32_t *nvalid, const uint8_t *codeT, int m, int n, const int32_t *cols, int ncols, int waves,
                          uint32_t *uoff, uint32_t *utt, float *uee, int32_t *nunion, const float *wlow, const float *wup, int ldw,
                          const void *tab, float *num_out, float *den_out) {
    const int64_t ldk = bx_ldk(m);
    // (diagnostics ride in the high bits of r0: MSA_LG_DBG & 16 -> no wave priorities)
    const int r0 = (tuning().bx_r0 >= 0 ? tuning().bx_r0 : LG_R0) | ((tuning().lg_dbg & 16) ? 0x20000 : 0);
    const int npairs = ncols / 2;
    if (npairs == 0) return 0;
    const int nr = npos + 1;
    lg2_union_kernel<<<(npairs + 3) / 4, 256, 0, s>>>(codeT, ldk, m, cols, npairs, (uint32_t)ldw * 4u, nr, uoff, utt, uee, nunion);
    const size_t dyn = (size_t)nr * nr * 128;
    if (waves < 1 || waves > LG2_WAVES) return (int)hipErrorInvalidValue;
    const unsigned grid = (unsigned)((npairs + waves - 1) / waves);
    const float *t = static_cast<const float *>(tab);
    const uint32_t wbytes = (uint32_t)(bx_wlow_rows(m) * (size_t)ldw * 4);
    const bool stamp = (tuning().sim_mode & 64) != 0;
    const void *k = stamp ? (const void *)similarity_lg2_kernel<true> : (const void *)similarity_lg2_kernel<false>;
    const int e = set_max_lds_once(k, (int)dyn);
    if (e) return e;
    if (stamp)
        similarity_lg2_kernel<true><<<grid, 64 * waves, dyn, s>>>(voff, vrow, vcode, nvalid, codeT, ldk, m, n, cols, npairs, uoff, utt, uee,
                                                                     nunion, nr, wlow, wbytes, wup, ldw, r0, t, num_out, den_out);
    else
        similarity_lg2_kernel<false><<<grid, 64 * waves, dyn, s>>>(voff, vrow, vcode, nvalid, codeT, ldk, m, n, cols, npairs, uoff, utt, uee,
                                                                      nunion, nr, wlow, wbytes, wup, ldw, r0, t, num_out, den_out);
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
    }
    return rc;
}

extern "C" int msa_debug_bx_records(unsigned int *out, int nwaves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bx_rec), sizeof(unsigned int) * 8 * (size_t)(nwaves < 16384 ? nwaves : 16384));
}

}  // namespace msak
