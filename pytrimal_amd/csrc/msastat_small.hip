// msastat_small.hip -- the layout kernels of the similarity pass (column-major codes, the compacted lists of every column's valid
// rows, the mean weights), the identity row statistics (sequential float32 sums through the binade test), and the kernels of
// small alignments: the flat similarity kernel, the one-launch front of the compact pipeline, the lane-per-column kernel of
// batches (msa_trim_batch).  The similarity kernel itself: msastat_simx.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "msastat_kernels.h"
#include "msastat_device.h"
#include "msastat_sums.h"

namespace msak {
namespace {

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
    // 1024 terms requested at a time, then four steps of 256: a row of up to 1024 identities waits for memory once (round 5
    // requested a chunk, added it up, requested the next: four round trips at 1000 sequences, and the step itself is ~40 instructions)
    for (int base = 0; base < m; base += 1024) {
        float x[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = base + 256 * q + 64 * c + lane;
                x[q][c] = (t < m && t != i) ? r[t] : 0.0f;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = base + 256 * q + 64 * c + lane;
                const float v = x[q][c];
                if (t < m && t != i) {
                    mx = mx < v ? v : mx;
                    mn = mn > v ? v : mn;
                }
            }
            if (base + 256 * q < m) s = chunk_step(s, x[q]);
        }
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
        for (int base = 0; base < m; base += 1024) {  // (1024 terms requested at a time: identity_rows_body)
            float xa[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int t = base + 256 * q + 64 * c + lane;
                    xa[q][c] = t < m ? src[t] : 0.0f;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (base + 256 * q < m) a = chunk_step(a, xa[q]);
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
// rows in the reference's order, 64 U at a time through scan_lanes (flat_add_run).  Its valid rows (index, code) are compacted into LDS from the
// column-major codes; W[j][k] a gather from the (L2-resident) upper triangle; the distance from the LDS table.  No lists, no
// predictor, no per-row state.  Two waves per column (one per sum); the numerator's wave writes MDK and Q itself (mdk_value).
constexpr int FLAT_ROWS_MAX = 512;
template <int U>
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
    // U CONSECUTIVE terms per lane and pass (one run of 64 U terms through flat_add_run: one scan per pass -- with four terms per
    // lane, what rounds 4 shipped, a column of 200 rows was 78 scans per sum), requested a pass ahead: with two waves per SIMD
    // (two per column, ~1000 columns) nothing else hides the gather's latency.
    int rt = U * lane;  // the lane's first term of the next pass
    const float fn = (float)(2 * nv - 1);
    auto request = [&](float (&w)[U], uint32_t (&ti)[U]) {
        uint32_t ea[U], eb[U];
        {
            // the lane's first term of the pass -> (row a, place p): the largest a with a nv - a (a + 1) / 2 <= t, from the root
            // of an exact integer below 2^24 (two corrections cover a root that is off by an ulp); its neighbours by a carry
            // into the next row.  Straight-line code: the gathers below leave back to back.  Terms behind the last one
            // read the last one's operands and count as zero (the caller masks them).
            const int t0 = rt;
            const int t = t0 < T ? t0 : T - 1;
            int a = (int)((fn - sqrtf(fn * fn - 8.0f * (float)t)) * 0.5f);
            a = a > nv - 2 ? nv - 2 : a;
            a -= (a * nv - a * (a + 1) / 2 > t) ? 1 : 0;
            a += (a + 1 <= nv - 2 && (a + 1) * nv - (a + 1) * (a + 2) / 2 <= t) ? 1 : 0;
            int pl = t - (a * nv - a * (a + 1) / 2);
#pragma unroll
            for (int i = 0; i < U; ++i) {
                ea[i] = mine[a], eb[i] = mine[a + 1 + pl];
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
            if (t0 + 64 * U < T) request(wn, tn);
            float x[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                if (t0 + U * lane + i >= T) w[i] = 0.0f;
                const float d = tab[ti[i]].x;  // (both waves read it: no branch around the LDS loads)
                x[i] = denominator ? w[i] : w[i] * d;
            }
            sum = flat_add_run<U>(sum, x, lane);
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
// (ROWS: rows of the LDS code array of a column block -- 512: 33 KB, or 1024: 66 KB of the CU's 160)
constexpr int COMPACT_TEAMS_MAX = 4;  // 64-row tiles a column block works on at once (a team of four waves each)
template <bool SIM, int ROWS>
__device__ __forceinline__ void compact_column_block(const CompactArgs &a, int b) {
    constexpr int LDC = ROWS + 4;  // bytes per column (4 past a multiple of 128: consecutive columns on different banks)
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
template <bool SIM, int ROWS>
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
    if (b < ncb) compact_column_block<SIM, ROWS>(a, b);
    else if (threadIdx.x < 256) row_nongap_body(a.raw, a.m, a.n, a.ld, nullptr, a.hres + a.h_rowtot, b - ncb);
}

// ---- the front kernel from 513 sequences on (round 6): NARROW column blocks -------------------------------------------------
// At 1000 x 4000 the kernel above is 63 column blocks of 1024 threads and 92 KB of LDS, each a chain of four passes over sixteen
// 64-row tiles: 41 us for one pass over 4 MB on a quarter of the chip, and in a batch its workgroups wait for three of a compute
// unit's similarity workgroups to leave (profiles/r05_c5_timeline.txt: 379 us).  Here a column block owns CW = 16 (or 32) columns
// over ALL rows and requests every one of its rows before it looks at the first:
//   * thread = one dword (four columns) of four CONSECUTIVE rows per sweep, NS sweeps: 4 NS dword loads in flight per thread, a
//     wave's load covers 16 rows x 16 bytes (CW = 16); the gap / indetermination counts are SWAR byte counters, widened to 16 bits
//     and summed over the wave's lanes of the same dword by shuffles, over the waves through LDS;
//   * the codes of a thread's four rows are one dword of the LDS array [column][row] (LDC4 = ROWS / 4 + 64 / CW dwords per column:
//     the lanes of a wave land on 64 different banks), written once, read back coalesced by the wave that writes the column out
//     (codes, compacted lists: as above);
//   * the bit planes come from blocks of their own (a thread per row and 64 columns: prep_planes_core) -- the rows are in device
//     memory at this size, a second read costs nothing, and a column block narrower than a plane word (32 columns) becomes possible;
//   * the non-ASCII verdict is the column blocks' (they see every byte the plane blocks see).
// 250 column blocks + 252 plane blocks + 250 row blocks of 256 active threads at 1000 x 4000 instead of 63 + 250.
template <int ROWS, int CW, int NT>
__device__ __forceinline__ void compact_column_block2(const CompactArgs &a, int b) {
    constexpr int WPR = CW / 4;               // dwords per row of the block
    constexpr int RG = NT / WPR;              // groups of four consecutive rows per sweep
    constexpr int NS = ROWS / (4 * RG);       // sweeps
    constexpr int LDC4 = ROWS / 4 + 64 / CW;  // dwords per column of the LDS code array
    constexpr int NW = NT / 64;
    static_assert(NS >= 1 && NS * 4 * RG == ROWS && (ROWS / 4) % 64 == 0, "shape of the narrow front kernel");
    __shared__ uint8_t lut[256];
    __shared__ uint32_t codes[CW * LDC4];
    __shared__ uint32_t cnt[2][2][NW][WPR];  // [gaps / indeterminations][even / odd bytes][wave][dword]
    __shared__ uint32_t firstbad[CW];
    __shared__ uint8_t skipc[CW];
    __shared__ int anybad;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int word = t % WPR, rg = t / WPR;
    const int m = a.m, n = a.n;
    const int64_t ld = a.ld, ldk = a.ldk;
    const int c0 = b * CW + 4 * word;
    const bool inb = c0 < n;
    // bytes of the thread's dword inside the alignment (columns < n)
    const uint32_t km = !inb ? 0u : (n - c0 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * (n - c0))) - 1u));
    uint32_t x[NS][4];
    {
        const uint8_t *p = a.raw + (inb ? c0 : 0);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * (s * RG + rg) + i;
                x[s][i] = (row < m && inb) ? *reinterpret_cast<const uint32_t *>(p + (size_t)row * ld) : 0u;
            }
    }
    for (int i = t; i < 256; i += NT) lut[i] = a.lut[i];
    if (t < CW) firstbad[t] = 0xFFFFFFFFu;
    if (t == 0) anybad = 0;
    __syncthreads();
    const uint32_t indet4 = a.indet4;
    uint32_t g = 0, xi = 0, bad = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * (s * RG + rg) + i;
            const uint32_t v = x[s][i];
            const uint32_t k4 = row < m ? km : 0u;
            g += (zero_bytes(v ^ 0x2d2d2d2du) & k4) >> 7;
            xi += (zero_bytes(v ^ indet4) & k4) >> 7;
            bad |= v & k4 & 0x80808080u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t code = BX_SKIP;
                if ((k4 >> (8 * k)) & 1u) {
                    const uint32_t byte = (v >> (8 * k)) & 0xFFu;
                    code = lut[byte];  // 8 x table row, 224 = skipped, 0xFE / 0xFF = bad symbol
                    if (code >= 0xFEu) {
                        atomicMin(&firstbad[4 * word + k], ((uint32_t)row << 16) | ((code & 1u) << 8) | byte);
                        code = BX_SKIP;
                    }
                }
                pk[k] |= code << (8 * i);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) codes[(4 * word + k) * LDC4 + s * RG + rg] = pk[k];
    }
    if (bad) anybad = 1;
    {  // byte counters (<= 4 NS per byte) -> 16-bit fields, summed over the wave's lanes that hold the same dword of other rows
        uint32_t f[4] = {g & 0x00FF00FFu, (g >> 8) & 0x00FF00FFu, xi & 0x00FF00FFu, (xi >> 8) & 0x00FF00FFu};
#pragma unroll
        for (int off = WPR; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] += (uint32_t)__shfl_xor((int)f[q], off, 64);
        if (lane < WPR) {
            cnt[0][0][wave][lane] = f[0], cnt[0][1][wave][lane] = f[1];
            cnt[1][0][wave][lane] = f[2], cnt[1][1][wave][lane] = f[3];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const bool mine = lane < CW;
        const int c = b * CW + lane;
        uint32_t G = 0, X = 0;
        if (mine) {
            const int wd = lane >> 2, odd = lane & 1, sh = (lane & 2) * 8;
            for (int w = 0; w < NW; ++w) G += (cnt[0][odd][w][wd] >> sh) & 0xFFFFu, X += (cnt[1][odd][w][wd] >> sh) & 0xFFFFu;
            if (c < n) {
                a.gaps[c] = (int32_t)G;
                a.indets[c] = (int32_t)X;
                a.hres[a.h_gaps + c] = (int32_t)G;
                a.hres[a.h_indets + c] = (int32_t)X;
            }
        }
        const bool skip = !mine || c >= n || (((float)(int32_t)G / (float)m) >= 0.8f);
        if (mine) skipc[lane] = skip ? 1 : 0;
        // the block's first bad residue: smallest column, then smallest row -- the key of sim_encode_cm, complemented
        unsigned long long key = ~0ull;
        if (!skip && firstbad[lane & (CW - 1)] != 0xFFFFFFFFu) key = ((unsigned long long)c << 40) | firstbad[lane & (CW - 1)];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), off, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            key = other < key ? other : key;
        }
        if (lane == 0) {
            const int ncb = a.ncols_pad / CW;
            key = key == ~0ull ? 0ull : ~key;
            a.hres[a.h_slots + 2 * b] = (int32_t)(uint32_t)key;
            a.hres[a.h_slots + 2 * b + 1] = (int32_t)(uint32_t)(key >> 32);
            a.hres[a.h_slots + 2 * ncb + b] = anybad;
        }
    }
    __syncthreads();
    // a wave per column, lane = row: the codes (coalesced) and the compacted lists, as sim_encode_cm and bx_compact write them
    const uint32_t ldw4 = (uint32_t)a.ldw * 4u;
    const int mtiles = (m + 63) / 64;
    const uint8_t *cbytes = reinterpret_cast<const uint8_t *>(codes);
    for (int q = wave; q < CW; q += NW) {
        const size_t col = (size_t)b * CW + q;
        const bool skip = skipc[q] != 0;
        uint8_t *ct = a.codeT + col * ldk;
        uint32_t *po = a.voff + col * ldk;
        uint16_t *pt = a.vtrow + col * ldk;
        int count = 0;
        for (int kb = 0; kb < mtiles * 64; kb += 64) {
            const int k = kb + lane;
            const uint32_t code = (k < m && !skip) ? (uint32_t)cbytes[q * (LDC4 * 4) + k] : BX_SKIP;
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
        for (int64_t e = count + lane; e < ldk; e += 64) {
            po[e] = a.big ? (uint32_t)m : (uint32_t)m * ldw4;  // row m of W: zeros
            pt[e] = (uint16_t)(a.skiprow * 256);               // the table's zero row
        }
        if (lane == 0) a.nvalid[col] = count;
    }
}
template <int ROWS, int CW, int NT>
__global__ __launch_bounds__(NT) void compact_front2_kernel(CompactArgs a) {
    const int b = (int)blockIdx.x;
    const int ncb = a.ncols_pad / CW;
    const int ncp = (a.nchunk + 1) / 2, npb = ncp * ((a.m_pad + 255) / 256);
    if (b == 0) {
        if (threadIdx.x < 32) a.flags[threadIdx.x] = 0;
        if (threadIdx.x == 0) a.scratch[0] = 0;  // the ticket of the identity statistics
        for (int i = threadIdx.x; i < a.m_pad + 64; i += NT) a.wsum[i] = 0u;
    }
    if (b < ncb) {
        // consecutive blocks go to different XCDs: the eight (four) blocks that share the 128-byte lines of a row are dealt to ONE
        // XCD -- XCD x = b % 8 owns a contiguous range of column groups -- so that its L2 fetches a line once
        const int x = b & 7, q = b >> 3, full = ncb >> 3, rem = ncb & 7;
        const int cg = a.xcd ? x * full + (x < rem ? x : rem) + q : b;
        compact_column_block2<ROWS, CW, NT>(a, cg);
    } else if (threadIdx.x < 256) {
        if (b < ncb + npb) {
            const int pb = b - ncb;
            (void)prep_planes_core(a.raw, a.m, a.n, a.ld, a.indet4, a.planes, a.nchunk, a.m_pad, pb % ncp, pb / ncp);
        } else row_nongap_body(a.raw, a.m, a.n, a.ld, nullptr, a.hres + a.h_rowtot, b - ncb - npb);
    }
}

// ---- codes + lists in ONE pass, any alignment whose sixteen-column block fits the LDS (round 6) -------------------------------
// sim_encode_cm + bx_compact are two passes with the column-major codes written to memory and read back in between (and byte loads
// behind a 64 x 64 transposition): 0.28 ms at 2000 x 10000 on the side stream, 0.60 at 3583 x 7287 -- beside a pair pass of 0.63,
// one step from the critical path.  The narrow column block of compact_front2_kernel does the same work in one pass: sixteen
// columns over ALL rows, dword loads of four consecutive rows per thread and sweep (the next sweep's requested before this one's
// are looked at), the codes in an LDS array [column][row], then a wave per column writes codes and lists out, coalesced.  The
// ">= 80 % gaps" cut comes from the gap counts of the kernel in front (windowed or not), the first bad residue goes to the
// context's key by atomicMax as the other encode kernels report it.  Dynamic LDS: 16 x (round_up(ceil(m / 4), 64) + 4) dwords
// (m = 2000: 33 KB, 3583: 58 KB, 8000: 131 KB): up to ~10 000 rows, beyond which the two-pass kernels stay.
__global__ __launch_bounds__(1024) void sim_lists_fused_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                               const uint8_t *__restrict__ lut_g, const int32_t *__restrict__ gaps_w,
                                                               uint8_t *__restrict__ codeT, int64_t ldk, int ncols_pad, uint32_t ldw4,
                                                               uint32_t *__restrict__ voff, uint16_t *__restrict__ vtrow, int skiprow,
                                                               int32_t *__restrict__ nvalid, int big, unsigned long long *__restrict__ err_key,
                                                               int ldc4) {
    constexpr int CW = 16, WPR = 4, NT = 1024, RG = NT / WPR, NW = NT / 64;
    extern __shared__ uint32_t fused_codes[];  // [CW][ldc4]
    __shared__ uint8_t lut[256];
    __shared__ uint8_t skipc[CW];
    const int ncb = ncols_pad / CW;
    // (the blocks that share the 128-byte lines of a row dealt to one XCD: compact_front2_kernel)
    const int bx = (int)blockIdx.x, xq = bx >> 3, xx = bx & 7, full = ncb >> 3, rem = ncb & 7;
    const int b = xx * full + (xx < rem ? xx : rem) + xq;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int word = t % WPR, rg = t / WPR;
    const int c0 = b * CW + 4 * word;
    if (t < 256) lut[t] = lut_g[t];
    if (t < CW) {
        const int c = b * CW + t;
        skipc[t] = (c >= n || (((float)gaps_w[c] / (float)m) >= 0.8f)) ? 1 : 0;
    }
    __syncthreads();
    const bool live = c0 < n && !(skipc[4 * word] & skipc[4 * word + 1] & skipc[4 * word + 2] & skipc[4 * word + 3]);
    const uint32_t km = c0 >= n ? 0u : (n - c0 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * (n - c0))) - 1u));
    uint32_t keep = 0;  // bytes of the dword whose column is evaluated
#pragma unroll
    for (int k = 0; k < 4; ++k) keep |= skipc[4 * word + k] ? 0u : (0xFFu << (8 * k));
    keep &= km;
    const uint8_t *p = raw + (c0 < n ? c0 : 0);
    const int nsweeps = (m + 4 * RG - 1) / (4 * RG);
    uint32_t nx[4];
    auto request = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * (s * RG + rg) + i;
            nx[i] = (live && row < m) ? *reinterpret_cast<const uint32_t *>(p + (size_t)row * ld) : 0u;
        }
    };
    request(0);
    for (int s = 0; s < nsweeps; ++s) {
        uint32_t x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = nx[i];
        if (s + 1 < nsweeps) request(s + 1);
        uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * (s * RG + rg) + i;
            const uint32_t k4 = row < m ? keep : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t code = BX_SKIP;
                if ((k4 >> (8 * k)) & 1u) {
                    const uint32_t byte = (x[i] >> (8 * k)) & 0xFFu;
                    code = lut[byte];  // 8 x table row, 224 = skipped, 0xFE / 0xFF = bad symbol
                    if (code >= 0xFEu) {
                        const unsigned long long key = ((unsigned long long)(c0 + k) << 40) | ((unsigned long long)row << 16) |
                                                       ((unsigned long long)(code & 1u) << 8) | byte;
                        atomicMax(err_key, ~key);  // (kept complemented: 0 = none, the largest complement = the first residue)
                        code = BX_SKIP;
                    }
                }
                pk[k] |= code << (8 * i);
            }
        }
        const int rgi = s * RG + rg;
        if (4 * rgi < m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) fused_codes[(4 * word + k) * ldc4 + rgi] = pk[k];
        }
    }
    __syncthreads();
    const int mtiles = (m + 63) / 64;
    const uint8_t *cbytes = reinterpret_cast<const uint8_t *>(fused_codes);
    for (int q = wave; q < CW; q += NW) {
        const size_t col = (size_t)b * CW + q;
        const bool skip = skipc[q] != 0;
        uint8_t *ct = codeT + col * ldk;
        uint32_t *po = voff + col * ldk;
        uint16_t *pt = vtrow + col * ldk;
        int count = 0;
        for (int kb = 0; kb < mtiles * 64; kb += 64) {
            const int k = kb + lane;
            const uint32_t code = (k < m && !skip) ? (uint32_t)cbytes[(size_t)q * ldc4 * 4 + k] : BX_SKIP;
            ct[k] = (uint8_t)code;
            const unsigned long long mask = __ballot(code != BX_SKIP);
            if (code != BX_SKIP) {
                const int pos = count + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                po[pos] = big ? (uint32_t)k : (uint32_t)k * ldw4;
                pt[pos] = (uint16_t)((code >> 3) * 256u);
            }
            count += __builtin_popcountll(mask);
        }
        for (int64_t k = (int64_t)mtiles * 64 + lane; k < ldk; k += 64) ct[k] = (uint8_t)BX_SKIP;
        for (int64_t e = count + lane; e < ldk; e += 64) {
            po[e] = big ? (uint32_t)m : (uint32_t)m * ldw4;  // row m of W: zeros
            pt[e] = (uint16_t)(skiprow * 256);               // the table's zero row
        }
        if (lane == 0) nvalid[col] = count;
    }
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

// OverlapTrimmer on a small alignment, behind the front kernel (compact_overlap, msastat_trim.hip): two launches whose results the
// kernels store into pinned host memory themselves -- where the ordinary sequence is the two overlap kernels, the device's decision,
// a memset, the column counts and three copies (the reference's ENOG411BWBU fixture, OverlapTrimmer(80, 0.8): 0.083 ms).
//   1. a wave per sequence: Cleaner::calculateSpuriousVector's closed form (overlap_rows_kernel, msastat_kernels.hip) with the
//      columns' three verdicts -- enough other sequences agree with a residue / a gap / an indetermination -- taken from the front
//      kernel's counts as the wave walks its row (four columns per lane and load: two 16-byte loads of counts beside the row's
//      dword), then the sequence's own verdict as the host will take it (the same float division and comparison);
//   2. the residues per column over the sequences that stay (what removeAllGapsSeqsAndCols asks for when sequences go): a
//      workgroup of sixteen waves per 256 columns, wave w on the rows w, w + 16, ..., byte counters (<= 64 rows per wave),
//      summed through LDS -- plain stores: no atomics, hence no memset.
__global__ __launch_bounds__(256) void overlap_small_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld, uint32_t indet4,
                                                            const int32_t *__restrict__ gaps, const int32_t *__restrict__ indets, int need,
                                                            float min_ov, int32_t *__restrict__ good, int32_t *__restrict__ h_good,
                                                            uint8_t *__restrict__ keep, uint8_t *__restrict__ h_keep) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m) return;
    typedef int i4 __attribute__((ext_vector_type(4)));
    const uint32_t *p = reinterpret_cast<const uint32_t *>(raw + (size_t)row * ld);
    int cnt = 0;
    for (int c4 = lane; c4 * 4 < n; c4 += 64) {
        const uint32_t x = p[c4];
        const i4 g4 = *reinterpret_cast<const i4 *>(gaps + 4 * c4), x4 = *reinterpret_cast<const i4 *>(indets + 4 * c4);
        const uint32_t isg = zero_bytes(x ^ 0x2d2d2d2du), isi = zero_bytes(x ^ indet4);
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
    if (lane == 0) {
        good[row] = cnt, h_good[row] = cnt;
        const uint8_t k = (static_cast<float>(cnt) / n) < min_ov ? 0 : 1;
        keep[row] = k, h_keep[row] = k;
    }
}
__global__ __launch_bounds__(1024) void col_nongap_small_kernel(const uint8_t *__restrict__ raw, int m, int n, int64_t ld,
                                                               const uint8_t *__restrict__ keep_seq, int32_t *__restrict__ col_nongap,
                                                               int32_t *__restrict__ h_col_nongap) {
    __shared__ uint32_t part[16][64];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int c4 = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    if ((int64_t)c4 * 4 < ld) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(raw) + c4;
        const size_t stride = (size_t)(ld >> 2);
#pragma unroll 4
        for (int r = wave; r < m; r += 16) {
            const uint32_t x = p[(size_t)r * stride];
            const uint32_t nongap = (~zero_bytes(x ^ 0x2d2d2d2du) & 0x80808080u) >> 7;
            acc += keep_seq[r] ? nongap : 0u;  // (wave-uniform)
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int d = (int)(threadIdx.x >> 2), k = (int)(threadIdx.x & 3);
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += (part[w][d] >> (8 * k)) & 0xFFu;
        const int c = (blockIdx.x * 64 + d) * 4 + k;
        if (c < n) col_nongap[c] = (int32_t)v, h_col_nongap[c] = (int32_t)v;
    }
}

// msa_trim_batch's engine: the columns of every alignment of a group by weight (valid rows, most first; the columns the ">= 80 %
// gaps" rule cuts have the fewest and come last), for the wave-per-column kernel's grid.  A workgroup per alignment, a counting sort
// in LDS: bins by gaps + indeterminations, exclusive prefix, scatter (the order inside a bin is whatever the atomics make it: any
// order is correct, the deal only sets the launch's time).
__global__ __launch_bounds__(256) void sort_columns_batch_kernel(const BAlign *__restrict__ table, const LgAlign *__restrict__ lg) {
    extern __shared__ int bins[];  // [m + 2]
    __shared__ int part[256];
    const BAlign d = batch_desc(table, (int)blockIdx.x);
    int32_t *cols = const_cast<int32_t *>(lg[blockIdx.x].cols);
    if (!cols) return;
    const int m = d.m, n = d.n, nb = m + 2, tid = (int)threadIdx.x;
    for (int i = tid; i < nb; i += 256) bins[i] = 0;
    __syncthreads();
    for (int j = tid; j < n; j += 256) atomicAdd(&bins[min(d.gaps[j] + d.indets[j], m) + 1], 1);
    __syncthreads();
    // inclusive prefix over the bins: a chunk per thread, the chunks' totals by one wave-sized pass
    const int per = (nb + 255) / 256, lo = min(tid * per, nb), hi = min(lo + per, nb);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += bins[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) {
            const int v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = part[tid];
    for (int i = lo; i < hi; ++i) {
        run += bins[i];
        bins[i] = run;  // bins[k + 1] = columns with a key <= k; bins[k] = where key k starts
    }
    __syncthreads();
    for (int j = tid; j < n; j += 256) cols[atomicAdd(&bins[min(d.gaps[j] + d.indets[j], m)], 1)] = j;
}

// A contiguous host matrix whose rows are no multiple of 16 bytes (5000 x 5000: the BASELINE's C4) came up in ONE linear copy;
// this lays the rows out at the device pitch: a thread per 16 destination bytes (byte loads: a source row starts anywhere),
// the padding columns zeroed.  50 MB through the HBM for a 25 MB matrix: ~ 20 us.
__global__ __launch_bounds__(256) void repitch_rows_kernel(const uint8_t *__restrict__ src, int64_t ld_src, uint8_t *__restrict__ dst,
                                                           int64_t ld_dst, int m, int n) {
    const int64_t per_row = ld_dst >> 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= per_row * m) return;
    const int row = (int)(t / per_row);
    const int c0 = (int)(t - (int64_t)row * per_row) * 16;
    const uint8_t *p = src + (size_t)row * ld_src + c0;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (c0 + 16 <= n) {
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    } else {
        for (int i = 0; c0 + i < n; ++i) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    }
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4 *>(dst + (size_t)row * ld_dst + c0) = u4{w[0], w[1], w[2], w[3]};
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

// codes + lists of every column in one pass (sim_lists_fused_kernel), where a sixteen-column block's codes fit the LDS; false: the
// caller runs launch_sim_encode_cm + launch_bx_compact (more than ~10 000 rows, or MSA_LISTS_FUSED=0)
bool launch_sim_lists_fused(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *lut, const int32_t *gaps_w, uint8_t *codeT,
                            int ldw, int npos, uint32_t *voff, uint16_t *vtrow, int32_t *nvalid, unsigned long long *err_key) {
    const int ldc4 = ((m + 3) / 4 + 63) / 64 * 64 + 4;
    const size_t dyn = (size_t)16 * ldc4 * 4;
    if (!gaps_w || tuning().lists_fused == 0 || dyn > (size_t)150 * 1024) return false;
    if (set_max_lds_once((const void *)sim_lists_fused_kernel, (int)dyn)) return false;
    const int ncp = bx_cols_pad(n);
    sim_lists_fused_kernel<<<(unsigned)(ncp / 16), 1024, dyn, s>>>(raw, m, n, ld, lut, gaps_w, codeT, bx_ldk(m), ncp, (uint32_t)ldw * 4u, voff, vtrow,
                                                                  npos, nvalid, lg_big(m, ldw) ? 1 : 0, err_key, ldc4);
    return true;
}

void launch_bx_compact(hipStream_t s, const uint8_t *codeT, int m, int n, int ldw, int npos, uint32_t *voff, uint16_t *vtrow,
                       int32_t *nvalid) {
    const int ncp = bx_cols_pad(n);
    bx_compact_kernel<<<(ncp + 3) / 4, 256, 0, s>>>(codeT, bx_ldk(m), m, ncp, (uint32_t)ldw * 4u, voff, vtrow, npos, nvalid,
                                                    lg_big(m, ldw) ? 1 : 0);
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
size_t compact_slot_words(int n) { return (size_t)3 * (bx_cols_pad(n) / 16) + 2; }  // (the narrowest column block: 16 columns)
size_t compact_scratch_words(int m, int n) { return (size_t)2 + (std::max(m, 1) + 127) / 128 * 128 + 64 + 8; }
// Columns per column block of the front kernel: 64 = compact_front_kernel (blocks of up to 1024 threads walking 64-row tiles, the
// planes from the tile in LDS: the kernel of the alignments whose rows lie in pinned host memory -- every byte crosses the link once
// -- and of up to 128 sequences), 16 = compact_front2_kernel from 129 sequences on when the rows are in device memory (every row of a
// block requested at once, planes from blocks of their own: profiles/r06_front_pairs_ab.txt, front kernel at 150 x 1200 16.9 -> 10.5 us,
// 500 x 2000 27.2 -> 15.0, 1000 x 4000 42.6 -> 17.3).  MSA_FRONT_CW / MSA_FRONT_FROM_M: A/B and tests.
#ifndef MSA_FRONT2_FROM_M
#define MSA_FRONT2_FROM_M 129
#endif
int compact_front_cw(int m, int n, bool sim, bool rows_in_host_memory) {
    (void)n;
    if (!sim) return 64;
    const Tuning &t = tuning();
    const int from = t.front_from_m > 0 ? t.front_from_m : MSA_FRONT2_FROM_M;
    if (t.front_cw == 64 || m < from || m > 1024 || (rows_in_host_memory && t.front_from_m <= 0)) return 64;
    return t.front_cw == 32 ? 32 : 16;
}
// threads per block of the narrow kernel: 512.  Alone, a block of 1024 threads (one sweep over 1024 rows) is 2 us faster at
// 1000 x 4000 (17.3 against 19.3 us); beside other contexts' similarity kernels -- five waves per SIMD that leave the register file
// no room for a sixth -- a block must find all of its waves' slots on ONE compute unit, and eight waves find them sooner than
// sixteen: the C5 batch 22.1 - 22.3 -> 21.75 ms (profiles/r06_c5_front_threads_ab.txt; 256 threads: 21.9), and at 1024 x 8000
// (501 column blocks) 512 is the faster alone as well (26.9 against 30.0 us).
int compact_front_nt(int m, int n, int cus) {
    (void)m, (void)n, (void)cus;
    if (tuning().front_nt > 0) return tuning().front_nt;
    return 512;
}
template <int ROWS>
static void launch_compact_front2(hipStream_t s, const CompactArgs &a, unsigned blocks, int nt) {
    if (a.cw == 32) {
        if (nt <= 256) compact_front2_kernel<ROWS, 32, 256><<<blocks, 256, 0, s>>>(a);
        else if (nt <= 512) compact_front2_kernel<ROWS, 32, 512><<<blocks, 512, 0, s>>>(a);
        else compact_front2_kernel<ROWS, 32, 1024><<<blocks, 1024, 0, s>>>(a);
    } else {
        constexpr int TOP = ROWS == 512 ? 512 : 1024;  // (sixteen columns of 512 rows are one sweep of 512 threads)
        if (nt <= 256) compact_front2_kernel<ROWS, 16, 256><<<blocks, 256, 0, s>>>(a);
        else if (nt <= 512 || TOP == 512) compact_front2_kernel<ROWS, 16, 512><<<blocks, 512, 0, s>>>(a);
        else compact_front2_kernel<ROWS, 16, TOP><<<blocks, TOP, 0, s>>>(a);
    }
}
void launch_compact_front(hipStream_t s, const CompactArgs &a) {
    if (a.sim && a.cw != 64) {
        const int ncp = (a.nchunk + 1) / 2;
        const unsigned blocks = (unsigned)(a.ncols_pad / a.cw + ncp * ((a.m_pad + 255) / 256) + (a.m + 3) / 4);
        const int nt = a.nt > 0 ? a.nt : 512;
        if (a.m > 512) launch_compact_front2<1024>(s, a, blocks, nt);
        else launch_compact_front2<512>(s, a, blocks, nt);
        return;
    }
    const unsigned blocks = (unsigned)(a.ncols_pad / 64 + (a.m + 3) / 4);
    const int teams = std::min(COMPACT_TEAMS_MAX, std::max(1, (a.m + 63) / 64));  // a team of four waves per 64-row tile, up to four
    if (a.sim && a.m > 512) compact_front_kernel<true, 1024><<<blocks, 256 * teams, 0, s>>>(a);
    else if (a.sim) compact_front_kernel<true, 512><<<blocks, 256 * teams, 0, s>>>(a);
    else compact_front_kernel<false, 512><<<blocks, 256 * teams, 0, s>>>(a);
}
// the flat similarity kernel: any alignment of up to FLAT_ROWS_MAX rows whose codes exist (A.codeT, A.wup, A.mdk_out, A.q_out)
int flat_rows_max() { return FLAT_ROWS_MAX; }
void launch_similarity_flat(hipStream_t s, const LgAlign &one, const void *tab) {
    // terms per lane and scan: eight, sixteen from ~110 sequences on (profiles/r05_flat_sweep.jsonl: four, what round 4 shipped, is 12 %
    // behind at 128 sequences and 25 % at 256; MSA_FLAT_U forces 4, 8 or 16: tools/flat_sweep.py)
    const int forced = tuning().flat_u;
    const int u = forced ? forced : (one.m >= 112 ? 16 : 8);
    const unsigned grid = (unsigned)((one.n + 1) / 2);
    const float *t = static_cast<const float *>(tab);
    if (one.n > 0) {
        if (u >= 16) similarity_flat_kernel<16><<<grid, 256, 0, s>>>(one, t);
        else if (u >= 8) similarity_flat_kernel<8><<<grid, 256, 0, s>>>(one, t);
        else similarity_flat_kernel<4><<<grid, 256, 0, s>>>(one, t);
    }
    LaunchNote &note = launch_note();
    note.sim_kind = 1, note.lg_split = 0, note.lg_launches = 1, note.lg_fin = 1;
}
// m <= 1024 (sixteen waves x 64 rows of byte counters); every pointer but raw / gaps / indets / keep in device AND pinned host memory
void launch_overlap_small(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, uint8_t indet, const int32_t *gaps,
                          const int32_t *indets, int need, float min_ov, int32_t *good, int32_t *h_good, uint8_t *keep, uint8_t *h_keep,
                          int32_t *col_nongap, int32_t *h_col_nongap) {
    overlap_small_kernel<<<(unsigned)((m + 3) / 4), 256, 0, s>>>(raw, m, n, ld, 0x01010101u * indet, gaps, indets, need, min_ov, good, h_good,
                                                                  keep, h_keep);
    launch_col_nongap_small(s, raw, m, n, ld, keep, col_nongap, h_col_nongap);
}
void launch_col_nongap_small(hipStream_t s, const uint8_t *raw, int m, int n, int64_t ld, const uint8_t *keep_seq, int32_t *col_nongap,
                             int32_t *h_col_nongap) {
    col_nongap_small_kernel<<<(unsigned)((ld / 4 + 63) / 64), 1024, 0, s>>>(raw, m, n, ld, keep_seq, col_nongap, h_col_nongap);
}
void launch_sort_columns_batch(hipStream_t s, const BAlign *table, const LgAlign *lg, int K, int max_m) {
    if (K > 0) sort_columns_batch_kernel<<<(unsigned)K, 256, (size_t)(max_m + 2) * sizeof(int), s>>>(table, lg);
}
void launch_repitch_rows(hipStream_t s, const uint8_t *src, int64_t ld_src, uint8_t *dst, int64_t ld_dst, int m, int n) {
    const int64_t threads = (ld_dst >> 4) * (int64_t)m;
    if (threads > 0) repitch_rows_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(src, ld_src, dst, ld_dst, m, n);
}
void launch_compact_identity(hipStream_t s, const CompactArgs &a) {
    compact_identity_kernel<<<(unsigned)((a.m + 3) / 4), 256, 0, s>>>(a);
}

// mean weight of every row over its later partners (m + 64 floats): the similarity kernel's predictor reads it
void launch_w_row_means(hipStream_t s, const float *wup, int m, int ldw, float *wbar) {
    w_row_means_kernel<<<(m + 64 + 3) / 4, 256, 0, s>>>(wup, m, ldw, wbar);
}

void launch_identity_stats(hipStream_t s, const float *ident, int m, int ldw, float *row_avg, float *row_max, float *out2,
                           float *row_min, int *gate) {
    identity_rows_kernel<<<(m + 3) / 4, 256, 0, s>>>(ident, m, ldw, row_avg, row_max, row_min);
    identity_final_kernel<<<1, 128, 0, s>>>(row_avg, row_max, m, out2, gate);
}

}  // namespace msak
