/*
 * msa_oracle_avx2.c -- AVX2 flavour of the two pairwise passes of the CPU restatement (TEST / BASELINE
 * INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline leg, never by the product).
 *
 * SURVEY.md section 8(d) asks for the CPU baseline in two flavours: scalar (msa_oracle.c, the reference's
 * platform=None) and SSE2/AVX2 intrinsics written from scratch to the same semantics (the reference links
 * trimAl's own SIMD object libraries, src/trimal/CMakeLists.txt:23-32, whose source is not in the tree).
 * Both functions are bit-identical to their scalar counterparts (tests/test_oracle_golden.py):
 *   orc_pair_counts_avx2  32 columns per step: byte compares, 0/-1 masks subtracted from byte counters,
 *                         _mm256_sad_epu8 every 255 steps (the horizontal-sum idiom CHANGELOG.md:211-214 names);
 *   orc_similarity_avx2   8 columns per vector, every lane carrying its own sequential float32 sums in the
 *                         reference's (j < k) order -- an invalid pair adds +0, which leaves a sum unchanged.
 * Build: -O3 -mavx2 -ffp-contract=off (oracle/Makefile).
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_INCORRECT_SYMBOL (-6)
#define ORC_E_UNDEFINED_SYMBOL (-7)

int orc_avx2_supported(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

static inline uint32_t hsum_sad(__m256i v) { /* sum of the four 64-bit lanes of a SAD result */
    __m128i s = _mm_add_epi64(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    return (uint32_t)(_mm_cvtsi128_si64(s) + _mm_extract_epi64(s, 1));
}

/* Cleaner::calculateSeqIdentity / Similarity::calculateMatrixIdentity integers (see msa_oracle.c). */
void orc_pair_counts_avx2(const uint8_t *a, int m, int n, int ld, uint8_t indet, uint32_t *hit, uint32_t *dst) {
    const int np = (n + 31) & ~31;
    uint8_t *raw = (uint8_t *)aligned_alloc(32, (size_t)m * np + 32);
    uint8_t *val = (uint8_t *)aligned_alloc(32, (size_t)m * np + 32);
    for (int i = 0; i < m; i++) {
        uint8_t *r = raw + (size_t)i * np, *v = val + (size_t)i * np;
        for (int c = 0; c < np; c++) {
            const uint8_t x = c < n ? a[(size_t)i * ld + c] : (uint8_t)'-';
            r[c] = x;
            v[c] = (x != '-' && x != indet) ? 0xFF : 0x00;
        }
    }
    const __m256i zero = _mm256_setzero_si256();
    for (int i = 0; i < m; i++) {
        hit[(size_t)i * m + i] = 0;
        dst[(size_t)i * m + i] = 0;
        const uint8_t *ri = raw + (size_t)i * np, *vi = val + (size_t)i * np;
        for (int j = i + 1; j < m; j++) {
            const uint8_t *rj = raw + (size_t)j * np, *vj = val + (size_t)j * np;
            uint32_t h = 0, d = 0;
            for (int c0 = 0; c0 < np; c0 += 32 * 255) {
                const int c1 = c0 + 32 * 255 < np ? c0 + 32 * 255 : np;
                __m256i hb = zero, db = zero; /* byte counters, at most 255 increments each */
                for (int c = c0; c < c1; c += 32) {
                    const __m256i xi = _mm256_load_si256((const __m256i *)(ri + c));
                    const __m256i xj = _mm256_load_si256((const __m256i *)(rj + c));
                    const __m256i cnt = _mm256_or_si256(_mm256_load_si256((const __m256i *)(vi + c)),
                                                        _mm256_load_si256((const __m256i *)(vj + c)));
                    db = _mm256_sub_epi8(db, cnt);
                    hb = _mm256_sub_epi8(hb, _mm256_and_si256(cnt, _mm256_cmpeq_epi8(xi, xj)));
                }
                h += hsum_sad(_mm256_sad_epu8(hb, zero));
                d += hsum_sad(_mm256_sad_epu8(db, zero));
            }
            hit[(size_t)i * m + j] = hit[(size_t)j * m + i] = h;
            dst[(size_t)i * m + j] = dst[(size_t)j * m + i] = d;
        }
    }
    free(raw);
    free(val);
}

/* Similarity::calculateVectors (see orc_similarity in msa_oracle.c): same arguments, same results. */
int orc_similarity_avx2(const uint8_t *a, int m, int n, int ld, uint8_t indet, const float *w, const int32_t *gaps_w,
                        const int32_t *vhash, const float *dist, int npos, float *mdk, float *q_out, int32_t *err) {
    /* residue -> matrix index per (row, column); -1 = skipped.  Errors in scan order of the scalar code:
     * column by column (columns cut by the gap rule are not examined), row by row. */
    int32_t *code = (int32_t *)aligned_alloc(32, sizeof(int32_t) * ((size_t)m * 8 + 8));
    float *dpad = (float *)calloc((size_t)(npos + 1) * (npos + 1), sizeof(float)); /* row / column npos: zeros */
    for (int x = 0; x < npos; x++)
        for (int y = 0; y < npos; y++) dpad[x * (npos + 1) + y] = dist[x * npos + y];
    const int np1 = npos + 1;
    for (int c0 = 0; c0 < n; c0 += 8) {
        int live[8];
        for (int l = 0; l < 8; l++) {
            const int c = c0 + l;
            live[l] = 0;
            if (c >= n) continue;
            if (q_out) q_out[c] = 0.0f;
            mdk[c] = 0.0f;
            if (gaps_w && ((float)gaps_w[c] / m) >= 0.8f) continue;
            live[l] = 1;
        }
        for (int l = 0; l < 8; l++) {
            const int c = c0 + l;
            for (int j = 0; j < m; j++) {
                int32_t v = npos; /* skipped: the zero row of dpad */
                if (live[l]) {
                    const uint8_t x = a[(size_t)j * ld + c];
                    if (!(x == '-' || x == indet)) {
                        const int up = (x >= 'a' && x <= 'z') ? x - 32 : x;
                        if (up < 'A' || up > 'Z' || vhash[up - 'A'] == -1) {
                            if (err) { err[0] = j; err[1] = c; err[2] = x; }
                            free(code);
                            free(dpad);
                            return (up < 'A' || up > 'Z') ? ORC_E_INCORRECT_SYMBOL : ORC_E_UNDEFINED_SYMBOL;
                        }
                        v = vhash[up - 'A'];
                    }
                }
                code[(size_t)j * 8 + l] = v;
            }
        }
        __m256 num = _mm256_setzero_ps(), den = _mm256_setzero_ps();
        const __m256i vnpos = _mm256_set1_epi32(npos), vnp1 = _mm256_set1_epi32(np1);
        for (int j = 0; j < m; j++) {
            const __m256i cj = _mm256_load_si256((const __m256i *)(code + (size_t)j * 8));
            const __m256i skipj = _mm256_cmpeq_epi32(cj, vnpos);
            if (_mm256_movemask_epi8(skipj) == -1) continue; /* no lane holds a residue in this row */
            const __m256i rowbase = _mm256_mullo_epi32(cj, vnp1);
            const float *wj = w + (size_t)j * m;
            for (int k = j + 1; k < m; k++) {
                const __m256i ck = _mm256_load_si256((const __m256i *)(code + (size_t)k * 8));
                /* both valid <=> neither index is npos; dpad is zero in row / column npos, so the product of a
                 * skipped pair is +0; the weight is masked explicitly for the denominator */
                const __m256i bad = _mm256_or_si256(skipj, _mm256_cmpeq_epi32(ck, vnpos));
                const __m256 d = _mm256_i32gather_ps(dpad, _mm256_add_epi32(rowbase, ck), 4);
                const __m256 wv = _mm256_set1_ps(wj[k]);
                num = _mm256_add_ps(num, _mm256_mul_ps(wv, d));
                den = _mm256_add_ps(den, _mm256_andnot_ps(_mm256_castsi256_ps(bad), wv));
            }
        }
        float nu[8], de[8];
        _mm256_storeu_ps(nu, num);
        _mm256_storeu_ps(de, den);
        for (int l = 0; l < 8; l++) {
            const int c = c0 + l;
            if (c >= n || !live[l] || de[l] == 0) continue;
            const float q = nu[l] / de[l];
            if (q_out) q_out[c] = q;
            const float v = (float)exp(-(double)q);
            mdk[c] = v > 1.0f ? 1.0f : v;
        }
    }
    free(code);
    free(dpad);
    return ORC_OK;
}
