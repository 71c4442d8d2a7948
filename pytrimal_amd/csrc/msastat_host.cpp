// msastat_host.cpp -- host-side selection logic of the trim path (product code).
//
// Mirrors what trimAl's Cleaner / statistics::Gaps / statistics::Similarity do with the
// statistic vectors once the kernels have produced them: cut points, column selection with
// recovery, strict/strictplus block logic, automated1 decision, representative clustering.
// All of it is O(n log n) or O(m * clusters) and stays on the CPU.  The reference declares
// these entry points in include/trimal/cleaner.pxd:16-46 and statistics.pxd:18-64; their
// bodies live in the un-vendored trimAl submodule, so the order of operations below follows
// SURVEY.md Appendix A (validated against the reference's surviving fixtures).
//
// Everything here is pure host code and is exported through the C ABI (include/msastat.h) so
// that it can be exercised without a device.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string_view>
#include <cctype>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "msastat.h"
#include "msastat_host.h"

namespace msah {

static inline int round_half_up(double x) { return static_cast<int>(x + 0.5); }

// mirror index used by both window functions (Gaps::applyWindow / Similarity::applyWindow)
static inline int mirror(int j, int n) { return j < 0 ? -j : (j >= n ? 2 * n - j - 2 : j); }

int window_i32(const int32_t *v, int n, int hw, int32_t *out) {
    if (hw > n / 4) return MSA_E_WINDOW_TOO_BIG;
    if (hw <= 0) {
        std::copy(v, v + n, out);
        return MSA_OK;
    }
    const int width = 2 * hw + 1;
    for (int c = 0; c < n; ++c) {
        int sum = 0;
        for (int d = -hw; d <= hw; ++d) sum += v[mirror(c + d, n)];
        out[c] = round_half_up(static_cast<double>(sum) / width);
    }
    return MSA_OK;
}

int window_f32(const float *v, int n, int hw, float *out) {
    if (hw > n / 4) return MSA_E_WINDOW_TOO_BIG;
    if (hw <= 0) {
        std::copy(v, v + n, out);
        return MSA_OK;
    }
    const float width = static_cast<float>(2 * hw + 1);
    for (int c = 0; c < n; ++c) {
        float sum = 0.0f;
        for (int d = -hw; d <= hw; ++d) sum += v[mirror(c + d, n)];
        out[c] = sum / width;
    }
    return MSA_OK;
}

GapHistogram::GapHistogram(const int32_t *gaps, int m_, int n_) : m(m_), n(n_), count(m_ + 2, 0), max_gaps(0) {
    for (int c = 0; c < n; ++c) {
        ++count[gaps[c]];
        max_gaps = std::max(max_gaps, gaps[c]);
    }
}

// Gaps::calcCutPoint
double GapHistogram::cut_point(float base_line, float gap_threshold) const {
    const double by_threshold = static_cast<double>(m) * gap_threshold;
    double wanted = round_half_up(static_cast<double>(n * base_line) / 100.0);
    if (wanted > n) wanted = n;
    int k = 0, seen = 0;
    for (; k < m; ++k) {
        seen += count[k];
        if (seen >= wanted) break;
    }
    double by_conservation = 0.0;
    if (count[k] != 0) by_conservation = static_cast<double>(k - (static_cast<float>(seen - wanted) / count[k]));
    return std::max(by_conservation, by_threshold);
}

// Gaps::calcCutPoint2ndSlope: walk triples of populated histogram bins; the running best ratio
// is held in an int (as upstream), so a candidate has to beat the truncated previous ratio.
int GapHistogram::cut_point_2nd_slope() const {
    std::vector<int> bins;  // populated gap counts, ascending
    for (int g = 0; g <= max_gaps; ++g)
        if (count[g]) bins.push_back(g);
    std::vector<float> slope(max_gaps + 2, -1.0f);
    int best = 0, best_ratio = 1;
    for (size_t t = 0; t + 2 < bins.size(); ++t) {
        const int first = bins[t], second = bins[t + 1], third = bins[t + 2];
        float s = static_cast<float>(third - first) / m;
        s /= static_cast<float>(count[third] + count[second]) / n;
        slope[third] = s;
        float reference = -1.0f;
        if (slope[first] != -1.0f) reference = slope[first];
        else if (slope[second] != -1.0f) reference = slope[second];
        if (reference != -1.0f && (s / reference) > best_ratio) {
            best_ratio = static_cast<int>(s / reference);
            best = first;
        }
    }
    return best;
}

// Similarity::calcCutPoint
double similarity_cut_point(const float *mdkw, int n, float base_line, float sim_threshold) {
    std::vector<float> sorted(mdkw, mdkw + n);
    std::sort(sorted.begin(), sorted.end());
    int at = static_cast<int>(static_cast<double>(n - 1) * (100.0 - base_line) / 100.0);
    at = std::min(std::max(at, 0), n - 1);  // upstream reads past the end when base_line is unset
    const double by_conservation = sorted[at];
    return by_conservation < sim_threshold ? by_conservation : sim_threshold;
}

// --------------------------------------------------------------------------------------------
// column selection with recovery (Cleaner::cleanByCutValue*)
// --------------------------------------------------------------------------------------------
namespace {

struct Recovery {
    const int32_t *gaps = nullptr;
    double gap_limit = 0;
    const float *sim = nullptr;
    float sim_limit = 0;
    bool admits(int c) const {
        if (gaps && sim) return sim[c] >= sim_limit || gaps[c] <= gap_limit;
        if (gaps) return gaps[c] <= gap_limit;
        return sim[c] >= sim_limit;
    }
};

// Re-add dropped columns next to sufficiently long kept blocks, sweeping outwards from the
// middle, with a minimum block size shrinking from 0.5 % of the alignment down to 0.
void recover(std::vector<uint8_t> &keep, int missing, const Recovery &rule) {
    const int n = static_cast<int>(keep.size());
    for (int block = round_half_up(0.005 * n); block >= 0 && missing > 0; --block) {
        int left = n / 2, right = left + 1;
        while ((left > 0 || right < n - 1) && missing > 0) {
            int p = left;
            while (p >= 0 && keep[p] && missing > 0) --p;
            if (left - p >= block)
                while (p >= 0 && !keep[p] && missing > 0) {
                    if (!rule.admits(p)) break;
                    keep[p] = 1;
                    --missing;
                    --p;
                }
            left = p;
            p = right;
            while (p < n && keep[p] && missing > 0) ++p;
            if (p - right >= block)
                while (p < n && !keep[p] && missing > 0) {
                    if (!rule.admits(p)) break;
                    keep[p] = 1;
                    --missing;
                    ++p;
                }
            right = p;
            --left;
            ++right;
        }
    }
}

int shortfall(float base_line, int kept, int n) {
    return round_half_up(((base_line / 100.0) - static_cast<float>(kept) / n) * n);
}

// value of the sorted vector at the base-line position.  The two flavours differ in where the
// float -> double promotion happens, exactly as upstream: gaps use (float)(n-1) * baseLine
// (a float product), similarity uses (float)(n-1) * (100.0 - baseLine) (a double product).
int32_t gap_value_at_baseline(const int32_t *v, int n, float base_line) {
    std::vector<int32_t> sorted(v, v + n);
    std::sort(sorted.begin(), sorted.end());
    return sorted[static_cast<int>(static_cast<float>(n - 1) * base_line / 100.0)];
}
float sim_value_at_baseline(const float *v, int n, float base_line) {
    std::vector<float> sorted(v, v + n);
    std::sort(sorted.begin(), sorted.end());
    return sorted[static_cast<int>(static_cast<float>(n - 1) * (100.0 - base_line) / 100.0)];
}

}  // namespace

void clean_gaps(const int32_t *gw, int n, double cut, float base_line, uint8_t *keep_out) {
    std::vector<uint8_t> keep(n);
    int kept = 0;
    for (int c = 0; c < n; ++c) kept += keep[c] = gw[c] <= cut;
    const int missing = shortfall(base_line, kept, n);
    if (missing > 0) {
        Recovery rule;
        rule.gaps = gw;
        rule.gap_limit = gap_value_at_baseline(gw, n, base_line);
        recover(keep, missing, rule);
    }
    std::copy(keep.begin(), keep.end(), keep_out);
}

void clean_similarity(const float *vw, int n, float cut, float base_line, uint8_t *keep_out) {
    std::vector<uint8_t> keep(n);
    int kept = 0;
    for (int c = 0; c < n; ++c) kept += keep[c] = vw[c] > cut;
    const int missing = shortfall(base_line, kept, n);
    if (missing > 0) {
        Recovery rule;
        rule.sim = vw;
        rule.sim_limit = sim_value_at_baseline(vw, n, base_line);
        recover(keep, missing, rule);
    }
    std::copy(keep.begin(), keep.end(), keep_out);
}

void clean_both(const int32_t *gw, const float *vw, int n, double cut_g, float cut_v, float base_line,
                uint8_t *keep_out) {
    std::vector<uint8_t> keep(n);
    int kept = 0;
    for (int c = 0; c < n; ++c) kept += keep[c] = (vw[c] > cut_v) && (gw[c] <= cut_g);
    const int missing = shortfall(base_line, kept, n);
    if (missing > 0) {
        Recovery rule;
        rule.gaps = gw;
        rule.gap_limit = gap_value_at_baseline(gw, n, base_line);
        rule.sim = vw;
        rule.sim_limit = sim_value_at_baseline(vw, n, base_line);
        recover(keep, missing, rule);
    }
    std::copy(keep.begin(), keep.end(), keep_out);
}

// --------------------------------------------------------------------------------------------
// strict / strictplus (Cleaner::cleanCombMethods + Cleaner::cleanStrict)
// --------------------------------------------------------------------------------------------
float comb_similarity_cut(const int32_t *gw, const float *mdkw, int n, int gap_cut) {
    std::vector<float> pool;
    pool.reserve(n);
    for (int c = 0; c < n; ++c)
        if (gw[c] <= gap_cut) pool.push_back(mdkw[c]);
    const int size = static_cast<int>(pool.size());
    // p20 / p80: the value of the LAST descending rank whose percentage (float division, as upstream) is
    // <= 20 / <= 80; 0 when no rank qualifies.  The percentage grows with the rank, so the two ranks
    // come from a scan over ranks alone and the values from two selections instead of a full sort.
    // (the percentage is monotone in the rank: start from the arithmetic guess and settle on the exact comparison)
    auto last_rank = [&](double limit) {
        auto pct = [&](int rank) { return (static_cast<float>(rank) / size) * 100.0; };
        int r = std::min(size, std::max(0, static_cast<int>(limit / 100.0 * size)));
        while (r < size && pct(r + 1) <= limit) ++r;
        while (r > 0 && pct(r) > limit) --r;
        return r;
    };
    const int r20 = size > 0 ? last_rank(20.0) : 0, r80 = size > 0 ? last_rank(80.0) : 0;
    // rank 1 = largest value.  Selection by a histogram over the leading bits of the (non-negative) floats -- their
    // bit patterns order like the values -- and a selection inside the one bucket that holds the rank: two linear
    // passes instead of two nth_element runs over the whole pool (negative or NaN values: plain nth_element).
    float p20 = 0.0f, p80 = 0.0f;
    bool plain = size < 2048;
    if (!plain) {
        constexpr int SHIFT = 18, NB = 1 << (31 - SHIFT);
        std::vector<uint32_t> hist(NB + 1, 0u);
        for (int i = 0; i < size && !plain; ++i) {
            uint32_t b;
            std::memcpy(&b, &pool[i], 4);
            if (b >> 31 || b > 0x7F800000u) plain = true;
            else ++hist[(b >> SHIFT) + 1];
        }
        if (!plain) {
            for (int i = 0; i < NB; ++i) hist[i + 1] += hist[i];  // hist[k] = elements in buckets below k
            std::vector<float> bucket;
            auto select = [&](int rank) -> float {  // the element at ascending position size - rank
                const uint32_t pos = static_cast<uint32_t>(size - rank);
                const int k = static_cast<int>(std::upper_bound(hist.begin(), hist.end(), pos) - hist.begin()) - 1;
                bucket.clear();
                for (int i = 0; i < size; ++i) {
                    uint32_t b;
                    std::memcpy(&b, &pool[i], 4);
                    if (static_cast<int>(b >> SHIFT) == k) bucket.push_back(pool[i]);
                }
                auto nth = bucket.begin() + (pos - hist[k]);
                std::nth_element(bucket.begin(), nth, bucket.end());
                return *nth;
            };
            if (r80 > 0) p80 = select(r80);
            if (r20 > 0) p20 = select(r20);
        }
    }
    if (plain) {  // the lower rank first, the higher one (r20 <= r80) then lies in the part above it
        auto from = pool.begin();
        if (r80 > 0) {
            auto nth = pool.begin() + (size - r80);
            std::nth_element(pool.begin(), nth, pool.end());
            p80 = *nth;
            from = nth;
        }
        if (r20 > 0) {
            auto nth = pool.begin() + (size - r20);
            std::nth_element(from, nth, pool.end());
            p20 = *nth;
        }
    }
    const double hi = std::log10(static_cast<double>(p20)), lo = std::log10(static_cast<double>(p80));
    return static_cast<float>(std::pow(10, ((hi - lo) / 10) + lo));
}

void clean_strict(const int32_t *gw, const float *mdkw, int n, int gap_cut, float sim_cut, bool variable,
                  uint8_t *keep) {
    // (two guard entries on either side, branch-free passes: the compiler vectorises them)
    std::vector<uint8_t> rej_store(static_cast<size_t>(n) + 4, 0);
    uint8_t *rejected = rej_store.data() + 2;
    for (int c = 0; c < n; ++c) rejected[c] = static_cast<uint8_t>((gw[c] > gap_cut) | (mdkw[c] < sim_cut));
    // rescue a rejected column when enough of its neighbours were accepted BEFORE any rescue
    for (int c = 2; c < n - 2; ++c) {
        const int around = rejected[c - 2] + rejected[c - 1] + rejected[c + 1] + rejected[c + 2];
        keep[c] = static_cast<uint8_t>(!rejected[c] | (around <= 1));
    }
    auto rej = [&](int c) { return static_cast<int>(rejected[c]); };
    for (int c : {0, 1, n - 2, n - 1})
        if (c >= 0 && c < n && (c < 2 || c >= n - 2)) keep[c] = !rejected[c];
    if (n > 2 && rejected[0]) keep[0] = (rej(1) + rej(2)) == 0;
    if (n > 3 && rejected[1]) keep[1] = (rej(0) + rej(2) + rej(3)) == 0;
    if (n >= 5) {
        if (rejected[n - 2]) keep[n - 2] = (rej(n - 4) + rej(n - 3) + rej(n - 1)) == 0;
        if (rejected[n - 1]) keep[n - 1] = (rej(n - 3) + rej(n - 2)) == 0;
    }
    int min_block = 5;
    if (variable) min_block = std::min(12, std::max(3, round_half_up(n * 0.01F)));
    // remove kept runs shorter than min_block.  As upstream, a run is examined when the dropped
    // column after it is met, so a run that reaches the last column always survives.
    int run = 0;
    for (int c = 0; c < n; ++c) {
        if (keep[c]) {
            ++run;
            continue;
        }
        if (run < min_block)
            for (int b = c - run; b < c; ++b) keep[b] = 0;
        run = 0;
    }
}

// Cleaner::selectMethod decision
int select_method(float avg_seq, float max_seq, int m) {
    if (avg_seq >= 0.55) return 1;
    if (avg_seq <= 0.38) return 2;
    if (m <= 20) return 1;
    if (max_seq >= 0.5 && max_seq <= 0.65) return 1;
    return 2;
}

// --------------------------------------------------------------------------------------------
// representative sequences
// --------------------------------------------------------------------------------------------
namespace {

struct LenIdx {
    int32_t len, idx;
};

// trimAl sorts (length, index) pairs with its own quicksort keyed on length only; the order
// it leaves equal lengths in is observable in the clustering below, so it is reproduced.
void length_sort(std::vector<LenIdx> &v, int lo, int hi) {
    if (lo >= hi || hi < 0) return;
    const int32_t pivot = v[hi].len;
    int i = lo - 1, j = hi;
    for (;;) {
        while (v[++i].len < pivot)
            if (i == hi) break;
        while (v[--j].len > pivot)
            if (j == 0) break;
        if (i >= j) break;
        std::swap(v[i], v[j]);
    }
    std::swap(v[i], v[hi]);
    length_sort(v, lo, i - 1);
    length_sort(v, i + 1, hi);
}

std::vector<LenIdx> by_length(const int32_t *lengths, int m) {
    std::vector<LenIdx> v(m);
    for (int i = 0; i < m; ++i) v[i] = {lengths[i], i};
    length_sort(v, 0, m - 1);
    return v;
}

}  // namespace

// processing order of calculateRepresentativeSeq: longest first, ties in trimAl's quicksort order
std::vector<int32_t> processing_order(const int32_t *lengths, int m) {
    const std::vector<LenIdx> order = by_length(lengths, m);
    std::vector<int32_t> seq_at(m);
    for (int t = 0; t < m; ++t) seq_at[t] = order[m - 1 - t].idx;
    return seq_at;
}

int representatives(const float *ident, int ldi, const int32_t *lengths, int m, float max_identity,
                    uint8_t *keep_seq) {
    const std::vector<LenIdx> order = by_length(lengths, m);
    std::vector<int> reps{order[m - 1].idx};
    for (int r = m - 2; r >= 0; --r) {
        const int s = order[r].idx;
        bool joins = false;
        float best = 0.0f;
        for (int rep : reps) {
            const float v = ident[static_cast<size_t>(s) * ldi + rep];
            if (v > max_identity && v > best) {
                best = v;
                joins = true;
            }
        }
        if (!joins) reps.push_back(s);
    }
    std::fill(keep_seq, keep_seq + m, 0);
    for (int rep : reps) keep_seq[rep] = 1;
    return static_cast<int>(reps.size());
}

// Cleaner::getCutPointClusters: bisection on the identity threshold, starting from the mean
// identity, giving up after the cluster count has stalled for more than 10 iterations.
float cutpoint_clusters(const float *ident, int ldi, const int32_t *lengths, int m, int clusters) {
    if (clusters == m) return 1;
    if (clusters == 1) return 0;
    float hi = 0, lo = 1, threshold = 0;
    for (int i = 0; i < m; ++i) {
        float row_max = 0, row_min = 1, row_sum = 0;
        for (int j = 0; j < m; ++j) {
            if (j == i) continue;
            const float v = ident[static_cast<size_t>(i) * ldi + j];
            if (row_max < v) row_max = v;
            if (row_min > v) row_min = v;
            row_sum += v;
        }
        threshold += row_sum / (m - 1);
        if (row_max > hi) hi = row_max;
        if (row_min < lo) lo = row_min;
    }
    threshold /= m;
    const std::vector<LenIdx> order = by_length(lengths, m);
    std::vector<int> reps(m);
    reps[0] = order[m - 1].idx;
    float previous = 0, stalled = 0;
    for (;;) {
        int count = 1;
        for (int r = m - 2; r >= 0; --r) {
            int j = 0;
            for (; j < count; ++j)
                if (ident[static_cast<size_t>(order[r].idx) * ldi + reps[j]] > threshold) break;
            if (j == count) reps[count++] = order[r].idx;
        }
        if (count == clusters || stalled > 10) break;
        if (count > clusters) hi = threshold;
        else lo = threshold;
        threshold = (hi + lo) / 2;
        if (previous != count) {
            stalled = 0;
            previous = static_cast<float>(count);
        } else ++stalled;
    }
    return threshold;
}

}  // namespace msah

// ---- C ABI ----------------------------------------------------------------------------------
// --------------------------------------------------------------------------------------------
// FASTA ingest (Alignment.load): text -> dense residue matrix
// --------------------------------------------------------------------------------------------
namespace {
inline bool fasta_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// Calls rec(start_of_name_line, end) at every header line and seq(ptr, end) for every other non-blank line
// that follows a header.  Lines are '\n'-separated; leading blanks of a line are skipped before the '>' test.
template <class Rec, class Seq>
void fasta_walk(const uint8_t *data, int64_t len, Rec &&rec, Seq &&seq) {
    const uint8_t *p = data, *end = data + len;
    bool in_record = false;
    while (p < end) {
        const uint8_t *eol = static_cast<const uint8_t *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!eol) eol = end;
        const uint8_t *q = p;
        while (q < eol && fasta_space(*q)) ++q;
        if (q < eol) {
            if (*q == '>') {
                rec(q + 1, eol);
                in_record = true;
            } else if (in_record) {
                seq(q, eol);
            }
        }
        p = eol + 1;
    }
}
}  // namespace

extern "C" int msa_fasta_scan(const uint8_t *data, int64_t len, int32_t *m_out, int32_t *n_out) {
    if (!data || len < 0 || !m_out || !n_out) return MSA_E_INVALID;
    int64_t m = 0, n0 = 0;
    fasta_walk(
        data, len, [&](const uint8_t *, const uint8_t *) { ++m; },
        [&](const uint8_t *a, const uint8_t *b) {
            if (m != 1) return;
            for (; a < b; ++a) n0 += !fasta_space(*a);
        });
    if (m > INT32_MAX || n0 > INT32_MAX) return MSA_E_INVALID;
    *m_out = static_cast<int32_t>(m);
    *n_out = static_cast<int32_t>(n0);
    return MSA_OK;
}

extern "C" int msa_fasta_fill(const uint8_t *data, int64_t len, int32_t m, int32_t n, uint8_t *matrix,
                              int64_t *name_off, int32_t *name_len, const uint8_t *valid, msa_err_detail *detail) {
    if (!data || len < 0 || m < 0 || n < 0 || (!matrix && (int64_t)m * n > 0) || !name_off || !name_len) return MSA_E_INVALID;
    int64_t row = -1, col = 0;
    int rc = MSA_OK;
    auto fail = [&](int code, int64_t r, int64_t c, int byte) {
        if (rc != MSA_OK) return;
        rc = code;
        if (detail) {
            detail->row = static_cast<int32_t>(r);
            detail->col = static_cast<int32_t>(c);
            detail->byte = byte;
        }
    };
    auto close_row = [&]() {
        if (row >= 0 && row < m && col != n) fail(MSA_E_LENGTH_MISMATCH, row, col, 0);
    };
    uint8_t cls[256];
    for (int ch = 0; ch < 256; ++ch) cls[ch] = fasta_space(static_cast<uint8_t>(ch)) ? 1 : ((valid && !valid[ch]) ? 2 : 0);
    fasta_walk(
        data, len,
        [&](const uint8_t *a, const uint8_t *b) {
            close_row();
            ++row;
            col = 0;
            if (row >= m) return;
            while (a < b && fasta_space(*a)) ++a;  // ">  name" keeps an empty name, as bytes.split() would not:
            const uint8_t *e = a;                   // the first field of the stripped line after '>'
            while (e < b && !fasta_space(*e)) ++e;
            name_off[row] = a - data;
            name_len[row] = static_cast<int32_t>(e - a);
        },
        [&](const uint8_t *a, const uint8_t *b) {
            if (row >= m || rc != MSA_OK) return;
            uint8_t *dst = matrix + static_cast<size_t>(row) * static_cast<size_t>(n);
            // fast path: the line fits the row; copy while classifying (0 residue, 1 blank, 2 not accepted)
            const int64_t room = n - col;
            if (b - a <= room) {
                uint8_t *out = dst + col;
                unsigned worst = 0;
                for (const uint8_t *q = a; q < b; ++q) {
                    const unsigned k = cls[*q];
                    *out = *q;
                    out += (k == 0);
                    worst |= k;
                }
                if (!(worst & 2)) {
                    col += out - (dst + col);
                    return;
                }
            }
            for (; a < b; ++a) {  // slow path: overlong row or a byte outside the accepted set
                const uint8_t ch = *a;
                if (cls[ch] == 1) continue;
                if (col < n) {
                    if (cls[ch] == 2) {
                        fail(MSA_E_BAD_RESIDUE, row, col, ch);
                        return;
                    }
                    dst[col] = ch;
                }
                ++col;
            }
        });
    close_row();
    if (rc == MSA_OK && row + 1 != m) return MSA_E_INVALID;
    return rc;
}

// --------------------------------------------------------------------------------------------
// Clustal ingest (Alignment.load(..., "clustal")): interleaved blocks -> dense residue matrix
// --------------------------------------------------------------------------------------------
namespace {
// Calls line(name_begin, name_end, res_begin, res_end) for every data line: not the header (first line), not
// blank, not a conservation line (one that starts with a blank); fields are separated by blanks, the first is
// the sequence name, the second its residues of this block, anything behind (a running count) is ignored.
// The header is the first line that is not blank; it must start with "CLUSTAL" or "MUSCLE" (either case), otherwise
// the text is not Clustal and the walk reports so (returns false) without calling `line`.
template <class Line>
bool clustal_walk(const uint8_t *data, int64_t len, Line &&line) {
    const uint8_t *p = data, *end = data + len;
    bool first = true;
    while (p < end) {
        const uint8_t *eol = static_cast<const uint8_t *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!eol) eol = end;
        const uint8_t *a = p;
        p = eol + 1;
        if (first) {
            const uint8_t *h = a;
            while (h < eol && fasta_space(*h)) ++h;
            if (h == eol) continue;  // blank lines in front of the header
            auto starts = [&](const char *word) {
                const size_t k = std::strlen(word);
                if (static_cast<size_t>(eol - h) < k) return false;
                for (size_t i = 0; i < k; ++i)
                    if (std::toupper(h[i]) != word[i]) return false;
                return true;
            };
            if (!starts("CLUSTAL") && !starts("MUSCLE")) return false;
            first = false;
            continue;
        }
        if (a == eol || fasta_space(*a)) continue;
        const uint8_t *ne = a;
        while (ne < eol && !fasta_space(*ne)) ++ne;
        const uint8_t *rb = ne;
        while (rb < eol && fasta_space(*rb)) ++rb;
        if (rb == eol) continue;  // a lone field: not a data line
        const uint8_t *re = rb;
        while (re < eol && !fasta_space(*re)) ++re;
        line(a, ne, rb, re);
    }
    return !first;  // (no header at all: not Clustal either)
}
inline bool same_name(const uint8_t *a, const uint8_t *ae, const uint8_t *b, int32_t blen) {
    return ae - a == blen && std::memcmp(a, b, static_cast<size_t>(blen)) == 0;
}
}  // namespace

extern "C" int msa_clustal_scan(const uint8_t *data, int64_t len, int32_t *m_out, int32_t *n_out) {
    if (!data || len < 0 || !m_out || !n_out) return MSA_E_INVALID;
    // the sequences are the names of the first block (it ends when any of its names comes back); the width is
    // what the first sequence collects over all blocks
    std::unordered_set<std::string_view> seen;
    const uint8_t *first = nullptr;
    int32_t first_len = 0;
    int64_t m = 0, n = 0;
    bool first_block = true;
    const bool is_clustal = clustal_walk(data, len, [&](const uint8_t *a, const uint8_t *ae, const uint8_t *rb, const uint8_t *re) {
        const std::string_view name(reinterpret_cast<const char *>(a), static_cast<size_t>(ae - a));
        if (!first) {
            first = a;
            first_len = static_cast<int32_t>(ae - a);
        }
        if (first_block) {
            if (seen.insert(name).second) ++m;
            else first_block = false;
        }
        if (same_name(a, ae, first, first_len)) n += re - rb;
    });
    if (!is_clustal) return MSA_E_INVALID;  // no "CLUSTAL ..." / "MUSCLE ..." header: the caller's own parser reports it
    if (m > INT32_MAX || n > INT32_MAX) return MSA_E_INVALID;
    *m_out = static_cast<int32_t>(m);
    *n_out = static_cast<int32_t>(n);
    return MSA_OK;
}

extern "C" int msa_clustal_fill(const uint8_t *data, int64_t len, int32_t m, int32_t n, uint8_t *matrix,
                                int64_t *name_off, int32_t *name_len, const uint8_t *valid, msa_err_detail *detail) {
    if (!data || len < 0 || m < 0 || n < 0 || (!matrix && (int64_t)m * n > 0) || !name_off || !name_len) return MSA_E_INVALID;
    std::vector<int64_t> col(static_cast<size_t>(m), 0);
    std::unordered_map<std::string_view, int32_t> rows;  // name -> row (the first block's names, as they appear)
    rows.reserve(static_cast<size_t>(m) * 2);
    int32_t known = 0, expect = 0;  // names seen so far; the row the next line should belong to (block order)
    int rc = MSA_OK;
    auto fail = [&](int code, int64_t r, int64_t c, int byte) {
        if (rc != MSA_OK) return;
        rc = code;
        if (detail) {
            detail->row = static_cast<int32_t>(r);
            detail->col = static_cast<int32_t>(c);
            detail->byte = byte;
        }
    };
    const bool is_clustal = clustal_walk(data, len, [&](const uint8_t *a, const uint8_t *ae, const uint8_t *rb, const uint8_t *re) {
        if (rc != MSA_OK) return;
        int32_t row = -1;
        if (expect < known && same_name(a, ae, data + name_off[expect], name_len[expect])) {
            row = expect;  // the usual case: every block lists the sequences in the same order
        } else {
            const std::string_view name(reinterpret_cast<const char *>(a), static_cast<size_t>(ae - a));
            const auto it = rows.find(name);
            if (it != rows.end()) row = it->second;
            if (row < 0) {
                if (known >= m) {  // a name the scan did not count (it appears after the first block only)
                    fail(MSA_E_INVALID, known, 0, 0);
                    return;
                }
                row = known++;
                name_off[row] = a - data;
                name_len[row] = static_cast<int32_t>(ae - a);
                rows.emplace(name, row);
            }
        }
        expect = row + 1 < m ? row + 1 : 0;
        uint8_t *dst = matrix + static_cast<size_t>(row) * static_cast<size_t>(n);
        for (const uint8_t *q = rb; q < re; ++q) {
            const int64_t c = col[row]++;
            if (c >= n) continue;  // reported as a length mismatch below
            if (valid && !valid[*q]) {
                fail(MSA_E_BAD_RESIDUE, row, c, *q);
                return;
            }
            dst[c] = *q;
        }
    });
    if (!is_clustal) return MSA_E_INVALID;
    if (rc != MSA_OK) return rc;
    if (known != m) return MSA_E_INVALID;
    for (int32_t r = 0; r < m; ++r)
        if (col[r] != n) {
            fail(MSA_E_LENGTH_MISMATCH, r, col[r], 0);
            return rc;
        }
    return MSA_OK;
}

extern "C" {

int msa_window_i32(const int32_t *v, int32_t n, int32_t hw, int32_t *out) { return msah::window_i32(v, n, hw, out); }
int msa_window_f32(const float *v, int32_t n, int32_t hw, float *out) { return msah::window_f32(v, n, hw, out); }

double msa_gaps_cutpoint(const int32_t *gaps, int32_t m, int32_t n, float base_line, float gap_threshold) {
    return msah::GapHistogram(gaps, m, n).cut_point(base_line, gap_threshold);
}
int32_t msa_gaps_cutpoint_2nd_slope(const int32_t *gaps, int32_t m, int32_t n) {
    return msah::GapHistogram(gaps, m, n).cut_point_2nd_slope();
}
double msa_similarity_cutpoint(const float *mdkw, int32_t n, float base_line, float sim_threshold) {
    return msah::similarity_cut_point(mdkw, n, base_line, sim_threshold);
}
int msa_clean_gaps(const int32_t *gw, int32_t n, double cut, float base_line, uint8_t *keep) {
    msah::clean_gaps(gw, n, cut, base_line, keep);
    return MSA_OK;
}
int msa_clean_similarity(const float *vw, int32_t n, float cut, float base_line, uint8_t *keep) {
    msah::clean_similarity(vw, n, cut, base_line, keep);
    return MSA_OK;
}
int msa_clean_both(const int32_t *gw, const float *vw, int32_t n, double cut_g, float cut_v, float base_line,
                   uint8_t *keep) {
    msah::clean_both(gw, vw, n, cut_g, cut_v, base_line, keep);
    return MSA_OK;
}
int msa_clean_strict(const int32_t *gaps, const int32_t *gw, const float *vw, int32_t m, int32_t n, int32_t variable,
                     uint8_t *keep, int32_t *gap_cut_out, float *sim_cut_out) {
    const int gap_cut = msah::GapHistogram(gaps, m, n).cut_point_2nd_slope();
    const float sim_cut = msah::comb_similarity_cut(gw, vw, n, gap_cut);
    msah::clean_strict(gw, vw, n, gap_cut, sim_cut, variable != 0, keep);
    if (gap_cut_out) *gap_cut_out = gap_cut;
    if (sim_cut_out) *sim_cut_out = sim_cut;
    return MSA_OK;
}
int32_t msa_select_method(float avg_seq, float max_seq, int32_t m) { return msah::select_method(avg_seq, max_seq, m); }
int msa_representatives(const float *ident, const int32_t *lengths, int32_t m, float max_identity, uint8_t *keep_seq,
                        int32_t *n_clusters) {
    const int k = msah::representatives(ident, m, lengths, m, max_identity, keep_seq);
    if (n_clusters) *n_clusters = k;
    return MSA_OK;
}
float msa_cutpoint_clusters(const float *ident, const int32_t *lengths, int32_t m, int32_t clusters) {
    return msah::cutpoint_clusters(ident, m, lengths, m, clusters);
}

}  // extern "C"
