// Internal C++ declarations of the host selection logic (msastat_host.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace msah {

int window_i32(const int32_t *v, int n, int hw, int32_t *out);
int window_f32(const float *v, int n, int hw, float *out);

struct GapHistogram {
    int m, n;
    std::vector<int> count;  // count[g] = columns with g gaps (m + 2 entries)
    int max_gaps;
    GapHistogram(const int32_t *gaps, int m, int n);
    double cut_point(float base_line, float gap_threshold) const;
    int cut_point_2nd_slope() const;
};

double similarity_cut_point(const float *mdkw, int n, float base_line, float sim_threshold);
void clean_gaps(const int32_t *gw, int n, double cut, float base_line, uint8_t *keep);
void clean_similarity(const float *vw, int n, float cut, float base_line, uint8_t *keep);
void clean_both(const int32_t *gw, const float *vw, int n, double cut_g, float cut_v, float base_line, uint8_t *keep);
float comb_similarity_cut(const int32_t *gw, const float *mdkw, int n, int gap_cut);
void clean_strict(const int32_t *gw, const float *mdkw, int n, int gap_cut, float sim_cut, bool variable,
                  uint8_t *keep);
int select_method(float avg_seq, float max_seq, int m);
std::vector<int32_t> processing_order(const int32_t *lengths, int m);
int representatives(const float *ident, int ldi, const int32_t *lengths, int m, float max_identity,
                    uint8_t *keep_seq);
float cutpoint_clusters(const float *ident, int ldi, const int32_t *lengths, int m, int clusters);

}  // namespace msah
