/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See the header of each oracle .c file.
 * PARITY UNPINNED (no golden vectors exist in the reference; saiga submodule absent).
 */
#ifndef SNK_ORACLE_H
#define SNK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DIST_INF 256

typedef struct orc_knn2
{
    int32_t idx1, dist1, idx2, dist2;
} orc_knn2;

typedef struct orc_kp64
{
    double x, y;
    float angle;
    int32_t octave;
} orc_kp64;

/* ---- match_oracle.c ---- */
int orc_hamming(const uint64_t a[4], const uint64_t b[4]);
void orc_bf_knn2(const uint64_t (*q)[4], int nq, const uint64_t (*t)[4], int nt, orc_knn2* out, int threads);
int orc_bf_filter(const orc_knn2* knn, int nq, int threshold, float ratio, int32_t (*pairs)[2]);
int orc_stereo_match(const orc_kp64* left, const uint64_t (*dl)[4], int nl, const orc_kp64* right,
                     const uint64_t (*dr)[4], int nr, double bf, const float* level_scale, int relaxed,
                     float* right_points, float* depth);

#ifdef __cplusplus
}
#endif
#endif
