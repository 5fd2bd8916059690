/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the descriptor-matching part of Snake-SLAM's hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors, and the arithmetic of
 * distance()/BruteForceMatcher lives in the absent, unpinned submodule darglein/saiga
 * (reference .gitmodules:1-3).  Functions whose algorithm is fully visible in the reference
 * follow it line by line and cite it; the saiga-side ones restate the published ORB-SLAM2 /
 * textbook definition and say so.
 *
 * Plain C, serial (optionally OpenMP over queries for the cpu_baseline timing, mirroring the
 * reference's matchKnn2_omp(num_tracking_threads) — Snake/Tracking/TrackingCoarse.cpp:351).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "snk_oracle.h"

/* Saiga::distance(DescriptorORB, DescriptorORB) -> int  (absent; 25 call sites, e.g.
 * Snake/Preprocess/Preprocess.cpp:192).  Definition: popcount of the xor over 256 bits. */
int orc_hamming(const uint64_t a[4], const uint64_t b[4])
{
    return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
           __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
}

/* BruteForceMatcher<DescriptorORB>::matchKnn2 (absent saiga; call site
 * Snake/Tracking/TrackingCoarse.cpp:350-351).  Sequential scan in ascending train index with
 * strict '<' (same update rule the in-repo matchers use, Preprocess.cpp:193-202). */
void orc_bf_knn2(const uint64_t (*q)[4], int nq, const uint64_t (*t)[4], int nt, orc_knn2* out, int threads)
{
#pragma omp parallel for num_threads(threads) if (threads > 1)
    for (int i = 0; i < nq; ++i)
    {
        int best = ORC_DIST_INF, second = ORC_DIST_INF, bi = -1, si = -1;
        for (int j = 0; j < nt; ++j)
        {
            int d = orc_hamming(q[i], t[j]);
            if (d < best)
            {
                second = best;
                si     = bi;
                best   = d;
                bi     = j;
            }
            else if (d < second)
            {
                second = d;
                si     = j;
            }
        }
        out[i].idx1  = bi;
        out[i].dist1 = best;
        out[i].idx2  = si;
        out[i].dist2 = second;
    }
}

/* BruteForceMatcher::filterMatches(threshold, ratio) (absent saiga; call site
 * Snake/Tracking/TrackingCoarse.cpp:352; `matches` consumed at :373-387 as (query, train)).
 * Definition chosen (operator strictness unknown in saiga): keep when dist1 <= threshold and
 * dist1 <= ratio * dist2, float arithmetic. */
/* Mirror of the library's snk_set_definition (include/snake_hip.h): the [DEFINED] comparison / rounding rules that a reader
 * of saiga may have to flip.  Same keys, same values, same defaults. */
static int g_def_th_strict = 0, g_def_ratio_strict = 0, g_def_iround = 0;
int orc_set_definition(const char* key, int value)
{
    if (key && strcmp(key, "bf_filter.threshold_strict") == 0 && (value == 0 || value == 1)) { g_def_th_strict = value; return 0; }
    if (key && strcmp(key, "bf_filter.ratio_strict") == 0 && (value == 0 || value == 1)) { g_def_ratio_strict = value; return 0; }
    if (key && strcmp(key, "iround.mode") == 0 && value >= 0 && value <= 2) { g_def_iround = value; return 0; }
    if (key && strcmp(key, "orb.response") == 0) return orc_orb_set_response(value);
    return 1;
}

int orc_bf_filter(const orc_knn2* knn, int nq, int threshold, float ratio, int32_t (*pairs)[2])
{
    int n = 0;
    for (int i = 0; i < nq; ++i)
    {
        if (knn[i].idx1 < 0) continue;
        const float rd2 = ratio * (float)knn[i].dist2;
        if (g_def_th_strict ? knn[i].dist1 >= threshold : knn[i].dist1 > threshold) continue;
        if (g_def_ratio_strict ? (float)knn[i].dist1 >= rd2 : (float)knn[i].dist1 > rd2) continue;
        pairs[n][0] = i;
        pairs[n][1] = knn[i].idx1;
        ++n;
    }
    return n;
}

/* Saiga::iRound (absent).  Definition chosen: floor(x + 0.5); "iround.mode" 1 = half away from zero, 2 = half to even. */
static int orc_iround(double x)
{
    if (g_def_iround == 0) return (int)floor(x + 0.5);
    if (g_def_iround == 1) return (int)(x < 0.0 ? -floor(0.5 - x) : floor(x + 0.5));
    return (int)rint(x);
}

/* Snake::Preprocess::StereoMatching — Snake/Preprocess/Preprocess.cpp:122-242, line by line.
 * `left`/`right` are already rectified (the loops at :140-150 are the caller's job, see
 * include/snake_hip.h).  Returns the number of matches. */
int orc_stereo_match(const orc_kp64* left, const uint64_t (*dl)[4], int nl, const orc_kp64* right,
                     const uint64_t (*dr)[4], int nr, double bf, const float* level_scale, int relaxed,
                     float* right_points, float* depth)
{
    if (nl <= 0 || nr <= 0) return 0;
    int min_y = 1023123; /* :132 */
    int max_y = -19284;  /* :133 */

    const float min_disp = 0;                 /* :137 */
    const float max_disp = (float)(bf * 0.5); /* :138 */

    for (int i = 0; i < nr; ++i) /* :145-150 */
    {
        int y = orc_iround(right[i].y);
        if (y < min_y) min_y = y;
        if (y > max_y) max_y = y;
    }

    /* :152-158 row_map: per rounded row, the right indices in insertion (= index) order.
     * Restated as a counting sort (stable), which yields the same per-row order. */
    int rows       = max_y - min_y + 1;
    int* row_start = (int*)calloc((size_t)rows + 1, sizeof(int));
    int* row_items = (int*)malloc((size_t)nr * sizeof(int));
    for (int i = 0; i < nr; ++i) row_start[orc_iround(right[i].y) - min_y + 1]++;
    for (int r = 0; r < rows; ++r) row_start[r + 1] += row_start[r];
    {
        int* fill = (int*)malloc((size_t)rows * sizeof(int));
        memcpy(fill, row_start, (size_t)rows * sizeof(int));
        for (int i = 0; i < nr; ++i) row_items[fill[orc_iround(right[i].y) - min_y]++] = i;
        free(fill);
    }

    int num_matches = 0; /* :160 */
    for (int i = 0; i < nl; ++i)
    {
        int y        = orc_iround(left[i].y); /* :165 */
        int y_center = y - min_y;             /* :166 */

        float r = ceilf(2.0f * level_scale[left[i].octave]); /* :169 */

        int best_id          = -1;  /* :171 */
        int best_dist        = 250; /* :172 */
        int second_best_dist = 250; /* :173 */

        for (int row = (int)((float)y_center - r); (float)row <= (float)y_center + r; ++row) /* :175 */
        {
            if (row < 0 || row >= rows) continue; /* :177 */
            for (int k = row_start[row]; k < row_start[row + 1]; ++k)
            {
                int other_id     = row_items[k];
                double disparity = left[i].x - right[other_id].x;                          /* :181 */
                if (disparity < (double)min_disp || disparity > (double)max_disp) continue; /* :182-185 */
                if (abs(left[i].octave - right[other_id].octave) > 1) continue;            /* :187-190 */
                int dist = orc_hamming(dl[i], dr[other_id]);                               /* :192 */
                if (dist < best_dist)                                                      /* :193-198 */
                {
                    second_best_dist = best_dist;
                    best_dist        = dist;
                    best_id          = other_id;
                }
                else if (dist < second_best_dist) /* :199-202 */
                {
                    second_best_dist = dist;
                }
            }
        }

        if (best_dist > (relaxed ? 75 : 40)) continue;                                     /* :207 */
        if ((double)best_dist > (relaxed ? 0.9 : 0.7) * (double)second_best_dist) continue; /* :209-212 */

        float angle1 = left[i].angle; /* :214 */
        float angle2 = right[best_id].angle;
        float rot    = fminf(fabsf(angle1 - angle2),
                             fminf(fabsf((angle1 + 365) - angle2), fabsf(angle1 - (angle2 + 365)))); /* :216-217 */
        if (rot > (relaxed ? 25 : 5)) continue;                                                   /* :219-222 */

        double right_point = right[best_id].x;        /* :225 */
        double disparity   = left[i].x - right_point; /* :227 */
        if (disparity <= 0.001)                       /* :229-233 */
        {
            disparity   = 0.001;
            right_point = left[i].x - disparity;
        }
        right_points[i] = (float)right_point;      /* :235 */
        depth[i]        = (float)(bf / disparity); /* :236 */
        num_matches++;
    }
    free(row_start);
    free(row_items);
    return num_matches;
}
