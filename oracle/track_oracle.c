/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of
 *   Preprocess::computeFeatureGrid + FeatureGrid2::create   (reference Snake/Preprocess/Preprocess.cpp:244-266)
 *   Features::GetFeaturesInArea*                              (reference Snake/Map/Features.cpp:13-76)
 *   SnakeORBMatcher::SearchByProjectionFrameFrame2 (coarse)   (reference Snake/Tracking/SnakeORBMatcher.cpp:191-354)
 *   SnakeORBMatcher::SearchByProjection2 (fine)               (reference Snake/Tracking/SnakeORBMatcher.cpp:365-526)
 *   SnakeORBMatcher::SearchByProjectionFrameToKeyframe        (reference Snake/Tracking/SnakeORBMatcher.cpp:71-188)
 * following the reference line by line.  PARITY UNPINNED for the saiga-side helpers they call
 * (FeatureGrid2, FeatureGridBounds2, ScalePyramid): those are [DEFINED] here:
 *   grid    20-px cells over [min,max) of the undistorted image; cell = clamp(floor((p-min)/20));
 *           cells are ordered x-major (id = cx*rows + cy, the order Features.cpp:17-21 walks them),
 *           features inside a cell keep their original order; inImage(p) = min <= p < max;
 *   pyramid Scale(l) = level_scale[l]; PredictScaleLevel = ref_level + log(ref_depth/dist)/log(f)
 *           clamped to [0, L-1]; ScaleForContiniousLevel(p) = exp(p log f); PredictionConsistent =
 *           |p - octave| <= 1; EstimateMinMaxDistance = [0.8 d s(l)/s(L-1), 1.2 d s(l)];
 *           log / exp are evaluated with fixed-order fp64 series (orc_det_log / orc_det_exp) so the
 *           GPU reproduces them bit for bit;
 *   camera  K.project3(p) = (fx x/z + cx, fy y/z + cy, z); LeftPointToRight(x, z) = x - bf/z.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "snk_oracle.h"

/* ---------- deterministic log / exp (fp64, +,-,*,/ only) ---------- */
double orc_det_log(double x)
{
    int e;
    double m = frexp(x, &e); /* m in [0.5, 1) */
    if (m < 0.70710678118654752440)
    {
        m *= 2.0;
        e -= 1;
    }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
    double s = 1.0 / 21.0;
    for (int k = 19; k >= 1; k -= 2) s = s * z2 + 1.0 / (double)k;
    return 2.0 * z * s + (double)e * 0.69314718055994530942;
}

double orc_det_exp(double y)
{
    const double k = floor(y * 1.44269504088896340736 + 0.5);
    const double f = y - k * 0.69314718055994530942;
    double s = 1.0;
    for (int n = 16; n >= 1; --n) s = 1.0 + s * f / (double)n;
    return ldexp(s, (int)k);
}

/* ---------- grid ---------- */
static int grid_dim(double lo, double hi)
{
    int n = (int)ceil((hi - lo) / 20.0);
    return n < 1 ? 1 : n;
}
static int cell_coord(double p, double lo, int n)
{
    int c = (int)floor((p - lo) / 20.0);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

void orc_grid_dims(const orc_grid_bounds* b, int* cols, int* rows)
{
    *cols = grid_dim(b->min_x, b->max_x);
    *rows = grid_dim(b->min_y, b->max_y);
}

/* FeatureGrid2::create (absent saiga) as used at Preprocess.cpp:246: perm[i] = new index of old
 * feature i; cell_start has cols*rows + 1 entries (x-major cell ids). */
void orc_feature_grid(const orc_kp64* kps, int n, const orc_grid_bounds* b, int32_t* perm, int32_t* cell_start)
{
    int cols, rows;
    orc_grid_dims(b, &cols, &rows);
    const int nc = cols * rows;
    memset(cell_start, 0, sizeof(int32_t) * (size_t)(nc + 1));
    int* cell = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i)
    {
        cell[i] = cell_coord(kps[i].x, b->min_x, cols) * rows + cell_coord(kps[i].y, b->min_y, rows);
        cell_start[cell[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) cell_start[c + 1] += cell_start[c];
    int* fill = (int*)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
    memcpy(fill, cell_start, sizeof(int) * (size_t)nc);
    for (int i = 0; i < n; ++i) perm[i] = fill[cell[i]]++;
    free(cell);
    free(fill);
}

static int in_image(const orc_grid_bounds* b, double x, double y)
{
    return x >= b->min_x && x < b->max_x && y >= b->min_y && y < b->max_y;
}

/* ---------- camera / pose helpers ---------- */
static void quat_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

typedef struct view_ctx
{
    double R[9], t[3], campos[3];
} view_ctx;

static void make_ctx(const double* pose, view_ctx* c)
{
    quat_R(pose, c->R);
    c->t[0] = pose[4];
    c->t[1] = pose[5];
    c->t[2] = pose[6];
    /* currentPose.inverse().translation() = -R^T t */
    c->campos[0] = -(c->R[0] * c->t[0] + c->R[3] * c->t[1] + c->R[6] * c->t[2]);
    c->campos[1] = -(c->R[1] * c->t[0] + c->R[4] * c->t[1] + c->R[7] * c->t[2]);
    c->campos[2] = -(c->R[2] * c->t[0] + c->R[5] * c->t[1] + c->R[8] * c->t[2]);
}

static void transform(const view_ctx* c, const double* p, double* o)
{
    o[0] = c->R[0] * p[0] + c->R[1] * p[1] + c->R[2] * p[2] + c->t[0];
    o[1] = c->R[3] * p[0] + c->R[4] * p[1] + c->R[5] * p[2] + c->t[1];
    o[2] = c->R[6] * p[0] + c->R[7] * p[1] + c->R[8] * p[2] + c->t[2];
}

/* Features.cpp:13-76: candidate enumeration, cells x-major then y, members in cell order.
 * mode 0: radius only (float r, :13-29); mode 1: octave window (:55-76); mode 2: predicted scale (:31-52). */
typedef int (*cand_fn)(void* ud, int pid);
static void for_candidates(const orc_frame_view* f, double px, double py, double r, double r2, int mode, int min_oct,
                           int max_oct, double pred, cand_fn fn, void* ud)
{
    const int cols = f->cols, rows = f->rows;
    const int cx0 = cell_coord(px - r, f->bounds.min_x, cols), cx1 = cell_coord(px + r, f->bounds.min_x, cols);
    const int cy0 = cell_coord(py - r, f->bounds.min_y, rows), cy1 = cell_coord(py + r, f->bounds.min_y, rows);
    for (int cx = cx0; cx <= cx1; ++cx)
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int pid = f->cell_start[cx * rows + cy]; pid < f->cell_start[cx * rows + cy + 1]; ++pid)
            {
                const orc_kp64* kp = &f->kps[pid];
                if (mode == 1 && (kp->octave < min_oct || kp->octave > max_oct)) continue;
                if (mode == 2 && fabs(pred - (double)kp->octave) > 1.0) continue;
                const double dx = kp->x - px, dy = kp->y - py;
                if (dx * dx + dy * dy < r2) fn(ud, pid);
            }
}

/* ---------- ComputeThreeMaxima, SnakeORBMatcher.cpp:27-68 ---------- */
static void three_maxima(const int* histo, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++)
    {
        const int s = histo[i];
        if (s > max1)
        {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        }
        else if (s > max2)
        {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        }
        else if (s > max3)
        {
            max3 = s; *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1)
    {
        *ind2 = -1;
        *ind3 = -1;
    }
    else if ((float)max3 < 0.1f * (float)max1)
    {
        *ind3 = -1;
    }
}

/* ---------- coarse: SearchByProjectionFrameFrame2, SnakeORBMatcher.cpp:191-354 ---------- */
typedef struct coarse_ud
{
    const orc_frame_view* f;
    const uint8_t* taken;
    const uint64_t* desc1;
    double ipx, z, bf, r;
    int best_dist, best_idx;
} coarse_ud;

static int coarse_cand(void* p, int i2)
{
    coarse_ud* u = (coarse_ud*)p;
    if (u->taken[i2]) return 0; /* :284 */
    if (u->f->right_points[i2] > 0) /* :286-295 */
    {
        const double disp = u->ipx - u->bf / u->z;
        const double er   = fabs(disp - (double)u->f->right_points[i2]);
        if (er > u->r * 0.5) return 0;
    }
    const int dist = orc_hamming(u->desc1, u->f->desc[i2]); /* :298 */
    if (dist < u->best_dist)
    {
        u->best_dist = dist;
        u->best_idx  = i2;
    }
    return 0;
}

/* The reference runs the per-point phase of both tracking matchers under `#pragma omp parallel for num_threads(threads)`
 * (SnakeORBMatcher.cpp:219, :379; threads = num_tracking_threads = 4, Settings.h:88).  Every iteration writes only its own
 * best[i] / bins[i] / visible[i] / valid, so the result does not depend on the thread count; bench.py's cpu_baseline sets 4. */
static int g_match_threads = 1;
void orc_set_match_threads(int n) { g_match_threads = n < 1 ? 1 : n; }

int orc_match_coarse(const orc_frame_view* f, const orc_camera* cam, const double* pose, const orc_lm_coarse* pts, int m,
                     float th, int feature_error, int direction, const float* level_scale, int n_levels, int32_t* match_idx)
{
    view_ctx c;
    make_ctx(pose, &c);
    int* bins         = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int* best         = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    const float factor = 1.0f / 30; /* :203 */
    (void)n_levels;
#pragma omp parallel for num_threads(g_match_threads) schedule(static)
    for (int i = 0; i < m; ++i) /* :221 */
    {
        best[i] = -1; /* :223 */
        const orc_lm_coarse* lmp = &pts[i];
        double pc[3];
        transform(&c, lmp->pos, pc);
        const double z   = pc[2];
        const double ipx = cam->fx * pc[0] / z + cam->cx, ipy = cam->fy * pc[1] / z + cam->cy; /* :229 */
        if (z <= 0) continue;                                                                  /* :234 */
        if (!in_image(&f->bounds, ipx, ipy)) continue;                                         /* :236 */
        const double PO[3] = {c.campos[0] - lmp->pos[0], c.campos[1] - lmp->pos[1], c.campos[2] - lmp->pos[2]};
        const double dist  = sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]); /* :240 */
        const double viewCos = (PO[0] * lmp->normal[0] + PO[1] * lmp->normal[1] + PO[2] * lmp->normal[2]) / dist; /* :244 */
        if (viewCos < 0.5) continue;                                                                              /* :246 */
        const int lvl = lmp->octave; /* :256 */
        float r       = th;          /* :259 */
        r *= level_scale[lvl];       /* :260 */
        int mn, mx;
        if (direction == 1) { mn = lvl - 1; mx = 100; }        /* :262-265 */
        else if (direction == 2) { mn = 0; mx = lvl; }          /* :266-269 */
        else { mn = lvl - 1; mx = lvl + 1; }                    /* :270-273 */
        coarse_ud u;
        u.f = f; u.taken = f->taken; u.desc1 = lmp->desc; u.ipx = ipx; u.z = z; u.bf = cam->bf; u.r = (double)r;
        u.best_dist = 256; u.best_idx = -1; /* :280-281 */
        for_candidates(f, ipx, ipy, (double)r, (double)r * (double)r, 1, mn, mx, 0.0, coarse_cand, &u);
        if (u.best_dist <= feature_error) /* :307 */
        {
            float rot = lmp->angle - f->kps[u.best_idx].angle; /* :311 */
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor); /* :313 */
            if (bin == 30) bin = 0;
            if (!(bin >= 0 && bin < 30)) continue; /* :316 asserts; angles outside [0,360) / NaN: unmatched */
            best[i] = u.best_idx;
            bins[i] = bin;
        }
    }
    /* serial resolve, first claimant wins (:321-332) */
    uint8_t* taken2 = (uint8_t*)malloc((size_t)(f->n > 0 ? f->n : 1));
    memcpy(taken2, f->taken, (size_t)f->n);
    int hist[30];
    memset(hist, 0, sizeof(hist));
    int matches = 0;
    for (int i = 0; i < m; ++i)
    {
        match_idx[i] = -1;
        if (best[i] == -1) continue;
        if (taken2[best[i]]) continue;
        taken2[best[i]] = 1;
        match_idx[i]    = best[i];
        hist[bins[i]]++;
        matches++;
    }
    /* rotation consistency (:334-351) */
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hist, 30, &ind1, &ind2, &ind3);
    for (int i = 0; i < m; ++i)
    {
        if (match_idx[i] < 0) continue;
        const int b = bins[i];
        if (b != ind1 && b != ind2 && b != ind3)
        {
            match_idx[i] = -1;
            matches--;
        }
    }
    free(bins);
    free(best);
    free(taken2);
    return matches;
}

/* ---------- fine: SearchByProjection2, SnakeORBMatcher.cpp:365-526 ---------- */
typedef struct fine_ud
{
    const orc_frame_view* f;
    const uint64_t* desc1;
    double ipx, z, bf, r;
    int best_dist, best_level, best_dist2, best_level2, best_idx;
} fine_ud;

static int fine_cand(void* p, int idx)
{
    fine_ud* u = (fine_ud*)p;
    if (u->f->taken[idx]) return 0; /* :473 */
    if (u->f->right_points[idx] > 0) /* :477-485 */
    {
        const double er = fabs((u->ipx - u->bf / u->z) - (double)u->f->right_points[idx]);
        if (er > u->r * 0.5) return 0;
    }
    const int dist = orc_hamming(u->desc1, u->f->desc[idx]); /* :490 */
    if (dist < u->best_dist) /* :492-499 */
    {
        u->best_dist2  = u->best_dist;
        u->best_dist   = dist;
        u->best_level2 = u->best_level;
        u->best_level  = u->f->kps[idx].octave;
        u->best_idx    = idx;
    }
    else if (dist < u->best_dist2) /* :500-504 */
    {
        u->best_level2 = u->f->kps[idx].octave;
        u->best_dist2  = dist;
    }
    return 0;
}

int orc_match_fine(const orc_frame_view* f, const orc_camera* cam, const double* pose, orc_lm_fine* pts, int m, float th,
                   float ratio, const float* level_scale, int n_levels, int32_t* match_idx, uint8_t* visible)
{
    view_ctx c;
    make_ctx(pose, &c);
    const int bFactor      = th != 1.0f; /* :374 */
    const double log_f     = orc_det_log(n_levels > 1 ? (double)level_scale[1] / (double)level_scale[0] : 1.2);
    const double s_last    = (double)level_scale[n_levels - 1];
    int* best              = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
#pragma omp parallel for num_threads(g_match_threads) schedule(static)
    for (int i = 0; i < m; ++i) /* :381 */
    {
        best[i]    = -1;
        visible[i] = 0;
        orc_lm_fine* lmp = &pts[i];
        if (!lmp->valid) continue; /* :385 */
        double pc[3];
        transform(&c, lmp->pos, pc);
        const double z = pc[2];
        if (z < 0) { lmp->valid = 0; continue; } /* :395-399 */
        const double ipx = cam->fx * pc[0] / z + cam->cx, ipy = cam->fy * pc[1] / z + cam->cy; /* :400 */
        if (!in_image(&f->bounds, ipx, ipy)) { lmp->valid = 0; continue; }                     /* :401-405 */
        const double PO[3] = {c.campos[0] - lmp->pos[0], c.campos[1] - lmp->pos[1], c.campos[2] - lmp->pos[2]};
        const double dist  = sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]); /* :408 */
        /* MATCHING_MIN_MAX_DISTANCE2 (SnakeGlobal.h:168): :411-417 */
        const double sref     = (double)level_scale[lmp->reference_scale_level];
        const double max_dist = 1.2 * (double)lmp->reference_depth * sref;
        const double min_dist = 0.8 * (double)lmp->reference_depth * sref / s_last;
        if (dist < min_dist || dist > max_dist) { lmp->valid = 0; continue; }
        const double viewCos = (PO[0] * lmp->normal[0] + PO[1] * lmp->normal[1] + PO[2] * lmp->normal[2]) / dist; /* :423 */
        if (viewCos < 0.5) { lmp->valid = 0; continue; }                                                          /* :426-430 */
        visible[i] = 1; /* lmp.mp->IncreaseVisible() :431 */
        const float vcf = (float)viewCos;          /* RadiusByViewingCos(float viewCos) :357-363 */
        float r = (double)vcf > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;                    /* :450 */
        double prediction = (double)lmp->reference_scale_level + orc_det_log((double)lmp->reference_depth / dist) / log_f; /* :451 */
        if (prediction < 0.0) prediction = 0.0;
        if (prediction > (double)(n_levels - 1)) prediction = (double)(n_levels - 1);
        r = (float)((double)r * orc_det_exp(prediction * log_f)); /* :452 (float r *= double scale) */
        fine_ud u;
        u.f = f; u.desc1 = lmp->desc; u.ipx = ipx; u.z = z; u.bf = cam->bf; u.r = (double)r;
        u.best_dist = 256; u.best_level = -1; u.best_dist2 = 256; u.best_level2 = -1; u.best_idx = -1; /* :462-466 */
        for_candidates(f, ipx, ipy, (double)r, (double)r * (double)r, 2, 0, 0, prediction, fine_cand, &u); /* :453 */
        if (u.best_dist <= 100) /* TH_HIGH :508 */
        {
            if (u.best_level == u.best_level2 && (float)u.best_dist > ratio * (float)u.best_dist2) continue; /* :510 */
            best[i] = u.best_idx;
        }
    }
    uint8_t* taken2 = (uint8_t*)malloc((size_t)(f->n > 0 ? f->n : 1));
    memcpy(taken2, f->taken, (size_t)f->n);
    int matches = 0;
    for (int i = 0; i < m; ++i) /* :516-524 */
    {
        match_idx[i] = -1;
        if (best[i] == -1) continue;
        if (taken2[best[i]]) continue;
        taken2[best[i]] = 1;
        match_idx[i]    = best[i];
        matches++;
    }
    free(best);
    free(taken2);
    return matches;
}

/* ---------- reloc: SearchByProjectionFrameToKeyframe, SnakeORBMatcher.cpp:71-188 ---------- */
typedef struct kf_ud
{
    const orc_frame_view* f;
    const uint8_t* taken;
    const uint64_t* desc1;
    double ipx, z, bf, r;
    int best_dist, best_idx;
} kf_ud;

static int kf_cand(void* p, int i2)
{
    kf_ud* u = (kf_ud*)p;
    if (u->taken[i2]) return 0; /* :125 */
    if (u->f->right_points[i2] > 0) /* :126-135 */
    {
        const double disp = u->ipx - u->bf / u->z;
        const float er    = (float)fabs(disp - (double)u->f->right_points[i2]); /* const float er :130 */
        if ((double)er > u->r * 0.5) return 0;
    }
    const int dist = orc_hamming(u->desc1, u->f->desc[i2]);
    if (dist < u->best_dist)
    {
        u->best_dist = dist;
        u->best_idx  = i2;
    }
    return 0;
}

/* skip[i] != 0: the keyframe has no map point at i, or the frame already holds it (:102-103).
 * Assignments are sequential: a feature taken by point i is unavailable to every later point (:150). */
int orc_match_keyframe(const orc_frame_view* f, const orc_camera* cam, const double* pose, const double (*pos)[3],
                       const uint64_t (*desc)[4], const uint8_t* skip, int m, float th, int feature_error, int32_t* match_idx)
{
    view_ctx c;
    make_ctx(pose, &c);
    uint8_t* taken2 = (uint8_t*)malloc((size_t)(f->n > 0 ? f->n : 1));
    memcpy(taken2, f->taken, (size_t)f->n);
    int matches = 0;
    for (int i = 0; i < m; ++i)
    {
        match_idx[i] = -1;
        if (skip[i]) continue;
        double pc[3];
        transform(&c, pos[i], pc);
        const double z   = pc[2];
        const double ipx = cam->fx * pc[0] / z + cam->cx, ipy = cam->fy * pc[1] / z + cam->cy; /* :107 */
        if (z <= 0) continue;                                                                  /* :112 */
        if (!in_image(&f->bounds, ipx, ipy)) continue;                                         /* :114 */
        kf_ud u;
        u.f = f; u.taken = taken2; u.desc1 = desc[i]; u.ipx = ipx; u.z = z; u.bf = cam->bf; u.r = (double)th; /* :117 */
        u.best_dist = 256; u.best_idx = -1;
        /* GetFeaturesInArea(indices, ip, th): float r, r2 = r*r in float (Features.cpp:13-17) */
        for_candidates(f, ipx, ipy, (double)th, (double)(th * th), 0, 0, 0, 0.0, kf_cand, &u);
        if (u.best_dist <= feature_error) /* :148 */
        {
            taken2[u.best_idx] = 1; /* :150 */
            match_idx[i]       = u.best_idx;
            matches++;
        }
    }
    free(taken2);
    return matches;
}

/* ================================================================================================
 * Keyframe-rate matchers of local mapping
 * ================================================================================================ */

/* ---------- MappingORBMatcher::Fuse, LocalMap<FusionPoint> overload —
 * Snake/LocalMapping/MappingORBMatcher.cpp:359-480 (call sites NeighbourSearch.cpp:177,188).
 * MATCHING_MIN_MAX_DISTANCE2 and MATCHING_CHECK_SCALE_CONSISTENCY2 are defined
 * (SnakeGlobal.h:168-169).  Every point is independent: best_idx[i] = feature of the keyframe the
 * point would be fused into (fuseCandidates gets (best_idx[i], pts[i].id) in point order) or -1. */
typedef struct fuse_ud
{
    const orc_frame_view* f;
    const uint64_t* desc1;
    double ipx, ipy, ur; /* projectStereo(np) = (u, v, u - bf / z) */
    float gate;          /* th_squared * observationFactor */
    int best_dist, best_idx;
} fuse_ud;

static int fuse_cand(void* p, int idx)
{
    fuse_ud* u        = (fuse_ud*)p;
    const orc_kp64* kp = &u->f->kps[idx];
    const double dx = u->ipx - kp->x, dy = u->ipy - kp->y;
    double e2 = dx * dx + dy * dy;
    if (u->f->right_points[idx] > 0) /* hasDepth(idx) :447 */
    {
        const double dr = u->ur - (double)u->f->right_points[idx];
        e2 += dr * dr; /* (ips - ips2).squaredNorm() :454 */
    }
    if (e2 > (double)u->gate) return 0; /* :455 / :460 */
    const int dist = orc_hamming(u->desc1, u->f->desc[idx]);
    if (dist < u->best_dist) /* :465 */
    {
        u->best_dist = dist;
        u->best_idx  = idx;
    }
    return 0;
}

int orc_match_fuse(const orc_frame_view* f, const orc_camera* cam, const double* pose, const orc_fusion_point* pts,
                   const uint8_t* point_mask, int m, float th, float obs_factor, int feature_th, const float* level_scale,
                   int n_levels, int32_t* best_idx)
{
    view_ctx c;
    make_ctx(pose, &c);
    const float th_squared = th * th; /* :370 */
    const double log_f     = orc_det_log(n_levels > 1 ? (double)level_scale[1] / (double)level_scale[0] : 1.2);
    const double s_last    = (double)level_scale[n_levels - 1];
    int fused              = 0;
    for (int i = 0; i < m; ++i) /* :376 */
    {
        best_idx[i] = -1;
        const orc_fusion_point* lmp = &pts[i];
        if (point_mask && !point_mask[i]) continue; /* :381 */
        double np[3];
        transform(&c, lmp->pos, np);
        if (np[2] <= 0) continue; /* :391 */
        const double ipx = cam->fx * np[0] / np[2] + cam->cx, ipy = cam->fy * np[1] / np[2] + cam->cy; /* :393 */
        if (!in_image(&f->bounds, ipx, ipy)) continue;                                                 /* :394 */
        const double PO[3] = {c.campos[0] - lmp->pos[0], c.campos[1] - lmp->pos[1], c.campos[2] - lmp->pos[2]};
        const double dist  = sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]); /* :397 */
        int rl             = lmp->reference_scale_level;
        rl                 = rl < 0 ? 0 : (rl >= n_levels ? n_levels - 1 : rl);
        const double sref     = (double)level_scale[rl];
        const double max_dist = 1.2 * (double)lmp->reference_depth * sref; /* EstimateMinMaxDistance :402-407 */
        const double min_dist = 0.8 * (double)lmp->reference_depth * sref / s_last;
        if (dist < min_dist || dist > max_dist) continue;
        if (PO[0] * lmp->normal[0] + PO[1] * lmp->normal[1] + PO[2] * lmp->normal[2] < 0.5 * dist) continue; /* :414 */
        const float observationFactor = lmp->observations <= 2 ? obs_factor : 1.0f;                           /* :420-424 */
        const float radius            = observationFactor * th;                                               /* :429 */
        double prediction = (double)lmp->reference_scale_level + orc_det_log((double)lmp->reference_depth / dist) / log_f; /* :433 */
        if (prediction < 0.0) prediction = 0.0;
        if (prediction > (double)(n_levels - 1)) prediction = (double)(n_levels - 1);
        fuse_ud u;
        u.f = f; u.desc1 = lmp->desc; u.ipx = ipx; u.ipy = ipy; u.ur = ipx - cam->bf / np[2];
        u.gate = th_squared * observationFactor;
        u.best_dist = 256; u.best_idx = -1; /* :447-448 */
        for_candidates(f, ipx, ipy, (double)radius, (double)radius * (double)radius, 2, 0, 0, prediction, fuse_cand, &u); /* :434 */
        if (u.best_idx >= 0 && u.best_dist <= feature_th) /* :472 */
        {
            best_idx[i] = u.best_idx;
            fused++;
        }
    }
    return fused;
}

/* ---------- MappingORBMatcher::SearchForTriangulationProject —
 * Snake/LocalMapping/MappingORBMatcher.cpp:168-249 (call site Triangulator.cpp:170).
 * kf1 side: plain arrays (any order); kf2 side: a frame view in grid order whose `taken` flags mean
 * "already has a map point" (:218) plus its normalized points.  tmp_flags is never set by the
 * reference, so every feature of kf1 is independent.  [DEFINED] Saiga::EpipolarDistanceSquared
 * (absent): squared distance of np2 to the epipolar line E * (np1, 1) in image 2. */
typedef struct tri_ud
{
    const orc_frame_view* f;
    const double (*np2)[2];
    const uint64_t* desc1;
    double l[3], th_chi2;
    int feature_distance, best_dist, best_idx;
} tri_ud;

static int tri_cand(void* p, int idx2)
{
    tri_ud* u = (tri_ud*)p;
    if (u->f->taken[idx2]) return 0; /* :218 */
    const double d      = u->np2[idx2][0] * u->l[0] + u->np2[idx2][1] * u->l[1] + u->l[2];
    const double disepi = d * d / (u->l[0] * u->l[0] + u->l[1] * u->l[1]);
    if (disepi > u->th_chi2) return 0; /* :224 */
    const int dist = orc_hamming(u->desc1, u->f->desc[idx2]);
    if (dist > u->feature_distance || dist > u->best_dist) return 0; /* :232 */
    u->best_idx  = idx2;
    u->best_dist = dist;
    return 0;
}

int orc_match_triangulation_project(const double* depth_grid, int grid_rows, int grid_cols, const double* pose1,
                                    const double* pose2, const orc_camera* cam, const orc_kp64* kps1,
                                    const double (*np1)[2], const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1,
                                    const orc_frame_view* f2, const double (*np2)[2], const double* E12,
                                    float epipolar_distance, int feature_distance, int32_t* match_idx2)
{
    view_ctx c1, c2;
    make_ctx(pose1, &c1);
    make_ctx(pose2, &c2);
    const double th_chi1 = (double)epipolar_distance / cam->fx; /* :174 */
    const double th_chi2 = th_chi1 * th_chi1;
    int nmatches         = 0;
    for (int i = 0; i < n1; ++i) /* :187 */
    {
        match_idx2[i] = -1;
        if (has_mp1[i]) continue; /* :191 */
        const int cx = cell_coord(kps1[i].x, f2->bounds.min_x, f2->cols), cy = cell_coord(kps1[i].y, f2->bounds.min_y, f2->rows);
        const int gr = cy / 4 < grid_rows ? cy / 4 : grid_rows - 1, gc = cx / 4 < grid_cols ? cx / 4 : grid_cols - 1;
        const double z = depth_grid[gr * grid_cols + gc]; /* :195 */
        /* wp = pose1^-1 * K.unproject(point, z) */
        const double pc[3] = {(kps1[i].x - cam->cx) / cam->fx * z, (kps1[i].y - cam->cy) / cam->fy * z, z};
        const double d[3]  = {pc[0] - c1.t[0], pc[1] - c1.t[1], pc[2] - c1.t[2]};
        const double wp[3] = {c1.R[0] * d[0] + c1.R[3] * d[1] + c1.R[6] * d[2], c1.R[1] * d[0] + c1.R[4] * d[1] + c1.R[7] * d[2],
                              c1.R[2] * d[0] + c1.R[5] * d[1] + c1.R[8] * d[2]};
        double p2[3];
        transform(&c2, wp, p2);
        const double ipx = cam->fx * p2[0] / p2[2] + cam->cx, ipy = cam->fy * p2[1] / p2[2] + cam->cy; /* :198 */
        if (!in_image(&f2->bounds, ipx, ipy)) continue;                                                 /* :201 */
        tri_ud u;
        u.f = f2; u.np2 = np2; u.desc1 = desc1[i];
        const double x = np1[i][0], y = np1[i][1];
        u.l[0] = E12[0] * x + E12[1] * y + E12[2];
        u.l[1] = E12[3] * x + E12[4] * y + E12[5];
        u.l[2] = E12[6] * x + E12[7] * y + E12[8];
        u.th_chi2 = th_chi2; u.feature_distance = feature_distance;
        u.best_dist = 50; u.best_idx = -1; /* TH_LOW :206-207 */
        for_candidates(f2, ipx, ipy, 20.0, 400.0, 0, 0, 0, 0.0, tri_cand, &u); /* GetFeaturesInArea(ip2, 20) :211 */
        if (u.best_idx >= 0)
        {
            match_idx2[i] = u.best_idx;
            nmatches++;
        }
    }
    return nmatches;
}

/* MappingORBMatcher::SearchForTriangulation2 — reference Snake/LocalMapping/MappingORBMatcher.cpp:14-99 (call site
 * Snake/LocalMapping/Triangulator.cpp:164).  The bag-of-words feature vector of a keyframe (an ordered map node id ->
 * feature indices, frame->bow_feature_vec) arrives flattened: node_id ascending, node_start[n_nodes + 1] offsets into
 * `features`.  tmp_flags (:26-27) is never set by the reference, so every feature of keyframe 1 is independent.
 * pairs receives (idx1, idx2) in the reference's emplace order; returns nmatches. */
int orc_match_triangulation_bow(const orc_camera* cam, const double* E12, const double (*np1)[2], const uint64_t (*desc1)[4],
                                const uint8_t* has_mp1, int n_nodes1, const uint32_t* node_id1, const int32_t* node_start1,
                                const int32_t* feat1, const double (*np2)[2], const uint64_t (*desc2)[4], const uint8_t* has_mp2,
                                int n_nodes2, const uint32_t* node_id2, const int32_t* node_start2, const int32_t* feat2,
                                float epipolar_distance, int feature_distance, int32_t (*pairs)[2])
{
    const double th_chi1 = (double)(epipolar_distance * 2) / cam->fx; /* :20 (float product, then double division) */
    const double th_chi2 = th_chi1 * th_chi1;
    int nmatches         = 0;
    int a = 0, b = 0;
    while (a < n_nodes1 && b < n_nodes2) /* :34 */
    {
        if (node_id1[a] == node_id2[b])
        {
            for (int u = node_start1[a]; u < node_start1[a + 1]; ++u) /* :38 */
            {
                const int idx1 = feat1[u];
                if (has_mp1[idx1]) continue; /* :43 */
                const double x = np1[idx1][0], y = np1[idx1][1];
                const double l0 = E12[0] * x + E12[1] * y + E12[2], l1 = E12[3] * x + E12[4] * y + E12[5],
                             l2 = E12[6] * x + E12[7] * y + E12[8];
                int best_dist = 50, best_idx2 = -1; /* TH_LOW :50-51 */
                for (int v = node_start2[b]; v < node_start2[b + 1]; ++v) /* :54 */
                {
                    const int idx2 = feat2[v];
                    if (has_mp2[idx2]) continue; /* :59 */
                    const int dist = orc_hamming(desc1[idx1], desc2[idx2]);
                    if (dist > feature_distance || dist > best_dist) continue; /* :66 */
                    const double d      = np2[idx2][0] * l0 + np2[idx2][1] * l1 + l2;
                    const double disepi = d * d / (l0 * l0 + l1 * l1);
                    if (disepi < th_chi2) /* :72 */
                    {
                        best_idx2 = idx2;
                        best_dist = dist;
                    }
                }
                if (best_idx2 >= 0) /* :79 */
                {
                    pairs[nmatches][0] = idx1;
                    pairs[nmatches][1] = best_idx2;
                    nmatches++;
                }
            }
            ++a;
            ++b;
        }
        else if (node_id1[a] < node_id2[b]) /* lower_bound jumps :88-97 == advance to the first id >= the other */
        {
            while (a < n_nodes1 && node_id1[a] < node_id2[b]) ++a;
        }
        else
        {
            while (b < n_nodes2 && node_id2[b] < node_id1[a]) ++b;
        }
    }
    return nmatches;
}

/* MappingORBMatcher::SearchForTriangulationBF — reference Snake/LocalMapping/MappingORBMatcher.cpp:102-165: every
 * unmatched feature of keyframe 1 against every unmatched feature of keyframe 2, epipolar gate 10 px first (:107, the
 * epipolarDistance argument is unused), then the descriptor gate.  match_idx2[i] = idx2 or -1. */
int orc_match_triangulation_bf(const orc_camera* cam, const double* E12, const double (*np1)[2], const uint64_t (*desc1)[4],
                               const uint8_t* has_mp1, int n1, const double (*np2)[2], const uint64_t (*desc2)[4],
                               const uint8_t* has_mp2, int n2, int feature_distance, int32_t* match_idx2)
{
    const double th_chi1 = 10 / cam->fx; /* :107 */
    const double th_chi2 = th_chi1 * th_chi1;
    int nmatches         = 0;
    for (int idx1 = 0; idx1 < n1; ++idx1) /* :116 */
    {
        match_idx2[idx1] = -1;
        if (has_mp1[idx1]) continue; /* :120 */
        const double x = np1[idx1][0], y = np1[idx1][1];
        const double l0 = E12[0] * x + E12[1] * y + E12[2], l1 = E12[3] * x + E12[4] * y + E12[5],
                     l2 = E12[6] * x + E12[7] * y + E12[8];
        int best_dist = 50, best_idx2 = -1;
        for (int idx2 = 0; idx2 < n2; ++idx2) /* :129 */
        {
            if (has_mp2[idx2]) continue; /* :134 */
            const double d      = np2[idx2][0] * l0 + np2[idx2][1] * l1 + l2;
            const double disepi = d * d / (l0 * l0 + l1 * l1);
            if (disepi > th_chi2) continue; /* :140 */
            const int dist = orc_hamming(desc1[idx1], desc2[idx2]);
            if (dist > feature_distance || dist > best_dist) continue; /* :148 */
            best_idx2 = idx2;
            best_dist = dist;
        }
        if (best_idx2 >= 0)
        {
            match_idx2[idx1] = best_idx2;
            nmatches++;
        }
    }
    return nmatches;
}

/* DeferredMapper::Relink — reference Snake/Optimizer/DeferredMapper.cpp:39-165, the per-observation search (the data
 * parallel part).  One query = one feature of the keyframe that holds a good map point (:63-64).  action[q]:
 * 0 = keep, 1 = erase the observation (:75-81), 2 = relink candidate best_idx[q] (:140-163; the caller looks at
 * kf->GetMapPoint(best_idx) at that moment, erases or relinks, and re-queries a point it moved to a LATER slot with the
 * recomputed descriptor, as the sequential loop of the reference would revisit it).
 * stereo_cam.LeftPointToRight(x, z) (absent saiga) is [DEFINED] as x - bf / z, as in the tracking matchers. */
typedef struct relink_ud
{
    const orc_frame_view* f;
    const orc_relink_query* q;
    const orc_camera* cam;
    double ipx, ipy, z, rep2;
    int feature_threshold, best_dist, best_idx;
} relink_ud;

static int relink_cand(void* p, int j)
{
    relink_ud* u = (relink_ud*)p;
    if (j == u->q->feature) return 0; /* :107 */
    const double dx = u->ipx - u->f->kps[j].x, dy = u->ipy - u->f->kps[j].y;
    const double error_squared = dx * dx + dy * dy;
    if (error_squared > u->rep2) return 0; /* :113 */
    if (u->f->right_points[j] > 0)         /* :115 */
    {
        const double disp = u->ipx - u->cam->bf / u->z;
        const double er   = disp - (double)u->f->right_points[j];
        if (er * er > u->rep2 * 2.0) return 0; /* :119 */
    }
    const int d2 = orc_hamming(u->q->desc, u->f->desc[j]);
    if (d2 < u->feature_threshold && d2 < u->best_dist) /* :126 */
    {
        u->best_dist = d2;
        u->best_idx  = j;
    }
    return 0;
}

int orc_match_relink(const orc_frame_view* f, const orc_camera* cam, const double* pose, const orc_relink_query* queries, int n,
                     float radius, double outlier_threshold, int feature_threshold, int32_t* action, int32_t* best_idx)
{
    view_ctx c;
    make_ctx(pose, &c);
    const double out2 = outlier_threshold * outlier_threshold; /* :54 */
    const float r2f   = radius * radius;                       /* Features.cpp:17 (float) */
    int changed       = 0;
    for (int q = 0; q < n; ++q)
    {
        const orc_relink_query* Q = &queries[q];
        action[q]   = 0;
        best_idx[q] = -1;
        double np[3];
        transform(&c, Q->pos, np); /* :70 */
        const double z   = np[2];
        const double ipx = cam->fx * np[0] / z + cam->cx, ipy = cam->fy * np[1] / z + cam->cy; /* :72 */
        const double ex = ipx - f->kps[Q->feature].x, ey = ipy - f->kps[Q->feature].y;
        const double rep2 = ex * ex + ey * ey;
        if (z <= 0 || rep2 > out2) /* :75 */
        {
            action[q] = 1;
            changed++;
            continue;
        }
        int feature_dist = orc_hamming(Q->desc, f->desc[Q->feature]);          /* :85 */
        if (feature_dist == 0 && Q->has_alt) feature_dist = orc_hamming(Q->desc, Q->alt_desc); /* :87-99 */
        relink_ud u;
        u.f = f; u.q = Q; u.cam = cam; u.ipx = ipx; u.ipy = ipy; u.z = z; u.rep2 = rep2;
        u.feature_threshold = feature_threshold; u.best_dist = feature_dist; u.best_idx = -1; /* :101-102 */
        for_candidates(f, ipx, ipy, (double)radius, (double)r2f, 0, 0, 0, 0.0, relink_cand, &u); /* :100 */
        if (u.best_idx != -1)
        {
            action[q]   = 2;
            best_idx[q] = u.best_idx;
            changed++;
        }
    }
    return changed;
}
