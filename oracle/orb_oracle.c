/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the ORB extractor the reference calls as
 * Saiga::ORBExtractor::Detect (call sites Snake/Preprocess/FeatureDetector.cpp:40-41,124,154;
 * GPU twin ORBExtractorGPU at :31-33,119,149).
 *
 * PARITY UNPINNED.  The extractor's source lives in the absent, unpinned submodule
 * darglein/saiga (reference .gitmodules:1-3) and the reference ships no golden vectors.  What is
 * restated here is the published ORB-SLAM2 extractor that saiga's descends from (scale pyramid,
 * cell-wise two-threshold FAST-9/16 with 3x3 non-max suppression, quadtree distribution,
 * intensity-centroid angle, 7x7 Gaussian, steered BRIEF-256), with every step the Snake side
 * does not fix DEFINED here in integer / explicitly ordered float arithmetic so that a GPU
 * implementation can match it bit for bit.  The definitions ("snk-orb v1") are listed in
 * DESIGN.md §ORB; each is marked [ORB-SLAM2] (follows the published algorithm) or [DEFINED]
 * (a choice of this repository).
 *
 * Plain C.  Optional OpenMP over pyramid levels only for the cpu_baseline timing
 * (the reference runs the extractor with fd_threads = 2, reference configs/euroc.ini:37).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "snk_oracle.h"

#define PATCH_SIZE 31
#define HALF_PATCH 15
#define EDGE_THRESHOLD 19
#define CELL_W 30
#define MIN_BORDER (EDGE_THRESHOLD - 3)

/* ----------------------------------------------------------------------------------------------
 * [ORB-SLAM2] learned BRIEF test pattern (256 pairs of points inside the 31x31 patch), the
 * table published with ORB (Rublee et al. 2011) and shipped by ORB-SLAM2 as bit_pattern_31_.
 * Written out from the public table; UNVERIFIED against saiga's copy.
 * ---------------------------------------------------------------------------------------------- */
const int8_t orc_brief_pattern[1024] = {
#include "brief_pattern.inc"
};

/* ---------------------------------------------------------------------------------------------- */
/* level geometry, [ORB-SLAM2] ORBextractor constructor arithmetic (float)                        */
/* ---------------------------------------------------------------------------------------------- */
int orc_orb_layout(const orc_orb_params* p, int w, int h, orc_orb_layout_t* L)
{
    if (p->n_levels < 1 || p->n_levels > ORC_MAX_LEVELS) return -1;
    if (!(p->scale_factor > 1.0f)) return -1;
    if (p->ini_th < 1 || p->min_th < 1 || p->ini_th > 254 || p->min_th > 254) return -1;
    L->n_levels = p->n_levels;
    L->scale[0] = 1.0f;
    for (int l = 1; l < p->n_levels; ++l) L->scale[l] = L->scale[l - 1] * p->scale_factor;
    for (int l = 0; l < p->n_levels; ++l)
    {
        float inv = 1.0f / L->scale[l];
        L->w[l]   = (int)lrintf((float)w * inv);
        L->h[l]   = (int)lrintf((float)h * inv);
    }
    float factor   = 1.0f / p->scale_factor;
    float nDesired = (float)p->nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)p->n_levels));
    int sum        = 0;
    for (int l = 0; l < p->n_levels - 1; ++l)
    {
        L->nfeat[l] = (int)lrintf(nDesired);
        sum += L->nfeat[l];
        nDesired *= factor;
    }
    L->nfeat[p->n_levels - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    return 0;
}

/* ---------------------------------------------------------------------------------------------- */
/* [DEFINED] bilinear down-scale of level l-1 to level l, 11-bit fixed-point weights.             */
/* Source coordinate as in OpenCV INTER_LINEAR: fx = (dx + 0.5) * (sw / dw) - 0.5 (double),       */
/* clamped; weights w1 = lrint(frac * 2048), w0 = 2048 - w1;                                      */
/* out = (sum of the 4 weighted taps + 2^21) >> 22.                                               */
/* ---------------------------------------------------------------------------------------------- */
void orc_resize_coords(int src, int dst, int32_t* ofs, int32_t* w1)
{
    double scale = (double)src / (double)dst;
    for (int d = 0; d < dst; ++d)
    {
        double f = ((double)d + 0.5) * scale - 0.5;
        int s    = (int)floor(f);
        f -= (double)s;
        if (s < 0)
        {
            s = 0;
            f = 0.0;
        }
        if (s >= src - 1)
        {
            s = src - 1;
            f = 0.0;
        }
        ofs[d] = s;
        w1[d]  = (int32_t)lrint(f * 2048.0);
    }
}

void orc_resize(const uint8_t* src, int sw, int sh, int spitch, uint8_t* dst, int dw, int dh, int dpitch)
{
    int32_t* xo = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw * 2);
    int32_t* yo = (int32_t*)malloc(sizeof(int32_t) * (size_t)dh * 2);
    orc_resize_coords(sw, dw, xo, xo + dw);
    orc_resize_coords(sh, dh, yo, yo + dh);
    for (int y = 0; y < dh; ++y)
    {
        int sy = yo[y], wy1 = yo[dh + y], wy0 = 2048 - wy1;
        int sy1           = sy + 1 < sh ? sy + 1 : sh - 1;
        const uint8_t* r0 = src + (size_t)sy * spitch;
        const uint8_t* r1 = src + (size_t)sy1 * spitch;
        for (int x = 0; x < dw; ++x)
        {
            int sx = xo[x], wx1 = xo[dw + x], wx0 = 2048 - wx1;
            int sx1    = sx + 1 < sw ? sx + 1 : sw - 1;
            int32_t v  = (r0[sx] * wx0 + r0[sx1] * wx1) * wy0 + (r1[sx] * wx0 + r1[sx1] * wx1) * wy1;
            dst[(size_t)y * dpitch + x] = (uint8_t)((v + (1 << 21)) >> 22);
        }
    }
    free(xo);
    free(yo);
}

/* ---------------------------------------------------------------------------------------------- */
/* FAST-9/16 score.  [ORB-SLAM2/OpenCV semantics] A pixel is a corner at threshold t iff 9        */
/* contiguous pixels of the 16-pixel radius-3 circle are all > c + t or all < c - t.  With        */
/* S = max over the 16 arcs of min over the arc of (ring - c), and the same for (c - ring),       */
/* corner(t) <=> S > t, and OpenCV's corner score ("largest t that keeps it a corner") is S - 1.  */
/* ---------------------------------------------------------------------------------------------- */
static const int8_t ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int8_t ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int orc_fast_score(const uint8_t* img, int pitch, int x, int y)
{
    int c = img[(size_t)y * pitch + x];
    int d[16];
    for (int i = 0; i < 16; ++i) d[i] = (int)img[(size_t)(y + ring_dy[i]) * pitch + (x + ring_dx[i])] - c;
    int best = -1000;
    for (int k = 0; k < 16; ++k)
    {
        int mn = 1000, mx = -1000;
        for (int i = 0; i < 9; ++i)
        {
            int v = d[(k + i) & 15];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        if (mn > best) best = mn;   /* bright arc */
        if (-mx > best) best = -mx; /* dark arc   */
    }
    return best;
}

/* Same value as orc_fast_score when the score exceeds `th`, any value <= th otherwise (only
 * scores above the cell threshold are ever compared).  Quick reject: every 9-arc of the circle
 * contains one pixel of each opposite pair, so S <= min_i max(d_i, d_{i+8}) (and the dark twin). */
static int fast_score_above(const uint8_t* img, int pitch, int x, int y, int th)
{
    const uint8_t* c = img + (size_t)y * pitch + x;
    int v  = c[0];
    int d0 = c[3 * pitch] - v, d8 = c[-3 * pitch] - v, d4 = c[3] - v, d12 = c[-3] - v;
    int ub_b = (d0 > d8 ? d0 : d8), t = (d4 > d12 ? d4 : d12);
    if (t < ub_b) ub_b = t;
    int ub_d = (-d0 > -d8 ? -d0 : -d8);
    t        = (-d4 > -d12 ? -d4 : -d12);
    if (t < ub_d) ub_d = t;
    if (ub_b <= th && ub_d <= th) return 0 < th ? 0 : th;
    return orc_fast_score(img, pitch, x, y);
}

/* [ORB-SLAM2] cell grid of ComputeKeyPointsOctTree: cells of ~30 px anchored at the border
 * EDGE_THRESHOLD-3; FAST runs per cell with iniThFAST and, if the cell yields nothing, again
 * with minThFAST; OpenCV's FAST applies 3x3 non-max suppression inside the cell's sub-image
 * (pixels outside the cell's detection range count as score 0).  Detection range of cell (i,j):
 * x in [19 + j*wCell, min(19 + (j+1)*wCell, w-19)), same for y.  */
void orc_cell_grid(int w, int h, orc_cell_grid_t* g)
{
    int width  = w - 2 * MIN_BORDER;
    int height = h - 2 * MIN_BORDER;
    g->n_cols  = width / CELL_W;
    g->n_rows  = height / CELL_W;
    if (g->n_cols < 1 || g->n_rows < 1 || w < 2 * EDGE_THRESHOLD + 1 || h < 2 * EDGE_THRESHOLD + 1)
    {
        g->n_cols = g->n_rows = 0;
        g->w_cell = g->h_cell = 1;
        return;
    }
    g->w_cell = (width + g->n_cols - 1) / g->n_cols;
    g->h_cell = (height + g->n_rows - 1) / g->n_rows;
}

/* Candidates of one level: NMS survivors above the cell's effective threshold.
 * Output order: cell-major (row, col), raster order inside the cell.  Returns the count (may
 * exceed cap; only the first cap are stored — callers use orc_orb_candidates for the capped,
 * well-defined list). */
static int level_candidates_raw(const uint8_t* img, int w, int h, int pitch, int ini_th, int min_th, orc_cand* out,
                                int* cell_count, int cap)
{
    orc_cell_grid_t g;
    orc_cell_grid(w, h, &g);
    int n       = 0;
    int16_t* S  = (int16_t*)malloc(sizeof(int16_t) * 64 * 64);
    for (int ci = 0; ci < g.n_rows; ++ci)
        for (int cj = 0; cj < g.n_cols; ++cj)
        {
            int x0 = EDGE_THRESHOLD + cj * g.w_cell, x1 = x0 + g.w_cell;
            int y0 = EDGE_THRESHOLD + ci * g.h_cell, y1 = y0 + g.h_cell;
            if (x1 > w - EDGE_THRESHOLD) x1 = w - EDGE_THRESHOLD;
            if (y1 > h - EDGE_THRESHOLD) y1 = h - EDGE_THRESHOLD;
            int cw = x1 - x0, ch = y1 - y0;
            int cnt = 0;
            if (cw > 0 && ch > 0)
            {
                if (cw > 62 || ch > 62) abort(); /* cells are at most 59 px by construction */
                /* score map with a zero ring around the cell */
                memset(S, 0, sizeof(int16_t) * 64 * 64);
                for (int y = 0; y < ch; ++y)
                    for (int x = 0; x < cw; ++x) S[(y + 1) * 64 + (x + 1)] = (int16_t)fast_score_above(img, pitch, x0 + x, y0 + y, min_th);
                for (int pass = 0; pass < 2 && cnt == 0; ++pass)
                {
                    int th = pass == 0 ? ini_th : min_th;
                    for (int y = 0; y < ch; ++y)
                        for (int x = 0; x < cw; ++x)
                        {
                            const int16_t* s = &S[(y + 1) * 64 + (x + 1)];
                            int v            = s[0];
                            if (v <= th) continue;
                            if (v > s[-1] && v > s[1] && v > s[-65] && v > s[-64] && v > s[-63] && v > s[63] && v > s[64] &&
                                v > s[65])
                            {
                                if (n + cnt < cap)
                                {
                                    out[n + cnt].x     = (uint16_t)(x0 + x);
                                    out[n + cnt].y     = (uint16_t)(y0 + y);
                                    out[n + cnt].score = (uint16_t)v;
                                    out[n + cnt].cell  = (uint16_t)(ci * g.n_cols + cj);
                                }
                                cnt++;
                            }
                        }
                }
            }
            if (cell_count) cell_count[ci * g.n_cols + cj] = cnt;
            n += cnt;
        }
    free(S);
    return n;
}

static int cand_cmp_strength(const void* a, const void* b)
{
    const orc_cand* p = (const orc_cand*)a;
    const orc_cand* q = (const orc_cand*)b;
    if (p->score != q->score) return p->score > q->score ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return 0;
}

/* [DEFINED] candidate budget of a level: every cell keeps its k strongest candidates (score desc,
 * then y, then x), with k the largest value <= ORC_CELL_SLOTS (64) for which the level total
 * sum(min(n_cell, k)) fits in `cap`.  (No effect unless a cell holds more than 64 candidates or
 * the level more than `cap`.)  Order of the result: cell-major, raster inside the cell; nothing
 * downstream depends on that order.  Returns the number stored. */
int orc_orb_candidates(const uint8_t* img, int w, int h, int pitch, int ini_th, int min_th, orc_cand* out, int cap)
{
    orc_cell_grid_t g;
    orc_cell_grid(w, h, &g);
    int ncell = g.n_cols * g.n_rows;
    if (ncell <= 0) return 0;
    int* cc   = (int*)calloc((size_t)ncell, sizeof(int));
    int total = level_candidates_raw(img, w, h, pitch, ini_th, min_th, out, cc, cap);
    int over  = 0;
    for (int c = 0; c < ncell; ++c) over |= cc[c] > ORC_CELL_SLOTS;
    if (total <= cap && !over)
    {
        free(cc);
        return total;
    }
    orc_cand* all = (orc_cand*)malloc(sizeof(orc_cand) * (size_t)total);
    level_candidates_raw(img, w, h, pitch, ini_th, min_th, all, cc, total);
    int k = ORC_CELL_SLOTS;
    for (;;)
    {
        long s = 0;
        for (int c = 0; c < ncell; ++c) s += cc[c] < k ? cc[c] : k;
        if (s <= cap || k == 0) break;
        --k;
    }
    int n = 0, base = 0;
    for (int c = 0; c < ncell; ++c)
    {
        int cnt = cc[c];
        if (cnt > k)
        {
            /* keep the k strongest, emitted in raster order */
            orc_cand* tmp = (orc_cand*)malloc(sizeof(orc_cand) * (size_t)cnt);
            memcpy(tmp, all + base, sizeof(orc_cand) * (size_t)cnt);
            qsort(tmp, (size_t)cnt, sizeof(orc_cand), cand_cmp_strength);
            for (int i = 0; i < cnt; ++i)
            {
                const orc_cand* c0 = &all[base + i];
                int keep           = 0;
                for (int j = 0; j < k; ++j)
                    if (tmp[j].x == c0->x && tmp[j].y == c0->y) keep = 1;
                if (keep) out[n++] = *c0;
            }
            free(tmp);
        }
        else
        {
            for (int i = 0; i < cnt; ++i) out[n++] = all[base + i];
        }
        base += cnt;
    }
    free(all);
    free(cc);
    return n;
}

/* ---------------------------------------------------------------------------------------------- */
/* [ORB-SLAM2] DistributeOctTree, restated with explicit node rectangles and lists.              */
/* [DEFINED]   integer node geometry, the tie-breaks ORB-SLAM2 leaves to pointer order, and the  */
/*             output order.                                                                      */
/*   region : [0,W) x [0,H) with W = w - 32, H = h - 32, coordinates relative to (16,16)          */
/*   roots  : nIni = max(1, floor(W/H + 0.5)); root i = [ceil(i*W/nIni), ceil((i+1)*W/nIni))      */
/*   split  : midx = x0 + ceil((x1-x0)/2), midy likewise; child = (x>=midx) + 2*(y>=midy)         */
/*   key    : root index followed by the 16 child digits of the point (2 bits each)               */
/*   careful phase order: count descending, then key-prefix ascending                             */
/*   kept point per node: highest score, then smallest y, then smallest x                         */
/*   output order: ascending key of the kept point                                                */
/* ---------------------------------------------------------------------------------------------- */
typedef struct qnode
{
    int x0, x1, y0, y1;
    int begin, count; /* points are kept in an index array, children partition it */
    uint64_t prefix;  /* key prefix, left-aligned like orc_point_key */
    int depth;
} qnode;

static int n_roots(int W, int H)
{
    int n = (2 * W + H) / (2 * H);
    return n < 1 ? 1 : (n > 255 ? 255 : n);
}

uint64_t orc_point_key(int x, int y, int W, int H)
{
    int nIni = n_roots(W, H);
    int root = (int)(((long)x * nIni) / W);
    int x0 = (int)(((long)root * W + nIni - 1) / nIni), x1 = (int)(((long)(root + 1) * W + nIni - 1) / nIni);
    int y0 = 0, y1 = H;
    uint64_t key = (uint64_t)root;
    for (int d = 0; d < 16; ++d)
    {
        int mx = x0 + (x1 - x0 + 1) / 2, my = y0 + (y1 - y0 + 1) / 2;
        int c  = (x >= mx ? 1 : 0) + (y >= my ? 2 : 0);
        if (x >= mx) x0 = mx; else x1 = mx;
        if (y >= my) y0 = my; else y1 = my;
        key = (key << 2) | (uint64_t)c;
    }
    return key;
}

static uint64_t prefix_of(uint64_t key, int depth)
{
    /* key = root(8 bits) : 16 digits(32 bits); keep root + `depth` digits, zero the rest */
    int drop = 2 * (16 - depth);
    return (key >> drop) << drop;
}

typedef struct exp_item
{
    int count;
    uint64_t prefix;
    int node;
} exp_item;

static int exp_cmp(const void* a, const void* b)
{
    const exp_item* p = (const exp_item*)a;
    const exp_item* q = (const exp_item*)b;
    if (p->count != q->count) return p->count > q->count ? -1 : 1;
    if (p->prefix != q->prefix) return p->prefix < q->prefix ? -1 : 1;
    return 0;
}

typedef struct qt
{
    qnode* nodes;
    int n_nodes, cap_nodes;
    uint8_t* alive;
    int n_alive;
    int* idx;  /* point indices, partitioned per node */
    int* tmp;
    const orc_cand* pts;
} qt;

static int qt_add(qt* t, qnode nd)
{
    if (t->n_nodes == t->cap_nodes)
    {
        t->cap_nodes *= 2;
        t->nodes = (qnode*)realloc(t->nodes, sizeof(qnode) * (size_t)t->cap_nodes);
        t->alive = (uint8_t*)realloc(t->alive, (size_t)t->cap_nodes);
    }
    t->nodes[t->n_nodes] = nd;
    t->alive[t->n_nodes] = 1;
    t->n_alive++;
    return t->n_nodes++;
}

/* split node `ni`; append its non-empty children; multi-point children are appended to `exp` */
static void qt_split(qt* t, int ni, exp_item* exp, int* n_exp)
{
    qnode nd = t->nodes[ni];
    int mx = nd.x0 + (nd.x1 - nd.x0 + 1) / 2, my = nd.y0 + (nd.y1 - nd.y0 + 1) / 2;
    int cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < nd.count; ++i)
    {
        const orc_cand* p = &t->pts[t->idx[nd.begin + i]];
        int px = p->x - MIN_BORDER, py = p->y - MIN_BORDER;
        cnt[(px >= mx ? 1 : 0) + (py >= my ? 2 : 0)]++;
    }
    int start[4], fill[4];
    start[0] = nd.begin;
    for (int c = 1; c < 4; ++c) start[c] = start[c - 1] + cnt[c - 1];
    memcpy(fill, start, sizeof(fill));
    for (int i = 0; i < nd.count; ++i)
    {
        int id            = t->idx[nd.begin + i];
        const orc_cand* p = &t->pts[id];
        int px = p->x - MIN_BORDER, py = p->y - MIN_BORDER;
        int c  = (px >= mx ? 1 : 0) + (py >= my ? 2 : 0);
        t->tmp[fill[c]++] = id;
    }
    memcpy(t->idx + nd.begin, t->tmp + nd.begin, sizeof(int) * (size_t)nd.count);
    t->alive[ni] = 0;
    t->n_alive--;
    for (int c = 0; c < 4; ++c)
    {
        if (cnt[c] == 0) continue;
        qnode ch;
        ch.x0     = (c & 1) ? mx : nd.x0;
        ch.x1     = (c & 1) ? nd.x1 : mx;
        ch.y0     = (c & 2) ? my : nd.y0;
        ch.y1     = (c & 2) ? nd.y1 : my;
        ch.begin  = start[c];
        ch.count  = cnt[c];
        ch.depth  = nd.depth + 1;
        ch.prefix = nd.prefix | ((uint64_t)c << (2 * (16 - ch.depth)));
        int id    = qt_add(t, ch);
        if (cnt[c] > 1)
        {
            exp[*n_exp].count  = cnt[c];
            exp[*n_exp].prefix = ch.prefix;
            exp[*n_exp].node   = id;
            (*n_exp)++;
        }
    }
}

typedef struct sel_item
{
    uint64_t key;
    int id;
} sel_item;
static int sel_cmp(const void* a, const void* b)
{
    const sel_item* p = (const sel_item*)a;
    const sel_item* q = (const sel_item*)b;
    return p->key < q->key ? -1 : (p->key > q->key ? 1 : 0);
}

/* Upper bound of the selection size for a w x h level: one careful split adds <= 3 nodes beyond N;
 * the unconditional first pass turns every root into <= 4 nodes (wide levels: 4 * roots can exceed N). */
int orc_orb_distribute_bound(int w, int h, int N)
{
    int W = w - 2 * MIN_BORDER, H = h - 2 * MIN_BORDER;
    if (W <= 0 || H <= 0 || N <= 0) return 0;
    int r = 4 * n_roots(W, H);
    return N + 3 > r ? N + 3 : r;
}

/* Select up to ~N of the n candidates of a w x h level.  out_idx receives candidate indices in
 * output order (capacity orc_orb_distribute_bound); returns the number selected. */
int orc_orb_distribute(const orc_cand* pts, int n, int w, int h, int N, int* out_idx)
{
    return orc_orb_distribute_ranked(pts, NULL, n, w, h, N, out_idx);
}

/* The same distribution with the kept point of a node chosen by `rank` (higher first, then smallest y, then smallest x)
 * instead of the FAST score: "orb.response" = 1 ranks by the Harris response (orc_harris_rank).  rank == NULL: the score. */
int orc_orb_distribute_ranked(const orc_cand* pts, const uint32_t* rank, int n, int w, int h, int N, int* out_idx)
{
    if (n <= 0 || N <= 0) return 0;
    int W = w - 2 * MIN_BORDER, H = h - 2 * MIN_BORDER;
    int nIni = n_roots(W, H);
    qt t;
    t.cap_nodes = 4 * n + 16 + nIni;
    t.nodes     = (qnode*)malloc(sizeof(qnode) * (size_t)t.cap_nodes);
    t.alive     = (uint8_t*)malloc((size_t)t.cap_nodes);
    t.n_nodes = t.n_alive = 0;
    t.idx = (int*)malloc(sizeof(int) * (size_t)n);
    t.tmp = (int*)malloc(sizeof(int) * (size_t)n);
    t.pts = pts;
    exp_item* exp  = (exp_item*)malloc(sizeof(exp_item) * (size_t)(4 * n + 16));
    exp_item* prev = (exp_item*)malloc(sizeof(exp_item) * (size_t)(4 * n + 16));
    int n_exp      = 0;

    /* roots */
    {
        int* rc = (int*)calloc((size_t)nIni + 1, sizeof(int));
        for (int i = 0; i < n; ++i) rc[(int)(((long)(pts[i].x - MIN_BORDER) * nIni) / W) + 1]++;
        for (int r = 0; r < nIni; ++r) rc[r + 1] += rc[r];
        int* fill = (int*)malloc(sizeof(int) * (size_t)nIni);
        memcpy(fill, rc, sizeof(int) * (size_t)nIni);
        for (int i = 0; i < n; ++i) t.idx[fill[(int)(((long)(pts[i].x - MIN_BORDER) * nIni) / W)]++] = i;
        for (int r = 0; r < nIni; ++r)
        {
            int cnt = rc[r + 1] - rc[r];
            if (cnt == 0) continue; /* ORB-SLAM2 erases empty roots */
            qnode nd;
            nd.x0     = (int)(((long)r * W + nIni - 1) / nIni);
            nd.x1     = (int)(((long)(r + 1) * W + nIni - 1) / nIni);
            nd.y0     = 0;
            nd.y1     = H;
            nd.begin  = rc[r];
            nd.count  = cnt;
            nd.depth  = 0;
            nd.prefix = (uint64_t)r << 32;
            qt_add(&t, nd);
        }
        free(rc);
        free(fill);
    }

    int finish = 0;
    while (!finish)
    {
        int prevSize = t.n_alive;
        n_exp        = 0;
        int upto     = t.n_nodes; /* children appended during the pass are not revisited */
        for (int i = 0; i < upto; ++i)
        {
            if (!t.alive[i] || t.nodes[i].count == 1) continue;
            qt_split(&t, i, exp, &n_exp);
        }
        if (t.n_alive >= N || t.n_alive == prevSize)
            finish = 1;
        else if (t.n_alive + n_exp * 3 > N)
        {
            while (!finish)
            {
                prevSize   = t.n_alive;
                int n_prev = n_exp;
                memcpy(prev, exp, sizeof(exp_item) * (size_t)n_prev);
                n_exp = 0;
                qsort(prev, (size_t)n_prev, sizeof(exp_item), exp_cmp);
                for (int j = 0; j < n_prev; ++j)
                {
                    qt_split(&t, prev[j].node, exp, &n_exp);
                    if (t.n_alive >= N) break;
                }
                if (t.n_alive >= N || t.n_alive == prevSize) finish = 1;
            }
        }
    }

    /* best point per node, output sorted by key */
    sel_item* sel = (sel_item*)malloc(sizeof(sel_item) * (size_t)t.n_alive);
    int ns        = 0;
    for (int i = 0; i < t.n_nodes; ++i)
    {
        if (!t.alive[i]) continue;
        const qnode* nd = &t.nodes[i];
        int best        = t.idx[nd->begin];
        for (int k = 1; k < nd->count; ++k)
        {
            int id = t.idx[nd->begin + k];
            const orc_cand *a = &pts[id], *b = &pts[best];
            const uint32_t ra = rank ? rank[id] : a->score, rb = rank ? rank[best] : b->score;
            if (ra > rb || (ra == rb && (a->y < b->y || (a->y == b->y && a->x < b->x)))) best = id;
        }
        sel[ns].id  = best;
        sel[ns].key = orc_point_key(pts[best].x - MIN_BORDER, pts[best].y - MIN_BORDER, W, H);
        ns++;
    }
    qsort(sel, (size_t)ns, sizeof(sel_item), sel_cmp);
    for (int i = 0; i < ns; ++i) out_idx[i] = sel[i].id;
    free(sel);
    free(exp);
    free(prev);
    free(t.nodes);
    free(t.alive);
    free(t.idx);
    free(t.tmp);
    (void)prefix_of;
    return ns;
}

/* ---------------------------------------------------------------------------------------------- */
/* orientation                                                                                    */
/* ---------------------------------------------------------------------------------------------- */
/* [ORB-SLAM2] umax table of the radius-15 disc (ORBextractor constructor). */
void orc_umax(int* umax /* 16 */)
{
    int v, v0;
    int vmax         = (int)floor(HALF_PATCH * sqrt(2.0) / 2 + 1);
    int vmin         = (int)ceil(HALF_PATCH * sqrt(2.0) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v)
    {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* [ORB-SLAM2] IC_Angle moments (integers, exact). */
void orc_ic_moments(const uint8_t* img, int pitch, int x, int y, int* m10, int* m01)
{
    int umax[16];
    orc_umax(umax);
    const uint8_t* c = img + (size_t)y * pitch + x;
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * c[u];
    for (int v = 1; v <= HALF_PATCH; ++v)
    {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u)
        {
            int val_plus = c[u + v * pitch], val_minus = c[u - v * pitch];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    *m10 = m_10;
    *m01 = m_01;
}

/* [ORB-SLAM2/OpenCV] cv::fastAtan2 polynomial (degrees in [0,360)), restated with a fixed
 * float operation order (no FMA) so the kernel can reproduce it bit for bit. */
float orc_fast_atan2(float y, float x)
{
    /* 0.9997878412794807f, -0.3258083974640975f, 0.1555786518463281f, -0.04432655554792128f, each
     * times (float)(180/pi), written as exact float literals */
    const float p1 = 0x1.ca44dep+5f, p3 = -0x1.2aaddcp+4f, p5 = 0x1.1d3f7ep+3f, p7 = -0x1.4515b2p+1f;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay)
    {
        c  = ay / (ax + 0x1p-52f);
        c2 = c * c;
        a  = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else
    {
        c  = ax / (ay + 0x1p-52f);
        c2 = c * c;
        a  = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.0f - a;
    if (y < 0) a = 360.0f - a;
    return a;
}

/* [DEFINED] sine / cosine of an angle in degrees, float, fixed operation order: exact octant
 * reduction, then Taylor polynomials on [0, 45] degrees (Horner, no FMA). */
void orc_sincos_deg(float deg, float* s_out, float* c_out)
{
    int q = 0;
    float r = deg;
    if (r >= 360.0f) r -= 360.0f;
    if (r >= 270.0f) { r -= 270.0f; q = 3; }
    else if (r >= 180.0f) { r -= 180.0f; q = 2; }
    else if (r >= 90.0f) { r -= 90.0f; q = 1; }
    int swap = 0;
    if (r > 45.0f) { r = 90.0f - r; swap = 1; }
    float x  = r * 0.017453292519943295f;
    float x2 = x * x;
    float s  = x + x * x2 * (-1.6666667e-1f + x2 * (8.3333333e-3f + x2 * (-1.9841270e-4f + x2 * 2.7557319e-6f)));
    float c  = 1.0f + x2 * (-0.5f + x2 * (4.1666667e-2f + x2 * (-1.3888889e-3f + x2 * 2.4801587e-5f)));
    if (swap) { float t = s; s = c; c = t; }
    switch (q)
    {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* blur + descriptor                                                                              */
/* ---------------------------------------------------------------------------------------------- */
static inline int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/* [DEFINED] 7x7 Gaussian, sigma = 2, as an 8-bit fixed-point separable kernel; the two passes are
 * exact integers and the single rounding is (sum + 2^15) >> 16.  Border: reflect-101 [ORB-SLAM2]. */
static const int gk7[7] = {18, 33, 49, 56, 49, 33, 18};

int orc_blur_at(const uint8_t* img, int w, int h, int pitch, int x, int y)
{
    int acc = 0;
    for (int dy = -3; dy <= 3; ++dy)
    {
        int yy  = reflect101(y + dy, h);
        int row = 0;
        for (int dx = -3; dx <= 3; ++dx) row += gk7[dx + 3] * img[(size_t)yy * pitch + reflect101(x + dx, w)];
        acc += gk7[dy + 3] * row;
    }
    return (acc + (1 << 15)) >> 16;
}

/* The same blur over a whole level (what ORB-SLAM2 does before computing descriptors): separable,
 * exact 16-bit intermediate, one rounding.  Bit-identical to orc_blur_at at every pixel. */
void orc_blur_image(const uint8_t* img, int w, int h, int pitch, uint8_t* dst, int dpitch)
{
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * (size_t)h);
    for (int y = 0; y < h; ++y)
    {
        const uint8_t* r = img + (size_t)y * pitch;
        uint16_t* t      = tmp + (size_t)y * w;
        for (int x = 0; x < w; ++x)
        {
            int acc = 0;
            if (x >= 3 && x < w - 3)
                for (int k = -3; k <= 3; ++k) acc += gk7[k + 3] * r[x + k];
            else
                for (int k = -3; k <= 3; ++k) acc += gk7[k + 3] * r[reflect101(x + k, w)];
            t[x] = (uint16_t)acc;
        }
    }
    for (int y = 0; y < h; ++y)
    {
        const uint16_t* rows[7];
        for (int k = -3; k <= 3; ++k) rows[k + 3] = tmp + (size_t)reflect101(y + k, h) * w;
        uint8_t* d = dst + (size_t)y * dpitch;
        for (int x = 0; x < w; ++x)
        {
            int acc = 0;
            for (int k = 0; k < 7; ++k) acc += gk7[k] * rows[k][x];
            d[x] = (uint8_t)((acc + (1 << 15)) >> 16);
        }
    }
    free(tmp);
}

void orc_descriptor_blurred(const uint8_t* blurred, int pitch, int x, int y, float angle_deg, uint64_t out[4])
{
    float a, b;
    orc_sincos_deg(angle_deg, &b, &a); /* a = cos, b = sin */
    out[0] = out[1] = out[2] = out[3] = 0;
    const uint8_t* c = blurred + (size_t)y * pitch + x;
    for (int i = 0; i < 256; ++i)
    {
        int t[2];
        for (int k = 0; k < 2; ++k)
        {
            float px = (float)orc_brief_pattern[4 * i + 2 * k], py = (float)orc_brief_pattern[4 * i + 2 * k + 1];
            int ry   = (int)lrintf(px * b + py * a);
            int rx   = (int)lrintf(px * a - py * b);
            t[k]     = c[ry * pitch + rx];
        }
        if (t[0] < t[1]) out[i >> 6] |= (uint64_t)1 << (i & 63);
    }
}

/* [ORB-SLAM2] computeOrbDescriptor on the blurred level image; bit b = (t0 < t1) of pair b. */
void orc_descriptor(const uint8_t* img, int w, int h, int pitch, int x, int y, float angle_deg, uint64_t out[4])
{
    float a, b;
    orc_sincos_deg(angle_deg, &b, &a); /* a = cos, b = sin */
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int i = 0; i < 256; ++i)
    {
        int t[2];
        for (int k = 0; k < 2; ++k)
        {
            float px = (float)orc_brief_pattern[4 * i + 2 * k], py = (float)orc_brief_pattern[4 * i + 2 * k + 1];
            int ry   = (int)lrintf(px * b + py * a);
            int rx   = (int)lrintf(px * a - py * b);
            t[k]     = orc_blur_at(img, w, h, pitch, x + rx, y + ry);
        }
        if (t[0] < t[1]) out[i >> 6] |= (uint64_t)1 << (i & 63);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Harris response ("orb.response" = 1)                                                           */
/* ---------------------------------------------------------------------------------------------- */
/* north_star names the Harris score among the extractor's steps; published ORB-SLAM2 ranks by the FAST score, OpenCV's ORB
 * (cv::ORB, HARRIS_SCORE: the default) by the Harris response of the FAST corners, and which of the two saiga's extractor
 * uses cannot be read here (reference Snake/Preprocess/FeatureDetector.cpp:31-41 shows the constructor call only).  So it is a
 * definition switch: "orb.response" 0 = FAST score (default, [ORB-SLAM2]), 1 = Harris.
 * [OpenCV] HarrisResponses (modules/features2d/src/orb.cpp), restated from the published algorithm: over the 7 x 7 block
 * around the corner, with the 3 x 3 Sobel derivatives
 *     Ix = 2 (p[y][x+1] - p[y][x-1]) + (p[y-1][x+1] - p[y-1][x-1]) + (p[y+1][x+1] - p[y+1][x-1])      (Iy likewise)
 * a = sum Ix^2, b = sum Iy^2, c = sum Ix Iy (exact integers, < 2^26) and, in float with every operation rounded in this order,
 *     response = (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale^4,   scale = 1 / (4 * 7 * 255)
 * [DEFINED] what the response is used for: the FAST stage is unchanged (corners, non-maximum suppression, the two thresholds
 * and the per-cell / per-level candidate budgets are all on the FAST score -- OpenCV likewise retains by FAST response before
 * it computes Harris); the point kept in a quadtree node is the one with the highest Harris response (then smallest y, x), and
 * KeyPoint::response is the Harris response. */
void orc_harris_abc(const uint8_t* img, int pitch, int x, int y, int32_t* a, int32_t* b, int32_t* c)
{
    int32_t sa = 0, sb = 0, sc = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx)
        {
            const uint8_t* p = img + (size_t)(y + dy) * pitch + (x + dx);
            int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-pitch + 1] - (int)p[-pitch - 1]) + ((int)p[pitch + 1] - (int)p[pitch - 1]);
            int Iy = ((int)p[pitch] - (int)p[-pitch]) * 2 + ((int)p[pitch - 1] - (int)p[-pitch - 1]) + ((int)p[pitch + 1] - (int)p[-pitch + 1]);
            sa += Ix * Ix;
            sb += Iy * Iy;
            sc += Ix * Iy;
        }
    *a = sa;
    *b = sb;
    *c = sc;
}

float orc_harris_response(const uint8_t* img, int pitch, int x, int y)
{
    int32_t a, b, c;
    orc_harris_abc(img, pitch, x, y, &a, &b, &c);
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float s4 = 0x1.bb9da2p-52f; /* ((1 / 7140) ^ 4 in float, multiplied left to right) */
    const float det = fa * fb - fc * fc;
    const float tr  = fa + fb;
    const float kt  = 0.04f * tr * tr;
    return (det - kt) * s4;
}

/* order-preserving map of a float onto uint32 (no NaN occurs: a, b, c are finite) */
uint32_t orc_harris_rank(float r)
{
    uint32_t u;
    memcpy(&u, &r, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static int g_def_orb_response = 0;
int orc_orb_set_response(int v)
{
    if (v != 0 && v != 1) return 1;
    g_def_orb_response = v;
    return 0;
}

/* ---------------------------------------------------------------------------------------------- */
/* whole extractor                                                                                */
/* ---------------------------------------------------------------------------------------------- */
int orc_orb_pyramid(const orc_orb_params* p, const uint8_t* img, int w, int h, int pitch, uint8_t** levels /* malloc'd */,
                    orc_orb_layout_t* L)
{
    if (orc_orb_layout(p, w, h, L) != 0) return -1;
    levels[0] = (uint8_t*)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y) memcpy(levels[0] + (size_t)y * w, img + (size_t)y * pitch, (size_t)w);
    for (int l = 1; l < L->n_levels; ++l)
    {
        levels[l] = (uint8_t*)malloc((size_t)L->w[l] * L->h[l] + 1);
        orc_resize(levels[l - 1], L->w[l - 1], L->h[l - 1], L->w[l - 1], levels[l], L->w[l], L->h[l], L->w[l]);
    }
    return 0;
}

int orc_orb_detect(const orc_orb_params* p, const uint8_t* img, int w, int h, int pitch, orc_keypoint* kps,
                   uint64_t (*desc)[4], int capacity, int level_cap, int threads)
{
    orc_orb_layout_t L;
    uint8_t* levels[ORC_MAX_LEVELS];
    if (orc_orb_pyramid(p, img, w, h, pitch, levels, &L) != 0) return -1;
    if (level_cap <= 0) level_cap = ORC_LEVEL_CAP;

    orc_keypoint* lk[ORC_MAX_LEVELS];
    uint64_t(*ld[ORC_MAX_LEVELS])[4];
    int ln[ORC_MAX_LEVELS];
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) if (threads > 1)
    for (int l = 0; l < L.n_levels; ++l)
    {
        int lw = L.w[l], lh = L.h[l];
        orc_cand* cand = (orc_cand*)malloc(sizeof(orc_cand) * (size_t)level_cap);
        int nc         = orc_orb_candidates(levels[l], lw, lh, lw, p->ini_th, p->min_th, cand, level_cap);
        int* sel       = (int*)malloc(sizeof(int) * (size_t)(orc_orb_distribute_bound(lw, lh, L.nfeat[l]) + 8));
        uint32_t* rank = NULL;
        float* hres    = NULL;
        if (g_def_orb_response == 1)
        {
            rank = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(nc + 1));
            hres = (float*)malloc(sizeof(float) * (size_t)(nc + 1));
            for (int i = 0; i < nc; ++i)
            {
                hres[i] = orc_harris_response(levels[l], lw, cand[i].x, cand[i].y);
                rank[i] = orc_harris_rank(hres[i]);
            }
        }
        int ns         = orc_orb_distribute_ranked(cand, rank, nc, lw, lh, L.nfeat[l], sel);
        uint8_t* blurred = (uint8_t*)malloc((size_t)lw * lh);
        orc_blur_image(levels[l], lw, lh, lw, blurred, lw);
        lk[l]          = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(ns + 1));
        ld[l]          = (uint64_t(*)[4])malloc(32 * (size_t)(ns + 1));
        ln[l]          = ns;
        for (int i = 0; i < ns; ++i)
        {
            const orc_cand* c = &cand[sel[i]];
            int m10, m01;
            orc_ic_moments(levels[l], lw, c->x, c->y, &m10, &m01);
            float angle = orc_fast_atan2((float)m01, (float)m10);
            orc_descriptor_blurred(blurred, lw, c->x, c->y, angle, ld[l][i]);
            lk[l][i].x        = (float)c->x * L.scale[l];
            lk[l][i].y        = (float)c->y * L.scale[l];
            lk[l][i].size     = (float)PATCH_SIZE * L.scale[l];
            lk[l][i].angle    = angle;
            lk[l][i].response = hres ? hres[sel[i]] : (float)(c->score - 1);
            lk[l][i].octave   = l;
        }
        free(cand);
        free(sel);
        free(blurred);
        free(rank);
        free(hres);
    }
    int n = 0, overflow = 0;
    for (int l = 0; l < L.n_levels; ++l)
    {
        for (int i = 0; i < ln[l]; ++i)
        {
            if (n < capacity)
            {
                kps[n] = lk[l][i];
                memcpy(desc[n], ld[l][i], 32);
                n++;
            }
            else
                overflow = 1;
        }
        free(lk[l]);
        free(ld[l]);
        free(levels[l]);
    }
    return overflow ? -2 : n;
}
