/* TEST INFRASTRUCTURE ONLY -- see goicp_oracle.h for the role and parity status of this file.
 *
 * Every function names the reference lines (relative to /root/reference) whose behaviour it restates.
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile); no FMA, float expressions evaluated in float.
 */
#include "goicp_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>

#define ORC_PI      3.1415926536     /* jly_goicp.h:44 */
#define ORC_SQRT3   1.732050808      /* jly_goicp.h:45 */
#define ORC_MAXLVL  20               /* jly_goicp.h:95 */
#define DT_INF      32767            /* jly_3ddt.h:20 */
#define NBINS       41

/* colour codes that own an entry in the (identity) compatibility map, jly_goicp.cpp:66-73 */
static int known_prop(int c) {
    return c == 8204959 || c == 30894 || c == 15219528 || c == 15231913 || c == 4646984 || c == 16741671 ||
           c == 7566712 || c == 0;
}

typedef struct { float x, y, z, w, ub, lb; } tnode;
typedef struct { float a, b, c, w, ub, lb; int l; } rnode;

/* ---------------------------------------------------------------------------------------------
 * Priority queues.  The reference uses std::priority_queue with operator< of jly_goicp.h:64-87
 * (lower lb first, ties -> larger w first).  Nodes that tie on (lb,w) are common (lb==0 at shallow
 * levels), so the pop order among equals matters; it is fixed by the binary-heap algorithm of the C++
 * runtime (sift the hole down to a leaf choosing the not-smaller child, then sift the value up), which
 * is restated here so that equal-key nodes leave the queue in the same order.
 * ------------------------------------------------------------------------------------------- */
static int t_lower(const tnode* p, const tnode* q) { return p->lb != q->lb ? p->lb > q->lb : p->w < q->w; }
static int r_lower(const rnode* p, const rnode* q) { return p->lb != q->lb ? p->lb > q->lb : p->w < q->w; }

#define HEAP_IMPL(NAME, T, LOWER)                                                            \
    typedef struct { T* a; long n, cap; } NAME;                                              \
    static void NAME##_up(T* a, long hole, long top, T v) {                                  \
        long par = (hole - 1) / 2;                                                           \
        while (hole > top && LOWER(&a[par], &v)) { a[hole] = a[par]; hole = par; par = (hole - 1) / 2; } \
        a[hole] = v;                                                                         \
    }                                                                                        \
    static void NAME##_push(NAME* h, T v) {                                                  \
        if (h->n == h->cap) { h->cap = h->cap ? 2 * h->cap : 64; h->a = (T*)realloc(h->a, sizeof(T) * h->cap); } \
        h->n++;                                                                              \
        NAME##_up(h->a, h->n - 1, 0, v);                                                     \
    }                                                                                        \
    static void NAME##_pop(NAME* h) {                                                        \
        if (h->n > 1) {                                                                      \
            T* a = h->a; long len = h->n - 1; T v = a[len]; a[len] = a[0];                   \
            long hole = 0, child = 0;                                                        \
            while (child < (len - 1) / 2) {                                                  \
                child = 2 * (child + 1);                                                     \
                if (LOWER(&a[child], &a[child - 1])) child--;                                \
                a[hole] = a[child]; hole = child;                                            \
            }                                                                                \
            if ((len & 1) == 0 && child == (len - 2) / 2) {                                  \
                child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1;           \
            }                                                                                \
            NAME##_up(a, hole, 0, v);                                                        \
        }                                                                                    \
        h->n--;                                                                              \
    }

HEAP_IMPL(theap, tnode, t_lower)
HEAP_IMPL(rheap, rnode, r_lower)

/* --------------------------------------------------------------------------------------------- */
struct orc_ctx {
    /* config (readConfig, jly_main.cpp:231-264) */
    float MSEThresh, trimFraction, reg, regN, regFPFH;
    int cfpfh, norm, ponderation, doTrim;
    rnode initRot; tnode initTrans;
    /* clouds */
    int Nm, Nd, NdAlloc;
    float *mxyz, *dxyz, *mfpfh, *dfpfh; int *mcol, *dcol;
    /* distance transform + cell tables */
    int S; double xMin, xMax, yMin, yMax, zMin, zMax, scale, expandFactor;
    float* dist; short* vhd; int* nearest3; int* cellc; int* cell_ptr; int* cell_idx;
    /* Initialize() products */
    int initialized;
    float *normData, *minDis, *maxRotDis, *weights, *rot; int* neighbors;
    int inlierNum; float SSEThresh;
    /* search state */
    float optError; double optR[9], optT[3]; int optComp;
    rnode optNodeRot; tnode optNodeTrans;
    int* icp_pts; double* icp_dis;   /* (id_data,id_model) + dis of the last ICP3D run */
    int* opt_pts;                    /* those of the last accepted run (icp3d.optPoints) */
    /* instrumentation */
    long long n_tpop, n_rpop, n_cube, n_corner;
    long long pop_budget;            /* 0: none; else InnerBnB stops after this many queue pops (orc_set_pop_budget) */
    float* trace; long long trace_n, trace_cap; int trace_on;
    char* log; size_t log_n, log_cap;
};

static void logf_(orc_ctx* c, const char* fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); int n = vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (n < 0) return;
    if (c->log_n + (size_t)n + 1 > c->log_cap) { c->log_cap = 2 * (c->log_cap + n + 256); c->log = (char*)realloc(c->log, c->log_cap); }
    memcpy(c->log + c->log_n, buf, (size_t)n + 1); c->log_n += (size_t)n;
}

orc_ctx* orc_new(void) {
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
    /* GoICP::GoICP(), jly_goicp.cpp:44-56 */
    c->initRot.a = c->initRot.b = c->initRot.c = (float)-ORC_PI; c->initRot.w = (float)(2 * ORC_PI); c->initRot.l = 0;
    c->doTrim = 1;
    return c;
}

static void free_init(orc_ctx* c) {
    free(c->normData); free(c->minDis); free(c->maxRotDis); free(c->weights); free(c->rot); free(c->neighbors);
    free(c->icp_pts); free(c->icp_dis); free(c->opt_pts);
    c->normData = c->minDis = c->maxRotDis = c->weights = c->rot = NULL; c->neighbors = NULL;
    c->icp_pts = c->opt_pts = NULL; c->icp_dis = NULL; c->initialized = 0;
}
static void free_dt(orc_ctx* c) {
    free(c->dist); free(c->vhd); free(c->nearest3); free(c->cellc); free(c->cell_ptr); free(c->cell_idx);
    c->dist = NULL; c->vhd = NULL; c->nearest3 = c->cellc = c->cell_ptr = c->cell_idx = NULL;
}
void orc_free(orc_ctx* c) {
    free_init(c); free_dt(c);
    free(c->mxyz); free(c->dxyz); free(c->mfpfh); free(c->dfpfh); free(c->mcol); free(c->dcol);
    free(c->trace); free(c->log); free(c);
}

void orc_set_config(orc_ctx* c, const double* v) {
    c->MSEThresh = (float)v[0];
    c->initRot.a = (float)v[1]; c->initRot.b = (float)v[2]; c->initRot.c = (float)v[3]; c->initRot.w = (float)v[4];
    c->initTrans.x = (float)v[5]; c->initTrans.y = (float)v[6]; c->initTrans.z = (float)v[7]; c->initTrans.w = (float)v[8];
    c->trimFraction = (float)v[9];
    c->reg = (float)v[10]; c->regN = (float)v[11]; c->regFPFH = (float)v[12];
    c->cfpfh = (int)v[13]; c->norm = (int)v[14]; c->ponderation = (int)v[15];
    c->doTrim = 1;
    if (c->trimFraction < 0.001) c->doTrim = 0;       /* jly_main.cpp:259-262 (double literal compare) */
    c->S = (int)v[16]; c->expandFactor = v[17];
}

static void* dup_(const void* p, size_t n) { void* q = malloc(n ? n : 1); if (p) memcpy(q, p, n); else memset(q, 0, n); return q; }

void orc_set_clouds(orc_ctx* c, int Nm, const float* mxyz, const int* mcol, const float* mfpfh,
                    int Nd, const float* dxyz, const int* dcol, const float* dfpfh) {
    free(c->mxyz); free(c->dxyz); free(c->mfpfh); free(c->dfpfh); free(c->mcol); free(c->dcol);
    c->Nm = Nm; c->Nd = c->NdAlloc = Nd;
    c->mxyz = (float*)dup_(mxyz, sizeof(float) * 3 * Nm); c->dxyz = (float*)dup_(dxyz, sizeof(float) * 3 * Nd);
    c->mcol = (int*)dup_(mcol, sizeof(int) * Nm);          c->dcol = (int*)dup_(dcol, sizeof(int) * Nd);
    c->mfpfh = (float*)dup_(mfpfh, sizeof(float) * NBINS * Nm); c->dfpfh = (float*)dup_(dfpfh, sizeof(float) * NBINS * Nd);
}
void orc_set_nd(orc_ctx* c, int Nd) { c->Nd = Nd; }

/* =============================================================================================
 * Distance transform
 * =========================================================================================== */
#define CELL(c_, z, y, x) ((((size_t)(z)) * (c_)->S + (y)) * (c_)->S + (x))

/* One neighbour of a propagation mask: where to read, what to add, and which bounds guard it.
 * (g* is the guard offset; it equals the read offset except for two entries of the reference that read
 *  [z+1][y][x] under a y+1 guard -- jly_3ddt.cpp:88-93 and :445-450 -- kept as they are.) */
typedef struct { signed char dz, dy, dx, id, ih, iv, gz, gy, gx; } mask_t;

/* jly_3ddt.cpp:508-712 */
static const mask_t F1[14] = {
    {-1,-1,-1, 1,1,1, -1,-1,-1}, {0,-1,-1, 0,1,1, 0,-1,-1}, {1,-1,-1, 1,1,1, 1,-1,-1},
    {-1,0,-1, 1,0,1, -1,0,-1},   {0,0,-1, 0,0,1, 0,0,-1},   {1,0,-1, 1,0,1, 1,0,-1},
    {-1,1,-1, 1,1,1, -1,1,-1},   {0,1,-1, 0,1,1, 0,1,-1},   {1,1,-1, 1,1,1, 1,1,-1},
    {-1,-1,0, 1,1,0, -1,-1,0},   {0,-1,0, 0,1,0, 0,-1,0},   {1,-1,0, 1,1,0, 1,-1,0},
    {0,0,0, 0,0,0, 0,0,0},       {-1,0,0, 1,0,0, -1,0,0} };
/* jly_3ddt.cpp:260-293 */
static const mask_t F2[2] = { {1,0,0, 1,0,0, 1,0,0}, {0,0,0, 0,0,0, 0,0,0} };
/* jly_3ddt.cpp:57-137 */
static const mask_t F3[5] = { {1,0,0, 1,0,0, 1,0,0}, {1,0,0, 1,1,0, 1,1,0}, {0,1,0, 0,1,0, 0,1,0},
                              {0,0,0, 0,0,0, 0,0,0}, {-1,1,0, 1,1,0, -1,1,0} };
/* jly_3ddt.cpp:139-175 */
static const mask_t F4[2] = { {-1,0,0, 1,0,0, -1,0,0}, {0,0,0, 0,0,0, 0,0,0} };
/* jly_3ddt.cpp:301-506 */
static const mask_t B1[14] = {
    {-1,-1,1, 1,1,1, -1,-1,1}, {0,-1,1, 0,1,1, 0,-1,1}, {1,-1,1, 1,1,1, 1,-1,1},
    {-1,0,1, 1,0,1, -1,0,1},   {0,0,1, 0,0,1, 0,0,1},   {1,0,1, 1,0,1, 1,0,1},
    {-1,1,1, 1,1,1, -1,1,1},   {0,1,1, 0,1,1, 0,1,1},   {1,1,1, 1,1,1, 1,1,1},
    {1,0,0, 1,0,0, 1,0,0},     {1,0,0, 1,1,0, 1,1,0},   {0,1,0, 0,1,0, 0,1,0},
    {0,0,0, 0,0,0, 0,0,0},     {-1,1,0, 1,1,0, -1,1,0} };
/* jly_3ddt.cpp:177-258 */
static const mask_t B3[5] = { {-1,-1,0, 1,1,0, -1,-1,0}, {0,-1,0, 0,1,0, 0,-1,0}, {1,-1,0, 1,1,0, 1,-1,0},
                              {0,0,0, 0,0,0, 0,0,0}, {-1,0,0, 1,0,0, -1,0,0} };

typedef struct { short v, h, d; float distance; } dcell;

/* 16-bit wrap of the `short = short + 1` stores and 32-bit wrap of the int sum of squares
 * (what the compiled reference does on x86-64; jly_3ddt.cpp:78-79 and siblings). */
static float mask_len(short v, short h, short d) {
    unsigned s = (unsigned)((int)v * v) + (unsigned)((int)h * h) + (unsigned)((int)d * d);
    return (float)sqrt((double)(int)s);     /* NaN for wrapped-negative sums: never selected */
}

static void propagate(orc_ctx* c, dcell* A, const mask_t* m, int nm, int z, int y, int x) {
    int S = c->S;
    dcell best; best.v = best.h = best.d = DT_INF; best.distance = DT_INF;
    /* NB: when no candidate wins, the reference returns an uninitialised v,h,d (jly_3ddt.cpp:70-73);
       this restatement returns the "infinite" cell instead (the behaviour of the -O3 reference build). */
    for (int k = 0; k < nm; k++) {
        int zz = z + m[k].gz, yy = y + m[k].gy, xx = x + m[k].gx;
        if (zz < 0 || zz >= S || yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
        const dcell* s = &A[CELL(c, z + m[k].dz, y + m[k].dy, x + m[k].dx)];
        dcell cand;
        cand.v = (short)(s->v + m[k].iv); cand.h = (short)(s->h + m[k].ih); cand.d = (short)(s->d + m[k].id);
        cand.distance = mask_len(cand.v, cand.h, cand.d);
        if (cand.distance < best.distance) best = cand;
    }
    A[CELL(c, z, y, x)] = best;
}

/* DEuclidean, jly_3ddt.cpp:716-750 */
static void sweep_all(orc_ctx* c, dcell* A) {
    int S = c->S, x, y, z;
    for (x = 0; x < S; x++) {
        for (y = 0; y < S; y++) {
            for (z = 0; z < S; z++) propagate(c, A, F1, 14, z, y, x);
            for (z = S - 1; z > -1; z--) propagate(c, A, F2, 2, z, y, x);
        }
        for (y = S - 1; y > -1; y--) {
            for (z = S - 1; z > -1; z--) propagate(c, A, F3, 5, z, y, x);
            for (z = 0; z < S; z++) propagate(c, A, F4, 2, z, y, x);
        }
    }
    for (x = S - 1; x > -1; x--) {
        for (y = S - 1; y > -1; y--) {
            for (z = S - 1; z > -1; z--) propagate(c, A, B1, 14, z, y, x);
            for (z = 0; z < S; z++) propagate(c, A, F4, 2, z, y, x);
        }
        for (y = 0; y < S; y++) {
            for (z = 0; z < S; z++) propagate(c, A, B3, 5, z, y, x);
            for (z = S - 1; z > -1; z--) propagate(c, A, F2, 2, z, y, x);
        }
    }
}

/* Nearest occupied cell guessed from |v|,|h|,|d|: sign combinations tried in the reference's fixed
 * order until a zero-distance cell is hit (jly_3ddt.cpp:1012-1132). Sign triples are (sx,sy,sz). */
static const signed char SGN3[8][3] = { {1,1,1}, {-1,1,1}, {1,-1,1}, {1,1,-1}, {-1,-1,1}, {-1,1,-1}, {1,-1,-1}, {-1,-1,-1} };
static const signed char SGN_YZ[4][3] = { {0,1,1}, {0,1,-1}, {0,-1,1}, {0,-1,-1} };   /* xD==0 */
static const signed char SGN_XZ[4][3] = { {1,0,1}, {1,0,-1}, {-1,0,1}, {-1,0,-1} };   /* yD==0 */
static const signed char SGN_XY[4][3] = { {1,1,0}, {-1,1,0}, {1,-1,0}, {-1,-1,0} };   /* zD==0 */
static const signed char SGN_X[2][3] = { {1,0,0}, {-1,0,0} };
static const signed char SGN_Y[2][3] = { {0,1,0}, {0,-1,0} };
static const signed char SGN_Z[2][3] = { {0,0,1}, {0,0,-1} };

static void nearest_occupied(orc_ctx* c, const dcell* A, int z, int y, int x, int* out3) {
    int S = c->S;
    const dcell* me = &A[CELL(c, z, y, x)];
    out3[0] = x; out3[1] = y; out3[2] = z;
    if (me->distance == 0) return;
    int xD = me->v, yD = me->h, zD = me->d;
    const signed char (*tab)[3]; int nt;
    if (xD == 0) {
        if (yD == 0) { tab = SGN_Z; nt = 2; } else if (zD == 0) { tab = SGN_Y; nt = 2; } else { tab = SGN_YZ; nt = 4; }
    } else if (yD == 0) {
        if (zD == 0) { tab = SGN_X; nt = 2; } else { tab = SGN_XZ; nt = 4; }
    } else if (zD == 0) { tab = SGN_XY; nt = 4; }
    else { tab = SGN3; nt = 8; }
    for (int k = 0; k < nt; k++) {
        int xx = x + tab[k][0] * xD, yy = y + tab[k][1] * yD, zz = z + tab[k][2] * zD;
        /* the reference only bounds-checks the side it steps towards: +: < dim, -: > -1 */
        if ((tab[k][0] > 0 && !(xx < S)) || (tab[k][0] < 0 && !(xx > -1))) continue;
        if ((tab[k][1] > 0 && !(yy < S)) || (tab[k][1] < 0 && !(yy > -1))) continue;
        if ((tab[k][2] > 0 && !(zz < S)) || (tab[k][2] < 0 && !(zz > -1))) continue;
        if (xx < 0 || xx >= S || yy < 0 || yy >= S || zz < 0 || zz >= S) continue;   /* (negative offsets never occur without UB) */
        if (A[CELL(c, zz, yy, xx)].distance == 0) { out3[0] = xx; out3[1] = yy; out3[2] = zz; return; }
    }
}

/* DT3D::Build (jly_3ddt.cpp:897-1137) followed by GoICP::assignCellColor (jly_goicp.cpp:951-969) */
void orc_build_dt(orc_ctx* c) {
    int S = c->S, Nm = c->Nm, i;
    size_t ncell = (size_t)S * S * S;
    double ef = c->expandFactor;
    double xMin, xMax, yMin, yMax, zMin, zMax;
    xMin = xMax = c->mxyz[0]; yMin = yMax = c->mxyz[1]; zMin = zMax = c->mxyz[2];
    for (i = 1; i < Nm; i++) {
        double x = c->mxyz[3*i], y = c->mxyz[3*i+1], z = c->mxyz[3*i+2];
        if (xMin > x) xMin = x; if (xMax < x) xMax = x;
        if (yMin > y) yMin = y; if (yMax < y) yMax = y;
        if (zMin > z) zMin = z; if (zMax < z) zMax = z;
    }
    double xC = (xMin + xMax) / 2, yC = (yMin + yMax) / 2, zC = (zMin + zMax) / 2;
    xMin = xC - ef * (xMax - xC); xMax = xC + ef * (xMax - xC);
    yMin = yC - ef * (yMax - yC); yMax = yC + ef * (yMax - yC);
    zMin = zC - ef * (zMax - zC); zMax = zC + ef * (zMax - zC);
    double mx = xMax - xMin > yMax - yMin ? xMax - xMin : yMax - yMin;
    mx = mx > zMax - zMin ? mx : zMax - zMin;
    c->xMin = xC - mx / 2; c->xMax = xC + mx / 2;
    c->yMin = yC - mx / 2; c->yMax = yC + mx / 2;
    c->zMin = zC - mx / 2; c->zMax = zC + mx / 2;
    c->scale = S / mx;

    free_dt(c);
    dcell* A = (dcell*)malloc(sizeof(dcell) * ncell);
    for (size_t k = 0; k < ncell; k++) { A[k].v = A[k].h = A[k].d = DT_INF; A[k].distance = DT_INF; }
    c->cellc = (int*)malloc(sizeof(int) * ncell);
    c->cell_ptr = (int*)calloc(ncell + 1, sizeof(int));
    c->cell_idx = (int*)malloc(sizeof(int) * (Nm ? Nm : 1));
    for (size_t k = 0; k < ncell; k++) c->cellc[k] = -2;
    int* cell_of = (int*)malloc(sizeof(int) * (Nm ? Nm : 1));
    for (i = 0; i < Nm; i++) {
        int x = (int)(((double)c->mxyz[3*i]   - c->xMin) * c->scale + 0.5);
        int y = (int)(((double)c->mxyz[3*i+1] - c->yMin) * c->scale + 0.5);
        int z = (int)(((double)c->mxyz[3*i+2] - c->zMin) * c->scale + 0.5);
        cell_of[i] = -1;
        if (x < 0 || x >= S || y < 0 || y >= S || z < 0 || z >= S) continue;   /* (reference would write out of bounds) */
        size_t k = CELL(c, z, y, x);
        cell_of[i] = (int)k; c->cellc[k] = -1; c->cell_ptr[k + 1]++;
        A[k].v = A[k].h = A[k].d = 0; A[k].distance = 0;
    }
    for (size_t k = 0; k < ncell; k++) c->cell_ptr[k + 1] += c->cell_ptr[k];
    { int* fill = (int*)calloc(ncell, sizeof(int));
      for (i = 0; i < Nm; i++) if (cell_of[i] >= 0) { int k = cell_of[i]; c->cell_idx[c->cell_ptr[k] + fill[k]++] = i; }
      free(fill); }
    free(cell_of);

    sweep_all(c, A);

    c->dist = (float*)malloc(sizeof(float) * ncell);
    c->vhd = (short*)malloc(sizeof(short) * 3 * ncell);
    c->nearest3 = (int*)malloc(sizeof(int) * 3 * ncell);
    /* distance /= scale happens cell by cell *while* the nearest-cell search reads neighbours' distances
       (only ==0 tests, which the division preserves), jly_3ddt.cpp:1003-1004 */
    for (size_t k = 0; k < ncell; k++) {
        float d = (float)((double)A[k].distance / c->scale);
        if (d < 0) d = 0;
        A[k].distance = d;
    }
    for (int z = 0; z < S; z++) for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) {
        size_t k = CELL(c, z, y, x);
        c->dist[k] = A[k].distance;
        c->vhd[3*k] = A[k].v; c->vhd[3*k+1] = A[k].h; c->vhd[3*k+2] = A[k].d;
        nearest_occupied(c, A, z, y, x, &c->nearest3[3*k]);
    }
    free(A);
    /* assignCellColor: single colour of the cell's points, or -1 if mixed */
    for (size_t k = 0; k < ncell; k++) if (c->cellc[k] == -1) {
        int p0 = c->cell_ptr[k], p1 = c->cell_ptr[k + 1];
        int prop = c->mcol[c->cell_idx[p0]], mixed = 0;
        for (int p = p0 + 1; p < p1; p++) if (c->mcol[c->cell_idx[p]] != prop) { mixed = 1; break; }
        c->cellc[k] = mixed ? -1 : prop;
    }
}

void orc_load_dt(orc_ctx* c, int S, const double* g, const float* dist, const int* nearest3,
                 const int* cellc, const int* cell_ptr, const int* cell_idx) {
    free_dt(c);
    size_t n = (size_t)S * S * S;
    c->S = S; c->xMin = g[0]; c->yMin = g[1]; c->zMin = g[2]; c->scale = g[3];
    c->xMax = c->xMin + S / c->scale; c->yMax = c->yMin + S / c->scale; c->zMax = c->zMin + S / c->scale;
    c->dist = (float*)dup_(dist, sizeof(float) * n);
    if (nearest3) c->nearest3 = (int*)dup_(nearest3, sizeof(int) * 3 * n);
    if (cellc) c->cellc = (int*)dup_(cellc, sizeof(int) * n);
    if (cell_ptr) { c->cell_ptr = (int*)dup_(cell_ptr, sizeof(int) * (n + 1)); c->cell_idx = (int*)dup_(cell_idx, sizeof(int) * (cell_ptr[n] ? cell_ptr[n] : 1)); }
}

void orc_dt_geom(orc_ctx* c, double* g) {
    g[0] = c->xMin; g[1] = c->xMax; g[2] = c->yMin; g[3] = c->yMax; g[4] = c->zMin; g[5] = c->zMax; g[6] = c->scale; g[7] = c->S;
}
void orc_dt_dump(orc_ctx* c, float* dist, short* vhd, int* nearest3, int* cellc) {
    size_t n = (size_t)c->S * c->S * c->S;
    if (dist) memcpy(dist, c->dist, sizeof(float) * n);
    if (vhd && c->vhd) memcpy(vhd, c->vhd, sizeof(short) * 3 * n);
    if (nearest3) memcpy(nearest3, c->nearest3, sizeof(int) * 3 * n);
    if (cellc) memcpy(cellc, c->cellc, sizeof(int) * n);
}
int orc_cell_points(orc_ctx* c, int x, int y, int z, int* out, int cap) {
    size_t k = CELL(c, z, y, x); int n = c->cell_ptr[k + 1] - c->cell_ptr[k];
    for (int i = 0; i < n && i < cap; i++) out[i] = c->cell_idx[c->cell_ptr[k] + i];
    return n;
}

/* DT3D::Distance, jly_3ddt.cpp:1139-1191 */
static float dt_distance(const orc_ctx* c, double X, double Y, double Z) {
    int S = c->S;
    int x = (int)((X - c->xMin) * c->scale + 0.5);
    int y = (int)((Y - c->yMin) * c->scale + 0.5);
    int z = (int)((Z - c->zMin) * c->scale + 0.5);
    if (x > -1 && x < S && y > -1 && y < S && z > -1 && z < S) return c->dist[CELL(c, z, y, x)];
    float a = 0, b = 0, cc = 0;
    if (x < 0) { a = (float)x; x = 0; } else if (x >= S) { a = (float)(x - S + 1); x = S - 1; }
    if (y < 0) { b = (float)y; y = 0; } else if (y >= S) { b = (float)(y - S + 1); y = S - 1; }
    if (z < 0) { cc = (float)z; z = 0; } else if (z >= S) { cc = (float)(z - S + 1); z = S - 1; }
    float r = sqrtf(a * a + b * b + cc * cc);                 /* float sqrt overload */
    return (float)((double)r / c->scale + (double)c->dist[CELL(c, z, y, x)]);
}
void orc_distance(orc_ctx* c, int n, const double* q, float* out) {
    for (int i = 0; i < n; i++) out[i] = dt_distance(c, q[3*i], q[3*i+1], q[3*i+2]);
}

/* =============================================================================================
 * Selection (jly_sorting.hpp:34-313): quickselect on median(st,mid,en) pivots that falls back to
 * median-of-medians when 5 partitions fail to halve the range; ranges of <=6 are insertion-sorted.
 * The permutation it leaves behind fixes the float summation order of every trimmed sum, so the exact
 * sequence of swaps is reproduced, including the absolute-index slip in the 3-element median helper.
 * =========================================================================================== */
static size_t med_ends(const float* d, size_t s, size_t e) {
    size_t m = (s + e) / 2;
    if (d[s] < d[e]) return d[m] < d[s] ? s : (d[m] < d[e] ? m : e);
    return d[m] < d[e] ? e : (d[m] < d[s] ? m : s);
}
static size_t med3(const float* d, size_t s) {
    const float* q = d + s;
    if (q[0] < q[2]) return q[1] < q[0] ? s : (q[1] < q[2] ? s + 1 : s + 2);
    /* reference reads data[1], data[2], data[0] of the WHOLE array here (jly_sorting.hpp:74-80) */
    return d[1] < d[2] ? s + 2 : (d[1] < d[0] ? s + 1 : s);
}
static size_t med5(float* d, size_t s) {
    float* q = d + s; float t;
    if (q[0] > q[1]) { t = q[0]; q[0] = q[1]; q[1] = t; }
    if (q[2] > q[3]) { t = q[2]; q[2] = q[3]; q[3] = t; }
    if (q[0] < q[2]) { t = q[4]; q[4] = q[0]; if (t < q[1]) q[0] = t; else { q[0] = q[1]; q[1] = t; } }
    else             { t = q[4]; q[4] = q[2]; if (t < q[3]) q[2] = t; else { q[2] = q[3]; q[3] = t; } }
    if (q[0] < q[2]) return q[1] < q[2] ? s + 1 : s + 2;
    return q[0] < q[3] ? s : s + 3;
}
static size_t med_of_meds(float* d, size_t s, size_t e) {
    size_t len = e - s + 1, groups = len / 5 + (len % 5 != 0), g = s, i, mi; float t;
    for (i = 0; i + 1 < groups; i++, g += 5) { mi = med5(d, g); t = d[s + i]; d[s + i] = d[mi]; d[mi] = t; }
    switch (e - g + 1) { case 3: case 4: mi = med3(d, g); break; case 5: mi = med5(d, g); break; default: mi = g; }
    t = d[s + i]; d[s + i] = d[mi]; d[mi] = t;
    if (groups > 5) return med_of_meds(d, s, s + groups - 1);
    switch (groups) { case 3: case 4: return med3(d, s); case 5: return med5(d, s); default: return s; }
}
static void select_k(float* d, size_t s, size_t e, size_t k) {
    size_t prev = e - s + 1, tries = 0; int quick = 1; float t;
    for (;;) {
        if (s >= e) return;
        if (e - s <= 5) {
            for (size_t i = s + 1; i <= e; i++) for (size_t j = i; j > s && d[j - 1] > d[j]; j--) { t = d[j - 1]; d[j - 1] = d[j]; d[j] = t; }
            return;
        }
        if (quick && tries++ == 5) { size_t len = e - s + 1; if (len * 2 > prev) quick = 0; prev = len; tries = 0; }
        size_t pi = quick ? med_ends(d, s, e) : med_of_meds(d, s, e);
        if (pi != s) { t = d[s]; d[s] = d[pi]; d[pi] = t; }
        float piv = d[s]; size_t lo = s + 1, hi = e;
        for (;;) {
            while (lo < hi && piv >= d[lo]) ++lo;
            while (lo < hi && piv <= d[hi]) --hi;
            if (lo >= hi) break;
            t = d[lo]; d[lo] = d[hi]; d[hi] = t;
        }
        size_t sp = d[lo] < piv ? lo : lo - 1;
        d[s] = d[sp]; d[sp] = piv;
        if (sp < k) s = sp + 1; else if (sp > k) e = sp - 1; else return;
    }
}
void orc_intro_select(float* data, long long n, long long k) { select_k(data, 0, (size_t)(n - 1), (size_t)k); }

/* =============================================================================================
 * Initialize (jly_goicp.cpp:180-267) incl. neighborsWeights (jly_goicp.cpp:1453-1498, 1097-1103)
 * =========================================================================================== */
static void neighbour_weights(orc_ctx* c) {
    int Nd = c->Nd, maxN = 0, minN = 100;
    float radius = 0.035f;
    while (maxN < 19) {            /* (never ends for Nd < 20, like the reference) */
        double lim = (double)sqrtf(radius);
        for (int i = 0; i < Nd; i++) {
            int cnt = 0;
            for (int j = 0; j < Nd; j++) {
                if (j == i) continue;
                double dx = (double)(c->dxyz[3*j] - c->dxyz[3*i]), dy = (double)(c->dxyz[3*j+1] - c->dxyz[3*i+1]), dz = (double)(c->dxyz[3*j+2] - c->dxyz[3*i+2]);
                double dist = sqrt(dx * dx + dy * dy + dz * dz);
                if (dist < lim) cnt++;
            }
            c->neighbors[i] = cnt;
            if (cnt > maxN) maxN = cnt;
            if (cnt < minN) minN = cnt;
        }
        radius = (float)(radius + 0.001);
    }
    if (minN == 0) minN = 1;
    for (int i = 0; i < Nd; i++) {
        if (c->neighbors[i] == 0) c->neighbors[i] = 1;
        float f = ((float)minN / (float)c->neighbors[i]) * 2;
        c->weights[i] += f;
    }
}

void orc_initialize(orc_ctx* c) {
    int Nd = c->Nd, i, j;
    free_init(c);
    c->normData = (float*)malloc(sizeof(float) * Nd);
    for (i = 0; i < Nd; i++) { float x = c->dxyz[3*i], y = c->dxyz[3*i+1], z = c->dxyz[3*i+2]; c->normData[i] = sqrtf(x * x + y * y + z * z); }
    c->maxRotDis = (float*)malloc(sizeof(float) * ORC_MAXLVL * Nd);
    for (i = 0; i < ORC_MAXLVL; i++) {
        float sigma = (float)((double)c->initRot.w / pow(2.0, i) / 2.0);
        float maxAngle = (float)(ORC_SQRT3 * sigma);
        if (maxAngle > ORC_PI) maxAngle = (float)ORC_PI;
        for (j = 0; j < Nd; j++) c->maxRotDis[(size_t)i * Nd + j] = 2 * sinf(maxAngle / 2) * c->normData[j];
    }
    c->minDis = (float*)malloc(sizeof(float) * Nd);
    c->rot = (float*)calloc((size_t)3 * Nd, sizeof(float));
    c->icp_pts = (int*)calloc((size_t)2 * Nd, sizeof(int)); c->icp_dis = (double*)calloc(Nd, sizeof(double));
    c->opt_pts = (int*)calloc((size_t)2 * Nd, sizeof(int));
    c->optNodeRot = c->initRot; c->optNodeTrans = c->initTrans;
    memset(c->optR, 0, sizeof c->optR); c->optR[0] = c->optR[4] = c->optR[8] = 1; memset(c->optT, 0, sizeof c->optT);
    if (c->doTrim) c->inlierNum = (int)(Nd * (1 - c->trimFraction)); else c->inlierNum = Nd;
    c->weights = (float*)malloc(sizeof(float) * Nd); c->neighbors = (int*)calloc(Nd, sizeof(int));
    for (i = 0; i < Nd; i++) c->weights[i] = 1;
    if (c->ponderation == 1) neighbour_weights(c);
    c->SSEThresh = c->MSEThresh * c->inlierNum;
    c->initialized = 1;
}

void  orc_get_weights(orc_ctx* c, float* w) { memcpy(w, c->weights, sizeof(float) * c->Nd); }
void  orc_set_weights(orc_ctx* c, const float* w) { memcpy(c->weights, w, sizeof(float) * c->Nd); }
void  orc_get_maxrotdis(orc_ctx* c, float* o) { memcpy(o, c->maxRotDis, sizeof(float) * ORC_MAXLVL * c->Nd); }
int   orc_inlier_num(orc_ctx* c) { return c->inlierNum; }
float orc_sse_thresh(orc_ctx* c) { return c->SSEThresh; }
void  orc_set_sse_thresh(orc_ctx* c, float v) { c->SSEThresh = v; }
void  orc_set_opt_error(orc_ctx* c, float e) { c->optError = e; }
float orc_get_opt_error(orc_ctx* c) { return c->optError; }
void  orc_set_init_trans(orc_ctx* c, float x, float y, float z, float w, float lb) {
    c->initTrans.x = x; c->initTrans.y = y; c->initTrans.z = z; c->initTrans.w = w; c->initTrans.lb = lb;
}
void orc_set_rotated(orc_ctx* c, const float* xyz) { memcpy(c->rot, xyz, sizeof(float) * 3 * c->Nd); }

/* jly_goicp.cpp:750-756 */
void orc_rotate(orc_ctx* c, const float* R) {
    for (int i = 0; i < c->Nd; i++) {
        float x = c->dxyz[3*i], y = c->dxyz[3*i+1], z = c->dxyz[3*i+2];
        c->rot[3*i]   = R[0] * x + R[1] * y + R[2] * z;
        c->rot[3*i+1] = R[3] * x + R[4] * y + R[5] * z;
        c->rot[3*i+2] = R[6] * x + R[7] * y + R[8] * z;
    }
}

/* jly_goicp.cpp:718-747 (float throughout) */
void orc_axis_angle(float v1, float v2, float v3, float* R) {
    float t = sqrtf(v1 * v1 + v2 * v2 + v3 * v3);
    if (!(t > 0)) { R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0; return; }
    v1 /= t; v2 /= t; v3 /= t;
    float ct = cosf(t), ct2 = 1 - ct, st = sinf(t);
    float t121 = v1 * v2 * ct2, t122 = v3 * st, t131 = v1 * v3 * ct2, t132 = v2 * st, t231 = v2 * v3 * ct2, t232 = v1 * st;
    R[0] = ct + v1 * v1 * ct2; R[1] = t121 - t122;        R[2] = t131 + t132;
    R[3] = t121 + t122;        R[4] = ct + v2 * v2 * ct2; R[5] = t231 - t232;
    R[6] = t131 - t132;        R[7] = t231 + t232;        R[8] = ct + v3 * v3 * ct2;
}

/* =============================================================================================
 * Corner penalty terms
 * =========================================================================================== */
static size_t occ_cell(const orc_ctx* c, float x, float y, float z) {
    /* checkCompatibility / computeFPFHDifference prologue, jly_goicp.cpp:974-988, 1642-1654 */
    int S = c->S;
    int rx = (int)(((double)x - c->xMin) * c->scale + 0.5);
    int ry = (int)(((double)y - c->yMin) * c->scale + 0.5);
    int rz = (int)(((double)z - c->zMin) * c->scale + 0.5);
    if (rx < 0) rx = 0; if (rx >= S) rx = S - 1;
    if (ry < 0) ry = 0; if (ry >= S) ry = S - 1;
    if (rz < 0) rz = 0; if (rz >= S) rz = S - 1;
    const int* n3 = &c->nearest3[3 * CELL(c, rz, ry, rx)];
    return CELL(c, n3[2], n3[1], n3[0]);
}

/* checkCompatibility -> checkProperty, jly_goicp.cpp:974-1000, 1068-1092 */
static int compatible_at(const orc_ctx* c, int i, float x, float y, float z) {
    size_t k = occ_cell(c, x, y, z);
    int src = c->dcol[i], tgt = c->cellc[k];
    if (tgt == -1) {
        for (int p = c->cell_ptr[k]; p < c->cell_ptr[k + 1]; p++) if (c->mcol[c->cell_idx[p]] == src) return 1;
        return 0;
    }
    return known_prop(src) && src == tgt;
}

/* checkCompatibilities, jly_goicp.cpp:919-928 */
int orc_check_compatibilities(orc_ctx* c, float x, float y, float z) {
    int bad = 0;
    for (int i = 0; i < c->Nd; i++) if (!compatible_at(c, i, c->rot[3*i] + x, c->rot[3*i+1] + y, c->rot[3*i+2] + z)) bad++;
    c->n_corner++;
    return bad;
}

static void fpfh_range(const orc_ctx* c, int* b0, int* b1) {
    *b0 = 0; *b1 = 0;
    if (c->cfpfh == 1) { *b0 = 0; *b1 = 41; } else if (c->cfpfh == 2) { *b0 = 0; *b1 = 33; } else if (c->cfpfh == 3) { *b0 = 33; *b1 = 41; }
}

/* sumFPFH -> computeFPFHDifference(false,...), jly_goicp.cpp:1689-1697, 1642-1681 */
float orc_sum_fpfh(orc_ctx* c, float x, float y, float z) {
    int b0, b1; fpfh_range(c, &b0, &b1);
    float sum = 0;
    for (int i = 0; i < c->Nd; i++) {
        size_t k = occ_cell(c, c->rot[3*i] + x, c->rot[3*i+1] + y, c->rot[3*i+2] + z);
        float minD = 1000000000;
        for (int p = c->cell_ptr[k]; p < c->cell_ptr[k + 1]; p++) {
            const float* f = &c->dfpfh[(size_t)NBINS * i]; const float* g = &c->mfpfh[(size_t)NBINS * c->cell_idx[p]];
            float diff = 0;
            for (int b = b0; b < b1; b++) diff += fabsf(f[b] - g[b]);
            if (diff < minD) minD = diff;
        }
        sum += minD;
    }
    c->n_corner++;
    return sum / c->Nd;
}

/* =============================================================================================
 * Per-cube sweep + bounds (jly_goicp.cpp:322-550)
 * =========================================================================================== */
void orc_eval_cube(orc_ctx* c, int level, float nx, float ny, float nz, float nw, float* ub_out, float* lb_out) {
    int Nd = c->Nd, i;
    const float* rho = level >= 0 ? &c->maxRotDis[(size_t)level * Nd] : NULL;
    float tx = nx + nw / 2, ty = ny + nw / 2, tz = nz + nw / 2;
    float maxTransDis = (float)(ORC_SQRT3 / 2.0 * nw);
    float* md = c->minDis;
    for (i = 0; i < Nd; i++) {
        float d = c->weights[i] * dt_distance(c, c->rot[3*i] + tx, c->rot[3*i+1] + ty, c->rot[3*i+2] + tz);
        if (rho) d -= rho[i];
        if (d < 0) d = 0;
        md[i] = d;
    }
    if (c->doTrim) select_k(md, 0, (size_t)(Nd - 1), (size_t)(c->inlierNum - 1));
    float ub = 0, lb = 0;
    for (i = 0; i < c->inlierNum; i++) { if (c->norm == 2) ub += md[i] * md[i]; if (c->norm == 1) ub += md[i]; }
    for (i = 0; i < c->inlierNum; i++) {
        float dis = md[i] - maxTransDis;
        if (dis > 0) { if (c->norm == 2) lb += dis * dis; if (c->norm == 1) lb += dis; }
    }
    c->n_cube++;
    /* corner terms: extremes over the cube's 8 corners (memoisation in the reference is value-neutral) */
    if (c->reg > 0 || (c->regFPFH > 0 && c->cfpfh != 0)) {
        int minI = 0, maxI = 0; float minF = 0, maxF = 0;
        for (int k = 0; k < 8; k++) {
            float cx = nx + (k & 1) * nw, cy = ny + (k >> 1 & 1) * nw, cz = nz + (k >> 2 & 1) * nw;
            if (c->reg > 0) {
                int v = orc_check_compatibilities(c, cx, cy, cz);
                if (k == 0) minI = maxI = v; else { if (v > maxI) maxI = v; if (v < minI) minI = v; }
            }
            if (c->regFPFH > 0) {
                int v = (int)orc_sum_fpfh(c, cx, cy, cz);       /* `int tempFPFH = it->fpfh`, jly_goicp.cpp:468,495 */
                if (k == 0) minF = maxF = (float)v; else { if (v > maxF) maxF = (float)v; if (v < minF) minF = (float)v; }
            }
        }
        if (c->reg > 0)     { ub += c->reg * (maxI * maxI);     lb += c->reg * (minI * minI); }
        if (c->regFPFH > 0) { ub += c->regFPFH * (maxF * maxF); lb += c->regFPFH * (minF * minF); }
    }
    *ub_out = ub; *lb_out = lb;
}

/* InnerBnB, jly_goicp.cpp:286-579 */
static float inner_bnb(orc_ctx* c, int level, tnode* out) {
    theap q = {0, 0, 0};
    float optErrorT = c->optError;
    long long pops = 0;
    theap_push(&q, c->initTrans);
    if (c->trace_on && c->trace_n < c->trace_cap) { memcpy(c->trace + 6 * c->trace_n, &c->initTrans, sizeof(tnode)); c->trace_n++; }
    for (;;) {
        if (q.n == 0) break;
        tnode par = q.a[0];
        theap_pop(&q); c->n_tpop++; pops++;
        if (optErrorT - par.lb < c->SSEThresh) break;
        if (c->pop_budget > 0 && pops > c->pop_budget) break;      /* test hook: bounded prefix of a deep search */
        tnode ch; ch.w = par.w / 2;
        for (int j = 0; j < 8; j++) {
            ch.x = par.x + (j & 1) * ch.w; ch.y = par.y + (j >> 1 & 1) * ch.w; ch.z = par.z + (j >> 2 & 1) * ch.w;
            float ub, lb;
            orc_eval_cube(c, level, ch.x, ch.y, ch.z, ch.w, &ub, &lb);
            if (ub < optErrorT) { optErrorT = ub; if (out) { *out = ch; /* ub/lb fields keep the parent's stale values in the reference */ } }
            if (lb >= optErrorT) continue;
            ch.ub = ub; ch.lb = lb;
            theap_push(&q, ch);
            if (c->trace_on && c->trace_n < c->trace_cap) { memcpy(c->trace + 6 * c->trace_n, &ch, sizeof ch); c->trace_n++; }
        }
    }
    free(q.a);
    return optErrorT;
}

void orc_set_pop_budget(orc_ctx* c, long long n) { c->pop_budget = n; }

float orc_inner_bnb(orc_ctx* c, int level, float* node_out, int want_node) {
    tnode n; memset(&n, 0, sizeof n);
    float r = inner_bnb(c, level, want_node ? &n : NULL);
    if (node_out) memcpy(node_out, &n, sizeof n);
    return r;
}

/* =============================================================================================
 * ICP (jly_icp3d.hpp:197-311) and the DT-based ICP error (jly_goicp.cpp:102-178)
 * =========================================================================================== */
/* 3x3 SVD by one-sided Jacobi (own implementation; the reference uses a Golub-Kahan routine, matrix.cpp:582).
 * Only the product V*diag(1,1,det)*U^T is consumed, which is unique for a non-degenerate H. */
static void svd3(const double* H, double* U, double* W, double* V) {
    double A[9]; memcpy(A, H, sizeof A);
    double Vm[9] = {1,0,0, 0,1,0, 0,0,1};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            double alpha = 0, beta = 0, gamma = 0;
            for (int i = 0; i < 3; i++) { alpha += A[3*i+p] * A[3*i+p]; beta += A[3*i+q] * A[3*i+q]; gamma += A[3*i+p] * A[3*i+q]; }
            if (gamma == 0) continue;
            off += fabs(gamma) / sqrt(alpha * beta + 1e-300);
            double zeta = (beta - alpha) / (2 * gamma);
            double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
            double cs = 1 / sqrt(1 + t * t), sn = cs * t;
            for (int i = 0; i < 3; i++) {
                double ap = A[3*i+p], aq = A[3*i+q]; A[3*i+p] = cs * ap - sn * aq; A[3*i+q] = sn * ap + cs * aq;
                double vp = Vm[3*i+p], vq = Vm[3*i+q]; Vm[3*i+p] = cs * vp - sn * vq; Vm[3*i+q] = sn * vp + cs * vq;
            }
        }
        if (off < 1e-15) break;
    }
    double w[3]; int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; j++) w[j] = sqrt(A[j] * A[j] + A[3+j] * A[3+j] + A[6+j] * A[6+j]);
    for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (w[ord[b]] > w[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    for (int j = 0; j < 3; j++) {
        int s = ord[j]; W[j] = w[s];
        for (int i = 0; i < 3; i++) { V[3*i+j] = Vm[3*i+s]; U[3*i+j] = w[s] > 0 ? A[3*i+s] / w[s] : 0; }
    }
    /* complete U for rank-deficient H: third column = cross of the first two */
    if (!(W[2] > 1e-300 * (W[0] + 1))) {
        if (!(W[1] > 1e-300 * (W[0] + 1))) {   /* rank <= 1: pick any orthonormal completion */
            double a0 = U[0], a1 = U[3], a2 = U[6];
            double n0 = sqrt(a0*a0 + a1*a1 + a2*a2);
            if (n0 == 0) { U[0] = 1; U[3] = 0; U[6] = 0; a0 = 1; a1 = a2 = 0; }
            double bx = fabs(a0) < 0.9 ? 1 : 0, by = fabs(a0) < 0.9 ? 0 : 1, bz = 0;
            double dp = bx*a0 + by*a1 + bz*a2; bx -= dp*a0; by -= dp*a1; bz -= dp*a2;
            double nb = sqrt(bx*bx + by*by + bz*bz); U[1] = bx/nb; U[4] = by/nb; U[7] = bz/nb;
        }
        U[2] = U[3]*U[7] - U[6]*U[4]; U[5] = U[6]*U[1] - U[0]*U[7]; U[8] = U[0]*U[4] - U[3]*U[1];
    }
}

static void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3*i+k] * B[3*k+j]; C[3*i+j] = s; }
}

/* R_ = V*diag(1,1,det(V U^T))*U^T with the determinant rounded to float, jly_icp3d.hpp:287-303 */
void orc_svd3_rot(const double* H, double* Rout) {
    double U[9], W[3], V[9], Ut[9], R0[9], D[9] = {1,0,0, 0,1,0, 0,0,1}, VD[9];
    svd3(H, U, W, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ut[3*i+j] = U[3*j+i];
    mat3_mul(V, Ut, R0);
    float a = (float)( R0[0] * (R0[4] * R0[8] - R0[5] * R0[7]));
    float b = (float)(-R0[1] * (R0[3] * R0[8] - R0[5] * R0[6]));
    float cc = (float)( R0[2] * (R0[3] * R0[7] - R0[4] * R0[6]));
    float det = a + b + cc;
    D[8] = det;
    mat3_mul(V, D, VD); mat3_mul(VD, Ut, Rout);
}

/* ---------------------------------------------------------------------------------------------
 * 1-NN with the search ORDER of the reference's kd-tree (vendored nanoflann 1.1.6: KDTreeSingleIndexAdaptor<
 * L2_Simple_Adaptor<float>, PointCloud<float>, 3>, leaf size 10, built by ICP3D::Build, jly_icp3d.hpp:141-169).
 * Restated because the result of an exact search still depends on the traversal: a candidate replaces the current best only
 * if it is strictly closer (KNNResultSet::addPoint, nanoflann.hpp:95-116; leaf test :1147), so among model points at exactly
 * the same float distance the one met first wins, and subtrees are pruned with float-accumulated box distances
 * (nanoflann.hpp:1137-1184). Tree construction: divideTree/middleSplit_/planeSplit (nanoflann.hpp:927-1110), including
 * middleSplit_'s habit of measuring the spread of the axis chosen so far rather than of the axis under test (:1047).
 * ------------------------------------------------------------------------------------------- */
typedef struct { int c1, c2; int left, right; int feat; float divlow, divhigh; } kdnode;   /* c1 < 0: leaf over vind[left, right) */
typedef struct { const float* pts; int n; int* vind; kdnode* nodes; int nn, cap; float lo[3], hi[3]; int root; } kdtree;

static void kd_minmax(const kdtree* T, const int* ind, long count, int f, float* mn, float* mx) {
    *mn = *mx = T->pts[3 * ind[0] + f];
    for (long i = 1; i < count; i++) { float v = T->pts[3 * ind[i] + f]; if (v < *mn) *mn = v; if (v > *mx) *mx = v; }
}
static void kd_plane_split(const kdtree* T, int* ind, long count, int f, float cutval, long* lim1, long* lim2) {
    long left = 0, right = count - 1; int t;
    for (;;) {
        while (left <= right && T->pts[3 * ind[left] + f] < cutval) ++left;
        while (right && left <= right && T->pts[3 * ind[right] + f] >= cutval) --right;
        if (left > right || !right) break;
        t = ind[left]; ind[left] = ind[right]; ind[right] = t; ++left; --right;
    }
    *lim1 = left; right = count - 1;
    for (;;) {
        while (left <= right && T->pts[3 * ind[left] + f] <= cutval) ++left;
        while (right && left <= right && T->pts[3 * ind[right] + f] > cutval) --right;
        if (left > right || !right) break;
        t = ind[left]; ind[left] = ind[right]; ind[right] = t; ++left; --right;
    }
    *lim2 = left;
}
static int kd_divide(kdtree* T, long left, long right, float* lo, float* hi) {
    if (T->nn == T->cap) { T->cap = 2 * T->cap + 16; T->nodes = (kdnode*)realloc(T->nodes, sizeof(kdnode) * T->cap); }
    int me = T->nn++;
    if (right - left <= 10) {
        T->nodes[me].c1 = T->nodes[me].c2 = -1; T->nodes[me].left = (int)left; T->nodes[me].right = (int)right;
        for (int i = 0; i < 3; i++) lo[i] = hi[i] = T->pts[3 * T->vind[left] + i];
        for (long k = left + 1; k < right; k++) for (int i = 0; i < 3; i++) {
            float v = T->pts[3 * T->vind[k] + i];
            if (lo[i] > v) lo[i] = v;
            if (hi[i] < v) hi[i] = v;
        }
        return me;
    }
    int* ind = T->vind + left; long count = right - left;
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int i = 1; i < 3; i++) { float span = hi[i] - lo[i]; if (span > max_span) max_span = span; }
    float max_spread = -1; int cutfeat = 0; float mn, mx;
    for (int i = 0; i < 3; i++) {
        float span = hi[i] - lo[i];
        if (span > (1 - EPS) * max_span) {
            kd_minmax(T, ind, count, cutfeat, &mn, &mx);           /* sic: the axis chosen so far */
            float spread = mx - mn;
            if (spread > max_spread) { cutfeat = i; max_spread = spread; }
        }
    }
    float split_val = (lo[cutfeat] + hi[cutfeat]) / 2, cutval;
    kd_minmax(T, ind, count, cutfeat, &mn, &mx);
    if (split_val < mn) cutval = mn; else if (split_val > mx) cutval = mx; else cutval = split_val;
    long lim1, lim2, idx;
    kd_plane_split(T, ind, count, cutfeat, cutval, &lim1, &lim2);
    if (lim1 > count / 2) idx = lim1; else if (lim2 < count / 2) idx = lim2; else idx = count / 2;
    float llo[3], lhi[3], rlo[3], rhi[3];
    memcpy(llo, lo, sizeof llo); memcpy(lhi, hi, sizeof lhi); memcpy(rlo, lo, sizeof rlo); memcpy(rhi, hi, sizeof rhi);
    lhi[cutfeat] = cutval;
    int c1 = kd_divide(T, left, left + idx, llo, lhi);
    rlo[cutfeat] = cutval;
    int c2 = kd_divide(T, left + idx, right, rlo, rhi);
    kdnode* nd = &T->nodes[me];
    nd->c1 = c1; nd->c2 = c2; nd->feat = cutfeat; nd->divlow = lhi[cutfeat]; nd->divhigh = rlo[cutfeat];
    for (int i = 0; i < 3; i++) { lo[i] = llo[i] < rlo[i] ? llo[i] : rlo[i]; hi[i] = lhi[i] > rhi[i] ? lhi[i] : rhi[i]; }
    return me;
}
static kdtree* kd_build(const float* pts, int n) {
    kdtree* T = (kdtree*)calloc(1, sizeof *T);
    T->pts = pts; T->n = n; T->vind = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) T->vind[i] = i;
    T->root = -1;
    if (n <= 0) return T;
    for (int i = 0; i < 3; i++) T->lo[i] = T->hi[i] = pts[i];       /* computeBoundingBox, nanoflann.hpp:894-917 */
    for (int k = 1; k < n; k++) for (int i = 0; i < 3; i++) { float v = pts[3 * k + i]; if (v < T->lo[i]) T->lo[i] = v; if (v > T->hi[i]) T->hi[i] = v; }
    T->root = kd_divide(T, 0, n, T->lo, T->hi);
    return T;
}
static void kd_free(kdtree* T) { if (T) { free(T->vind); free(T->nodes); free(T); } }
typedef struct { float best; int idx; } kdres;
static void kd_search(const kdtree* T, kdres* r, const float* q, int node, float mindistsq, float* dists) {
    const kdnode* nd = &T->nodes[node];
    if (nd->c1 < 0) {
        const float worst = r->best;
        for (int i = nd->left; i < nd->right; i++) {
            const int m = T->vind[i];
            const float d0 = q[0] - T->pts[3 * m], d1 = q[1] - T->pts[3 * m + 1], d2 = q[2] - T->pts[3 * m + 2];
            const float d = d0 * d0 + d1 * d1 + d2 * d2;
            if (d < worst && r->best > d) { r->best = d; r->idx = m; }
        }
        return;
    }
    const int f = nd->feat; const float val = q[f];
    const float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int bestc, otherc; float cut;
    if ((diff1 + diff2) < 0) { bestc = nd->c1; otherc = nd->c2; cut = (val - nd->divhigh) * (val - nd->divhigh); }
    else { bestc = nd->c2; otherc = nd->c1; cut = (val - nd->divlow) * (val - nd->divlow); }
    kd_search(T, r, q, bestc, mindistsq, dists);
    const float dst = dists[f];
    mindistsq = mindistsq + cut - dst;
    dists[f] = cut;
    if (mindistsq * 1.0f <= r->best) kd_search(T, r, q, otherc, mindistsq, dists);
    dists[f] = dst;
}
static int kd_nn(const kdtree* T, const float* q, float* dist_out) {
    kdres r; r.best = FLT_MAX; r.idx = 0;
    float dists[3] = {0, 0, 0}, distsq = 0;
    for (int i = 0; i < 3; i++) {                                    /* computeInitialDistances, nanoflann.hpp:1113-1131 */
        if (q[i] < T->lo[i]) { dists[i] = (q[i] - T->lo[i]) * (q[i] - T->lo[i]); distsq += dists[i]; }
        if (q[i] > T->hi[i]) { dists[i] = (q[i] - T->hi[i]) * (q[i] - T->hi[i]); distsq += dists[i]; }
    }
    if (T->root >= 0) kd_search(T, &r, q, T->root, distsq, dists);
    *dist_out = r.best;
    return r.idx;
}
/* test hook: the tree's 1-NN for n query points */
void orc_kd_nn(const float* model, int Nm, const float* q, int n, int* id, float* d) {
    kdtree* T = kd_build(model, Nm);
    for (int i = 0; i < n; i++) id[i] = kd_nn(T, q + 3 * i, &d[i]);
    kd_free(T);
}

typedef struct { double dis; int id_data, id_model; } pref;

static void stable_sort_dis(pref* p, pref* tmp, int n) {   /* qsort+cmp of jly_icp3d.hpp:172-176,254 under glibc (merge sort, ties keep order) */
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (p[i].dis > p[j].dis) ? p[j++] : p[i++];
            while (i < mid) tmp[k++] = p[i++];
            while (j < hi) tmp[k++] = p[j++];
        }
        memcpy(p, tmp, sizeof(pref) * n);
    }
}

/* ICP3D<float>::Run, jly_icp3d.hpp:197-311 (1-NN through the kd-tree restatement above: the reference's visiting order) */
static float icp_run(orc_ctx* c, double* R, double* t) {
    int n = c->Nd, num, i;
    const float* data = c->dxyz; const float* model = c->mxyz;
    kdtree* kd = kd_build(model, c->Nm);          /* icp3d.Build(M_icp, Nm), jly_goicp.cpp:231 (the tree is a pure function of the model) */
    float err_diff = c->MSEThresh / 10000;
    if (c->doTrim) num = (int)(n * (1 - c->trimFraction)); else num = n;
    pref* pts = (pref*)malloc(sizeof(pref) * n); pref* tmp = (pref*)malloc(sizeof(pref) * n);
    double* p_m = (double*)malloc(sizeof(double) * 3 * (num ? num : 1)); double* p_d = (double*)malloc(sizeof(double) * 3 * (num ? num : 1));
    double mu_m[3] = {0, 0, 0}, mu_d[3] = {0, 0, 0};
    float err = -1, err_new = 0;
    for (size_t iter = 0; iter < 10000; iter++) {
        float r00 = (float)R[0], r01 = (float)R[1], r02 = (float)R[2], r10 = (float)R[3], r11 = (float)R[4], r12 = (float)R[5];
        float r20 = (float)R[6], r21 = (float)R[7], r22 = (float)R[8], t0 = (float)t[0], t1 = (float)t[1], t2 = (float)t[2];
        err_new = 0;
        for (i = 0; i < n; i++) {
            float q0 = r00 * data[3*i] + r01 * data[3*i+1] + r02 * data[3*i+2] + t0;
            float q1 = r10 * data[3*i] + r11 * data[3*i+1] + r12 * data[3*i+2] + t1;
            float q2 = r20 * data[3*i] + r21 * data[3*i+1] + r22 * data[3*i+2] + t2;
            float qv[3] = {q0, q1, q2}, best;
            int bi = kd_nn(kd, qv, &best);                      /* kdtree->knnSearch(query, 1, ...), jly_icp3d.hpp:245 */
            pts[i].dis = best; pts[i].id_data = i; pts[i].id_model = bi;
        }
        if (c->doTrim) stable_sort_dis(pts, tmp, n);
        for (i = 0; i < num; i++) {
            int im = pts[i].id_model, id = pts[i].id_data;
            p_m[3*i] = model[3*im]; mu_m[0] += p_m[3*i];
            p_m[3*i+1] = model[3*im+1]; mu_m[1] += p_m[3*i+1];
            p_m[3*i+2] = model[3*im+2]; mu_m[2] += p_m[3*i+2];
            p_d[3*i]   = r00 * data[3*id] + r01 * data[3*id+1] + r02 * data[3*id+2] + t0; mu_d[0] += p_d[3*i];
            p_d[3*i+1] = r10 * data[3*id] + r11 * data[3*id+1] + r12 * data[3*id+2] + t1; mu_d[1] += p_d[3*i+1];
            p_d[3*i+2] = r20 * data[3*id] + r21 * data[3*id+1] + r22 * data[3*id+2] + t2; mu_d[2] += p_d[3*i+2];
            err_new = (float)(err_new + pts[i].dis);
        }
        if (err > 0 && err - err_new < err_diff * num) break;
        err = err_new;
        /* means: divided by n (not num) and NOT reset between iterations, jly_icp3d.hpp:260-262,278-279 */
        for (int k = 0; k < 3; k++) { mu_m[k] = mu_m[k] / (double)(float)n; mu_d[k] = mu_d[k] / (double)(float)n; }
        double H[9] = {0,0,0, 0,0,0, 0,0,0};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            double s = 0;
            for (i = 0; i < num; i++) s += (p_d[3*i+a] - mu_d[a]) * (p_m[3*i+b] - mu_m[b]);
            H[3*a+b] = s;
        }
        double Rn[9], Rc[9], tc[3], t_[3];
        orc_svd3_rot(H, Rn);
        for (int a = 0; a < 3; a++) { double s = 0; for (int k = 0; k < 3; k++) s += Rn[3*a+k] * mu_d[k]; t_[a] = mu_m[a] - s; }
        mat3_mul(Rn, R, Rc);
        for (int a = 0; a < 3; a++) { double s = 0; for (int k = 0; k < 3; k++) s += Rn[3*a+k] * t[k]; tc[a] = s + t_[a]; }
        memcpy(R, Rc, sizeof Rc); memcpy(t, tc, sizeof tc);
    }
    for (i = 0; i < n; i++) { c->icp_pts[2*i] = pts[i].id_data; c->icp_pts[2*i+1] = pts[i].id_model; c->icp_dis[i] = pts[i].dis; }
    free(pts); free(tmp); free(p_m); free(p_d); kd_free(kd);
    return err_new;
}

/* countCompatibilities, jly_goicp.cpp:890-914 */
static int count_incompat(const orc_ctx* c, const int* pts) {
    int bad = 0;
    for (int i = 0; i < c->Nd; i++) { int s = c->dcol[pts[2*i]], t = c->mcol[pts[2*i+1]]; if (!(known_prop(s) && s == t)) bad++; }
    return bad;
}

/* GoICP::ICP, jly_goicp.cpp:102-178 */
static float icp_error(orc_ctx* c, double* R, double* t) {
    int Nd = c->Nd, i, b0, b1; fpfh_range(c, &b0, &b1);
    icp_run(c, R, t);
    float error = 0, fpfh = 0;
    for (i = 0; i < Nd; i++) {
        float px = c->dxyz[3*i], py = c->dxyz[3*i+1], pz = c->dxyz[3*i+2];
        float x = (float)(R[0] * px + R[1] * py + R[2] * pz + t[0]);
        float y = (float)(R[3] * px + R[4] * py + R[5] * pz + t[1]);
        float z = (float)(R[6] * px + R[7] * py + R[8] * pz + t[2]);
        if (!c->doTrim) {
            float dis = c->weights[i] * dt_distance(c, x, y, z);
            if (c->norm == 2) error += dis * dis;
            if (c->norm == 1) error += dis;
        } else c->minDis[i] = dt_distance(c, x, y, z);
        if (c->cfpfh != 0) {
            const float* f = &c->dfpfh[(size_t)NBINS * c->icp_pts[2*i]]; const float* g = &c->mfpfh[(size_t)NBINS * c->icp_pts[2*i+1]];
            float d = 0;
            for (int b = b0; b < b1; b++) d += fabsf(f[b] - g[b]);
            fpfh += d;
        }
    }
    fpfh = fpfh / Nd;
    if (c->reg > 0) { int inc = count_incompat(c, c->icp_pts); error += c->reg * (inc * inc); }
    if (c->regFPFH > 0) error += c->regFPFH * (fpfh * fpfh);
    if (c->doTrim) {
        select_k(c->minDis, 0, (size_t)(Nd - 1), (size_t)(c->inlierNum - 1));
        for (i = 0; i < c->inlierNum; i++) error += c->minDis[i] * c->minDis[i];
    }
    return error;
}

float orc_icp(orc_ctx* c, double* R, double* t, int* corr, double* dis) {
    float e = icp_error(c, R, t);
    if (corr) memcpy(corr, c->icp_pts, sizeof(int) * 2 * c->Nd);
    if (dis) memcpy(dis, c->icp_dis, sizeof(double) * c->Nd);
    return e;
}

/* updateCompatibilities, jly_goicp.cpp:933-946 */
static int update_compat(const orc_ctx* c) {
    int bad = 0;
    for (int i = 0; i < c->Nd; i++) {
        float px = c->dxyz[3*i], py = c->dxyz[3*i+1], pz = c->dxyz[3*i+2];
        float x = (float)(c->optR[0] * px + c->optR[1] * py + c->optR[2] * pz + c->optT[0]);
        float y = (float)(c->optR[3] * px + c->optR[4] * py + c->optR[5] * pz + c->optT[1]);
        float z = (float)(c->optR[6] * px + c->optR[7] * py + c->optR[8] * pz + c->optT[2]);
        if (!compatible_at(c, i, x, y, z)) bad++;
    }
    return bad;
}

/* OuterBnB, jly_goicp.cpp:582-876 */
static float outer_bnb(orc_ctx* c) {
    int Nd = c->Nd, i, j;
    rheap q = {0, 0, 0};
    float lb = 0, ub, error;
    c->optError = 0;
    for (i = 0; i < Nd; i++) c->minDis[i] = c->weights[i] * dt_distance(c, c->dxyz[3*i], c->dxyz[3*i+1], c->dxyz[3*i+2]);
    if (c->doTrim) select_k(c->minDis, 0, (size_t)(Nd - 1), (size_t)(c->inlierNum - 1));
    for (i = 0; i < c->inlierNum; i++) { if (c->norm == 2) c->optError += c->minDis[i] * c->minDis[i]; if (c->norm == 1) c->optError += c->minDis[i]; }
    if (c->reg > 0) c->optError += c->reg * (Nd * Nd);
    if (c->regFPFH > 0) c->optError += c->regFPFH * (100 * 8 * 100 * 8);
    logf_(c, "Error*: %g (Init)\n", (double)c->optError);

    double R_icp[9], t_icp[3];
    memcpy(R_icp, c->optR, sizeof R_icp); memcpy(t_icp, c->optT, sizeof t_icp);
    error = icp_error(c, R_icp, t_icp);
    memcpy(c->opt_pts, c->icp_pts, sizeof(int) * 2 * Nd);          /* icp3d.optPoints = icp3d.points, :635 */
    if (error < c->optError) {
        c->optError = error; memcpy(c->optR, R_icp, sizeof R_icp); memcpy(c->optT, t_icp, sizeof t_icp);
        c->optComp = count_incompat(c, c->opt_pts);
        logf_(c, "Error*: %g (ICP)\n", (double)error);
    }
    rheap_push(&q, c->initRot);
    for (;;) {
        if (q.n == 0) { logf_(c, "Rotation Queue Empty\n"); break; }
        rnode par = q.a[0];
        rheap_pop(&q); c->n_rpop++;
        if ((c->optError - par.lb) <= c->SSEThresh) { logf_(c, "Threshold reached LB %g\n", (double)par.lb); break; }
        rnode ch; ch.w = par.w / 2; ch.l = par.l + 1;
        for (j = 0; j < 8; j++) {
            ch.a = par.a + (j & 1) * ch.w; ch.b = par.b + (j >> 1 & 1) * ch.w; ch.c = par.c + (j >> 2 & 1) * ch.w;
            float v1 = ch.a + ch.w / 2, v2 = ch.b + ch.w / 2, v3 = ch.c + ch.w / 2;
            if ((double)sqrtf(v1 * v1 + v2 * v2 + v3 * v3) - ORC_SQRT3 * ch.w / 2 > ORC_PI) continue;
            float R[9];
            orc_axis_angle(v1, v2, v3, R);
            if (sqrtf(v1 * v1 + v2 * v2 + v3 * v3) > 0) orc_rotate(c, R); else memcpy(c->rot, c->dxyz, sizeof(float) * 3 * Nd);
            tnode nt; memset(&nt, 0, sizeof nt);
            ub = inner_bnb(c, -1, &nt);
            if (ub < c->optError) {
                c->optError = ub; c->optNodeRot = ch; c->optNodeTrans = nt;
                for (i = 0; i < 9; i++) c->optR[i] = R[i];
                c->optT[0] = nt.x + nt.w / 2; c->optT[1] = nt.y + nt.w / 2; c->optT[2] = nt.z + nt.w / 2;
                if (c->nearest3) c->optComp = update_compat(c);
                logf_(c, "Error*: %g (BNB)\n", (double)c->optError);
                memcpy(R_icp, c->optR, sizeof R_icp); memcpy(t_icp, c->optT, sizeof t_icp);
                error = icp_error(c, R_icp, t_icp);
                if (error < c->optError) {
                    c->optError = error; memcpy(c->optR, R_icp, sizeof R_icp); memcpy(c->optT, t_icp, sizeof t_icp);
                    memcpy(c->opt_pts, c->icp_pts, sizeof(int) * 2 * Nd);
                    c->optComp = count_incompat(c, c->opt_pts);
                    logf_(c, "Error*: %g (ICP)\n", (double)error);
                }
                /* drop queued rotation cubes whose lb can no longer win, :843-853 */
                rheap q2 = {0, 0, 0};
                while (q.n) { rnode nd = q.a[0]; rheap_pop(&q); if (nd.lb < c->optError) rheap_push(&q2, nd); else break; }
                free(q.a); q = q2;
            }
            lb = inner_bnb(c, ch.l, NULL);
            if (lb >= c->optError) continue;
            ch.ub = ub; ch.lb = lb;
            rheap_push(&q, ch);
        }
    }
    free(q.a);
    return c->optError;
}

float orc_register(orc_ctx* c, double* out) {
    c->log_n = 0; if (c->log) c->log[0] = 0;
    orc_initialize(c);
    float e = outer_bnb(c);
    for (int i = 0; i < 9; i++) out[i] = c->optR[i];
    for (int i = 0; i < 3; i++) out[9 + i] = c->optT[i];
    out[12] = c->optError; out[13] = c->optComp; out[14] = c->Nd; out[15] = c->inlierNum;
    return e;
}

int orc_last_log(orc_ctx* c, char* buf, int cap) {
    int n = (int)c->log_n;
    if (buf && cap > 0) { int m = n < cap - 1 ? n : cap - 1; if (m > 0) memcpy(buf, c->log, m); buf[m] = 0; }
    return n;
}

void orc_trace_begin(orc_ctx* c, long long cap) {
    free(c->trace); c->trace = (float*)malloc(sizeof(float) * 6 * (cap ? cap : 1)); c->trace_cap = cap; c->trace_n = 0; c->trace_on = 1;
}
long long orc_trace_end(orc_ctx* c, float* out, long long cap) {
    c->trace_on = 0;
    long long n = c->trace_n < cap ? c->trace_n : cap;
    memcpy(out, c->trace, sizeof(float) * 6 * n);
    return c->trace_n;
}
void orc_counters(orc_ctx* c, long long* o, int reset) {
    o[0] = c->n_tpop; o[1] = c->n_rpop; o[2] = c->n_cube; o[3] = c->n_corner;
    if (reset) c->n_tpop = c->n_rpop = c->n_cube = c->n_corner = 0;
}
