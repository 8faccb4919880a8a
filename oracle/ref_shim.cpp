// TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.
//
// C-ABI instrumentation shim around the UNMODIFIED reference sources, which are compiled
// where they lie under /root/reference (see oracle/Makefile; output oracle/_ref/libgoicp_ref.so).
// Nothing from the reference is copied here: this file only *includes* its headers / its
// jly_goicp.cpp translation unit with two instrumentation macros:
//   * `private` -> `public`, so the private seams named in SURVEY.md 8(b)
//     (InnerBnB, ICP, Initialize, OuterBnB, weights, maxRotDis, pDataTemp) can be driven one by one;
//   * `priority_queue` -> a thin logging wrapper, so every TRANSNODE the reference pushes
//     ((cube) -> (ub, lb) as computed by jly_goicp.cpp:343-550) can be recorded as a known answer;
//   * `malloc(n)` -> `calloc(1,n)` for that one translation unit (see the note at the include).
//
// Used by: tests/ (oracle pinning), tests/golden/make_golden.py (fixture generation) and
// bench.py's cpu_baseline leg ("kind": "reference").
#include <queue>
#include <vector>
#include <map>
#include <string>
#include <sstream>
#include <iostream>
#include <cstring>
#include <cstdlib>
#include <cmath>

// ---- push-trace wrapper -------------------------------------------------------------------
struct RefTraceRec { float x, y, z, w, ub, lb; };
static std::vector<RefTraceRec> g_trace;
static bool g_trace_on = false;
static size_t g_trace_cap = 0;
static long long g_trans_pops = 0;
static long long g_rot_pops = 0;
static long long g_trans_empty = 0;   // InnerBnB loops that ended on an empty queue

#include <algorithm>
#include <stdexcept>
#include <limits>
#include <cassert>
#include <cstdio>
#include <fstream>
#include <random>
#include <iterator>
#include <ctime>

#define private public
#include "jly_goicp.h"     // resolved through -I/root/reference (compiled in place, never copied)

template <class T> struct trace_log;
template <> struct trace_log<TRANSNODE> {
    static void push(const TRANSNODE& n) {
        if (g_trace_on && g_trace.size() < g_trace_cap) {
            RefTraceRec r = {n.x, n.y, n.z, n.w, n.ub, n.lb};
            g_trace.push_back(r);
        }
    }
    static void pop() { g_trans_pops++; }
    static void empty_hit() { g_trans_empty++; }
};
template <> struct trace_log<ROTNODE> {
    static void push(const ROTNODE&) {}
    static void pop() { g_rot_pops++; }
    static void empty_hit() {}
};

template <class T>
class traced_priority_queue {
    std::priority_queue<T> q_;
public:
    bool empty() const { bool e = q_.empty(); if (e) trace_log<T>::empty_hit(); return e; }
    size_t size() const { return q_.size(); }
    const T& top() const { return q_.top(); }
    void pop() { trace_log<T>::pop(); q_.pop(); }
    void push(const T& v) { trace_log<T>::push(v); q_.push(v); }
};

#define priority_queue traced_priority_queue
// The reference leaves `std::vector` members of malloc'd POINT3D arrays unconstructed and later
// `delete`s them (jly_goicp.cpp:212-213,271-273), which only works on the zero pages a fresh process
// gets from malloc. Inside a long-lived test process the heap is dirty, so hand it zeroed blocks.
#define malloc(n) calloc(1, (n))
#include "jly_goicp.cpp"   // the reference translation unit itself, unmodified
#undef malloc
#undef priority_queue
#undef private

#include "jly_sorting.hpp"

// ---- handle -------------------------------------------------------------------------------
struct RefHandle {
    GoICP g;
    bool initialized;
    std::string log;
    RefHandle() : initialized(false) {}
};

namespace {
struct CoutCapture {
    std::streambuf* old;
    std::ostringstream ss;
    CoutCapture() { old = std::cout.rdbuf(ss.rdbuf()); }
    ~CoutCapture() { std::cout.rdbuf(old); }
};
}

extern "C" {

void* ref_new() { return new RefHandle(); }

// Leaks exactly what the reference leaks (DT cell tables); the handle itself is freed.
void ref_free(void* hv) {
    RefHandle* h = (RefHandle*)hv;
    if (h->initialized) { CoutCapture cap; h->g.Clear(); h->initialized = false; }
    delete[] h->g.pModel; h->g.pModel = NULL;
    delete[] h->g.pData;  h->g.pData = NULL;
    delete h;
}

// cfg[18]: MSEThresh, rotMinX, rotMinY, rotMinZ, rotWidth, transMinX, transMinY, transMinZ, transWidth,
//          trimFraction, regularization, regularizationNeighbors, regularizationFPFH, cfpfh, norm,
//          ponderation, distTransSize, distTransExpandFactor     (mirrors readConfig, jly_main.cpp:231-264)
void ref_set_config(void* hv, const double* cfg) {
    GoICP& g = ((RefHandle*)hv)->g;
    g.MSEThresh = (float)cfg[0];
    g.initNodeRot.a = (float)cfg[1]; g.initNodeRot.b = (float)cfg[2]; g.initNodeRot.c = (float)cfg[3];
    g.initNodeRot.w = (float)cfg[4];
    g.initNodeTrans.x = (float)cfg[5]; g.initNodeTrans.y = (float)cfg[6]; g.initNodeTrans.z = (float)cfg[7];
    g.initNodeTrans.w = (float)cfg[8];
    g.trimFraction = (float)cfg[9];
    g.regularization = (float)cfg[10];
    g.regularizationNeighbors = (float)cfg[11];
    g.regularizationFPFH = (float)cfg[12];
    g.cfpfh = (int)cfg[13];
    g.norm = (int)cfg[14];
    g.ponderation = (int)cfg[15];
    g.doTrim = true;
    if (g.trimFraction < 0.001) g.doTrim = false;
    g.dt.SIZE = (int)cfg[16];
    g.dt.expandFactor = cfg[17];
}

static POINT3D* make_cloud(int n, const float* xyz, const int* col, const float* fpfh) {
    POINT3D* p = new POINT3D[n];   // constructed (the reference's caller mallocs; same field values)
    for (int i = 0; i < n; i++) {
        p[i].x = xyz[3*i]; p[i].y = xyz[3*i+1]; p[i].z = xyz[3*i+2];
        p[i].c = col ? col[i] : 0;
        p[i].neighbors = 0; p[i].density = 0;
        p[i].cfpfh.resize(41);
        for (int j = 0; j < 41; j++) p[i].cfpfh[j] = fpfh ? fpfh[41*i+j] : 0.0f;
    }
    return p;
}

void ref_set_clouds(void* hv, int Nm, const float* mxyz, const int* mcol, const float* mfpfh,
                    int Nd, const float* dxyz, const int* dcol, const float* dfpfh) {
    GoICP& g = ((RefHandle*)hv)->g;
    g.pModel = make_cloud(Nm, mxyz, mcol, mfpfh); g.Nm = Nm;
    g.pData  = make_cloud(Nd, dxyz, dcol, dfpfh); g.Nd = Nd;
}

void ref_set_nd(void* hv, int Nd) { ((RefHandle*)hv)->g.Nd = Nd; }

void ref_build_dt(void* hv) { CoutCapture cap; ((RefHandle*)hv)->g.BuildDT(); }

// geom[8]: xMin,xMax,yMin,yMax,zMin,zMax,scale,SIZE
void ref_dt_geom(void* hv, double* geom) {
    DT3D& d = ((RefHandle*)hv)->g.dt;
    geom[0]=d.xMin; geom[1]=d.xMax; geom[2]=d.yMin; geom[3]=d.yMax; geom[4]=d.zMin; geom[5]=d.zMax;
    geom[6]=d.scale; geom[7]=d.SIZE;
}

// dist[S^3] (x fastest), vhd[S^3*3] as int16 (v,h,d), nearest[S^3*3] (cx,cy,cz), cellc[S^3]; any may be NULL.
void ref_dt_dump(void* hv, float* dist, short* vhd, int* nearest, int* cellc) {
    DT3D& d = ((RefHandle*)hv)->g.dt;
    int S = d.SIZE;
    size_t k = 0;
    for (int z = 0; z < S; z++) for (int y = 0; y < S; y++) for (int x = 0; x < S; x++, k++) {
        if (dist) dist[k] = d.A.data[z][y][x].distance;
        if (vhd) { vhd[3*k]=d.A.data[z][y][x].v; vhd[3*k+1]=d.A.data[z][y][x].h; vhd[3*k+2]=d.A.data[z][y][x].d; }
        if (nearest) { nearest[3*k]=d.emptyCells[z][y][x].cx; nearest[3*k+1]=d.emptyCells[z][y][x].cy; nearest[3*k+2]=d.emptyCells[z][y][x].cz; }
        if (cellc) cellc[k] = d.cellPoints[z][y][x].c;
    }
}

// Adopt a distance grid built elsewhere (geometric path only: the cell tables stay unbuilt, so the handle must
// run with regularization == regularizationFPFH == 0 and only through ref_inner_bnb / ref_distance).
void ref_load_dt(void* hv, int S, const double* geom4, const float* dist) {
    DT3D& d = ((RefHandle*)hv)->g.dt;
    d.SIZE = S; d.xMin = geom4[0]; d.yMin = geom4[1]; d.zMin = geom4[2]; d.scale = geom4[3];
    d.xMax = d.xMin + S / d.scale; d.yMax = d.yMin + S / d.scale; d.zMax = d.zMin + S / d.scale;
    if (d.A.data == NULL) d.A.Init(S, S, S);
    size_t k = 0;
    for (int z = 0; z < S; z++) for (int y = 0; y < S; y++) for (int x = 0; x < S; x++, k++) {
        d.A.data[z][y][x].distance = dist[k]; d.A.data[z][y][x].v = d.A.data[z][y][x].h = d.A.data[z][y][x].d = 0;
    }
}

int ref_cell_points(void* hv, int x, int y, int z, int* out, int cap) {
    DT3D& d = ((RefHandle*)hv)->g.dt;
    const std::vector<int>& v = d.cellPoints[z][y][x].points;
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = v[i];
    return n;
}

// DT3D::Distance on n query points (double xyz triplets) -> out[n]
void ref_distance(void* hv, int n, const double* q, float* out) {
    DT3D& d = ((RefHandle*)hv)->g.dt;
    int cx, cy, cz;
    for (int i = 0; i < n; i++) out[i] = d.Distance(q[3*i], q[3*i+1], q[3*i+2], cx, cy, cz);
}

void ref_initialize(void* hv) {
    RefHandle* h = (RefHandle*)hv;
    CoutCapture cap;
    if (h->initialized) h->g.Clear();
    h->g.Initialize();
    h->initialized = true;
}

void ref_get_weights(void* hv, float* w) { GoICP& g = ((RefHandle*)hv)->g; memcpy(w, g.weights, sizeof(float)*g.Nd); }
void ref_set_weights(void* hv, const float* w) { GoICP& g = ((RefHandle*)hv)->g; memcpy(g.weights, w, sizeof(float)*g.Nd); }
void ref_get_maxrotdis(void* hv, float* out /*[20*Nd]*/) {
    GoICP& g = ((RefHandle*)hv)->g;
    for (int l = 0; l < MAXROTLEVEL; l++) memcpy(out + (size_t)l*g.Nd, g.maxRotDis[l], sizeof(float)*g.Nd);
}
int   ref_inlier_num(void* hv) { return ((RefHandle*)hv)->g.inlierNum; }
float ref_sse_thresh(void* hv) { return ((RefHandle*)hv)->g.SSEThresh; }
void  ref_set_sse_thresh(void* hv, float v) { ((RefHandle*)hv)->g.SSEThresh = v; }
void  ref_set_opt_error(void* hv, float e) { ((RefHandle*)hv)->g.optError = e; }
float ref_get_opt_error(void* hv) { return ((RefHandle*)hv)->g.optError; }
void  ref_set_init_trans(void* hv, float x, float y, float z, float w, float lb) {
    GoICP& g = ((RefHandle*)hv)->g;
    g.initNodeTrans.x = x; g.initNodeTrans.y = y; g.initNodeTrans.z = z; g.initNodeTrans.w = w; g.initNodeTrans.lb = lb;
}

// pDataTemp <- xyz (the rotated cloud the reference computes at jly_goicp.cpp:750-756)
void ref_set_rotated(void* hv, const float* xyz) {
    GoICP& g = ((RefHandle*)hv)->g;
    for (int i = 0; i < g.Nd; i++) { g.pDataTemp[i].x = xyz[3*i]; g.pDataTemp[i].y = xyz[3*i+1]; g.pDataTemp[i].z = xyz[3*i+2]; }
}

// GoICP::InnerBnB with maxRotDis[level] (level>=0) or NULL (level<0); node_out[6]=x,y,z,w,ub,lb.
float ref_inner_bnb(void* hv, int level, float* node_out, int want_node) {
    GoICP& g = ((RefHandle*)hv)->g;
    TRANSNODE n; memset(&n, 0, sizeof(n));
    CoutCapture cap;
    float r = g.InnerBnB(level >= 0 ? g.maxRotDis[level] : NULL, want_node ? &n : NULL);
    if (node_out) { node_out[0]=n.x; node_out[1]=n.y; node_out[2]=n.z; node_out[3]=n.w; node_out[4]=n.ub; node_out[5]=n.lb; }
    return r;
}

int   ref_check_compatibilities(void* hv, float x, float y, float z) { return ((RefHandle*)hv)->g.checkCompatibilities(x, y, z); }
float ref_sum_fpfh(void* hv, float x, float y, float z) { return ((RefHandle*)hv)->g.sumFPFH(x, y, z); }

// GoICP::ICP(R,t): R[9], t[3] in/out (double); corr[Nd*2] = (id_data,id_model) of icp3d.points, dis[Nd]
float ref_icp(void* hv, double* R, double* t, int* corr, double* dis) {
    GoICP& g = ((RefHandle*)hv)->g;
    Matrix Rm(3,3), tm(3,1);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rm.val[i][j] = R[3*i+j]; tm.val[i][0] = t[i]; }
    float e;
    { CoutCapture cap; e = g.ICP(Rm, tm); }
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3*i+j] = Rm.val[i][j]; t[i] = tm.val[i][0]; }
    for (int i = 0; i < g.Nd; i++) {
        if (corr) { corr[2*i] = g.icp3d.points[i].id_data; corr[2*i+1] = g.icp3d.points[i].id_model; }
        if (dis) dis[i] = g.icp3d.points[i].dis;
    }
    return e;
}

// Whole registration. out[16]: optR[9], optT[3], optError, optComp, Nd, inlierNum
float ref_register(void* hv, double* out) {
    RefHandle* h = (RefHandle*)hv;
    GoICP& g = h->g;
    float e;
    { CoutCapture cap; e = g.Register(); h->log = cap.ss.str(); }
    h->initialized = false;   // Register() ends with Clear()
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out[3*i+j] = g.optR.val[i][j]; out[9+i] = g.optT.val[i][0]; }
    out[12] = g.optError; out[13] = g.optComp; out[14] = g.Nd; out[15] = g.inlierNum;
    return e;
}

// stdout of the last ref_register (the reference's "Error*: ..." progress lines)
int ref_last_log(void* hv, char* buf, int cap) {
    RefHandle* h = (RefHandle*)hv;
    int n = (int)h->log.size();
    if (buf && cap > 0) { int m = n < cap-1 ? n : cap-1; memcpy(buf, h->log.data(), m); buf[m] = 0; }
    return n;
}

void ref_trace_begin(long long cap) { g_trace.clear(); g_trace_cap = (size_t)cap; g_trace_on = true; }
long long ref_trace_end(float* out /*[n*6]*/, long long cap) {
    g_trace_on = false;
    long long n = (long long)g_trace.size();
    for (long long i = 0; i < n && i < cap; i++) memcpy(out + 6*i, &g_trace[i], sizeof(RefTraceRec));
    return n;
}
void ref_counters(long long* out, int reset) {
    out[0] = g_trans_pops; out[1] = g_rot_pops; out[2] = g_trans_empty;
    if (reset) { g_trans_pops = 0; g_rot_pops = 0; g_trans_empty = 0; }
}

// jly_sorting.hpp:229 on a caller-owned array
void ref_intro_select(float* data, long long n, long long k) { intro_select(data, (size_t)0, (size_t)(n-1), (size_t)k); }

// Matrix::svd (matrix.cpp:582) of a 3x3; U,W,V row-major doubles
void ref_svd3(const double* H, double* U, double* W, double* V) {
    Matrix h(3,3), u, w, v;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) h.val[i][j] = H[3*i+j];
    h.svd(u, w, v);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) { U[3*i+j] = u.val[i][j]; V[3*i+j] = v.val[i][j]; } W[i] = w.val[i][0]; }
}

} // extern "C"
