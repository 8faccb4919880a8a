// host_goicp.cpp -- GoICP::Register() / BuildDT() with the reference's control flow on the host and every
// error evaluation on the MI355X through include/goicp_engine.h.
//
// Host-side pieces restated from the reference (float/double mixing kept expression by expression, see
// SURVEY.md 8(a')): Initialize (jly_goicp.cpp:180-267), neighborsWeights (:1453-1498), OuterBnB (:582-876),
// ICP (:102-178), ICP3D::Run's iteration (jly_icp3d.hpp:197-311) with the 1-NN search on the device.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <array>
#include <chrono>
#include <limits>
#include <map>
#include <vector>

#include "goicp.hpp"
#include "kabsch.hpp"

namespace goicp {
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

properties string_to_prop(const std::string& t) {
    if (t == "OG") return OG; if (t == "N") return N; if (t == "O") return O; if (t == "NZ") return NZ;
    if (t == "CZ") return CZ; if (t == "CA") return CA; if (t == "DU") return DU; if (t == "OD1") return OD1;
    if (t == "C") return C;
    return OG;
}

std::ostream& operator<<(std::ostream& out, const Matrix& M) {
    if (M.m == 0 || M.n == 0) { out << "[empty matrix]"; return out; }
    char buf[64];
    for (int i = 0; i < M.m; i++) {
        for (int j = 0; j < M.n; j++) { snprintf(buf, sizeof buf, "%12.7f ", M.val[i][j]); out << buf; }
        if (i < M.m - 1) out << std::endl;
    }
    return out;
}

static bool known_prop(int c) {   // keys of the identity compatibility map (reference jly_goicp.cpp:66-73)
    return c == OG || c == N || c == O || c == NZ || c == CZ || c == CA || c == DU || c == OD1;
}

// -------------------------------------------------------------------------------------------------
// selection with the reference's swap sequence (jly_sorting.hpp:34-313)
// -------------------------------------------------------------------------------------------------
namespace {
size_t pivot_ends(const float* d, size_t s, size_t e) {
    const size_t m = (s + e) / 2;
    if (d[s] < d[e]) return d[m] < d[s] ? s : (d[m] < d[e] ? m : e);
    return d[m] < d[e] ? e : (d[m] < d[s] ? m : s);
}
size_t median3(const float* d, size_t s) {
    const float* q = d + s;
    if (q[0] < q[2]) return q[1] < q[0] ? s : (q[1] < q[2] ? s + 1 : s + 2);
    return d[1] < d[2] ? s + 2 : (d[1] < d[0] ? s + 1 : s);     // absolute indices, as in jly_sorting.hpp:74-80
}
size_t median5(float* d, size_t s) {
    float* q = d + s; float t;
    if (q[0] > q[1]) std::swap(q[0], q[1]);
    if (q[2] > q[3]) std::swap(q[2], q[3]);
    if (q[0] < q[2]) { t = q[4]; q[4] = q[0]; if (t < q[1]) q[0] = t; else { q[0] = q[1]; q[1] = t; } }
    else { t = q[4]; q[4] = q[2]; if (t < q[3]) q[2] = t; else { q[2] = q[3]; q[3] = t; } }
    if (q[0] < q[2]) return q[1] < q[2] ? s + 1 : s + 2;
    return q[0] < q[3] ? s : s + 3;
}
size_t median_of_medians(float* d, size_t s, size_t e) {
    const size_t len = e - s + 1, groups = len / 5 + (len % 5 != 0);
    size_t g = s, i = 0, mi;
    for (; i + 1 < groups; i++, g += 5) { mi = median5(d, g); std::swap(d[s + i], d[mi]); }
    switch (e - g + 1) { case 3: case 4: mi = median3(d, g); break; case 5: mi = median5(d, g); break; default: mi = g; }
    std::swap(d[s + i], d[mi]);
    if (groups > 5) return median_of_medians(d, s, s + groups - 1);
    switch (groups) { case 3: case 4: return median3(d, s); case 5: return median5(d, s); default: return s; }
}
}  // namespace

void intro_select(float* d, size_t s, size_t e, size_t k) {
    size_t prev = e - s + 1, tries = 0;
    bool quick = true;
    for (;;) {
        if (s >= e) return;
        if (e - s <= 5) {
            for (size_t i = s + 1; i <= e; i++) for (size_t j = i; j > s && d[j - 1] > d[j]; j--) std::swap(d[j - 1], d[j]);
            return;
        }
        if (quick && tries++ == 5) { const size_t len = e - s + 1; if (len * 2 > prev) quick = false; prev = len; tries = 0; }
        const size_t pi = quick ? pivot_ends(d, s, e) : median_of_medians(d, s, e);
        if (pi != s) std::swap(d[s], d[pi]);
        const float piv = d[s];
        size_t lo = s + 1, hi = e;
        for (;;) {
            while (lo < hi && piv >= d[lo]) ++lo;
            while (lo < hi && piv <= d[hi]) --hi;
            if (lo >= hi) break;
            std::swap(d[lo], d[hi]);
        }
        const size_t sp = d[lo] < piv ? lo : lo - 1;
        d[s] = d[sp]; d[sp] = piv;
        if (sp < k) s = sp + 1; else if (sp > k) e = sp - 1; else return;
    }
}

// -------------------------------------------------------------------------------------------------
// binary heaps with std::priority_queue's element order (ties on (lb,w) are frequent)
// -------------------------------------------------------------------------------------------------
namespace {
inline bool lower(const ROTNODE& p, const ROTNODE& q) { return p.lb != q.lb ? p.lb > q.lb : p.w < q.w; }   // jly_goicp.h:64-72

struct RotHeap {
    std::vector<ROTNODE> a;
    bool empty() const { return a.empty(); }
    const ROTNODE& top() const { return a[0]; }
    void up(size_t hole, const ROTNODE& v) {
        while (hole > 0) { const size_t par = (hole - 1) / 2; if (!lower(a[par], v)) break; a[hole] = a[par]; hole = par; }
        a[hole] = v;
    }
    void push(const ROTNODE& v) { a.push_back(v); up(a.size() - 1, v); }
    void pop() {
        if (a.size() > 1) {
            const long len = (long)a.size() - 1;
            const ROTNODE v = a[len];
            a[len] = a[0];
            long hole = 0, child = 0;
            while (child < (len - 1) / 2) { child = 2 * (child + 1); if (lower(a[child], a[child - 1])) child--; a[hole] = a[child]; hole = child; }
            if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; }
            up((size_t)hole, v);
        }
        a.pop_back();
    }
};
}  // namespace

// -------------------------------------------------------------------------------------------------
GoICP::GoICP() {
    // reference jly_goicp.cpp:44-56
    initNodeRot.a = (float)-PI; initNodeRot.b = (float)-PI; initNodeRot.c = (float)-PI; initNodeRot.w = (float)(2 * PI); initNodeRot.l = 0;
    initNodeRot.lb = 0; initNodeRot.ub = 0;
    initNodeTrans.x = initNodeTrans.y = initNodeTrans.z = -0.5f; initNodeTrans.w = 1.0f; initNodeTrans.lb = 0; initNodeTrans.ub = 0;
    optNodeRot = initNodeRot; optNodeTrans = initNodeTrans;
    optR = Matrix::eye(3); optT = Matrix(3, 1);
    doTrim = true;
}

GoICP::~GoICP() { Clear(); }

void GoICP::Clear() {
    if (eng_) { goicp_engine_destroy(eng_); eng_ = nullptr; }
    kd_order_.clear();
}

void GoICP::log_error(const char* tag, float v) {
    trace.emplace_back(tag, v);
    if (verbose) std::cout << "Error*: " << v << " (" << tag << ")" << std::endl;
}

// reference jly_goicp.cpp:79-99 + assignCellColor (:951-969)
void GoICP::BuildDT() {
    Clear();                      // an engine of the previous grid (it may have adopted a device-built grid in place) goes first
    status = GOICP_OK; error_message.clear();
    for (int i = 0; i < Nm; i++) if (!std::isfinite(pModel[i].x) || !std::isfinite(pModel[i].y) || !std::isfinite(pModel[i].z)) {
        status = GOICP_ERR_ARG; error_message = "model cloud holds a non-finite coordinate"; return;
    }
    std::vector<double> x(Nm), y(Nm), z(Nm);
    for (int i = 0; i < Nm; i++) { x[i] = pModel[i].x; y[i] = pModel[i].y; z[i] = pModel[i].z; }
    if (dt_mode == 1) {
        status = dt.BuildGpu(x.data(), y.data(), z.data(), Nm, device_id);
        if (status != GOICP_OK) { error_message = status == GOICP_ERR_NO_DEVICE ? "no usable HIP device for the device DT build" : "device DT build failed"; return; }
    } else {
        dt.ReleaseGpu();
        // the host builder compares candidates on integer squared lengths, proven equal to the reference's float compares up to S = 1000 (host_dt.cpp)
        if (dt.SIZE < 1 || dt.SIZE > 1000) { status = GOICP_ERR_ARG; error_message = "distTransSize outside 1..1000 (host builder); dt_mode 1 accepts up to 1290"; dt.dist.clear(); dt.nearest.clear(); dt.cell_color.clear(); dt.cell_ptr.clear(); return; }
        dt.Build(x.data(), y.data(), z.data(), Nm);
    }
    assignCellColor();
}

void GoICP::assignCellColor() {
    if (dt.gpu) {               // sparse tables of the device-built grid
        for (size_t o = 0; o < dt.occ_cells.size(); o++) {
            const int p0 = dt.occ_ptr[o], p1 = dt.occ_ptr[o + 1];
            const int prop = pModel[dt.cell_idx[p0]].c;
            bool mixed = false;
            for (int p = p0 + 1; p < p1; p++) if (pModel[dt.cell_idx[p]].c != prop) { mixed = true; break; }
            dt.occ_color[o] = mixed ? -1 : prop;
        }
        return;
    }
    const size_t ncell = dt.cell_color.size();
    for (size_t k = 0; k < ncell; k++) if (dt.cell_color[k] == -1) {
        const int p0 = dt.cell_ptr[k], p1 = dt.cell_ptr[k + 1];
        const int prop = pModel[dt.cell_idx[p0]].c;
        bool mixed = false;
        for (int p = p0 + 1; p < p1; p++) if (pModel[dt.cell_idx[p]].c != prop) { mixed = true; break; }
        dt.cell_color[k] = mixed ? -1 : prop;
    }
}

// reference jly_goicp.cpp:1453-1498 with isNeighbor (:1097-1103)
void GoICP::neighborsWeights() {
    int maxN = 0, minN = 100;
    float radius = 0.035f;
    int guard = 0;
    // O(Nd^2) pair tests per radius step: on the device for large clouds (same float/double arithmetic, integer counts)
    goicp_nbr_ctx* nb = nullptr;
    std::vector<int32_t> cnts;
    if (Nd >= 1024) {
        std::vector<float> xyz((size_t)3 * Nd);
        for (int i = 0; i < Nd; i++) { xyz[3 * i] = pData[i].x; xyz[3 * i + 1] = pData[i].y; xyz[3 * i + 2] = pData[i].z; }
        if (goicp_nbr_create(device_id, xyz.data(), Nd, &nb) != GOICP_OK) nb = nullptr;
        cnts.resize(Nd);
    }
    if (!nb && Nd >= 20) {
        // Host path for cavity-sized clouds. The reference recounts all Nd^2 pairs for every radius step until some point has
        // 19 neighbours (typically 30-70 steps). Counts only grow with the radius, so: the loop ends at the first step whose
        // limit exceeds the smallest 19th-nearest distance; the counts it leaves are those of that step; and its running
        // minimum is the minimum of the FIRST step. Same radius sequence (float accumulation), same distance expression.
        auto dist_ij = [&](int i, int j) {
            const double dx = (double)(pData[j].x - pData[i].x), dy = (double)(pData[j].y - pData[i].y), dz = (double)(pData[j].z - pData[i].z);
            return std::sqrt(dx * dx + dy * dy + dz * dz);
        };
        std::vector<double> D((size_t)Nd * Nd), row(Nd);
        double d19min = 1e300;
        for (int i = 0; i < Nd; i++) {
            int m = 0;
            for (int j = 0; j < Nd; j++) { const double d = dist_ij(i, j); D[(size_t)i * Nd + j] = d; if (j != i) row[m++] = d; }
            std::nth_element(row.begin(), row.begin() + 18, row.begin() + m);
            if (row[18] < d19min) d19min = row[18];
        }
        auto count_all = [&](double lim, bool track_min) {
            for (int i = 0; i < Nd; i++) {
                int cnt = 0;
                const double* di = &D[(size_t)i * Nd];
                for (int j = 0; j < Nd; j++) cnt += (j != i && di[j] < lim) ? 1 : 0;
                pData[i].neighbors = cnt;
                if (cnt > maxN) maxN = cnt;
                if (track_min && cnt < minN) minN = cnt;
            }
        };
        for (int step = 0;; step++) {
            const double lim = (double)std::sqrt(radius);
            const bool last = d19min < lim;
            if (step == 0 || last) count_all(lim, step == 0);
            if (last) break;
            radius = (float)(radius + 0.001);
        }
    }
    while (maxN < 19) {
        if (Nd < 20 && ++guard > 100000) break;      // the reference never terminates for Nd < 20
        const double lim = (double)std::sqrt(radius);
        if (nb && goicp_nbr_count(nb, lim, cnts.data()) == GOICP_OK) {
            for (int i = 0; i < Nd; i++) {
                const int cnt = cnts[i];
                pData[i].neighbors = cnt;
                if (cnt > maxN) maxN = cnt;
                if (cnt < minN) minN = cnt;
            }
        } else {
            for (int i = 0; i < Nd; i++) {
                int cnt = 0;
                for (int j = 0; j < Nd; j++) {
                    if (j == i) continue;
                    const double dx = (double)(pData[j].x - pData[i].x), dy = (double)(pData[j].y - pData[i].y), dz = (double)(pData[j].z - pData[i].z);
                    if (std::sqrt(dx * dx + dy * dy + dz * dz) < lim) cnt++;
                }
                pData[i].neighbors = cnt;
                if (cnt > maxN) maxN = cnt;
                if (cnt < minN) minN = cnt;
            }
        }
        radius = (float)(radius + 0.001);
    }
    if (nb) goicp_nbr_destroy(nb);
    if (minN == 0) minN = 1;
    for (int i = 0; i < Nd; i++) {
        if (pData[i].neighbors == 0) pData[i].neighbors = 1;
        const float f = ((float)minN / (float)pData[i].neighbors) * 2;
        weights[i] += f;
    }
}

// reference jly_goicp.cpp:180-267, then the tables are staged to the device
void GoICP::Initialize() {
    Clear();
    { const char* e = getenv("GOICP_DEVICE_ICP"); if (e) device_icp = atoi(e) != 0; }      // A/B knob, read once per registration
    { const char* e = getenv("GOICP_ICP_LOOP_MAX"); if (e) icp_loop_max = atoi(e); }
    status = GOICP_OK; error_message.clear();
    normData_.resize(Nd);
    for (int i = 0; i < Nd; i++) normData_[i] = std::sqrt(pData[i].x * pData[i].x + pData[i].y * pData[i].y + pData[i].z * pData[i].z);
    maxRotDis.resize((size_t)MAXROTLEVEL * Nd);
    for (int i = 0; i < MAXROTLEVEL; i++) {
        const float sigma = (float)(initNodeRot.w / std::pow(2.0, i) / 2.0);
        float maxAngle = (float)(SQRT3 * sigma);
        if (maxAngle > PI) maxAngle = (float)PI;
        for (int j = 0; j < Nd; j++) maxRotDis[(size_t)i * Nd + j] = 2 * std::sin(maxAngle / 2) * normData_[j];
    }
    minDis_.resize(Nd);
    optNodeRot = initNodeRot; optNodeTrans = initNodeTrans;
    optR = Matrix::eye(3); optT = Matrix(3, 1);
    if (doTrim) inlierNum = (int)(Nd * (1 - trimFraction)); else inlierNum = Nd;
    weights.assign(Nd, 1.0f);
    if (ponderation == 1) neighborsWeights();
    SSEThresh = MSEThresh * inlierNum;

    // ---- stage everything the hot path reads into HBM (once per registration)
    std::vector<float>& mxyz = mxyz_; std::vector<float>& dxyz = dxyz_; std::vector<float> mf, df;      // the coordinate tables stay for the ICP loop
    mxyz.resize((size_t)3 * Nm); dxyz.resize((size_t)3 * Nd);
    std::vector<int32_t> mcol(Nm), dcol(Nd);
    for (int i = 0; i < Nm; i++) { mxyz[3 * i] = pModel[i].x; mxyz[3 * i + 1] = pModel[i].y; mxyz[3 * i + 2] = pModel[i].z; mcol[i] = pModel[i].c; }
    for (int i = 0; i < Nd; i++) { dxyz[3 * i] = pData[i].x; dxyz[3 * i + 1] = pData[i].y; dxyz[3 * i + 2] = pData[i].z; dcol[i] = pData[i].c; }
    const bool need_fpfh = regularizationFPFH > 0;
    if (need_fpfh) {
        mf.assign((size_t)GOICP_FPFH_BINS * Nm, 0.f); df.assign((size_t)GOICP_FPFH_BINS * Nd, 0.f);
        for (int i = 0; i < Nm; i++) for (size_t b = 0; b < pModel[i].cfpfh.size() && b < GOICP_FPFH_BINS; b++) mf[(size_t)GOICP_FPFH_BINS * i + b] = pModel[i].cfpfh[b];
        for (int i = 0; i < Nd; i++) for (size_t b = 0; b < pData[i].cfpfh.size() && b < GOICP_FPFH_BINS; b++) df[(size_t)GOICP_FPFH_BINS * i + b] = pData[i].cfpfh[b];
    }
    goicp_engine_desc d;
    std::memset(&d, 0, sizeof d);
    d.S = dt.SIZE; d.xMin = dt.xMin; d.yMin = dt.yMin; d.zMin = dt.zMin; d.scale = dt.scale;
    // The cell tables feed the penalty terms; without them (geometric error) the only consumer is the per-point compatibility test behind the
    // "Compatibilities" count of the result, which the host then does itself on its own copy of the tables (updateCompatibilities): a 300^3
    // nearest-cell map is 108 MB of upload and 270 MB of device tables saved per registration.
    host_compat_ = !dt.gpu && !(regularization > 0) && !(regularizationFPFH > 0);
    if (dt.gpu) {               // device-resident grid: adopted in place, sparse cell tables
        d.dt_dist_dev = goicp_dtgpu_dist(dt.gpu); d.nearest_occ_dev = goicp_dtgpu_nearest(dt.gpu);
        d.n_occ = (int32_t)dt.occ_cells.size(); d.occ_cells = dt.occ_cells.data(); d.occ_color = dt.occ_color.data(); d.occ_ptr = dt.occ_ptr.data(); d.cell_idx = dt.cell_idx.data();
    } else {
        d.dt_dist = dt.dist.data();
        if (!host_compat_) { d.nearest_occ = dt.nearest.data(); d.cell_color = dt.cell_color.data(); d.cell_ptr = dt.cell_ptr.data(); d.cell_idx = dt.cell_idx.data(); }
    }
    d.Nm = Nm; d.model_xyz = mxyz.data(); d.model_color = mcol.data(); d.model_fpfh = need_fpfh ? mf.data() : nullptr;
    d.Nd = Nd; d.data_xyz = dxyz.data(); d.data_color = dcol.data(); d.data_fpfh = need_fpfh ? df.data() : nullptr;
    d.weights = weights.data(); d.maxRotDis = maxRotDis.data();
    d.inlierNum = inlierNum; d.doTrim = doTrim ? 1 : 0; d.norm = norm; d.cfpfh = cfpfh;
    d.regularization = regularization; d.regularizationNeighbors = regularizationNeighbors; d.regularizationFPFH = regularizationFPFH;
    d.device_id = device_id; d.parity_mode = parity_mode;
    status = goicp_engine_create(&d, &eng_);
    if (status != GOICP_OK) {
        eng_ = nullptr;
        error_message = status == GOICP_ERR_NO_DEVICE ? "no usable HIP device (the MI355X engine has no CPU fallback)" : "goicp_engine_create failed (unsupported configuration)";
    }
}

// reference jly_goicp.cpp:890-914 on explicit correspondence lists
int GoICP::countCompatibilities(const std::vector<int32_t>& idd, const std::vector<int32_t>& idm) const {
    int bad = 0;
    for (int i = 0; i < Nd; i++) { const int s = pData[idd[i]].c, t = pModel[idm[i]].c; if (!(known_prop(s) && s == t)) bad++; }
    return bad;
}

// reference jly_goicp.cpp:933-946: per-point test on the device, count on the host
int GoICP::updateCompatibilities() {
    double R[9], t[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = optR.val[i][j]; t[i] = optT.val[i][0]; }
    if (host_compat_) {
        // checkCompatibility per point (reference jly_goicp.cpp:937-946, 974-1000) on the host tables: (float)(R*p + t), the clamped cell, its nearest
        // occupied cell, and the colour test of that cell -- a mixed cell accepts the colours of its points, a uniform one its own (known) colour
        const int S = dt.SIZE;
        const size_t ncell = (size_t)S * S * S;
        auto cell_of = [&](float v, double mn) { return (int)(((double)v - mn) * dt.scale + 0.5); };
        int bad = 0;
        for (int i = 0; i < Nd; i++) {
            const float px = pData[i].x, py = pData[i].y, pz = pData[i].z;
            const float x = (float)(R[0] * px + R[1] * py + R[2] * pz + t[0]);
            const float y = (float)(R[3] * px + R[4] * py + R[5] * pz + t[1]);
            const float z = (float)(R[6] * px + R[7] * py + R[8] * pz + t[2]);
            int cx = cell_of(x, dt.xMin), cy = cell_of(y, dt.yMin), cz = cell_of(z, dt.zMin);
            cx = cx < 0 ? 0 : (cx >= S ? S - 1 : cx); cy = cy < 0 ? 0 : (cy >= S ? S - 1 : cy); cz = cz < 0 ? 0 : (cz >= S ? S - 1 : cz);
            const int32_t nk = dt.nearest[((size_t)cz * S + cy) * S + cx];
            bool ok = false;
            if (nk >= 0 && (size_t)nk < ncell) {
                const int col = dt.cell_color[nk], c = pData[i].c;
                if (col == -1) { for (int q = dt.cell_ptr[nk]; q < dt.cell_ptr[nk + 1] && !ok; q++) ok = pModel[dt.cell_idx[q]].c == c; }
                else if (col != -2 && known_prop(col)) ok = c == col;
            }
            if (!ok) bad++;
        }
        return bad;
    }
    std::vector<float> dist(Nd); std::vector<int32_t> ok(Nd);
    status = goicp_engine_point_eval(eng_, R, t, dist.data(), ok.data());
    int bad = 0;
    for (int i = 0; i < Nd; i++) if (!ok[i]) bad++;
    return bad;
}

// ICP3D<float>::Run (reference jly_icp3d.hpp:197-311): correspondences from the device, the 3x3 algebra here
float GoICP::icp_run(Matrix& R, Matrix& t) {
    const int n = Nd;
    const float err_diff = MSEThresh / 10000;      // jly_goicp.cpp:232
    int num;
    if (doTrim) num = (int)(n * (1 - trimFraction)); else num = n;
    // icp3d.points as packed (squared distance bits << 32 | data index) words: a non-negative float orders like its bit pattern, so the
    // numeric order of the words is the order (dis, id_data) -- the order glibc's stable merge sort leaves under the reference's comparator
    std::vector<uint64_t>& ka = sort_a_; std::vector<uint64_t>& kb = sort_b_;
    ka.resize(n); kb.resize(n);
    std::vector<int32_t> ids(n), tied(n); std::vector<float> dsq(n);
    icp_pm_.resize((size_t)3 * std::max(num, 1)); icp_pd_.resize((size_t)3 * std::max(num, 1));
    double* const p_m = icp_pm_.data(); double* const p_d = icp_pd_.data();
    const float* const mx = mxyz_.data(); const float* const dx = dxyz_.data();
    double mu_m[3] = {0, 0, 0}, mu_d[3] = {0, 0, 0};
    double Rd[9], td[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rd[3 * i + j] = R.val[i][j]; td[i] = t.val[i][0]; }
    float err = -1, err_new = 0;
    stats.icp_calls++;
    size_t iter = 0;
    bool finished = false;
    auto pack_in_index_order = [&]() {
        for (int i = 0; i < n; i++) { uint32_t k; std::memcpy(&k, &dsq[i], 4); ka[i] = ((uint64_t)k << 32) | (uint32_t)i; }
    };
    auto pack_sorted = [&]() {
        // qsort with a comparator that never reports equality (jly_icp3d.hpp:172-176,254); under glibc's merge sort ties keep
        // their order, i.e. a stable sort by dis = the numeric order of the packed words (unique, so any sorting method gives it).
        // Two stable counting passes over key bits 10-20 and 21-31 (one read builds both histograms), then an insertion pass on the
        // whole words for the low 10 key bits: runs of equal upper bits are short, and exact ties are already in index order. A cloud
        // that defeats this (thousands of distinct distances inside one 2^-13 relative band) goes to std::sort after 16 n moves.
        uint32_t cnt[2][2048];
        std::memset(cnt, 0, sizeof cnt);
        for (int i = 0; i < n; i++) {
            uint32_t k; std::memcpy(&k, &dsq[i], 4);
            ka[i] = ((uint64_t)k << 32) | (uint32_t)i; cnt[0][(k >> 10) & 2047u]++; cnt[1][k >> 21]++;
        }
        for (int p = 0; p < 2; p++) {
            uint32_t run = 0;
            for (int b = 0; b < 2048; b++) { const uint32_t c = cnt[p][b]; cnt[p][b] = run; run += c; }
            const int sh = 42 + 11 * p;
            for (int i = 0; i < n; i++) { const uint64_t v = ka[i]; kb[cnt[p][(uint32_t)(v >> sh) & 2047u]++] = v; }
            ka.swap(kb);
        }
        long moves = 0; const long budget = 16L * n;
        int i = 1;
        for (; i < n && moves <= budget; i++) {
            const uint64_t v = ka[i]; int j = i - 1;
            while (j >= 0 && ka[j] > v) { ka[j + 1] = ka[j]; j--; }
            ka[j + 1] = v; moves += i - 1 - j;
        }
        if (i < n) std::sort(ka.begin(), ka.end());
    };
    if (device_icp && n <= icp_loop_max) {
        // cavity-sized clouds: the whole loop in one launch (k_icp_loop: same sums in the same order -- for trimmed runs the order of the sort by
        // distance --, same 3x3 algebra). It hands back early only at an iteration with an exact distance tie between model points, which the
        // kd-tree order below decides.
        goicp_icp_state S;
        std::memcpy(S.R, Rd, sizeof Rd); std::memcpy(S.t, td, sizeof td);
        for (int k = 0; k < 3; k++) { S.mu_m[k] = 0; S.mu_d[k] = 0; }
        S.err = err; S.err_new = err_new; S.iters = 0; S.status = 0;
        const int rc = doTrim ? goicp_engine_icp_loop_trimmed(eng_, &S, err_diff, 10000, num, ids.data(), dsq.data())
                              : goicp_engine_icp_loop(eng_, &S, err_diff, 10000, ids.data(), dsq.data());
        if (rc == GOICP_OK) {
            stats.icp_iters += (uint64_t)S.iters; iter = (size_t)S.iters;
            std::memcpy(Rd, S.R, sizeof Rd); std::memcpy(td, S.t, sizeof td);
            for (int k = 0; k < 3; k++) { mu_m[k] = S.mu_m[k]; mu_d[k] = S.mu_d[k]; }
            err = S.err; err_new = S.err_new;
            if (S.status != 1) { if (doTrim) pack_sorted(); else pack_in_index_order(); finished = true; }
        } else if (rc != GOICP_ERR_ARG) { status = rc; return std::numeric_limits<float>::quiet_NaN(); }
    }
    for (; !finished && iter < 10000; iter++) {
        stats.icp_iters++;
        const float r00 = (float)Rd[0], r01 = (float)Rd[1], r02 = (float)Rd[2], r10 = (float)Rd[3], r11 = (float)Rd[4], r12 = (float)Rd[5];
        const float r20 = (float)Rd[6], r21 = (float)Rd[7], r22 = (float)Rd[8], t0 = (float)td[0], t1 = (float)td[1], t2 = (float)td[2];
        const double t_nn0 = now_ms();
        status = goicp_engine_icp_nn_ties(eng_, Rd, td, ids.data(), dsq.data(), tied.data());
        icp_nn_ms_ += now_ms() - t_nn0;
        if (status != GOICP_OK) return std::numeric_limits<float>::quiet_NaN();
        for (int i = 0; i < n; i++) if (tied[i] > 1) {
            // several model points at exactly this distance: the reference's kd-tree keeps the first one it visits
            // (nanoflann.hpp:1147), so ask the restated tree which one that is
            if (!kd_order_.built()) kd_order_.build(mx, Nm);
            const float x = dx[3 * i], y = dx[3 * i + 1], z = dx[3 * i + 2];
            const float q[3] = {r00 * x + r01 * y + r02 * z + t0, r10 * x + r11 * y + r12 * z + t1, r20 * x + r21 * y + r22 * z + t2};   // jly_icp3d.hpp:239-241
            ids[i] = kd_order_.nearest(q);
        }
        if (!doTrim) pack_in_index_order(); else pack_sorted();
        err_new = 0;
        for (int i = 0; i < num; i++) {
            const uint64_t v = ka[i];
            const int id = (int)(uint32_t)v, im = ids[id];
            const uint32_t kbits = (uint32_t)(v >> 32); float dis; std::memcpy(&dis, &kbits, 4);
            p_m[3 * i] = mx[3 * im]; mu_m[0] += p_m[3 * i];
            p_m[3 * i + 1] = mx[3 * im + 1]; mu_m[1] += p_m[3 * i + 1];
            p_m[3 * i + 2] = mx[3 * im + 2]; mu_m[2] += p_m[3 * i + 2];
            const float x = dx[3 * id], y = dx[3 * id + 1], z = dx[3 * id + 2];
            p_d[3 * i] = r00 * x + r01 * y + r02 * z + t0; mu_d[0] += p_d[3 * i];
            p_d[3 * i + 1] = r10 * x + r11 * y + r12 * z + t1; mu_d[1] += p_d[3 * i + 1];
            p_d[3 * i + 2] = r20 * x + r21 * y + r22 * z + t2; mu_d[2] += p_d[3 * i + 2];
            // the reference adds the double dis to the float sum and rounds back (jly_icp3d.hpp:268-276); dis holds a float value, and
            // rounding the 53-bit sum of two floats to 24 bits is the float sum itself (53 >= 2*24+2), so a float add gives the same bits
            err_new = err_new + dis;
        }
        if (err > 0 && err - err_new < err_diff * num) break;
        err = err_new;
        // the running means are divided by n (not num) and never reset (jly_icp3d.hpp:260-262,278-279)
        for (int k = 0; k < 3; k++) { mu_m[k] = mu_m[k] / (double)(float)n; mu_d[k] = mu_d[k] / (double)(float)n; }
        // the nine sums of the cross-covariance, each in index order (one pass, nine independent accumulators)
        double H[9];
        {
            double h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0, h8 = 0;
            const double m0 = mu_m[0], m1 = mu_m[1], m2 = mu_m[2], d0 = mu_d[0], d1 = mu_d[1], d2 = mu_d[2];
            for (int i = 0; i < num; i++) {
                const double a0 = p_d[3 * i] - d0, a1 = p_d[3 * i + 1] - d1, a2 = p_d[3 * i + 2] - d2;
                const double b0 = p_m[3 * i] - m0, b1 = p_m[3 * i + 1] - m1, b2 = p_m[3 * i + 2] - m2;
                h0 += a0 * b0; h1 += a0 * b1; h2 += a0 * b2; h3 += a1 * b0; h4 += a1 * b1; h5 += a1 * b2; h6 += a2 * b0; h7 += a2 * b1; h8 += a2 * b2;
            }
            H[0] = h0; H[1] = h1; H[2] = h2; H[3] = h3; H[4] = h4; H[5] = h5; H[6] = h6; H[7] = h7; H[8] = h8;
        }
        double Rn[9], Rc[9], tc[3], t_[3];
        kabsch_rotation(H, Rn);
        for (int a = 0; a < 3; a++) { double s = 0; for (int k = 0; k < 3; k++) s += Rn[3 * a + k] * mu_d[k]; t_[a] = mu_m[a] - s; }
        mul3(Rn, Rd, Rc);
        for (int a = 0; a < 3; a++) { double s = 0; for (int k = 0; k < 3; k++) s += Rn[3 * a + k] * td[k]; tc[a] = s + t_[a]; }
        std::memcpy(Rd, Rc, sizeof Rc); std::memcpy(td, tc, sizeof tc);
    }
    icp_id_data_.resize(n); icp_id_model_.resize(n); icp_dis_.resize(n);
    for (int i = 0; i < n; i++) {
        const uint64_t v = ka[i]; const int id = (int)(uint32_t)v;
        const uint32_t kbits = (uint32_t)(v >> 32); float dis; std::memcpy(&dis, &kbits, 4);
        icp_id_data_[i] = id; icp_id_model_[i] = ids[id]; icp_dis_[i] = dis;
    }
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R.val[i][j] = Rd[3 * i + j]; t.val[i][0] = td[i]; }
    return err_new;
}

// reference jly_goicp.cpp:102-178: DT lookups on the device, sums in the reference's order on the host
float GoICP::ICP(Matrix& R_icp, Matrix& t_icp) {
    if (!eng_) return std::numeric_limits<float>::quiet_NaN();
    const double t_icp0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    struct Acc { double* a; double t0; ~Acc() { *a += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; } } acc_{&icp_ms_, t_icp0};
    icp_run(R_icp, t_icp);
    if (status != GOICP_OK) return std::numeric_limits<float>::quiet_NaN();
    double R[9], t[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = R_icp.val[i][j]; t[i] = t_icp.val[i][0]; }
    std::vector<float> dist(Nd);
    status = goicp_engine_point_eval(eng_, R, t, dist.data(), nullptr);
    if (status != GOICP_OK) return std::numeric_limits<float>::quiet_NaN();
    int b0 = 0, b1 = 0;
    if (cfpfh == 1) { b0 = 0; b1 = 41; } else if (cfpfh == 2) { b0 = 0; b1 = 33; } else if (cfpfh == 3) { b0 = 33; b1 = 41; }
    float error = 0, fpfh = 0;
    for (int i = 0; i < Nd; i++) {
        if (!doTrim) {
            const float dis = weights[i] * dist[i];
            if (norm == 2) error += dis * dis;
            if (norm == 1) error += dis;
        } else minDis_[i] = dist[i];
        if (cfpfh != 0) {
            const std::vector<float>& f = pData[icp_id_data_[i]].cfpfh; const std::vector<float>& g = pModel[icp_id_model_[i]].cfpfh;
            float d = 0;
            for (int b = b0; b < b1; b++) d += std::fabs((b < (int)f.size() ? f[b] : 0.f) - (b < (int)g.size() ? g[b] : 0.f));
            fpfh += d;
        }
    }
    fpfh = fpfh / Nd;
    if (regularization > 0) { const int inc = countCompatibilities(icp_id_data_, icp_id_model_); error += regularization * (inc * inc); }
    if (regularizationFPFH > 0) error += regularizationFPFH * (fpfh * fpfh);
    if (doTrim) {
        intro_select(minDis_.data(), 0, (size_t)(Nd - 1), (size_t)(inlierNum - 1));
        for (int i = 0; i < inlierNum; i++) error += minDis_[i] * minDis_[i];
    }
    return error;
}

static void axis_angle(float v1, float v2, float v3, float* R) {   // reference jly_goicp.cpp:718-747
    const float t = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    if (!(t > 0)) { R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0; return; }
    v1 /= t; v2 /= t; v3 /= t;
    const float ct = std::cos(t), ct2 = 1 - ct, st = std::sin(t);
    const float t121 = v1 * v2 * ct2, t122 = v3 * st, t131 = v1 * v3 * ct2, t132 = v2 * st, t231 = v2 * v3 * ct2, t232 = v1 * st;
    R[0] = ct + v1 * v1 * ct2; R[1] = t121 - t122; R[2] = t131 + t132;
    R[3] = t121 + t122; R[4] = ct + v2 * v2 * ct2; R[5] = t231 - t232;
    R[6] = t131 - t132; R[7] = t231 + t232; R[8] = ct + v3 * v3 * ct2;
}

static goicp_bnb_job make_job(const float* R9, int level, float E, float sse, const TRANSNODE& init) {
    goicp_bnb_job j;
    std::memcpy(j.R, R9, sizeof j.R);
    j.rot_level = level; j.optError_in = E; j.SSEThresh = sse;
    j.init[0] = init.x; j.init[1] = init.y; j.init[2] = init.z; j.init[3] = init.w; j.init_lb = init.lb;
    j.max_pops = 0; j.flags = 0;
    return j;
}

float GoICP::InnerBnB(const float* R9, int rot_level, TRANSNODE* out) {
    if (!eng_) return std::numeric_limits<float>::quiet_NaN();
    goicp_bnb_job j = make_job(R9, rot_level, optError, SSEThresh, initNodeTrans);
    goicp_bnb_result r;
    status = goicp_engine_inner_bnb(eng_, &j, &r);
    stats.inner_calls++; stats.inner_launches++; stats.inner_jobs_launched++; stats.cube_evals += r.cube_evals; stats.trans_pops += r.pops; stats.pops_launched += r.pops;
    if (out && r.found) { out->x = r.node[0]; out->y = r.node[1]; out->z = r.node[2]; out->w = r.node[3]; }
    return r.optErrorT;
}


namespace {
struct PopPlan {
    ROTNODE ch[8]; float Rj[8][9]; bool live[8];
    goicp_bnb_result res[16]; goicp_bnb_job job[16]; float res_E[16]; bool have[16];
};
typedef std::array<uint32_t, 5> PlanKey;
PlanKey plan_key(const ROTNODE& n) {
    PlanKey k; float f[4] = {n.a, n.b, n.c, n.w};
    for (int i = 0; i < 4; i++) std::memcpy(&k[i], &f[i], 4);
    k[4] = (uint32_t)n.l;
    return k;
}
// children of a rotation node, their pi-ball test and rotation matrices (reference jly_goicp.cpp:703-747)
void expand_plan(const ROTNODE& par, PopPlan& pl) {
    for (int j = 0; j < 8; j++) {
        ROTNODE& nd = pl.ch[j];
        nd.w = par.w / 2; nd.l = par.l + 1;
        nd.a = par.a + (j & 1) * nd.w; nd.b = par.b + (j >> 1 & 1) * nd.w; nd.c = par.c + (j >> 2 & 1) * nd.w;
        nd.ub = nd.lb = 0;
        const float v1 = nd.a + nd.w / 2, v2 = nd.b + nd.w / 2, v3 = nd.c + nd.w / 2;
        pl.live[j] = !((double)std::sqrt(v1 * v1 + v2 * v2 + v3 * v3) - SQRT3 * nd.w / 2 > PI);      // pi-ball skip (:723)
        if (pl.live[j]) axis_angle(v1, v2, v3, pl.Rj[j]);
    }
    for (int k = 0; k < 16; k++) pl.have[k] = false;
}
}  // namespace


// reference jly_goicp.cpp:582-876
float GoICP::OuterBnB() {
    std::map<PlanKey, PopPlan> plan_cache;
    RotHeap queueRot;
    float lb = 0, ub = 0, error;
    const float nanv = std::numeric_limits<float>::quiet_NaN();

    // initial error (jly_goicp.cpp:602-626): lookups on the device, sum in the reference's order
    {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
        std::vector<float> dist(Nd);
        status = goicp_engine_point_eval(eng_, I, z3, dist.data(), nullptr);
        if (status != GOICP_OK) return nanv;
        for (int i = 0; i < Nd; i++) minDis_[i] = weights[i] * dist[i];
        if (doTrim) intro_select(minDis_.data(), 0, (size_t)(Nd - 1), (size_t)(inlierNum - 1));
        optError = 0;
        for (int i = 0; i < inlierNum; i++) { if (norm == 2) optError += minDis_[i] * minDis_[i]; if (norm == 1) optError += minDis_[i]; }
        if (regularization > 0) optError += regularization * (Nd * Nd);
        if (regularizationFPFH > 0) optError += regularizationFPFH * (100 * 8 * 100 * 8);
        log_error("Init", optError);
    }
    Matrix R_icp = optR, t_icp = optT;
    error = ICP(R_icp, t_icp);
    if (status != GOICP_OK) return nanv;
    opt_id_data_ = icp_id_data_; opt_id_model_ = icp_id_model_;         // icp3d.optPoints = icp3d.points (:635)
    if (error < optError) {
        optError = error; optR = R_icp; optT = t_icp;
        optComp = countCompatibilities(opt_id_data_, opt_id_model_);
        log_error("ICP", error);
    }
    queueRot.push(initNodeRot);
    while (true) {
        if (queueRot.empty()) break;
        const ROTNODE par = queueRot.top();
        queueRot.pop();
        stats.rot_pops++;
        if ((optError - par.lb) <= SSEThresh) break;
        if (wall_budget_s > 0 && (now_ms() - t_start_ms_) * 1e-3 > wall_budget_s) { budget_hit = true; break; }
        // The 16 InnerBnB calls of a rotation pop (8 children x {ub, lb}) are pure functions of (child rotation, level,
        // incumbent), so they can be evaluated ahead of time: one launch carries the jobs of this pop AND of the next
        // `speculate`-1 nodes at the head of the rotation queue. Results are cached per node and consumed strictly in the
        // reference's order; a result is used only if the incumbent it was computed with is still current and the job ran to
        // completion (speculative jobs carry a pop budget) -- otherwise it is launched again. Search results are unchanged.
        PopPlan plan;
        {
            auto it = plan_cache.find(plan_key(par));
            if (it != plan_cache.end()) { plan = it->second; plan_cache.erase(it); }
            else expand_plan(par, plan);
        }
        ROTNODE* ch = plan.ch; float (*Rj)[9] = plan.Rj; bool* live = plan.live;
        auto add_jobs = [&](PopPlan& pl, int first, bool spec, std::vector<goicp_bnb_job>& jobs, std::vector<std::pair<PopPlan*, int>>& where) {
            for (int k = first; k < 16; k++) {
                const int j = k >> 1;
                if (!pl.live[j]) continue;
                if (pl.have[k] && pl.res_E[k] == optError && pl.res[k].status == GOICP_OK) continue;
                goicp_bnb_job jb = make_job(pl.Rj[j], (k & 1) ? pl.ch[j].l : -1, optError, SSEThresh, initNodeTrans);
                if (spec || !jobs.empty()) jb.max_pops = spec_pop_budget;      // only the very first job is certain to be needed as is
                else if (search_pop_budget) { jb.max_pops = search_pop_budget; jb.flags = GOICP_JOB_REGROW; }
                jobs.push_back(jb); where.emplace_back(&pl, k);
                if (speculate <= 0) break;
            }
        };
        auto launch_from = [&](int first) -> bool {
            std::vector<goicp_bnb_job> jobs; std::vector<std::pair<PopPlan*, int>> where;
            add_jobs(plan, first, false, jobs, where);
            if (speculate > 1 && !queueRot.empty()) {
                RotHeap peek = queueRot;
                // Launch size: up to eight times the jobs the device holds at once (2048 at most) -- most jobs are short, so the later workgroups start
                // while the few long searches of the launch are still running (a hard cavity pair with 37 M cube evaluations: 1.56 s
                // at one wave of workgroups, 0.98 s at four, 0.89 s here). The depth grows with the length of the search: easy pairs end within a few
                // launches and should not pay for jobs that are never consumed (8 nodes for launches 0-2, then 16, 32, 64, and
                // `speculate` = 128 from the 24th launch on).
                const int max_jobs = std::min(2048, std::max(32, 8 * goicp_engine_resident_jobs(eng_)));
                const int depth = std::min(speculate, stats.inner_launches >= 24 ? 128 : 8 << std::min(3, (int)(stats.inner_launches / 3)));
                for (int m = 1; m < depth && !peek.empty() && (int)jobs.size() + 16 <= max_jobs; m++) {
                    const ROTNODE nd = peek.top(); peek.pop();
                    if ((optError - nd.lb) <= SSEThresh) break;            // the search would stop before expanding it
                    auto key = plan_key(nd);
                    auto it = plan_cache.find(key);
                    if (it == plan_cache.end()) { it = plan_cache.emplace(key, PopPlan()).first; expand_plan(nd, it->second); }
                    add_jobs(it->second, 0, true, jobs, where);
                }
            }
            if (jobs.empty()) return true;
            std::vector<goicp_bnb_result> out(jobs.size());
            const int rc = goicp_engine_inner_bnb_batch(eng_, jobs.data(), (int)jobs.size(), out.data());
            stats.inner_launches++; stats.inner_jobs_launched += jobs.size();
            if (rc != GOICP_OK && rc != GOICP_ERR_LIMIT && rc != GOICP_ERR_OVERFLOW) { status = rc; return false; }
            for (size_t q = 0; q < jobs.size(); q++) {
                stats.pops_launched += out[q].pops;
                PopPlan& pl = *where[q].first; const int k = where[q].second;
                pl.res[k] = out[q]; pl.job[k] = jobs[q]; pl.res_E[k] = optError; pl.have[k] = true;
            }
            return true;
        };
        auto get = [&](int k) -> const goicp_bnb_result* {
            if (!(plan.have[k] && plan.res_E[k] == optError && plan.res[k].status == GOICP_OK)) {
                plan.have[k] = false;
                if (!launch_from(k)) return nullptr;
                if (plan.res[k].status == GOICP_ERR_LIMIT && search_pop_budget && plan.job[k].max_pops == search_pop_budget) {
                    budget_hit = true;                   // this search is beyond the caller's budget: stop here with the best pose so far
                    stats.trans_pops += plan.res[k].pops; stats.cube_evals += plan.res[k].cube_evals;
                    return nullptr;
                }
                if (plan.res[k].status != GOICP_OK) {
                    status = plan.res[k].status;
                    char buf[160];
                    snprintf(buf, sizeof buf, "InnerBnB job failed (status %d, rot level %d, incumbent %g, %u pops)", plan.res[k].status, plan.job[k].rot_level,
                             (double)plan.job[k].optError_in, plan.res[k].pops);
                    error_message = buf;
                    return nullptr;
                }
            }
            stats.inner_calls++; stats.cube_evals += plan.res[k].cube_evals; stats.trans_pops += plan.res[k].pops;
            if (record_jobs) { job_log.push_back(plan.job[k]); result_log.push_back(plan.res[k]); }
            return &plan.res[k];
        };
        for (int j = 0; j < 8; j++) {
            if (!live[j]) continue;
            ROTNODE& nodeRot = ch[j];
            const goicp_bnb_result* r = get(2 * j);
            if (!r) return budget_hit ? optError : nanv;
            ub = r->optErrorT;
            if (ub < optError) {
                TRANSNODE nodeTrans = optNodeTrans;
                if (r->found) { nodeTrans.x = r->node[0]; nodeTrans.y = r->node[1]; nodeTrans.z = r->node[2]; nodeTrans.w = r->node[3]; }
                optError = ub; optNodeRot = nodeRot; optNodeTrans = nodeTrans;
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) optR.val[a][b] = Rj[j][3 * a + b];
                optT.val[0][0] = optNodeTrans.x + optNodeTrans.w / 2;
                optT.val[1][0] = optNodeTrans.y + optNodeTrans.w / 2;
                optT.val[2][0] = optNodeTrans.z + optNodeTrans.w / 2;
                optComp = updateCompatibilities();
                if (status != GOICP_OK) return nanv;
                log_error("BNB", optError);
                R_icp = optR; t_icp = optT;
                error = ICP(R_icp, t_icp);
                if (status != GOICP_OK) return nanv;
                if (error < optError) {
                    optError = error; optR = R_icp; optT = t_icp;
                    opt_id_data_ = icp_id_data_; opt_id_model_ = icp_id_model_;
                    optComp = countCompatibilities(opt_id_data_, opt_id_model_);
                    log_error("ICP", error);
                }
                // drop queued rotation cubes that can no longer win (:843-853)
                RotHeap q2;
                while (!queueRot.empty()) { const ROTNODE nd = queueRot.top(); queueRot.pop(); if (nd.lb < optError) q2.push(nd); else break; }
                queueRot = q2;
                plan_cache.clear();          // every cached result was computed with the old incumbent
            }
            r = get(2 * j + 1);
            if (!r) return budget_hit ? optError : nanv;
            lb = r->optErrorT;
            if (lb >= optError) continue;
            nodeRot.ub = ub; nodeRot.lb = lb;
            queueRot.push(nodeRot);
        }
    }
    return optError;
}

float GoICP::Register() {
    trace.clear(); stats = RegisterStats(); job_log.clear(); result_log.clear(); icp_ms_ = 0; icp_nn_ms_ = 0;
    const bool timing = getenv("GOICP_TIMING") != nullptr;
    const double t0 = now_ms();
    t_start_ms_ = t0; budget_hit = false;
    // non-finite coordinates would walk the (reference-order) selection and sorting loops out of their arrays: refuse them
    for (int i = 0; i < Nd; i++) if (!std::isfinite(pData[i].x) || !std::isfinite(pData[i].y) || !std::isfinite(pData[i].z)) {
        status = GOICP_ERR_ARG; error_message = "data cloud holds a non-finite coordinate"; return std::numeric_limits<float>::quiet_NaN();
    }
    for (int i = 0; i < Nm; i++) if (!std::isfinite(pModel[i].x) || !std::isfinite(pModel[i].y) || !std::isfinite(pModel[i].z)) {
        status = GOICP_ERR_ARG; error_message = "model cloud holds a non-finite coordinate"; return std::numeric_limits<float>::quiet_NaN();
    }
    Initialize();
    const double t1 = now_ms();
    if (!eng_) return std::numeric_limits<float>::quiet_NaN();
    OuterBnB();
    const double t2 = now_ms();
    if (timing) {
        uint64_t rounds = 0, launched = 0;
        goicp_engine_search_stats(eng_, &rounds, &launched);
        fprintf(stderr, "[goicp timing] Initialize+engine %.1f ms, OuterBnB %.1f ms (ICP host+device %.1f ms, of which 1-NN launch+wait %.1f ms); search rounds %llu, cube evals launched %llu, consumed %llu; "
                        "launches %llu, jobs launched %llu consumed %llu, queue pops launched %llu consumed %llu, rot pops %llu\n",
                t1 - t0, t2 - t1, icp_ms_, icp_nn_ms_, (unsigned long long)rounds, (unsigned long long)launched, (unsigned long long)stats.cube_evals,
                (unsigned long long)stats.inner_launches, (unsigned long long)stats.inner_jobs_launched, (unsigned long long)stats.inner_calls,
                (unsigned long long)stats.pops_launched, (unsigned long long)stats.trans_pops, (unsigned long long)stats.rot_pops);
    }
    double ms = 0; uint64_t ce = 0, lk = 0;
    goicp_engine_stats(eng_, &ce, &lk, &ms);
    stats.engine_kernel_ms = ms;
    goicp_engine_search_stats(eng_, &stats.search_rounds, &stats.cube_evals_launched);
    stats.slab_regrowths = goicp_engine_slab_regrowths(eng_);
    stats.seconds = (now_ms() - t0) * 1e-3;
    Clear();
    if (status != GOICP_OK) return std::numeric_limits<float>::quiet_NaN();
    return optError;
}

}  // namespace goicp
