// host_dt.cpp -- distance transform + cell tables, host side (staged once per pair into HBM).
//
// Restates DT3D::Build (reference jly_3ddt.cpp:897-1137): 8SED-style vector propagation of (v,h,d)
// offsets in two x-sweeps with four z/y sub-scans each (DEuclidean, jly_3ddt.cpp:716-750), distance =
// sqrt(v^2+h^2+d^2)/scale, per-cell point lists and the nearest-occupied-cell guess from |v|,|h|,|d|.
// Where the reference returns an uninitialised candidate (jly_3ddt.cpp:70-73 and siblings) this builder
// returns the "infinite" cell, which is what the compiled reference produces in practice (bit-identical
// grids at 20^3..300^3, see tests/test_oracle_golden.py and tests/test_host_vs_oracle.py).
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <thread>
#include <vector>

#include "goicp.hpp"

namespace goicp {
namespace {

constexpr int DT_INF = 32767;   // reference jly_3ddt.h:20


// one neighbour of a propagation mask: read offset (dz,dy,dx), increments (id,ih,iv), bounds-guard offset
struct Tap { signed char dz, dy, dx, id, ih, iv, gz, gy, gx; };

// jly_3ddt.cpp:508-712
const Tap kF1[14] = {{-1, -1, -1, 1, 1, 1, -1, -1, -1}, {0, -1, -1, 0, 1, 1, 0, -1, -1}, {1, -1, -1, 1, 1, 1, 1, -1, -1},
                     {-1, 0, -1, 1, 0, 1, -1, 0, -1},   {0, 0, -1, 0, 0, 1, 0, 0, -1},   {1, 0, -1, 1, 0, 1, 1, 0, -1},
                     {-1, 1, -1, 1, 1, 1, -1, 1, -1},   {0, 1, -1, 0, 1, 1, 0, 1, -1},   {1, 1, -1, 1, 1, 1, 1, 1, -1},
                     {-1, -1, 0, 1, 1, 0, -1, -1, 0},   {0, -1, 0, 0, 1, 0, 0, -1, 0},   {1, -1, 0, 1, 1, 0, 1, -1, 0},
                     {0, 0, 0, 0, 0, 0, 0, 0, 0},       {-1, 0, 0, 1, 0, 0, -1, 0, 0}};
const Tap kF2[2] = {{1, 0, 0, 1, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}};                      // :260-293
// the second tap reads [z+1][y][x] under a y+1 guard in the reference (:88-93); kept
const Tap kF3[5] = {{1, 0, 0, 1, 0, 0, 1, 0, 0}, {1, 0, 0, 1, 1, 0, 1, 1, 0}, {0, 1, 0, 0, 1, 0, 0, 1, 0},
                    {0, 0, 0, 0, 0, 0, 0, 0, 0}, {-1, 1, 0, 1, 1, 0, -1, 1, 0}};                    // :57-137
const Tap kF4[2] = {{-1, 0, 0, 1, 0, 0, -1, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}};                    // :139-175
const Tap kB1[14] = {{-1, -1, 1, 1, 1, 1, -1, -1, 1}, {0, -1, 1, 0, 1, 1, 0, -1, 1}, {1, -1, 1, 1, 1, 1, 1, -1, 1},
                     {-1, 0, 1, 1, 0, 1, -1, 0, 1},   {0, 0, 1, 0, 0, 1, 0, 0, 1},   {1, 0, 1, 1, 0, 1, 1, 0, 1},
                     {-1, 1, 1, 1, 1, 1, -1, 1, 1},   {0, 1, 1, 0, 1, 1, 0, 1, 1},   {1, 1, 1, 1, 1, 1, 1, 1, 1},
                     {1, 0, 0, 1, 0, 0, 1, 0, 0},     {1, 0, 0, 1, 1, 0, 1, 1, 0},   {0, 1, 0, 0, 1, 0, 0, 1, 0},
                     {0, 0, 0, 0, 0, 0, 0, 0, 0},     {-1, 1, 0, 1, 1, 0, -1, 1, 0}};               // :301-506
const Tap kB3[5] = {{-1, -1, 0, 1, 1, 0, -1, -1, 0}, {0, -1, 0, 0, 1, 0, 0, -1, 0}, {1, -1, 0, 1, 1, 0, 1, -1, 0},
                    {0, 0, 0, 0, 0, 0, 0, 0, 0}, {-1, 0, 0, 1, 0, 0, -1, 0, 0}};                    // :177-258

// ---- storage -----------------------------------------------------------------------------------------------------
// The propagation visits z in its innermost loops, so the working copy is kept z-fastest (the reference's own layout makes
// every inner step a S^2-cell stride), as three int16 planes with a one-cell border of "infinite" cells that replaces the
// bounds tests. Candidates are compared on the exact integer v^2+h^2+d^2: the reference compares (float)sqrt((double)sum),
// which is strictly monotone in the sum for the sums that can win (sqrt < 2048, i.e. S <= 1000), and an "infinite"
// candidate (components >= 32767) can never be strictly smaller than the initial best 3*32767^2, exactly as its NaN /
// >32767 lengths never win in the reference.
constexpr uint32_t S_INF = 3u * 32767u * 32767u;

struct Grid {
    int S, P;                                  // P = S + 2 (padded)
    std::vector<int16_t> v, h, d;
    size_t idx(int z, int y, int x) const { return ((size_t)(x + 1) * P + (size_t)(y + 1)) * P + (size_t)(z + 1); }
    int16_t* row(std::vector<int16_t>& a, int y, int x) { return a.data() + ((size_t)(x + 1) * P + (size_t)(y + 1)) * P + 1; }
    bool zero(int z, int y, int x) const { const size_t k = idx(z, y, x); return v[k] == 0 && h[k] == 0 && d[k] == 0; }
};

struct RowBuf {                                // best candidate per z of a run of taps that do not depend on the running scan
    std::vector<uint32_t> s; std::vector<int32_t> v, h, d;
    void reserve(int n) { s.resize(n); v.resize(n); h.resize(n); d.resize(n); }
    void reset(int n) { for (int z = 0; z < n; z++) { s[z] = S_INF; v[z] = DT_INF; h[z] = DT_INF; d[z] = DT_INF; } }
};

// one tap applied to a whole z-row (vectorisable: contiguous int16 loads widened to 32-bit lanes, integer arithmetic, blends)
__attribute__((target_clones("avx2", "default")))
void relax_row(uint32_t* __restrict__ bs, int32_t* __restrict__ bv, int32_t* __restrict__ bh, int32_t* __restrict__ bd,
                      const int16_t* __restrict__ sv, const int16_t* __restrict__ sh, const int16_t* __restrict__ sd, int iv, int ih, int id, int n) {
#pragma clang loop vectorize(enable) interleave(enable)
    for (int z = 0; z < n; z++) {
        const int32_t cv = sv[z] + iv, ch = sh[z] + ih, cd = sd[z] + id;
        const uint32_t cs = (uint32_t)(cv * cv) + (uint32_t)(ch * ch) + (uint32_t)(cd * cd);
        const bool better = cs < bs[z];
        bs[z] = better ? cs : bs[z];
        bv[z] = better ? cv : bv[z];
        bh[z] = better ? ch : bh[z];
        bd[z] = better ? cd : bd[z];
    }
}

// A reference mask split once into: run A (taps before the first tap that reads the row being scanned), the sequential
// taps (dy==dx==0, dz!=0: they see cells already updated by this scan), run C (taps after them). Tap order is kept inside
// each part and across parts, so ties resolve as in the reference (earlier tap wins).
struct ScanPlan {
    struct T { std::ptrdiff_t off; int iv, ih, id, gy, dz; };
    std::vector<T> A, Sq, C;
    bool ascending;
};
template <int N>
ScanPlan make_plan(const Tap (&taps)[N], bool ascending, int P) {
    ScanPlan pl; pl.ascending = ascending;
    int firstSeq = N, lastSeq = -1;
    for (int k = 0; k < N; k++) if (taps[k].dy == 0 && taps[k].dx == 0 && taps[k].dz != 0) { if (firstSeq == N) firstSeq = k; lastSeq = k; }
    for (int k = 0; k < N; k++) {
        const Tap& t = taps[k];
        ScanPlan::T e{((std::ptrdiff_t)t.dx * P + t.dy) * P + t.dz, t.iv, t.ih, t.id, t.gy, t.dz};
        const bool seq = t.dy == 0 && t.dx == 0 && t.dz != 0;
        if (seq) pl.Sq.push_back(e); else if (k < firstSeq) pl.A.push_back(e); else pl.C.push_back(e);
    }
    (void)lastSeq;
    return pl;
}

void scan_row(Grid& g, const ScanPlan& pl, int y, int x, RowBuf& A, RowBuf& C) {
    const int S = g.S;
    int16_t* rv = g.row(g.v, y, x); int16_t* rh = g.row(g.h, y, x); int16_t* rd = g.row(g.d, y, x);
    A.reset(S);
    for (const ScanPlan::T& t : pl.A)
        if (y + t.gy >= 0 && y + t.gy < S) relax_row(A.s.data(), A.v.data(), A.h.data(), A.d.data(), rv + t.off, rh + t.off, rd + t.off, t.iv, t.ih, t.id, S);
    const bool haveC = !pl.C.empty();
    if (haveC) {
        C.reset(S);
        for (const ScanPlan::T& t : pl.C)
            if (y + t.gy >= 0 && y + t.gy < S) relax_row(C.s.data(), C.v.data(), C.h.data(), C.d.data(), rv + t.off, rh + t.off, rd + t.off, t.iv, t.ih, t.id, S);
    }
    // sequential taps live in this row (guards on y decided once)
    int ns = 0; int sdz[2], siv[2], sih[2], sid[2];
    for (const ScanPlan::T& t : pl.Sq) if (y + t.gy >= 0 && y + t.gy < S) { sdz[ns] = t.dz; siv[ns] = t.iv; sih[ns] = t.ih; sid[ns] = t.id; ns++; }
    const int z0 = pl.ascending ? 0 : S - 1, z1 = pl.ascending ? S : -1, dzs = pl.ascending ? 1 : -1;
    const uint32_t* As = A.s.data(); const int32_t* Av = A.v.data(); const int32_t* Ah = A.h.data(); const int32_t* Ad = A.d.data();
    const uint32_t* Cs = C.s.data(); const int32_t* Cv = C.v.data(); const int32_t* Ch = C.h.data(); const int32_t* Cd = C.d.data();
    for (int z = z0; z != z1; z += dzs) {
        uint32_t bs = As[z]; int32_t bv = Av[z], bh = Ah[z], bd = Ad[z];
        for (int k = 0; k < ns; k++) {
            const int32_t cv = rv[z + sdz[k]] + siv[k], ch = rh[z + sdz[k]] + sih[k], cd = rd[z + sdz[k]] + sid[k];
            const uint32_t cs = (uint32_t)(cv * cv) + (uint32_t)(ch * ch) + (uint32_t)(cd * cd);
            if (cs < bs) { bs = cs; bv = cv; bh = ch; bd = cd; }
        }
        if (haveC && Cs[z] < bs) { bs = Cs[z]; bv = Cv[z]; bh = Ch[z]; bd = Cd[z]; }
        rv[z] = (int16_t)bv; rh[z] = (int16_t)bh; rd[z] = (int16_t)bd;
    }
}

void propagate_all(Grid& g) {   // DEuclidean, jly_3ddt.cpp:716-750
    const int S = g.S, P = g.P;
    RowBuf A, C; A.reserve(S); C.reserve(S);
    const ScanPlan pF1 = make_plan(kF1, true, P), pF2 = make_plan(kF2, false, P), pF3 = make_plan(kF3, false, P), pF4 = make_plan(kF4, true, P);
    const ScanPlan pB1 = make_plan(kB1, false, P), pB3 = make_plan(kB3, true, P);
    for (int x = 0; x < S; x++) {
        for (int y = 0; y < S; y++) { scan_row(g, pF1, y, x, A, C); scan_row(g, pF2, y, x, A, C); }
        for (int y = S - 1; y >= 0; y--) { scan_row(g, pF3, y, x, A, C); scan_row(g, pF4, y, x, A, C); }
    }
    for (int x = S - 1; x >= 0; x--) {
        for (int y = S - 1; y >= 0; y--) { scan_row(g, pB1, y, x, A, C); scan_row(g, pF4, y, x, A, C); }
        for (int y = 0; y < S; y++) { scan_row(g, pB3, y, x, A, C); scan_row(g, pF2, y, x, A, C); }
    }
}

// sign combinations (sx,sy,sz) in the order the reference tries them, jly_3ddt.cpp:1012-1132
const signed char kS3[8][3] = {{1, 1, 1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
const signed char kSyz[4][3] = {{0, 1, 1}, {0, 1, -1}, {0, -1, 1}, {0, -1, -1}};
const signed char kSxz[4][3] = {{1, 0, 1}, {1, 0, -1}, {-1, 0, 1}, {-1, 0, -1}};
const signed char kSxy[4][3] = {{1, 1, 0}, {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0}};
const signed char kSx[2][3] = {{1, 0, 0}, {-1, 0, 0}};
const signed char kSy[2][3] = {{0, 1, 0}, {0, -1, 0}};
const signed char kSz[2][3] = {{0, 0, 1}, {0, 0, -1}};

int32_t nearest_occupied(const Grid& g, int z, int y, int x) {
    const int S = g.S;
    const size_t me = g.idx(z, y, x);
    const int32_t self = (int32_t)(((size_t)z * S + y) * S + x);
    if (g.v[me] == 0 && g.h[me] == 0 && g.d[me] == 0) return self;
    const int xD = g.v[me], yD = g.h[me], zD = g.d[me];
    const signed char(*tab)[3]; int nt;
    if (xD == 0) { if (yD == 0) { tab = kSz; nt = 2; } else if (zD == 0) { tab = kSy; nt = 2; } else { tab = kSyz; nt = 4; } }
    else if (yD == 0) { if (zD == 0) { tab = kSx; nt = 2; } else { tab = kSxz; nt = 4; } }
    else if (zD == 0) { tab = kSxy; nt = 4; }
    else { tab = kS3; nt = 8; }
    for (int k = 0; k < nt; k++) {
        const int xx = x + tab[k][0] * xD, yy = y + tab[k][1] * yD, zz = z + tab[k][2] * zD;
        if ((unsigned)xx >= (unsigned)S || (unsigned)yy >= (unsigned)S || (unsigned)zz >= (unsigned)S) continue;
        if (g.zero(zz, yy, xx)) return (int32_t)(((size_t)zz * S + yy) * S + xx);
    }
    return self;
}

}  // namespace

void DT3D::Build(const double* _x, const double* _y, const double* _z, int num) {
    const int S = SIZE;
    const size_t ncell = (size_t)S * S * S;
    xMin = xMax = _x[0]; yMin = yMax = _y[0]; zMin = zMax = _z[0];
    for (int i = 1; i < num; i++) {
        if (xMin > _x[i]) xMin = _x[i];
        if (xMax < _x[i]) xMax = _x[i];
        if (yMin > _y[i]) yMin = _y[i];
        if (yMax < _y[i]) yMax = _y[i];
        if (zMin > _z[i]) zMin = _z[i];
        if (zMax < _z[i]) zMax = _z[i];
    }
    const double xC = (xMin + xMax) / 2, yC = (yMin + yMax) / 2, zC = (zMin + zMax) / 2;
    xMin = xC - expandFactor * (xMax - xC); xMax = xC + expandFactor * (xMax - xC);
    yMin = yC - expandFactor * (yMax - yC); yMax = yC + expandFactor * (yMax - yC);
    zMin = zC - expandFactor * (zMax - zC); zMax = zC + expandFactor * (zMax - zC);
    double mx = xMax - xMin > yMax - yMin ? xMax - xMin : yMax - yMin;
    mx = mx > zMax - zMin ? mx : zMax - zMin;
    xMin = xC - mx / 2; xMax = xC + mx / 2;
    yMin = yC - mx / 2; yMax = yC + mx / 2;
    zMin = zC - mx / 2; zMax = zC + mx / 2;
    scale = S / mx;

    Grid g; g.S = S; g.P = S + 2;
    const size_t np = (size_t)g.P * g.P * g.P;
    g.v.assign(np, (int16_t)DT_INF); g.h.assign(np, (int16_t)DT_INF); g.d.assign(np, (int16_t)DT_INF);
    cell_color.assign(ncell, -2);
    cell_ptr.assign(ncell + 1, 0);
    cell_idx.assign((size_t)(num > 0 ? num : 1), 0);
    std::vector<int64_t> cell_of((size_t)num, -1);
    for (int i = 0; i < num; i++) {
        const int x = (int)((_x[i] - xMin) * scale + 0.5), y = (int)((_y[i] - yMin) * scale + 0.5), z = (int)((_z[i] - zMin) * scale + 0.5);
        if ((unsigned)x >= (unsigned)S || (unsigned)y >= (unsigned)S || (unsigned)z >= (unsigned)S) continue;   // (the reference would index out of bounds)
        const size_t k = ((size_t)z * S + y) * S + x;
        cell_of[i] = (int64_t)k; cell_color[k] = -1; cell_ptr[k + 1]++;
        const size_t q = g.idx(z, y, x);
        g.v[q] = 0; g.h[q] = 0; g.d[q] = 0;
    }
    for (size_t k = 0; k < ncell; k++) cell_ptr[k + 1] += cell_ptr[k];
    {
        std::vector<int32_t> fill(ncell, 0);
        for (int i = 0; i < num; i++) if (cell_of[i] >= 0) { const size_t k = (size_t)cell_of[i]; cell_idx[cell_ptr[k] + fill[k]++] = i; }
    }
    const auto tA = std::chrono::steady_clock::now();
    propagate_all(g);
    const auto tB = std::chrono::steady_clock::now();
    dist.resize(ncell); nearest.resize(ncell);
    // distances + nearest-occupied map: independent per cell -> host threads over z slabs
    auto finish_slab = [&](int ylo, int yhi) {
        // the working copy is z-fastest, the outputs are x-fastest: go through 32x32 (x,z) tiles so that both sides move whole cache lines
        constexpr int TB = 32;
        float td[TB][TB]; int32_t tn[TB][TB];
        for (int y = ylo; y < yhi; y++) for (int xb = 0; xb < S; xb += TB) for (int zb = 0; zb < S; zb += TB) {
            const int xe = std::min(S, xb + TB), ze = std::min(S, zb + TB);
            for (int x = xb; x < xe; x++) for (int z = zb; z < ze; z++) {
                const size_t q = g.idx(z, y, x);
                const int cv = g.v[q], ch = g.h[q], cd = g.d[q];
                // (float)sqrt((double)(v^2+h^2+d^2)) as stored by the propagation, then /scale in double (jly_3ddt.cpp:79,1003)
                const float len = cv == DT_INF ? (float)DT_INF : (float)std::sqrt((double)(cv * cv + ch * ch + cd * cd));
                float dd = (float)((double)len / scale);
                if (dd < 0) dd = 0;
                td[z - zb][x - xb] = dd;
                tn[z - zb][x - xb] = nearest_occupied(g, z, y, x);
            }
            for (int z = zb; z < ze; z++) {
                float* drow = dist.data() + ((size_t)z * S + y) * S + xb;
                int32_t* nrow = nearest.data() + ((size_t)z * S + y) * S + xb;
                for (int x = 0; x < xe - xb; x++) { drow[x] = td[z - zb][x]; nrow[x] = tn[z - zb][x]; }
            }
        }
    };
    {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 1;
        if (nt > 16) nt = 16;
        if ((size_t)S * S * S < (size_t)1 << 18) nt = 1;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(finish_slab, (int)((size_t)S * t / nt), (int)((size_t)S * (t + 1) / nt));
        for (auto& t : th) t.join();
    }
    if (getenv("GOICP_TIMING")) {
        const auto tC = std::chrono::steady_clock::now();
        fprintf(stderr, "[goicp timing] DT propagate %.0f ms, distances+nearest map %.0f ms\n", std::chrono::duration<double, std::milli>(tB - tA).count(),
                std::chrono::duration<double, std::milli>(tC - tB).count());
    }
}


// ---- dt_mode 1: exact transform on the device ---------------------------------------------------------------------------------
void DT3D::ReleaseGpu() {
    if (gpu) { goicp_dtgpu_destroy(gpu); gpu = nullptr; }
}

int DT3D::BuildGpu(const double* _x, const double* _y, const double* _z, int num, int device) {
    ReleaseGpu();
    dist.clear(); nearest.clear(); cell_color.clear(); cell_ptr.clear();
    const int S = SIZE;
    if (num <= 0 || S < 1 || S > GOICP_MAX_GRID) return GOICP_ERR_ARG;
    // grid geometry: reference jly_3ddt.cpp:899-930 (same expressions as DT3D::Build above)
    xMin = xMax = _x[0]; yMin = yMax = _y[0]; zMin = zMax = _z[0];
    for (int i = 1; i < num; i++) {
        if (xMin > _x[i]) xMin = _x[i];
        if (xMax < _x[i]) xMax = _x[i];
        if (yMin > _y[i]) yMin = _y[i];
        if (yMax < _y[i]) yMax = _y[i];
        if (zMin > _z[i]) zMin = _z[i];
        if (zMax < _z[i]) zMax = _z[i];
    }
    const double xC = (xMin + xMax) / 2, yC = (yMin + yMax) / 2, zC = (zMin + zMax) / 2;
    xMin = xC - expandFactor * (xMax - xC); xMax = xC + expandFactor * (xMax - xC);
    yMin = yC - expandFactor * (yMax - yC); yMax = yC + expandFactor * (yMax - yC);
    zMin = zC - expandFactor * (zMax - zC); zMax = zC + expandFactor * (zMax - zC);
    double mx = xMax - xMin > yMax - yMin ? xMax - xMin : yMax - yMin;
    mx = mx > zMax - zMin ? mx : zMax - zMin;
    xMin = xC - mx / 2; xMax = xC + mx / 2;
    yMin = yC - mx / 2; yMax = yC + mx / 2;
    zMin = zC - mx / 2; zMax = zC + mx / 2;
    scale = S / mx;
    // seed cells ROUND((p - min) * scale) (jly_3ddt.cpp:976-995) and their point lists, occupied cells in ascending order
    std::vector<std::pair<int32_t, int32_t>> cp; cp.reserve((size_t)num);
    for (int i = 0; i < num; i++) {
        const int x = (int)((_x[i] - xMin) * scale + 0.5), y = (int)((_y[i] - yMin) * scale + 0.5), z = (int)((_z[i] - zMin) * scale + 0.5);
        if ((unsigned)x >= (unsigned)S || (unsigned)y >= (unsigned)S || (unsigned)z >= (unsigned)S) continue;
        cp.emplace_back((int32_t)(((size_t)z * S + y) * S + x), (int32_t)i);
    }
    std::stable_sort(cp.begin(), cp.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) { return a.first < b.first; });
    occ_cells.clear(); occ_ptr.assign(1, 0); cell_idx.clear(); occ_color.clear();
    for (size_t k = 0; k < cp.size(); k++) {
        if (occ_cells.empty() || occ_cells.back() != cp[k].first) { if (!occ_cells.empty()) occ_ptr.push_back((int32_t)cell_idx.size()); occ_cells.push_back(cp[k].first); }
        cell_idx.push_back(cp[k].second);
    }
    occ_ptr.push_back((int32_t)cell_idx.size());
    occ_color.assign(occ_cells.size(), -1);
    if (occ_cells.empty()) return GOICP_ERR_ARG;
    if (cell_idx.empty()) cell_idx.push_back(0);
    // value per squared cell distance: float((double)(float)sqrt((double)d2) / scale)   (jly_3ddt.cpp:28,79,1003)
    const int nt = 3 * (S - 1) * (S - 1) + 1;
    std::vector<float> table((size_t)nt);
    for (int n = 0; n < nt; n++) table[(size_t)n] = (float)((double)(float)std::sqrt((double)n) / scale);
    return goicp_dtgpu_build(device, S, occ_cells.data(), (int)occ_cells.size(), table.data(), nt, &gpu);
}

}  // namespace goicp
