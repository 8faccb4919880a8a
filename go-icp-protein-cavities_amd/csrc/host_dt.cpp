// host_dt.cpp -- distance transform + cell tables, host side (staged once per pair into HBM).
//
// Restates DT3D::Build (reference jly_3ddt.cpp:897-1137): 8SED-style vector propagation of (v,h,d)
// offsets in two x-sweeps with four z/y sub-scans each (DEuclidean, jly_3ddt.cpp:716-750), distance =
// sqrt(v^2+h^2+d^2)/scale, per-cell point lists and the nearest-occupied-cell guess from |v|,|h|,|d|.
// Where the reference returns an uninitialised candidate (jly_3ddt.cpp:70-73 and siblings) this builder
// returns the "infinite" cell, which is what the compiled reference produces in practice (bit-identical
// grids at 20^3..300^3, see tests/test_oracle_golden.py and tests/test_host_vs_oracle.py).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "goicp.hpp"

namespace goicp {
namespace {

constexpr int DT_INF = 32767;   // reference jly_3ddt.h:20

struct Cell { short v, h, d; float distance; };

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

struct Grid {
    int S;
    std::vector<Cell> a;
    Cell& at(int z, int y, int x) { return a[((size_t)z * S + y) * S + x]; }
};

inline float tap_length(short v, short h, short d) {
    // int sum of squares with 32-bit wrap, as the compiled reference computes it (jly_3ddt.cpp:79)
    unsigned s = (unsigned)((int)v * v) + (unsigned)((int)h * h) + (unsigned)((int)d * d);
    return (float)std::sqrt((double)(int)s);
}

template <int N>
inline void relax(Grid& g, const Tap (&taps)[N], int z, int y, int x) {
    const int S = g.S;
    Cell best{(short)DT_INF, (short)DT_INF, (short)DT_INF, (float)DT_INF};
    for (int k = 0; k < N; k++) {
        const Tap& t = taps[k];
        const int zz = z + t.gz, yy = y + t.gy, xx = x + t.gx;
        if ((unsigned)zz >= (unsigned)S || (unsigned)yy >= (unsigned)S || (unsigned)xx >= (unsigned)S) continue;
        const Cell& s = g.at(z + t.dz, y + t.dy, x + t.dx);
        Cell c;
        c.v = (short)(s.v + t.iv); c.h = (short)(s.h + t.ih); c.d = (short)(s.d + t.id);
        c.distance = tap_length(c.v, c.h, c.d);
        if (c.distance < best.distance) best = c;
    }
    g.at(z, y, x) = best;
}

void propagate_all(Grid& g) {   // DEuclidean, jly_3ddt.cpp:716-750
    const int S = g.S;
    for (int x = 0; x < S; x++) {
        for (int y = 0; y < S; y++) {
            for (int z = 0; z < S; z++) relax(g, kF1, z, y, x);
            for (int z = S - 1; z >= 0; z--) relax(g, kF2, z, y, x);
        }
        for (int y = S - 1; y >= 0; y--) {
            for (int z = S - 1; z >= 0; z--) relax(g, kF3, z, y, x);
            for (int z = 0; z < S; z++) relax(g, kF4, z, y, x);
        }
    }
    for (int x = S - 1; x >= 0; x--) {
        for (int y = S - 1; y >= 0; y--) {
            for (int z = S - 1; z >= 0; z--) relax(g, kB1, z, y, x);
            for (int z = 0; z < S; z++) relax(g, kF4, z, y, x);
        }
        for (int y = 0; y < S; y++) {
            for (int z = 0; z < S; z++) relax(g, kB3, z, y, x);
            for (int z = S - 1; z >= 0; z--) relax(g, kF2, z, y, x);
        }
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

int32_t nearest_occupied(Grid& g, int z, int y, int x) {
    const int S = g.S;
    const Cell& me = g.at(z, y, x);
    const int32_t self = (int32_t)(((size_t)z * S + y) * S + x);
    if (me.distance == 0) return self;
    const int xD = me.v, yD = me.h, zD = me.d;
    const signed char(*tab)[3]; int nt;
    if (xD == 0) { if (yD == 0) { tab = kSz; nt = 2; } else if (zD == 0) { tab = kSy; nt = 2; } else { tab = kSyz; nt = 4; } }
    else if (yD == 0) { if (zD == 0) { tab = kSx; nt = 2; } else { tab = kSxz; nt = 4; } }
    else if (zD == 0) { tab = kSxy; nt = 4; }
    else { tab = kS3; nt = 8; }
    for (int k = 0; k < nt; k++) {
        const int xx = x + tab[k][0] * xD, yy = y + tab[k][1] * yD, zz = z + tab[k][2] * zD;
        if ((unsigned)xx >= (unsigned)S || (unsigned)yy >= (unsigned)S || (unsigned)zz >= (unsigned)S) continue;
        if (g.at(zz, yy, xx).distance == 0) return (int32_t)(((size_t)zz * S + yy) * S + xx);
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

    Grid g; g.S = S;
    g.a.assign(ncell, Cell{(short)DT_INF, (short)DT_INF, (short)DT_INF, (float)DT_INF});
    cell_color.assign(ncell, -2);
    cell_ptr.assign(ncell + 1, 0);
    cell_idx.assign((size_t)(num > 0 ? num : 1), 0);
    std::vector<int64_t> cell_of((size_t)num, -1);
    for (int i = 0; i < num; i++) {
        const int x = (int)((_x[i] - xMin) * scale + 0.5), y = (int)((_y[i] - yMin) * scale + 0.5), z = (int)((_z[i] - zMin) * scale + 0.5);
        if ((unsigned)x >= (unsigned)S || (unsigned)y >= (unsigned)S || (unsigned)z >= (unsigned)S) continue;   // (the reference would index out of bounds)
        const size_t k = ((size_t)z * S + y) * S + x;
        cell_of[i] = (int64_t)k; cell_color[k] = -1; cell_ptr[k + 1]++;
        g.a[k] = Cell{0, 0, 0, 0.f};
    }
    for (size_t k = 0; k < ncell; k++) cell_ptr[k + 1] += cell_ptr[k];
    {
        std::vector<int32_t> fill(ncell, 0);
        for (int i = 0; i < num; i++) if (cell_of[i] >= 0) { const size_t k = (size_t)cell_of[i]; cell_idx[cell_ptr[k] + fill[k]++] = i; }
    }
    propagate_all(g);
    dist.resize(ncell); nearest.resize(ncell);
    for (size_t k = 0; k < ncell; k++) {
        float d = (float)((double)g.a[k].distance / scale);     // jly_3ddt.cpp:1003
        if (d < 0) d = 0;
        g.a[k].distance = d;
        dist[k] = d;
    }
    for (int z = 0; z < S; z++) for (int y = 0; y < S; y++) for (int x = 0; x < S; x++)
        nearest[((size_t)z * S + y) * S + x] = nearest_occupied(g, z, y, x);
}

}  // namespace goicp
