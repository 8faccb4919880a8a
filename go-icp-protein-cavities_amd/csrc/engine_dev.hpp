// engine_dev.hpp -- device side shared by the kernel translation units of libgoicp_hip.so (k_bnb.hip, k_fr.hip, k_eval.hip, engine.hip):
// tables, the DT lookup, the tile sweeps (sweep_tile / sweep_stream), the reference-order selection, the corner-lattice terms and the
// replayed priority queue. Everything here is __device__ inline code or plain structs; the kernels and their launchers live in the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include <mutex>

#include "goicp_engine.h"
#include "pool.hpp"
#include "kabsch.hpp"

#define TC 8            // cubes per workgroup tile (the 8 children of one queue pop)
#define CAND_CAP 128     // threshold-bin members refined inside one wave; larger bins take the generic byte passes
#define NLMAX 64        // lattice points per tile: 27 (shared child corners) or 8*TC (free cubes)
#define SQRT3_D 1.732050808   // jly_goicp.h:45
// Diagnosis bits of GOICP_DBG (cycle stamps: 32, 64; ablations that make results WRONG: 1 no selection, 2 gathers forced into a 4 KB window,
// 128 no fused pop) exist only in a `make PROBES=1` build; the shipped library compiles them out.
#ifdef GOICP_PROBES
#define DBG_ON(T, bit) ((((T).dbg) & (bit)) != 0)
#else
#define DBG_ON(T, bit) false
#endif

struct DevTables {
    int S;
    double xMin, yMin, zMin, scale;
    const float* dist;
    // 2-byte coded copy of the grid (see k_build_codes): code = squared cell distance d2 when dist == lut[d2] bit for bit and d2 < 65535,
    // else 0xFFFF (the value is then read from `dist`). Twice the cells per cache line of the gather stream; null = not built.
    const uint16_t* code; const float* lut;
    const uint16_t* cmask;
    const int* occ;          // [S^3] compact id of the nearest occupied cell, -1 if none
    const int* occ_ptr;      // [nOcc+1]
    const int* occ_pts;      // model ids
    const float* mfpfh;      // [Nm*41]
    const float* dfpfh;      // [Nd*41]
    const float* fpE;        // [Nd*nOcc] min over the model points of occupied cell o of the L1 distance to data point i's descriptor
    int nOcc;
    const float* dxyz;       // [Nd*3]
    const uint8_t* dcls;     // [Nd] colour class of each data point
    const float* weights;    // [Nd]
    const float* maxRotDis;  // [20*Nd]
    const int* perm; const float* maxRotDisP;   // metric kernel (order-free sums): slot clouds hold point perm[i] at position i (far points first, so that the
                                                // rare out-of-grid lanes share wave-iterations); maxRotDisP = the radius rows in that order. Null = identity.
    int Nd, Nm, inlierNum, doTrim, norm, b0, b1;
    float reg, regFPFH;
    int use_compat, use_fpfh, exact;
    int exact_big;           // parity_mode 2, 1025..5120 points, geometric error: reference-order sums / intro_select on residual rows in HBM (xrows)
    int exact_stream;        // parity_mode 2 on a streamed cloud: the same on [2 * TC][NdP] rows per resident workgroup (srows)
    float* xrows; int xslots; // [xslots][2 * TC * NdP] floats: addend / residual rows + selection scratch of one workgroup each
    float4* srot;            // streaming path (Nd > 8192): per-workgroup rotated cloud [NdP]
    float* srows;            // streaming path, trimmed runs: per-workgroup residual rows [TC][NdP]
    int NdP;                 // row pitch of the two above
    int regsel;              // reference-order selection (trimmed parity runs): partition rounds on ranges of <= 64 elements run in registers, one element per lane
    int dbg;                 // GOICP_DBG ablation bits (diagnosis only; results are wrong when set)
    unsigned long long* dbg_times;   // [8] per-phase cycle sums of job 0, lane 0 (GOICP_DBG & 32)
};

struct __attribute__((aligned(8))) HeapNode { float x, y, z, w, ub, lb; };

// order-free engines (geometric error above 1024 points, not parity_mode 2) keep their clouds by decreasing norm: position i holds point T.perm[i]
__device__ __forceinline__ int src_point(const DevTables& T, int i) { return T.perm ? T.perm[i] : i; }
__device__ __forceinline__ const float* rho_rows(const DevTables& T) { return T.perm ? T.maxRotDisP : T.maxRotDis; }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_of(float v, double mn, double scale) {
    // ROUND((x - xMin) * scale), reference jly_3ddt.cpp:30,1142-1144: double math, truncation toward zero
    return (int)(((double)v - mn) * scale + 0.5);
}

// DT3D::Distance (reference jly_3ddt.cpp:1139-1191) split in two so that a lane can put many gathers in flight
// before it consumes the first one: dt_index() does the (double) cell arithmetic and yields the clamped cell plus
// the out-of-grid extension length (0 inside the grid); dt_finish() applies the extension to the fetched value.
__device__ __forceinline__ unsigned dt_index(const DevTables& T, float X, float Y, float Z, float& ext) {
    const int S = T.S;
    int x = cell_of(X, T.xMin, T.scale), y = cell_of(Y, T.yMin, T.scale), z = cell_of(Z, T.zMin, T.scale);
    const int ax = x < 0 ? x : (x >= S ? x - S + 1 : 0);
    const int ay = y < 0 ? y : (y >= S ? y - S + 1 : 0);
    const int az = z < 0 ? z : (z >= S ? z - S + 1 : 0);
    x = min(max(x, 0), S - 1); y = min(max(y, 0), S - 1); z = min(max(z, 0), S - 1);
    ext = 0.f;
    if ((ax | ay | az) != 0) { const float a = (float)ax, b = (float)ay, c = (float)az; ext = sqrtf(a * a + b * b + c * c); }
    return (unsigned)((z * S + y) * S + x);
}
// one axis of dt_index(): unclamped cell coordinate; clamping is a single v_med3_i32 and the out-of-grid offset is only
// formed on the (rare) lanes that need it
__device__ __forceinline__ int clamp_cell(int x, int S) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(S - 1));
    return r;
}
__device__ __forceinline__ int grid_off(int x, int S) { return x < 0 ? x : (x >= S ? x - S + 1 : 0); }
// extension length from UNCLAMPED cell coordinates (0 when the cell is inside the grid)
__device__ __forceinline__ float dt_ext_raw(int x, int y, int z, int S) {
    float ext = 0.f;
    if (((unsigned)x >= (unsigned)S) | ((unsigned)y >= (unsigned)S) | ((unsigned)z >= (unsigned)S)) {
        const float a = (float)grid_off(x, S), b = (float)grid_off(y, S), c = (float)grid_off(z, S);
        ext = sqrtf(a * a + b * b + c * c);
    }
    return ext;
}
__device__ __forceinline__ float dt_finish(const DevTables& T, float v, float ext) {
    if (ext > 0.f) v = (float)((double)ext / T.scale + (double)v);      // float sqrt, double divide and add, rounded to float
    return v;
}
__device__ __forceinline__ float dt_lookup(const DevTables& T, float X, float Y, float Z) {
    float ext;
    const unsigned idx = dt_index(T, X, Y, Z, ext);
    return dt_finish(T, T.dist[idx], ext);
}

#define LUT_LDS 8192      // leading LUT entries a sweep kernel keeps in LDS (squared cell distances below 8192: within 90 cells of the model)
// value of cell idx through the coded grid: code -> LUT (LDS copy for small codes, global beyond), escape -> the float grid
__device__ __forceinline__ float dt_decode(const DevTables& T, unsigned idx, unsigned c, const float* lut_lds) {
    if (c < (unsigned)LUT_LDS && lut_lds) {
        const __attribute__((address_space(3))) float* q = (const __attribute__((address_space(3))) float*)lut_lds;
        return q[c];
    }
    return c != 0xFFFFu ? T.lut[c] : T.dist[idx];
}

__device__ __forceinline__ size_t clamped_cell(const DevTables& T, float X, float Y, float Z) {
    // checkCompatibility / computeFPFHDifference prologue (reference jly_goicp.cpp:974-985, 1642-1651)
    const int S = T.S;
    int x = cell_of(X, T.xMin, T.scale), y = cell_of(Y, T.yMin, T.scale), z = cell_of(Z, T.zMin, T.scale);
    x = x < 0 ? 0 : (x >= S ? S - 1 : x);
    y = y < 0 ? 0 : (y >= S ? S - 1 : y);
    z = z < 0 ? 0 : (z >= S ? S - 1 : z);
    return ((size_t)z * S + y) * S + x;
}

// min over the model points of the nearest occupied cell of the L1 distance between c-FPFH rows
// (computeFPFHDifference(false,...), reference jly_goicp.cpp:1642-1681). The value depends only on (data point,
// occupied cell), not on the pose, so it is tabulated once per engine (k_fpfh_table) and the hot path does one gather.
__device__ __forceinline__ float fpfh_min(const DevTables& T, int i, float X, float Y, float Z) {
    const int o = T.occ[clamped_cell(T, X, Y, Z)];
    return o < 0 ? 1000000000.f : T.fpE[(size_t)i * T.nOcc + o];
}


__device__ __forceinline__ double cube_sum(double v) {       // over the 8 lanes of a wave that hold the same (lane & 7)
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Histogram increment for a whole wave. Residual sets are often dominated by one value (every residual that the
// rotation-uncertainty radius clamps to 0 is bit-identical), and 64 lanes adding to one LDS word serialise; so the
// lanes that share the first taker's key are counted with one ballot and added once, the others add individually.
__device__ __forceinline__ void hist_add(unsigned* row, unsigned key, bool take, int ln) {
    const unsigned long long tk = __ballot(take);
    if (!tk) return;
    const int leader = __ffsll((long long)tk) - 1;
    const unsigned k0 = (unsigned)__shfl((int)key, leader, 64);
    const unsigned long long same = __ballot(take && key == k0);
    if (ln == leader) atomicAdd(&row[k0], (unsigned)__popcll(same));
    if (take && key != k0) atomicAdd(&row[key], 1u);
}

// ---------------------------------------------------------------------------------------------
// intro_select of the reference (jly_sorting.hpp:34-313) with its exact swap sequence, run by one wave on a residual row in
// LDS. Parity mode uses it for trimmed runs of small clouds: the permutation it leaves behind fixes the order of the
// float sums that follow (jly_goicp.cpp:389-415), so the bounds come out bit-identical to the reference's.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sel_pivot_ends(const float* d, int s, int e) {
    const int m = (s + e) / 2;
    if (d[s] < d[e]) return d[m] < d[s] ? s : (d[m] < d[e] ? m : e);
    return d[m] < d[e] ? e : (d[m] < d[s] ? m : s);
}
__device__ inline int sel_med3(const float* d, int s) {
    if (d[s] < d[s + 2]) return d[s + 1] < d[s] ? s : (d[s + 1] < d[s + 2] ? s + 1 : s + 2);
    return d[1] < d[2] ? s + 2 : (d[1] < d[0] ? s + 1 : s);            // absolute indices 0..2, as in jly_sorting.hpp:74-80
}
__device__ inline int sel_med5(float* d, int s) {
    float* q = d + s; float t;
    if (q[0] > q[1]) { t = q[0]; q[0] = q[1]; q[1] = t; }
    if (q[2] > q[3]) { t = q[2]; q[2] = q[3]; q[3] = t; }
    if (q[0] < q[2]) { t = q[4]; q[4] = q[0]; if (t < q[1]) q[0] = t; else { q[0] = q[1]; q[1] = t; } }
    else { t = q[4]; q[4] = q[2]; if (t < q[3]) q[2] = t; else { q[2] = q[3]; q[3] = t; } }
    if (q[0] < q[2]) return q[1] < q[2] ? s + 1 : s + 2;
    return q[0] < q[3] ? s : s + 3;
}
// Wave-parallel form of the same selection: one wave per residual row. The reference's partition loop pairs the i-th element
// > pivot from the left with the i-th element < pivot from the right and swaps them while they have not crossed; that final
// arrangement (and the pivot's resting place) is a pure function of the row, so it is computed with ballots / prefix counts
// instead of a cell-by-cell scan, and the result is bit-identical to the sequential routine (tests/csrc/partition_check.c). `idx` is scratch of n ints.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the same hand-off between the lanes of a wave when the data may live in HBM (rows of parity_mode 2): workgroup scope makes the compiler wait for the
// outstanding vector-memory operations as well (a CU's waves share its L1, so nothing has to be written back or invalidated)
__device__ __forceinline__ void wave_sync_mem() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// sel_median_of_medians for a whole wave. The group medians of a pass are independent of each other (a group is only modified by its own
// sel_med5 and, later, by front swaps of groups 5 times further on), so every lane takes groups of its own: sel_med5 in registers with the
// reference's in-place moves written back. The front swaps "d[s+i] <-> d[median of group i]" keep the sequential result when they are applied
// level by level: swap i can only meet swap floor(i/5) (whose group holds the slot s+i), so swaps [0,1), [1,5), [5,25), [25,125), ... are
// independent inside a level. The last (short) group and the final choice stay on lane 0 (sel_med3 reads absolute indices, jly_sorting.hpp:74-80).
// idx: scratch of at least (e - s + 1) / 5 + 1 ints. Returns the pivot's index (uniform).
template <bool MEM>
__device__ __forceinline__ int wave_median_of_medians(float* d, int* idx, int s, int e, int lane) {
    for (;;) {
        const int len = e - s + 1, groups = len / 5 + (len % 5 != 0);
        const int nfull = groups - 1;                                  // groups that are certainly complete (the loop of the reference)
        for (int it = lane; it < nfull; it += 64) {
            float* q = d + s + 5 * it;
            float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], t;
            if (q0 > q1) { t = q0; q0 = q1; q1 = t; }
            if (q2 > q3) { t = q2; q2 = q3; q3 = t; }
            int m;
            if (q0 < q2) { t = q4; q4 = q0; if (t < q1) q0 = t; else { q0 = q1; q1 = t; } }
            else { t = q4; q4 = q2; if (t < q3) q2 = t; else { q2 = q3; q3 = t; } }
            if (q0 < q2) m = q1 < q2 ? 1 : 2; else m = q0 < q3 ? 0 : 3;
            q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3; q[4] = q4;
            idx[it] = s + 5 * it + m;
        }
        (MEM ? wave_sync_mem() : wave_sync_lds());
        for (int lo = 0, hi = 1; lo < nfull; lo = hi, hi *= 5) {
            for (int i = lo + lane; i < min(hi, nfull); i += 64) { const int mi = idx[i]; const float t = d[s + i]; d[s + i] = d[mi]; d[mi] = t; }
            (MEM ? wave_sync_mem() : wave_sync_lds());
        }
        int ret = 0;
        if (lane == 0) {
            const int g = s + 5 * nfull, rest = e - g + 1;
            const int mi = (rest == 3 || rest == 4) ? sel_med3(d, g) : (rest == 5 ? sel_med5(d, g) : g);
            const float t = d[s + nfull]; d[s + nfull] = d[mi]; d[mi] = t;
            if (groups <= 5) ret = (groups == 3 || groups == 4) ? sel_med3(d, s) : (groups == 5 ? sel_med5(d, s) : s);
        }
        (MEM ? wave_sync_mem() : wave_sync_lds());
        if (groups > 5) { e = s + groups - 1; continue; }
        return __shfl(ret, 0, 64);
    }
}
template <bool MEM = false>
__device__ __forceinline__ void wave_intro_select(float* d, int* idx, int n, int k, int lane, unsigned long long* st = nullptr /* diagnosis: [0] rows [1] rounds [2] median-of-medians rounds [3] cycles in them [4] cycles in the final insertion sort */,
                                                  bool regsel = false /* ranges of <= 64 elements finish in registers, one element per lane (tests/csrc/partition_check.c, reg()); default on (T.regsel), GOICP_REGSEL=0 switches it off */) {
    int s = 0, e = n - 1;
    int prev = e - s + 1, tries = 0; bool quick = true;
    unsigned long long n_rounds = 0, n_mom = 0, c_mom = 0, c_ins = 0, c_a = 0, c_b = 0, c_c = 0, c_d = 0, tq = 0;
    for (;;) {
        if (s >= e) break;
        n_rounds++;
        if (e - s <= 5) {
            const unsigned long long t0 = st ? clock64() : 0ull;
            // (the reference's insertion sort of the last <= 6 elements: the sorted arrangement is a function of the values alone, so the
            // elements are fetched together, sorted in registers and written back by lane 0 instead of ~15 dependent round trips)
            {
                const int cnt = e - s + 1;
                float w[6];
#pragma unroll
                for (int q = 0; q < 6; q++) w[q] = q < cnt ? d[s + q] : __builtin_inff();
#pragma unroll
                for (int i = 1; i < 6; i++) {
#pragma unroll
                    for (int j = i; j > 0; j--) { const float a = w[j - 1], b = w[j]; const bool sw = a > b; w[j - 1] = sw ? b : a; w[j] = sw ? a : b; }
                }
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 6; q++) if (q < cnt) d[s + q] = w[q];
                }
            }
            if (st) c_ins += clock64() - t0;
            break;
        }
        if (regsel && quick && e - s + 1 <= 64) {
            // ---- register regime: lane i holds d[s0 + i]; the same rounds with ballots, rank scatters and two gathers instead of LDS round trips
            const int s0 = s, len0 = e - s + 1, kr = k - s0;
            float w = lane < len0 ? d[s0 + lane] : 0.f;
            int rs = 0, re = len0 - 1;
            bool done = false;
            const unsigned long long lo_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane)), hi_mask = lane == 63 ? 0ull : (~0ull << (lane + 1));
            for (;;) {
                if (rs >= re) { done = true; break; }
                n_rounds++;
                if (re - rs <= 5) {
                    // the reference's insertion sort of the last <= 6 elements = stable ascending order: rank by comparisons, one scatter
                    const bool in = lane >= rs && lane <= re;
                    int rank = 0;
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        const int pq = min(rs + q, re);
                        const float wq = __shfl(w, pq, 64);
                        if (rs + q <= re) rank += (wq < w || (!(w < wq) && pq < lane)) ? 1 : 0;      // (wq == w in the comparison's own sense: neither is below the other)
                    }
                    const int dst = in ? rs + rank : lane;
                    w = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(w)));
                    done = true; break;
                }
                if (tries++ == 5) { const int len = re - rs + 1; if (len * 2 > prev) quick = false; prev = len; tries = 0; }
                if (!quick) break;                                     // no progress: the median-of-medians rounds run on the row in memory
                const int m = (rs + re) / 2;
                const float ds = __shfl(w, rs, 64), dm = __shfl(w, m, 64), de = __shfl(w, re, 64);
                int pi;
                if (ds < de) pi = dm < ds ? rs : (dm < de ? m : re);
                else pi = dm < de ? re : (dm < ds ? m : rs);
                const float piv = pi == rs ? ds : (pi == m ? dm : de);
                if (pi != rs) { if (lane == pi) w = ds; if (lane == rs) w = piv; }
                const bool in = lane > rs && lane <= re;
                const unsigned long long g = __ballot(in && w > piv), l = __ballot(in && w < piv);
                const int nG = __popcll(g);
                const bool isg = (g >> lane) & 1ull, isl = (l >> lane) & 1ull;
                const int g_below = __popcll(g & lo_mask), l_above = __popcll(l & hi_mask);
                const bool swp = isg ? l_above > g_below : (isl ? g_below > l_above : false);
                const int t = __popcll(__ballot(swp && isg));
                // pair i = (i-th '>' from the left, i-th '<' from the right): both lane ids to lane i (the other lanes fill the permutation up)
                const int dstA = isg ? g_below : nG + __popcll(~g & lo_mask);
                const int dstB = isl ? l_above : __popcll(l) + __popcll(~l & hi_mask);
                const int A = __builtin_amdgcn_ds_permute(dstA << 2, lane), B = __builtin_amdgcn_ds_permute(dstB << 2, lane);
                const int P = A | (B << 8);
                const int q = __builtin_amdgcn_ds_bpermute((isg ? g_below : l_above) << 2, P);
                const int partner = isg ? (q >> 8) & 255 : q & 255;
                const float nv = __int_as_float(__builtin_amdgcn_ds_bpermute((swp ? partner : lane) << 2, __float_as_int(w)));
                if (swp) w = nv;
                const int iA = __shfl(B, max(t - 1, 0), 64), iB = __shfl(A, min(t, 63), 64);
                const float dend = __shfl(w, re, 64);
                const int hi0 = t > 0 ? iA : re;
                int lo1 = hi0; bool lo_is_greater;
                if (t < nG && iB < hi0) { lo1 = iB; lo_is_greater = true; }
                else lo_is_greater = t > 0 ? true : !(dend < piv);
                const int sp = lo_is_greater ? lo1 - 1 : lo1;
                const float vsp = __shfl(w, sp, 64);
                if (lane == rs) w = vsp;
                if (lane == sp) w = piv;
                if (sp < kr) rs = sp + 1; else if (sp > kr) re = sp - 1; else { done = true; break; }
            }
            if (lane < len0) d[s0 + lane] = w;
            (MEM ? wave_sync_mem() : wave_sync_lds());
            if (done) break;
            s = s0 + rs; e = s0 + re;                                  // (quick is false from here on: the regime is not entered again)
        }
        if (quick && tries++ == 5) { const int len = e - s + 1; if (len * 2 > prev) quick = false; prev = len; tries = 0; }
        int pi; float piv, dfront;
        if (st) tq = clock64();
        if (quick) {
            // sel_pivot_ends with its three (uniform) reads kept: the pivot's value is one of them
            const int m = (s + e) / 2;
            const float ds = d[s], dm = d[m], de = d[e];
            if (ds < de) pi = dm < ds ? s : (dm < de ? m : e);
            else pi = dm < de ? e : (dm < ds ? m : s);
            piv = pi == s ? ds : (pi == m ? dm : de);
            dfront = ds;
        } else {
            int tmp = 0;
            const unsigned long long t0 = st ? clock64() : 0ull;
            tmp = wave_median_of_medians<MEM>(d, idx, s, e, lane);      // fallback of a selection that does not make progress: the reference's in-place group medians
            if (st) { c_mom += clock64() - t0; n_mom++; }
            pi = tmp;
            piv = d[pi]; dfront = d[s];
        }
        // pivot to the front. No hand-off is needed around it: the three reads above have returned (the pivot's index is computed from
        // them), the slices below never read d[s], and the one slot they could see too early (d[pi]) is patched in registers.
        if (!quick) (MEM ? wave_sync_mem() : wave_sync_lds());
        if (pi != s && lane == 0) { d[pi] = dfront; d[s] = piv; }
        if (st) { const unsigned long long now = clock64(); c_a += now - tq; tq = now; }
        // The reference's partition loop pairs the i-th element > pivot from the left with the i-th element < pivot from the right and swaps
        // them while they have not crossed. Ranks by position: the elements of (s, e] are taken 64 at a time (lane = position within the
        // slice), so a rank is "set bits of the slice's ballot below my lane + the totals of the slices before" -- no scan, no LDS.
        const int L = e - s;                                           // elements besides the pivot
        int nG = 0, nLt = 0;
        float v4[4]; unsigned long long g4[4], l4[4];                  // the first four slices stay in registers (every row of <= 257 elements)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            g4[q] = 0ull; l4[q] = 0ull;
            if (q > 0 && 64 * q >= L) continue;                        // (uniform: short ranges take one slice)
            const int p = s + 1 + lane + 64 * q;
            const bool in = p <= e;
            v4[q] = in ? (p == pi ? dfront : d[p]) : piv;
            g4[q] = __ballot(in && v4[q] > piv); l4[q] = __ballot(in && v4[q] < piv);
            nG += __popcll(g4[q]); nLt += __popcll(l4[q]);
        }
        for (int base = 256; base < L; base += 256) {                  // longer rows: count the rest, four slices (reads) at a time
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int p = s + 1 + lane + base + 64 * q; v[q] = p <= e ? (p == pi ? dfront : d[p]) : piv; }
#pragma unroll
            for (int q = 0; q < 4; q++) { nG += __popcll(__ballot(v[q] > piv)); nLt += __popcll(__ballot(v[q] < piv)); }
        }
        // idx[i] = position of the i-th '>' from the left; idx[L-1-i] = position of the i-th '<' from the right (nG + nLt <= L)
        {
            int bg = 0, bl = 0;                                        // '>' / '<' met in the slices before
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q > 0 && 64 * q >= L) continue;
                const int p = s + 1 + lane + 64 * q;
                const int below_g = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(g4[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)g4[q], 0u));
                const int below_l = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(l4[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)l4[q], 0u));
                if ((g4[q] >> lane) & 1ull) idx[bg + below_g] = p;
                if ((l4[q] >> lane) & 1ull) idx[L - 1 - (nLt - 1 - (bl + below_l))] = p;
                bg += __popcll(g4[q]); bl += __popcll(l4[q]);
            }
            for (int base = 256; base < L; base += 256) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const int p = s + 1 + lane + base + 64 * q; v[q] = p <= e ? (p == pi ? dfront : d[p]) : piv; }      // (out of range: equal to the pivot, in neither set)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int p = s + 1 + lane + base + 64 * q;
                    const unsigned long long gm = __ballot(v[q] > piv), lm = __ballot(v[q] < piv);
                    const int below_g = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(gm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)gm, 0u));
                    const int below_l = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
                    if ((gm >> lane) & 1ull) idx[bg + below_g] = p;
                    if ((lm >> lane) & 1ull) idx[L - 1 - (nLt - 1 - (bl + below_l))] = p;
                    bg += __popcll(gm); bl += __popcll(lm);
                }
            }
        }
        (MEM ? wave_sync_mem() : wave_sync_lds());
        if (st) { const unsigned long long now = clock64(); c_b += now - tq; tq = now; }
        // pair i is swapped iff its two members have not crossed; pairs are disjoint, so one lane per pair is race-free
        int t = 0;
        for (int base = 0, m = min(nG, nLt); base < m; base += 64) {
            const int i = base + lane;
            const bool act = i < m;
            const int a_ = act ? idx[i] : 0, b_ = act ? idx[L - 1 - i] : 0;
            const bool sw = act && a_ < b_;
            t += __popcll(__ballot(sw));
            if (sw) { const float va = d[a_]; d[a_] = d[b_]; d[b_] = va; }
        }
        (MEM ? wave_sync_mem() : wave_sync_lds());
        if (st) { const unsigned long long now = clock64(); c_c += now - tq; tq = now; }
        // where the reference's scan pointers end up, and where the pivot comes to rest (jly_sorting.hpp:282-303); the three reads go out together
        const int iA = idx[max(L - t, 0)], iB = idx[min(t, L - 1)];
        const float dend = d[e];
        const int hi0 = t > 0 ? iA : e;
        int lo1 = hi0; bool lo_is_greater;
        if (t < nG && iB < hi0) { lo1 = iB; lo_is_greater = true; }
        else lo_is_greater = t > 0 ? true : !(dend < piv);            // t == 0: d[e] is untouched; '<' means the pivot swaps with it
        const int sp = lo_is_greater ? lo1 - 1 : lo1;
        if (lane == 0) { d[s] = d[sp]; d[sp] = piv; }
        (MEM ? wave_sync_mem() : wave_sync_lds());
        if (st) { const unsigned long long now = clock64(); c_d += now - tq; tq = now; }
        if (sp < k) s = sp + 1; else if (sp > k) e = sp - 1; else break;
    }
    (MEM ? wave_sync_mem() : wave_sync_lds());
    if (st && lane == 0) { atomicAdd(&st[0], 1ull); atomicAdd(&st[1], n_rounds); atomicAdd(&st[2], n_mom); atomicAdd(&st[3], c_mom); atomicAdd(&st[4], c_ins); atomicAdd(&st[5], c_a); atomicAdd(&st[6], c_b); atomicAdd(&st[7], c_c); }
}

template <int BLOCK>
struct TileShm {
    float cube[TC][4];        // x, y, z, w
    float cen[TC][3];         // centre translation
    float mtd[TC];            // maxTransDis
    unsigned hist[TC][1024];  // radix-select histograms (first pass 1024 bins, later passes 256)
    unsigned prefix[TC];
    int rem[TC];
    unsigned maxb[TC];        // bit pattern of the largest residual per cube
    unsigned wmax[BLOCK / 64][TC];   // per-wave maxima (plain stores: no initialisation to order against)
    int fallback;
    int zc[TC];               // residuals that are exactly 0 (clamped), counted with ballots instead of LDS adds
    int ncand[TC];            // members of the threshold bin collected for the in-wave refinement
    unsigned cand[TC][CAND_CAP];
    double red[BLOCK / 64][2 * TC];
    float ub[TC], lb[TC];
    float ub2[TC], lb2[TC];   // fused form: bounds of the first of the two calls
    float lat[NLMAX][3];
    int cnt[NLMAX];
    float fpv[NLMAX];
    int ctl[4];               // [0] done flag, [1] status
    unsigned himask;          // fixed-window selection (XP): bit j = cube j has residuals above the window
};

// Evaluate the TC cubes whose descriptors sit in sh.cube[] against the PPT points each lane holds in
// registers. Results: sh.ub[], sh.lb[] (geometric part only; the penalty part is added by the caller).
// Restates reference jly_goicp.cpp:343-415.
// NT = cubes of the tile (TC everywhere today; 4-cube tiles with two 512-lane workgroups per CU measured no faster)
// JIT = true: the lane's points are not held in registers; point m is fetched from `jit_rot` / `jit_rho` just before its gathers
// (one point ahead), which frees the registers a second workgroup per CU needs (512 lanes x 10 points, k_eval_cubes only).
// DUAL = true (just-in-time, non-parity form): the tile is evaluated for TWO calls that share rotation and cubes -- the ub-call
// (no rotation uncertainty) and an lb-call (radius row dual_rho) -- from ONE pass of gathers: results of the first in sh.ub2 / sh.lb2,
// of the second in sh.ub / sh.lb. OuterBnB's two InnerBnB calls per child rotation (jly_goicp.cpp:768,861) read the same DT cells.
// ---- fixed-window selection of the metric kernel (sweep_tile<..., XP = true>) ----
// The inlier threshold is found on a histogram of the residuals' bit patterns. With a FIXED window -- key = bits >> 17 (sign, exponent, 6 mantissa
// bits) relative to 2^-12, i.e. 1024 bins over the 16 octaves [2^-12, 2^4) -- no pass for the per-cube maximum is needed, and the histogram of
// the first selection is accumulated INSIDE the gather loop (LDS adds beside the outstanding gathers, nothing waits for them). Counters are 16 bits
// (a cube has at most 8192 residuals), two bins per LDS word: the 8 histograms take the lower half of sh.hist, the transposition scratch of
// the 8 waves the upper half. A threshold that falls outside the window or into a crowded bin takes the generic byte passes, as before.
#define FW_SHIFT 17
#define FW_KBASE ((int)(0x39800000u >> FW_SHIFT))        // key of 2^-12
#define FW_HP_WORDS 512                                   // packed words per cube histogram
template <int BLOCK, bool H32>
__device__ __forceinline__ void fw_hist_reset(TileShm<BLOCK>& sh) {   // every lane of the workgroup; a workgroup barrier must follow before the first add
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4_t lds_u128;
    lds_u128* hp = (lds_u128*)&sh.hist[0][0];
    const u32x4_t z = {0u, 0u, 0u, 0u};
    for (int k = threadIdx.x; k < TC * (H32 ? 1024 : FW_HP_WORDS) / 4; k += BLOCK) hp[k] = z;   // (32-bit bins: the whole of sh.hist, no transposition scratch in that variant)
    if (threadIdx.x < TC) { sh.prefix[threadIdx.x] = 0u; sh.zc[threadIdx.x] = 0; sh.ncand[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { sh.fallback = 0; sh.himask = 0u; }
}

// XP = true (NT = 8, at most 8 waves, coded grid): the gathers of a point are issued TRANSPOSED. With lane = point every gather instruction
// carries 64 unrelated cells (64 line look-ups, and a point's 8 look-ups -- 1-3 cache lines of one grid row for a tile of x-neighbours -- go
// out as 8 separate requests of which the L1 merges only some: 4.4 L2 requests per point and tile measured for 2.1 lines touched). Here the 8
// lanes of a group swap their cell indices through LDS (the wave's own 4 KB row of sh.hist, which finish() clears afterwards) so that
// instruction k carries the 8 cells of point k of every group: 8 lanes share 1-3 lines, one request each. The codes travel back the same way.
// Values, arithmetic and the residual registers are untouched (lane = point everywhere else).
template <int BLOCK, int PPT, bool OCT, int NT, bool JIT = false, bool DUAL = false, bool XBIG = false /* JIT kernels: the parity_mode 2 instantiation (rows in HBM) */, bool XP = false,
          int NORM = 0 /* XP path: 2 or 1 = the norm is known at compile time (no per-value selects), 0 = read T.norm */>
__device__ __forceinline__ void sweep_tile(const DevTables& T, TileShm<BLOCK>& sh, const float (&px)[PPT], const float (&py)[PPT],
                                           const float (&pz)[PPT], const float (&pw)[PPT], const float (&prho)[PPT], bool use_rho,
                                           float* rbuf /* LDS, exact mode only */, bool row_mode = false /* !OCT: the cubes share y and z */,
                                           const float4* __restrict__ jit_rot = nullptr, const float* __restrict__ jit_rho = nullptr,
                                           const float* lut_lds = nullptr /* LDS copy of T.lut[0, LUT_LDS) or null */,
                                           const float* __restrict__ dual_rho = nullptr) {
    const int tid = threadIdx.x;
    const int Nd = T.Nd;
    // the two forms of the fast path (measured, 149 760 cube-evals per launch): a tile evaluated for ONE call spends most of its time gathering and gains
    // from the transposed gathers (1.82 vs 1.93 ms), whose scratch takes half of sh.hist and leaves room for 16-bit bins only; a fused (no radius,
    // radius) tile is selection-heavy and gains more from plain 32-bit bins -- a shift, a subtract, a clamp and one LDS add per value (1.21 vs 1.35 ms)
    constexpr bool XPOSE = XP && !DUAL, H32 = XP && DUAL;
    float r[NT][PPT];
    // gathers: the NT lookups of one point are issued back to back (no consumer in between), so every lane keeps NT loads
    // in flight. A lookup that leaves the grid only sets a flag bit; its extension length (jly_3ddt.cpp:1150-1187) is
    // formed on the rare lanes that need it, after the fetch.
    // Lanes past Nd carry a zero point with zero weight (their residuals are never counted).
    const int S = T.S;
    auto cells = [&](float X, float Y, float Z, unsigned (&idx)[NT]) -> unsigned {
        unsigned oob = 0u;
        if (OCT) {
            // the 8 cubes are the 2x2x2 children of one node: child j sits at (j&1, j>>1&1, j>>2&1), so only two distinct
            // centre coordinates exist per axis and the (double) cell arithmetic is done 6 times instead of 24
            int rx[2], ry[2], rz[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                rx[q] = cell_of(X + sh.cen[q][0], T.xMin, T.scale);
                ry[q] = cell_of(Y + sh.cen[2 * q][1], T.yMin, T.scale);
                rz[q] = cell_of(Z + sh.cen[4 * q][2], T.zMin, T.scale);
            }
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int x = rx[j & 1], y = ry[j >> 1 & 1], z = rz[j >> 2 & 1];
                idx[j] = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(x, S));
                oob |= ((((unsigned)x >= (unsigned)S) | ((unsigned)y >= (unsigned)S) | ((unsigned)z >= (unsigned)S)) ? 1u : 0u) << j;
            }
        } else if (row_mode) {
            // the 8 cubes form a row (same y and z centre, e.g. 8 x-neighbours of a level grid): the y and z cell
            // arithmetic is shared, 10 axis computations instead of 24
            const int y = cell_of(Y + sh.cen[0][1], T.yMin, T.scale), z = cell_of(Z + sh.cen[0][2], T.zMin, T.scale);
            const int rowbase = (clamp_cell(z, S) * S + clamp_cell(y, S)) * S;
            const unsigned yz_out = (((unsigned)y >= (unsigned)S) | ((unsigned)z >= (unsigned)S)) ? 1u : 0u;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int x = cell_of(X + sh.cen[j][0], T.xMin, T.scale);
                idx[j] = (unsigned)(rowbase + clamp_cell(x, S));
                oob |= (yz_out | (((unsigned)x >= (unsigned)S) ? 1u : 0u)) << j;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int x = cell_of(X + sh.cen[j][0], T.xMin, T.scale), y = cell_of(Y + sh.cen[j][1], T.yMin, T.scale), z = cell_of(Z + sh.cen[j][2], T.zMin, T.scale);
                idx[j] = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(x, S));
                oob |= ((((unsigned)x >= (unsigned)S) | ((unsigned)y >= (unsigned)S) | ((unsigned)z >= (unsigned)S)) ? 1u : 0u) << j;
            }
        }
        return oob;
    };
    // diagnosis (build with -DGOICP_TILE_STAMPS, run with GOICP_DBG & 64; tools/tile_phases.py): cycle stamps of wave 0 / lane 0 of every 64th
    // workgroup, summed into T.dbg_times[0..7]. Compiled out by default: the stamp state costs the metric kernel registers it does not have.
#ifdef GOICP_TILE_STAMPS
    const bool stamp = DBG_ON(T, 64) && T.dbg_times && tid == 0 && (blockIdx.x & 63) == 0;
    unsigned long long tl_ = stamp ? clock64() : 0ull;
    const unsigned long long tw0_ = stamp ? wall_clock64() : 0ull;      // [8]: the same tile in 100 MHz wall ticks -> in-kernel clock = cycles / ticks x 100 MHz
#define TSTAMP(k) do { if (stamp) { const unsigned long long now_ = clock64(); atomicAdd(&T.dbg_times[k], now_ - tl_); tl_ = now_; } } while (0)
#else
    constexpr bool stamp = false;
#define TSTAMP(k) do { } while (0)
#endif
    // fixed-window histogram adds (XP): zeros and out-of-window flags are wave-uniform counts (ballots), flushed once per wave
    int fw_z[NT]; unsigned fw_hi = 0u;
#pragma unroll
    for (int j = 0; j < NT; j++) fw_z[j] = 0;
    auto fw_add = [&](int j, float d) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) unsigned lds_u32;
        const unsigned bits = __float_as_uint(d);
        const bool nz = bits != 0u;
        fw_z[j] += __popcll(__ballot(!nz));            // (includes the lanes past Nd: BLOCK * PPT - Nd of them per cube, subtracted at the scan)
        int k = (int)(bits >> FW_SHIFT) - FW_KBASE;
        if constexpr (H32) {
            // 32-bit bins: shift, subtract, clamp, one LDS add; residuals above the window simply land in bin 1023, which is never trusted (scan)
            k = min(max(k, 0), 1023);
            if (nz) atomicAdd((unsigned*)((lds_u32*)&sh.hist[0][0] + j * 1024 + k), 1u);
        } else {
            fw_hi |= (__ballot(nz && k > 1023) ? 1u : 0u) << j;
            k = min(max(k, 0), 1023);
            if (nz) atomicAdd((unsigned*)((lds_u32*)&sh.hist[0][0] + j * FW_HP_WORDS + (k >> 1)), 1u << ((k & 1) * 16));
        }
    };
    auto fw_flush = [&]() __attribute__((always_inline)) {
        if ((tid & 63) == 0) {
#pragma unroll
            for (int j = 0; j < NT; j++) if (fw_z[j]) atomicAdd(&sh.zc[j], fw_z[j]);
            if (fw_hi) atomicOr(&sh.himask, fw_hi);
        }
#pragma unroll
        for (int j = 0; j < NT; j++) fw_z[j] = 0;
        fw_hi = 0u;
    };
    float4 pnext = make_float4(0.f, 0.f, 0.f, 0.f); float rnext = 0.f;
    if (JIT) { const int i0 = min(tid, Nd - 1); pnext = jit_rot[i0]; rnext = jit_rho ? jit_rho[i0] : 0.f; }
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        float X, Y, Z, W, RH;
        if (JIT) {
            X = pnext.x; Y = pnext.y; Z = pnext.z; W = pnext.w; RH = rnext;
            if (XP && tid + m * BLOCK >= Nd) W = 0.f;      // lanes past Nd: residual exactly 0 (weight 0, radius >= 0); no per-value validity tests below
            if (m + 1 < PPT) { const int in = min(tid + (m + 1) * BLOCK, Nd - 1); pnext = jit_rot[in]; rnext = jit_rho ? jit_rho[in] : 0.f; }
        } else { X = px[m]; Y = py[m]; Z = pz[m]; W = pw[m]; RH = prho[m]; }
        unsigned idx[NT]; float v[NT];
        const unsigned oob = cells(X, Y, Z, idx);
        auto decode_all = [&](const unsigned (&c)[NT]) __attribute__((always_inline)) {
            // codes below LUT_LDS (within 90 cells of the model: nearly all) come from the LDS table without a branch per value; the rest --
            // far cells through the global table, escapes through the float grid -- are patched under ONE wave-wide test
            if (XP && lut_lds) {
                typedef __attribute__((address_space(3))) float lds_f32;
                const lds_f32* q = (const lds_f32*)lut_lds;
                bool far = false;
#pragma unroll
                for (int j = 0; j < NT; j++) far = far || c[j] >= (unsigned)LUT_LDS;
#pragma unroll
                for (int j = 0; j < NT; j++) v[j] = q[min(c[j], (unsigned)(LUT_LDS - 1))];
                if (__any(far ? 1 : 0)) {
#pragma unroll
                    for (int j = 0; j < NT; j++) if (c[j] >= (unsigned)LUT_LDS) v[j] = c[j] != 0xFFFFu ? T.lut[c[j]] : T.dist[idx[j]];
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < NT; j++) v[j] = dt_decode(T, idx[j], c[j], lut_lds);
        };
        if (XPOSE && T.code) {
            static_assert(!XP || (NT == 8 && BLOCK / 64 <= TC), "transposed gathers: 8-cube tiles, one hist row per wave");
            typedef __attribute__((address_space(3))) unsigned lds_u32;
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) u32x4_t lds_u128;
            // (upper half of sh.hist; row k of group g is stored at row (k + g) & 7, so that the transposed side -- lane (g, p) touches row k,
            // column p -- spreads a half-wave over all 32 banks without padding, and the lane = point side stays 16-byte aligned)
            const int gg = (tid & 63) >> 3, pp = tid & 7;
            lds_u32* xs = (lds_u32*)&sh.hist[0][0] + TC * FW_HP_WORDS + (tid >> 6) * 512 + gg * 64;
            const int p8 = ((pp + gg) & 7) * 8;
            { u32x4_t w0 = {idx[0], idx[1], idx[2], idx[3]}; *(lds_u128*)(xs + p8) = w0; }
            { u32x4_t w1 = {idx[4], idx[5], idx[6], idx[7]}; *(lds_u128*)(xs + p8 + 4) = w1; }
            wave_sync_lds();
            unsigned t[NT], c[NT];
#pragma unroll
            for (int k = 0; k < NT; k++) t[k] = xs[((k + gg) & 7) * 8 + pp];
#pragma unroll
            for (int k = 0; k < NT; k++) c[k] = T.code[t[k]];
#pragma unroll
            for (int k = 0; k < NT; k++) xs[((k + gg) & 7) * 8 + pp] = c[k];
            wave_sync_lds();
            const u32x4_t ca = *(lds_u128*)(xs + p8), cb = *(lds_u128*)(xs + p8 + 4);
            c[0] = ca.x; c[1] = ca.y; c[2] = ca.z; c[3] = ca.w; c[4] = cb.x; c[5] = cb.y; c[6] = cb.z; c[7] = cb.w;
            decode_all(c);
        } else if (T.code) {
            // coded grid: 2-byte gathers (half the cache lines of the float grid), then the value from the LUT
            unsigned c[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) c[j] = T.code[DBG_ON(T, 2) ? (idx[j] & 1023u) : idx[j]];
            decode_all(c);
        } else {
#pragma unroll
            for (int j = 0; j < NT; j++) v[j] = T.dist[DBG_ON(T, 2) ? (idx[j] & 1023u) : idx[j]];
        }
        if (__any((int)oob)) {
            {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                if ((oob >> j) & 1u) {
                    float ext; (void)dt_index(T, X + sh.cen[j][0], Y + sh.cen[j][1], Z + sh.cen[j][2], ext);
                    v[j] = dt_finish(T, v[j], ext);
                }
            }
            }
        }
#pragma unroll
        for (int j = 0; j < NT; j++) {
            float d = W * v[j];
            if (use_rho) d -= RH;
            if (d < 0.f) d = 0.f;
            if (JIT) asm volatile("" : "+v"(d));           // the residual is formed HERE (otherwise the compiler may sink these operations to the
            r[j][m] = d;                                   // first use and keep value, weight and radius alive instead: registers, spills)
        }
        if constexpr (XP) {
            // fixed-window histogram of the first selection, accumulated here beside the outstanding gathers (the caller has cleared it: fw_hist_reset)
            if (T.doTrim) {
#pragma unroll
                for (int j = 0; j < NT; j++) fw_add(j, r[j][m]);
            }
        }
        if (JIT) __builtin_amdgcn_sched_barrier(0);        // keep the point fetches one ahead instead of all up front
    }
    if constexpr (XP) { if (T.doTrim) fw_flush(); }
    TSTAMP(0);
    // everything behind the gathers (selection + sums -> sh.ub / sh.lb) as one block, so that the fused form can run it twice on the same
    // distance values: first for the call without rotation uncertainty, then again after subtracting the radius of the other call
    auto finish = [&](const bool have_hist) __attribute__((always_inline)) {      // (out of line the residual registers would go through scratch memory)
    if constexpr (XP) {
    if (T.doTrim) {
        // fixed-window selection (see fw_hist_reset): four workgroup barriers when the histogram came with the gathers, six otherwise
        const int wv = tid >> 6, ln = tid & 63;
        const int nrm = NORM ? NORM : T.norm;
        constexpr int NW = BLOCK / 64;
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) u32x4_t lds_u128;
        if (!have_hist) {
            __syncthreads();                          // (the histograms were cleared by the caller, behind the previous selection's last barrier: fw_hist_reset)
#pragma unroll
            for (int m = 0; m < PPT; m++) {
#pragma unroll
                for (int j = 0; j < NT; j++) fw_add(j, r[j][m]);
            }
            fw_flush();
        }
        __syncthreads();                                                            // all adds are in
        TSTAMP(2);
        for (int j = wv; j < NT; j += NW) {             // one wave scans the 1024 bins of a cube: 16 per lane = two 16-byte reads
            unsigned c[16]; int mine = 0;
            if constexpr (H32) {
                const lds_u128* hp = (const lds_u128*)&sh.hist[0][0] + j * 256 + 4 * ln;
                const u32x4_t w0 = hp[0], w1 = hp[1], w2 = hp[2], w3 = hp[3];
                c[0] = w0.x; c[1] = w0.y; c[2] = w0.z; c[3] = w0.w; c[4] = w1.x; c[5] = w1.y; c[6] = w1.z; c[7] = w1.w;
                c[8] = w2.x; c[9] = w2.y; c[10] = w2.z; c[11] = w2.w; c[12] = w3.x; c[13] = w3.y; c[14] = w3.z; c[15] = w3.w;
            } else {
            const lds_u128* hp = (const lds_u128*)&sh.hist[0][0] + j * (FW_HP_WORDS / 4) + 2 * ln;
            const u32x4_t wa = hp[0], wb = hp[1];
            c[0] = wa.x & 0xFFFFu; c[1] = wa.x >> 16; c[2] = wa.y & 0xFFFFu; c[3] = wa.y >> 16; c[4] = wa.z & 0xFFFFu; c[5] = wa.z >> 16; c[6] = wa.w & 0xFFFFu; c[7] = wa.w >> 16;
            c[8] = wb.x & 0xFFFFu; c[9] = wb.x >> 16; c[10] = wb.y & 0xFFFFu; c[11] = wb.y >> 16; c[12] = wb.z & 0xFFFFu; c[13] = wb.z >> 16; c[14] = wb.w & 0xFFFFu; c[15] = wb.w >> 16;
            }
#pragma unroll
            for (int q = 0; q < 16; q++) mine += (int)c[q];
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (ln >= off) incl += t; }
            const int zc = sh.zc[j] - (BLOCK * PPT - Nd);     // residuals that are exactly 0, without the lanes past Nd
            const int rank = T.inlierNum - 1;
            if (rank < zc) {                                  // the threshold is one of the zeros
                if (ln == 0) { sh.prefix[j] = 0u; sh.rem[j] = rank; sh.ncand[j] = -1; }
                continue;
            }
            const int rem = rank - zc;
            const unsigned long long over = __ballot(incl > rem);
            const int owner = over ? __ffsll((long long)over) - 1 : 63;
            if (ln == owner) {
                int rr = rem - (incl - mine);
                int b = 0;
#pragma unroll
                for (int q = 0; q < 15; q++) { if (b == q && rr >= (int)c[q]) { rr -= (int)c[q]; b = q + 1; } }
                unsigned cb = c[0];
#pragma unroll
                for (int q = 1; q < 16; q++) if (b == q) cb = c[q];
                const int bin = 16 * ln + b;
                // threshold below the window (bin 0 also holds everything under 2^-12), above it (bin 1023 holds everything from 2^4 on), or in a
                // crowded bin: the generic byte passes decide
                if (bin == 0 || (bin == 1023 && (H32 || ((sh.himask >> j) & 1u))) || cb > CAND_CAP) sh.fallback = 1;
                sh.prefix[j] = (unsigned)(FW_KBASE + bin) << FW_SHIFT;
                sh.rem[j] = rr;                               // rank among the members of the threshold bin
            }
        }
        __syncthreads();
        TSTAMP(3);
        if (!sh.fallback) {
            // one pass over the registers: everything below the threshold bin goes into the sums (lane-local float partials of <= PPT addends, fp64
            // across lanes), the bin's members (a handful) into the candidate list
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const unsigned pk = sh.prefix[j] >> FW_SHIFT;
                const bool known = sh.ncand[j] < 0;           // threshold already known (zero): nothing below it
                const float mtd = sh.mtd[j];
                float fub = 0.f, flb = 0.f;
                const unsigned pkc = known ? 0xFFFFFFFFu : pk;     // (threshold already known: no candidates)
#pragma unroll
                for (int m = 0; m < PPT; m++) {
                    // zeros (clamped residuals, lanes past Nd) have key 0: they pass the test and add 0 to both sums (mtd >= 0)
                    const unsigned bits = __float_as_uint(r[j][m]);
                    const unsigned key = bits >> FW_SHIFT;
                    const float dm = key < pk ? r[j][m] : 0.f;
                    const float em = fmaxf(dm - mtd, 0.f);
                    fub += nrm == 2 ? dm * dm : (nrm == 1 ? dm : 0.f);
                    flb += nrm == 2 ? em * em : (nrm == 1 ? em : 0.f);
                    if (key == pkc) { const int pos = atomicAdd(&sh.ncand[j], 1); if (pos < CAND_CAP) sh.cand[j][pos] = bits; }
                }
                const double u = wave_sum((double)fub), l = wave_sum((double)flb);
                if (ln == 0) { sh.red[wv][2 * j] = u; sh.red[wv][2 * j + 1] = l; }
            }
            __syncthreads();
            TSTAMP(4);
            // wave j finishes cube j: low 17 bits of the threshold by bisection over the candidates; the candidates below it and the copies of the
            // threshold that belong to the inlier set join the partial sums of all waves
            for (int j = wv; j < NT; j += NW) {
                const int nc = sh.ncand[j];
                const float mtd = sh.mtd[j];
                unsigned pre = sh.prefix[j]; int remj = sh.rem[j];
                double cu = 0.0, cl = 0.0;
                if (nc >= 0) {
                    const bool v0 = ln < nc, v1 = ln + 64 < nc;
                    const unsigned c0 = v0 ? sh.cand[j][ln] : 0u, c1 = v1 ? sh.cand[j][ln + 64] : 0u;
#pragma unroll
                    for (int b = FW_SHIFT - 1; b >= 0; b--) {
                        const unsigned cnd = pre | (1u << b);
                        const int cnt = __popcll(__ballot(v0 && c0 < cnd)) + __popcll(__ballot(v1 && c1 < cnd));
                        if (cnt <= remj) pre = cnd;               // the rem-th member is >= cnd
                    }
                    const int less = __popcll(__ballot(v0 && c0 < pre)) + __popcll(__ballot(v1 && c1 < pre));
                    remj -= less;                                 // rank among the copies of the threshold
                    // (a lane's two candidate terms are added in double: which members of the bin share a lane depends on the order of the LDS
                    // adds that collected them, and a float sum of two terms would round differently from run to run)
                    if (v0 && c0 < pre) { const float d = __uint_as_float(c0), e = d - mtd; cu += (double)(nrm == 2 ? d * d : (nrm == 1 ? d : 0.f)); if (e > 0.f) cl += (double)(nrm == 2 ? e * e : (nrm == 1 ? e : 0.f)); }
                    if (v1 && c1 < pre) { const float d = __uint_as_float(c1), e = d - mtd; cu += (double)(nrm == 2 ? d * d : (nrm == 1 ? d : 0.f)); if (e > 0.f) cl += (double)(nrm == 2 ? e * e : (nrm == 1 ? e : 0.f)); }
                }
                cu += ln < NW ? sh.red[ln][2 * j] : 0.0;
                cl += ln < NW ? sh.red[ln][2 * j + 1] : 0.0;
                double u = wave_sum(cu), l = wave_sum(cl);
                if (ln == 0) {
                    const float d = __uint_as_float(pre);
                    const float e = d - mtd;
                    const double n = (double)(remj + 1);          // copies of the threshold inside the inlier set
                    if (nrm == 2) { u += n * (double)(d * d); if (e > 0.f) l += n * (double)(e * e); }
                    else if (nrm == 1) { u += n * (double)d; if (e > 0.f) l += n * (double)e; }
                    sh.ub[j] = (float)u; sh.lb[j] = (float)l;
                }
            }
            __syncthreads();
            TSTAMP(5);
#ifdef GOICP_TILE_STAMPS
            if (stamp && (DUAL ? !have_hist : have_hist)) { atomicAdd(&T.dbg_times[7], 1ull); atomicAdd(&T.dbg_times[8], wall_clock64() - tw0_); }
#endif
            return;
        }
        // generic path: four byte passes over the residual registers, MSB first (256-bin histograms in the rows of sh.hist -- the fixed-window
        // histograms and the transposition scratch are not needed any more), then the order-free sums below
        unsigned thr[NT]; int below[NT];
        if (tid < NT) { sh.prefix[tid] = 0u; sh.rem[tid] = T.inlierNum - 1; }
        for (int pass = 3; pass >= 0; pass--) {
            __syncthreads();
            for (int k = tid; k < NT * 256; k += BLOCK) sh.hist[k >> 8][k & 255] = 0u;
            __syncthreads();
            const unsigned himask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
#pragma unroll
            for (int m = 0; m < PPT; m++) {
                const bool valid = tid + m * BLOCK < Nd;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const unsigned bits = __float_as_uint(r[j][m]);
                    const bool take = valid && ((bits ^ sh.prefix[j]) & himask) == 0u;
                    hist_add(sh.hist[j], (bits >> (8 * pass)) & 255u, take, ln);
                }
            }
            __syncthreads();
            for (int j = wv; j < NT; j += NW) {
                const unsigned c0 = sh.hist[j][4 * ln], c1 = sh.hist[j][4 * ln + 1], c2 = sh.hist[j][4 * ln + 2], c3 = sh.hist[j][4 * ln + 3];
                const int mine = (int)(c0 + c1 + c2 + c3);
                int incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (ln >= off) incl += t; }
                const int rem = sh.rem[j];
                const unsigned long long over = __ballot(incl > rem);
                const int owner = over ? __ffsll((long long)over) - 1 : 63;
                if (ln == owner) {
                    int rr = rem - (incl - mine);
                    unsigned dgt = 4u * ln;
                    if (rr >= (int)c0) { rr -= (int)c0; dgt++; if (rr >= (int)c1) { rr -= (int)c1; dgt++; if (rr >= (int)c2) { rr -= (int)c2; dgt++; } } }
                    sh.prefix[j] |= dgt << (8 * pass);
                    sh.rem[j] = rr;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; j++) { thr[j] = sh.prefix[j]; below[j] = T.inlierNum - 1 - sh.rem[j]; }   // rem = rank among the copies of thr
#pragma unroll
        for (int j = 0; j < NT; j++) {
            float fub = 0.f, flb = 0.f;
            const float mtd = sh.mtd[j];
            const unsigned thrj = thr[j];
#pragma unroll
            for (int m = 0; m < PPT; m++) {
                const float d = r[j][m];
                const bool in = (tid + m * BLOCK < Nd) && __float_as_uint(d) < thrj;
                const float e = d - mtd;
                const float tu = T.norm == 2 ? d * d : (T.norm == 1 ? d : 0.f);
                const float tl = T.norm == 2 ? e * e : (T.norm == 1 ? e : 0.f);
                fub += in ? tu : 0.f;
                flb += (in && e > 0.f) ? tl : 0.f;
            }
            const double u = wave_sum((double)fub), l = wave_sum((double)flb);
            if (ln == 0) { sh.red[wv][2 * j] = u; sh.red[wv][2 * j + 1] = l; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; j++) {
            if (tid == j) {
                double u = 0.0, l = 0.0;
                for (int w = 0; w < NW; w++) { u += sh.red[w][2 * j]; l += sh.red[w][2 * j + 1]; }
                const float d = __uint_as_float(thr[j]);
                const float e = d - sh.mtd[j];
                const double n = (double)(T.inlierNum - below[j]);          // the copies of the threshold value that belong to the inlier set
                if (T.norm == 2) { u += n * (double)(d * d); if (e > 0.f) l += n * (double)(e * e); }
                else if (T.norm == 1) { u += n * (double)d; if (e > 0.f) l += n * (double)e; }
                sh.ub[j] = (float)u; sh.lb[j] = (float)l;
            }
        }
        __syncthreads();
        return;
    }
    }
    // (the just-in-time kernels carry the reference-order branches only in their parity_mode 2 instantiation: beside the fast path they cost it registers)
    if ((JIT ? XBIG : (BLOCK * PPT <= 1024 && T.exact)) && T.doTrim) {
        // parity mode, trimmed: residual rows to LDS, one lane per cube runs the reference's selection (same swaps, same final
        // order), then the inlier sums are taken sequentially in float over that order -- bit-exact with jly_goicp.cpp:384-415
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const int i = tid + m * BLOCK;
            if (i < Nd) {
#pragma unroll
                for (int j = 0; j < NT; j++) rbuf[j * Nd + i] = r[j][m];
            }
        }
        __syncthreads();
        for (int j = tid >> 6; j < NT; j += BLOCK / 64)
            wave_intro_select<JIT>(rbuf + j * Nd, reinterpret_cast<int*>(rbuf + NT * Nd) + j * Nd, Nd, T.inlierNum - 1, tid & 63, DBG_ON(T, 32) && T.dbg_times ? T.dbg_times + 8 : nullptr, T.regsel != 0);      // (JIT: rows in HBM)
        __syncthreads();
        if (tid < 2 * NT) {
            const int j = tid & (NT - 1);
            const bool is_lb = tid >= NT;
            const float* rr = rbuf + j * Nd;
            const float sub = is_lb ? sh.mtd[j] : 0.f;
            const bool sq = T.norm == 2, lin = T.norm == 1;
            float acc = 0.f;
            auto add = [&](float d) {
                const float e = d - sub;                           // (d - 0 == d exactly for the ub chain)
                float term = sq ? e * e : (lin ? e : 0.f);
                if (is_lb && !(e > 0.f)) term = 0.f;
                acc += term;
            };
            int i = 0;
            for (; i + 16 <= T.inlierNum; i += 16) {                // 16 LDS reads in flight, then the 16 dependent adds
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = rr[i + q];
#pragma unroll
                for (int q = 0; q < 16; q++) add(v[q]);
            }
            for (; i < T.inlierNum; i++) add(rr[i]);
            if (is_lb) sh.lb[j] = acc; else sh.ub[j] = acc;
        }
        __syncthreads();
        return;
    }
    if ((JIT ? XBIG : T.exact != 0) && !T.doTrim) {
        // reference order: float, sequential over the point index (bit-exact with jly_goicp.cpp:393-415). Every lane first
        // turns its residuals into the two addends (d*d and max(d-mtd,0)^2, or the norm-1 forms) in parallel; 16 lanes then
        // run the 16 sequential float sums (8 ub chains, 8 lb chains), one add per element. Skipping a term (e <= 0) and
        // adding +0.0f leave the same bits because the running sums are never negative.
        float* tub = rbuf; float* tlb = rbuf + NT * Nd;
        const bool sq = T.norm == 2, lin = T.norm == 1;
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const int i = tid + m * BLOCK;
            if (i < Nd) {
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const float d = r[j][m], e = d - sh.mtd[j];
                    tub[j * Nd + i] = sq ? d * d : (lin ? d : 0.f);
                    tlb[j * Nd + i] = e > 0.f ? (sq ? e * e : (lin ? e : 0.f)) : 0.f;
                }
            }
        }
        __syncthreads();
        if (tid < 2 * NT) {
            const float* tt = rbuf + tid * Nd;          // lanes 0-7: ub rows, lanes 8-15: lb rows
            float acc = 0.f;
            int i = 0;
            for (; i + 16 <= Nd; i += 16) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = tt[i + q];
#pragma unroll
                for (int q = 0; q < 16; q++) acc += v[q];
            }
            for (; i < Nd; i++) acc += tt[i];
            if (tid < NT) sh.ub[tid] = acc; else sh.lb[tid - NT] = acc;
        }
        __syncthreads();
        return;
    }
    // ---- inlier threshold per cube = the inlierNum-th smallest residual, by MSB-first radix select on the float bit
    //      pattern (residuals are >= 0, so the pattern orders like the value) with LDS histograms. Replaces intro_select
    //      (reference jly_sorting.hpp:229): the selected multiset is the same, the order is not needed (order-free sums).
    unsigned thr[NT]; int below[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) { thr[j] = 0u; below[j] = 0; }
    if (T.doTrim && !DBG_ON(T, 1)) {
        const int wv = tid >> 6, ln = tid & 63;
        constexpr int NW = BLOCK / 64;
        // Five workgroup barriers for the whole selection + sums. Everything initialised here is first used after barrier 1.
        for (int k = tid; k < NT * 1024; k += BLOCK) (&sh.hist[0][0])[k] = 0u;
        if (tid < NT) { sh.prefix[tid] = 0u; sh.rem[tid] = T.inlierNum - 1; sh.zc[tid] = 0; sh.ncand[tid] = 0; }
        if (tid == 0) sh.fallback = 0;
        // (1) largest residual per cube (fixes the 8-octave window of the histogram): per-wave maxima by plain stores
#pragma unroll
        for (int j = 0; j < NT; j++) {
            unsigned mx = 0u;
#pragma unroll
            for (int m = 0; m < PPT; m++) if (tid + m * BLOCK < Nd) mx = max(mx, __float_as_uint(r[j][m]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_down((int)mx, off, 64));
            if (ln == 0) sh.wmax[wv][j] = mx;
        }
        __syncthreads();                                                            // barrier 1
        TSTAMP(1);
        int base16[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            unsigned mx = ln < NW ? sh.wmax[ln][j] : 0u;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));   // NW <= 16
            mx = (unsigned)__shfl((int)mx, 0, 64);
            base16[j] = max(0, (int)(mx >> 16) - 1023);
            if (tid == j) sh.maxb[j] = mx;
        }
        // (2) ONE histogram pass: 10-bit digit = top 16 bits (sign, exponent, 7 mantissa bits) relative to the cube's maximum:
        //     1024 monotone bins over the 8 octaves below the maximum, so the values spread over hundreds of LDS words
        //     instead of piling onto the 2-3 exponent bytes a plain byte pass would see. Residuals clamped to exactly 0
        //     (frequent once a rotation-uncertainty radius is subtracted) are bit-identical and would serialise on one
        //     LDS word, so they are counted with ballots and kept out of the histogram.
#pragma unroll
        for (int j = 0; j < NT; j++) {
            int z = 0;
#pragma unroll
            for (int m = 0; m < PPT; m++) {
                const bool valid = tid + m * BLOCK < Nd;
                const unsigned bits = __float_as_uint(r[j][m]);
                z += __popcll(__ballot(valid && bits == 0u));
                if (valid && bits != 0u) atomicAdd(&sh.hist[j][max(0, (int)(bits >> 16) - base16[j])], 1u);
            }
            if (ln == 0 && z) atomicAdd(&sh.zc[j], z);
        }
        __syncthreads();                                                            // barrier 2
        TSTAMP(2);
        for (int j = wv; j < NT; j += NW) {             // (3) one wave scans the 1024 bins of a cube: 16 per lane
            unsigned c[16]; int mine = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) { c[q] = sh.hist[j][16 * ln + q]; mine += (int)c[q]; }
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (ln >= off) incl += t; }
            const int zc = sh.zc[j];
            const int rank = sh.rem[j];
            if (rank < zc) {                                  // the threshold is one of the zeros
                if (ln == 0) { sh.prefix[j] = 0u; sh.rem[j] = rank; sh.ncand[j] = -1; }
                continue;
            }
            const int rem = rank - zc;
            const unsigned long long over = __ballot(incl > rem);
            const int owner = over ? __ffsll((long long)over) - 1 : 63;
            if (ln == owner) {
                int rr = rem - (incl - mine);
                int b = 0;
#pragma unroll
                for (int q = 0; q < 15; q++) { if (b == q && rr >= (int)c[q]) { rr -= (int)c[q]; b = q + 1; } }
                unsigned cb = c[0];
#pragma unroll
                for (int q = 1; q < 16; q++) if (b == q) cb = c[q];
                const int bin = 16 * ln + b;
                const int b16 = max(0, (int)(sh.maxb[j] >> 16) - 1023);
                // threshold below the window, or a crowded bin: the generic byte passes decide
                if ((bin == 0 && b16 > 0) || cb > CAND_CAP) sh.fallback = 1;
                sh.prefix[j] = (unsigned)(b16 + bin) << 16;
                sh.rem[j] = rr;                               // rank among the members of the threshold bin
            }
        }
        __syncthreads();                                                            // barrier 3
        TSTAMP(3);
        if (!sh.fallback) {
            // (4) one pass over the registers: everything below the threshold bin goes into the sums (lane-local float partials
            //     of <= PPT addends, fp64 across lanes), the bin's members (a handful) into the candidate list
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const unsigned p16 = sh.prefix[j] >> 16;
                const bool known = sh.ncand[j] < 0;           // threshold already known (zero): nothing below it
                const float mtd = sh.mtd[j];
                float fub = 0.f, flb = 0.f;
#pragma unroll
                for (int m = 0; m < PPT; m++) {
                    const float d = r[j][m];
                    const unsigned bits = __float_as_uint(d);
                    const bool valid = tid + m * BLOCK < Nd && bits != 0u;      // zeros add nothing to either sum
                    const unsigned b16 = bits >> 16;
                    const bool in = valid && b16 < p16;
                    const float e = d - mtd;
                    const float tu = T.norm == 2 ? d * d : (T.norm == 1 ? d : 0.f);
                    const float tl = T.norm == 2 ? e * e : (T.norm == 1 ? e : 0.f);
                    fub += in ? tu : 0.f;
                    flb += (in && e > 0.f) ? tl : 0.f;
                    if (valid && !known && b16 == p16) { const int pos = atomicAdd(&sh.ncand[j], 1); if (pos < CAND_CAP) sh.cand[j][pos] = bits; }
                }
                const double u = wave_sum((double)fub), l = wave_sum((double)flb);
                if (ln == 0) { sh.red[wv][2 * j] = u; sh.red[wv][2 * j + 1] = l; }
            }
            __syncthreads();                                                        // barrier 4
            TSTAMP(4);
            // (5) wave j finishes cube j: low 16 bits of the threshold by bisection over the candidates, the candidates below it and
            //     the copies of the threshold that belong to the inlier set join the partial sums of all waves
            for (int j = wv; j < NT; j += NW) {
                const int nc = sh.ncand[j];
                const float mtd = sh.mtd[j];
                unsigned pre = sh.prefix[j]; int remj = sh.rem[j];
                double cu = 0.0, cl = 0.0;
                if (nc >= 0) {
                    const bool v0 = ln < nc, v1 = ln + 64 < nc;
                    const unsigned c0 = v0 ? sh.cand[j][ln] : 0u, c1 = v1 ? sh.cand[j][ln + 64] : 0u;
#pragma unroll
                    for (int b = 15; b >= 0; b--) {
                        const unsigned cnd = pre | (1u << b);
                        const int cnt = __popcll(__ballot(v0 && c0 < cnd)) + __popcll(__ballot(v1 && c1 < cnd));
                        if (cnt <= remj) pre = cnd;               // the rem-th member is >= cnd
                    }
                    const int less = __popcll(__ballot(v0 && c0 < pre)) + __popcll(__ballot(v1 && c1 < pre));
                    remj -= less;                                 // rank among the copies of the threshold
                    // (a lane's two candidate terms are added in double: which members of the bin share a lane depends on the order of the LDS
                    // adds that collected them, and a float sum of two terms would round differently from run to run)
                    if (v0 && c0 < pre) { const float d = __uint_as_float(c0), e = d - mtd; cu += (double)(T.norm == 2 ? d * d : (T.norm == 1 ? d : 0.f)); if (e > 0.f) cl += (double)(T.norm == 2 ? e * e : (T.norm == 1 ? e : 0.f)); }
                    if (v1 && c1 < pre) { const float d = __uint_as_float(c1), e = d - mtd; cu += (double)(T.norm == 2 ? d * d : (T.norm == 1 ? d : 0.f)); if (e > 0.f) cl += (double)(T.norm == 2 ? e * e : (T.norm == 1 ? e : 0.f)); }
                }
                // partial sums of all waves (lane w holds wave w's), plus this wave's candidate terms
                cu += ln < NW ? sh.red[ln][2 * j] : 0.0;
                cl += ln < NW ? sh.red[ln][2 * j + 1] : 0.0;
                double u = wave_sum(cu), l = wave_sum(cl);
                if (ln == 0) {
                    const float d = __uint_as_float(pre);
                    const float e = d - mtd;
                    const double n = (double)(remj + 1);          // copies of the threshold inside the inlier set
                    if (T.norm == 2) { u += n * (double)(d * d); if (e > 0.f) l += n * (double)(e * e); }
                    else if (T.norm == 1) { u += n * (double)d; if (e > 0.f) l += n * (double)e; }
                    sh.ub[j] = (float)u; sh.lb[j] = (float)l;
                }
            }
            __syncthreads();                                                        // barrier 5
            TSTAMP(5);
#ifdef GOICP_TILE_STAMPS
            if (stamp) { atomicAdd(&T.dbg_times[7], 1ull); atomicAdd(&T.dbg_times[8], wall_clock64() - tw0_); }
#endif
            return;
        }
        const bool generic = sh.fallback != 0;
        if (generic) { if (tid < NT) { sh.prefix[tid] = 0u; sh.rem[tid] = T.inlierNum - 1; } }
        // (4) generic path: four byte passes, MSB first, when the window missed or a bin was crowded
        for (int pass = generic ? 3 : -1; pass >= 0; pass--) {
            __syncthreads();
            for (int k = tid; k < NT * 256; k += BLOCK) sh.hist[k >> 8][k & 255] = 0u;
            __syncthreads();
            const unsigned himask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
#pragma unroll
            for (int m = 0; m < PPT; m++) {
                const bool valid = tid + m * BLOCK < Nd;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const unsigned bits = __float_as_uint(r[j][m]);
                    const bool take = valid && ((bits ^ sh.prefix[j]) & himask) == 0u;
                    hist_add(sh.hist[j], (bits >> (8 * pass)) & 255u, take, ln);
                }
            }
            __syncthreads();
            for (int j = wv; j < NT; j += BLOCK / 64) {
                const unsigned c0 = sh.hist[j][4 * ln], c1 = sh.hist[j][4 * ln + 1], c2 = sh.hist[j][4 * ln + 2], c3 = sh.hist[j][4 * ln + 3];
                const int mine = (int)(c0 + c1 + c2 + c3);
                int incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (ln >= off) incl += t; }
                const int rem = sh.rem[j];
                const unsigned long long over = __ballot(incl > rem);
                const int owner = over ? __ffsll((long long)over) - 1 : 63;
                if (ln == owner) {
                    int rr = rem - (incl - mine);
                    unsigned dgt = 4u * ln;
                    if (rr >= (int)c0) { rr -= (int)c0; dgt++; if (rr >= (int)c1) { rr -= (int)c1; dgt++; if (rr >= (int)c2) { rr -= (int)c2; dgt++; } } }
                    sh.prefix[j] |= dgt << (8 * pass);
                    sh.rem[j] = rr;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; j++) { thr[j] = sh.prefix[j]; below[j] = T.inlierNum - 1 - sh.rem[j]; }   // rem = rank among the copies of thr
    }
    // ---- sums in fp64 (order-free), float at the end; one cube at a time to keep registers low
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        // a lane's own PPT addends are summed in float (PPT <= 16 terms), everything across lanes in fp64
        float fub = 0.f, flb = 0.f;
        const float mtd = sh.mtd[j];
        const unsigned thrj = thr[j];
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const float d = r[j][m];
            const bool in = (tid + m * BLOCK < Nd) && (!T.doTrim || __float_as_uint(d) < thrj);
            const float e = d - mtd;
            const float tu = T.norm == 2 ? d * d : (T.norm == 1 ? d : 0.f);
            const float tl = T.norm == 2 ? e * e : (T.norm == 1 ? e : 0.f);
            fub += in ? tu : 0.f;
            flb += (in && e > 0.f) ? tl : 0.f;
        }
        const double u = wave_sum((double)fub), l = wave_sum((double)flb);
        if (lane == 0) { sh.red[wave][2 * j] = u; sh.red[wave][2 * j + 1] = l; }
    }
    __syncthreads();
    // every lane holds thr[]/below[] (wave-uniform); lane j finishes cube j
#pragma unroll
    for (int j = 0; j < NT; j++) {
        if (tid == j) {
            double u = 0.0, l = 0.0;
            for (int w = 0; w < BLOCK / 64; w++) { u += sh.red[w][2 * j]; l += sh.red[w][2 * j + 1]; }
            if (T.doTrim) {
                // the copies of the threshold value that belong to the inlier set
                const float d = __uint_as_float(thr[j]);
                const float e = d - sh.mtd[j];
                const double n = (double)(T.inlierNum - below[j]);
                if (T.norm == 2) { u += n * (double)(d * d); if (e > 0.f) l += n * (double)(e * e); }
                else if (T.norm == 1) { u += n * (double)d; if (e > 0.f) l += n * (double)e; }
            }
            sh.ub[j] = (float)u; sh.lb[j] = (float)l;
        }
    }
    __syncthreads();
    };
    finish(true);
    if constexpr (DUAL) {
        // (two call sites rather than a loop: inside a loop the compiler hoists the block's lane-constant addresses and spills the residuals)
        if (tid < NT) { sh.ub2[tid] = sh.ub[tid]; sh.lb2[tid] = sh.lb[tid]; }        // (ordered by the barriers of the next finish())
        // the second selection's histograms are cleared here, beside the radius fetches below: every wave has left the first selection (its last
        // phase ends with a workgroup barrier), nothing reads the histograms or the per-cube counters any more
        if constexpr (XP) { if (T.doTrim) fw_hist_reset<BLOCK, H32>(sh); }
        // second bound of the same cubes: minDis -= maxRotDisL; clamp (reference jly_goicp.cpp:375-381) on the products weights*DT kept in r
        int tid2 = tid;
        asm volatile("" : "+v"(tid2));                     // (keeps the radius fetches below the first finish(): hoisted above it they cost PPT registers)
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const float rh = dual_rho[min(tid2 + m * BLOCK, Nd - 1)];
#pragma unroll
            for (int j = 0; j < NT; j++) { float d = r[j][m] - rh; if (d < 0.f) d = 0.f; r[j][m] = d; }
        }
        finish(false);
    }
}

#undef TSTAMP

// Penalty terms at the nl translations in sh.lat[]: incompatibility counts (checkCompatibilities, reference
// jly_goicp.cpp:919-928) into sh.cnt[], c-FPFH means (sumFPFH, jly_goicp.cpp:1689-1697) into sh.fpv[].
template <int BLOCK, int PPT>
__device__ __forceinline__ void lattice_terms(const DevTables& T, TileShm<BLOCK>& sh, int nl, const float (&px)[PPT], const float (&py)[PPT],
                                              const float (&pz)[PPT], float* fpbuf /* global scratch [NLMAX*Nd] */, bool do_compat = true) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int Nd = T.Nd;
    if (T.use_compat && do_compat) {
        for (int l = tid; l < nl; l += BLOCK) sh.cnt[l] = 0;
        __syncthreads();
        unsigned cls[PPT];
#pragma unroll
        for (int m = 0; m < PPT; m++) cls[m] = (tid + m * BLOCK < Nd) ? (unsigned)T.dcls[tid + m * BLOCK] : 0u;
        for (int l = 0; l < nl; l++) {
            const float lx = sh.lat[l][0], ly = sh.lat[l][1], lz = sh.lat[l][2];
            int wave_cnt = 0;
#pragma unroll
            for (int m = 0; m < PPT; m++) {
                bool bad = false;
                if (tid + m * BLOCK < Nd) {
                    unsigned mask = T.cmask[clamped_cell(T, px[m] + lx, py[m] + ly, pz[m] + lz)];
                    bad = ((mask >> cls[m]) & 1u) == 0u;
                }
                wave_cnt += __popcll(__ballot(bad));
            }
            if (lane == 0 && wave_cnt) atomicAdd(&sh.cnt[l], wave_cnt);
        }
        __syncthreads();
    }
    if (T.use_fpfh) {
        for (int l = 0; l < nl; l++) {
            const float lx = sh.lat[l][0], ly = sh.lat[l][1], lz = sh.lat[l][2];
#pragma unroll
            for (int m = 0; m < PPT; m++) {
                const int i = tid + m * BLOCK;
                if (i < Nd) fpbuf[(size_t)l * Nd + i] = fpfh_min(T, i, px[m] + lx, py[m] + ly, pz[m] + lz);
            }
        }
        __syncthreads();
        // the mean is truncated to int by the caller (reference jly_goicp.cpp:468,495), so the float sum must
        // round exactly like the reference's: sequential over the point index
        for (int l = tid; l < nl; l += BLOCK) {
            const float* f = fpbuf + (size_t)l * Nd;
            float s = 0.f;
            for (int i = 0; i < Nd; i++) s += f[i];
            sh.fpv[l] = s / (float)Nd;
        }
        __syncthreads();
    }
}

// Incompatibility counts at the 27 points of the 3x3x3 corner lattice of a node's children (k_inner_bnb).
// Lattice point l = a + 3b + 9c has coordinates (X[a], Y[b], Z[c]), so the (double) cell arithmetic is done 9 times
// per data point instead of 81, and the 27 mask gathers of a point are issued together before any of them is used.
template <int BLOCK, int PPT>
__device__ __forceinline__ void lattice_compat_oct(const DevTables& T, TileShm<BLOCK>& sh, const float (&px)[PPT], const float (&py)[PPT], const float (&pz)[PPT]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int Nd = T.Nd, S = T.S;
    if (tid < 27) sh.cnt[tid] = 0;
    __syncthreads();
    int wc[27];
#pragma unroll
    for (int l = 0; l < 27; l++) wc[l] = 0;
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        const int i = min(tid + m * BLOCK, Nd - 1);
        const bool valid = tid + m * BLOCK < Nd;
        const unsigned cls = (unsigned)T.dcls[i];
        int cx[3], cy[3], cz[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            cx[a] = min(max(cell_of(px[m] + sh.lat[a][0], T.xMin, T.scale), 0), S - 1);
            cy[a] = min(max(cell_of(py[m] + sh.lat[3 * a][1], T.yMin, T.scale), 0), S - 1);
            cz[a] = min(max(cell_of(pz[m] + sh.lat[9 * a][2], T.zMin, T.scale), 0), S - 1);
        }
        unsigned mk[27];
#pragma unroll
        for (int l = 0; l < 27; l++) mk[l] = T.cmask[(size_t)((cz[l / 9] * S + cy[(l / 3) % 3]) * S + cx[l % 3])];
#pragma unroll
        for (int l = 0; l < 27; l++) wc[l] += __popcll(__ballot(valid && ((mk[l] >> cls) & 1u) == 0u));
    }
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < 27; l++) if (wc[l]) atomicAdd(&sh.cnt[l], wc[l]);
    }
    __syncthreads();
}

// c-FPFH means at the 27 points of the children's corner lattice (k_inner_bnb). Per data point: 9 axis computations,
// the 27 nearest-occupied-cell ids fetched together, the point's own descriptor row held in registers, and a lattice
// point that resolves to the same occupied cell as its predecessor reuses the predecessor's minimum (frequent at deep
// levels, where the lattice spacing is below the cell size). The per-lattice-point mean is then summed sequentially in
// float (the reference truncates it to int, so it must round the same way) from rows kept in LDS when they fit.
template <int BLOCK, int PPT>
__device__ __forceinline__ void lattice_fpfh_oct(const DevTables& T, TileShm<BLOCK>& sh, const float (&px)[PPT], const float (&py)[PPT], const float (&pz)[PPT],
                                                 float* rows /* [27*Nd], LDS or global */) {
    const int tid = threadIdx.x;
    const int Nd = T.Nd, S = T.S;
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        const int i = min(tid + m * BLOCK, Nd - 1);
        const bool valid = tid + m * BLOCK < Nd;
        int cx[3], cy[3], cz[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            cx[a] = min(max(cell_of(px[m] + sh.lat[a][0], T.xMin, T.scale), 0), S - 1);
            cy[a] = min(max(cell_of(py[m] + sh.lat[3 * a][1], T.yMin, T.scale), 0), S - 1);
            cz[a] = min(max(cell_of(pz[m] + sh.lat[9 * a][2], T.zMin, T.scale), 0), S - 1);
        }
        int o[27];
#pragma unroll
        for (int l = 0; l < 27; l++) o[l] = T.occ[(size_t)((cz[l / 9] * S + cy[(l / 3) % 3]) * S + cx[l % 3])];
        float v[27];
        const float* Erow = T.fpE + (size_t)i * T.nOcc;
#pragma unroll
        for (int l = 0; l < 27; l++) v[l] = Erow[max(o[l], 0)];            // 27 gathers in flight
#pragma unroll
        for (int l = 0; l < 27; l++) if (valid) rows[(size_t)l * Nd + i] = o[l] < 0 ? 1000000000.f : v[l];
    }
    __syncthreads();
    if (tid < 27) {
        const float* fr = rows + (size_t)tid * Nd;
        float acc = 0.f;
        int i = 0;
        for (; i + 16 <= Nd; i += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = fr[i + q];
#pragma unroll
            for (int q = 0; q < 16; q++) acc += v[q];
        }
        for (; i < Nd; i++) acc += fr[i];
        sh.fpv[tid] = acc / (float)Nd;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// pop_eval_fused: everything a queue pop of a cavity-sized search evaluates (k_inner_bnb, parity mode, untrimmed, penalty terms) in ONE
// gather phase and ONE sequential-sum phase. sweep_tile + lattice_compat_oct + lattice_fpfh_oct do the same work as three phases of their
// own (gathers -> barrier -> 16 / 27 sequential chains -> barrier, three times over: 7.2 k + 7.9 k + 13.6 k cycles of a 40 k-cycle pop at 238
// points). Here a point's 8 distance lookups, 27 compatibility masks and 27 nearest-occupied-cell ids are in flight together (then the 27
// table entries), the addends / c-FPFH rows go to LDS, and after one barrier the 16 bound chains (wave 0) and the 27 c-FPFH chains (wave 1)
// run side by side. Same values, same order of every float sum: same bits (reference jly_goicp.cpp:343-415, 431-550, 919-928, 1642-1697).
// Expects sh.cube / sh.cen / sh.mtd / sh.lat set and sh.cnt[0..26] zeroed behind a barrier.
// ---------------------------------------------------------------------------------------------
template <int BLOCK, int PPT>
__device__ __forceinline__ void pop_eval_fused(const DevTables& T, TileShm<BLOCK>& sh, const float (&px)[PPT], const float (&py)[PPT], const float (&pz)[PPT],
                                               const float (&pw)[PPT], const float (&prho)[PPT], bool use_rho, float* rbuf /* LDS: 2 x TC x Nd addends */,
                                               float* fprows /* [27 x Nd], LDS or global */) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int Nd = T.Nd, S = T.S;
    const bool sq = T.norm == 2, lin = T.norm == 1;
    float* tub = rbuf; float* tlb = rbuf + TC * Nd;
    int wc[27];
#pragma unroll
    for (int l = 0; l < 27; l++) wc[l] = 0;
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        const int i = tid + m * BLOCK;
        const bool valid = i < Nd;
        const int ic = min(i, Nd - 1);
        const float X = px[m], Y = py[m], Z = pz[m];
        // children: two centre coordinates per axis (sweep_tile, OCT)
        int rx[2], ry[2], rz[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            rx[q] = cell_of(X + sh.cen[q][0], T.xMin, T.scale);
            ry[q] = cell_of(Y + sh.cen[2 * q][1], T.yMin, T.scale);
            rz[q] = cell_of(Z + sh.cen[4 * q][2], T.zMin, T.scale);
        }
        unsigned idx[TC]; unsigned oob = 0u;
#pragma unroll
        for (int j = 0; j < TC; j++) {
            const int x = rx[j & 1], y = ry[j >> 1 & 1], z = rz[j >> 2 & 1];
            idx[j] = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(x, S));
            oob |= ((((unsigned)x >= (unsigned)S) | ((unsigned)y >= (unsigned)S) | ((unsigned)z >= (unsigned)S)) ? 1u : 0u) << j;
        }
        // corner lattice: three coordinates per axis (lattice_compat_oct / lattice_fpfh_oct)
        int cx[3], cy[3], cz[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            cx[a] = min(max(cell_of(X + sh.lat[a][0], T.xMin, T.scale), 0), S - 1);
            cy[a] = min(max(cell_of(Y + sh.lat[3 * a][1], T.yMin, T.scale), 0), S - 1);
            cz[a] = min(max(cell_of(Z + sh.lat[9 * a][2], T.zMin, T.scale), 0), S - 1);
        }
        // ---- all first-level gathers of the point
        float v[TC]; unsigned cd[TC];
        if (T.code) {
#pragma unroll
            for (int j = 0; j < TC; j++) cd[j] = T.code[idx[j]];
        } else {
#pragma unroll
            for (int j = 0; j < TC; j++) v[j] = T.dist[idx[j]];
        }
        unsigned mk[27]; int oc[27];
        if (T.use_compat) {
#pragma unroll
            for (int l = 0; l < 27; l++) mk[l] = T.cmask[(size_t)((cz[l / 9] * S + cy[(l / 3) % 3]) * S + cx[l % 3])];
        }
        if (T.use_fpfh) {
#pragma unroll
            for (int l = 0; l < 27; l++) oc[l] = T.occ[(size_t)((cz[l / 9] * S + cy[(l / 3) % 3]) * S + cx[l % 3])];
        }
        const unsigned cls = (unsigned)T.dcls[ic];
        // ---- second level: decoded distances, table entries
        if (T.code) {
#pragma unroll
            for (int j = 0; j < TC; j++) v[j] = dt_decode(T, idx[j], cd[j], nullptr);
        }
        float fe[27];
        if (T.use_fpfh) {
            const float* Erow = T.fpE + (size_t)ic * T.nOcc;
#pragma unroll
            for (int l = 0; l < 27; l++) fe[l] = Erow[max(oc[l], 0)];
        }
        if (__any((int)oob)) {
#pragma unroll
            for (int j = 0; j < TC; j++) {
                // (extension length from the cell coordinates already at hand -- dt_index would redo the three double conversions)
                if ((oob >> j) & 1u) v[j] = dt_finish(T, v[j], dt_ext_raw(rx[j & 1], ry[j >> 1 & 1], rz[j >> 2 & 1], S));
            }
        }
        // ---- addends of the two bounds (sweep_tile, parity mode, untrimmed)
        if (valid) {
#pragma unroll
            for (int j = 0; j < TC; j++) {
                float d = pw[m] * v[j];
                if (use_rho) d -= prho[m];
                if (d < 0.f) d = 0.f;
                const float e = d - sh.mtd[j];
                tub[j * Nd + i] = sq ? d * d : (lin ? d : 0.f);
                tlb[j * Nd + i] = e > 0.f ? (sq ? e * e : (lin ? e : 0.f)) : 0.f;
            }
        }
        if (T.use_compat) {
#pragma unroll
            for (int l = 0; l < 27; l++) wc[l] += __popcll(__ballot(valid && ((mk[l] >> cls) & 1u) == 0u));
        }
        if (T.use_fpfh && valid) {
#pragma unroll
            for (int l = 0; l < 27; l++) fprows[(size_t)l * Nd + i] = oc[l] < 0 ? 1000000000.f : fe[l];
        }
    }
    if (T.use_compat && lane == 0) {
#pragma unroll
        for (int l = 0; l < 27; l++) if (wc[l]) atomicAdd(&sh.cnt[l], wc[l]);
    }
    __syncthreads();
    // ---- the sequential float sums: 16 bound chains on wave 0, 27 c-FPFH chains on wave 1, side by side
    if (tid < 2 * TC) {
        const float* tt = rbuf + tid * Nd;          // lanes 0-7: ub rows, lanes 8-15: lb rows
        float acc = 0.f;
        int i = 0;
        for (; i + 16 <= Nd; i += 16) {
            float t16[16];
#pragma unroll
            for (int q = 0; q < 16; q++) t16[q] = tt[i + q];
#pragma unroll
            for (int q = 0; q < 16; q++) acc += t16[q];
        }
        for (; i < Nd; i++) acc += tt[i];
        if (tid < TC) sh.ub[tid] = acc; else sh.lb[tid - TC] = acc;
    } else if (T.use_fpfh && tid >= 64 && tid < 64 + 27) {
        const float* fr = fprows + (size_t)(tid - 64) * Nd;
        float acc = 0.f;
        int i = 0;
        for (; i + 16 <= Nd; i += 16) {
            float t16[16];
#pragma unroll
            for (int q = 0; q < 16; q++) t16[q] = fr[i + q];
#pragma unroll
            for (int q = 0; q < 16; q++) acc += t16[q];
        }
        for (; i < Nd; i++) acc += fr[i];
        sh.fpv[tid - 64] = acc / (float)Nd;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// binary heap with the element order of std::priority_queue<TRANSNODE> (reference jly_goicp.h:75-87):
// the hole is sifted to a leaf along the not-lower child, then the displaced value is sifted up.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool node_lower(const HeapNode& p, const HeapNode& q) { return p.lb != q.lb ? p.lb > q.lb : p.w < q.w; }

// node of the round-based search (k_fr_*): the queue entry plus the slot that holds its children's bounds once they are known
struct __attribute__((aligned(16))) HeapNodeX { float x, y, z, w, ub, lb; int exp; int pad; };

// storage: the first `hl` nodes live in LDS (the hot top of the heap), the rest in the job's global slab
// Node moves go through explicit LDS / global address spaces: a plain `i < hl ? lds[i] : glb[i]` makes the compiler merge the two
// pointers into a generic one and emit flat_load / flat_store, whose latency (and the double s_waitcnt they need) dominated the queue
// work (k_fr_select: 10.8 k cycles per pop before, see profiles/).
typedef unsigned qv4 __attribute__((ext_vector_type(4)));
typedef unsigned qv2 __attribute__((ext_vector_type(2)));
template <class N> struct NodeWords { typedef qv2 V; };
template <> struct NodeWords<HeapNodeX> { typedef qv4 V; };
template <class N>
struct HeapStoreT {
    N* lds; N* glb; int hl;
    typedef typename NodeWords<N>::V V;
    static constexpr int NV = sizeof(N) / sizeof(V);
    union Pun { N n; V v[NV]; __device__ Pun() {} };
    __device__ __forceinline__ N get(int i) const {
        Pun u;
        if (i < hl) {
            const __attribute__((address_space(3))) V* q = (const __attribute__((address_space(3))) V*)(lds + i);
#pragma unroll
            for (int k = 0; k < NV; k++) u.v[k] = q[k];
        } else {
            const __attribute__((address_space(1))) V* q = (const __attribute__((address_space(1))) V*)(glb + i);
#pragma unroll
            for (int k = 0; k < NV; k++) u.v[k] = q[k];
        }
        return u.n;
    }
    // the same for callers that know the slot is inside the LDS window (no second address space, no branch)
    template <bool LDSONLY> __device__ __forceinline__ N get_t(int i) const {
        if constexpr (LDSONLY) {
            Pun u;
            const __attribute__((address_space(3))) V* q = (const __attribute__((address_space(3))) V*)(lds + i);
#pragma unroll
            for (int k = 0; k < NV; k++) u.v[k] = q[k];
            return u.n;
        } else return get(i);
    }
    template <bool LDSONLY> __device__ __forceinline__ void set_t(int i, const N& v) const {
        if constexpr (LDSONLY) {
            Pun u; u.n = v;
            __attribute__((address_space(3))) V* q = (__attribute__((address_space(3))) V*)(lds + i);
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] = u.v[k];
        } else set(i, v);
    }
    __device__ __forceinline__ void set(int i, const N& v) const {
        Pun u; u.n = v;
        if (i < hl) {
            __attribute__((address_space(3))) V* q = (__attribute__((address_space(3))) V*)(lds + i);
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] = u.v[k];
        } else {
            __attribute__((address_space(1))) V* q = (__attribute__((address_space(1))) V*)(glb + i);
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] = u.v[k];
        }
    }
};
typedef HeapStoreT<HeapNode> HeapStore;
// (the reference's `lb != lb' ? lb > lb' : w < w'` as mask arithmetic -- the same truth table, NaNs included -- so that a wave-wide compare is three
// v_cmp and two scalar mask operations instead of a chain of selects)
__device__ __forceinline__ bool key_lower(float plb, float pw, float qlb, float qw) { return (plb > qlb) | ((plb == qlb) & (pw < qw)); }
struct HeapKey { float lb, w; };
template <class N>
__device__ __forceinline__ HeapKey heap_key(const HeapStoreT<N>& a, int i) {
    HeapKey k;
    if (i < a.hl) {
        const __attribute__((address_space(3))) float* q = (const __attribute__((address_space(3))) float*)(a.lds + i);
        k.w = q[3]; k.lb = q[5];
    } else {
        const __attribute__((address_space(1))) float* q = (const __attribute__((address_space(1))) float*)(a.glb + i);
        k.w = q[3]; k.lb = q[5];
    }
    return k;
}
template <bool LDSONLY, class N>
__device__ __forceinline__ HeapKey heap_key_t(const HeapStoreT<N>& a, int i) {
    if constexpr (LDSONLY) {
        HeapKey k;
        const __attribute__((address_space(3))) float* q = (const __attribute__((address_space(3))) float*)(a.lds + i);
        k.w = q[3]; k.lb = q[5];
        return k;
    } else return heap_key(a, i);
}
template <class N>
__device__ __forceinline__ void heap_up(const HeapStoreT<N>& a, int hole, const N& v) {
    int par = (hole - 1) / 2;
    while (hole > 0) {
        const HeapKey pk = heap_key(a, par);                      // compare on (lb, w) only; move whole nodes only when needed
        if (!key_lower(pk.lb, pk.w, v.lb, v.w)) break;
        a.set(hole, a.get(par)); hole = par; par = (hole - 1) / 2;
    }
    a.set(hole, v);
}
// heap_up for a whole wave (uniform hole and v): the ancestors of the hole are known up front, so lane k fetches ancestor k,
// one ballot tells how many of them v passes (the leading run of "ancestor is lower"), those ancestors drop one slot down
// the path in parallel and v lands above them -- the arrangement std::push_heap's loop produces, in one round trip to
// the queue instead of one per level.
template <bool LDSONLY, class N>
__device__ __forceinline__ void heap_up_wave_t(const HeapStoreT<N>& a, int hole, const N& v, int lane) {
    const int depth = 31 - __clz(hole + 1);                      // ancestors: ((hole+1) >> (k+1)) - 1, k = 0 .. depth-1
    const bool act = lane < depth;
    N nk = v; bool rises = false;
    if (act) { nk = a.template get_t<LDSONLY>(((hole + 1) >> (lane + 1)) - 1); rises = key_lower(nk.lb, nk.w, v.lb, v.w); }
    const unsigned long long up = __ballot(rises);
    const int m = min(depth, __ffsll((long long)~up) - 1);       // lanes >= depth never rise, so ~up is never 0
    wave_sync_lds();
    if (lane < m) a.template set_t<LDSONLY>(((hole + 1) >> lane) - 1, nk);      // ancestor k drops into the slot below it on the path
    if (lane == 63) a.template set_t<LDSONLY>(((hole + 1) >> m) - 1, v);
    wave_sync_lds();
}
template <class N>
__device__ __forceinline__ void heap_up_wave(const HeapStoreT<N>& a, int hole_, const N& v, int lane) {
    const int hole = __builtin_amdgcn_readfirstlane(hole_);      // (uniform by contract)
    if (hole < a.hl) heap_up_wave_t<true>(a, hole, v, lane);     // the whole path sits inside the LDS window: straight-line code
    else heap_up_wave_t<false>(a, hole, v, lane);
}
template <class N>
__device__ __forceinline__ void heap_pop(const HeapStoreT<N>& a, int& n) {
    if (n > 1) {
        const int len = n - 1;
        const N v = a.get(len);
        a.set(len, a.get(0));
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            const HeapKey k1 = heap_key(a, child), k0 = heap_key(a, child - 1);
            if (key_lower(k1.lb, k1.w, k0.lb, k0.w)) child--;
            a.set(hole, a.get(child)); hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a.set(hole, a.get(child - 1)); hole = child - 1; }
        heap_up(a, hole, v);
    }
    n--;
}

// heap_pop for a whole wave (uniform n). std::pop_heap sifts the hole to a leaf along the not-lower child WITHOUT looking at the displaced
// value (libstdc++ __adjust_heap), so the path is a function of sibling comparisons alone: lane l takes node l of the 63-node subtree under
// the hole and compares that node's two children (keys only), one ballot holds the 63 choices, and a scalar walk over the ballot's bits
// descends six levels per round trip to the queue. Lane k then fetches the node that moves up at level k (whole), one more ballot tells how
// far the displaced last element rises back along the same path (the trailing run of "path node is lower", as in heap_up_wave), and the
// stores go out together: path nodes above the landing slot move up one level, the value lands, nodes below it stay where they are.
// Same arrangement as heap_pop / std::pop_heap; two round trips for 12 levels instead of one per level.
template <bool LDSONLY, class N>
__device__ __forceinline__ void heap_pop_wave_t(const HeapStoreT<N>& a, const int len, int lane) {
    {
        const N v = a.template get_t<LDSONLY>(len);
        const N top = a.template get_t<LDSONLY>(0);
        const int half = (len - 1) / 2;                              // nodes below this index have two children
        const int sd = 31 - __clz(lane + 1), sj = lane + 1 - (1 << sd);   // this lane's node of a 63-node subtree: depth, offset in its level
        int hole = 0, lvl = 0, from = 0, to = 0;
        while (hole < half) {
            const int nidx = ((hole + 1) << sd) - 1 + sj;
            bool right = false;
            if (lane < 63 && nidx < half) {
                const HeapKey k1 = heap_key_t<LDSONLY>(a, 2 * nidx + 2), k0 = heap_key_t<LDSONLY>(a, 2 * nidx + 1);
                right = !key_lower(k1.lb, k1.w, k0.lb, k0.w);
            }
            const unsigned long long choice = __ballot(right);
            int rel = 0;
#pragma unroll
            for (int s = 0; s < 6; s++) {
                if (hole >= half) break;
                const int bit = (int)((choice >> rel) & 1ull);
                const int child = 2 * hole + 1 + bit;
                if (lane == lvl) { from = child; to = hole; }
                hole = child; rel = 2 * rel + 1 + bit; lvl++;
            }
        }
        if ((len & 1) == 0 && hole == (len - 2) / 2) {               // the last inner node has a single child
            const int child = 2 * hole + 1;
            if (lane == lvl) { from = child; to = hole; }
            hole = child; lvl++;
        }
        N mine = v; bool rises = false;
        if (lane < lvl) { mine = a.template get_t<LDSONLY>(from); rises = key_lower(mine.lb, mine.w, v.lb, v.w); }
        const unsigned long long stays = ~__ballot(rises) & ((1ull << lvl) - 1ull);    // (lvl <= 32: the queue holds < 2^31 nodes)
        const int m = stays ? 64 - __clzll((long long)stays) : 0;   // path nodes m .. lvl-1 let the value pass: it lands in slot path[m]
        const int land = m == lvl ? hole : __builtin_amdgcn_readlane(to, m);
        if (lane < m) a.template set_t<LDSONLY>(to, mine);
        if (lane == 63) a.template set_t<LDSONLY>(land, v);
        if (lane == 62) a.template set_t<LDSONLY>(len, top);
        wave_sync_lds();
    }
}
template <class N>
__device__ __forceinline__ void heap_pop_wave(const HeapStoreT<N>& a, int& n, int lane) {
    if (n > 1) {
        const int len = __builtin_amdgcn_readfirstlane(n) - 1;       // (uniform by contract: keeps the walk on the scalar unit)
        if (len < a.hl) heap_pop_wave_t<true>(a, len, lane);         // the whole queue sits inside the LDS window: straight-line code
        else heap_pop_wave_t<false>(a, len, lane);
    }
    n--;
}

// ---------------------------------------------------------------------------------------------
// sweep_stream: the tile sweep for clouds that do not fit a workgroup's registers (Nd > 8192; geometric error only, fp64 sums).
// The workgroup walks the rotated cloud in chunks of 1024 points (coalesced float4 reads), 8 gathers per lane and chunk.
// Untrimmed runs accumulate the two sums on the fly. Trimmed runs park the residuals in a per-workgroup row buffer in HBM
// (8 x Nd floats; 288 GB make that affordable) and find the exact inlier threshold with the same histogram select as
// sweep_tile: one pass over the rows for the 1024-bin window, one pass that sums everything below the threshold bin and
// collects the bin's members, in-wave bisection of the low 16 bits; the generic byte passes re-read the rows.
// ---------------------------------------------------------------------------------------------
// PARTIAL = true (untrimmed runs only): the workgroup sweeps the points [p0, p1) of the cloud and leaves its 16 fp64 partial sums
// (ub, lb per cube) in part_out instead of finishing the bounds -- one node's sweep split over several workgroups (k_fr_expand).
template <bool OCT, int BLOCK = 1024, bool PARTIAL = false>
__device__ __forceinline__ void sweep_stream(const DevTables& T, TileShm<BLOCK>& sh, const float4* __restrict__ rot, const float* __restrict__ rho_row,
                                             float* __restrict__ rows, int NdP, int p0 = 0, int p1 = 0, double* __restrict__ part_out = nullptr,
                                             const float* __restrict__ dual_rho = nullptr /* untrimmed, !PARTIAL: second call's radius row (see sweep_tile DUAL) */) {
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int Nd = PARTIAL ? p1 : T.Nd, S = T.S;
    const bool row_mode = !OCT && sh.ctl[2];
    const bool sq = T.norm == 2, lin = T.norm == 1;
    const bool trim = !PARTIAL && T.doTrim != 0;
    if (!PARTIAL && T.exact_stream) {
        // parity_mode 2 on a streamed cloud: the addend rows (untrimmed: 8 ub + 8 lb rows) or the residual rows (trimmed: 8 rows + 8 rows of
        // selection scratch) of the tile go to this workgroup's [2 * TC][NdP] rows in HBM; then the reference-order arithmetic of sweep_tile's
        // parity branches runs on them: one wave per cube for intro_select, 16 lanes for the 16 sequential float sums (jly_goicp.cpp:384-415).
        constexpr int PB = BLOCK / 8, U = 8;
        const int j = tid & 7, q = tid >> 3;
        const float cx = sh.cen[j][0], cy = sh.cen[j][1], cz = sh.cen[j][2], mtd = sh.mtd[j];
        for (int b = 0; b < Nd; b += PB * U) {
            unsigned idx[U]; float w[U], rh[U], ext[U]; bool valid[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int i = b + q + k * PB;
                valid[k] = i < Nd;
                const float4 p = rot[min(i, Nd - 1)];
                rh[k] = rho_row ? rho_row[min(i, Nd - 1)] : 0.f;
                const int x = cell_of(p.x + cx, T.xMin, T.scale), y = cell_of(p.y + cy, T.yMin, T.scale), z = cell_of(p.z + cz, T.zMin, T.scale);
                idx[k] = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(x, S));
                ext[k] = dt_ext_raw(x, y, z, S);
                w[k] = p.w;
            }
            float v[U];
            if (T.code) {
                unsigned c[U];
#pragma unroll
                for (int k = 0; k < U; k++) c[k] = T.code[idx[k]];
#pragma unroll
                for (int k = 0; k < U; k++) v[k] = dt_decode(T, idx[k], c[k], nullptr);
            } else {
#pragma unroll
                for (int k = 0; k < U; k++) v[k] = T.dist[idx[k]];
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                float d = w[k] * dt_finish(T, v[k], ext[k]);
                if (rho_row) d -= rh[k];
                if (d < 0.f) d = 0.f;
                const int i = b + q + k * PB;
                if (valid[k]) {
                    if (trim) rows[(size_t)j * NdP + i] = d;
                    else {
                        const float e = d - mtd;
                        rows[(size_t)j * NdP + i] = sq ? d * d : (lin ? d : 0.f);
                        rows[(size_t)(TC + j) * NdP + i] = e > 0.f ? (sq ? e * e : (lin ? e : 0.f)) : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        if (trim) {
            if (wv < TC) wave_intro_select<true>(rows + (size_t)wv * NdP, reinterpret_cast<int*>(rows + (size_t)(TC + wv) * NdP), Nd, T.inlierNum - 1, ln);
            __syncthreads();
        }
        if (tid < 2 * TC) {
            const int cj = tid & (TC - 1);
            const bool is_lb = tid >= TC;
            const float* rr = trim ? rows + (size_t)cj * NdP : rows + (size_t)tid * NdP;
            const float sub = (trim && is_lb) ? sh.mtd[cj] : 0.f;
            const int cnt = trim ? T.inlierNum : Nd;
            float acc = 0.f;
            auto add = [&](float d) {
                if (!trim) { acc += d; return; }                    // (addends formed above)
                const float e = d - sub;                            // (d - 0 == d exactly for the ub chain)
                float term = sq ? e * e : (lin ? e : 0.f);
                if (is_lb && !(e > 0.f)) term = 0.f;
                acc += term;
            };
            int i = 0;
            for (; i + 32 <= cnt; i += 32) {                        // 32 reads in flight, then the 32 dependent adds
                float t32[32];
#pragma unroll
                for (int k = 0; k < 32; k++) t32[k] = rr[i + k];
#pragma unroll
                for (int k = 0; k < 32; k++) add(t32[k]);
            }
            for (; i < cnt; i++) add(rr[i]);
            if (is_lb) sh.lb[cj] = acc; else sh.ub[cj] = acc;
        }
        __syncthreads();
        return;
    }
    if (!trim) {
        // Untrimmed runs keep nothing per point, so the sweep runs with lane = (point, cube): lane l adds up cube l & 7 over the points
        // (l >> 3) + BLOCK/8 * k. The 8 lanes of a point read cells of nearby grid rows / one row, i.e. 8-32 cache lines per wave instruction
        // instead of 64 unrelated ones -- the vector L1 works through the lines of an instruction one by one (tools/gather_modes.py: the same
        // lookups in 1.85 ms instead of 3.77 ms on the configs[2] tiles). Two fp64 accumulators per lane, 8 lookups in flight.
        constexpr int PB = BLOCK / 8, U = 8;
        const int j = tid & 7, q = tid >> 3;
        const float cx = sh.cen[j][0], cy = sh.cen[j][1], cz = sh.cen[j][2], mtd = sh.mtd[j];
        double su1 = 0.0, sl1 = 0.0, su2 = 0.0, sl2 = 0.0;        // (su2, sl2: the second call of a fused pair -- same lookups, radius dual_rho)
        for (int b = (PARTIAL ? p0 : 0); b < Nd; b += PB * U) {
            unsigned idx[U]; float w[U], rh[U], rh2[U], ext[U]; bool valid[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int i = b + q + k * PB;
                valid[k] = i < Nd;
                const float4 p = rot[min(i, Nd - 1)];
                rh[k] = rho_row ? rho_row[min(i, Nd - 1)] : 0.f;
                rh2[k] = dual_rho ? dual_rho[min(i, Nd - 1)] : 0.f;
                const int x = cell_of(p.x + cx, T.xMin, T.scale), y = cell_of(p.y + cy, T.yMin, T.scale), z = cell_of(p.z + cz, T.zMin, T.scale);
                idx[k] = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(x, S));
                ext[k] = dt_ext_raw(x, y, z, S);
                w[k] = p.w;
            }
            float v[U];
            if (T.code) {
                unsigned c[U];
#pragma unroll
                for (int k = 0; k < U; k++) c[k] = T.code[idx[k]];
#pragma unroll
                for (int k = 0; k < U; k++) v[k] = dt_decode(T, idx[k], c[k], nullptr);
            } else {
#pragma unroll
                for (int k = 0; k < U; k++) v[k] = T.dist[idx[k]];
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                float d = w[k] * dt_finish(T, v[k], ext[k]);
                if (rho_row) d -= rh[k];
                if (d < 0.f) d = 0.f;
                const float e = d - mtd;
                const float tu = sq ? d * d : (lin ? d : 0.f);
                const float tl = sq ? e * e : (lin ? e : 0.f);
                if (valid[k]) { su1 += (double)tu; if (e > 0.f) sl1 += (double)tl; }
                if (dual_rho) {                               // minDis -= maxRotDisL; clamp (reference jly_goicp.cpp:375-381) on the same product
                    float d2 = d - rh2[k];
                    if (d2 < 0.f) d2 = 0.f;
                    const float e2 = d2 - mtd;
                    const float tu2 = sq ? d2 * d2 : (lin ? d2 : 0.f);
                    const float tl2 = sq ? e2 * e2 : (lin ? e2 : 0.f);
                    if (valid[k]) { su2 += (double)tu2; if (e2 > 0.f) sl2 += (double)tl2; }
                }
            }
        }
        if (dual_rho) {
            // first call of the pair -> sh.ub2 / sh.lb2 (one more trip through the per-wave partials)
            su1 = cube_sum(su1); sl1 = cube_sum(sl1);
            if (ln < TC) { sh.red[wv][2 * ln] = su1; sh.red[wv][2 * ln + 1] = sl1; }
            __syncthreads();
            if (tid < TC) {
                double u = 0.0, l = 0.0;
                for (int w2 = 0; w2 < BLOCK / 64; w2++) { u += sh.red[w2][2 * tid]; l += sh.red[w2][2 * tid + 1]; }
                sh.ub2[tid] = (float)u; sh.lb2[tid] = (float)l;
            }
            __syncthreads();
            su1 = su2; sl1 = sl2;
        }
        su1 = cube_sum(su1); sl1 = cube_sum(sl1);
        if (ln < TC) { sh.red[wv][2 * ln] = su1; sh.red[wv][2 * ln + 1] = sl1; }
        __syncthreads();
        if (tid < TC) {
            double u = 0.0, l = 0.0;
            for (int w2 = 0; w2 < BLOCK / 64; w2++) { u += sh.red[w2][2 * tid]; l += sh.red[w2][2 * tid + 1]; }
            if (PARTIAL) { part_out[2 * tid] = u; part_out[2 * tid + 1] = l; }
            else { sh.ub[tid] = (float)u; sh.lb[tid] = (float)l; }
        }
        __syncthreads();
        return;
    }
    double su[TC], sl[TC]; unsigned mx[TC];
#pragma unroll
    for (int j = 0; j < TC; j++) { su[j] = 0.0; sl[j] = 0.0; mx[j] = 0u; }
    auto add_terms = [&](int j, float d) {           // the two addends of one residual (jly_goicp.cpp:393-415), into lane-local fp64
        const float e = d - sh.mtd[j];
        const float tu = sq ? d * d : (lin ? d : 0.f);
        const float tl = sq ? e * e : (lin ? e : 0.f);
        su[j] += (double)tu;
        if (e > 0.f) sl[j] += (double)tl;
    };
    // (only trimmed, order-free runs get here.) Pass 0, lane = (point, cube) as in the untrimmed sweep above: the residual rows go to HBM, every
    // lane keeps the largest residual of its cube (the window of the histogram pass that follows, which reads the rows back lane = point).
    {
        constexpr int PB = BLOCK / 8, U = 8;
        const int j = tid & 7, q = tid >> 3;
        const float cx = sh.cen[j][0], cy = sh.cen[j][1], cz = sh.cen[j][2];
        unsigned mx1 = 0u;
        for (int b = 0; b < Nd; b += PB * U) {
            unsigned idx[U]; float w[U], rh[U], ext[U]; bool valid[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int i = b + q + k * PB;
                valid[k] = i < Nd;
                const float4 p = rot[min(i, Nd - 1)];
                rh[k] = rho_row ? rho_row[min(i, Nd - 1)] : 0.f;
                const int x = cell_of(p.x + cx, T.xMin, T.scale), y = cell_of(p.y + cy, T.yMin, T.scale), z = cell_of(p.z + cz, T.zMin, T.scale);
                idx[k] = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(x, S));
                ext[k] = dt_ext_raw(x, y, z, S);
                w[k] = p.w;
            }
            float v[U];
            if (T.code) {
                unsigned c[U];
#pragma unroll
                for (int k = 0; k < U; k++) c[k] = T.code[idx[k]];
#pragma unroll
                for (int k = 0; k < U; k++) v[k] = dt_decode(T, idx[k], c[k], nullptr);
            } else {
#pragma unroll
                for (int k = 0; k < U; k++) v[k] = T.dist[idx[k]];
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                float d = w[k] * dt_finish(T, v[k], ext[k]);
                if (rho_row) d -= rh[k];
                if (d < 0.f) d = 0.f;
                if (valid[k]) { rows[(size_t)j * NdP + b + q + k * PB] = d; mx1 = max(mx1, __float_as_uint(d)); }
            }
        }
#pragma unroll
        for (int jj = 0; jj < TC; jj++) mx[jj] = jj == j ? mx1 : 0u;
    }
    unsigned thr[TC]; int below[TC];
#pragma unroll
    for (int j = 0; j < TC; j++) { thr[j] = 0xFFFFFFFFu; below[j] = 0; }
    if (trim) {
        for (int k = tid; k < TC * 1024; k += BLOCK) (&sh.hist[0][0])[k] = 0u;
        if (tid < TC) { sh.prefix[tid] = 0u; sh.rem[tid] = T.inlierNum - 1; sh.maxb[tid] = 0u; sh.zc[tid] = 0; sh.ncand[tid] = 0; }
        if (tid == 0) sh.fallback = 0;
        __syncthreads();                                   // also orders the row stores before the row loads below
#pragma unroll
        for (int j = 0; j < TC; j++) {
            unsigned m = mx[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_down((int)m, off, 64));
            if (ln == 0) atomicMax(&sh.maxb[j], m);
        }
        __syncthreads();
        // pass 1 over the rows: 1024-bin window below each cube's maximum, exact zeros counted with ballots
        {
            int base16[TC], z[TC];
#pragma unroll
            for (int j = 0; j < TC; j++) { base16[j] = max(0, (int)(sh.maxb[j] >> 16) - 1023); z[j] = 0; }
            for (int b = 0; b < Nd; b += BLOCK) {
                const int i = b + tid;
                const bool valid = i < Nd;
                unsigned bits[TC];
#pragma unroll
                for (int j = 0; j < TC; j++) bits[j] = valid ? __float_as_uint(rows[(size_t)j * NdP + i]) : 0u;
#pragma unroll
                for (int j = 0; j < TC; j++) {
                    z[j] += __popcll(__ballot(valid && bits[j] == 0u));
                    if (valid && bits[j] != 0u) atomicAdd(&sh.hist[j][max(0, (int)(bits[j] >> 16) - base16[j])], 1u);
                }
            }
#pragma unroll
            for (int j = 0; j < TC; j++) if (ln == 0 && z[j]) atomicAdd(&sh.zc[j], z[j]);
        }
        __syncthreads();
        for (int j = wv; j < TC; j += BLOCK / 64) {          // one wave scans the 1024 bins of a cube: 16 per lane
            unsigned c[16]; int mine = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) { c[q] = sh.hist[j][16 * ln + q]; mine += (int)c[q]; }
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (ln >= off) incl += t; }
            const int zc = sh.zc[j];
            const int rank = sh.rem[j];
            if (rank < zc) {                                  // the threshold is one of the zeros
                if (ln == 0) { sh.prefix[j] = 0u; sh.rem[j] = rank; sh.ncand[j] = -1; }
                continue;
            }
            const int rem = rank - zc;
            const unsigned long long over = __ballot(incl > rem);
            const int owner = over ? __ffsll((long long)over) - 1 : 63;
            if (ln == owner) {
                int rr = rem - (incl - mine);
                int bq = 0;
#pragma unroll
                for (int q = 0; q < 15; q++) { if (bq == q && rr >= (int)c[q]) { rr -= (int)c[q]; bq = q + 1; } }
                unsigned cb = c[0];
#pragma unroll
                for (int q = 1; q < 16; q++) if (bq == q) cb = c[q];
                const int bin = 16 * ln + bq;
                const int base16 = max(0, (int)(sh.maxb[j] >> 16) - 1023);
                if ((bin == 0 && base16 > 0) || cb > CAND_CAP) sh.fallback = 1;
                sh.prefix[j] = (unsigned)(base16 + bin) << 16;
                sh.rem[j] = rr;
            }
        }
        __syncthreads();
        const bool generic = sh.fallback != 0;
        if (!generic) {
            // pass 2 over the rows: everything below the threshold bin is summed, the bin's members are collected
            unsigned p16[TC]; bool known[TC];
#pragma unroll
            for (int j = 0; j < TC; j++) { p16[j] = sh.prefix[j] >> 16; known[j] = sh.ncand[j] < 0; }
            for (int b = 0; b < Nd; b += BLOCK) {
                const int i = b + tid;
                if (i < Nd) {
                    float d[TC];
#pragma unroll
                    for (int j = 0; j < TC; j++) d[j] = rows[(size_t)j * NdP + i];
#pragma unroll
                    for (int j = 0; j < TC; j++) {
                        const unsigned bits = __float_as_uint(d[j]);
                        if (bits == 0u) continue;               // zeros add nothing to either sum
                        const unsigned b16 = bits >> 16;
                        if (b16 < p16[j]) add_terms(j, d[j]);
                        else if (b16 == p16[j] && !known[j]) { const int pos = atomicAdd(&sh.ncand[j], 1); if (pos < CAND_CAP) sh.cand[j][pos] = bits; }
                    }
                }
            }
            __syncthreads();
            for (int j = wv; j < TC; j += BLOCK / 64) {
                const int nc = sh.ncand[j];
                if (nc < 0) continue;
                const bool v0 = ln < nc, v1 = ln + 64 < nc;
                const unsigned c0 = v0 ? sh.cand[j][ln] : 0u, c1 = v1 ? sh.cand[j][ln + 64] : 0u;
                const int rem = sh.rem[j];
                unsigned pre = sh.prefix[j];
#pragma unroll
                for (int bb = 15; bb >= 0; bb--) {
                    const unsigned cnd = pre | (1u << bb);
                    const int cnt = __popcll(__ballot(v0 && c0 < cnd)) + __popcll(__ballot(v1 && c1 < cnd));
                    if (cnt <= rem) pre = cnd;
                }
                const int less = __popcll(__ballot(v0 && c0 < pre)) + __popcll(__ballot(v1 && c1 < pre));
                if (ln == 0) { sh.prefix[j] = pre; sh.rem[j] = rem - less; }
                // the bin members below the threshold join this wave's partial sums of cube j
#pragma unroll
                for (int jj = 0; jj < TC; jj++) {
                    if (jj == j) {
                        if (v0 && c0 < pre) add_terms(jj, __uint_as_float(c0));
                        if (v1 && c1 < pre) add_terms(jj, __uint_as_float(c1));
                    }
                }
            }
            __syncthreads();
        } else {
            if (tid < TC) { sh.prefix[tid] = 0u; sh.rem[tid] = T.inlierNum - 1; }
            for (int pass = 3; pass >= 0; pass--) {
                __syncthreads();
                for (int k = tid; k < TC * 256; k += BLOCK) sh.hist[k >> 8][k & 255] = 0u;
                __syncthreads();
                const unsigned himask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
                unsigned pf[TC];
#pragma unroll
                for (int j = 0; j < TC; j++) pf[j] = sh.prefix[j];
                for (int b = 0; b < Nd; b += BLOCK) {
                    const int i = b + tid;
                    const bool valid = i < Nd;
#pragma unroll
                    for (int j = 0; j < TC; j++) {
                        const unsigned bits = valid ? __float_as_uint(rows[(size_t)j * NdP + i]) : 0u;
                        const bool take = valid && ((bits ^ pf[j]) & himask) == 0u;
                        hist_add(sh.hist[j], (bits >> (8 * pass)) & 255u, take, ln);
                    }
                }
                __syncthreads();
                for (int j = wv; j < TC; j += BLOCK / 64) {
                    const unsigned c0 = sh.hist[j][4 * ln], c1 = sh.hist[j][4 * ln + 1], c2 = sh.hist[j][4 * ln + 2], c3 = sh.hist[j][4 * ln + 3];
                    const int mine = (int)(c0 + c1 + c2 + c3);
                    int incl = mine;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (ln >= off) incl += t; }
                    const int rem = sh.rem[j];
                    const unsigned long long over = __ballot(incl > rem);
                    const int owner = over ? __ffsll((long long)over) - 1 : 63;
                    if (ln == owner) {
                        int rr = rem - (incl - mine);
                        unsigned dgt = 4u * ln;
                        if (rr >= (int)c0) { rr -= (int)c0; dgt++; if (rr >= (int)c1) { rr -= (int)c1; dgt++; if (rr >= (int)c2) { rr -= (int)c2; dgt++; } } }
                        sh.prefix[j] |= dgt << (8 * pass);
                        sh.rem[j] = rr;
                    }
                }
            }
            __syncthreads();
            unsigned tj[TC];
#pragma unroll
            for (int j = 0; j < TC; j++) tj[j] = sh.prefix[j];
            for (int b = 0; b < Nd; b += BLOCK) {
                const int i = b + tid;
                if (i < Nd) {
#pragma unroll
                    for (int j = 0; j < TC; j++) { const float d = rows[(size_t)j * NdP + i]; if (__float_as_uint(d) < tj[j]) add_terms(j, d); }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TC; j++) { thr[j] = sh.prefix[j]; below[j] = T.inlierNum - 1 - sh.rem[j]; }
    }
#pragma unroll
    for (int j = 0; j < TC; j++) {
        const double u = wave_sum(su[j]), l = wave_sum(sl[j]);
        if (ln == 0) { sh.red[wv][2 * j] = u; sh.red[wv][2 * j + 1] = l; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TC; j++) {
        if (tid == j) {
            double u = 0.0, l = 0.0;
            for (int w = 0; w < BLOCK / 64; w++) { u += sh.red[w][2 * j]; l += sh.red[w][2 * j + 1]; }
            if (PARTIAL) { part_out[2 * j] = u; part_out[2 * j + 1] = l; }
            else {
                if (trim) {
                    const float d = __uint_as_float(thr[j]);
                    const float e = d - sh.mtd[j];
                    const double n = (double)(T.inlierNum - below[j]);
                    if (sq) { u += n * (double)(d * d); if (e > 0.f) l += n * (double)(e * e); }
                    else if (lin) { u += n * (double)d; if (e > 0.f) l += n * (double)e; }
                }
                sh.ub[j] = (float)u; sh.lb[j] = (float)l;
            }
        }
    }
    __syncthreads();
}


// ---- records of the round-based search (k_fr.hip), shared with the engine handle ----
struct __attribute__((aligned(16))) FrExp { float ub[TC], lb[TC]; int child_base; int pad[3]; };   // child_base: first of the 8 records of the children's own children, or -1
struct FrItem { int job, eid, split, pad; float x, y, z, w; };
struct FrState {
    float optErrorT, bx, by, bz, bw;
    int found, status, done;
    unsigned pops, cubes, spec;
    int hn, ne, pend0, pad0, pad1;
};
struct FrCtx {
    const goicp_bnb_job* jobs; FrState* st; goicp_bnb_result* results;
    HeapNodeX* heaps; int heap_cap;
    FrExp* exps; int exp_cap;
    FrItem* wl; int wl_cap; unsigned* wl_count;      // wl_count[round], zeroed per batch of rounds
    unsigned* active;                                  // active[round] = searches still running after that round's select
    double* parts; int nsplit;                         // [job][eid][split][16] partial sums (streamed untrimmed clouds), else null / 1
    int ngroups;                                       // work items per node and point slice: 1 (all 8 children) or 2 (4 children each)
    float4* srot; int NdP;                             // rotated clouds [job][NdP]
    unsigned max_pops;
    unsigned long long* dbg;                           // [8] cycle sums of the busiest-looking phases of k_fr_select (GOICP_DBG & 32), else null
};
#define FR_LDS_NODES 4095          // queue nodes staged in LDS by k_fr_select (32 B each: 128 KB)
#define FR_SELECT_THREADS 256

