// engine.hip -- MI355X (gfx950 / CDNA4) Go-ICP inner-BnB error-evaluation engine.
//
// Hand-written HIP kernels for the hot path named in BASELINE.json:north_star; C ABI in
// include/goicp_engine.h (which cites the reference seam each entry point replaces).
// Compile with -ffp-contract=off: the reference arithmetic has no FMA (SURVEY.md 8(a')).
//
// Kernels
//   k_inner_bnb   one workgroup = one whole InnerBnB call (reference jly_goicp.cpp:286-579): the data
//                 cloud is rotated in-kernel and lives in registers (PPT points per lane), each queue pop
//                 sweeps the 8 child cubes (8*Nd DT gathers), finds the inlier threshold with an LDS histogram select,
//                 evaluates the 27 shared corner-lattice penalty terms, and lane 0 replays the reference's
//                 prune/push decisions on a binary heap that reproduces std::priority_queue's order.
//   k_eval_cubes  the same sweep for an arbitrary list of cubes (8 cubes per workgroup), for callers that
//                 drive the queue themselves and for the cube-evals/s benchmark.
//   k_rotate, k_corner_terms, k_point_eval, k_icp_nn, k_fpfh_table, k_build_cmask, k_neighbor_counts   helpers (see goicp_engine.h).
//
// Data layout in HBM (staged once per engine): DT distance grid float[S^3] x-fastest; compat mask
// uint16[S^3] (bit c set <=> a data point of colour class c is compatible with the nearest occupied cell of
// that cell -- jly_goicp.cpp:974-1000,1068-1092 folded into one gather); nearest-occupied compact id
// int32[S^3] + CSR of model points per occupied cell + c-FPFH rows (only when regularizationFPFH>0).
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
    int Nd, Nm, inlierNum, doTrim, norm, b0, b1;
    float reg, regFPFH;
    int use_compat, use_fpfh, exact;
    int exact_big;           // parity_mode 2, 1025..5120 points, geometric error: reference-order sums / intro_select on residual rows in HBM (xrows)
    int exact_stream;        // parity_mode 2 on a streamed cloud: the same on [2 * TC][NdP] rows per resident workgroup (srows)
    float* xrows; int xslots; // [xslots][2 * TC * NdP] floats: addend / residual rows + selection scratch of one workgroup each
    float4* srot;            // streaming path (Nd > 8192): per-workgroup rotated cloud [NdP]
    float* srows;            // streaming path, trimmed runs: per-workgroup residual rows [TC][NdP]
    int NdP;                 // row pitch of the two above
    int dbg;                 // GOICP_DBG ablation bits (diagnosis only; results are wrong when set)
    unsigned long long* dbg_times;   // [8] per-phase cycle sums of job 0, lane 0 (GOICP_DBG & 32)
};

struct __attribute__((aligned(8))) HeapNode { float x, y, z, w, ub, lb; };

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

// E[i][o] = min_p sum_b |f_i[b] - g_p[b]| over the model points p of occupied cell o, bins b0..b1-1 added in order (float)
__global__ void k_fpfh_table(DevTables T, float* __restrict__ E) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (size_t)T.Nd * T.nOcc) return;
    const int i = (int)(k / T.nOcc), o = (int)(k % T.nOcc);
    const float* f = T.dfpfh + (size_t)GOICP_FPFH_BINS * i;
    float minD = 1000000000.f;
    for (int p = T.occ_ptr[o]; p < T.occ_ptr[o + 1]; p++) {
        const float* g = T.mfpfh + (size_t)GOICP_FPFH_BINS * T.occ_pts[p];
        float diff = 0.f;
        for (int b = T.b0; b < T.b1; b++) diff += fabsf(f[b] - g[b]);
        if (diff < minD) minD = diff;
    }
    E[k] = minD;
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
                                                  bool regsel = false /* experimental (GOICP_DBG bit 256): ranges of <= 64 elements finish in registers, one element per lane (tests/csrc/partition_check.c, reg()) */) {
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
template <int BLOCK, int PPT, bool OCT, int NT, bool JIT = false, bool DUAL = false, bool XBIG = false /* JIT kernels: the parity_mode 2 instantiation (rows in HBM) */>
__device__ __forceinline__ void sweep_tile(const DevTables& T, TileShm<BLOCK>& sh, const float (&px)[PPT], const float (&py)[PPT],
                                           const float (&pz)[PPT], const float (&pw)[PPT], const float (&prho)[PPT], bool use_rho,
                                           float* rbuf /* LDS, exact mode only */, bool row_mode = false /* !OCT: the cubes share y and z */,
                                           const float4* __restrict__ jit_rot = nullptr, const float* __restrict__ jit_rho = nullptr,
                                           const float* lut_lds = nullptr /* LDS copy of T.lut[0, LUT_LDS) or null */,
                                           const float* __restrict__ dual_rho = nullptr) {
    const int tid = threadIdx.x;
    const int Nd = T.Nd;
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
    const bool stamp = (T.dbg & 64) && T.dbg_times && tid == 0 && (blockIdx.x & 63) == 0;
    unsigned long long tl_ = stamp ? clock64() : 0ull;
#define TSTAMP(k) do { if (stamp) { const unsigned long long now_ = clock64(); atomicAdd(&T.dbg_times[k], now_ - tl_); tl_ = now_; } } while (0)
#else
    constexpr bool stamp = false;
#define TSTAMP(k) do { } while (0)
#endif
    float4 pnext = make_float4(0.f, 0.f, 0.f, 0.f); float rnext = 0.f;
    if (JIT) { const int i0 = min(tid, Nd - 1); pnext = jit_rot[i0]; rnext = jit_rho ? jit_rho[i0] : 0.f; }
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        float X, Y, Z, W, RH;
        if (JIT) {
            X = pnext.x; Y = pnext.y; Z = pnext.z; W = pnext.w; RH = rnext;
            if (m + 1 < PPT) { const int in = min(tid + (m + 1) * BLOCK, Nd - 1); pnext = jit_rot[in]; rnext = jit_rho ? jit_rho[in] : 0.f; }
        } else { X = px[m]; Y = py[m]; Z = pz[m]; W = pw[m]; RH = prho[m]; }
        unsigned idx[NT]; float v[NT];
        const unsigned oob = cells(X, Y, Z, idx);
        if (T.code) {
            // coded grid: 2-byte gathers (half the cache lines of the float grid), then the value from the LUT
            unsigned c[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) c[j] = T.code[(T.dbg & 2) ? (idx[j] & 1023u) : idx[j]];
#pragma unroll
            for (int j = 0; j < NT; j++) v[j] = dt_decode(T, idx[j], c[j], lut_lds);
        } else {
#pragma unroll
            for (int j = 0; j < NT; j++) v[j] = T.dist[(T.dbg & 2) ? (idx[j] & 1023u) : idx[j]];
        }
        if (__any((int)oob)) {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                if ((oob >> j) & 1u) {
                    float ext; (void)dt_index(T, X + sh.cen[j][0], Y + sh.cen[j][1], Z + sh.cen[j][2], ext);
                    v[j] = dt_finish(T, v[j], ext);
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
        if (JIT) __builtin_amdgcn_sched_barrier(0);        // keep the point fetches one ahead instead of all up front
    }
    TSTAMP(0);
    // everything behind the gathers (selection + sums -> sh.ub / sh.lb) as one block, so that the fused form can run it twice on the same
    // distance values: first for the call without rotation uncertainty, then again after subtracting the radius of the other call
    auto finish = [&]() __attribute__((always_inline)) {      // (out of line the residual registers would go through scratch memory)
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
            wave_intro_select<JIT>(rbuf + j * Nd, reinterpret_cast<int*>(rbuf + NT * Nd) + j * Nd, Nd, T.inlierNum - 1, tid & 63, (T.dbg & 32) && T.dbg_times ? T.dbg_times + 8 : nullptr, (T.dbg & 256) != 0);      // (JIT: rows in HBM)
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
    if (T.doTrim && !(T.dbg & 1)) {
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
            if (stamp) atomicAdd(&T.dbg_times[7], 1ull);
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
    finish();
    if constexpr (DUAL) {
        // (two call sites rather than a loop: inside a loop the compiler hoists the block's lane-constant addresses and spills the residuals)
        if (tid < NT) { sh.ub2[tid] = sh.ub[tid]; sh.lb2[tid] = sh.lb[tid]; }        // (ordered by the barriers of the next finish())
        // second bound of the same cubes: minDis -= maxRotDisL; clamp (reference jly_goicp.cpp:375-381) on the products weights*DT kept in r
        int tid2 = tid;
        asm volatile("" : "+v"(tid2));                     // (keeps the radius fetches below the first finish(): hoisted above it they cost PPT registers)
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const float rh = dual_rho[min(tid2 + m * BLOCK, Nd - 1)];
#pragma unroll
            for (int j = 0; j < NT; j++) { float d = r[j][m] - rh; if (d < 0.f) d = 0.f; r[j][m] = d; }
        }
        finish();
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
__device__ __forceinline__ bool key_lower(float plb, float pw, float qlb, float qw) { return plb != qlb ? plb > qlb : pw < qw; }
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
template <class N>
__device__ __forceinline__ void heap_up_wave(const HeapStoreT<N>& a, int hole, const N& v, int lane) {
    const int depth = 31 - __clz(hole + 1);                      // ancestors: ((hole+1) >> (k+1)) - 1, k = 0 .. depth-1
    const bool act = lane < depth;
    N nk = v; bool rises = false;
    if (act) { nk = a.get(((hole + 1) >> (lane + 1)) - 1); rises = key_lower(nk.lb, nk.w, v.lb, v.w); }
    const unsigned long long up = __ballot(rises);
    const int m = min(depth, __ffsll((long long)~up) - 1);       // lanes >= depth never rise, so ~up is never 0
    wave_sync_lds();
    if (lane < m) a.set(((hole + 1) >> lane) - 1, nk);            // ancestor k drops into the slot below it on the path
    if (lane == 0) a.set(((hole + 1) >> m) - 1, v);
    wave_sync_lds();
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

// heap_pop for a whole wave (uniform n): every lane walks the same path with broadcast reads, both children of a level are
// fetched whole in one round trip (the chosen one moves up without a second read), lane 0 does the stores, and the displaced
// last element rises with heap_up_wave. Same arrangement as heap_pop / std::pop_heap.
template <class N>
__device__ __forceinline__ void heap_pop_wave(const HeapStoreT<N>& a, int& n, int lane) {
    if (n > 1) {
        const int len = n - 1;
        const N v = a.get(len);
        const N top = a.get(0);
        // the walk down only READS (the next level's children do not depend on this level's move); lane k remembers the node that
        // moves up at level k and all moves are stored together afterwards -- one LDS round trip per level instead of a read-write pair
        N mine = v; int mine_to = -1;
        int hole = 0, child = 0, lvl = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            const N c1 = a.get(child), c0 = a.get(child - 1);
            const bool left = key_lower(c1.lb, c1.w, c0.lb, c0.w);
            if (left) child--;
            if (lane == lvl) { mine = left ? c0 : c1; mine_to = hole; }
            hole = child; lvl++;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            const N c0 = a.get(child - 1);
            if (lane == lvl) { mine = c0; mine_to = hole; }
            hole = child - 1; lvl++;
        }
        if (mine_to >= 0) a.set(mine_to, mine);
        if (lane == 63) a.set(len, top);
        wave_sync_lds();
        heap_up_wave(a, hole, v, lane);
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

// ---------------------------------------------------------------------------------------------
// k_inner_bnb
// ---------------------------------------------------------------------------------------------
// PEN = false compiles the penalty-term code (27-point lattice gathers) out: a purely geometric search then keeps its
// cloud and residuals in registers without spilling.
// STREAM = true (BLOCK 1024, PPT 1, PEN false): clouds above 8192 points; the rotated cloud lives in the workgroup's slice of
// T.srot and every pop streams it through sweep_stream instead of holding it in registers.
// (256-lane geometries: at most 256 registers per lane, so that two searches share a CU)
// JIT = true (BLOCK 512, PPT 4, PEN false): geometric searches of 1025..2048 points; the rotated cloud also lives in T.srot, but the
// residuals stay in registers (sweep_tile fetches point m just before its gathers) -> 128 VGPRs, two searches per CU.
template <int BLOCK, int PPT, bool PEN, bool STREAM = false, bool JIT = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == 256 ? 2 : (JIT ? 4 : 1))) void k_inner_bnb(DevTables T, const goicp_bnb_job* __restrict__ jobs, goicp_bnb_result* __restrict__ results,
                                                      HeapNode* heaps, int heap_cap, float* fpscratch, unsigned max_pops, int heap_lds_nodes, int rbuf_floats, int fp_lds) {
    __shared__ TileShm<BLOCK> sh;
    __shared__ HeapNode parent;
    extern __shared__ __attribute__((aligned(16))) float rbuf[];   // [rbuf_floats] parity-mode residual rows, then the LDS part of the heap
    const int tid = threadIdx.x;
    const int Nd = T.Nd;
    const goicp_bnb_job job = jobs[blockIdx.x];
    HeapStore heap;
    heap.glb = heaps + (size_t)blockIdx.x * heap_cap; heap.lds = reinterpret_cast<HeapNode*>(rbuf + rbuf_floats); heap.hl = heap_lds_nodes;
    float* fpbuf = fpscratch ? fpscratch + (size_t)blockIdx.x * NLMAX * Nd : nullptr;
    if (fp_lds == 1) fpbuf = rbuf + rbuf_floats + (size_t)heap_lds_nodes * (sizeof(HeapNode) / sizeof(float));     // c-FPFH rows behind the LDS heap
    if (fp_lds == 2) fpbuf = reinterpret_cast<float*>(&sh.hist[0][0]);      // parity mode never touches the select histograms: the rows live there

    // K5 fused: pDataTemp = R*p, float, left to right (reference jly_goicp.cpp:750-756)
    float px[PPT], py[PPT], pz[PPT], pw[PPT], prho[PPT];
    const bool use_rho = job.rot_level >= 0;
    float4* srot = nullptr; float* srows = nullptr; const float* srho = nullptr;
    if constexpr (STREAM || JIT) {
        srot = T.srot + (size_t)blockIdx.x * T.NdP;
        srows = T.srows ? T.srows + (size_t)blockIdx.x * (T.exact_stream ? 2 : 1) * TC * T.NdP : nullptr;
        srho = use_rho ? T.maxRotDis + (size_t)job.rot_level * Nd : nullptr;
        for (int i = tid; i < Nd; i += BLOCK) {
            const float x = T.dxyz[3 * i], y = T.dxyz[3 * i + 1], z = T.dxyz[3 * i + 2];
            float4 p;
            p.x = job.R[0] * x + job.R[1] * y + job.R[2] * z;
            p.y = job.R[3] * x + job.R[4] * y + job.R[5] * z;
            p.z = job.R[6] * x + job.R[7] * y + job.R[8] * z;
            p.w = T.weights[i];
            srot[i] = p;
        }
#pragma unroll
        for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
        __syncthreads();
    } else {
        float x[PPT], y[PPT], z[PPT], rh[PPT];
        const float* rho_row = T.maxRotDis + (size_t)(use_rho ? job.rot_level : 0) * Nd;
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const int i = min(tid + m * BLOCK, Nd - 1);
            x[m] = T.dxyz[3 * i]; y[m] = T.dxyz[3 * i + 1]; z[m] = T.dxyz[3 * i + 2]; pw[m] = T.weights[i]; rh[m] = rho_row[i];
        }
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            px[m] = job.R[0] * x[m] + job.R[1] * y[m] + job.R[2] * z[m];
            py[m] = job.R[3] * x[m] + job.R[4] * y[m] + job.R[5] * z[m];
            pz[m] = job.R[6] * x[m] + job.R[7] * y[m] + job.R[8] * z[m];
            prho[m] = use_rho ? rh[m] : 0.f;
        }
    }

    // lane-0 state
    float optErrorT = job.optError_in;
    float bx = 0.f, by = 0.f, bz = 0.f, bw = 0.f;
    int found = 0, hn = 0, status = GOICP_OK;
    unsigned pops = 0, cubes = 0;
    if (tid == 0) {
        HeapNode root; root.x = job.init[0]; root.y = job.init[1]; root.z = job.init[2]; root.w = job.init[3]; root.ub = 0.f; root.lb = job.init_lb;
        heap.set(0, root);
    }
    hn = 1;
    const bool penalties = PEN && (T.use_compat || T.use_fpfh);
    const bool timing = (T.dbg & 32) && T.dbg_times && blockIdx.x == 0 && tid == 0;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = timing ? clock64() : 0ull;
#define STAMP(k) do { if (timing) { const unsigned long long now_ = clock64(); tph[k] += now_ - tlast; tlast = now_; } } while (0)

    for (;;) {
        if (tid < 64) {
            // wave 0, uniform control flow (every lane carries the same search state): pop with the wave-wide walk of heap_pop_wave
            int done = 0;
            if (hn == 0) done = 1;
            else {
                const HeapNode par = heap.get(0);
                heap_pop_wave(heap, hn, tid);
                pops++;
                if (optErrorT - par.lb < job.SSEThresh) done = 1;             // reference jly_goicp.cpp:317
                else if (pops > (job.max_pops ? job.max_pops : max_pops)) { done = 1; status = GOICP_ERR_LIMIT; }
                else if (hn + 8 > heap_cap) { done = 1; status = GOICP_ERR_OVERFLOW; }
                if (tid == 0) parent = par;
            }
            if (tid == 0) sh.ctl[0] = done;
        }
        STAMP(0);
        __syncthreads();
        if (sh.ctl[0]) break;
        // children (reference jly_goicp.cpp:322-333) and the 3x3x3 lattice of their corners (:436-437,486-488)
        if (tid < TC) {
            const int j = tid;
            const float w = parent.w / 2;
            const float x = parent.x + (j & 1) * w, y = parent.y + (j >> 1 & 1) * w, z = parent.z + (j >> 2 & 1) * w;
            sh.cube[j][0] = x; sh.cube[j][1] = y; sh.cube[j][2] = z; sh.cube[j][3] = w;
            sh.cen[j][0] = x + w / 2; sh.cen[j][1] = y + w / 2; sh.cen[j][2] = z + w / 2;
            sh.mtd[j] = (float)(SQRT3_D / 2.0 * w);
        }
        if (penalties && tid >= 64 && tid < 64 + 27) {
            const int l = tid - 64, ax = l % 3, ay = (l / 3) % 3, az = l / 9;
            sh.cnt[l] = 0;                                                      // (pop_eval_fused adds into it behind the next barrier)
            const float w = parent.w / 2;
            // coordinate k of the lattice along one axis: p, p+w, (p+w)+w  == child origin + corner offset
            const float x1 = parent.x + w, y1 = parent.y + w, z1 = parent.z + w;
            sh.lat[l][0] = ax == 0 ? parent.x : (ax == 1 ? x1 : x1 + w);
            sh.lat[l][1] = ay == 0 ? parent.y : (ay == 1 ? y1 : y1 + w);
            sh.lat[l][2] = az == 0 ? parent.z : (az == 1 ? z1 : z1 + w);
        }
        __syncthreads();
        STAMP(1);
        bool fused_pop = false;
        if constexpr (PEN && BLOCK == 256 && !STREAM && !JIT) fused_pop = penalties && T.exact && !T.doTrim && !(T.dbg & 128);
        if constexpr (STREAM) sweep_stream<true>(T, sh, srot, srho, srows, T.NdP);
        else if constexpr (JIT) sweep_tile<BLOCK, PPT, true, TC, true>(T, sh, px, py, pz, pw, prho, use_rho, rbuf, false, srot, srho);
        else if constexpr (PEN && BLOCK == 256) {
            // cavity-sized clouds, parity mode, untrimmed, penalty terms: one gather phase + one sequential-sum phase for all three evaluations
            if (fused_pop) pop_eval_fused<BLOCK, PPT>(T, sh, px, py, pz, pw, prho, use_rho, rbuf, fpbuf);
            else sweep_tile<BLOCK, PPT, true, TC>(T, sh, px, py, pz, pw, prho, use_rho, rbuf);
        } else sweep_tile<BLOCK, PPT, true, TC>(T, sh, px, py, pz, pw, prho, use_rho, rbuf);
        STAMP(2);
        if (PEN && T.use_compat && !fused_pop) lattice_compat_oct<BLOCK, PPT>(T, sh, px, py, pz);
        STAMP(3);
        if (PEN && T.use_fpfh && !fused_pop) lattice_fpfh_oct<BLOCK, PPT>(T, sh, px, py, pz, fpbuf);
        STAMP(4);
        if (penalties && tid < 64) {
            // lanes 0-7 of wave 0 fold the corner terms of their child into sh.ub/sh.lb in parallel (only wave 0 reads them next,
            // so a wave-level fence replaces a workgroup barrier) ...
            if (tid < TC) {
                const int j = tid;
                float ub = sh.ub[j], lb = sh.lb[j];
                // extremes over the cube's 8 corners (reference jly_goicp.cpp:431-549)
                int minI = 0, maxI = 0; float minF = 0.f, maxF = 0.f;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int l = ((j >> 2 & 1) + (c >> 2 & 1)) * 9 + ((j >> 1 & 1) + (c >> 1 & 1)) * 3 + ((j & 1) + (c & 1));
                    if (T.use_compat) { int v = sh.cnt[l]; if (c == 0) { minI = maxI = v; } else { if (v > maxI) maxI = v; if (v < minI) minI = v; } }
                    if (T.use_fpfh) { int v = (int)sh.fpv[l]; if (c == 0) { minF = maxF = (float)v; } else { if (v > maxF) maxF = (float)v; if (v < minF) minF = (float)v; } }
                }
                if (T.use_compat) { ub += T.reg * (float)(maxI * maxI); lb += T.reg * (float)(minI * minI); }
                if (T.use_fpfh) { ub += T.regFPFH * (maxF * maxF); lb += T.regFPFH * (minF * minF); }
                sh.ub[j] = ub; sh.lb[j] = lb;
            }
            wave_sync_lds();
        }
        if (tid < 64) {
            // ... and wave 0 replays the order-dependent part with uniform control flow (every lane carries the same search
            // state; lane 0's copy is the one reported): incumbent update, prune, push (reference jly_goicp.cpp:554-575)
            cubes += 8;
            for (int j = 0; j < TC; j++) {
                const float ub = sh.ub[j], lb = sh.lb[j];
                if (ub < optErrorT) { optErrorT = ub; bx = sh.cube[j][0]; by = sh.cube[j][1]; bz = sh.cube[j][2]; bw = sh.cube[j][3]; found = 1; }
                if (lb >= optErrorT) continue;                                  // reference jly_goicp.cpp:568
                HeapNode nd; nd.x = sh.cube[j][0]; nd.y = sh.cube[j][1]; nd.z = sh.cube[j][2]; nd.w = sh.cube[j][3]; nd.ub = ub; nd.lb = lb;
                hn++;
                heap_up_wave(heap, hn - 1, nd, tid);
            }
        }
        STAMP(5);
        // (the barrier at the top of the next iteration orders lane 0's heap work before anyone reads `parent`)
    }
    if (timing) { for (int k = 0; k < 8; k++) T.dbg_times[k] += tph[k]; T.dbg_times[7] += pops; }
#undef STAMP
    if (tid == 0) {
        goicp_bnb_result r;
        r.optErrorT = optErrorT; r.node[0] = bx; r.node[1] = by; r.node[2] = bz; r.node[3] = bw;
        r.found = found; r.status = status; r.pops = pops; r.cube_evals = cubes;
        results[blockIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// Round-based InnerBnB for large clouds (k_fr_init / k_fr_select / k_fr_expand): one search uses MANY workgroups.
//
// k_inner_bnb gives a search one workgroup; above cavity size a queue pop then costs tens of microseconds (8 x Nd gathers + the
// inlier selection on one CU) and a registration waits for its longest chain of pops while most of the chip idles. The bounds of a
// cube are pure functions of (rotation, level, cube) (reference jly_goicp.cpp:325-415), so the children of SEVERAL queue nodes can
// be evaluated at once and the reference's pop / prune / push sequence replayed afterwards from the stored values:
//   k_fr_select  one wave per search. Replays jly_goicp.cpp:303-575 on the search's queue for as long as the node on top already has
//                its children's bounds (same heap arrangement, same tie order, same stop test as k_inner_bnb), then marks up to K
//                nodes near the top of the queue that lack them and appends them to the launch-wide work list.
//   k_fr_expand  persistent workgroups drain the work list: one item = the 8 children of one node (one sweep tile over the search's
//                rotated cloud), or, for streamed untrimmed clouds, one slice of the points of such a tile (fp64 partial sums that the
//                next k_fr_select adds up in slice order).
// Rounds are separated by kernel boundaries (no in-kernel cross-workgroup hand-off); the host enqueues batches of rounds and looks
// at the count of unfinished searches between batches. Results are those of the sequential search: speculation only decides WHEN
// a node's children are evaluated, never which nodes the replay pops.
// ---------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void k_fr_init(DevTables T, FrCtx C) {
    const int job = blockIdx.x, tid = threadIdx.x;
    const goicp_bnb_job jb = C.jobs[job];
    float4* srot = C.srot + (size_t)job * C.NdP;
    for (int i = tid; i < T.Nd; i += 256) {            // K5: pDataTemp = R*p, float, left to right (reference jly_goicp.cpp:750-756)
        const float x = T.dxyz[3 * i], y = T.dxyz[3 * i + 1], z = T.dxyz[3 * i + 2];
        float4 p;
        p.x = jb.R[0] * x + jb.R[1] * y + jb.R[2] * z;
        p.y = jb.R[3] * x + jb.R[4] * y + jb.R[5] * z;
        p.z = jb.R[6] * x + jb.R[7] * y + jb.R[8] * z;
        p.w = T.weights[i];
        srot[i] = p;
    }
    if (tid == 0) {
        HeapNodeX root; root.x = jb.init[0]; root.y = jb.init[1]; root.z = jb.init[2]; root.w = jb.init[3]; root.ub = 0.f; root.lb = jb.init_lb; root.exp = -1; root.pad = 0;
        C.heaps[(size_t)job * C.heap_cap] = root;
        FrState st; st.optErrorT = jb.optError_in; st.bx = st.by = st.bz = st.bw = 0.f; st.found = 0; st.status = GOICP_OK; st.done = 0;
        st.pops = st.cubes = st.spec = 0u; st.hn = 1; st.ne = 0; st.pend0 = 0; st.pad0 = st.pad1 = 0;
        C.st[job] = st;
    }
}

__global__ __launch_bounds__(FR_SELECT_THREADS) void k_fr_select(FrCtx C, int round, int K, int depth) {
    __shared__ HeapNodeX top[FR_LDS_NODES];
    __shared__ FrExp ecache[64];
    __shared__ int s_hi;
    const int job = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool w0 = tid < 64;                       // wave 0 replays and selects; the other waves only move the queue in and out of LDS
    FrState st = C.st[job];
    if (st.done) return;
    const bool timing = C.dbg != nullptr && w0;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = timing ? clock64() : 0ull;
    const unsigned long long wall0 = timing ? wall_clock64() : 0ull;
#define FSTAMP(k) do { if (timing) { const unsigned long long now_ = clock64(); tph[k] += now_ - tlast; tlast = now_; } } while (0)
    const goicp_bnb_job jb = C.jobs[job];
    HeapNodeX* hg = C.heaps + (size_t)job * C.heap_cap;
    FrExp* ex = C.exps + (size_t)job * C.exp_cap;
    // children's bounds of the nodes expanded in the previous round whose sweeps were split over several workgroups: add the slices up
    if (C.parts) {
        for (int e = st.pend0 + (tid >> 4); e < st.ne; e += FR_SELECT_THREADS / 16) {
            const int k = tid & 15;
            const double* pp = C.parts + ((size_t)job * C.exp_cap + e) * C.nsplit * 16 + k;
            double acc = 0.0;
            for (int q = 0; q < C.nsplit; q++) acc += pp[(size_t)q * 16];
            if (k & 1) ex[e].lb[k >> 1] = (float)acc; else ex[e].ub[k >> 1] = (float)acc;
        }
        __threadfence_block();
    }
    st.pend0 = st.ne;
    // the queue (its first FR_LDS_NODES nodes) is staged in LDS for the replay and written back afterwards, by all waves
    const int nl = min(st.hn, FR_LDS_NODES);
    {
        const qv4* src = reinterpret_cast<const qv4*>(hg); qv4* dst = reinterpret_cast<qv4*>(top);
        for (int i = tid; i < 2 * nl; i += FR_SELECT_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    HeapStoreT<HeapNodeX> heap; heap.lds = top; heap.glb = hg; heap.hl = FR_LDS_NODES;
    int hn = st.hn, hi_water = nl;
    if (w0) {
    // the children's bounds of the nodes near the top of the queue are fetched now, all at once (one lane each), instead of one
    // dependent global read per replayed pop; pops of nodes that sat deeper fall back to the global read
    int my_eid = -1;
    if (lane < 63 && lane < hn) my_eid = top[lane].exp;
    if (my_eid >= 0) ecache[lane] = ex[my_eid];
    wave_sync_lds();
    FSTAMP(0);
    const unsigned pop_limit = jb.max_pops ? jb.max_pops : C.max_pops;
    for (;;) {
        if (hn == 0) { st.done = 1; break; }
        const HeapNodeX par = heap.get(0);
        // (the search ends on these three: the queue is not used again, so the pop itself is only counted)
        if (st.optErrorT - par.lb < jb.SSEThresh) { st.pops++; st.done = 1; break; }                                   // reference jly_goicp.cpp:317
        if (st.pops + 1 > pop_limit) { st.pops++; st.done = 1; st.status = GOICP_ERR_LIMIT; break; }
        if (hn - 1 + 8 > C.heap_cap || (par.exp < 0 && st.ne + 9 > C.exp_cap)) { st.pops++; st.done = 1; st.status = GOICP_ERR_OVERFLOW; break; }
        if (par.exp < 0) break;                                         // its children have not been evaluated yet
        FSTAMP(1);
        heap_pop_wave(heap, hn, lane);
        FSTAMP(2);
        st.pops++; st.cubes += 8;
        const unsigned long long hit = __ballot(my_eid == par.exp);
        FrExp E;
        {   // explicit address spaces (a pointer select would turn these into flat loads)
            qv4* ev = reinterpret_cast<qv4*>(&E);
            if (hit) {
                const __attribute__((address_space(3))) qv4* q = (const __attribute__((address_space(3))) qv4*)(ecache + (__ffsll((long long)hit) - 1));
#pragma unroll
                for (int k = 0; k < (int)(sizeof(FrExp) / 16); k++) ev[k] = q[k];
            } else {
                const __attribute__((address_space(1))) qv4* q = (const __attribute__((address_space(1))) qv4*)(ex + par.exp);
#pragma unroll
                for (int k = 0; k < (int)(sizeof(FrExp) / 16); k++) ev[k] = q[k];
            }
        }
        if (timing) { tph[6] += 1; if (!hit) tph[7] += 1; }
        const float w = par.w / 2;
#pragma unroll
        for (int j = 0; j < TC; j++) {                                  // incumbent update, prune, push (reference jly_goicp.cpp:554-575)
            const float ub = E.ub[j], lb = E.lb[j];
            const float cx = par.x + (j & 1) * w, cy = par.y + (j >> 1 & 1) * w, cz = par.z + (j >> 2 & 1) * w;
            if (ub < st.optErrorT) { st.optErrorT = ub; st.bx = cx; st.by = cy; st.bz = cz; st.bw = w; st.found = 1; }
            if (lb >= st.optErrorT) continue;
            HeapNodeX nd; nd.x = cx; nd.y = cy; nd.z = cz; nd.w = w; nd.ub = ub; nd.lb = lb; nd.pad = 0;
            nd.exp = E.child_base >= 0 ? E.child_base + j : -1;        // this child's own children were evaluated together with it
            hn++;
            heap_up_wave(heap, hn - 1, nd, lane);
        }
        hi_water = max(hi_water, min(hn, FR_LDS_NODES));
        FSTAMP(3);
    }
    wave_sync_lds();
    FSTAMP(1);
    // nodes to expand next: the (up to K) best keys without children's bounds among the first 63 queue slots (the true head of
    // the queue is always among them; deeper slots are picked up in later rounds). depth 2: the 8 children of a chosen node are
    // expanded in the same round (9 tiles per node), so a search that keeps diving into fresh children needs half the rounds.
    if (!st.done) {
        const bool have = lane < 63 && lane < hn;
        HeapNodeX me; me.lb = 0.f; me.w = 0.f; me.exp = 0; me.x = me.y = me.z = me.ub = 0.f; me.pad = 0;
        if (have) me = heap.get(lane);
        // a node whose bound already meets the stop test can never be expanded (the incumbent only falls): not worth a sweep
        const bool want = have && me.exp < 0 && !(st.optErrorT - me.lb < jb.SSEThresh);
        int rank = 0;
        // (only the wanted slots count, and their keys come through v_readlane -- a scalar lane index -- instead of 3 x 63 trips through the LDS crossbar)
        for (unsigned long long wm = __ballot(want); wm; wm &= wm - 1) {
            const int m = __ffsll((long long)wm) - 1;
            const float olb = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.lb), m)), ow = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.w), m));
            const bool better = olb != me.lb ? olb < me.lb : (ow != me.w ? ow > me.w : m < lane);
            rank += better ? 1 : 0;
        }
        const int slots = depth >= 2 ? 9 : 1;                          // expansion records per chosen node
        const bool sel = want && rank < K && st.ne + (rank + 1) * slots <= C.exp_cap;
        const int nsel = __popcll(__ballot(sel));
        if (nsel) {
            // (ranks of the selected lanes are 0 .. nsel-1: a selected lane's betters are all selected too)
            unsigned base = 0;
            const int per_tile = C.nsplit * C.ngroups, per_node = per_tile * slots;
            if (lane == 0) base = atomicAdd(&C.wl_count[round], (unsigned)(nsel * per_node));
            base = (unsigned)__shfl((int)base, 0, 64);
            if (sel) {
                const int eid = st.ne + rank * slots;
                const unsigned at = base + (unsigned)(rank * per_node);
                if (at + per_node <= (unsigned)C.wl_cap) {
                    const float hw = me.w / 2;
                    for (int t = 0; t < slots; t++) {
                        FrItem it; it.job = job; it.eid = eid + t; it.pad = slots == 9 && t == 0 ? eid + 1 : -1;    // pad = child_base of the record
                        if (t == 0) { it.x = me.x; it.y = me.y; it.z = me.z; it.w = me.w; }
                        else { const int j = t - 1; it.x = me.x + (j & 1) * hw; it.y = me.y + (j >> 1 & 1) * hw; it.z = me.z + (j >> 2 & 1) * hw; it.w = hw; }
                        for (int q = 0; q < per_tile; q++) { it.split = q; C.wl[at + t * per_tile + q] = it; }
                    }
                    me.exp = eid;
                    heap.set(lane, me);
                }
            }
            st.ne += nsel * slots; st.spec += 8u * nsel * slots;
        }
    }
    wave_sync_lds();
    FSTAMP(4);
    if (lane == 0) s_hi = hi_water;
    }
    __syncthreads();
    {
        const int hw2 = 2 * s_hi;
        const qv4* src = reinterpret_cast<const qv4*>(top); qv4* dst = reinterpret_cast<qv4*>(hg);
        for (int i = tid; i < hw2; i += FR_SELECT_THREADS) dst[i] = src[i];
    }
    FSTAMP(5);
    if (timing && lane == 0) {
        // per phase: summed over all searches and rounds (atomic adds); [6] replayed pops, [7] of which missed the prefetch
        for (int k = 0; k < 8; k++) if (tph[k]) atomicAdd(&C.dbg[k], tph[k]);
        atomicAdd(&C.dbg[8], wall_clock64() - wall0);
    }
#undef FSTAMP
    if (tid == 0) {
        st.hn = hn;
        C.st[job] = st;
        if (st.done) {
            goicp_bnb_result r;
            r.optErrorT = st.optErrorT; r.node[0] = st.bx; r.node[1] = st.by; r.node[2] = st.bz; r.node[3] = st.bw;
            r.found = st.found; r.status = st.status; r.pops = st.pops; r.cube_evals = st.cubes;
            C.results[job] = r;
        } else atomicAdd(&C.active[round], 1u);
    }
}

// MODE 0: register-resident residuals, points fetched just in time (sweep_tile, BLOCK 512 x PPT points per lane, Nd <= 5120); an item is
//         NT = 8 / ngroups children of the node (two items of 4 children each keep the residuals within 128 registers at 2 workgroups per CU)
// MODE 1: streamed cloud (sweep_stream, trimmed or untrimmed)          MODE 2: streamed untrimmed cloud, one slice of the points per item
template <int BLOCK, int PPT, int MODE, int NT, bool XBIG = false>
__device__ __forceinline__ void fr_expand_item(const DevTables& T, const FrCtx& C, TileShm<BLOCK>& sh, const FrItem item, float* rows, float* lutl) {
    const int tid = threadIdx.x;
    if (MODE == 0 && T.code) { for (int i = tid; i < LUT_LDS; i += BLOCK) lutl[i] = T.lut[i]; }   // (ordered by the barrier behind the cube set-up)
    const int level = C.jobs[item.job].rot_level;
    const float4* rot = C.srot + (size_t)item.job * C.NdP;
    const float* rho = level >= 0 ? T.maxRotDis + (size_t)level * T.Nd : nullptr;
    const int j0 = MODE == 0 ? item.split * NT : 0;        // first child of this item
    if (tid < TC) {                                     // children of the node (reference jly_goicp.cpp:322-333)
        const int j = (tid < NT ? tid : 0) + j0;        // rows past NT repeat child j0 (sweep_tile reads the centre of row 4 for the second z value)
        const float w = item.w / 2;
        const float x = item.x + (j & 1) * w, y = item.y + (j >> 1 & 1) * w, z = item.z + (j >> 2 & 1) * w;
        sh.cube[tid][0] = x; sh.cube[tid][1] = y; sh.cube[tid][2] = z; sh.cube[tid][3] = w;
        sh.cen[tid][0] = x + w / 2; sh.cen[tid][1] = y + w / 2; sh.cen[tid][2] = z + w / 2;
        sh.mtd[tid] = (float)(SQRT3_D / 2.0 * w);
    }
    __syncthreads();
    FrExp* out = C.exps + (size_t)item.job * C.exp_cap + item.eid;
    if constexpr (MODE == 0) {
        float px[PPT], py[PPT], pz[PPT], pw[PPT], prho[PPT];
#pragma unroll
        for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
        sweep_tile<BLOCK, PPT, true, NT, true, false, XBIG>(T, sh, px, py, pz, pw, prho, level >= 0, rows, false, rot, rho, T.code ? lutl : nullptr);
        if (tid < NT) { out->ub[j0 + tid] = sh.ub[tid]; out->lb[j0 + tid] = sh.lb[tid]; }
        if (tid == 0 && item.split == 0) out->child_base = item.pad;
    } else if constexpr (MODE == 1) {
        sweep_stream<true, BLOCK>(T, sh, rot, rho, rows, T.NdP);
        if (tid < TC) { out->ub[tid] = sh.ub[tid]; out->lb[tid] = sh.lb[tid]; }
        if (tid == 0) out->child_base = item.pad;
    } else {
        const int per = ((T.Nd + C.nsplit - 1) / C.nsplit + 63) & ~63;
        const int p0 = min(item.split * per, T.Nd), p1 = min(p0 + per, T.Nd);
        double* part = C.parts + (((size_t)item.job * C.exp_cap + item.eid) * C.nsplit + item.split) * 16;
        sweep_stream<true, BLOCK, true>(T, sh, rot, rho, nullptr, T.NdP, p0, p1, part);
        if (tid == 0 && item.split == 0) out->child_base = item.pad;
    }
}

template <int BLOCK, int PPT, int MODE, int NT, bool XBIG = false>
__global__ __launch_bounds__(BLOCK, (MODE == 0 ? 4 : (BLOCK == 512 ? 4 : 1))) void k_fr_expand(DevTables T, FrCtx C, int round, int item0) {
    __shared__ TileShm<BLOCK> sh;
    __shared__ float lutl[MODE == 0 ? LUT_LDS : 1];
    const unsigned n = min(C.wl_count[round], (unsigned)C.wl_cap);
    if constexpr (MODE == 1) {
        // a residual-row scratch per resident workgroup: persistent workgroups walk the list
        float* rows = T.srows ? T.srows + (size_t)blockIdx.x * (T.exact_stream ? 2 : 1) * TC * T.NdP : nullptr;
        for (unsigned it = blockIdx.x; it < n; it += gridDim.x) {
            fr_expand_item<BLOCK, PPT, MODE, NT, false>(T, C, sh, C.wl[it], rows, lutl);
            __syncthreads();
        }
    } else {
        // one item per workgroup (the host sizes the grid from the number of live searches). Inside a loop the compiler hoists the
        // per-lane point addresses of all PPT fetches out of it and the residuals no longer fit the 128 registers of 2 workgroups per CU.
        // (parity_mode 2: the launch goes out in chunks of T.xslots items, each workgroup with a slot of row scratch in HBM)
        if (blockIdx.x + item0 < n) fr_expand_item<BLOCK, PPT, MODE, NT, XBIG>(T, C, sh, C.wl[blockIdx.x + item0], XBIG ? T.xrows + (size_t)blockIdx.x * 2 * TC * T.NdP : nullptr, lutl);
    }
}

// ---------------------------------------------------------------------------------------------
// k_eval_cubes_stream: k_eval_cubes for clouds above 8192 points. Persistent workgroups walk the (tile, group) list so
// that the residual-row scratch (trimmed runs) is per resident workgroup, not per tile.
// ---------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, (BLOCK == 512 ? 4 : 1)) void k_eval_cubes_stream(DevTables T, const float4* __restrict__ rot_all, int slot0, int level0,
                                                            const int2* __restrict__ groups, int ngroups, const float* __restrict__ cubes, int n,
                                                            float* __restrict__ ub_out, float* __restrict__ lb_out,
                                                            const int4* __restrict__ pairs /* untrimmed runs: fused (no radius, radius) group pairs as in k_eval_cubes_geo, or null */) {
    __shared__ TileShm<BLOCK> sh;
    const int tid = threadIdx.x;
    const int Nd = T.Nd;
    float* rows = T.srows ? T.srows + (size_t)blockIdx.x * (T.exact_stream ? 2 : 1) * TC * T.NdP : nullptr;
    const int tiles = (n + TC - 1) / TC;
    const long long total = (long long)tiles * ngroups;            // (ngroups = number of pairs when `pairs` is given)
    for (long long w = blockIdx.x; w < total; w += gridDim.x) {
        const int g = (int)(w / tiles), base = (int)(w % tiles) * TC;
        int slot = slot0, rot_level = level0, row_a = g, row_b = g;
        if (pairs) { const int4 pp = pairs[g]; slot = pp.x; rot_level = pp.y; row_a = pp.z; row_b = pp.w; }
        else if (groups) { const int2 gg = groups[g]; slot = gg.x; rot_level = gg.y; }
        if (tid < TC) {
            const int j = tid, c = min(base + j, n - 1);
            const float x = cubes[4 * c], y = cubes[4 * c + 1], z = cubes[4 * c + 2], wd = cubes[4 * c + 3];
            sh.cube[j][0] = x; sh.cube[j][1] = y; sh.cube[j][2] = z; sh.cube[j][3] = wd;
            sh.cen[j][0] = x + wd / 2; sh.cen[j][1] = y + wd / 2; sh.cen[j][2] = z + wd / 2;
            sh.mtd[j] = (float)(SQRT3_D / 2.0 * wd);
        }
        __syncthreads();
        if (tid == 0) {
            bool row = true;
            for (int j = 1; j < TC; j++) row = row && sh.cen[j][1] == sh.cen[0][1] && sh.cen[j][2] == sh.cen[0][2];
            sh.ctl[2] = row ? 1 : 0;
        }
        __syncthreads();
        if (pairs) {
            sweep_stream<false, BLOCK>(T, sh, rot_all + (size_t)slot * Nd, nullptr, rows, T.NdP, 0, 0, nullptr, T.maxRotDis + (size_t)rot_level * Nd);
            if (tid < TC && base + tid < n) {
                ub_out[(size_t)row_a * n + base + tid] = sh.ub2[tid]; lb_out[(size_t)row_a * n + base + tid] = sh.lb2[tid];
                ub_out[(size_t)row_b * n + base + tid] = sh.ub[tid];  lb_out[(size_t)row_b * n + base + tid] = sh.lb[tid];
            }
        } else {
            sweep_stream<false, BLOCK>(T, sh, rot_all + (size_t)slot * Nd, rot_level >= 0 ? T.maxRotDis + (size_t)rot_level * Nd : nullptr, rows, T.NdP);
            if (tid < TC && base + tid < n) { ub_out[(size_t)g * n + base + tid] = sh.ub[tid]; lb_out[(size_t)g * n + base + tid] = sh.lb[tid]; }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k_eval_cubes: 8 free cubes per workgroup; cloud = rotation slot prepared by k_rotate
// ---------------------------------------------------------------------------------------------
template <int BLOCK, int PPT, int NT, bool JIT = false>
__global__ __launch_bounds__(BLOCK, (JIT ? (NT == 8 ? 4 : (NT == 4 ? 6 : 8)) : 1)) void k_eval_cubes(DevTables T, const float4* __restrict__ rot_all /*[slots][Nd] x,y,z,weight*/, int slot0, int level0,
                                                       const int2* __restrict__ groups /* (slot, rot_level) per blockIdx.y, or null */,
                                                       const float* __restrict__ cubes, int n, float* __restrict__ ub_out, float* __restrict__ lb_out,
                                                       float* fpscratch) {
    __shared__ TileShm<BLOCK> sh;
    __shared__ float lutl[JIT ? LUT_LDS : 1];           // leading entries of the coded grid's value table (just-in-time form only)
    extern __shared__ float rbuf[];
    const int tid = threadIdx.x;
    const int Nd = T.Nd;
    const int base = blockIdx.x * NT;
    if (JIT && T.code) { for (int i = tid; i < LUT_LDS; i += BLOCK) lutl[i] = T.lut[i]; }      // (ordered by the barrier behind the cube set-up)
    // one launch can sweep the same cube list for many (rotation, rho-level) pairs: blockIdx.y picks the pair
    int slot = slot0, rot_level = level0;
    if (groups) { const int2 g = groups[blockIdx.y]; slot = g.x; rot_level = g.y; }
    const float4* __restrict__ rot = rot_all + (size_t)slot * Nd;
    ub_out += (size_t)blockIdx.y * n; lb_out += (size_t)blockIdx.y * n;
    float* fpbuf = fpscratch ? fpscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NLMAX * Nd : nullptr;
    float px[PPT], py[PPT], pz[PPT], pw[PPT], prho[PPT];
    const bool use_rho = rot_level >= 0;
    const float* rho_row_jit = use_rho ? T.maxRotDis + (size_t)rot_level * Nd : nullptr;
    if (JIT) {
#pragma unroll
        for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
    } else {   // all loads first (clamped index instead of a branch), one wait for the lot
        float4 p[PPT]; float rh[PPT];
        const float* rho_row = T.maxRotDis + (size_t)(use_rho ? rot_level : 0) * Nd;
#pragma unroll
        for (int m = 0; m < PPT; m++) { const int i = min(tid + m * BLOCK, Nd - 1); p[m] = rot[i]; rh[m] = rho_row[i]; }
#pragma unroll
        for (int m = 0; m < PPT; m++) { px[m] = p[m].x; py[m] = p[m].y; pz[m] = p[m].z; pw[m] = p[m].w; prho[m] = use_rho ? rh[m] : 0.f; }
    }
    if (tid < NT) {
        const int j = tid, c = min(base + j, n - 1);          // the tail tile re-evaluates the last cube
        const float x = cubes[4 * c], y = cubes[4 * c + 1], z = cubes[4 * c + 2], w = cubes[4 * c + 3];
        sh.cube[j][0] = x; sh.cube[j][1] = y; sh.cube[j][2] = z; sh.cube[j][3] = w;
        sh.cen[j][0] = x + w / 2; sh.cen[j][1] = y + w / 2; sh.cen[j][2] = z + w / 2;
        sh.mtd[j] = (float)(SQRT3_D / 2.0 * w);
    }
    const bool penalties = T.use_compat || T.use_fpfh;
    __syncthreads();
    // every lane decides for itself whether the tile is a row of x-neighbours (broadcast LDS reads, no second barrier)
    bool row = true;
#pragma unroll
    for (int j = 1; j < NT; j++) row = row && sh.cen[j][1] == sh.cen[0][1] && sh.cen[j][2] == sh.cen[0][2];
    if (penalties && tid < 8 * NT) {                        // (read by lattice_terms, behind the sweep's barriers)
        const int j = tid >> 3, c = tid & 7;
        sh.lat[tid][0] = sh.cube[j][0] + (c & 1) * sh.cube[j][3];
        sh.lat[tid][1] = sh.cube[j][1] + (c >> 1 & 1) * sh.cube[j][3];
        sh.lat[tid][2] = sh.cube[j][2] + (c >> 2 & 1) * sh.cube[j][3];
    }
    sweep_tile<BLOCK, PPT, false, NT, JIT>(T, sh, px, py, pz, pw, prho, use_rho, rbuf, row, rot, rho_row_jit, (JIT && T.code) ? lutl : nullptr);
    if (penalties) lattice_terms<BLOCK, PPT>(T, sh, 8 * NT, px, py, pz, fpbuf);
    if (tid < NT && base + tid < n) {
        const int j = tid;
        float ub = sh.ub[j], lb = sh.lb[j];
        if (penalties) {
            int minI = 0, maxI = 0; float minF = 0.f, maxF = 0.f;
            for (int c = 0; c < 8; c++) {
                const int l = j * 8 + c;
                if (T.use_compat) { int v = sh.cnt[l]; if (c == 0) { minI = maxI = v; } else { if (v > maxI) maxI = v; if (v < minI) minI = v; } }
                if (T.use_fpfh) { int v = (int)sh.fpv[l]; if (c == 0) { minF = maxF = (float)v; } else { if (v > maxF) maxF = (float)v; if (v < minF) minF = (float)v; } }
            }
            if (T.use_compat) { ub += T.reg * (float)(maxI * maxI); lb += T.reg * (float)(minI * minI); }
            if (T.use_fpfh) { ub += T.regFPFH * (maxF * maxF); lb += T.regFPFH * (minF * minF); }
        }
        ub_out[base + j] = ub; lb_out[base + j] = lb;
    }
}

// Diagnosis only (goicp_engine_debug_gather_probe): the DT gathers of k_eval_cubes' row tiles -- same grid, same rotated clouds, same
// cube list, same (double) cell arithmetic, 8 lookups per point in flight -- with nothing else: no residuals kept, no selection.
// Its run time is the memory-system floor of the metric kernel's ACTUAL access pattern (unlike tools/probes/gather_probe.hip,
// whose synthetic tiles repeat and stay L2-resident).
// MODE 0: the tile's real lookups (8 per point, x-neighbours: about 2 cache lines per point, the other 6 lookups find their line already
//         requested).  The other modes separate what the memory system charges for, on the same points and tiles:
// MODE 1: only the lookups of cubes 0 and 4 (2 loads per point, about 2 lines: the misses alone);
// MODE 2: 8 loads per point, all of cube 0's cell (1 line per point, 7 loads that find it requested);
// MODE 3: 8 loads per point on 8 different grid rows (8 lines per point, no load finds its line requested);
// MODE 4, 5: the real lookups with lane = (point, cube) instead of lane = point (see below).
template <int BLOCK, int PPT, int MODE = 0>
__global__ __launch_bounds__(BLOCK) void k_gather_probe(DevTables T, const float4* __restrict__ rot_all, const int2* __restrict__ groups,
                                                        const float* __restrict__ cubes, int n, float* __restrict__ out) {
    __shared__ float cen[TC][3];
    const int tid = threadIdx.x, Nd = T.Nd, S = T.S;
    const int base = blockIdx.x * TC;
    const int2 g = groups[blockIdx.y];
    const float4* __restrict__ rot = rot_all + (size_t)g.x * Nd;
    if (tid < TC) {
        const int c = min(base + tid, n - 1);
        const float w = cubes[4 * c + 3];
        cen[tid][0] = cubes[4 * c] + w / 2; cen[tid][1] = cubes[4 * c + 1] + w / 2; cen[tid][2] = cubes[4 * c + 2] + w / 2;
    }
    __syncthreads();
    float acc = 0.f;
    if constexpr (MODE == 4 || MODE == 5) {
        // lane = (point, cube): the 8 lanes of a point read cells of one grid row 33 cells apart at most (1-2 lines per 8 lanes, 8-16 per
        // wave instruction instead of 64). MODE 5: the y / z cell arithmetic is done by one lane of the 8 and handed to the others.
        const int j = tid & 7, q = tid >> 3;
        const float cx = cen[j][0];
#pragma unroll 8
        for (int m = 0; m < TC * PPT; m++) {
            const float4 p = rot[min(q + m * (BLOCK / 8), Nd - 1)];
            int y, z;
            if (MODE == 5) {
                int yz = (j == 0) ? cell_of(p.y + cen[0][1], T.yMin, T.scale) : ((j == 1) ? cell_of(p.z + cen[0][2], T.zMin, T.scale) : 0);
                y = __shfl(yz, (tid & 63 & ~7), 64); z = __shfl(yz, (tid & 63 & ~7) + 1, 64);
            } else { y = cell_of(p.y + cen[0][1], T.yMin, T.scale); z = cell_of(p.z + cen[0][2], T.zMin, T.scale); }
            const unsigned ix = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y, S)) * S + clamp_cell(cell_of(p.x + cx, T.xMin, T.scale), S));
            acc += T.code ? (float)T.code[ix] : T.dist[ix];
        }
        if (acc == 12345.678f) out[blockIdx.x] = acc;
        return;
    }
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        const float4 p = rot[min(tid + m * BLOCK, Nd - 1)];
        const int y = cell_of(p.y + cen[0][1], T.yMin, T.scale), z = cell_of(p.z + cen[0][2], T.zMin, T.scale);
        const int rowbase = (clamp_cell(z, S) * S + clamp_cell(y, S)) * S;
        float v[TC];
#pragma unroll
        for (int j = 0; j < TC; j++) {
            if (MODE == 1 && (j & 3) != 0) { v[j] = 0.f; continue; }
            const int jj = (MODE == 2 || MODE == 3) ? 0 : j;
            unsigned ix = (unsigned)(rowbase + clamp_cell(cell_of(p.x + cen[jj][0], T.xMin, T.scale), S));
            if (MODE == 3) ix = (unsigned)((clamp_cell(z, S) * S + clamp_cell(y + 3 * j - 10, S)) * S + clamp_cell(cell_of(p.x + cen[0][0], T.xMin, T.scale), S));
            if (MODE == 2) { const volatile uint16_t* q = T.code + ix; const volatile float* qf = T.dist + ix; v[j] = T.code ? (float)*q : *qf; }   // (volatile: eight separate loads)
            else v[j] = T.code ? (float)T.code[ix] : T.dist[ix];
        }
#pragma unroll
        for (int j = 0; j < TC; j++) acc += v[j];
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// k_eval_cubes_geo: the cube sweep of clouds of 1025..5120 points without penalty terms (512 lanes, points fetched just in time) -- the
// metric kernel. A kernel of its own rather than an instantiation of k_eval_cubes: without the penalty-lattice and parity-mode code next
// to it the 80 residuals stay in registers (106 VGPRs, no spills; the general kernel spilled 26).
// DUAL = false: groups[y] = (slot, rot_level), output row y.
// DUAL = true : PAIRS of groups that share rotation slot and cube list -- the call without rotation uncertainty (rot_level -1) and a call
//               with it, as OuterBnB's two InnerBnB calls per child rotation do (reference jly_goicp.cpp:768,861): one pass of gathers, two
//               bound pairs per cube (sweep_tile<..., DUAL>). groups[y] = (slot, rot_level of the second call, output row of the first
//               call, output row of the second).
template <int PPT, bool DUAL, bool XBIG = false>
__global__ __launch_bounds__(512, 4) void k_eval_cubes_geo(DevTables T, const float4* __restrict__ rot_all, const int4* __restrict__ groups,
                                                           const float* __restrict__ cubes, int n, float* __restrict__ ub_out, float* __restrict__ lb_out,
                                                           int tiles, int ngroups, int xcd_mode, int lin0 /* first workgroup of this launch (chunked launches of the exact mode) */) {
    __shared__ TileShm<512> sh;
    __shared__ float lutl[LUT_LDS];
    // 1-D grid; workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MB L2), so `blockIdx.x & 7` labels the workgroups
    // that share an L2 (speed only, MI355X_MICROARCH.md "Workgroup dispatch"). xcd_mode 0: workgroup b = tile b % tiles of group b / tiles;
    // 1: every XCD label takes a contiguous eighth of the (group, tile) list -- whole rotations; 2: a contiguous eighth of the TILES
    // (neighbouring translations) of every group. The grid is padded to a multiple of 8 labels; the surplus workgroups leave at once.
    int bx, by;
    {
        const int lin = blockIdx.x, lab = lin & 7, slot = lin >> 3;
        if (xcd_mode == 1) {
            const int total = tiles * ngroups, per = (total + 7) >> 3, L = lab * per + slot;
            if (slot >= per || L >= total) return;
            bx = L % tiles; by = L / tiles;
        } else if (xcd_mode == 2) {
            const int tper = (tiles + 7) >> 3;
            by = slot / tper; bx = lab * tper + slot % tper;
            if (by >= ngroups || bx >= tiles) return;
        } else { const int l2 = lin + lin0; if (l2 >= tiles * ngroups) return; bx = l2 % tiles; by = l2 / tiles; }
    }
    // parity_mode 2: this workgroup's rows in HBM (the launch holds at most T.xslots workgroups)
    float* xr = XBIG ? T.xrows + (size_t)blockIdx.x * 2 * TC * T.NdP : nullptr;
    const int tid = threadIdx.x, Nd = T.Nd, base = bx * TC;
    if (T.code) { for (int i = tid; i < LUT_LDS; i += 512) lutl[i] = T.lut[i]; }     // (ordered by the barrier behind the cube set-up)
    int4 g = groups[by];
    // (wave-uniform: kept in scalar registers, so that the output addresses formed from them at the end cost no vector registers in between)
    g.x = __builtin_amdgcn_readfirstlane(g.x); g.y = __builtin_amdgcn_readfirstlane(g.y); g.z = __builtin_amdgcn_readfirstlane(g.z); g.w = __builtin_amdgcn_readfirstlane(g.w);
    const float4* __restrict__ rot = rot_all + (size_t)g.x * Nd;
    const float* __restrict__ rho = g.y >= 0 ? T.maxRotDis + (size_t)g.y * Nd : nullptr;
    if (tid < TC) {
        const int j = tid, c = min(base + j, n - 1);          // the tail tile re-evaluates the last cube
        const float x = cubes[4 * c], y = cubes[4 * c + 1], z = cubes[4 * c + 2], w = cubes[4 * c + 3];
        sh.cube[j][0] = x; sh.cube[j][1] = y; sh.cube[j][2] = z; sh.cube[j][3] = w;
        sh.cen[j][0] = x + w / 2; sh.cen[j][1] = y + w / 2; sh.cen[j][2] = z + w / 2;
        sh.mtd[j] = (float)(SQRT3_D / 2.0 * w);
    }
    __syncthreads();
    // every lane decides for itself whether the tile is a row of x-neighbours (broadcast LDS reads, no second barrier)
    bool row = true;
#pragma unroll
    for (int j = 1; j < TC; j++) row = row && sh.cen[j][1] == sh.cen[0][1] && sh.cen[j][2] == sh.cen[0][2];
    float px[PPT], py[PPT], pz[PPT], pw[PPT], prho[PPT];
#pragma unroll
    for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
    if constexpr (DUAL) {
        sweep_tile<512, PPT, false, TC, true, true, XBIG>(T, sh, px, py, pz, pw, prho, false, xr, row, rot, nullptr, T.code ? lutl : nullptr, rho);
        // (results out by lane id, from every wave alike -- identical values to identical addresses: the workitem id would have to survive the
        // whole sweep in a register the kernel does not have, i.e. in scratch)
        const int ol = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (ol < TC && base + ol < n) {
            ub_out[(size_t)g.z * n + base + ol] = sh.ub2[ol]; lb_out[(size_t)g.z * n + base + ol] = sh.lb2[ol];
            ub_out[(size_t)g.w * n + base + ol] = sh.ub[ol];  lb_out[(size_t)g.w * n + base + ol] = sh.lb[ol];
        }
    } else {
        sweep_tile<512, PPT, false, TC, true, false, XBIG>(T, sh, px, py, pz, pw, prho, g.y >= 0, xr, row, rot, rho, T.code ? lutl : nullptr);
        const int ol = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (ol < TC && base + ol < n) { ub_out[(size_t)by * n + base + ol] = sh.ub[ol]; lb_out[(size_t)by * n + base + ol] = sh.lb[ol]; }
    }
}

// lattice terms only, at n free translations (NLMAX per workgroup)
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void k_corner_terms(DevTables T, const float4* __restrict__ rot, const float* __restrict__ xyz, int n,
                                                         int* __restrict__ cnt_out, float* __restrict__ fp_out, float* fpscratch) {
    __shared__ TileShm<BLOCK> sh;
    const int tid = threadIdx.x, Nd = T.Nd, base = blockIdx.x * NLMAX;
    const int nl = min(NLMAX, n - base);
    float* fpbuf = fpscratch ? fpscratch + (size_t)blockIdx.x * NLMAX * Nd : nullptr;
    float px[PPT], py[PPT], pz[PPT];
#pragma unroll
    for (int m = 0; m < PPT; m++) {
        const int i = tid + m * BLOCK;
        px[m] = py[m] = pz[m] = 0.f;
        if (i < Nd) { float4 p = rot[i]; px[m] = p.x; py[m] = p.y; pz[m] = p.z; }
    }
    for (int l = tid; l < nl; l += BLOCK) { sh.lat[l][0] = xyz[3 * (base + l)]; sh.lat[l][1] = xyz[3 * (base + l) + 1]; sh.lat[l][2] = xyz[3 * (base + l) + 2]; sh.cnt[l] = 0; sh.fpv[l] = 0.f; }
    __syncthreads();
    lattice_terms<BLOCK, PPT>(T, sh, nl, px, py, pz, fpbuf);
    for (int l = tid; l < nl; l += BLOCK) { cnt_out[base + l] = T.use_compat ? sh.cnt[l] : 0; fp_out[base + l] = T.use_fpfh ? sh.fpv[l] : 0.f; }
}

// engine-creation helpers: compact id of every occupied cell, then the S^3-sized resolution of nearest_occ
__global__ void k_scatter_occ(const int* __restrict__ cells, int n, int* __restrict__ occ_id) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n) occ_id[cells[o]] = o;
}
__global__ void k_build_cmask(const int* __restrict__ nearest, const int* __restrict__ occ_id, const uint16_t* __restrict__ occ_mask,
                              unsigned long long ncell, uint16_t* __restrict__ cmask, int* __restrict__ occ_out) {
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncell) return;
    const int nk = nearest[k];
    const int o = (nk >= 0 && (unsigned long long)nk < ncell) ? occ_id[nk] : -1;
    cmask[k] = o < 0 ? (uint16_t)0 : occ_mask[o];
    if (occ_out) occ_out[k] = o;
}

// coded grid: d2 = round((dist * scale)^2) is accepted as the cell's code only if the LUT gives back the cell's float bit for bit
__global__ void k_build_codes(const float* __restrict__ dist, const float* __restrict__ lut, int lut_n, double scale, unsigned long long ncell,
                              uint16_t* __restrict__ code, unsigned long long* __restrict__ escapes) {
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool esc = false;
    if (k < ncell) {
        const float d = dist[k];
        const double t = (double)d * scale;
        const double n = rint(t * t);
        unsigned c = 0xFFFFu;
        if (n >= 0.0 && n < (double)lut_n && n < 65535.0) { const unsigned ni = (unsigned)n; if (__float_as_uint(lut[ni]) == __float_as_uint(d)) c = ni; }
        code[k] = (uint16_t)c;
        esc = c == 0xFFFFu;
    }
    const unsigned long long b = __ballot(esc);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(escapes, (unsigned long long)__popcll(b));
}

// K5: rotated cloud + weight as float4 (reference jly_goicp.cpp:750-756)
__global__ void k_rotate(DevTables T, const float* __restrict__ R, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.Nd) return;
    const float x = T.dxyz[3 * i], y = T.dxyz[3 * i + 1], z = T.dxyz[3 * i + 2];
    float4 p;
    p.x = R[0] * x + R[1] * y + R[2] * z;
    p.y = R[3] * x + R[4] * y + R[5] * z;
    p.z = R[6] * x + R[7] * y + R[8] * z;
    p.w = T.weights[i];
    out[i] = p;
}

// (float)(R*p + t) with double R,t (reference jly_goicp.cpp:120-122, 937-939), DT lookup, compat flag
struct RtArgD { double v[12]; };                 // double R[9], t[3] by value
__global__ void k_point_eval(DevTables T, RtArgD rt, float* __restrict__ dist, int* __restrict__ compat) {
    const double* Rt = rt.v;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.Nd) return;
    const float px = T.dxyz[3 * i], py = T.dxyz[3 * i + 1], pz = T.dxyz[3 * i + 2];
    const float x = (float)(Rt[0] * px + Rt[1] * py + Rt[2] * pz + Rt[9]);
    const float y = (float)(Rt[3] * px + Rt[4] * py + Rt[5] * pz + Rt[10]);
    const float z = (float)(Rt[6] * px + Rt[7] * py + Rt[8] * pz + Rt[11]);
    dist[i] = dt_lookup(T, x, y, z);
    if (compat) {
        int ok = 1;
        if (T.cmask) { unsigned mask = T.cmask[clamped_cell(T, x, y, z)]; ok = (mask >> T.dcls[i]) & 1u; }
        compat[i] = ok;
    }
}

// exhaustive 1-NN in float with the distance expression of the reference's kd-tree metric
// (jly_icp3d.hpp:54-60). One wave per query point: lane l scans model points l, l+64, ... keeping its closest point and how
// many of its points sit at exactly that distance; the wave then reduces on (distance, index). The reference's kd-tree keeps
// the FIRST point it meets among exact ties (nanoflann.hpp:1147), which depends on the tree, so the kernel also reports the
// number of model points at the minimal distance: where it exceeds one, the host names the reference's choice
// (host_kdorder.hpp); id_out then holds the lowest such index.
struct RtArg { float v[12]; };                    // float R[9], t[3], passed by value (no staging copy per ICP iteration)
__global__ __launch_bounds__(256) void k_icp_nn(DevTables T, const float* __restrict__ mxyz, RtArg rt,
                                                 int* __restrict__ id_out, float* __restrict__ d_out, int* __restrict__ tie_out) {
    const float* Rt = rt.v;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= T.Nd) return;                         // whole waves leave together
    const float d0 = T.dxyz[3 * i], d1 = T.dxyz[3 * i + 1], d2 = T.dxyz[3 * i + 2];
    const float q0 = Rt[0] * d0 + Rt[1] * d1 + Rt[2] * d2 + Rt[9];
    const float q1 = Rt[3] * d0 + Rt[4] * d1 + Rt[5] * d2 + Rt[10];
    const float q2 = Rt[6] * d0 + Rt[7] * d1 + Rt[8] * d2 + Rt[11];
    float best = INFINITY; int bi = 0x7fffffff, cnt = 0;
    for (int m = lane; m < T.Nm; m += 64) {
        const float e0 = q0 - mxyz[3 * m], e1 = q1 - mxyz[3 * m + 1], e2 = q2 - mxyz[3 * m + 2];
        const float d = e0 * e0 + e1 * e1 + e2 * e2;
        if (d < best) { best = d; bi = m; cnt = 1; }
        else if (d == best) cnt++;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off, 64); const int oi = __shfl_down(bi, off, 64); const int oc = __shfl_down(cnt, off, 64);
        if (ob < best) { best = ob; bi = oi; cnt = oc; }
        else if (ob == best) { cnt += oc; if (oi < bi) bi = oi; }
    }
    if (lane == 0) { id_out[i] = bi; d_out[i] = best; if (tie_out) tie_out[i] = cnt; }
}

// ---------------------------------------------------------------------------------------------
// k_icp_loop: ICP3D<float>::Run (reference jly_icp3d.hpp:197-311) as ONE launch for cavity-sized, untrimmed clouds: the whole
// iteration -- exhaustive 1-NN (the search of k_icp_nn), the running centroids (never reset, divided by n: the reference's quirks), the
// error sum, the 3x3 covariance, Kabsch and the pose update -- repeats inside one workgroup until the reference's stop test fires. Every
// sum whose order matters runs sequentially over the point index on its own lane (7 + 9 chains), so the poses and the error are the bits
// the host loop (host_goicp.cpp: icp_run) produces. Why: an iteration used to be one launch + one synchronisation; beside other engines'
// search launches (batch driver) that round trip took 180 us instead of 20 us and ICP was 60 % of a cavity registration.
// An iteration that meets an exact distance tie between model points is NOT decided here (the reference's kd-tree order decides it,
// host_kdorder.hpp): the kernel returns the state before that iteration with status 1 and the host loop carries on from it.
// ---------------------------------------------------------------------------------------------
struct IcpState { double R[9], t[3], mu_m[3], mu_d[3]; float err, err_new; int iters, status; };
__global__ __launch_bounds__(1024) void k_icp_loop(DevTables T, const float* __restrict__ mxyz, IcpState st0, float err_diff, int max_iters,
                                                    IcpState* __restrict__ st_out, int* __restrict__ ids_out, float* __restrict__ dsq_out) {
    extern __shared__ float icp_lds[];
    const int n = T.Nd, Nm = T.Nm, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NWV = blockDim.x >> 6;
    float* mx = icp_lds;                  // [3 Nm] model
    float* pd = mx + 3 * Nm;              // [3 n]  R*d + t of the current pose (float expression of jly_icp3d.hpp:239-241)
    float* pm = pd + 3 * n;               // [3 n]  matched model point
    float* dsq = pm + 3 * n;              // [n]
    int* ids = reinterpret_cast<int*>(dsq + n);
    float* dx = reinterpret_cast<float*>(ids + n);        // [3 n] data cloud
    float* pbest = dx + 3 * n;            // [parts * n] partial searches: closest distance, its lowest index, how many points sit at it
    int* pidx = reinterpret_cast<int*>(pbest + 1024);
    int* pcnt = pidx + 1024;
    __shared__ double sR[9], st[3], smu_m[3], smu_d[3], sH[9];
    __shared__ float serr, serr_new;
    __shared__ int sties, sstop;
    for (int k = tid; k < 3 * Nm; k += blockDim.x) mx[k] = mxyz[k];
    for (int k = tid; k < 3 * n; k += blockDim.x) dx[k] = T.dxyz[k];
    // search layout: lane = (query, part of the model): every lane scans its part front to back, so "first point met at the closest distance"
    // is the lowest index and no lane-to-lane reduction is needed; the parts of a query are merged in ascending order afterwards
    const int parts = min(4, max(1, 1024 / n)), chunk = (Nm + parts - 1) / parts;
    const int qi = tid % n, qp = tid / n;
    if (tid == 0) {
        for (int k = 0; k < 9; k++) sR[k] = st0.R[k];
        for (int k = 0; k < 3; k++) { st[k] = st0.t[k]; smu_m[k] = st0.mu_m[k]; smu_d[k] = st0.mu_d[k]; }
        serr = st0.err; serr_new = st0.err_new; sties = 0; sstop = 0;
    }
    __syncthreads();
    int iters = st0.iters, status = 0;
    // diagnosis (GOICP_DBG & 32): cycles of lane 0 per phase into T.dbg_times[8..12] (search, sums, covariance, 3x3 algebra), [13] iterations
    const bool timing = (T.dbg & 32) && T.dbg_times && tid == 0;
    unsigned long long tph[4] = {0, 0, 0, 0}, tlast = timing ? clock64() : 0ull;
#define ISTAMP(k) do { if (timing) { const unsigned long long now_ = clock64(); tph[k] += now_ - tlast; tlast = now_; } } while (0)
    for (;;) {
        if (iters >= max_iters) { status = 2; break; }
        iters++;
        // ---- correspondences: one wave per query, lanes over the model points (same arithmetic and reduction as k_icp_nn)
        const float r00 = (float)sR[0], r01 = (float)sR[1], r02 = (float)sR[2], r10 = (float)sR[3], r11 = (float)sR[4], r12 = (float)sR[5];
        const float r20 = (float)sR[6], r21 = (float)sR[7], r22 = (float)sR[8], t0 = (float)st[0], t1 = (float)st[1], t2 = (float)st[2];
        if (qp < parts) {
            const float d0 = dx[3 * qi], d1 = dx[3 * qi + 1], d2 = dx[3 * qi + 2];
            const float q0 = r00 * d0 + r01 * d1 + r02 * d2 + t0;
            const float q1 = r10 * d0 + r11 * d1 + r12 * d2 + t1;
            const float q2 = r20 * d0 + r21 * d1 + r22 * d2 + t2;
            float best = INFINITY; int bi = 0x7fffffff, cnt = 0;
            const int m1 = min((qp + 1) * chunk, Nm);
            int m = qp * chunk;
            for (; m + 4 <= m1; m += 4) {                       // (all lanes of a part read the same model point: LDS broadcast)
                float e[12];
#pragma unroll
                for (int k = 0; k < 12; k++) e[k] = mx[3 * m + k];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float e0 = q0 - e[3 * k], e1 = q1 - e[3 * k + 1], e2 = q2 - e[3 * k + 2];
                    const float d = e0 * e0 + e1 * e1 + e2 * e2;
                    if (d < best) { best = d; bi = m + k; cnt = 1; }
                    else if (d == best) cnt++;
                }
            }
            for (; m < m1; m++) {
                const float e0 = q0 - mx[3 * m], e1 = q1 - mx[3 * m + 1], e2 = q2 - mx[3 * m + 2];
                const float d = e0 * e0 + e1 * e1 + e2 * e2;
                if (d < best) { best = d; bi = m; cnt = 1; }
                else if (d == best) cnt++;
            }
            pbest[tid] = best; pidx[tid] = bi; pcnt[tid] = cnt;
        }
        __syncthreads();
        if (tid < n) {
            float best = pbest[tid]; int bi = pidx[tid], cnt = pcnt[tid];
            for (int p = 1; p < parts; p++) {                   // ascending parts: at equal distance the earlier (lower) index stays
                const float ob = pbest[p * n + tid]; const int oi = pidx[p * n + tid], oc = pcnt[p * n + tid];
                if (ob < best) { best = ob; bi = oi; cnt = oc; }
                else if (ob == best) cnt += oc;
            }
            const int i = tid;
            const float d0 = dx[3 * i], d1 = dx[3 * i + 1], d2 = dx[3 * i + 2];
            ids[i] = bi; dsq[i] = best;
            pd[3 * i] = r00 * d0 + r01 * d1 + r02 * d2 + t0; pd[3 * i + 1] = r10 * d0 + r11 * d1 + r12 * d2 + t1; pd[3 * i + 2] = r20 * d0 + r21 * d1 + r22 * d2 + t2;
            pm[3 * i] = mx[3 * bi]; pm[3 * i + 1] = mx[3 * bi + 1]; pm[3 * i + 2] = mx[3 * bi + 2];
            if (cnt > 1) sties = 1;
        }
        __syncthreads();
        ISTAMP(0);
        if (sties) { status = 1; iters--; break; }            // this iteration belongs to the host (kd-tree order among tied model points)
        // ---- the seven running sums, each sequential over the point index on a lane of its own (jly_icp3d.hpp:256-276)
        if (tid < 6) {
            const float* src = tid < 3 ? pm + tid : pd + (tid - 3);
            double acc = tid < 3 ? smu_m[tid] : smu_d[tid - 3];
            int i = 0;
            for (; i + 32 <= n; i += 32) {                     // 32 reads in flight, then the 32 dependent adds
                float v[32];
#pragma unroll
                for (int q = 0; q < 32; q++) v[q] = src[3 * (i + q)];
#pragma unroll
                for (int q = 0; q < 32; q++) acc += (double)v[q];
            }
            for (; i + 8 <= n; i += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = src[3 * (i + q)];
#pragma unroll
                for (int q = 0; q < 8; q++) acc += (double)v[q];
            }
            for (; i < n; i++) acc += (double)src[3 * i];
            if (tid < 3) smu_m[tid] = acc; else smu_d[tid - 3] = acc;
        } else if (tid == 6) {
            float e = 0.f;
            int i = 0;
            for (; i + 32 <= n; i += 32) {
                float v[32];
#pragma unroll
                for (int q = 0; q < 32; q++) v[q] = dsq[i + q];
#pragma unroll
                for (int q = 0; q < 32; q++) e = (float)((double)e + (double)v[q]);
            }
            for (; i + 8 <= n; i += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = dsq[i + q];
#pragma unroll
                for (int q = 0; q < 8; q++) e = (float)((double)e + (double)v[q]);
            }
            for (; i < n; i++) e = (float)((double)e + (double)dsq[i]);
            serr_new = e;
        }
        __syncthreads();
        if (tid == 0) {
            const float err = serr, err_new = serr_new;
            if (err > 0 && err - err_new < err_diff * n) sstop = 1;            // jly_icp3d.hpp:272 (num = n: untrimmed)
            else {
                serr = err_new;
                for (int k = 0; k < 3; k++) { smu_m[k] = smu_m[k] / (double)(float)n; smu_d[k] = smu_d[k] / (double)(float)n; }
            }
        }
        __syncthreads();
        ISTAMP(1);
        if (sstop) { status = 0; break; }
        // ---- covariance: nine sequential chains (jly_icp3d.hpp:281)
        if (tid < 9) {
            const int a = tid / 3, b = tid % 3;
            const double ma = smu_d[a], mb = smu_m[b];
            double acc = 0;
            int i = 0;
            for (; i + 16 <= n; i += 16) {                     // 32 reads in flight; the products do not wait for the running sum
                float va[16], vb[16];
#pragma unroll
                for (int q = 0; q < 16; q++) { va[q] = pd[3 * (i + q) + a]; vb[q] = pm[3 * (i + q) + b]; }
                double pr[16];
#pragma unroll
                for (int q = 0; q < 16; q++) pr[q] = ((double)va[q] - ma) * ((double)vb[q] - mb);
#pragma unroll
                for (int q = 0; q < 16; q++) acc += pr[q];
            }
            for (; i + 8 <= n; i += 8) {
                float va[8], vb[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { va[q] = pd[3 * (i + q) + a]; vb[q] = pm[3 * (i + q) + b]; }
#pragma unroll
                for (int q = 0; q < 8; q++) acc += ((double)va[q] - ma) * ((double)vb[q] - mb);
            }
            for (; i < n; i++) acc += ((double)pd[3 * i + a] - ma) * ((double)pm[3 * i + b] - mb);
            sH[tid] = acc;
        }
        __syncthreads();
        ISTAMP(2);
        if (tid == 0) {
            double H[9], Rn[9], Rd[9], Rc[9], td[3], tc[3], t_[3], mum[3], mud[3];
            for (int k = 0; k < 9; k++) { H[k] = sH[k]; Rd[k] = sR[k]; }
            for (int k = 0; k < 3; k++) { td[k] = st[k]; mum[k] = smu_m[k]; mud[k] = smu_d[k]; }
            goicp::kabsch_rotation(H, Rn);
            for (int a = 0; a < 3; a++) { double s_ = 0; for (int k = 0; k < 3; k++) s_ += Rn[3 * a + k] * mud[k]; t_[a] = mum[a] - s_; }
            goicp::mul3(Rn, Rd, Rc);
            for (int a = 0; a < 3; a++) { double s_ = 0; for (int k = 0; k < 3; k++) s_ += Rn[3 * a + k] * td[k]; tc[a] = s_ + t_[a]; }
            for (int k = 0; k < 9; k++) sR[k] = Rc[k];
            for (int k = 0; k < 3; k++) st[k] = tc[k];
        }
        __syncthreads();
        ISTAMP(3);
    }
    if (timing) { for (int k = 0; k < 4; k++) atomicAdd(&T.dbg_times[8 + k], tph[k]); atomicAdd(&T.dbg_times[12], (unsigned long long)(iters - st0.iters)); }
#undef ISTAMP
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) { ids_out[i] = ids[i]; dsq_out[i] = dsq[i]; }
    if (tid == 0) {
        IcpState o;
        for (int k = 0; k < 9; k++) o.R[k] = sR[k];
        for (int k = 0; k < 3; k++) { o.t[k] = st[k]; o.mu_m[k] = smu_m[k]; o.mu_d[k] = smu_d[k]; }
        o.err = serr; o.err_new = serr_new; o.iters = iters; o.status = status;
        *st_out = o;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct goicp_engine {
    int device;
    hipStream_t stream;
    int upload_defer;                         // goicp_engine_create: uploads are waited for once, at its end (their sources live that long)
    hipStream_t stream_bulk;                  // search launches of cavity-sized engines (see cache_stream_aux); null: they use `stream`
    hipStream_t stream_hi;                    // high-priority stream for the small latency-critical kernels (ICP 1-NN, point evaluation): beside other
                                              // engines' bulk search launches they would otherwise queue behind thousands of search workgroups
    hipEvent_t ev_wait; int wait_block;       // host waits: spin on the stream (default) or sleep on a blocking event (GOICP_WAIT=block; more host threads than cores)
    void* pinned;                             // PINNED_BYTES of host memory (may be null), travels with the stream in the pool
    hipEvent_t ev0, ev1;
    DevTables T;
    std::vector<void*> allocs;
    // device buffers
    float* d_mxyz; float4* d_rot; float* d_R; int nslots;
    goicp_bnb_job* d_jobs; goicp_bnb_result* d_results; int job_cap;
    HeapNode* d_heaps; int heap_cap; size_t heap_jobs; HeapNode* d_heap_big; int heap_big_cap;
    float* d_fpscratch; size_t fp_blocks;
    float* d_cubes; float* d_ub; float* d_lb; int cube_cap; int2* d_groups; std::vector<int2> groups_host;
    int4* d_pairs; int4* d_groups4; int4 one_group; int npairs; int fuse;     // fused evaluation of (rot_level -1, rot_level >= 0) group pairs of one slot (k_eval_cubes_dual)
    float* d_dist; int* d_int; int* d_tie; float* d_lat; int lat_cap; int* d_cnt; float* d_fp;
    int block, ppt; size_t dyn_lds; size_t dyn_lds_bnb; int heap_lds_nodes; int rbuf_floats; int fp_lds;
    int resident_jobs;                       // search workgroups resident at once (CUs x occupancy of the chosen k_inner_bnb)
    int streamed; size_t stream_wgs, stream_rot_wgs;   // streaming path: scratch sized for this many resident workgroups (rows / rotated clouds)
    int last_fused;                          // the last goicp_engine_eval_cubes* call evaluated its (no radius, radius) group pairs from one pass of gathers
    int xcd_mode;                            // workgroup -> (group, tile) map of k_eval_cubes_geo (GOICP_XCD, read once per engine)
    int jit_eval;                            // clouds of 1025..5120 points without penalty terms: the just-in-time sweep kernels (GOICP_JIT=0 disables; read once per engine)
    int jit_bnb;                             // geometric searches of 1025..2048 points: k_inner_bnb<512, 4, false, false, true>
    // round-based search (k_fr_*): geometric searches of more than 1024 points
    int fr_mode;                             // 0 off, else 1 + MODE of k_fr_expand
    int fr_ppt, fr_block, fr_nsplit, fr_wgs, fr_heap_cap, fr_depth2;
    FrState* fr_st; HeapNodeX* fr_heaps; FrExp* fr_exps; FrItem* fr_wl; unsigned* fr_cnt; double* fr_parts; float4* fr_srot; goicp_bnb_result* fr_results;
    size_t fr_jobs, fr_cap_nodes, fr_cap_exps; int fr_wl_cap;
    uint64_t fr_rounds, fr_spec_cubes;
    unsigned max_pops;
    uint64_t n_cubes, n_lookups; double kernel_ms;
    std::string err, kernel_name;
    std::vector<int4> pairs_host, groups4_host;
    unsigned long long code_escapes;          // cells of the coded grid that fall back to the float grid
};

// ---------------------------------------------------------------------------------------------
// Device-memory cache. One engine lives for one pair, so a batch creates and destroys engines all the time; hipMalloc and
// hipFree synchronise the device and serialise the host threads that run registrations side by side. Freed blocks are
// therefore parked per (device, size class) and handed out again; at most GOICP_CACHE_GB (default 16) gigabytes stay parked.
// A block is only parked after its engine's stream has been synchronised, so the next owner may use it on any stream.
// ---------------------------------------------------------------------------------------------
namespace {
struct DevCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void*> parked;
    std::map<void*, std::pair<int, size_t>> live;
    size_t parked_bytes = 0, limit = 0; bool init = false, off = false;
    struct Queue { int device; hipStream_t stream; hipEvent_t ev0, ev1; void* pinned; };   // pinned: PINNED_BYTES of host memory the device writes results into
    std::vector<Queue> queues;                 // idle (stream, event, event) triples of destroyed engines
    struct Aux { int device, kind; hipStream_t stream; };
    std::vector<Aux> aux_streams;              // idle auxiliary streams (cache_stream_aux)
    static size_t size_class(size_t n) {
        if (n <= 256) return 256;
        if (n < ((size_t)1 << 20)) { size_t p2 = 256; while (p2 < n) p2 <<= 1; return p2; }
        return (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    }
    void setup() {
        if (init) return;
        init = true;
        const char* e = getenv("GOICP_CACHE_GB");
        const double gb = e ? atof(e) : 16.0;
        off = gb <= 0.0; limit = (size_t)(gb * 1073741824.0);
    }
};
DevCache g_cache;
}  // namespace
hipError_t cache_malloc(void** out, size_t bytes, int device) {
    const size_t cls = DevCache::size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        g_cache.setup();
        if (!g_cache.off) {
            auto it = g_cache.parked.find({device, cls});
            if (it != g_cache.parked.end()) {
                *out = it->second; g_cache.parked.erase(it); g_cache.parked_bytes -= cls;
                g_cache.live[*out] = {device, cls};
                return hipSuccess;
            }
        }
    }
    const hipError_t e = hipMalloc(out, cls);
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_cache.mu); g_cache.live[*out] = {device, cls}; }
    return e;
}
void cache_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        auto it = g_cache.live.find(p);
        if (it != g_cache.live.end()) {
            const auto key = it->second;
            g_cache.live.erase(it);
            if (!g_cache.off && g_cache.parked_bytes + key.second <= g_cache.limit) { g_cache.parked.emplace(key, p); g_cache.parked_bytes += key.second; return; }
        }
    }
    (void)hipFree(p);
}
#define PINNED_BYTES ((size_t)1 << 20)
hipError_t cache_queue(int device, hipStream_t* st, hipEvent_t* e0, hipEvent_t* e1, void** pinned) {
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        g_cache.setup();
        for (size_t k = 0; k < g_cache.queues.size(); k++) if (g_cache.queues[k].device == device) {
            *st = g_cache.queues[k].stream; *e0 = g_cache.queues[k].ev0; *e1 = g_cache.queues[k].ev1; *pinned = g_cache.queues[k].pinned;
            g_cache.queues.erase(g_cache.queues.begin() + k);
            return hipSuccess;
        }
    }
    // non-blocking: a plain stream would synchronise with the legacy default stream, i.e. with every synchronous hipMemcpy / hipMemset any
    // other host thread issues -- engines that run side by side (batch driver) then wait for each other's kernels
    hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(e0);
    if (e == hipSuccess) e = hipEventCreate(e1);
    if (e == hipSuccess && hipHostMalloc(pinned, PINNED_BYTES) != hipSuccess) *pinned = nullptr;   // optional: callers fall back to copies
    return e;
}
void cache_queue_return(int device, hipStream_t st, hipEvent_t e0, hipEvent_t e1, void* pinned) {
    if (st && e0 && e1) {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        if (!g_cache.off && g_cache.queues.size() < 64) { g_cache.queues.push_back({device, st, e0, e1, pinned}); return; }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (pinned) (void)hipHostFree(pinned);
}
// auxiliary streams of an engine, pooled like the rest. kind 0: high priority, for the small latency-critical kernels (ICP 1-NN, point
// evaluation). kind 1: the bulk stream of cavity-sized engines' search launches (k_inner_bnb, hundreds to thousands of workgroups that hold a
// CU's whole register file two at a time), confined to all but GOICP_CU_RESERVE (default 16) CUs -- beside other engines' launches the small
// kernels otherwise wait for search workgroups to retire (configs[3] batch, 16 registrations in flight: 313 -> 390 pairs/s).
static hipError_t cache_stream_aux(int device, int kind, hipStream_t* st) {
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        for (size_t k = 0; k < g_cache.aux_streams.size(); k++) if (g_cache.aux_streams[k].device == device && g_cache.aux_streams[k].kind == kind) {
            *st = g_cache.aux_streams[k].stream; g_cache.aux_streams.erase(g_cache.aux_streams.begin() + k); return hipSuccess;
        }
    }
    if (kind == 0) {
        int lo = 0, hi = 0;                                    // (numerically lower = higher priority)
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        if (hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    }
    int reserve = 16, ncu = 0;
    { const char* r = getenv("GOICP_CU_RESERVE"); if (r) reserve = atoi(r); }
    if (reserve > 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && 2 * reserve <= ncu) {
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int c = 0; c < ncu - reserve; c++) mask[(size_t)c / 32] |= 1u << (c % 32);
        if (hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();                               // (a runtime without CU masks: an ordinary stream does the job, only slower in a batch)
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
static void cache_stream_aux_return(int device, int kind, hipStream_t st) {
    if (!st) return;
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        if (!g_cache.off && g_cache.aux_streams.size() < 128) { g_cache.aux_streams.push_back({device, kind, st}); return; }
    }
    (void)hipStreamDestroy(st);
}
extern "C" void goicp_engine_release_cached_memory(void) {
    std::vector<void*> blocks;
    std::vector<DevCache::Queue> qs;
    std::vector<DevCache::Aux> aux;
    { std::lock_guard<std::mutex> lk(g_cache.mu); for (auto& kv : g_cache.parked) blocks.push_back(kv.second); g_cache.parked.clear(); g_cache.parked_bytes = 0; qs.swap(g_cache.queues); aux.swap(g_cache.aux_streams); }
    for (auto& q : aux) (void)hipStreamDestroy(q.stream);
    for (void* p : blocks) (void)hipFree(p);
    for (auto& q : qs) { (void)hipEventDestroy(q.ev0); (void)hipEventDestroy(q.ev1); (void)hipStreamDestroy(q.stream); if (q.pinned) (void)hipHostFree(q.pinned); }
}

static int fail(goicp_engine* h, int code, const std::string& msg) { if (h) h->err = msg; return code; }
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, GOICP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// host wait for everything queued on `st` so far
static hipError_t wait_stream(goicp_engine* h, hipStream_t st) {
    if (h->wait_block && h->ev_wait) {
        hipError_t e = hipEventRecord(h->ev_wait, st);
        return e == hipSuccess ? hipEventSynchronize(h->ev_wait) : e;
    }
    return hipStreamSynchronize(st);
}

template <class Tp>
static int dev_upload(goicp_engine* h, const Tp* src, size_t n, Tp** out) {
    Tp* p = nullptr;
    HIPCHK(h, cache_malloc((void**)&p, std::max<size_t>(n, 1) * sizeof(Tp), h->device));
    h->allocs.push_back(p);
    if (src && n) {
        // on the engine's own (non-blocking) stream: a copy on the default stream would wait for the kernels of every other engine in the
        // process. goicp_engine_create keeps every source alive and waits ONCE at its end (upload_defer); elsewhere the copy is waited for here.
        HIPCHK(h, hipMemcpyAsync(p, src, n * sizeof(Tp), hipMemcpyHostToDevice, h->stream));
        if (!h->upload_defer) HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    *out = p;
    return GOICP_OK;
}

static bool known_prop(int c) {   // keys of the identity compatibility map, reference jly_goicp.cpp:66-73
    return c == 8204959 || c == 30894 || c == 15219528 || c == 15231913 || c == 4646984 || c == 16741671 || c == 7566712 || c == 0;
}

extern "C" const char* goicp_engine_last_error(const goicp_engine* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" void goicp_engine_destroy(goicp_engine* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);          // parked blocks must be idle
    if (h->stream_hi) { (void)hipStreamSynchronize(h->stream_hi); cache_stream_aux_return(h->device, 0, h->stream_hi); }
    if (h->stream_bulk) { (void)hipStreamSynchronize(h->stream_bulk); cache_stream_aux_return(h->device, 1, h->stream_bulk); }
    if (h->ev_wait) (void)hipEventDestroy(h->ev_wait);
    for (void* p : h->allocs) cache_free(p);
    if (h->d_heaps) cache_free(h->d_heaps);
    if (h->d_heap_big) cache_free(h->d_heap_big);
    if (h->T.srot) cache_free(h->T.srot);
    if (h->T.srows) cache_free(h->T.srows);
    if (h->d_fpscratch) cache_free(h->d_fpscratch);
    { void* fr[] = {h->fr_st, h->fr_heaps, h->fr_exps, h->fr_wl, h->fr_cnt, h->fr_parts, h->fr_srot, h->fr_results}; for (void* p : fr) if (p) cache_free(p); }
    cache_queue_return(h->device, h->stream, h->ev0, h->ev1, h->pinned);   // (already synchronised above)
    delete h;
}

#define DISPATCH_GEOM(h, CALL)                                               \
    do {                                                                     \
        if (h->block == 256 && h->ppt == 1) { CALL(256, 1); }                \
        else if (h->block == 256 && h->ppt == 2) { CALL(256, 2); }           \
        else if (h->block == 512 && h->ppt == 1) { CALL(512, 1); }           \
        else if (h->block == 1024 && h->ppt == 1) { CALL(1024, 1); }         \
        else if (h->block == 1024 && h->ppt == 2) { CALL(1024, 2); }         \
        else if (h->block == 1024 && h->ppt == 4) { CALL(1024, 4); }         \
        else if (h->block == 1024 && h->ppt == 5) { CALL(1024, 5); }         \
        else { CALL(512, 16); }                                              \
    } while (0)

// The dynamic-LDS ceiling of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-process property of the instantiation, and
// engines of different cloud sizes share instantiations while they run side by side on host threads (batch driver): it is therefore
// raised ONCE per instantiation to everything the CU has left beside the kernel's static LDS, never to one engine's own need.
static hipError_t set_max_dyn_lds(const void* fn) {
    static std::mutex mu; static std::map<std::pair<int, const void*>, int> done;      // per (device, instantiation)
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const std::pair<int, const void*> key(dev, fn);
    if (done.count(key)) return hipSuccess;
    hipFuncAttributes a;
    e = hipFuncGetAttributes(&a, fn);
    if (e != hipSuccess) return e;
    const int room = 160 * 1024 - (int)a.sharedSizeBytes;
    if (room <= 0) return hipErrorInvalidValue;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, room);
    if (e == hipSuccess) done[key] = room;
    return e;
}
static bool env_is_zero(const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; }

static void pick_geometry(goicp_engine* h) {
    // lanes x points-per-lane; the residuals of a tile (TC x PPT floats per lane) stay in registers across the
    // selection passes, so large clouds use 512-lane workgroups (256 VGPRs per lane) instead of 1024 (128 VGPRs)
    const int Nd = h->T.Nd;
    // above 5120 points the register-resident geometry would be 512 lanes x 16 points (spills, 66-81 G lookups/s measured);
    // the streamed sweep does 109-118 G lookups/s there, so geometric runs switch to it early. Penalty runs keep registers up to 8192.
    h->streamed = Nd > 8192 || (Nd > 5120 && !h->T.use_compat && !h->T.use_fpfh);
    if (h->streamed) { h->block = 1024; h->ppt = 1; }          // cloud streamed from HBM per tile (sweep_stream)
    // trimmed runs in parity mode: one wave per cube row runs the reference's selection, and at 256 lanes a wave would take two rows one after the
    // other -- 512 lanes give each of the 8 rows a wave of its own (one search per CU instead of two, each pop about twice as fast)
    else if (h->T.exact && h->T.doTrim && Nd <= 512 && !env_is_zero("GOICP_TRIM512")) { h->block = 512; h->ppt = 1; }
    else if (Nd <= 256) { h->block = 256; h->ppt = 1; }
    else if (Nd <= 512) { h->block = 256; h->ppt = 2; }
    else if (Nd <= 1024) { h->block = 1024; h->ppt = 1; }
    else if (Nd <= 2048) { h->block = 1024; h->ppt = 2; }
    else if (Nd <= 4096) { h->block = 1024; h->ppt = 4; }
    else if (Nd <= 5120) { h->block = 1024; h->ppt = 5; }
    else { h->block = 512; h->ppt = 16; }
    h->dyn_lds = h->T.exact ? (size_t)2 * TC * Nd * sizeof(float) : 0;      // ub and lb addend rows (untrimmed) / residual rows (trimmed)
    // k_inner_bnb keeps the top of its queue in LDS behind the residual rows (24 B per node)
    h->rbuf_floats = (int)(h->dyn_lds / sizeof(float));
    h->heap_lds_nodes = 512;              // 12 KB (1024 and 2048 measured no faster): with the residual rows of a cloud of up to ~450
                                          // points and the tile state a search stays below 80 KB of LDS, so two share a CU
    { const char* e = getenv("GOICP_HEAP_LDS"); if (e && atoi(e) >= 0 && atoi(e) <= 2048) h->heap_lds_nodes = atoi(e); }   // test knob
    h->dyn_lds_bnb = h->dyn_lds + (size_t)h->heap_lds_nodes * sizeof(HeapNode);
    h->fp_lds = 0;
    // parity mode (Nd <= 1024) leaves the 32 KB of select histograms unused: the 27 c-FPFH rows of a cavity cloud (Nd <= 303) fit
    // there, which keeps a search at <= 80 KB of LDS, i.e. two searches per CU
    if (h->T.use_fpfh && h->T.exact && (size_t)27 * Nd * sizeof(float) <= sizeof(unsigned) * TC * 1024) h->fp_lds = 2;
    else if (h->T.use_fpfh) {
        // larger clouds: rows in LDS only if that does not cost the second search per CU (measured on a 379-point cloud with
        // 37 M cube evaluations: rows in global scratch + two searches per CU 0.77 s, rows in LDS + one search per CU 0.89 s)
        const size_t tile_state = h->block == 256 ? sizeof(TileShm<256>) : (h->block == 512 ? sizeof(TileShm<512>) : sizeof(TileShm<1024>));
        const bool two_without_rows = h->block == 256 && h->dyn_lds_bnb + tile_state + 256 <= 80 * 1024;
        if (!two_without_rows && h->dyn_lds_bnb + (size_t)27 * Nd * sizeof(float) + sizeof(TileShm<1024>) + 1024 <= 160 * 1024) { h->fp_lds = 1; h->dyn_lds_bnb += (size_t)27 * Nd * sizeof(float); }
    }
}

extern "C" int goicp_engine_create(const goicp_engine_desc* d, goicp_engine** out) {
    if (!d || !out) return GOICP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || d->device_id < 0 || d->device_id >= ndev) return GOICP_ERR_NO_DEVICE;
    if (d->S <= 0 || d->S > GOICP_MAX_GRID || (!d->dt_dist && !d->dt_dist_dev) || d->Nd <= 0 || !d->data_xyz || !d->weights || !d->maxRotDis || d->Nm < 0) return GOICP_ERR_ARG;
    if (d->regularizationNeighbors > 0.f) return GOICP_ERR_ARG;
    if (d->Nd > (1 << 20)) return GOICP_ERR_ARG;  // up to 8192 points the cloud is register-resident; above that it is streamed (geometric error only)
    if (d->doTrim && (d->inlierNum < 1 || d->inlierNum > d->Nd)) return GOICP_ERR_ARG;
    const bool use_compat = d->regularization > 0.f;
    const bool use_fpfh = d->regularizationFPFH > 0.f;        // entered for any cfpfh once regularizationFPFH>0 (jly_goicp.cpp:460)
    if (d->Nd > 8192 && (use_compat || use_fpfh)) return GOICP_ERR_ARG;   // streamed clouds: geometric error only (the reference's own penalty constant overflows int above 46 340 points)
    const bool sparse = d->dt_dist_dev != nullptr;             // device-resident grid + sparse cell tables (goicp_dtgpu_build)
    const bool have_tables = d->cell_idx && d->model_color && d->data_color &&
                             (sparse ? (d->nearest_occ_dev && d->n_occ > 0 && d->occ_cells && d->occ_color && d->occ_ptr) : (d->nearest_occ && d->cell_color && d->cell_ptr));
    if ((use_compat || use_fpfh) && !have_tables) return GOICP_ERR_ARG;
    if (use_fpfh && (!d->model_fpfh || !d->data_fpfh)) return GOICP_ERR_ARG;

    goicp_engine* h = new goicp_engine();
    memset(&h->T, 0, sizeof h->T);
    h->device = d->device_id; h->stream = nullptr; h->ev0 = h->ev1 = nullptr; h->stream_hi = nullptr; h->stream_bulk = nullptr; h->ev_wait = nullptr; h->wait_block = 0; h->upload_defer = 1;
    h->streamed = 0; h->stream_wgs = 0; h->stream_rot_wgs = 0; h->pinned = nullptr;
    h->d_heaps = nullptr; h->d_fpscratch = nullptr; h->heap_cap = 8192; h->heap_jobs = 0; h->fp_blocks = 0; h->d_heap_big = nullptr; h->heap_big_cap = 0;
    h->n_cubes = h->n_lookups = 0; h->kernel_ms = 0; h->max_pops = 200000000u;
    h->code_escapes = 0; h->last_fused = 0;
    h->fr_mode = 0; h->fr_st = nullptr; h->fr_heaps = nullptr; h->fr_exps = nullptr; h->fr_wl = nullptr; h->fr_cnt = nullptr; h->fr_parts = nullptr; h->fr_srot = nullptr; h->fr_results = nullptr;
    h->fr_jobs = 0; h->fr_cap_nodes = 0; h->fr_cap_exps = 0; h->fr_wl_cap = 0; h->fr_rounds = 0; h->fr_spec_cubes = 0;
#define CR(call) do { int rc_ = (call); if (rc_ != GOICP_OK) { std::string m = h->err; goicp_engine_destroy(h); fprintf(stderr, "goicp_engine_create: %s\n", m.c_str()); return rc_; } } while (0)
#define CRH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "goicp_engine_create: %s: %s\n", #call, hipGetErrorString(e_)); goicp_engine_destroy(h); return GOICP_ERR_NO_DEVICE; } } while (0)
    {   // experiment knob: how this process's host threads wait for the device (spin is the runtime's default with idle cores)
        static std::once_flag sched_once;
        std::call_once(sched_once, [] {
            const char* e = getenv("GOICP_SCHED");
            if (e && !strcmp(e, "yield")) (void)hipSetDeviceFlags(hipDeviceScheduleYield);
            else if (e && !strcmp(e, "block")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
            (void)hipGetLastError();
        });
    }
    CRH(hipSetDevice(h->device));
    CRH(cache_queue(h->device, &h->stream, &h->ev0, &h->ev1, &h->pinned));
    CRH(cache_stream_aux(h->device, 0, &h->stream_hi));
    { const char* e = getenv("GOICP_WAIT"); h->wait_block = (e && !strcmp(e, "block")) ? 1 : 0; }
    if (h->wait_block) CRH(hipEventCreateWithFlags(&h->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
    DevTables& T = h->T;
    T.S = d->S; T.xMin = d->xMin; T.yMin = d->yMin; T.zMin = d->zMin; T.scale = d->scale;
    T.Nd = d->Nd; T.Nm = d->Nm; T.doTrim = d->doTrim ? 1 : 0; T.inlierNum = d->doTrim ? d->inlierNum : d->Nd; T.norm = d->norm;
    T.b0 = 0; T.b1 = 0;
    if (d->cfpfh == 1) { T.b0 = 0; T.b1 = 41; } else if (d->cfpfh == 2) { T.b0 = 0; T.b1 = 33; } else if (d->cfpfh == 3) { T.b0 = 33; T.b1 = 41; }
    T.reg = d->regularization; T.regFPFH = d->regularizationFPFH;
    T.use_compat = use_compat; T.use_fpfh = use_fpfh;
    { const char* e = getenv("GOICP_DBG"); T.dbg = e ? atoi(e) : 0; }
    { const char* e = getenv("GOICP_MAX_POPS"); if (e) h->max_pops = (unsigned)atoll(e); }
    { const char* e = getenv("GOICP_HEAP_CAP"); if (e && atoi(e) >= 16) h->heap_cap = atoi(e); }      // test knob: small queue slabs
    T.NdP = (d->Nd + 1023) & ~1023; T.srot = nullptr; T.srows = nullptr;
    T.exact = (d->parity_mode && d->Nd <= 1024) ? 1 : 0;   // 2*TC*Nd floats of LDS next to the static tile state and the LDS heap
    // parity_mode 2: reference-order float sums / the reference's intro_select also for clouds of 1025..5120 points (geometric error), on rows in HBM
    T.exact_big = (d->parity_mode == 2 && d->Nd > 1024 && d->Nd <= 5120 && !use_compat && !use_fpfh) ? 1 : 0;
    T.exact_stream = (d->parity_mode == 2 && d->Nd > 5120 && !use_compat && !use_fpfh) ? 1 : 0;      // (geometric runs above 5120 points are streamed, pick_geometry)
    T.xrows = nullptr; T.xslots = 0;
    if (T.exact_big) { T.xslots = 1024; CR(dev_upload<float>(h, nullptr, (size_t)T.xslots * 2 * TC * T.NdP, &T.xrows)); }
    const size_t ncell = (size_t)d->S * d->S * d->S;
    if (sparse) T.dist = (const float*)d->dt_dist_dev;        // adopted in place (owned by the caller's goicp_dtgpu handle)
    else { float* p_dist; CR(dev_upload(h, d->dt_dist, ncell, &p_dist)); T.dist = p_dist; }
    T.code = nullptr; T.lut = nullptr;
    std::vector<float> lut;                     // (sources of deferred uploads live to the end of this function)
    std::vector<int> occ_cells, occ_ptr(1, 0), occ_pts;
    std::vector<uint16_t> occ_mask;
    {   // 2-byte coded copy of the grid for the gather stream (grids of 100^3 cells and more; GOICP_CODED=0 switches it off)
        const char* e = getenv("GOICP_CODED");
        if (d->S >= 100 && !(e && atoi(e) == 0)) {
            const long long full = 3ll * (d->S - 1) * (d->S - 1) + 1;
            const int lut_n = (int)std::min<long long>(full, 65535);
            lut.assign((size_t)std::max(lut_n, LUT_LDS), 0.f);
            for (int n = 0; n < lut_n; n++) lut[(size_t)n] = (float)((double)(float)sqrt((double)n) / d->scale);     // jly_3ddt.cpp:28,79,1003
            float* p_lut; CR(dev_upload(h, lut.data(), lut.size(), &p_lut));
            uint16_t* p_code; CR(dev_upload<uint16_t>(h, nullptr, ncell, &p_code));
            unsigned long long* p_esc; CR(dev_upload<unsigned long long>(h, nullptr, 1, &p_esc));
            CRH(hipMemsetAsync(p_esc, 0, sizeof(unsigned long long), h->stream));
            hipLaunchKernelGGL(k_build_codes, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, h->stream, T.dist, p_lut, lut_n, d->scale, (unsigned long long)ncell, p_code, p_esc);
            CRH(hipGetLastError());
            unsigned long long esc = 0;
            CRH(hipMemcpyAsync(&esc, p_esc, sizeof esc, hipMemcpyDeviceToHost, h->stream));
            CRH(hipStreamSynchronize(h->stream));
            h->code_escapes = esc;
            if ((double)esc <= 0.25 * (double)ncell) { T.code = p_code; T.lut = p_lut; }      // mostly escapes: the float grid alone is the better stream
        }
    }
    float* p_dxyz; CR(dev_upload(h, d->data_xyz, (size_t)3 * d->Nd, &p_dxyz)); T.dxyz = p_dxyz;
    float* p_w; CR(dev_upload(h, d->weights, (size_t)d->Nd, &p_w)); T.weights = p_w;
    float* p_mrd; CR(dev_upload(h, d->maxRotDis, (size_t)GOICP_MAXROTLEVEL * d->Nd, &p_mrd)); T.maxRotDis = p_mrd;
    CR(dev_upload(h, d->model_xyz, (size_t)3 * d->Nm, &h->d_mxyz));

    // colour classes: the 8 known properties first, then whatever else appears (<= 16 classes in all)
    std::map<int, int> cls;
    auto cls_of = [&](int c) -> int {
        auto it = cls.find(c);
        if (it != cls.end()) return it->second;
        int id = (int)cls.size(); cls[c] = id; return id;
    };
    std::vector<uint8_t> dcls(d->Nd, 0);
    if (d->data_color) for (int i = 0; i < d->Nd; i++) dcls[i] = (uint8_t)cls_of(d->data_color[i]);
    if (have_tables) {      // the compat mask also serves goicp_engine_point_eval when no penalty is active
        std::vector<int> mcls(d->Nm);
        for (int i = 0; i < d->Nm; i++) mcls[i] = cls_of(d->model_color[i]);
        if (cls.size() > 16) { h->err = "more than 16 distinct colour codes"; CR(GOICP_ERR_ARG); }
        // compat mask per cell, resolving emptyCells -> cellPoints (reference jly_goicp.cpp:986-1000,1068-1092): the per-occupied-
        // cell masks are formed on the host (a few hundred cells), the S^3-sized gather through nearest_occ runs on the device
        auto add_cell = [&](int k, int col, int p0, int p1) {
            occ_cells.push_back(k);
            uint16_t m = 0;
            if (col == -1) { for (int p = p0; p < p1; p++) m |= (uint16_t)(1u << mcls[d->cell_idx[p]]); }
            else if (known_prop(col)) { auto it = cls.find(col); if (it != cls.end()) m = (uint16_t)(1u << it->second); }
            occ_mask.push_back(m);
            for (int p = p0; p < p1; p++) occ_pts.push_back(d->cell_idx[p]);
            occ_ptr.push_back((int)occ_pts.size());
        };
        if (sparse) { for (int o = 0; o < d->n_occ; o++) add_cell(d->occ_cells[o], d->occ_color[o], d->occ_ptr[o], d->occ_ptr[o + 1]); }
        else {
            for (size_t k = 0; k < ncell; k++) {
                const int col = d->cell_color[k];
                if (col == -2) continue;
                add_cell((int)k, col, d->cell_ptr[k], d->cell_ptr[k + 1]);
            }
        }
        const int nOcc = (int)occ_cells.size();
        int* p_near;
        if (sparse) p_near = (int*)d->nearest_occ_dev;            // read-only here
        else CR(dev_upload(h, d->nearest_occ, ncell, &p_near));
        int* p_occid; CR(dev_upload<int>(h, nullptr, ncell, &p_occid));
        int* p_cells; CR(dev_upload(h, occ_cells.data(), (size_t)nOcc, &p_cells));
        uint16_t* p_om; CR(dev_upload(h, occ_mask.data(), (size_t)nOcc, &p_om));
        uint16_t* p_cm; CR(dev_upload<uint16_t>(h, nullptr, ncell, &p_cm));
        int* p_occ = nullptr;
        if (use_fpfh) CR(dev_upload<int>(h, nullptr, ncell, &p_occ));
        CRH(hipMemsetAsync(p_occid, 0xFF, ncell * sizeof(int), h->stream));
        if (nOcc) hipLaunchKernelGGL(k_scatter_occ, dim3((nOcc + 255) / 256), dim3(256), 0, h->stream, p_cells, nOcc, p_occid);
        hipLaunchKernelGGL(k_build_cmask, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, h->stream, p_near, p_occid, p_om, (unsigned long long)ncell, p_cm, p_occ);
        CRH(hipGetLastError());
        {   // nearest_occ range check happened on the device (out-of-range entries resolve to "no occupied cell")
            T.cmask = p_cm;
        }
        if (use_fpfh) {
            T.occ = p_occ;
            int* p_op; CR(dev_upload(h, occ_ptr.data(), occ_ptr.size(), &p_op)); T.occ_ptr = p_op;
            int* p_oi; CR(dev_upload(h, occ_pts.data(), occ_pts.size(), &p_oi)); T.occ_pts = p_oi;
            float* p_mf; CR(dev_upload(h, d->model_fpfh, (size_t)GOICP_FPFH_BINS * d->Nm, &p_mf)); T.mfpfh = p_mf;
            float* p_df; CR(dev_upload(h, d->data_fpfh, (size_t)GOICP_FPFH_BINS * d->Nd, &p_df)); T.dfpfh = p_df;
            // pose-independent table of descriptor distances (data point x occupied cell), built once on the device
            T.nOcc = (int)occ_ptr.size() - 1;
            const size_t ne = (size_t)d->Nd * (size_t)std::max(T.nOcc, 1);
            float* p_E; CR(dev_upload<float>(h, nullptr, ne, &p_E));
            if (T.nOcc > 0) {
                hipLaunchKernelGGL(k_fpfh_table, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream, T, p_E);
                CRH(hipGetLastError());
            }
            T.fpE = p_E;
        }
    }
    if (cls.size() > 16) { h->err = "more than 16 distinct colour codes"; CR(GOICP_ERR_ARG); }
    uint8_t* p_dc; CR(dev_upload(h, dcls.data(), (size_t)d->Nd, &p_dc)); T.dcls = p_dc;

    h->nslots = 16;
    CR(dev_upload<float4>(h, nullptr, (size_t)h->nslots * d->Nd, &h->d_rot));
    CR(dev_upload<float>(h, nullptr, 16, &h->d_R));
    h->job_cap = 4096;
    CR(dev_upload<goicp_bnb_job>(h, nullptr, h->job_cap, &h->d_jobs));
    CR(dev_upload<goicp_bnb_result>(h, nullptr, h->job_cap, &h->d_results));
    CR(dev_upload<int2>(h, nullptr, 4096, &h->d_groups));
    CR(dev_upload<int4>(h, nullptr, 2048, &h->d_pairs)); h->npairs = 0;
    CR(dev_upload<int4>(h, nullptr, 4096, &h->d_groups4));
    { const char* e = getenv("GOICP_FUSE"); h->fuse = !(e && atoi(e) == 0); }      // A/B knob, read once per engine
    { const char* e = getenv("GOICP_XCD"); h->xcd_mode = e ? std::max(0, std::min(atoi(e), 2)) : 1; }   // 1 measured 1.5-1.7 % faster than 0, 2 slower (tools/gather_modes.py)
    h->cube_cap = 1 << 16;
    CR(dev_upload<float>(h, nullptr, (size_t)4 * h->cube_cap, &h->d_cubes));
    CR(dev_upload<float>(h, nullptr, h->cube_cap, &h->d_ub));
    CR(dev_upload<float>(h, nullptr, h->cube_cap, &h->d_lb));
    CR(dev_upload<float>(h, nullptr, d->Nd, &h->d_dist));
    CR(dev_upload<int>(h, nullptr, d->Nd, &h->d_int));
    CR(dev_upload<int>(h, nullptr, d->Nd, &h->d_tie));
    h->lat_cap = 1 << 14;
    CR(dev_upload<float>(h, nullptr, (size_t)3 * h->lat_cap, &h->d_lat));
    CR(dev_upload<int>(h, nullptr, h->lat_cap, &h->d_cnt));
    CR(dev_upload<float>(h, nullptr, h->lat_cap, &h->d_fp));
    { unsigned long long* p_t; CR(dev_upload<unsigned long long>(h, nullptr, 16, &p_t)); CRH(hipMemsetAsync(p_t, 0, 128, h->stream)); T.dbg_times = p_t; }
    pick_geometry(h);
    h->jit_eval = (h->block == 1024 && h->ppt >= 2 && !T.exact && !T.use_compat && !T.use_fpfh && !env_is_zero("GOICP_JIT")) ? 1 : 0;
    if (h->dyn_lds_bnb + sizeof(TileShm<1024>) + 64 > 160 * 1024) { h->err = "LDS budget exceeded"; CR(GOICP_ERR_ARG); }
    {
        // static tile state + residual rows + LDS heap can exceed the 64 KB a launch gets without asking
#define SETATTR(B, P) do { CRH(set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<B, P, true>))); \
                           CRH(set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<B, P, false>))); \
                           if (h->dyn_lds) CRH(set_max_dyn_lds(reinterpret_cast<const void*>(&k_eval_cubes<B, P, TC>))); } while (0)
        DISPATCH_GEOM(h, SETATTR);
#undef SETATTR
        if (h->streamed) CRH(set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<1024, 1, false, true>)));
        // measured (tools/mid_perf.py): 2 000 points 15.2 -> 12.9 ms of kernel time, 4 000 points no change, 5 000 points 7 % slower
        // (122 spilled registers), so only the 1025..2048-point geometry switches. GOICP_JIT_BNB=0 disables.
        h->jit_bnb = (!h->streamed && h->block == 1024 && h->ppt == 2 && !T.exact && !T.use_compat && !T.use_fpfh && !env_is_zero("GOICP_JIT_BNB")) ? 1 : 0;
        if (h->jit_bnb) CRH(set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<512, 4, false, false, true>)));
        {
            int ncu = 0; CRH(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device));   // (a full hipGetDeviceProperties costs ~1 ms per engine)
            int nb = 0;
            const bool pen = T.use_compat || T.use_fpfh;
#define OCC(B, P) do { if (pen) CRH(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<B, P, true>, B, h->dyn_lds_bnb)); \
                       else CRH(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<B, P, false>, B, h->dyn_lds_bnb)); } while (0)
            if (h->streamed) CRH(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<1024, 1, false, true>, 1024, h->dyn_lds_bnb));
            else if (h->jit_bnb) CRH(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<512, 4, false, false, true>, 512, h->dyn_lds_bnb));
            else DISPATCH_GEOM(h, OCC);
#undef OCC
            h->resident_jobs = std::max(1, nb) * std::max(1, ncu);
            // searches of more than 1024 points without penalty terms run round by round over many workgroups (k_fr_*)
            h->fr_mode = 0; h->fr_nsplit = 1; h->fr_ppt = 4; h->fr_block = 512; h->fr_wgs = 2 * std::max(1, ncu); h->fr_heap_cap = 8192;
            h->fr_depth2 = (getenv("GOICP_FR_DEPTH") && atoi(getenv("GOICP_FR_DEPTH")) == 2) ? 1 : 0;   // two-level expansion: measured slower on configs[2] (more sweeps per round than rounds saved); knob kept for deep searches
            const bool fr_off = getenv("GOICP_FR") && atoi(getenv("GOICP_FR")) == 0;          // A/B knob, read once per engine
            if ((!fr_off || T.exact_big || T.exact_stream) && !T.exact && !T.use_compat && !T.use_fpfh && d->Nd > 1024) {
                if (!h->streamed) { h->fr_mode = 1; h->fr_block = 512; h->fr_ppt = d->Nd <= 2048 ? 4 : (d->Nd <= 4096 ? 8 : 10); h->fr_wgs = 2 * std::max(1, ncu); }
                else if (T.doTrim || T.exact_stream) { h->fr_mode = 2;      // (parity_mode 2: whole tiles, the sums run over all points in index order)
                    h->fr_block = d->Nd > 16384 ? 1024 : 512; h->fr_wgs = (h->fr_block == 512 ? 2 : 1) * std::max(1, ncu); }
                else { h->fr_mode = 3; h->fr_block = 1024; h->fr_nsplit = std::min(16, std::max(1, (d->Nd + 6143) / 6144)); h->fr_wgs = std::max(1, ncu); }
                { const char* e = getenv("GOICP_HEAP_CAP"); if (e && atoi(e) >= 16) h->fr_heap_cap = atoi(e); }
                h->resident_jobs = std::max(h->resident_jobs, 32);
            }
        }
    }
    CRH(hipStreamSynchronize(h->stream));           // every deferred upload and set-up kernel
    h->upload_defer = 0;
#undef CR
#undef CRH
    *out = h;
    return GOICP_OK;
}

extern "C" int goicp_engine_debug_times(goicp_engine* h, unsigned long long* out8) {
    if (!h || !out8) return GOICP_ERR_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyAsync(out8, h->T.dbg_times, 128, hipMemcpyDeviceToHost, h->stream));      /* 16 words */
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GOICP_OK;
}

extern "C" int goicp_engine_last_eval_fused(const goicp_engine* h) { return h ? h->last_fused : 0; }
extern "C" int goicp_engine_num_slots(const goicp_engine* h) { return h ? h->nslots : 0; }
extern "C" int goicp_engine_resident_jobs(const goicp_engine* h) { return h ? h->resident_jobs : 0; }
extern "C" void* goicp_engine_stream(const goicp_engine* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int goicp_engine_synchronize(goicp_engine* h) {
    if (!h) return GOICP_ERR_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GOICP_OK;
}

extern "C" int goicp_engine_stats(const goicp_engine* h, uint64_t* cube_evals, uint64_t* lookups, double* kernel_ms) {
    if (!h) return GOICP_ERR_ARG;
    if (cube_evals) *cube_evals = h->n_cubes;
    if (lookups) *lookups = h->n_lookups;
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    return GOICP_OK;
}

extern "C" int goicp_engine_search_stats(const goicp_engine* h, uint64_t* rounds, uint64_t* cube_evals_launched) {
    if (!h) return GOICP_ERR_ARG;
    if (rounds) *rounds = h->fr_rounds;
    if (cube_evals_launched) *cube_evals_launched = h->fr_spec_cubes;
    return GOICP_OK;
}

extern "C" int goicp_engine_set_rotation(goicp_engine* h, int slot, const float* R9) {
    if (!h || !R9 || slot < 0 || slot >= h->nslots) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_R, R9, 9 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_rotate, dim3((h->T.Nd + 255) / 256), dim3(256), 0, h->stream, h->T, h->d_R, h->d_rot + (size_t)slot * h->T.Nd);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));      // d_R is reused by the next call
    return GOICP_OK;
}

static int ensure_fpscratch(goicp_engine* h, size_t blocks) {
    if (!h->T.use_fpfh) return GOICP_OK;
    if (blocks <= h->fp_blocks) return GOICP_OK;
    if (h->d_fpscratch) { HIPCHK(h, hipStreamSynchronize(h->stream)); cache_free(h->d_fpscratch); h->d_fpscratch = nullptr; h->fp_blocks = 0; }
    HIPCHK(h, cache_malloc((void**)&h->d_fpscratch, blocks * NLMAX * (size_t)h->T.Nd * sizeof(float), h->device));
    h->fp_blocks = blocks;
    return GOICP_OK;
}

static int ensure_stream_scratch(goicp_engine* h, size_t wgs, bool need_rot, bool need_rows = true) {
    // per resident workgroup: rotated cloud (search kernel only) + TC residual rows (trimmed runs only)
    if (need_rot && wgs > h->stream_rot_wgs) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->T.srot) { cache_free(h->T.srot); h->T.srot = nullptr; }
        h->stream_rot_wgs = 0;
        HIPCHK(h, cache_malloc((void**)&h->T.srot, wgs * (size_t)h->T.NdP * sizeof(float4), h->device));
        h->stream_rot_wgs = wgs;
    }
    if (need_rows && (h->T.doTrim || h->T.exact_stream) && wgs > h->stream_wgs) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->T.srows) { cache_free(h->T.srows); h->T.srows = nullptr; }
        h->stream_wgs = 0;
        HIPCHK(h, cache_malloc((void**)&h->T.srows, wgs * (size_t)(h->T.exact_stream ? 2 : 1) * TC * h->T.NdP * sizeof(float), h->device));
        h->stream_wgs = wgs;
    }
    return GOICP_OK;
}

static int launch_eval(goicp_engine* h, int slot, int rot_level, const int2* groups_dev, int ngroups, const void* cubes_dev, int n, void* ub_dev, void* lb_dev) {
    if (h->streamed) {
        // untrimmed runs: group pairs (no radius / radius, same slot) of a multi-group call are evaluated from one pass of gathers when all pair up
        const bool dual = !h->T.doTrim && !h->T.exact_stream && groups_dev == h->d_groups && h->npairs > 0 && h->npairs * 2 == ngroups;
        const int ng = dual ? h->npairs : ngroups;
        const int4* ptab = dual ? h->d_pairs : nullptr;
        h->last_fused = dual ? 1 : 0;
        const long long total = (long long)((n + TC - 1) / TC) * ng;
        const int wgs = (int)std::min<long long>(total, 1024);         // persistent workgroups: a few per CU (two 512-lane ones are resident)
        int rcs = ensure_stream_scratch(h, (size_t)wgs, false);
        if (rcs != GOICP_OK) return rcs;
        // two 512-lane workgroups per CU overlap gathering and selection (+12 % at 6 000 points); from 16 K points on the sweep is
        // bound by the row and grid traffic itself and the 1024-lane form is as fast or faster
        if (h->T.Nd > 16384) hipLaunchKernelGGL(k_eval_cubes_stream<1024>, dim3(wgs), dim3(1024), 0, h->stream, h->T, h->d_rot, slot, rot_level, groups_dev, ng, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, ptab);
        else hipLaunchKernelGGL(k_eval_cubes_stream<512>, dim3(wgs), dim3(512), 0, h->stream, h->T, h->d_rot, slot, rot_level, groups_dev, ng, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, ptab);
        HIPCHK(h, hipGetLastError());
        h->n_cubes += (uint64_t)n * ngroups; h->n_lookups += (uint64_t)n * ngroups * h->T.Nd;
        return GOICP_OK;
    }
    h->last_fused = 0;
    const int tiles = (n + TC - 1) / TC;
    int rc = ensure_fpscratch(h, (size_t)tiles * ngroups);
    if (rc != GOICP_OK) return rc;
#define CALL(B, P) hipLaunchKernelGGL((k_eval_cubes<B, P, TC>), dim3(tiles, ngroups), dim3(B), h->dyn_lds, h->stream, h->T, h->d_rot, slot, rot_level, groups_dev, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, h->d_fpscratch)
    // clouds of 1025..5120 points without penalty terms: 512 lanes x twice the points, the points fetched just in time instead of
    // held in registers -> 128 VGPRs, two workgroups per CU, one gathering while the other selects (3.89 -> 3.52 ms at C3)
    const bool jit = h->jit_eval != 0;
    if (jit) {
        // geometric just-in-time sweep: its own kernel; group pairs (no radius / radius, same slot) of a multi-group call fused when all pair up
        const bool dual = groups_dev == h->d_groups && h->npairs > 0 && h->npairs * 2 == ngroups;
        h->last_fused = dual ? 1 : 0;
        const int4* gtab = h->d_pairs;
        if (!dual) {
            // (single-group calls and unpaired tables: one int4 per group, staged here; the table of a repeated multi-group call is cached)
            if (!groups_dev) { h->one_group = make_int4(slot, rot_level, 0, 0); HIPCHK(h, hipMemcpyAsync(h->d_pairs + 2047, &h->one_group, sizeof(int4), hipMemcpyHostToDevice, h->stream)); gtab = h->d_pairs + 2047; }
            else gtab = h->d_groups4;
        }
        const int ng = dual ? h->npairs : ngroups, xm = h->xcd_mode;
        const long long wgs = xm == 1 ? 8ll * (((long long)tiles * ng + 7) / 8) : (xm == 2 ? 8ll * ((tiles + 7) / 8) * ng : (long long)tiles * ng);
#define CALLG(P2) do { if (dual) hipLaunchKernelGGL((k_eval_cubes_geo<P2, true>), dim3((unsigned)wgs), dim3(512), 0, h->stream, h->T, h->d_rot, gtab, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, tiles, ng, xm, 0); \
                       else hipLaunchKernelGGL((k_eval_cubes_geo<P2, false>), dim3((unsigned)wgs), dim3(512), 0, h->stream, h->T, h->d_rot, gtab, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, tiles, ng, xm, 0); } while (0)
        if (h->T.exact_big) {
            // parity_mode 2: every workgroup owns a slot of row scratch, so the launch goes out in chunks of T.xslots workgroups
            const long long total = (long long)tiles * ng;
            for (long long l0 = 0; l0 < total; l0 += h->T.xslots) {
                const unsigned cw = (unsigned)std::min<long long>(h->T.xslots, total - l0);
#define CALLX(P2) do { if (dual) hipLaunchKernelGGL((k_eval_cubes_geo<P2, true, true>), dim3(cw), dim3(512), 0, h->stream, h->T, h->d_rot, gtab, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, tiles, ng, 0, (int)l0); \
                       else hipLaunchKernelGGL((k_eval_cubes_geo<P2, false, true>), dim3(cw), dim3(512), 0, h->stream, h->T, h->d_rot, gtab, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, tiles, ng, 0, (int)l0); } while (0)
                if (h->ppt == 2) CALLX(4); else if (h->ppt == 4) CALLX(8); else CALLX(10);
#undef CALLX
            }
        } else if (h->ppt == 2) CALLG(4); else if (h->ppt == 4) CALLG(8); else CALLG(10);
#undef CALLG
        HIPCHK(h, hipGetLastError());
        h->n_cubes += (uint64_t)n * ngroups; h->n_lookups += (uint64_t)n * ngroups * h->T.Nd;
        return GOICP_OK;
    }
    DISPATCH_GEOM(h, CALL);
#undef CALL
    HIPCHK(h, hipGetLastError());
    h->n_cubes += (uint64_t)n * ngroups; h->n_lookups += (uint64_t)n * ngroups * h->T.Nd;
    return GOICP_OK;
}

extern "C" int goicp_engine_debug_gather_probe(goicp_engine* h, const void* cubes_dev, int n, void* scratch_dev, int variant, float* ms_out) {
    // diagnosis: the gathers of one goicp_engine_eval_cubes_multi_dev launch (the group table of the last such call) and nothing else
    if (!h || !cubes_dev || !scratch_dev || n <= 0 || h->groups_host.empty()) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const int tiles = (n + TC - 1) / TC, ngroups = (int)h->groups_host.size();
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
#define PROBE(B, P, M) hipLaunchKernelGGL((k_gather_probe<B, P, M>), dim3(tiles, ngroups), dim3(B), 0, h->stream, h->T, h->d_rot, h->d_groups, (const float*)cubes_dev, n, (float*)scratch_dev)
    if (variant == 0) PROBE(512, 10, 0);
    else if (variant == 1) PROBE(1024, 5, 0);
    else if (variant == 2) PROBE(512, 10, 1);
    else if (variant == 3) PROBE(512, 10, 2);
    else if (variant == 4) PROBE(512, 10, 3);
    else if (variant == 5) PROBE(512, 10, 4);
    else if (variant == 6) PROBE(512, 10, 5);
    else return GOICP_ERR_ARG;
#undef PROBE
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f; (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    if (ms_out) *ms_out = ms;
    return GOICP_OK;
}

extern "C" const char* goicp_engine_eval_kernel_name(goicp_engine* h) {
    // the instantiation launch_eval() picks for this engine, spelled as rocprofv3 prints it
    if (!h) return "";
    char buf[96];
    const bool jit = h->jit_eval != 0;
    if (h->streamed) snprintf(buf, sizeof buf, "k_eval_cubes_stream<%d>", h->T.Nd > 16384 ? 1024 : 512);
    else if (jit) snprintf(buf, sizeof buf, "k_eval_cubes_geo<%d,%s,%s>", 2 * h->ppt, h->npairs > 0 ? "true" : "false", h->T.exact_big ? "true" : "false");   // (DUAL: the group table of the last multi-group call paired up; XBIG: parity_mode 2)
    else snprintf(buf, sizeof buf, "k_eval_cubes<%d,%d,8,false>", h->block, h->ppt);
    h->kernel_name = buf;
    return h->kernel_name.c_str();
}

extern "C" int goicp_engine_eval_cubes_dev(goicp_engine* h, int slot, int rot_level, const void* cubes_dev, int n, void* ub_dev, void* lb_dev) {
    if (!h || slot < 0 || slot >= h->nslots || rot_level >= GOICP_MAXROTLEVEL || n < 0 || (n && (!cubes_dev || !ub_dev || !lb_dev))) return GOICP_ERR_ARG;
    if (n == 0) return GOICP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    return launch_eval(h, slot, rot_level, nullptr, 1, cubes_dev, n, ub_dev, lb_dev);
}

extern "C" int goicp_engine_eval_cubes_multi_dev(goicp_engine* h, const int32_t* slots, const int32_t* rot_levels, int ngroups,
                                                 const void* cubes_dev, int n, void* ub_dev, void* lb_dev) {
    if (!h || !slots || !rot_levels || ngroups <= 0 || ngroups > 4096 || n < 0 || (n && (!cubes_dev || !ub_dev || !lb_dev))) return GOICP_ERR_ARG;
    if (n == 0) return GOICP_OK;
    std::vector<int2> g((size_t)ngroups);
    for (int k = 0; k < ngroups; k++) {
        if (slots[k] < 0 || slots[k] >= h->nslots || rot_levels[k] >= GOICP_MAXROTLEVEL) return GOICP_ERR_ARG;
        g[k].x = slots[k]; g[k].y = rot_levels[k];
    }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->groups_host.size() != g.size() || memcmp(h->groups_host.data(), g.data(), g.size() * sizeof(int2)) != 0) {
        // new group table: stage it (the staging copy lives in the handle, so no host sync is needed on the launch path)
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->groups_host = g;
        HIPCHK(h, hipMemcpyAsync(h->d_groups, h->groups_host.data(), (size_t)ngroups * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        h->groups4_host.resize((size_t)ngroups);
        for (int k = 0; k < ngroups; k++) h->groups4_host[(size_t)k] = make_int4(g[k].x, g[k].y, k, k);
        HIPCHK(h, hipMemcpyAsync(h->d_groups4, h->groups4_host.data(), (size_t)ngroups * sizeof(int4), hipMemcpyHostToDevice, h->stream));
        // groups that share a slot, one without rotation uncertainty and one with: evaluated from one pass of gathers when ALL groups pair up
        h->npairs = 0;
        if (h->fuse && (ngroups & 1) == 0 && ngroups / 2 <= 2048) {
            std::vector<int4> pairs; std::vector<char> used((size_t)ngroups, 0);
            for (int a = 0; a < ngroups; a++) {
                if (used[a] || g[a].y >= 0) continue;
                for (int b = 0; b < ngroups; b++) if (!used[b] && b != a && g[b].x == g[a].x && g[b].y >= 0) { used[a] = used[b] = 1; pairs.push_back(make_int4(g[a].x, g[b].y, a, b)); break; }
            }
            if ((int)pairs.size() * 2 == ngroups) {
                h->pairs_host = pairs;
                HIPCHK(h, hipMemcpyAsync(h->d_pairs, h->pairs_host.data(), pairs.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
                h->npairs = (int)pairs.size();
            }
        }
    }
    return launch_eval(h, 0, -1, h->d_groups, ngroups, cubes_dev, n, ub_dev, lb_dev);
}

extern "C" int goicp_engine_eval_cubes(goicp_engine* h, int slot, int rot_level, const float* cubes, int n, float* ub, float* lb) {
    if (!h || n < 0 || (n && (!cubes || !ub || !lb))) return GOICP_ERR_ARG;
    for (int off = 0; off < n; off += h->cube_cap) {
        const int m = std::min(h->cube_cap, n - off);
        HIPCHK(h, hipMemcpyAsync(h->d_cubes, cubes + (size_t)4 * off, (size_t)4 * m * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        int rc = goicp_engine_eval_cubes_dev(h, slot, rot_level, h->d_cubes, m, h->d_ub, h->d_lb);
        if (rc != GOICP_OK) return rc;
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipMemcpyAsync(ub + off, h->d_ub, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(lb + off, h->d_lb, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0.f; (void)hipEventElapsedTime(&ms, h->ev0, h->ev1); h->kernel_ms += ms;
    }
    return GOICP_OK;
}

extern "C" int goicp_engine_corner_terms(goicp_engine* h, int slot, const float* xyz, int n, int32_t* incompat, float* fpfh) {
    if (!h || slot < 0 || slot >= h->nslots || n < 0 || (n && !xyz)) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const float4* rot = h->d_rot + (size_t)slot * h->T.Nd;
    for (int off = 0; off < n; off += h->lat_cap) {
        const int m = std::min(h->lat_cap, n - off);
        const int blocks = (m + NLMAX - 1) / NLMAX;
        int rc = ensure_fpscratch(h, (size_t)blocks);
        if (rc != GOICP_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(h->d_lat, xyz + (size_t)3 * off, (size_t)3 * m * sizeof(float), hipMemcpyHostToDevice, h->stream));
#define CALL(B, P) hipLaunchKernelGGL((k_corner_terms<B, P>), dim3(blocks), dim3(B), 0, h->stream, h->T, rot, h->d_lat, m, h->d_cnt, h->d_fp, h->d_fpscratch)
        DISPATCH_GEOM(h, CALL);
#undef CALL
        HIPCHK(h, hipGetLastError());
        if (incompat) HIPCHK(h, hipMemcpyAsync(incompat + off, h->d_cnt, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (fpfh) HIPCHK(h, hipMemcpyAsync(fpfh + off, h->d_fp, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return GOICP_OK;
}

static int ensure_heaps(goicp_engine* h, size_t jobs) {
    // per-job queue slabs of the batch path (fixed capacity; a job that outgrows its slab is re-run alone on the big slab)
    if (h->d_heaps && jobs <= h->heap_jobs) return GOICP_OK;
    if (h->d_heaps) { HIPCHK(h, hipStreamSynchronize(h->stream)); cache_free(h->d_heaps); h->d_heaps = nullptr; }
    HIPCHK(h, cache_malloc((void**)&h->d_heaps, jobs * (size_t)h->heap_cap * sizeof(HeapNode), h->device));
    h->heap_jobs = jobs;
    return GOICP_OK;
}

static int ensure_big_heap(goicp_engine* h, int cap) {
    if (h->d_heap_big && cap <= h->heap_big_cap) return GOICP_OK;
    if (h->d_heap_big) { HIPCHK(h, hipStreamSynchronize(h->stream)); cache_free(h->d_heap_big); h->d_heap_big = nullptr; }
    HIPCHK(h, cache_malloc((void**)&h->d_heap_big, (size_t)cap * sizeof(HeapNode), h->device));
    h->heap_big_cap = cap;
    return GOICP_OK;
}

static int run_bnb(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results, HeapNode* heaps, int cap) {
    if (!h->stream_bulk && h->T.Nd <= 1024 && !h->streamed) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, cache_stream_aux(h->device, 1, &h->stream_bulk)); }
    hipStream_t sq = h->stream_bulk ? h->stream_bulk : h->stream;      // (every call returns synchronised, so consecutive calls may use different streams)
    int rc = ensure_fpscratch(h, (size_t)njobs);
    if (rc != GOICP_OK) return rc;
    if (h->streamed) { rc = ensure_stream_scratch(h, (size_t)njobs, true); if (rc != GOICP_OK) return rc; }
    if (h->jit_bnb) { rc = ensure_stream_scratch(h, (size_t)njobs, true, false); if (rc != GOICP_OK) return rc; }
    // job descriptors and results go through the pooled pinned buffer when they fit: the kernel reads its job and writes its
    // result over the bus, so a launch costs no staging copies
    const bool zero_copy = h->pinned && (size_t)njobs * (sizeof(goicp_bnb_job) + sizeof(goicp_bnb_result)) <= PINNED_BYTES;
    goicp_bnb_job* k_jobs = h->d_jobs; goicp_bnb_result* k_results = h->d_results;
    if (zero_copy) {
        k_jobs = (goicp_bnb_job*)h->pinned; k_results = (goicp_bnb_result*)(k_jobs + njobs);
        memcpy(k_jobs, jobs, (size_t)njobs * sizeof(goicp_bnb_job));
    } else HIPCHK(h, hipMemcpyAsync(h->d_jobs, jobs, (size_t)njobs * sizeof(goicp_bnb_job), hipMemcpyHostToDevice, sq));
    HIPCHK(h, hipEventRecord(h->ev0, sq));
#define CALL(B, P) do { if (h->T.use_compat || h->T.use_fpfh) hipLaunchKernelGGL((k_inner_bnb<B, P, true>), dim3(njobs), dim3(B), h->dyn_lds_bnb, sq, h->T, k_jobs, k_results, heaps, cap, h->d_fpscratch, h->max_pops, h->heap_lds_nodes, h->rbuf_floats, h->fp_lds); \
                        else hipLaunchKernelGGL((k_inner_bnb<B, P, false>), dim3(njobs), dim3(B), h->dyn_lds_bnb, sq, h->T, k_jobs, k_results, heaps, cap, h->d_fpscratch, h->max_pops, h->heap_lds_nodes, h->rbuf_floats, h->fp_lds); } while (0)
#define CALLJ(P2) hipLaunchKernelGGL((k_inner_bnb<512, P2, false, false, true>), dim3(njobs), dim3(512), h->dyn_lds_bnb, sq, h->T, k_jobs, k_results, heaps, cap, h->d_fpscratch, h->max_pops, h->heap_lds_nodes, h->rbuf_floats, h->fp_lds)
    if (h->streamed) hipLaunchKernelGGL((k_inner_bnb<1024, 1, false, true>), dim3(njobs), dim3(1024), h->dyn_lds_bnb, sq, h->T, k_jobs, k_results, heaps, cap, h->d_fpscratch, h->max_pops, h->heap_lds_nodes, h->rbuf_floats, h->fp_lds);
    else if (h->jit_bnb) CALLJ(4);
    else DISPATCH_GEOM(h, CALL);
#undef CALLJ
#undef CALL
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, sq));
    if (!zero_copy) HIPCHK(h, hipMemcpyAsync(results, h->d_results, (size_t)njobs * sizeof(goicp_bnb_result), hipMemcpyDeviceToHost, sq));
    HIPCHK(h, wait_stream(h, sq));
    if (zero_copy) memcpy(results, k_results, (size_t)njobs * sizeof(goicp_bnb_result));
    float ms = 0.f; (void)hipEventElapsedTime(&ms, h->ev0, h->ev1); h->kernel_ms += ms;
    return GOICP_OK;
}

// ---- round-based search: buffers, launch loop ------------------------------------------------------------------
#define FR_KMAX 16
#define FR_ROUNDS_MAX 64
static int ensure_fr(goicp_engine* h, size_t jobs, size_t cap, size_t ecap) {
    // per-search arrays (state, results, rotated cloud, work list) grow with the number of searches; the queue and expansion slabs
    // with searches x capacity (a search that outgrew its slab is re-run alone on a larger one: few searches, large capacity)
    if (jobs > h->fr_jobs) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        { void* fr[] = {h->fr_st, h->fr_wl, h->fr_srot, h->fr_results}; for (void* p : fr) if (p) cache_free(p); }
        h->fr_st = nullptr; h->fr_wl = nullptr; h->fr_srot = nullptr; h->fr_results = nullptr; h->fr_jobs = 0;
        const size_t wl_cap = std::min<size_t>(jobs * FR_KMAX * 9 * (size_t)h->fr_nsplit * 2, (size_t)1 << 26);
        HIPCHK(h, cache_malloc((void**)&h->fr_st, jobs * sizeof(FrState), h->device));
        HIPCHK(h, cache_malloc((void**)&h->fr_results, jobs * sizeof(goicp_bnb_result), h->device));
        HIPCHK(h, cache_malloc((void**)&h->fr_wl, wl_cap * sizeof(FrItem), h->device));
        HIPCHK(h, cache_malloc((void**)&h->fr_srot, jobs * (size_t)h->T.NdP * sizeof(float4), h->device));
        h->fr_jobs = jobs; h->fr_wl_cap = (int)wl_cap;
    }
    if (!h->fr_cnt) HIPCHK(h, cache_malloc((void**)&h->fr_cnt, 2 * FR_ROUNDS_MAX * sizeof(unsigned), h->device));
    if (jobs * cap > h->fr_cap_nodes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->fr_heaps) cache_free(h->fr_heaps);
        h->fr_heaps = nullptr; h->fr_cap_nodes = 0;
        HIPCHK(h, cache_malloc((void**)&h->fr_heaps, jobs * cap * sizeof(HeapNodeX), h->device));
        h->fr_cap_nodes = jobs * cap;
    }
    if (jobs * ecap > h->fr_cap_exps) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        { void* fr[] = {h->fr_exps, h->fr_parts}; for (void* p : fr) if (p) cache_free(p); }
        h->fr_exps = nullptr; h->fr_parts = nullptr; h->fr_cap_exps = 0;
        HIPCHK(h, cache_malloc((void**)&h->fr_exps, jobs * ecap * sizeof(FrExp), h->device));
        if (h->fr_mode == 3) HIPCHK(h, cache_malloc((void**)&h->fr_parts, jobs * ecap * (size_t)h->fr_nsplit * 16 * sizeof(double), h->device));
        h->fr_cap_exps = jobs * ecap;
    }
    return GOICP_OK;
}

static void launch_fr_expand(goicp_engine* h, const FrCtx& C, int round, int max_items) {
    const dim3 grid((unsigned)(h->fr_mode == 2 ? std::min(h->fr_wgs, max_items) : max_items));
    if (h->fr_mode == 1 && h->T.exact_big) {
        for (int i0 = 0; i0 < max_items; i0 += h->T.xslots) {
            const dim3 gc((unsigned)std::min(h->T.xslots, max_items - i0));
#define FRC(B, P) hipLaunchKernelGGL((k_fr_expand<B, P, 0, 4, true>), gc, dim3(B), 0, h->stream, h->T, C, round, i0)
            if (h->fr_ppt == 4) FRC(512, 4); else if (h->fr_ppt == 8) FRC(512, 8); else FRC(512, 10);
#undef FRC
        }
        return;
    }
#define FRX(B, P, M, NTT) hipLaunchKernelGGL((k_fr_expand<B, P, M, NTT>), grid, dim3(B), 0, h->stream, h->T, C, round, 0)
    if (h->fr_mode == 1) { if (h->fr_ppt == 4) FRX(512, 4, 0, 4); else if (h->fr_ppt == 8) FRX(512, 8, 0, 4); else FRX(512, 10, 0, 4); }
    else if (h->fr_mode == 2) { if (h->fr_block == 512) FRX(512, 1, 1, 8); else FRX(1024, 1, 1, 8); }
    else FRX(1024, 1, 2, 8);
#undef FRX
}

// njobs searches, round by round. `cap` = queue / expansion-slot capacity per search.
static int run_fr(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results, int cap) {
    // expansion records per search: speculation (several nodes per round, two levels deep) consumes them faster than the queue grows
    const int ecap = (int)std::min<long long>((long long)cap * (njobs <= 128 ? 8 : 2), 1ll << 27);
    int rc = ensure_fr(h, (size_t)njobs, (size_t)cap, (size_t)ecap);
    if (rc != GOICP_OK) return rc;
    if (h->fr_mode == 2) { rc = ensure_stream_scratch(h, (size_t)h->fr_wgs, false); if (rc != GOICP_OK) return rc; }
    HIPCHK(h, hipMemcpyAsync(h->d_jobs, jobs, (size_t)njobs * sizeof(goicp_bnb_job), hipMemcpyHostToDevice, h->stream));
    FrCtx C;
    C.jobs = h->d_jobs; C.st = h->fr_st; C.results = h->fr_results; C.heaps = h->fr_heaps; C.heap_cap = cap; C.exps = h->fr_exps; C.exp_cap = ecap;
    C.wl = h->fr_wl; C.wl_cap = h->fr_wl_cap; C.wl_count = h->fr_cnt; C.active = h->fr_cnt + FR_ROUNDS_MAX; C.parts = h->fr_parts; C.nsplit = h->fr_nsplit; C.ngroups = h->fr_mode == 1 ? 2 : 1;
    C.srot = h->fr_srot; C.NdP = h->T.NdP; C.max_pops = h->max_pops; C.dbg = (h->T.dbg & 32) ? h->T.dbg_times : nullptr;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(k_fr_init, dim3(njobs), dim3(256), 0, h->stream, h->T, C);
    HIPCHK(h, hipGetLastError());
    unsigned* host_active = h->pinned ? (unsigned*)h->pinned : nullptr;
    unsigned active = (unsigned)njobs, active_tmp = 0;
    int rounds = 4;
    const int per_tile = h->fr_nsplit * C.ngroups;
    const int target = 4 * h->fr_wgs;                     // work items per round the chip absorbs without queueing for long
    for (int batch = 0; active > 0; batch++) {
        // few live searches: every chosen node is expanded two levels deep (9 tiles) and up to FR_KMAX nodes are chosen per search;
        // many live searches: one level, as many nodes as still fit the target
        // (measured on the 16 searches of a configs[2] rotation pop: one level 9.3 ms / 184 rounds, two levels 11.9 ms / 120 rounds)
        int depth = ((long long)active * 9 * per_tile * 8 <= target && h->fr_depth2) ? 2 : 1;
        const int slots = depth == 2 ? 9 : 1;
        const int K = std::max(1, std::min(FR_KMAX, target / (int)std::max(1ll, (long long)active * slots * per_tile)));
        HIPCHK(h, hipMemsetAsync(h->fr_cnt, 0, 2 * FR_ROUNDS_MAX * sizeof(unsigned), h->stream));
        for (int r = 0; r < rounds; r++) {
            hipLaunchKernelGGL(k_fr_select, dim3(njobs), dim3(FR_SELECT_THREADS), 0, h->stream, C, r, K, depth);
            launch_fr_expand(h, C, r, (int)std::min<long long>((long long)active * K * slots * per_tile, (long long)C.wl_cap));
        }
        HIPCHK(h, hipGetLastError());
        h->fr_rounds += (uint64_t)rounds;
        if (host_active) { HIPCHK(h, hipMemcpyAsync(host_active, C.active + (rounds - 1), sizeof(unsigned), hipMemcpyDeviceToHost, h->stream)); }
        else HIPCHK(h, hipMemcpyAsync(&active_tmp, C.active + (rounds - 1), sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        active = host_active ? *host_active : active_tmp;
        rounds = std::min(32, rounds * 2);
        if (batch > 100000) return fail(h, GOICP_ERR_HIP, "round-based search did not terminate");
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(results, h->fr_results, (size_t)njobs * sizeof(goicp_bnb_result), hipMemcpyDeviceToHost, h->stream));
    std::vector<FrState> sts((size_t)njobs);
    HIPCHK(h, hipMemcpyAsync(sts.data(), h->fr_st, (size_t)njobs * sizeof(FrState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (const FrState& st : sts) h->fr_spec_cubes += st.spec;          // cube evaluations launched, consumed by the replay or not
    float ms = 0.f; (void)hipEventElapsedTime(&ms, h->ev0, h->ev1); h->kernel_ms += ms;
    return GOICP_OK;
}

extern "C" int goicp_engine_inner_bnb_batch(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results) {
    if (!h || njobs < 0 || (njobs && (!jobs || !results))) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    for (int j = 0; j < njobs; j++) if (jobs[j].rot_level >= GOICP_MAXROTLEVEL) return GOICP_ERR_ARG;
    int worst = GOICP_OK;
    const int job_cap = h->streamed ? std::min(h->job_cap, 256) : h->job_cap;
    for (int off = 0; off < njobs; off += job_cap) {
        const int m = std::min(job_cap, njobs - off);
        int rc;
        if (h->fr_mode) rc = run_fr(h, jobs + off, m, results + off, h->fr_heap_cap);
        else {
            rc = ensure_heaps(h, (size_t)m);
            if (rc != GOICP_OK) return rc;
            rc = run_bnb(h, jobs + off, m, results + off, h->d_heaps, h->heap_cap);
        }
        if (rc != GOICP_OK) return rc;
        for (int j = 0; j < m; j++) {
            goicp_bnb_result& r = results[off + j];
            // a job whose queue outgrew its slab is re-run alone on a larger one (the search is deterministic); jobs that carry a
            // pop budget are speculative and are simply reported as overflowed
            int cap = h->fr_mode ? h->fr_heap_cap : h->heap_cap;
            while (r.status == GOICP_ERR_OVERFLOW && jobs[off + j].max_pops == 0 && cap < (1 << 26)) {
                cap *= 16;
                if (h->fr_mode) rc = run_fr(h, jobs + off + j, 1, &r, cap);
                else {
                    rc = ensure_big_heap(h, cap);
                    if (rc != GOICP_OK) return rc;
                    rc = run_bnb(h, jobs + off + j, 1, &r, h->d_heap_big, cap);
                }
                if (rc != GOICP_OK) return rc;
            }
            h->n_cubes += r.cube_evals; h->n_lookups += (uint64_t)r.cube_evals * h->T.Nd;
            if (r.status != GOICP_OK) worst = r.status;
        }
    }
    return worst;
}

extern "C" int goicp_engine_inner_bnb(goicp_engine* h, const goicp_bnb_job* job, goicp_bnb_result* result) {
    return goicp_engine_inner_bnb_batch(h, job, 1, result);
}

extern "C" int goicp_engine_point_eval(goicp_engine* h, const double* R9, const double* t3, float* dist, int32_t* compat) {
    if (!h || !R9 || !t3 || !dist) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    RtArgD rt; memcpy(rt.v, R9, 9 * sizeof(double)); memcpy(rt.v + 9, t3, 3 * sizeof(double));
    const size_t Nd = (size_t)h->T.Nd;
    if (h->pinned && Nd * 8 <= PINNED_BYTES) {
        float* pd = (float*)h->pinned; int* pc = (int*)h->pinned + Nd;
        hipLaunchKernelGGL(k_point_eval, dim3((h->T.Nd + 255) / 256), dim3(256), 0, h->stream_hi, h->T, rt, pd, compat ? pc : nullptr);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, wait_stream(h, h->stream_hi));
        memcpy(dist, pd, Nd * sizeof(float));
        if (compat) memcpy(compat, pc, Nd * sizeof(int));
    } else {
        hipLaunchKernelGGL(k_point_eval, dim3((h->T.Nd + 255) / 256), dim3(256), 0, h->stream_hi, h->T, rt, h->d_dist, compat ? h->d_int : nullptr);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(dist, h->d_dist, (size_t)h->T.Nd * sizeof(float), hipMemcpyDeviceToHost, h->stream_hi));
        if (compat) HIPCHK(h, hipMemcpyAsync(compat, h->d_int, (size_t)h->T.Nd * sizeof(int), hipMemcpyDeviceToHost, h->stream_hi));
        HIPCHK(h, wait_stream(h, h->stream_hi));
    }
    h->n_lookups += (uint64_t)h->T.Nd;
    return GOICP_OK;
}

extern "C" int goicp_engine_icp_nn_ties(goicp_engine* h, const double* R9, const double* t3, int32_t* id_model, float* dist_sq, int32_t* n_tied) {
    if (!h || !R9 || !t3 || !id_model || !dist_sq || h->T.Nm <= 0) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    RtArg rt;
    for (int k = 0; k < 9; k++) rt.v[k] = (float)R9[k];
    for (int k = 0; k < 3; k++) rt.v[9 + k] = (float)t3[k];
    const size_t Nd = (size_t)h->T.Nd;
    if (h->pinned && Nd * 12 <= PINNED_BYTES) {
        // small clouds (every ICP iteration of a cavity pair comes through here): the kernel writes its result arrays
        // straight into pinned host memory, so an iteration is one launch and one stream synchronisation
        int* pid = (int*)h->pinned; float* pd = (float*)h->pinned + Nd; int* pt = (int*)h->pinned + 2 * Nd;
        hipLaunchKernelGGL(k_icp_nn, dim3((h->T.Nd + 3) / 4), dim3(256), 0, h->stream_hi, h->T, h->d_mxyz, rt, pid, pd, n_tied ? pt : nullptr);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, wait_stream(h, h->stream_hi));
        memcpy(id_model, pid, Nd * sizeof(int)); memcpy(dist_sq, pd, Nd * sizeof(float));
        if (n_tied) memcpy(n_tied, pt, Nd * sizeof(int));
        return GOICP_OK;
    }
    hipLaunchKernelGGL(k_icp_nn, dim3((h->T.Nd + 3) / 4), dim3(256), 0, h->stream_hi, h->T, h->d_mxyz, rt, h->d_int, h->d_dist, n_tied ? h->d_tie : nullptr);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(id_model, h->d_int, (size_t)h->T.Nd * sizeof(int), hipMemcpyDeviceToHost, h->stream_hi));
    HIPCHK(h, hipMemcpyAsync(dist_sq, h->d_dist, (size_t)h->T.Nd * sizeof(float), hipMemcpyDeviceToHost, h->stream_hi));
    if (n_tied) HIPCHK(h, hipMemcpyAsync(n_tied, h->d_tie, (size_t)h->T.Nd * sizeof(int), hipMemcpyDeviceToHost, h->stream_hi));
    HIPCHK(h, wait_stream(h, h->stream_hi));
    return GOICP_OK;
}

static_assert(sizeof(IcpState) == sizeof(goicp_icp_state), "goicp_icp_state");
extern "C" int goicp_engine_icp_loop(goicp_engine* h, goicp_icp_state* state, float err_diff, int max_iters, int32_t* id_model, float* dist_sq) {
    if (!h || !state || !id_model || !dist_sq || max_iters < 0) return GOICP_ERR_ARG;
    const size_t n = (size_t)h->T.Nd, nm = (size_t)h->T.Nm;
    const size_t lds = (3 * nm + 11 * n + 3 * 1024) * sizeof(float);
    // cavity class only: untrimmed, everything in one workgroup's LDS, results through the pinned buffer
    if (h->T.doTrim || h->T.Nm <= 0 || n > 1024 || nm > 2048 || lds > 120 * 1024 || !h->pinned || sizeof(IcpState) + 8 * n + 64 > PINNED_BYTES) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (lds > 48 * 1024) HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_icp_loop)));
    IcpState st0; memcpy(&st0, state, sizeof st0);
    IcpState* po = (IcpState*)h->pinned;
    int* pid = (int*)((char*)h->pinned + ((sizeof(IcpState) + 63) & ~(size_t)63)); float* pdq = (float*)(pid + n);
    hipLaunchKernelGGL(k_icp_loop, dim3(1), dim3(1024), lds, h->stream_hi, h->T, h->d_mxyz, st0, err_diff, max_iters, po, pid, pdq);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, wait_stream(h, h->stream_hi));
    memcpy(state, po, sizeof st0);
    memcpy(id_model, pid, n * sizeof(int)); memcpy(dist_sq, pdq, n * sizeof(float));
    return GOICP_OK;
}

extern "C" int goicp_engine_icp_nn(goicp_engine* h, const double* R9, const double* t3, int32_t* id_model, float* dist_sq) {
    return goicp_engine_icp_nn_ties(h, R9, t3, id_model, dist_sq, nullptr);
}

// ---------------------------------------------------------------------------------------------
// neighbour counts for the point weighting (GoICP::neighborsWeights, reference jly_goicp.cpp:1453-1498 with isNeighbor
// :1097-1103): count_i = #{ j != i : sqrt(dx^2+dy^2+dz^2) < lim } with the differences formed in float and everything
// else in double, exactly as the reference's pow/sqrt expression. O(Nd^2) per radius step -> one kernel per step.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_neighbor_counts(const float* __restrict__ xyz, int n, double lim, int* __restrict__ counts) {
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float xi = 0.f, yi = 0.f, zi = 0.f;
    if (i < n) { xi = xyz[3 * i]; yi = xyz[3 * i + 1]; zi = xyz[3 * i + 2]; }
    int cnt = 0;
    for (int base = 0; base < n; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < n) { sx[threadIdx.x] = xyz[3 * j]; sy[threadIdx.x] = xyz[3 * j + 1]; sz[threadIdx.x] = xyz[3 * j + 2]; }
        __syncthreads();
        const int m = min(256, n - base);
        for (int q = 0; q < m; q++) {
            const double dx = (double)(sx[q] - xi), dy = (double)(sy[q] - yi), dz = (double)(sz[q] - zi);
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            cnt += (base + q != i && d < lim) ? 1 : 0;
        }
    }
    if (i < n) counts[i] = cnt;
}

struct goicp_nbr_ctx { int device; int n; float* d_xyz; int* d_cnt; hipStream_t stream; hipEvent_t ev0, ev1; void* pinned; };

extern "C" int goicp_nbr_create(int device_id, const float* xyz, int n, goicp_nbr_ctx** out) {
    if (!xyz || n <= 0 || !out) return GOICP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return GOICP_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return GOICP_ERR_NO_DEVICE;
    goicp_nbr_ctx* c = new goicp_nbr_ctx();
    c->device = device_id; c->n = n; c->d_xyz = nullptr; c->d_cnt = nullptr; c->stream = nullptr; c->ev0 = c->ev1 = nullptr; c->pinned = nullptr;
    if (cache_queue(device_id, &c->stream, &c->ev0, &c->ev1, &c->pinned) != hipSuccess ||
        cache_malloc((void**)&c->d_xyz, (size_t)3 * n * sizeof(float), device_id) != hipSuccess || cache_malloc((void**)&c->d_cnt, (size_t)n * sizeof(int), device_id) != hipSuccess ||
        hipMemcpyAsync(c->d_xyz, xyz, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        if (c->d_xyz) cache_free(c->d_xyz);
        if (c->d_cnt) cache_free(c->d_cnt);
        cache_queue_return(device_id, c->stream, c->ev0, c->ev1, c->pinned);
        delete c; return GOICP_ERR_HIP;
    }
    *out = c;
    return GOICP_OK;
}
extern "C" int goicp_nbr_count(goicp_nbr_ctx* c, double lim, int32_t* counts) {
    if (!c || !counts) return GOICP_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return GOICP_ERR_HIP;
    hipLaunchKernelGGL(k_neighbor_counts, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->d_xyz, c->n, lim, c->d_cnt);
    if (hipGetLastError() != hipSuccess) return GOICP_ERR_HIP;
    if (hipMemcpyAsync(counts, c->d_cnt, (size_t)c->n * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return GOICP_ERR_HIP;
    return GOICP_OK;
}
extern "C" void goicp_nbr_destroy(goicp_nbr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    cache_free(c->d_xyz); cache_free(c->d_cnt);
    cache_queue_return(c->device, c->stream, c->ev0, c->ev1, c->pinned);
    delete c;
}
