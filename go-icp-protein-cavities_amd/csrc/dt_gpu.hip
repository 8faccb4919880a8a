// dt_gpu.hip -- exact Euclidean distance transform + nearest-occupied-cell map of the DT grid, built ON the MI355X.
//
// Opt-in replacement (GoICP::dt_mode = 1) for the host builder of csrc/host_dt.cpp. The reference's DT3D::Build (reference
// jly_3ddt.cpp:897-1137) propagates (v,h,d) offsets cell by cell in a fixed scan order -- a chain of dependent updates with no
// parallelism to give a GPU, and an APPROXIMATION of the Euclidean transform (a handful of cells differ, plus the
// phantom seed of its uninitialised first candidate). This builder computes the EXACT transform instead, separably and in
// integers, so it is not bit-identical to the reference grid in those cells and is therefore not the default: parity runs keep
// the host builder. What it keeps from the reference: the grid geometry, the seed cells ROUND((p - min) * scale)
// (jly_3ddt.cpp:976-995), the value stored per cell, float((double)(float)sqrt((double)d2) / scale) (jly_3ddt.cpp:28,79,1003)
// -- taken from a host-computed table indexed by the integer squared cell distance d2, so no device sqrt/divide is involved --
// and the meaning of the nearest-occupied-cell map (jly_3ddt.cpp:1012-1132: the cell with distance 0 the offsets point at).
//
// Three passes, all on squared integer distances (x fastest, as the engine's grid):
//   pass 1 (x)  rows that hold seeds: for every cell the nearest seed in its own row                        -> fx   int16[S^3]
//   pass 2 (y)  slabs z that hold seeds: min over the slab's seeded rows y' of (x-fx)^2 + (y-y')^2         -> g2   int32[S^3], src int32[S^3]
//   pass 3 (z)  min over the seeded slabs z' of g2 + (z-z')^2, slab values staged through LDS              -> dist float[S^3], nearest int32[S^3]
// Only seeded rows / slabs are visited (their lists come from the host: the cloud seeds a few thousand of the S^2 rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#include "goicp_engine.h"
#include "pool.hpp"

struct goicp_dtgpu {
    int device, S;
    float* d_dist; int32_t* d_nearest;
};

namespace {
#define DTCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "goicp_dtgpu: %s: %s\n", #call, hipGetErrorString(e_)); ok = false; } } while (0)

__global__ void k_dt_mark(const int32_t* __restrict__ cells, int n, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[cells[i]] = 1;
}

// pass 1: one workgroup per seeded row; every cell looks outwards for the nearest seed of the row (ties: the lower x)
__global__ __launch_bounds__(256) void k_dt_rows(const uint8_t* __restrict__ mask, const int32_t* __restrict__ rows, int S, int16_t* __restrict__ fx) {
    extern __shared__ uint8_t rowm[];
    const size_t base = (size_t)rows[blockIdx.x] * S;
    for (int x = threadIdx.x; x < S; x += blockDim.x) rowm[x] = mask[base + x];
    __syncthreads();
    for (int x = threadIdx.x; x < S; x += blockDim.x) {
        int best = -1;
        for (int d = 0; d < S; d++) {
            const int a = x - d, b = x + d;
            if (a >= 0 && rowm[a]) { best = a; break; }
            if (b < S && rowm[b]) { best = b; break; }
            if (a < 0 && b >= S) break;
        }
        fx[base + x] = (int16_t)best;
    }
}

// pass 2: one workgroup per (seeded slab z, 4 rows y, 64 columns x); candidates = the seeded rows of the slab
__global__ __launch_bounds__(256) void k_dt_slabs(const int16_t* __restrict__ fx, const int32_t* __restrict__ slabs, const int32_t* __restrict__ slab_ptr,
                                                  const int32_t* __restrict__ slab_rows, int S, int32_t* __restrict__ g2, int32_t* __restrict__ src) {
    const int z = slabs[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= S || y >= S) return;
    const int r0 = slab_ptr[blockIdx.z], r1 = slab_ptr[blockIdx.z + 1];
    int best = 0x7fffffff, bsrc = 0;
    for (int r = r0; r < r1; r++) {
        const int yy = slab_rows[r];                                   // row index inside the slab
        const int sx = fx[((size_t)z * S + yy) * S + x];
        const int dx = x - sx, dy = y - yy;
        const int d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; bsrc = (yy << 16) | sx; }
    }
    const size_t k = ((size_t)z * S + y) * S + x;
    g2[k] = best; src[k] = bsrc;
}

// pass 3: one workgroup per (row y, 64 columns x): the seeded slabs' g2 values of the tile go through LDS, every thread owns z = q, q+4, ...
__global__ __launch_bounds__(256) void k_dt_columns(const int32_t* __restrict__ g2, const int32_t* __restrict__ src, const int32_t* __restrict__ slabs, int nslab, int S,
                                                    const float* __restrict__ table, float* __restrict__ dist, int32_t* __restrict__ nearest) {
    extern __shared__ int32_t tile[];                                 // [nslab][64]
    const int xl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + xl, y = blockIdx.y;
    const int xc = min(x, S - 1);
    for (int k = q; k < nslab; k += 4) tile[k * 64 + xl] = g2[((size_t)slabs[k] * S + y) * S + xc];
    __syncthreads();
    if (x >= S) return;
    for (int z = q; z < S; z += 4) {
        int best = 0x7fffffff, bk = 0;
        for (int k = 0; k < nslab; k++) {
            const int dz = z - slabs[k];
            const int d2 = tile[k * 64 + xl] + dz * dz;
            if (d2 < best) { best = d2; bk = k; }
        }
        const int zs = slabs[bk];
        const int s2 = src[((size_t)zs * S + y) * S + x];
        const size_t k = ((size_t)z * S + y) * S + x;
        dist[k] = table[best];
        nearest[k] = (zs * S + (s2 >> 16)) * S + (s2 & 0xffff);
    }
}
}  // namespace

extern "C" int goicp_dtgpu_build(int device, int S, const int32_t* occ_cells, int n_occ, const float* dist_of_d2, int n_table, goicp_dtgpu** out) {
    if (!out) return GOICP_ERR_ARG;
    *out = nullptr;
    if (S < 1 || S > GOICP_MAX_GRID || !occ_cells || n_occ < 1 || !dist_of_d2 || n_table < 3 * (S - 1) * (S - 1) + 1) return GOICP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return GOICP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return GOICP_ERR_NO_DEVICE;
    const size_t ncell = (size_t)S * S * S;
    // seeded rows (z*S + y, ascending) and, per seeded slab, the row indices y of its seeded rows
    std::vector<int32_t> rows; rows.reserve((size_t)n_occ);
    for (int i = 0; i < n_occ; i++) { if (occ_cells[i] < 0 || (size_t)occ_cells[i] >= ncell) return GOICP_ERR_ARG; rows.push_back(occ_cells[i] / S); }
    std::sort(rows.begin(), rows.end()); rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    std::vector<int32_t> slabs, slab_ptr(1, 0), slab_rows;
    for (int32_t r : rows) {
        const int z = r / S;
        if (slabs.empty() || slabs.back() != z) { if (!slabs.empty()) slab_ptr.push_back((int32_t)slab_rows.size()); slabs.push_back(z); }
        slab_rows.push_back(r % S);
    }
    slab_ptr.push_back((int32_t)slab_rows.size());
    const int nslab = (int)slabs.size();
    if ((size_t)nslab * 256 > 160 * 1024 - 1024) return GOICP_ERR_ARG;          // (S <= GOICP_MAX_GRID: cannot happen below 636 seeded slabs)

    bool ok = true;
    goicp_dtgpu* h = new goicp_dtgpu(); h->device = device; h->S = S; h->d_dist = nullptr; h->d_nearest = nullptr;
    uint8_t* d_mask = nullptr; int16_t* d_fx = nullptr; int32_t *d_g2 = nullptr, *d_src = nullptr, *d_cells = nullptr, *d_rows = nullptr, *d_slabs = nullptr, *d_sptr = nullptr, *d_srows = nullptr;
    float* d_table = nullptr;
    // blocks, stream and events come from the process-wide pool (pool.hpp): no hipMalloc / hipFree / default-stream work per build, so
    // builds on several host threads (batch driver, dt_mode = 1) neither synchronise the device nor wait for each other's kernels
    hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; void* pinned = nullptr;
    DTCHK(cache_queue(device, &st, &e0, &e1, &pinned));
    DTCHK(cache_malloc((void**)&h->d_dist, ncell * sizeof(float), device));
    DTCHK(cache_malloc((void**)&h->d_nearest, ncell * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_mask, ncell, device));
    DTCHK(cache_malloc((void**)&d_fx, ncell * sizeof(int16_t), device));
    DTCHK(cache_malloc((void**)&d_g2, ncell * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_src, ncell * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_cells, (size_t)n_occ * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_rows, rows.size() * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_slabs, (size_t)nslab * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_sptr, slab_ptr.size() * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_srows, slab_rows.size() * sizeof(int32_t), device));
    DTCHK(cache_malloc((void**)&d_table, (size_t)n_table * sizeof(float), device));
    if (ok) {
        DTCHK(hipMemsetAsync(d_mask, 0, ncell, st));
        DTCHK(hipMemcpyAsync(d_cells, occ_cells, (size_t)n_occ * sizeof(int32_t), hipMemcpyHostToDevice, st));
        DTCHK(hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        DTCHK(hipMemcpyAsync(d_slabs, slabs.data(), (size_t)nslab * sizeof(int32_t), hipMemcpyHostToDevice, st));
        DTCHK(hipMemcpyAsync(d_sptr, slab_ptr.data(), slab_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        DTCHK(hipMemcpyAsync(d_srows, slab_rows.data(), slab_rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        DTCHK(hipMemcpyAsync(d_table, dist_of_d2, (size_t)n_table * sizeof(float), hipMemcpyHostToDevice, st));
    }
    if (ok) {
        hipLaunchKernelGGL(k_dt_mark, dim3((n_occ + 255) / 256), dim3(256), 0, st, d_cells, n_occ, d_mask);
        hipLaunchKernelGGL(k_dt_rows, dim3((unsigned)rows.size()), dim3(256), (size_t)((S + 15) & ~15), st, d_mask, d_rows, S, d_fx);
        hipLaunchKernelGGL(k_dt_slabs, dim3((S + 63) / 64, (S + 3) / 4, nslab), dim3(256), 0, st, d_fx, d_slabs, d_sptr, d_srows, S, d_g2, d_src);
        const size_t lds = (size_t)nslab * 64 * sizeof(int32_t);
        if (lds > 64 * 1024) DTCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dt_columns), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));   // (raised to the ceiling, never lowered: builds run side by side)
        hipLaunchKernelGGL(k_dt_columns, dim3((S + 63) / 64, S), dim3(256), lds, st, d_g2, d_src, d_slabs, nslab, S, d_table, h->d_dist, h->d_nearest);
        DTCHK(hipGetLastError());
    }
    if (st) (void)hipStreamSynchronize(st);                    // also on the error path: parked blocks must be idle
    void* tmp[] = {d_mask, d_fx, d_g2, d_src, d_cells, d_rows, d_slabs, d_sptr, d_srows, d_table};
    for (void* p : tmp) if (p) cache_free(p);
    cache_queue_return(device, st, e0, e1, pinned);
    if (!ok) { if (h->d_dist) cache_free(h->d_dist); if (h->d_nearest) cache_free(h->d_nearest); delete h; return GOICP_ERR_HIP; }
    *out = h;
    return GOICP_OK;
}

extern "C" const void* goicp_dtgpu_dist(const goicp_dtgpu* h) { return h ? h->d_dist : nullptr; }
extern "C" const void* goicp_dtgpu_nearest(const goicp_dtgpu* h) { return h ? h->d_nearest : nullptr; }

extern "C" int goicp_dtgpu_download(const goicp_dtgpu* h, float* dist, int32_t* nearest) {
    if (!h) return GOICP_ERR_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return GOICP_ERR_HIP;
    const size_t ncell = (size_t)h->S * h->S * h->S;
    if (dist && hipMemcpy(dist, h->d_dist, ncell * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return GOICP_ERR_HIP;
    if (nearest && hipMemcpy(nearest, h->d_nearest, ncell * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return GOICP_ERR_HIP;
    return GOICP_OK;
}

extern "C" void goicp_dtgpu_destroy(goicp_dtgpu* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    // (the engine that adopted the grid has been destroyed -- and its stream synchronised -- before this is called)
    if (h->d_dist) cache_free(h->d_dist);
    if (h->d_nearest) cache_free(h->d_nearest);
    delete h;
}
