// Micro-benchmark behind DESIGN.md's gather-layout choice: the same 8 x 5000 DT lookups per tile issued as
//   A: lane = point, 8 back-to-back gathers per lane (x-neighbour cubes -> the same 2-3 cache lines, still pending)
//   B: lane = (point, cube): the 8 lookups of a point sit in 8 adjacent lanes of ONE instruction
// build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define S 300
#define NP 5000
__global__ __launch_bounds__(1024) void kA(const float* __restrict__ dt, const float4* __restrict__ pts, const float4* __restrict__ tiles, float step, float* out) {
    const float4 tl = tiles[blockIdx.x];
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < 5; m++) {
        const int i = min((int)threadIdx.x + 1024 * m, NP - 1);
        const float4 p = pts[i];
        const int y = min(max((int)(p.y + tl.y), 0), S - 1), z = min(max((int)(p.z + tl.z), 0), S - 1);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const int x = min(max((int)(p.x + tl.x + j * step), 0), S - 1); v[j] = dt[(z * S + y) * S + x]; }
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j];
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void kB(const float* __restrict__ dt, const float4* __restrict__ pts, const float4* __restrict__ tiles, float step, float* out) {
    const float4 tl = tiles[blockIdx.x];
    const int j = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const float xo = tl.x + j * step;
    float acc = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < 40; k0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = min(slot + 128 * (k0 + k), NP - 1);
            const float4 p = pts[i];
            const int y = min(max((int)(p.y + tl.y), 0), S - 1), z = min(max((int)(p.z + tl.z), 0), S - 1);
            const int x = min(max((int)(p.x + xo), 0), S - 1);
            v[k] = dt[(z * S + y) * S + x];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
int main() {
    const int ntiles = 18720;
    std::vector<float> dt((size_t)S * S * S);
    for (size_t i = 0; i < dt.size(); i++) dt[i] = (float)(i % 977);
    std::vector<float4> pts(NP), tiles(ntiles);
    srand(1);
    for (int i = 0; i < NP; i++) {      // shell of radius ~40 cells around the grid centre
        float a = rand() / (float)RAND_MAX * 6.2831853f, c = rand() / (float)RAND_MAX * 2 - 1, s = sqrtf(1 - c * c), r = 36 + 4 * (rand() / (float)RAND_MAX);
        pts[i] = make_float4(150 + r * s * cosf(a), 150 + r * s * sinf(a), 150 + r * c, 0);
    }
    for (int t = 0; t < ntiles; t++) { int q = t % 585; tiles[t] = make_float4(-37.5f + (q & 1) * 37.5f, -37.5f + ((q >> 1) & 15) * 4.7f, -37.5f + ((q >> 5) % 16) * 4.7f, 0); }
    float *d_dt, *d_out; float4 *d_pts, *d_tiles;
    hipMalloc(&d_dt, dt.size() * 4); hipMalloc(&d_out, ntiles * 4); hipMalloc(&d_pts, NP * 16); hipMalloc(&d_tiles, ntiles * 16);
    hipMemcpy(d_dt, dt.data(), dt.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_pts, pts.data(), NP * 16, hipMemcpyHostToDevice); hipMemcpy(d_tiles, tiles.data(), ntiles * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(kA, dim3(ntiles), dim3(1024), 0, 0, d_dt, d_pts, d_tiles, 4.7f, d_out);
            else hipLaunchKernelGGL(kB, dim3(ntiles), dim3(1024), 0, 0, d_dt, d_pts, d_tiles, 4.7f, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s rep %d: %.3f ms  (%.1f G lookups/s)\n", which ? "B lane=(point,cube)" : "A lane=point      ", rep, ms, ntiles * 8.0 * NP / ms / 1e6);
        }
    }
    return 0;
}
