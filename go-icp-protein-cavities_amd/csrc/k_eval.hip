// k_eval.hip -- the cube-list sweeps (k_eval_cubes*, the metric kernel k_eval_cubes_geo) and the corner-term kernel, with their launchers.
#include "engine_host.hpp"

// ---------------------------------------------------------------------------------------------
// k_eval_cubes_stream: k_eval_cubes for clouds above 8192 points. Persistent workgroups walk the (tile, group) list so
// that the residual-row scratch (trimmed runs) is per resident workgroup, not per tile.
// ---------------------------------------------------------------------------------------------
// DUAL (a template tag so that profilers list the fused and the plain launches as different kernels): `pairs` holds fused (no radius, radius)
// group pairs as in k_eval_cubes_geo.
template <int BLOCK, bool DUAL>
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
        if (DUAL) { const int4 pp = pairs[g]; slot = pp.x; rot_level = pp.y; row_a = pp.z; row_b = pp.w; }
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
        if (DUAL) {
            sweep_stream<false, BLOCK>(T, sh, rot_all + (size_t)slot * Nd, nullptr, rows, T.NdP, 0, 0, nullptr, rho_rows(T) + (size_t)rot_level * Nd);
            if (tid < TC && base + tid < n) {
                ub_out[(size_t)row_a * n + base + tid] = sh.ub2[tid]; lb_out[(size_t)row_a * n + base + tid] = sh.lb2[tid];
                ub_out[(size_t)row_b * n + base + tid] = sh.ub[tid];  lb_out[(size_t)row_b * n + base + tid] = sh.lb[tid];
            }
        } else {
            sweep_stream<false, BLOCK>(T, sh, rot_all + (size_t)slot * Nd, rot_level >= 0 ? rho_rows(T) + (size_t)rot_level * Nd : nullptr, rows, T.NdP);
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
    const float* rho_row_jit = use_rho ? rho_rows(T) + (size_t)rot_level * Nd : nullptr;
    if (JIT) {
#pragma unroll
        for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
    } else {   // all loads first (clamped index instead of a branch), one wait for the lot
        float4 p[PPT]; float rh[PPT];
        const float* rho_row = rho_rows(T) + (size_t)(use_rho ? rot_level : 0) * Nd;
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

#ifdef GOICP_PROBES
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
#endif  // GOICP_PROBES

// k_eval_cubes_geo: the cube sweep of clouds of 1025..5120 points without penalty terms (512 lanes, points fetched just in time) -- the
// metric kernel. A kernel of its own rather than an instantiation of k_eval_cubes: without the penalty-lattice and parity-mode code next
// to it the 80 residuals stay in registers (106 VGPRs, no spills; the general kernel spilled 26).
// DUAL = false: groups[y] = (slot, rot_level), output row y.
// DUAL = true : PAIRS of groups that share rotation slot and cube list -- the call without rotation uncertainty (rot_level -1) and a call
//               with it, as OuterBnB's two InnerBnB calls per child rotation do (reference jly_goicp.cpp:768,861): one pass of gathers, two
//               bound pairs per cube (sweep_tile<..., DUAL>). groups[y] = (slot, rot_level of the second call, output row of the first
//               call, output row of the second).
template <int PPT, bool DUAL, bool XBIG = false, int NORM = 0 /* 2: the squared norm compiled in (the common case); 0: T.norm read at run time */>
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
    const float* __restrict__ rho = g.y >= 0 ? rho_rows(T) + (size_t)g.y * Nd : nullptr;      // (slot clouds of this kernel are stored in T.perm order)
    if (tid < TC) {
        const int j = tid, c = min(base + j, n - 1);          // the tail tile re-evaluates the last cube
        const float x = cubes[4 * c], y = cubes[4 * c + 1], z = cubes[4 * c + 2], w = cubes[4 * c + 3];
        sh.cube[j][0] = x; sh.cube[j][1] = y; sh.cube[j][2] = z; sh.cube[j][3] = w;
        sh.cen[j][0] = x + w / 2; sh.cen[j][1] = y + w / 2; sh.cen[j][2] = z + w / 2;
        sh.mtd[j] = (float)(SQRT3_D / 2.0 * w);
    }
    if (!XBIG && T.doTrim) fw_hist_reset<512, DUAL>(sh);           // the first selection's histogram is filled beside the gathers (sweep_tile<..., XP>)
    __syncthreads();
    // every lane decides for itself whether the tile is a row of x-neighbours (broadcast LDS reads, no second barrier)
    bool row = true;
#pragma unroll
    for (int j = 1; j < TC; j++) row = row && sh.cen[j][1] == sh.cen[0][1] && sh.cen[j][2] == sh.cen[0][2];
    float px[PPT], py[PPT], pz[PPT], pw[PPT], prho[PPT];
#pragma unroll
    for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
    if constexpr (DUAL) {
        sweep_tile<512, PPT, false, TC, true, true, XBIG, !XBIG, NORM>(T, sh, px, py, pz, pw, prho, false, xr, row, rot, nullptr, T.code ? lutl : nullptr, rho);
        // (results out by lane id, from every wave alike -- identical values to identical addresses: the workitem id would have to survive the
        // whole sweep in a register the kernel does not have, i.e. in scratch)
        const int ol = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (ol < TC && base + ol < n) {
            ub_out[(size_t)g.z * n + base + ol] = sh.ub2[ol]; lb_out[(size_t)g.z * n + base + ol] = sh.lb2[ol];
            ub_out[(size_t)g.w * n + base + ol] = sh.ub[ol];  lb_out[(size_t)g.w * n + base + ol] = sh.lb[ol];
        }
    } else {
        sweep_tile<512, PPT, false, TC, true, false, XBIG, !XBIG, NORM>(T, sh, px, py, pz, pw, prho, g.y >= 0, xr, row, rot, rho, T.code ? lutl : nullptr);
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


// ---- host: configuration and launch ----
int eval_configure(goicp_engine* h) {
    if (!h->dyn_lds) return GOICP_OK;         // parity runs keep 2 * TC rows of Nd floats next to the static tile state
#define SETATTR(B, P) HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_eval_cubes<B, P, TC>)))
    DISPATCH_GEOM(h, SETATTR);
#undef SETATTR
    return GOICP_OK;
}

int launch_eval(goicp_engine* h, int slot, int rot_level, const int2* groups_dev, int ngroups, const void* cubes_dev, int n, void* ub_dev, void* lb_dev) {
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
#define CALLS(B, D) hipLaunchKernelGGL((k_eval_cubes_stream<B, D>), dim3(wgs), dim3(B), 0, h->stream, h->T, h->d_rot, slot, rot_level, groups_dev, ng, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, ptab)
        if (h->T.Nd > 16384) { if (dual) CALLS(1024, true); else CALLS(1024, false); }
        else { if (dual) CALLS(512, true); else CALLS(512, false); }
#undef CALLS
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
#define CALLGN(P2, D, NRM) hipLaunchKernelGGL((k_eval_cubes_geo<P2, D, false, NRM>), dim3((unsigned)wgs), dim3(512), 0, h->stream, h->T, h->d_rot, gtab, (const float*)cubes_dev, n, (float*)ub_dev, (float*)lb_dev, tiles, ng, xm, 0)
#define CALLG(P2) do { if (h->T.norm == 2) { if (dual) CALLGN(P2, true, 2); else CALLGN(P2, false, 2); } else { if (dual) CALLGN(P2, true, 0); else CALLGN(P2, false, 0); } } while (0)
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
#undef CALLGN
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

#ifdef GOICP_PROBES
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
#endif  // GOICP_PROBES

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

