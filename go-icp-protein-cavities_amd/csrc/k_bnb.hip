// k_bnb.hip -- k_inner_bnb: one workgroup = one whole InnerBnB call (reference jly_goicp.cpp:286-579), and its launcher.
#include "engine_host.hpp"

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
        srho = use_rho ? rho_rows(T) + (size_t)job.rot_level * Nd : nullptr;
        for (int i = tid; i < Nd; i += BLOCK) {
            const int s = src_point(T, i);
            const float x = T.dxyz[3 * s], y = T.dxyz[3 * s + 1], z = T.dxyz[3 * s + 2];
            float4 p;
            p.x = job.R[0] * x + job.R[1] * y + job.R[2] * z;
            p.y = job.R[3] * x + job.R[4] * y + job.R[5] * z;
            p.z = job.R[6] * x + job.R[7] * y + job.R[8] * z;
            p.w = T.weights[s];
            srot[i] = p;
        }
#pragma unroll
        for (int m = 0; m < PPT; m++) { px[m] = py[m] = pz[m] = pw[m] = prho[m] = 0.f; }
        __syncthreads();
    } else {
        float x[PPT], y[PPT], z[PPT], rh[PPT];
        const float* rho_row = rho_rows(T) + (size_t)(use_rho ? job.rot_level : 0) * Nd;
#pragma unroll
        for (int m = 0; m < PPT; m++) {
            const int i = min(tid + m * BLOCK, Nd - 1), s = src_point(T, i);
            x[m] = T.dxyz[3 * s]; y[m] = T.dxyz[3 * s + 1]; z[m] = T.dxyz[3 * s + 2]; pw[m] = T.weights[s]; rh[m] = rho_row[i];
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
    const bool timing = DBG_ON(T, 32) && T.dbg_times && blockIdx.x == 0 && tid == 0;
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
        if constexpr (PEN && BLOCK == 256 && !STREAM && !JIT) fused_pop = penalties && T.exact && !T.doTrim && !DBG_ON(T, 128);
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


// ---- host: configuration and launch ----
int bnb_configure(goicp_engine* h, int ncu) {
    // static tile state + residual rows + LDS heap can exceed the 64 KB a launch gets without asking
#define SETATTR(B, P) do { HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<B, P, true>))); \
                           HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<B, P, false>))); } while (0)
    DISPATCH_GEOM(h, SETATTR);
#undef SETATTR
    if (h->streamed) HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<1024, 1, false, true>)));
    if (h->jit_bnb) HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_inner_bnb<512, 4, false, false, true>)));
    int nb = 0;
    const bool pen = h->T.use_compat || h->T.use_fpfh;
#define OCC(B, P) do { if (pen) HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<B, P, true>, B, h->dyn_lds_bnb)); \
                       else HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<B, P, false>, B, h->dyn_lds_bnb)); } while (0)
    if (h->streamed) HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<1024, 1, false, true>, 1024, h->dyn_lds_bnb));
    else if (h->jit_bnb) HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inner_bnb<512, 4, false, false, true>, 512, h->dyn_lds_bnb));
    else DISPATCH_GEOM(h, OCC);
#undef OCC
    h->resident_jobs = std::max(1, nb) * std::max(1, ncu);
    return GOICP_OK;
}

int run_bnb(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results, HeapNode* heaps, int cap) {
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

