// k_fr.hip -- round-based InnerBnB for clouds above 1024 points: one search on many workgroups.
#include "engine_host.hpp"

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
__global__ __launch_bounds__(256) void k_fr_init(DevTables T, FrCtx C) {
    const int job = blockIdx.x, tid = threadIdx.x;
    const goicp_bnb_job jb = C.jobs[job];
    float4* srot = C.srot + (size_t)job * C.NdP;
    for (int i = tid; i < T.Nd; i += 256) {            // K5: pDataTemp = R*p, float, left to right (reference jly_goicp.cpp:750-756)
        const int s = src_point(T, i);
        const float x = T.dxyz[3 * s], y = T.dxyz[3 * s + 1], z = T.dxyz[3 * s + 2];
        float4 p;
        p.x = jb.R[0] * x + jb.R[1] * y + jb.R[2] * z;
        p.y = jb.R[3] * x + jb.R[4] * y + jb.R[5] * z;
        p.z = jb.R[6] * x + jb.R[7] * y + jb.R[8] * z;
        p.w = T.weights[s];
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
    const float* rho = level >= 0 ? rho_rows(T) + (size_t)level * T.Nd : nullptr;
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


// ---- round-based search: buffers, launch loop ------------------------------------------------------------------
#define FR_KMAX 63          // upper limit of the nodes a search may have expanded per round (k_fr_select looks at the first 63 queue slots); the engine's own limit is h->fr_kmax
#define FR_ROUNDS_MAX 64
static int ensure_fr(goicp_engine* h, size_t jobs, size_t cap, size_t ecap) {
    // per-search arrays (state, results, rotated cloud, work list) grow with the number of searches; the queue and expansion slabs
    // with searches x capacity (a search that outgrew its slab is re-run alone on a larger one: few searches, large capacity)
    if (jobs > h->fr_jobs) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        { void* fr[] = {h->fr_st, h->fr_wl, h->fr_srot, h->fr_results}; for (void* p : fr) if (p) cache_free(p); }
        h->fr_st = nullptr; h->fr_wl = nullptr; h->fr_srot = nullptr; h->fr_results = nullptr; h->fr_jobs = 0;
        const size_t wl_cap = std::min<size_t>(jobs * (size_t)h->fr_kmax * 9 * (size_t)h->fr_nsplit * 2, (size_t)1 << 26);
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
int run_fr(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results, int cap) {
    // expansion records per search: speculation (several nodes per round, two levels deep) consumes them faster than the queue grows
    const int ecap = (int)std::min<long long>((long long)cap * (njobs <= 128 ? 8 : 2), 1ll << 27);
    int rc = ensure_fr(h, (size_t)njobs, (size_t)cap, (size_t)ecap);
    if (rc != GOICP_OK) return rc;
    if (h->fr_mode == 2) { rc = ensure_stream_scratch(h, (size_t)h->fr_wgs, false); if (rc != GOICP_OK) return rc; }
    HIPCHK(h, hipMemcpyAsync(h->d_jobs, jobs, (size_t)njobs * sizeof(goicp_bnb_job), hipMemcpyHostToDevice, h->stream));
    FrCtx C;
    C.jobs = h->d_jobs; C.st = h->fr_st; C.results = h->fr_results; C.heaps = h->fr_heaps; C.heap_cap = cap; C.exps = h->fr_exps; C.exp_cap = ecap;
    C.wl = h->fr_wl; C.wl_cap = h->fr_wl_cap; C.wl_count = h->fr_cnt; C.active = h->fr_cnt + FR_ROUNDS_MAX; C.parts = h->fr_parts; C.nsplit = h->fr_nsplit; C.ngroups = h->fr_mode == 1 ? 2 : 1;
    C.srot = h->fr_srot; C.NdP = h->T.NdP; C.max_pops = h->max_pops; C.dbg = DBG_ON(h->T, 32) ? h->T.dbg_times : nullptr;
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
        const int K = std::max(1, std::min(h->fr_kmax, target / (int)std::max(1ll, (long long)active * slots * per_tile)));
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

