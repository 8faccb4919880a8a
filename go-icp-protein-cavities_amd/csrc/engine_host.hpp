// engine_host.hpp -- host side shared by the translation units of libgoicp_hip.so: the engine handle, the pooled-resource helpers and the
// per-family launchers (defined next to their kernels so that each kernel family compiles in its own translation unit).
#pragma once
#include "engine_dev.hpp"

struct goicp_engine {
    int device;
    hipStream_t stream;
    int upload_defer;                         // goicp_engine_create: uploads are waited for once, at its end (their sources live that long)
    hipStream_t stream_bulk;                  // search launches of cavity-sized engines (see cache_stream_aux); null: they use `stream`
    hipStream_t stream_hi;                    // high-priority stream for the small latency-critical kernels (ICP 1-NN, point evaluation): beside other
                                              // engines' bulk search launches they would otherwise queue behind thousands of search workgroups
    hipEvent_t ev_wait; int wait_block;       // host waits: 0 spin on the stream (default), 1 sleep on a blocking event (GOICP_WAIT=block), 2 poll with naps (GOICP_WAIT=nap[:spin_us[:nap_us]])
    int wait_spin_us, wait_nap_us;
    hipEvent_t ev_hi; int hi_shared;           // stream_hi is one of a few shared by all engines (GOICP_HI_SHARE): waits go through this engine's own event
    void* pinned;                             // PINNED_BYTES of host memory (may be null), travels with the stream in the pool
    hipEvent_t ev0, ev1;
    DevTables T;
    std::vector<void*> allocs;
    // device buffers
    float* d_mxyz; float4* d_rot; float* d_R; int nslots;
    float4* d_mxs; int* d_mxb; int* d_prev; float mx_x0, mx_inv; int nn_slab;   // model sorted along x + bucket starts + last match per data point (k_icp_nn_slab)
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
    int fr_ppt, fr_block, fr_nsplit, fr_wgs, fr_heap_cap, fr_depth2, fr_kmax;
    FrState* fr_st; HeapNodeX* fr_heaps; FrExp* fr_exps; FrItem* fr_wl; unsigned* fr_cnt; double* fr_parts; float4* fr_srot; goicp_bnb_result* fr_results;
    size_t fr_jobs, fr_cap_nodes, fr_cap_exps; int fr_wl_cap;
    uint64_t fr_rounds, fr_spec_cubes, n_regrow;
    unsigned max_pops;
    uint64_t n_cubes, n_lookups; double kernel_ms;
    std::string err, kernel_name;
    std::vector<int4> pairs_host, groups4_host;
    unsigned long long code_escapes;          // cells of the coded grid that fall back to the float grid
};


#include "pool.hpp"
inline int fail(goicp_engine* h, int code, const std::string& msg) { if (h) h->err = msg; return code; }
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, GOICP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
// host wait for everything queued on `st` so far
inline hipError_t wait_stream(goicp_engine* h, hipStream_t st) {
    if (h->hi_shared && st == h->stream_hi) {                             // other engines' small kernels share the stream: wait for our own work only
        const hipError_t e = hipEventRecord(h->ev_hi, st);
        if (e != hipSuccess) return e;
        return h->wait_block == 2 ? wait_event_nap(h->ev_hi, h->wait_spin_us, h->wait_nap_us) : hipEventSynchronize(h->ev_hi);
    }
    if (h->wait_block == 2) return wait_stream_nap(st, h->wait_spin_us, h->wait_nap_us);
    if (h->wait_block && h->ev_wait) {
        hipError_t e = hipEventRecord(h->ev_wait, st);
        return e == hipSuccess ? hipEventSynchronize(h->ev_wait) : e;
    }
    return hipStreamSynchronize(st);
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


inline bool env_is_zero(const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; }
int ensure_fpscratch(goicp_engine* h, size_t blocks);           // engine.hip
int ensure_stream_scratch(goicp_engine* h, size_t wgs, bool need_rot, bool need_rows = true);

// per-family entry points
int bnb_configure(goicp_engine* h, int ncu);                    // k_bnb.hip: LDS ceilings of the chosen k_inner_bnb instantiation, h->resident_jobs
int run_bnb(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results, HeapNode* heaps, int cap);
int run_fr(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results, int cap);      // k_fr.hip
int eval_configure(goicp_engine* h);                            // k_eval.hip
int launch_eval(goicp_engine* h, int slot, int rot_level, const int2* groups_dev, int ngroups, const void* cubes_dev, int n, void* ub_dev, void* lb_dev);
