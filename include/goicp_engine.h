/* goicp_engine.h -- C ABI of the MI355X (gfx950) Go-ICP inner-BnB error-evaluation engine.
 *
 * Drop-in boundary: the reference has no FFI layer; its seam is the C++ class GoICP
 * (reference jly_goicp.h:112-218).  Each entry point below replaces one private seam of that class:
 *
 *   goicp_engine_create        <- the device-side copy of what GoICP::BuildDT() (jly_goicp.cpp:79-99) and
 *                                 GoICP::Initialize() (jly_goicp.cpp:180-267) leave behind: DT grid + cell
 *                                 tables (jly_3ddt.h:64-139), weights[], maxRotDis[][], inlierNum, flags
 *   goicp_engine_set_rotation  <- pDataTemp = R*pData                         (jly_goicp.cpp:750-756)
 *   goicp_engine_eval_cubes    <- the per-cube sweep of GoICP::InnerBnB       (jly_goicp.cpp:322-550):
 *                                 DT3D::Distance (jly_3ddt.cpp:1139-1191) per point, clamp, trim
 *                                 (intro_select, jly_sorting.hpp:229), ub/lb sums, corner penalty terms
 *                                 (checkCompatibilities jly_goicp.cpp:919, sumFPFH jly_goicp.cpp:1689)
 *   goicp_engine_inner_bnb*    <- float GoICP::InnerBnB(float*, TRANSNODE*)   (jly_goicp.cpp:286-579),
 *                                 called from OuterBnB at jly_goicp.cpp:768 and :861
 *   goicp_engine_point_eval    <- the Nd DT lookups of the initial error (jly_goicp.cpp:602-605), of
 *                                 GoICP::ICP's error step (jly_goicp.cpp:117-137) and the per-point
 *                                 compatibility test of updateCompatibilities (jly_goicp.cpp:933-946)
 *   goicp_engine_icp_nn        <- the 1-NN correspondence search of ICP3D<float>::Run
 *                                 (jly_icp3d.hpp:236-250; nanoflann kd-tree replaced by exhaustive search)
 *
 * Conventions: every call returns a GOICP_* status (never exits the process); the caller owns all host
 * buffers; a handle is bound to one GPU and one HIP stream and is not thread-safe; distinct handles are
 * independent.  All pointers are plain host pointers unless the name ends in _dev.
 * There is no CPU fallback: without a usable HIP device goicp_engine_create fails with GOICP_ERR_NO_DEVICE.
 */
#ifndef GOICP_ENGINE_H
#define GOICP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOICP_OK               0
#define GOICP_ERR_ARG          1   /* bad argument / unsupported configuration */
#define GOICP_ERR_NO_DEVICE    2   /* no HIP device, or HIP runtime failure at create */
#define GOICP_ERR_HIP          3   /* HIP runtime error (see goicp_engine_last_error) */
#define GOICP_ERR_OVERFLOW     4   /* device queue capacity exceeded even after regrowth */
#define GOICP_ERR_LIMIT        5   /* max_pops safety bound reached */

#define GOICP_MAXROTLEVEL 20       /* jly_goicp.h:95 */
#define GOICP_FPFH_BINS   41       /* jly_main.cpp:301 */
#define GOICP_MAX_GRID    1290     /* largest grid side: linear cell indices are int32 throughout (1290^3 < 2^31 <= 1291^3); larger S -> GOICP_ERR_ARG */

typedef struct goicp_engine goicp_engine;

typedef struct goicp_engine_desc {
    /* distance transform (DT3D, jly_3ddt.h:112-139): float distance per cell, x fastest */
    int32_t S;
    double xMin, yMin, zMin, scale;
    const float*   dt_dist;        /* [S^3] */
    /* cell tables for the penalty terms; all NULL when both regularization and regularizationFPFH are 0 */
    const int32_t* nearest_occ;    /* [S^3] linear index (z*S+y)*S+x of the nearest occupied cell (emptyCells) */
    const int32_t* cell_color;     /* [S^3] -2 unoccupied, -1 mixed, else property code (cellPoints[].c) */
    const int32_t* cell_ptr;       /* [S^3+1] CSR offsets into cell_idx (cellPoints[].points) */
    const int32_t* cell_idx;       /* [cell_ptr[S^3]] model point ids */
    /* clouds */
    int32_t Nm;
    const float*   model_xyz;      /* [Nm*3] */
    const int32_t* model_color;    /* [Nm] or NULL */
    const float*   model_fpfh;     /* [Nm*41] or NULL */
    int32_t Nd;
    const float*   data_xyz;       /* [Nd*3] */
    const int32_t* data_color;     /* [Nd] or NULL */
    const float*   data_fpfh;      /* [Nd*41] or NULL */
    /* Initialize() products */
    const float*   weights;        /* [Nd] */
    const float*   maxRotDis;      /* [20*Nd] */
    int32_t inlierNum;
    int32_t doTrim;
    int32_t norm;                  /* 1 or 2 */
    int32_t cfpfh;                 /* 0 / 1 (41 bins) / 2 (33) / 3 (8 colour bins) */
    float   regularization;
    float   regularizationNeighbors;   /* must be 0 (direct-neighbour penalty is out of scope) */
    float   regularizationFPFH;
    int32_t device_id;
    /* 1: reference-order arithmetic for clouds of Nd <= 1024 points (every cavity cloud): float sums taken sequentially over
     *    the point index, and for trimmed runs the reference's own selection (intro_select, same swap sequence) on the
     *    device -- (ub, lb), InnerBnB results and whole registrations are bit-identical to the reference. Larger clouds fall
     *    back to 0.  0: order-free fp64 sums and an exact histogram selection (same inlier multiset; bounds within n*eps of
     *    the reference, so a search may settle on a different epsilon-optimal node).
     * 2: as 1, and ALSO for geometric runs (no penalty terms) above 1024 points (register-resident or streamed): the addend / residual rows of a tile go to HBM
     *    and the same sequential sums / intro_select run on them. Bit-identical bounds, searches and registrations (the configs[2]
     *    pair: same error to the last bit, same queue pops) at about twice the Register() time of mode 1 (0.124 s against 0.066 s;
     *    reference 10.4 s); a stand-alone sweep of many cubes is far slower (row scratch for 1024 tiles at a time: 94 ms against 2 ms). */
    int32_t parity_mode;
    /* Optional device-resident grid (built by goicp_dtgpu_build below): when dt_dist_dev is non-NULL the engine adopts these
     * buffers in place -- nothing is copied or uploaded, and they must outlive the engine; dt_dist / nearest_occ / cell_color /
     * cell_ptr are then ignored and the cell tables come in sparse form (occupied cells only). */
    const void*    dt_dist_dev;    /* device float[S^3] */
    const void*    nearest_occ_dev;/* device int32[S^3]: linear index of the nearest occupied cell */
    int32_t        n_occ;          /* occupied cells */
    const int32_t* occ_cells;      /* [n_occ] linear cell index, ascending */
    const int32_t* occ_color;      /* [n_occ] -1 mixed, else property code (cellPoints[].c after assignCellColor) */
    const int32_t* occ_ptr;        /* [n_occ+1] CSR offsets into cell_idx */
} goicp_engine_desc;

typedef struct goicp_bnb_job {
    float   R[9];          /* rotation applied to the data cloud before the search (row-major) */
    int32_t rot_level;     /* >=0: subtract maxRotDis[rot_level][i]; -1: no rotation uncertainty (ub call) */
    float   optError_in;   /* incumbent at entry (optErrorT = optError, jly_goicp.cpp:297) */
    float   SSEThresh;
    float   init[4];       /* initNodeTrans x,y,z,w */
    float   init_lb;       /* initNodeTrans.lb (0 in the reference) */
    uint32_t max_pops;     /* 0: engine default; else stop with GOICP_ERR_LIMIT after this many queue pops (speculative jobs, caller budgets) */
    uint32_t flags;        /* GOICP_JOB_* */
} goicp_bnb_job;
/* a job that outgrows its queue slab is re-run alone on a 16x larger one when it carries no pop budget (max_pops == 0); a job WITH a budget is
 * taken for speculative and simply reported as GOICP_ERR_OVERFLOW -- unless it asks for the re-run with this flag */
#define GOICP_JOB_REGROW 1u

typedef struct goicp_bnb_result {
    float    optErrorT;    /* return value of InnerBnB */
    float    node[4];      /* best cube x,y,z,w (valid when found != 0) */
    int32_t  found;        /* some cube improved on optError_in */
    int32_t  status;       /* GOICP_OK / GOICP_ERR_OVERFLOW / GOICP_ERR_LIMIT */
    uint32_t pops;         /* queue pops */
    uint32_t cube_evals;   /* cubes evaluated (8 per expanded node) */
} goicp_bnb_result;

int  goicp_engine_create(const goicp_engine_desc* desc, goicp_engine** out);
void goicp_engine_destroy(goicp_engine* h);
const char* goicp_engine_last_error(const goicp_engine* h);

/* number of rotation slots for eval_cubes (each holds one rotated copy of the data cloud on the device) */
int goicp_engine_set_rotation(goicp_engine* h, int slot, const float* R9);
int goicp_engine_num_slots(const goicp_engine* h);
/* InnerBnB jobs (one workgroup each) that are resident on the device at once for this engine's geometry: a batch of at most
 * this many jobs runs as one wave of workgroups. The host's speculative batching sizes its launches with it. */
int goicp_engine_resident_jobs(const goicp_engine* h);
/* Device memory, streams, events and pinned buffers of destroyed engines are parked per process and reused by the next engine
 * (hipMalloc/hipFree/hipStreamCreate synchronise the device); this returns the parked blocks and streams to the runtime.
 * GOICP_CACHE_GB (default 16, 0 = off) bounds what is parked. Call it when a batch of cavity-sized registrations is over and the
 * process goes on to other GPU work: the batch's CU-masked search streams slow every later launch down for as long as they exist
 * (measured: a configs[2]-sized Register() 60-72 ms instead of 43 ms). */
void goicp_engine_release_cached_memory(void);

/* (cube) -> (ub, lb) for n cubes given as x,y,z,w quadruples, rotation slot `slot`, rot_level as in the job */
int goicp_engine_eval_cubes(goicp_engine* h, int slot, int rot_level, const float* cubes, int n, float* ub, float* lb);
/* same on device-resident buffers, asynchronous on the engine's stream (bench / pipelines) */
int goicp_engine_eval_cubes_dev(goicp_engine* h, int slot, int rot_level, const void* cubes_dev, int n, void* ub_dev, void* lb_dev);
/* one launch for ngroups (rotation slot, rot_level) pairs over the same n cubes; ub/lb are [ngroups][n] on the device */
int goicp_engine_eval_cubes_multi_dev(goicp_engine* h, const int32_t* slots, const int32_t* rot_levels, int ngroups,
                                      const void* cubes_dev, int n, void* ub_dev, void* lb_dev);
/* per-lattice-point penalty terms at n translations (x,y,z triplets): incompatibility counts and c-FPFH means */
int goicp_engine_corner_terms(goicp_engine* h, int slot, const float* xyz, int n, int32_t* incompat, float* fpfh);

/* reference-order best-first translation search, fully device-resident; jobs run concurrently */
int goicp_engine_inner_bnb(goicp_engine* h, const goicp_bnb_job* job, goicp_bnb_result* result);
int goicp_engine_inner_bnb_batch(goicp_engine* h, const goicp_bnb_job* jobs, int njobs, goicp_bnb_result* results);

/* q_i = (float)(R*p_i + t) (double R,t); dist[i] = DT(q_i) (unweighted); compat[i] = 1/0 (may be NULL) */
int goicp_engine_point_eval(goicp_engine* h, const double* R9, const double* t3, float* dist, int32_t* compat);
/* q_i = R*p_i + t in float (R,t rounded from double); nearest model point id and squared distance. Models of >= 1024 points are searched
 * through a copy sorted along x, bounded by each point's match of the previous call on this engine (same results as the exhaustive
 * search, GOICP_NN_SLAB=0 forces that one) */
int goicp_engine_icp_nn(goicp_engine* h, const double* R9, const double* t3, int32_t* id_model, float* dist_sq);
/* the same search (kdtree->knnSearch(query, 1, ...), reference jly_icp3d.hpp:245) plus n_tied[i] = number of model points at
 * exactly the minimal float distance of query i (id_model[i] is then the lowest of them). The reference's kd-tree keeps the
 * first such point it visits (reference nanoflann.hpp:1137-1184); the host-side GoICP::ICP resolves those queries with its
 * restatement of that visiting order (csrc/host_kdorder.hpp). */
int goicp_engine_icp_nn_ties(goicp_engine* h, const double* R9, const double* t3, int32_t* id_model, float* dist_sq, int32_t* n_tied);

/* ICP3D<float>::Run (reference jly_icp3d.hpp:197-311) as one launch, for untrimmed clouds of at most 1024 data / 2048 model points:
 * iterates from `state` (pose R, t; the running centroids mu_m, mu_d that the reference never resets; err = the previous
 * iteration's error sum, -1 before the first; iters = iterations done so far) until the reference's stop test fires (status 0: R, t are
 * the pose BEFORE the last correspondence search, err_new its error sum, id_model / dist_sq its correspondences), max_iters is
 * reached (status 2), or an iteration meets an exact distance tie between model points (status 1: the state is the one before that
 * iteration; the caller decides the tie in the reference's kd-tree order and carries on). Sums whose order matters are taken
 * sequentially over the point index, the 3x3 algebra is the host's (kabsch.hpp): same bits as the host loop.
 * GOICP_ERR_ARG when the configuration is outside the above (the caller then iterates with goicp_engine_icp_nn_ties). */
typedef struct { double R[9], t[3], mu_m[3], mu_d[3]; float err, err_new; int32_t iters, status; } goicp_icp_state;
int goicp_engine_icp_loop(goicp_engine* h, goicp_icp_state* state, float err_diff, int max_iters, int32_t* id_model, float* dist_sq);
/* The same for TRIMMED runs (reference jly_icp3d.hpp:254-262): every iteration orders the points by (squared distance, data index) -- the order
 * the reference's qsort leaves under its comparator without equality -- and takes the error sum, the centroids and the cross-covariance over
 * the first `num` = (int)(Nd * (1 - trimFraction)) of them, sequentially in that order; the centroids are still divided by Nd and the stop test
 * uses `num`. id_model / dist_sq come back by data index, as above. */
int goicp_engine_icp_loop_trimmed(goicp_engine* h, goicp_icp_state* state, float err_diff, int max_iters, int num, int32_t* id_model, float* dist_sq);

int goicp_engine_synchronize(goicp_engine* h);
/* cumulative counters since create: cube evaluations, point lookups, kernel milliseconds (HIP events) */
int goicp_engine_stats(const goicp_engine* h, uint64_t* cube_evals, uint64_t* lookups, double* kernel_ms);
/* round-based searches (clouds above 1024 points, geometric error): select/expand rounds launched so far and cube evaluations
 * launched by them (>= the evaluations the replay consumed: the difference is speculation that was not needed); 0 otherwise */
int goicp_engine_search_stats(const goicp_engine* h, uint64_t* rounds, uint64_t* cube_evals_launched);
/* searches (goicp_engine_inner_bnb*) that outgrew their queue slab and were re-run alone on a 16x larger one, since engine creation */
uint64_t goicp_engine_slab_regrowths(const goicp_engine* h);
#ifdef GOICP_PROBES
/* diagnosis build only (make PROBES=1; absent from the shipped library): run ONLY the distance-grid gathers of the last
 * goicp_engine_eval_cubes_multi_dev launch shape (same grid, clouds, cubes, cell arithmetic; nothing kept) and report the kernel time --
 * the memory-system floor of the sweep's access pattern. scratch_dev: >= n/8 floats. */
int goicp_engine_debug_gather_probe(goicp_engine* h, const void* cubes_dev, int n, void* scratch_dev, int variant, float* ms_out);
#endif
/* 1 if the last goicp_engine_eval_cubes_multi_dev call evaluated its (rot_level -1, rot_level >= 0) group pairs of one slot from ONE pass of
 * gathers (fused sweep: OuterBnB's ub / lb calls of a rotation read the same DT cells, reference jly_goicp.cpp:768,861), else 0 */
int goicp_engine_last_eval_fused(const goicp_engine* h);
/* name of the cube-sweep kernel instantiation goicp_engine_eval_cubes* launches for this handle (as profilers print it) */
const char* goicp_engine_eval_kernel_name(goicp_engine* h);
/* opaque hipStream_t of the handle, for callers that time with their own HIP events */
void* goicp_engine_stream(const goicp_engine* h);

/* ---- exact distance transform on the device (opt-in, NOT bit-identical to DT3D::Build) -----------------
 * The reference's DT3D::Build (reference jly_3ddt.cpp:897-1137) is a sequential vector propagation that approximates the
 * Euclidean transform; this builds the exact one (separable, integer squared distances) with the same geometry, seeds and
 * stored value: dist[cell] = dist_of_d2[d2] where d2 = squared cell distance to the nearest occupied cell and the table holds
 * float((double)(float)sqrt((double)d2) / scale) (jly_3ddt.cpp:28,79,1003) computed by the caller; nearest[cell] = linear index
 * of that occupied cell (the emptyCells map of jly_3ddt.cpp:1012-1132). occ_cells = linear indices of the occupied cells. */
typedef struct goicp_dtgpu goicp_dtgpu;
int  goicp_dtgpu_build(int device_id, int S, const int32_t* occ_cells, int n_occ, const float* dist_of_d2, int n_table, goicp_dtgpu** out);
const void* goicp_dtgpu_dist(const goicp_dtgpu* h);       /* device float[S^3] */
const void* goicp_dtgpu_nearest(const goicp_dtgpu* h);    /* device int32[S^3] */
int  goicp_dtgpu_download(const goicp_dtgpu* h, float* dist, int32_t* nearest);   /* either may be NULL */
void goicp_dtgpu_destroy(goicp_dtgpu* h);

/* ---- point-weight precompute (GoICP::neighborsWeights, reference jly_goicp.cpp:1453-1498) ----
 * counts[i] = #{ j != i : ||p_j - p_i|| < lim } with the reference's float/double mixing; one call per radius step. */
typedef struct goicp_nbr_ctx goicp_nbr_ctx;
int  goicp_nbr_create(int device_id, const float* xyz, int n, goicp_nbr_ctx** out);
int  goicp_nbr_count(goicp_nbr_ctx* ctx, double lim, int32_t* counts);
void goicp_nbr_destroy(goicp_nbr_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
