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
#include "engine_host.hpp"

#include "pool.hpp"
#include "kabsch.hpp"
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
    const int s = src_point(T, i);
    const float x = T.dxyz[3 * s], y = T.dxyz[3 * s + 1], z = T.dxyz[3 * s + 2];
    float4 p;
    p.x = R[0] * x + R[1] * y + R[2] * z;
    p.y = R[3] * x + R[4] * y + R[5] * z;
    p.z = R[6] * x + R[7] * y + R[8] * z;
    p.w = T.weights[s];
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
    // four model points per lane and trip, their loads issued together: a wave walks its points alone on its SIMD most of the time, so a
    // trip costs one memory latency however little it computes (one point per trip: 78 latencies per query at 5 000 model points)
    for (int m0 = lane; m0 < T.Nm; m0 += 256) {
        float x[4], y[4], z[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int m = min(m0 + 64 * k, T.Nm - 1); x[k] = mxyz[3 * m]; y[k] = mxyz[3 * m + 1]; z[k] = mxyz[3 * m + 2]; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int m = m0 + 64 * k;
            const float e0 = q0 - x[k], e1 = q1 - y[k], e2 = q2 - z[k];
            const float d = e0 * e0 + e1 * e1 + e2 * e2;
            const bool ok = m < T.Nm, lt = ok & (d < best), eq = ok & (d == best);
            cnt = lt ? 1 : cnt + (eq ? 1 : 0); bi = lt ? m : bi; best = lt ? d : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off, 64); const int oi = __shfl_down(bi, off, 64); const int oc = __shfl_down(cnt, off, 64);
        if (ob < best) { best = ob; bi = oi; cnt = oc; }
        else if (ob == best) { cnt += oc; if (oi < bi) bi = oi; }
    }
    if (lane == 0) { id_out[i] = bi; d_out[i] = best; if (tie_out) tie_out[i] = cnt; }
}

// The same search with the model sorted along x (ICP3D::Run above cavity size: thousands of model points, hundreds of iterations).
// ANY model point bounds the answer from above, and the point matched in the previous iteration bounds it tightly: with
// dub = |q - model[prev]|^2, a point whose (q.x - x)^2 alone exceeds dub cannot be the nearest (float rounding is monotone, so the full
// three-term sum is not smaller than its first term), and the survivors are a contiguous run of the x-sorted model, found through a table
// of NN_BUCKETS bucket starts (one bucket of slack either side covers the rounding of the run's ends). Distances come from the same
// expression on the same floats as k_icp_nn, the minimum, the lowest original index among exact ties and the tie count are those of the
// full scan: every point at the minimal distance lies inside the run. An inlier scans a few per cent of the model, an outlier most of it.
#define NN_BUCKETS 4096
__global__ __launch_bounds__(256) void k_icp_nn_slab(DevTables T, const float* __restrict__ mxyz, const float4* __restrict__ mxs, const int* __restrict__ mxb,
                                                      float x0, float inv, RtArg rt, int* __restrict__ prev,
                                                      int* __restrict__ id_out, float* __restrict__ d_out, int* __restrict__ tie_out) {
    const float* Rt = rt.v;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= T.Nd) return;                         // whole waves leave together
    const float d0 = T.dxyz[3 * i], d1 = T.dxyz[3 * i + 1], d2 = T.dxyz[3 * i + 2];
    const float q0 = Rt[0] * d0 + Rt[1] * d1 + Rt[2] * d2 + Rt[9];
    const float q1 = Rt[3] * d0 + Rt[4] * d1 + Rt[5] * d2 + Rt[10];
    const float q2 = Rt[6] * d0 + Rt[7] * d1 + Rt[8] * d2 + Rt[11];
    const int p = prev[i];
    const float f0 = q0 - mxyz[3 * p], f1 = q1 - mxyz[3 * p + 1], f2 = q2 - mxyz[3 * p + 2];
    const float dub = f0 * f0 + f1 * f1 + f2 * f2;
    const float rr = sqrtf(dub) * 1.000002f;       // |q.x - x| of a survivor, with room for the roundings of the difference, the square and the root
    const float bl = floorf((q0 - rr - x0) * inv), bh = floorf((q0 + rr - x0) * inv);
    const int blo = (int)fminf(fmaxf(bl - 1.f, 0.f), (float)(NN_BUCKETS - 1)), bhi = (int)fminf(fmaxf(bh + 1.f, 0.f), (float)(NN_BUCKETS - 1));
    const int s0 = mxb[blo], s1 = mxb[bhi + 1];
    float best = INFINITY; int bi = 0x7fffffff, cnt = 0;
    for (int t0 = s0 + lane; t0 < s1; t0 += 256) {            // four points per lane and trip, loads issued together (see k_icp_nn)
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = mxs[min(t0 + 64 * k, s1 - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float e0 = q0 - v[k].x, e1 = q1 - v[k].y, e2 = q2 - v[k].z;
            const float d = e0 * e0 + e1 * e1 + e2 * e2;
            const int m = __float_as_int(v[k].w);
            const bool ok = t0 + 64 * k < s1, lt = ok & (d < best), eq = ok & (d == best);
            cnt = lt ? 1 : cnt + (eq ? 1 : 0); bi = lt ? m : (eq ? min(bi, m) : bi); best = lt ? d : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off, 64); const int oi = __shfl_down(bi, off, 64); const int oc = __shfl_down(cnt, off, 64);
        if (ob < best) { best = ob; bi = oi; cnt = oc; }
        else if (ob == best) { cnt += oc; if (oi < bi) bi = oi; }
    }
    if (lane == 0) { id_out[i] = bi; d_out[i] = best; if (tie_out) tie_out[i] = cnt; prev[i] = bi; }
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
__global__ __launch_bounds__(1024) void k_icp_loop(DevTables T, const float* __restrict__ mxyz, IcpState st0, float err_diff, int max_iters, int trim_num,
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
    // trimmed runs (trim_num > 0; jly_icp3d.hpp:254-262): the sums run over the trim_num closest points in the order of the reference's sort by distance --
    // qsort under a comparator without equality = glibc's stable merge sort = ascending (distance, data index). Sorted copies live behind the rest.
    const bool trimmed = trim_num > 0;
    const int cnt_sum = trimmed ? trim_num : n;
    float* pd2 = reinterpret_cast<float*>(pcnt + 1024);   // [3 n] pd in sorted order   (allocated only for trimmed runs)
    float* pm2 = pd2 + 3 * n;                             // [3 n] pm in sorted order
    float* dsq2 = pm2 + 3 * n;                            // [n]   distances, ascending
    const float* spd = trimmed ? pd2 : pd; const float* spm = trimmed ? pm2 : pm; const float* sdsq = trimmed ? dsq2 : dsq;
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
    const bool timing = DBG_ON(T, 32) && T.dbg_times && tid == 0;
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
        if (trimmed) {
            // rank of point i = points that sort before it; the keys (distance, index) are distinct, so the ranks are a permutation
            if (tid < n) {
                const float di = dsq[tid];
                int r = 0;
                for (int j = 0; j < n; j++) { const float dj = dsq[j]; r += (dj < di || (dj == di && j < tid)) ? 1 : 0; }      // (all lanes read the same word: broadcast)
                pd2[3 * r] = pd[3 * tid]; pd2[3 * r + 1] = pd[3 * tid + 1]; pd2[3 * r + 2] = pd[3 * tid + 2];
                pm2[3 * r] = pm[3 * tid]; pm2[3 * r + 1] = pm[3 * tid + 1]; pm2[3 * r + 2] = pm[3 * tid + 2];
                dsq2[r] = di;
            }
            __syncthreads();
        }
        // ---- the seven running sums, each sequential over the point index on a lane of its own (jly_icp3d.hpp:256-276)
        if (tid < 6) {
            const float* src = tid < 3 ? spm + tid : spd + (tid - 3);
            double acc = tid < 3 ? smu_m[tid] : smu_d[tid - 3];
            int i = 0;
            for (; i + 32 <= cnt_sum; i += 32) {                     // 32 reads in flight, then the 32 dependent adds
                float v[32];
#pragma unroll
                for (int q = 0; q < 32; q++) v[q] = src[3 * (i + q)];
#pragma unroll
                for (int q = 0; q < 32; q++) acc += (double)v[q];
            }
            for (; i + 8 <= cnt_sum; i += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = src[3 * (i + q)];
#pragma unroll
                for (int q = 0; q < 8; q++) acc += (double)v[q];
            }
            for (; i < cnt_sum; i++) acc += (double)src[3 * i];
            if (tid < 3) smu_m[tid] = acc; else smu_d[tid - 3] = acc;
        } else if (tid == 6) {
            float e = 0.f;
            int i = 0;
            for (; i + 32 <= cnt_sum; i += 32) {
                float v[32];
#pragma unroll
                for (int q = 0; q < 32; q++) v[q] = sdsq[i + q];
#pragma unroll
                for (int q = 0; q < 32; q++) e = (float)((double)e + (double)v[q]);
            }
            for (; i + 8 <= cnt_sum; i += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = sdsq[i + q];
#pragma unroll
                for (int q = 0; q < 8; q++) e = (float)((double)e + (double)v[q]);
            }
            for (; i < cnt_sum; i++) e = (float)((double)e + (double)sdsq[i]);
            serr_new = e;
        }
        __syncthreads();
        if (tid == 0) {
            const float err = serr, err_new = serr_new;
            if (err > 0 && err - err_new < err_diff * cnt_sum) sstop = 1;      // jly_icp3d.hpp:272 (num points in the sums; the means below still divide by n)
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
            for (; i + 16 <= cnt_sum; i += 16) {                     // 32 reads in flight; the products do not wait for the running sum
                float va[16], vb[16];
#pragma unroll
                for (int q = 0; q < 16; q++) { va[q] = spd[3 * (i + q) + a]; vb[q] = spm[3 * (i + q) + b]; }
                double pr[16];
#pragma unroll
                for (int q = 0; q < 16; q++) pr[q] = ((double)va[q] - ma) * ((double)vb[q] - mb);
#pragma unroll
                for (int q = 0; q < 16; q++) acc += pr[q];
            }
            for (; i + 8 <= cnt_sum; i += 8) {
                float va[8], vb[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { va[q] = spd[3 * (i + q) + a]; vb[q] = spm[3 * (i + q) + b]; }
#pragma unroll
                for (int q = 0; q < 8; q++) acc += ((double)va[q] - ma) * ((double)vb[q] - mb);
            }
            for (; i < cnt_sum; i++) acc += ((double)spd[3 * i + a] - ma) * ((double)spm[3 * i + b] - mb);
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
    if (h->stream_hi && !h->hi_shared) { (void)hipStreamSynchronize(h->stream_hi); cache_stream_aux_return(h->device, 0, h->stream_hi); }
    if (h->ev_hi) { (void)hipEventSynchronize(h->ev_hi); (void)hipEventDestroy(h->ev_hi); }
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
    h->device = d->device_id; h->stream = nullptr; h->ev0 = h->ev1 = nullptr; h->stream_hi = nullptr; h->stream_bulk = nullptr; h->ev_wait = nullptr; h->ev_hi = nullptr; h->hi_shared = 0; h->wait_block = 0; h->upload_defer = 1;
    h->streamed = 0; h->stream_wgs = 0; h->stream_rot_wgs = 0; h->pinned = nullptr;
    h->d_heaps = nullptr; h->d_fpscratch = nullptr; h->heap_cap = 8192; h->heap_jobs = 0; h->fp_blocks = 0; h->d_heap_big = nullptr; h->heap_big_cap = 0;
    h->n_cubes = h->n_lookups = 0; h->kernel_ms = 0; h->max_pops = 200000000u;
    h->code_escapes = 0; h->last_fused = 0;
    h->fr_mode = 0; h->fr_st = nullptr; h->fr_heaps = nullptr; h->fr_exps = nullptr; h->fr_wl = nullptr; h->fr_cnt = nullptr; h->fr_parts = nullptr; h->fr_srot = nullptr; h->fr_results = nullptr;
    h->fr_jobs = 0; h->fr_cap_nodes = 0; h->fr_cap_exps = 0; h->fr_wl_cap = 0; h->fr_rounds = 0; h->fr_spec_cubes = 0; h->n_regrow = 0;
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
    h->hi_shared = hi_share_count() > 0 ? 1 : 0; h->ev_hi = nullptr;
    if (h->hi_shared) CRH(hipEventCreateWithFlags(&h->ev_hi, hipEventDisableTiming));
    { const char* e = getenv("GOICP_WAIT"); h->wait_block = (e && !strcmp(e, "block")) ? 1 : 0; h->wait_spin_us = 40; h->wait_nap_us = 40;
      if (e && !strncmp(e, "nap", 3)) { h->wait_block = 2; int a = 0, b = 0; const int k = sscanf(e, "nap:%d:%d", &a, &b); if (k >= 1 && a >= 0) h->wait_spin_us = a; if (k >= 2 && b > 0) h->wait_nap_us = b; } }
    if (h->wait_block) CRH(hipEventCreateWithFlags(&h->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
    DevTables& T = h->T;
    T.S = d->S; T.xMin = d->xMin; T.yMin = d->yMin; T.zMin = d->zMin; T.scale = d->scale;
    T.Nd = d->Nd; T.Nm = d->Nm; T.doTrim = d->doTrim ? 1 : 0; T.inlierNum = d->doTrim ? d->inlierNum : d->Nd; T.norm = d->norm;
    T.b0 = 0; T.b1 = 0;
    if (d->cfpfh == 1) { T.b0 = 0; T.b1 = 41; } else if (d->cfpfh == 2) { T.b0 = 0; T.b1 = 33; } else if (d->cfpfh == 3) { T.b0 = 33; T.b1 = 41; }
    T.reg = d->regularization; T.regFPFH = d->regularizationFPFH;
    T.use_compat = use_compat; T.use_fpfh = use_fpfh;
    { const char* e = getenv("GOICP_DBG"); T.dbg = e ? atoi(e) : 0; }
    T.regsel = env_is_zero("GOICP_REGSEL") ? 0 : 1;      // (round 2: opt-in behind a debug bit; every == test is green with it, 70.8 k -> 67.9 k cycles per trimmed parity pop)
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
    // 1-NN of ICP3D::Run above cavity size: the model sorted along x, bucket starts, and the last match of every data point (k_icp_nn_slab)
    std::vector<float4> mxs; std::vector<int> mxb;
    h->nn_slab = 0; h->d_mxs = nullptr; h->d_mxb = nullptr; h->d_prev = nullptr; h->mx_x0 = 0.f; h->mx_inv = 0.f;
    if (d->Nm >= 1024 && !env_is_zero("GOICP_NN_SLAB")) {
        float xmin = d->model_xyz[0], xmax = d->model_xyz[0]; bool finite = true;
        for (int m = 0; m < d->Nm; m++) { const float x = d->model_xyz[3 * (size_t)m]; finite = finite && std::isfinite(x); xmin = std::min(xmin, x); xmax = std::max(xmax, x); }
        if (finite && xmax > xmin) {                         // (a model without extent along x, or with a non-finite x, keeps the exhaustive kernel)
            h->mx_x0 = xmin; h->mx_inv = (float)NN_BUCKETS / (xmax - xmin);
            std::vector<int> ord((size_t)d->Nm);
            for (int m = 0; m < d->Nm; m++) ord[(size_t)m] = m;
            std::sort(ord.begin(), ord.end(), [&](int a, int b) { const float xa = d->model_xyz[3 * (size_t)a], xb = d->model_xyz[3 * (size_t)b]; return xa != xb ? xa < xb : a < b; });
            mxs.resize((size_t)d->Nm); mxb.assign(NN_BUCKETS + 1, 0);
            for (int s = 0; s < d->Nm; s++) {
                const int m = ord[(size_t)s]; const float* q = d->model_xyz + 3 * (size_t)m;
                float4 v; v.x = q[0]; v.y = q[1]; v.z = q[2]; memcpy(&v.w, &m, 4);
                mxs[(size_t)s] = v;
                // the kernel's own bucket expression (monotone in x, so the buckets are runs of the sorted model)
                const int b = (int)std::min(std::max(floorf((q[0] - h->mx_x0) * h->mx_inv), 0.f), (float)(NN_BUCKETS - 1));
                mxb[(size_t)b + 1]++;
            }
            for (int b = 0; b < NN_BUCKETS; b++) mxb[(size_t)b + 1] += mxb[(size_t)b];
            CR(dev_upload(h, mxs.data(), mxs.size(), &h->d_mxs));
            CR(dev_upload(h, mxb.data(), mxb.size(), &h->d_mxb));
            CR(dev_upload<int>(h, nullptr, (size_t)d->Nd, &h->d_prev));
            HIPCHK(h, hipMemsetAsync(h->d_prev, 0, (size_t)d->Nd * sizeof(int), h->stream));
            h->nn_slab = 1;
        }
    }

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
    T.perm = nullptr; T.maxRotDisP = nullptr;
    std::vector<int> perm; std::vector<float> mrdP;
    if (d->Nd > 1024 && !T.exact && !T.exact_big && !T.exact_stream && !T.use_compat && !T.use_fpfh && !env_is_zero("GOICP_SORT")) {
        // order-free engines (every sweep and search kernel of a geometric run above 1024 points): the clouds are stored by decreasing norm. A look-up leaves the grid only for points far from the centre, and a wave
        // pays for the out-of-grid extension (double-precision sqrt / divide) whenever ONE of its lanes needs it: with the far points next to each
        // other few waves do. The sums of this path are order-free (fp64), the inlier selection is a multiset: results do not depend on the order.
        perm.resize((size_t)d->Nd);
        std::vector<float> nrm((size_t)d->Nd);
        for (int i = 0; i < d->Nd; i++) { perm[(size_t)i] = i; const float* q = d->data_xyz + 3 * (size_t)i; nrm[(size_t)i] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2]; }
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return nrm[(size_t)a] > nrm[(size_t)b]; });
        mrdP.resize((size_t)GOICP_MAXROTLEVEL * d->Nd);
        for (int l = 0; l < GOICP_MAXROTLEVEL; l++) for (int i = 0; i < d->Nd; i++) mrdP[(size_t)l * d->Nd + i] = d->maxRotDis[(size_t)l * d->Nd + perm[(size_t)i]];
        int* p_perm; CR(dev_upload(h, perm.data(), perm.size(), &p_perm)); T.perm = p_perm;
        float* p_mp; CR(dev_upload(h, mrdP.data(), mrdP.size(), &p_mp)); T.maxRotDisP = p_mp;
    }
    if (h->dyn_lds_bnb + sizeof(TileShm<1024>) + 64 > 160 * 1024) { h->err = "LDS budget exceeded"; CR(GOICP_ERR_ARG); }
    {
        // static tile state + residual rows + LDS heap can exceed the 64 KB a launch gets without asking: each kernel family raises the
        // ceiling of the instantiation this engine will launch (k_eval.hip, k_bnb.hip)
        CR(eval_configure(h));
        // measured (tools/mid_perf.py): 2 000 points 15.2 -> 12.9 ms of kernel time, 4 000 points no change, 5 000 points 7 % slower
        // (122 spilled registers), so only the 1025..2048-point geometry switches. GOICP_JIT_BNB=0 disables.
        h->jit_bnb = (!h->streamed && h->block == 1024 && h->ppt == 2 && !T.exact && !T.use_compat && !T.use_fpfh && !env_is_zero("GOICP_JIT_BNB")) ? 1 : 0;
        {
            int ncu = 0; CRH(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device));   // (a full hipGetDeviceProperties costs ~1 ms per engine)
            CR(bnb_configure(h, ncu));                 // sets h->resident_jobs
            // searches of more than 1024 points without penalty terms run round by round over many workgroups (k_fr_*)
            h->fr_mode = 0; h->fr_nsplit = 1; h->fr_ppt = 4; h->fr_block = 512; h->fr_wgs = 2 * std::max(1, ncu); h->fr_heap_cap = 8192;
            h->fr_kmax = 32; { const char* e = getenv("GOICP_FR_KMAX"); if (e && atoi(e) >= 1 && atoi(e) <= 63) h->fr_kmax = atoi(e); }
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

extern "C" uint64_t goicp_engine_slab_regrowths(const goicp_engine* h) { return h ? h->n_regrow : 0; }

extern "C" int goicp_engine_set_rotation(goicp_engine* h, int slot, const float* R9) {
    if (!h || !R9 || slot < 0 || slot >= h->nslots) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_R, R9, 9 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_rotate, dim3((h->T.Nd + 255) / 256), dim3(256), 0, h->stream, h->T, h->d_R, h->d_rot + (size_t)slot * h->T.Nd);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));      // d_R is reused by the next call
    return GOICP_OK;
}

int ensure_fpscratch(goicp_engine* h, size_t blocks) {
    if (!h->T.use_fpfh) return GOICP_OK;
    if (blocks <= h->fp_blocks) return GOICP_OK;
    if (h->d_fpscratch) { HIPCHK(h, hipStreamSynchronize(h->stream)); cache_free(h->d_fpscratch); h->d_fpscratch = nullptr; h->fp_blocks = 0; }
    HIPCHK(h, cache_malloc((void**)&h->d_fpscratch, blocks * NLMAX * (size_t)h->T.Nd * sizeof(float), h->device));
    h->fp_blocks = blocks;
    return GOICP_OK;
}

int ensure_stream_scratch(goicp_engine* h, size_t wgs, bool need_rot, bool need_rows) {
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

extern "C" const char* goicp_engine_eval_kernel_name(goicp_engine* h) {
    // the instantiation launch_eval() picks for this engine, spelled as rocprofv3 prints it
    if (!h) return "";
    char buf[96];
    const bool jit = h->jit_eval != 0;
    if (h->streamed) snprintf(buf, sizeof buf, "k_eval_cubes_stream<%d,%s>", h->T.Nd > 16384 ? 1024 : 512, (h->last_fused ? "true" : "false"));
    else if (jit) snprintf(buf, sizeof buf, "k_eval_cubes_geo<%d,%s,%s,%d>", 2 * h->ppt, h->npairs > 0 ? "true" : "false", h->T.exact_big ? "true" : "false", (!h->T.exact_big && h->T.norm == 2) ? 2 : 0);   // (DUAL: the group table of the last multi-group call paired up; XBIG: parity_mode 2)
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
            while (r.status == GOICP_ERR_OVERFLOW && (jobs[off + j].max_pops == 0 || (jobs[off + j].flags & GOICP_JOB_REGROW)) && cap < (1 << 26)) {
                cap *= 16; h->n_regrow++;
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
        if (h->nn_slab) hipLaunchKernelGGL(k_icp_nn_slab, dim3((h->T.Nd + 3) / 4), dim3(256), 0, h->stream_hi, h->T, h->d_mxyz, h->d_mxs, h->d_mxb, h->mx_x0, h->mx_inv, rt, h->d_prev, pid, pd, n_tied ? pt : nullptr);
        else hipLaunchKernelGGL(k_icp_nn, dim3((h->T.Nd + 3) / 4), dim3(256), 0, h->stream_hi, h->T, h->d_mxyz, rt, pid, pd, n_tied ? pt : nullptr);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, wait_stream(h, h->stream_hi));
        memcpy(id_model, pid, Nd * sizeof(int)); memcpy(dist_sq, pd, Nd * sizeof(float));
        if (n_tied) memcpy(n_tied, pt, Nd * sizeof(int));
        return GOICP_OK;
    }
    if (h->nn_slab) hipLaunchKernelGGL(k_icp_nn_slab, dim3((h->T.Nd + 3) / 4), dim3(256), 0, h->stream_hi, h->T, h->d_mxyz, h->d_mxs, h->d_mxb, h->mx_x0, h->mx_inv, rt, h->d_prev, h->d_int, h->d_dist, n_tied ? h->d_tie : nullptr);
    else hipLaunchKernelGGL(k_icp_nn, dim3((h->T.Nd + 3) / 4), dim3(256), 0, h->stream_hi, h->T, h->d_mxyz, rt, h->d_int, h->d_dist, n_tied ? h->d_tie : nullptr);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(id_model, h->d_int, (size_t)h->T.Nd * sizeof(int), hipMemcpyDeviceToHost, h->stream_hi));
    HIPCHK(h, hipMemcpyAsync(dist_sq, h->d_dist, (size_t)h->T.Nd * sizeof(float), hipMemcpyDeviceToHost, h->stream_hi));
    if (n_tied) HIPCHK(h, hipMemcpyAsync(n_tied, h->d_tie, (size_t)h->T.Nd * sizeof(int), hipMemcpyDeviceToHost, h->stream_hi));
    HIPCHK(h, wait_stream(h, h->stream_hi));
    return GOICP_OK;
}

static_assert(sizeof(IcpState) == sizeof(goicp_icp_state), "goicp_icp_state");
static int icp_loop_impl(goicp_engine* h, goicp_icp_state* state, float err_diff, int max_iters, int trim_num, int32_t* id_model, float* dist_sq) {
    if (!h || !state || !id_model || !dist_sq || max_iters < 0) return GOICP_ERR_ARG;
    const size_t n = (size_t)h->T.Nd, nm = (size_t)h->T.Nm;
    if (trim_num < 0 || (size_t)trim_num > n) return GOICP_ERR_ARG;
    const size_t lds = (3 * nm + (trim_num > 0 ? 18 : 11) * n + 3 * 1024) * sizeof(float);
    // cavity class only: everything in one workgroup's LDS, results through the pinned buffer
    if (h->T.Nm <= 0 || n > 1024 || nm > 2048 || lds > 120 * 1024 || !h->pinned || sizeof(IcpState) + 8 * n + 64 > PINNED_BYTES) return GOICP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (lds > 48 * 1024) HIPCHK(h, set_max_dyn_lds(reinterpret_cast<const void*>(&k_icp_loop)));
    IcpState st0; memcpy(&st0, state, sizeof st0);
    IcpState* po = (IcpState*)h->pinned;
    int* pid = (int*)((char*)h->pinned + ((sizeof(IcpState) + 63) & ~(size_t)63)); float* pdq = (float*)(pid + n);
    hipLaunchKernelGGL(k_icp_loop, dim3(1), dim3(1024), lds, h->stream_hi, h->T, h->d_mxyz, st0, err_diff, max_iters, trim_num, po, pid, pdq);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, wait_stream(h, h->stream_hi));
    memcpy(state, po, sizeof st0);
    memcpy(id_model, pid, n * sizeof(int)); memcpy(dist_sq, pdq, n * sizeof(float));
    return GOICP_OK;
}
extern "C" int goicp_engine_icp_loop(goicp_engine* h, goicp_icp_state* state, float err_diff, int max_iters, int32_t* id_model, float* dist_sq) {
    if (h && h->T.doTrim) return GOICP_ERR_ARG;               // a trimmed engine iterates over its closest points: goicp_engine_icp_loop_trimmed
    return icp_loop_impl(h, state, err_diff, max_iters, 0, id_model, dist_sq);
}
extern "C" int goicp_engine_icp_loop_trimmed(goicp_engine* h, goicp_icp_state* state, float err_diff, int max_iters, int num, int32_t* id_model, float* dist_sq) {
    if (num <= 0) return GOICP_ERR_ARG;
    return icp_loop_impl(h, state, err_diff, max_iters, num, id_model, dist_sq);
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

