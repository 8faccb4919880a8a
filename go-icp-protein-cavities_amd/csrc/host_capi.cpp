// host_capi.cpp -- extern "C" wrappers of include/goicp_c.h over the GoICP class and the file pipeline.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include "goicp.hpp"
#include <algorithm>
#include "goicp_c.h"
#include "host_io.hpp"

using namespace goicp;

struct goicp_handle {
    GoICP g;
    std::vector<POINT3D> model, data;
    double cfg[18];
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" {

goicp_handle* goicp_new(void) {
    goicp_handle* h = new goicp_handle();
    std::memset(h->cfg, 0, sizeof h->cfg);
    return h;
}
void goicp_free(goicp_handle* h) { delete h; }

int goicp_set_config(goicp_handle* h, const double* cfg) {
    if (!h || !cfg) return GOICP_ERR_ARG;
    std::memcpy(h->cfg, cfg, sizeof h->cfg);
    apply_config(h->g, cfg);
    return GOICP_OK;
}
int goicp_read_config(goicp_handle* h, const char* path) {
    if (!h || !path) return GOICP_ERR_ARG;
    double cfg[18];
    if (!read_config(path, cfg)) return GOICP_ERR_ARG;
    return goicp_set_config(h, cfg);
}
int goicp_get_config(goicp_handle* h, double* cfg) { if (!h || !cfg) return GOICP_ERR_ARG; std::memcpy(cfg, h->cfg, sizeof h->cfg); return GOICP_OK; }

int goicp_set_options(goicp_handle* h, int device_id, int parity_mode, int speculate, int verbose) {
    if (!h) return GOICP_ERR_ARG;
    h->g.device_id = device_id; h->g.parity_mode = parity_mode; h->g.speculate = speculate; h->g.verbose = verbose;
    return GOICP_OK;
}

static void fill(std::vector<POINT3D>& v, int n, const float* xyz, const int32_t* col, const float* fp) {
    v.assign((size_t)n, POINT3D());
    for (int i = 0; i < n; i++) {
        v[i].x = xyz[3 * i]; v[i].y = xyz[3 * i + 1]; v[i].z = xyz[3 * i + 2];
        v[i].c = col ? col[i] : 0;
        v[i].cfpfh.assign(41, 0.f);
        if (fp) for (int j = 0; j < 41; j++) v[i].cfpfh[j] = fp[41 * i + j];
    }
}

int goicp_set_clouds(goicp_handle* h, int Nm, const float* mxyz, const int32_t* mcol, const float* mfpfh,
                     int Nd, const float* dxyz, const int32_t* dcol, const float* dfpfh) {
    if (!h || Nm <= 0 || Nd <= 0 || !mxyz || !dxyz) return GOICP_ERR_ARG;
    // non-finite coordinates walk the reference's selection and sorting loops out of their arrays; refuse them here
    for (size_t k = 0; k < (size_t)3 * Nm; k++) if (!std::isfinite(mxyz[k])) { h->g.status = GOICP_ERR_ARG; h->g.error_message = "model cloud holds a non-finite coordinate"; return GOICP_ERR_ARG; }
    for (size_t k = 0; k < (size_t)3 * Nd; k++) if (!std::isfinite(dxyz[k])) { h->g.status = GOICP_ERR_ARG; h->g.error_message = "data cloud holds a non-finite coordinate"; return GOICP_ERR_ARG; }
    fill(h->model, Nm, mxyz, mcol, mfpfh); fill(h->data, Nd, dxyz, dcol, dfpfh);
    h->g.pModel = h->model.data(); h->g.Nm = Nm; h->g.pData = h->data.data(); h->g.Nd = Nd;
    return GOICP_OK;
}
int goicp_set_nd(goicp_handle* h, int Nd) { if (!h || Nd <= 0 || Nd > (int)h->data.size()) return GOICP_ERR_ARG; h->g.Nd = Nd; return GOICP_OK; }

int goicp_build_dt(goicp_handle* h) { if (!h || !h->g.pModel) return GOICP_ERR_ARG; h->g.BuildDT(); return GOICP_OK; }

int goicp_load_dt(goicp_handle* h, int S, const double* g4, const float* dist, const int32_t* nearest, const int32_t* cell_color,
                  const int32_t* cell_ptr, const int32_t* cell_idx) {
    if (!h || S <= 0 || !g4 || !dist) return GOICP_ERR_ARG;
    DT3D& d = h->g.dt;
    h->g.Clear();                 // an engine that adopted the device-built grid in place must be gone (its streams drained) before the blocks go back to the pool
    d.ReleaseGpu();
    const size_t n = (size_t)S * S * S;
    d.SIZE = S; d.xMin = g4[0]; d.yMin = g4[1]; d.zMin = g4[2]; d.scale = g4[3];
    d.xMax = d.xMin + S / d.scale; d.yMax = d.yMin + S / d.scale; d.zMax = d.zMin + S / d.scale;
    d.dist.assign(dist, dist + n);
    if (nearest && cell_color && cell_ptr && cell_idx) {
        d.nearest.assign(nearest, nearest + n); d.cell_color.assign(cell_color, cell_color + n);
        d.cell_ptr.assign(cell_ptr, cell_ptr + n + 1); d.cell_idx.assign(cell_idx, cell_idx + cell_ptr[n]);
    } else {
        d.nearest.resize(n); for (size_t k = 0; k < n; k++) d.nearest[k] = (int32_t)k;
        d.cell_color.assign(n, -2); d.cell_ptr.assign(n + 1, 0); d.cell_idx.assign(1, 0);
    }
    return GOICP_OK;
}
int goicp_dt_geom(goicp_handle* h, double* g) {
    if (!h || !g) return GOICP_ERR_ARG;
    const DT3D& d = h->g.dt;
    g[0] = d.xMin; g[1] = d.xMax; g[2] = d.yMin; g[3] = d.yMax; g[4] = d.zMin; g[5] = d.zMax; g[6] = d.scale; g[7] = d.SIZE;
    return GOICP_OK;
}
int goicp_set_icp_one_launch(goicp_handle* h, int on) { if (!h) return GOICP_ERR_ARG; h->g.device_icp = on != 0; return GOICP_OK; }
int goicp_set_dt_mode(goicp_handle* h, int mode) { if (!h || (mode != 0 && mode != 1)) return GOICP_ERR_ARG; h->g.dt_mode = mode; return GOICP_OK; }

int goicp_dt_dump(goicp_handle* h, float* dist, int32_t* nearest, int32_t* cell_color) {
    if (!h) return GOICP_ERR_ARG;
    const DT3D& d = h->g.dt;
    if (d.gpu) {
        const int rc = goicp_dtgpu_download(d.gpu, dist, nearest);
        if (rc != GOICP_OK) return rc;
        if (cell_color) {
            const size_t n = (size_t)d.SIZE * d.SIZE * d.SIZE;
            for (size_t k = 0; k < n; k++) cell_color[k] = -2;
            for (size_t o = 0; o < d.occ_cells.size(); o++) cell_color[d.occ_cells[o]] = d.occ_color[o];
        }
        return GOICP_OK;
    }
    if (dist) std::memcpy(dist, d.dist.data(), d.dist.size() * sizeof(float));
    if (nearest) std::memcpy(nearest, d.nearest.data(), d.nearest.size() * sizeof(int32_t));
    if (cell_color) std::memcpy(cell_color, d.cell_color.data(), d.cell_color.size() * sizeof(int32_t));
    return GOICP_OK;
}

int goicp_initialize(goicp_handle* h) { if (!h || !h->g.pData) return GOICP_ERR_ARG; h->g.Initialize(); return h->g.status; }
goicp_engine* goicp_get_engine(goicp_handle* h) { return h ? h->g.engine() : nullptr; }
void goicp_clear(goicp_handle* h) { if (h) h->g.Clear(); }
int goicp_get_weights(goicp_handle* h, float* w) { if (!h || !w) return GOICP_ERR_ARG; std::memcpy(w, h->g.weights.data(), h->g.weights.size() * sizeof(float)); return GOICP_OK; }
int goicp_get_maxrotdis(goicp_handle* h, float* m) { if (!h || !m) return GOICP_ERR_ARG; std::memcpy(m, h->g.maxRotDis.data(), h->g.maxRotDis.size() * sizeof(float)); return GOICP_OK; }
int goicp_inlier_num(goicp_handle* h) { return h ? h->g.inlierNum : 0; }
float goicp_sse_thresh(goicp_handle* h) { return h ? h->g.SSEThresh : 0.f; }
void goicp_set_opt_error(goicp_handle* h, float e) { if (h) h->g.optError = e; }
void goicp_set_init_trans(goicp_handle* h, float x, float y, float z, float w, float lb) {
    if (!h) return;
    h->g.initNodeTrans.x = x; h->g.initNodeTrans.y = y; h->g.initNodeTrans.z = z; h->g.initNodeTrans.w = w; h->g.initNodeTrans.lb = lb;
}

float goicp_icp(goicp_handle* h, double* R9, double* t3, int32_t* corr) {
    if (!h || !R9 || !t3) return std::numeric_limits<float>::quiet_NaN();
    Matrix R(3, 3), t(3, 1);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R.val[i][j] = R9[3 * i + j]; t.val[i][0] = t3[i]; }
    const float e = h->g.ICP(R, t);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R9[3 * i + j] = R.val[i][j]; t3[i] = t.val[i][0]; }
    if (corr) {                      // icp3d.points of the run just made (reference jly_icp3d.hpp:247-249, after the trimmed sort)
        const std::vector<int32_t>& idd = h->g.icp_data_ids(); const std::vector<int32_t>& idm = h->g.icp_model_ids();
        for (size_t i = 0; i < idd.size() && i < idm.size(); i++) { corr[2 * i] = idd[i]; corr[2 * i + 1] = idm[i]; }
    }
    return e;
}

int goicp_icp_distances(goicp_handle* h, double* dis) {
    if (!h || !dis) return GOICP_ERR_ARG;
    const std::vector<double>& d = h->g.icp_distances();
    for (size_t i = 0; i < d.size(); i++) dis[i] = d[i];
    return (int)d.size();
}

int goicp_kdorder_nn(const float* model_xyz, int Nm, const float* queries_xyz, int n, int32_t* id_model, float* dist_sq) {
    if (!model_xyz || Nm <= 0 || n < 0 || (n && (!queries_xyz || !id_model))) return GOICP_ERR_ARG;
    KdOrder kd; kd.build(model_xyz, Nm);
    for (int i = 0; i < n; i++) { float d = 0; id_model[i] = kd.nearest(queries_xyz + 3 * (size_t)i, &d); if (dist_sq) dist_sq[i] = d; }
    return GOICP_OK;
}

static void pack16(const GoICP& g, double* out) {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out[3 * i + j] = g.optR.val[i][j]; out[9 + i] = g.optT.val[i][0]; }
    out[12] = g.optError; out[13] = g.optComp; out[14] = g.Nd; out[15] = g.inlierNum;
}

float goicp_register(goicp_handle* h, double* out16) {
    if (!h || !h->g.pData || !h->g.pModel) return std::numeric_limits<float>::quiet_NaN();
    const float e = h->g.Register();
    if (out16) pack16(h->g, out16);
    return e;
}
int goicp_status(goicp_handle* h) { return h ? h->g.status : GOICP_ERR_ARG; }
const char* goicp_error_message(goicp_handle* h) { return h ? h->g.error_message.c_str() : "null handle"; }

int goicp_get_trace(goicp_handle* h, int32_t* tags, float* values, int cap) {
    if (!h) return 0;
    const int n = (int)h->g.trace.size();
    for (int i = 0; i < n && i < cap; i++) {
        const std::string& t = h->g.trace[i].first;
        if (tags) tags[i] = t == "Init" ? 0 : (t == "ICP" ? 1 : 2);
        if (values) values[i] = h->g.trace[i].second;
    }
    return n;
}
int goicp_get_stats(goicp_handle* h, uint64_t* s, double* ms) {
    if (!h) return GOICP_ERR_ARG;
    const RegisterStats& st = h->g.stats;
    if (s) { s[0] = st.rot_pops; s[1] = st.inner_calls; s[2] = st.inner_launches; s[3] = st.inner_jobs_launched; s[4] = st.cube_evals; s[5] = st.trans_pops; s[6] = st.icp_calls; s[7] = st.icp_iters; }
    if (ms) *ms = st.engine_kernel_ms;
    return GOICP_OK;
}

int goicp_get_search_stats(goicp_handle* h, uint64_t* s) {
    if (!h || !s) return GOICP_ERR_ARG;
    const RegisterStats& st = h->g.stats;
    s[0] = st.pops_launched; s[1] = st.search_rounds; s[2] = st.cube_evals_launched; s[3] = st.slab_regrowths; s[4] = (uint64_t)(st.seconds * 1e6); s[5] = h->g.budget_hit ? 1 : 0;
    return GOICP_OK;
}
int goicp_set_budget(goicp_handle* h, double wall_seconds, uint32_t search_pops) {
    if (!h || !(wall_seconds >= 0)) return GOICP_ERR_ARG;
    h->g.wall_budget_s = wall_seconds; h->g.search_pop_budget = search_pops;
    return GOICP_OK;
}
int goicp_budget_hit(goicp_handle* h) { return h && h->g.budget_hit ? 1 : 0; }

void goicp_record_jobs(goicp_handle* h, int on) { if (h) h->g.record_jobs = on; }
int goicp_get_job_log(goicp_handle* h, goicp_bnb_job* jobs, goicp_bnb_result* results, int cap) {
    if (!h) return 0;
    const int n = (int)h->g.job_log.size();
    for (int i = 0; i < n && i < cap; i++) { if (jobs) jobs[i] = h->g.job_log[i]; if (results) results[i] = h->g.result_log[i]; }
    return n;
}

// ---- file pipeline (reference jly_main.cpp:54-179) -------------------------------------------------------
static std::string stem_between(const std::string& s) {   // substr(find("/")+1, find(".")-find("/")-1), jly_main.cpp:65-66
    const size_t a = s.find("/"), b = s.find(".");
    return s.substr(a + 1, b - a - 1);
}

// "<output stem>_rescaled.txt": the reference cuts the output path at its first '.' (jly_main.cpp:155). Here the cut is the first '.'
// of the FILE NAME, so that a dot in a directory ("./out", "run.1/") cannot map every row to one file; for the reference's own layout
// (output/similarK.txt) the names are the same.
static std::string rescaled_name(const std::string& o) {
    const size_t slash = o.find_last_of('/');
    const size_t dot = o.find('.', slash == std::string::npos ? 0 : slash + 1);
    return o.substr(0, dot) + "_rescaled.txt";
}

static int finish(GoICP& g, double* res, double t_reg, double t_dt, const double resc[3]) {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) res[3 * i + j] = g.optR.val[i][j]; res[9 + i] = g.optT.val[i][0]; }
    res[12] = g.optError; res[13] = g.Nd - g.optComp; res[14] = g.Nd; res[15] = t_reg; res[16] = t_dt;
    res[17] = resc[0]; res[18] = resc[1]; res[19] = resc[2];
    res[20] = (double)g.stats.cube_evals; res[21] = (double)g.stats.trans_pops; res[22] = (double)g.stats.rot_pops; res[23] = g.status;
    return g.status;
}

static std::atomic<int> g_batch_speculate{-1};
void goicp_set_batch_speculation(int depth) { g_batch_speculate.store(depth); }
static std::atomic<int> g_batch_icp_loop{-1};
void goicp_set_batch_icp_one_launch(int on) { g_batch_icp_loop.store(on); }

// what the file pipeline's registrations of this process launched and what their replays consumed (speculation cost of a batch)
static std::atomic<uint64_t> g_tot[6];
static void add_totals(const GoICP& g) {
    const uint64_t v[6] = {1, g.stats.pops_launched, g.stats.trans_pops, g.stats.inner_jobs_launched, g.stats.inner_calls, g.stats.inner_launches};
    for (int k = 0; k < 6; k++) g_tot[k].fetch_add(v[k], std::memory_order_relaxed);
}
void goicp_batch_search_totals(uint64_t* out6, int reset) {
    for (int k = 0; k < 6; k++) { const uint64_t v = reset ? g_tot[k].exchange(0, std::memory_order_relaxed) : g_tot[k].load(std::memory_order_relaxed); if (out6) out6[k] = v; }
}

int goicp_run_pair_files(const char* base_dir, const char* target_mol2, const char* source_mol2, int nd_down, const char* config_path,
                         const char* output_path, int pair, int device_id, int verbose, double* res) {
    if (!target_mol2 || !source_mol2 || !config_path || !output_path || !res) return GOICP_ERR_ARG;
    const std::string base = (base_dir && base_dir[0]) ? std::string(base_dir) + "/" : std::string();
    GoICP g;
    g.device_id = device_id; g.verbose = verbose;
    { const int v = g_batch_icp_loop.load(); if (v >= 0) g.device_icp = v != 0; }
    { const int d = g_batch_speculate.load(); if (d >= 0) g.speculate = d; }                              // what the batch driver chose for its number of registrations in flight
    { const char* e = getenv("GOICP_SPECULATE"); if (e) g.speculate = std::max(0, atoi(e)); }           // experiment knobs of the batch path
    { const char* e = getenv("GOICP_SPEC_POPS"); if (e && atoi(e) > 0) g.spec_pop_budget = (uint32_t)atoi(e); }
    double cfg[18];
    if (!read_config(config_path, cfg)) return GOICP_ERR_ARG;
    apply_config(g, cfg);
    const std::string modelF = target_mol2, dataF = source_mol2;
    const std::string proteinData = stem_between(dataF), proteinModel = stem_between(modelF);
    std::vector<Atom> src, tgt;
    if (!read_mol2(base + dataF, src) || !read_mol2(base + modelF, tgt) || src.empty() || tgt.empty()) return GOICP_ERR_ARG;
    double smean[3], tmean[3];
    const double ss = normalise(src, smean), ts = normalise(tgt, tmean);
    const double scale = ss >= ts ? ss : ts;
    for (Atom& p : src) { p.x /= scale; p.y /= scale; p.z /= scale; }
    for (Atom& p : tgt) { p.x /= scale; p.y /= scale; p.z /= scale; }
    const std::string srcN = "cavitiesN/" + proteinData + "_sim" + std::to_string(pair) + "N.xyz";
    const std::string tgtN = "cavitiesN/" + proteinModel + "_sim" + std::to_string(pair) + "N.xyz";
    if (!write_xyzc(base + srcN, src) || !write_xyzc(base + tgtN, tgt)) return GOICP_ERR_ARG;
    auto cf = [&](const std::string& f) { return base + "cfpfh/" + f.substr(f.find("/") + 1, f.find_last_of("_") - f.find("/") - 1) + ".cfpfh"; };
    std::vector<POINT3D> model, data;
    if (!load_xyzc(base + tgtN, cf(tgtN), model) || !load_xyzc(base + srcN, cf(srcN), data)) return GOICP_ERR_ARG;
    g.pModel = model.data(); g.Nm = (int)model.size(); g.pData = data.data(); g.Nd = (int)data.size();
    double t0 = now_s();
    g.BuildDT();
    const double t_dt = now_s() - t0;
    if (nd_down > 0 && nd_down <= g.Nd) g.Nd = nd_down;
    t0 = now_s();
    g.Register();
    const double t_reg = now_s() - t0;
    double resc[3] = {0, 0, 0};
    if (g.status == GOICP_OK) {
        write_output(output_path, t_reg, g);
        rescaled_translation(g, scale, smean, tmean, resc);
        write_rescaled(rescaled_name(output_path), t_reg, g, resc);
    } else if (verbose) std::cerr << "goicp: " << g.error_message << std::endl;
    add_totals(g);
    return finish(g, res, t_reg, t_dt, resc);
}

int goicp_run_txt_files(const char* model_txt, const char* data_txt, int nd_down, const char* config_path, const char* output_path,
                        int device_id, int verbose, double* res) {
    if (!model_txt || !data_txt || !config_path || !res) return GOICP_ERR_ARG;
    GoICP g;
    g.device_id = device_id; g.verbose = verbose;
    double cfg[18];
    if (!read_config(config_path, cfg)) return GOICP_ERR_ARG;
    apply_config(g, cfg);
    std::vector<POINT3D> model, data;
    if (!load_txt(model_txt, model) || !load_txt(data_txt, data) || model.empty() || data.empty()) return GOICP_ERR_ARG;
    g.pModel = model.data(); g.Nm = (int)model.size(); g.pData = data.data(); g.Nd = (int)data.size();
    double t0 = now_s();
    g.BuildDT();
    const double t_dt = now_s() - t0;
    if (nd_down > 0 && nd_down <= g.Nd) g.Nd = nd_down;
    t0 = now_s();
    g.Register();
    const double t_reg = now_s() - t0;
    const double resc[3] = {0, 0, 0};
    if (g.status == GOICP_OK && output_path && output_path[0]) write_output(output_path, t_reg, g);
    return finish(g, res, t_reg, t_dt, resc);
}

float goicp_compute_rmsd(const char* aligned, const char* rot) { return (aligned && rot) ? compute_rmsd(aligned, rot) : NAN; }
int goicp_apply_transformation_protein(const char* output_txt, const char* protein, const char* out_mol2) {
    if (!output_txt || !protein || !out_mol2) return GOICP_ERR_ARG;
    return apply_transformation_protein(output_txt, protein, out_mol2) ? GOICP_OK : GOICP_ERR_ARG;
}

}  // extern "C"
