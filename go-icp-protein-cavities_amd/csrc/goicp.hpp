// goicp.hpp -- host-side mirror of the reference's registration API for the accelerated path.
//
// Same class name, public fields and call sequence as the reference's `class GoICP`
// (reference jly_goicp.h:112-218; caller: jly_main.cpp:61-139): default-construct, fill the public
// configuration fields, assign pModel/Nm/pData/Nd, BuildDT(), optionally lower Nd, Register(), then read
// optR / optT / optError / optComp.  The outer rotation branch-and-bound stays on the host (this file's
// .cpp); every error evaluation goes through the C ABI of include/goicp_engine.h to the HIP kernels.
// There is no CPU evaluation path: Register() returns NaN and sets `status` when no MI355X is usable.
#ifndef GOICP_HPP
#define GOICP_HPP

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "goicp_engine.h"
#include "host_kdorder.hpp"

namespace goicp {

constexpr double PI = 3.1415926536;       // reference jly_goicp.h:44
constexpr double SQRT3 = 1.732050808;     // reference jly_goicp.h:45
constexpr int MAXROTLEVEL = 20;           // reference jly_goicp.h:95

// physico-chemical property codes (reference transformation.hpp:36)
enum properties { OG = 8204959, N = 30894, O = 15219528, NZ = 15231913, CZ = 4646984, CA = 16741671, DU = 7566712, OD1 = 0, C = 1 };
properties string_to_prop(const std::string& type);   // reference transformation.cpp:18-47

struct POINT3D {                          // reference jly_goicp.h:47-56
    float x = 0, y = 0, z = 0;
    int c = 0;
    int neighbors = 0;
    float density = 0;
    std::vector<float> cfpfh;
};

struct ROTNODE { float a, b, c, w; float ub, lb; int l; };      // reference jly_goicp.h:59-73
struct TRANSNODE { float x, y, z, w; float ub, lb; };           // reference jly_goicp.h:75-87

// minimal stand-in for the reference's Matrix results (`optR.val[i][j]`, `optT.val[i][0]`, operator<<
// printing "%12.7f " per element -- reference matrix.cpp:812-827)
struct Matrix {
    int m = 0, n = 0;
    double val[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    Matrix() {}
    Matrix(int m_, int n_) : m(m_), n(n_) {}
    static Matrix eye(int k) { Matrix M(k, k); for (int i = 0; i < k; i++) M.val[i][i] = 1; return M; }
};
std::ostream& operator<<(std::ostream& out, const Matrix& M);

// Distance transform + cell tables (reference jly_3ddt.h:112-139), flattened x-fastest for staging to HBM.
class DT3D {
public:
    int SIZE = 300;                        // reference jly_3ddt.cpp (DT3D::DT3D) default via config
    double scale = 0;
    double expandFactor = 2.0;
    double xMin = 0, xMax = 0, yMin = 0, yMax = 0, zMin = 0, zMax = 0;
    void Build(const double* x, const double* y, const double* z, int num);   // reference jly_3ddt.cpp:897-1137
    std::vector<float> dist;               // [S^3] A.data[z][y][x].distance
    std::vector<int32_t> nearest;          // [S^3] emptyCells -> linear index of the nearest occupied cell
    std::vector<int32_t> cell_color;       // [S^3] cellPoints[].c  (-2 empty, -1 mixed after assignCellColor)
    std::vector<int32_t> cell_ptr;         // [S^3+1] CSR of cellPoints[].points
    std::vector<int32_t> cell_idx;
    // --- addition of this build: the exact transform built and kept on the device (csrc/dt_gpu.hip, GoICP::dt_mode = 1). The dense
    // host vectors above then stay empty; the cell tables exist in sparse form (occupied cells only) and cell_idx is shared.
    goicp_dtgpu* gpu = nullptr;
    std::vector<int32_t> occ_cells, occ_color, occ_ptr;
    int BuildGpu(const double* x, const double* y, const double* z, int num, int device);   // GOICP_* status
    void ReleaseGpu();
    DT3D() {}
    ~DT3D() { ReleaseGpu(); }
    DT3D(const DT3D&) = delete;
    DT3D& operator=(const DT3D&) = delete;
};

struct RegisterStats {
    uint64_t rot_pops = 0, inner_calls = 0, inner_launches = 0, inner_jobs_launched = 0, cube_evals = 0, trans_pops = 0, icp_calls = 0, icp_iters = 0;
    double engine_kernel_ms = 0;
    uint64_t pops_launched = 0;   // queue pops of every launched job, consumed by the replay or not (speculation cost)
    uint64_t search_rounds = 0, cube_evals_launched = 0;   // round-based searches (clouds above 1024 points): select/expand rounds, cube evaluations launched
    uint64_t slab_regrowths = 0;  // searches re-run on a 16x larger queue slab
    double seconds = 0;           // wall time of Register()
};

class GoICP {
public:
    int Nm = 0, Nd = 0;
    POINT3D* pModel = nullptr;
    POINT3D* pData = nullptr;

    ROTNODE initNodeRot;
    TRANSNODE initNodeTrans;
    DT3D dt;
    ROTNODE optNodeRot;
    TRANSNODE optNodeTrans;

    GoICP();
    ~GoICP();
    float Register();
    void BuildDT();

    float MSEThresh = 0.001f;
    float SSEThresh = 0;
    float icpThresh = 0;
    float optError = 0;
    Matrix optR;
    Matrix optT;
    float trimFraction = 0;
    int inlierNum = 0;
    bool doTrim = true;

    float regularization = 0;
    float regularizationNeighbors = 0;
    float regularizationFPFH = 0;
    int optComp = 0;
    int norm = 2;
    int ponderation = 0;
    int cfpfh = 0;

    // --- additions of this build (not in the reference) ---
    int device_id = 0;
    int parity_mode = 1;          // see goicp_engine_desc::parity_mode
    int dt_mode = 0;              // 0: DT3D::Build restated on the host (bit-identical to the reference grid); 1: exact Euclidean transform built on the device (opt-in, see csrc/dt_gpu.hip)
    int speculate = 128;           // 0: one InnerBnB per launch; 1: the <=16 calls of a rotation pop per launch; M>1: also those of the next M-1 queue nodes
    // ICP3D::Run of cavity-sized clouds (<= 1024 data / 2048 model points) as ONE launch per call (k_icp_loop, one workgroup) instead of one search
    // launch per iteration with the sums on the host. Same bits either way. One CU against the whole chip plus a host core: a registration that has
    // the GPU to itself is slower with it (shipped pair 1: 5.2 vs 4.3 ms; 1024 points: 11.6 vs 2.1 ms), registrations that share the GPU are faster
    // (configs[3] batch, 14 in flight: 634 vs 519 pairs/s) -- so it is off here and the batch driver switches it on (goicp_set_batch_icp_one_launch).
    bool device_icp = false;
    int icp_loop_max = 1024;      // data points up to which ICP3D::Run goes out as one launch (GOICP_ICP_LOOP_MAX)
    uint32_t spec_pop_budget = 20000;   // queue-pop budget of a speculative job (re-run unbounded if it turns out to be needed)
    // budgets for searches that cannot be run to convergence (BASELINE configs[4]: 50 000 points, MSEThresh 1e-5): Register() then returns the best
    // pose found so far with budget_hit = true and status OK. 0 = none (the reference's behaviour).
    double wall_budget_s = 0;     // checked before every rotation-queue pop
    uint32_t search_pop_budget = 0;   // translation-queue pops allowed to ONE InnerBnB call
    bool budget_hit = false;
    int verbose = 0;              // echo the reference's "Error*: ..." progress lines to stdout
    int status = GOICP_OK;        // last engine status
    std::string error_message;
    RegisterStats stats;
    std::vector<std::pair<std::string, float>> trace;   // ("Init"|"ICP"|"BNB", Error*) in order
    int record_jobs = 0;          // keep the consumed InnerBnB jobs (for CPU-baseline replays)
    std::vector<goicp_bnb_job> job_log;
    std::vector<goicp_bnb_result> result_log;

    // state shared with tests through the C API
    std::vector<float> weights;            // [Nd]
    std::vector<float> maxRotDis;          // [20*Nd]
    void Initialize();                     // reference jly_goicp.cpp:180-267 (+ engine creation)
    void Clear();
    goicp_engine* engine() { return eng_; }
    // icp3d.points of the last ICP3D run (reference jly_icp3d.hpp:80-85,247-249): ids and squared distances, in the order Run leaves them
    const std::vector<int32_t>& icp_data_ids() const { return icp_id_data_; }
    const std::vector<int32_t>& icp_model_ids() const { return icp_id_model_; }
    const std::vector<double>& icp_distances() const { return icp_dis_; }
    float ICP(Matrix& R_icp, Matrix& t_icp);                      // reference jly_goicp.cpp:102-178
    float InnerBnB(const float* R9, int rot_level, TRANSNODE* nodeTransOut);   // reference jly_goicp.cpp:286-579 (one job)

private:
    goicp_engine* eng_ = nullptr;
    double icp_ms_ = 0, icp_nn_ms_ = 0, t_start_ms_ = 0;
    std::vector<float> normData_, minDis_;
    std::vector<int32_t> icp_id_data_, icp_id_model_, opt_id_data_, opt_id_model_;
    std::vector<double> icp_dis_;
    std::vector<uint64_t> sort_a_, sort_b_;                    // ICP3D's points as packed (distance bits, data index) words
    std::vector<double> icp_pm_, icp_pd_;                      // the matched model / transformed data points of one ICP iteration
    std::vector<float> mxyz_, dxyz_;                           // packed coordinates of the clouds (set by Initialize)
    bool host_compat_ = false;             // geometric runs on a host-built grid: the compatibility count is taken on the host (no cell tables on the device)
    KdOrder kd_order_;                     // the reference kd-tree's visiting order, built on the first exact distance tie
    float OuterBnB();                      // reference jly_goicp.cpp:582-876
    float icp_run(Matrix& R, Matrix& t);   // reference jly_icp3d.hpp:197-311 (1-NN on the device)
    void neighborsWeights();               // reference jly_goicp.cpp:1453-1498
    int countCompatibilities(const std::vector<int32_t>& id_data, const std::vector<int32_t>& id_model) const;   // :890-914
    int updateCompatibilities();           // reference jly_goicp.cpp:933-946
    void assignCellColor();                // reference jly_goicp.cpp:951-969
    void log_error(const char* tag, float v);
};

// jly_sorting.hpp:229-313 restated (the permutation fixes the float summation order of trimmed sums)
void intro_select(float* data, size_t st, size_t en, size_t k);
// V*diag(1,1,det)*U^T of the cross-covariance H = U W V^T (jly_icp3d.hpp:283-303)

}  // namespace goicp
#endif
