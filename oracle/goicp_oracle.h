/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's hot path.
 *
 * Plain-C, single-threaded restatement of the Go-ICP inner-BnB error-evaluation path of
 * guillaumebaldi/Go-ICP-protein-cavities, written from the behavioural description in SURVEY.md 8(a)/8(a')
 * (file:line citations into /root/reference are given per function in goicp_oracle.c).
 * Parity status: PINNED -- bit-compared against the compiled reference (oracle/_ref, built by oracle/Makefile
 * from the sources where they lie) and against the reference's shipped goldens (output/similar1.txt,
 * demo/output.txt, cavitiesN/ *.xyz); see tests/test_oracle_golden.py and tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (go-icp-protein-cavities_amd/) never links or calls it.
 */
#ifndef GOICP_ORACLE_H
#define GOICP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_new(void);
void     orc_free(orc_ctx*);

/* cfg[18] in the order of jly_main.cpp:236-264 (see oracle/refbind.py CFG_KEYS) */
void orc_set_config(orc_ctx*, const double* cfg);
void orc_set_clouds(orc_ctx*, int Nm, const float* mxyz, const int* mcol, const float* mfpfh,
                    int Nd, const float* dxyz, const int* dcol, const float* dfpfh);
void orc_set_nd(orc_ctx*, int Nd);

/* DT3D::Build restated (jly_3ddt.cpp:897-1137) + assignCellColor (jly_goicp.cpp:951-969) */
void orc_build_dt(orc_ctx*);
/* ... or adopt tables produced elsewhere (the reference's own grid, for hot-path parity runs).
 * nearest3 = (cx,cy,cz) per cell; cell_ptr/cell_idx = CSR cell -> model point ids. geom = xMin,yMin,zMin,scale */
void orc_load_dt(orc_ctx*, int S, const double* geom4, const float* dist, const int* nearest3,
                 const int* cellc, const int* cell_ptr, const int* cell_idx);
void orc_dt_geom(orc_ctx*, double* geom8);
void orc_dt_dump(orc_ctx*, float* dist, short* vhd, int* nearest3, int* cellc);
int  orc_cell_points(orc_ctx*, int x, int y, int z, int* out, int cap);
void orc_distance(orc_ctx*, int n, const double* q, float* out);

void  orc_initialize(orc_ctx*);                 /* GoICP::Initialize, jly_goicp.cpp:180-267 */
void  orc_get_weights(orc_ctx*, float* w);
void  orc_set_weights(orc_ctx*, const float* w);
void  orc_get_maxrotdis(orc_ctx*, float* out);  /* [20*Nd] */
int   orc_inlier_num(orc_ctx*);
float orc_sse_thresh(orc_ctx*);
void  orc_set_sse_thresh(orc_ctx*, float);
void  orc_set_opt_error(orc_ctx*, float);
float orc_get_opt_error(orc_ctx*);
void  orc_set_init_trans(orc_ctx*, float x, float y, float z, float w, float lb);
void  orc_set_rotated(orc_ctx*, const float* xyz);
void  orc_rotate(orc_ctx*, const float* R9);    /* pDataTemp = R*pData, jly_goicp.cpp:750-756 */

/* one translation cube -> (ub, lb) incl. corner penalty terms, jly_goicp.cpp:322-550; level<0: no rho */
void  orc_eval_cube(orc_ctx*, int level, float x, float y, float z, float w, float* ub, float* lb);
float orc_inner_bnb(orc_ctx*, int level, float* node_out6, int want_node);   /* jly_goicp.cpp:286-579 */
void orc_set_pop_budget(orc_ctx*, long long n);   /* test hook: InnerBnB returns after n queue pops (0 = unbounded), like a job's max_pops on the device */
int   orc_check_compatibilities(orc_ctx*, float x, float y, float z);        /* jly_goicp.cpp:919-928 */
float orc_sum_fpfh(orc_ctx*, float x, float y, float z);                     /* jly_goicp.cpp:1689-1697 */
float orc_icp(orc_ctx*, double* R9, double* t3, int* corr, double* dis);     /* jly_goicp.cpp:102-178 */
/* 1-NN of n queries in the visiting order of the reference's kd-tree (nanoflann.hpp:927-1184, jly_icp3d.hpp:141-169) */
void orc_kd_nn(const float* model_xyz, int Nm, const float* queries_xyz, int n, int* id_model, float* dist_sq);
float orc_register(orc_ctx*, double* out16);                                 /* jly_goicp.cpp:878-885 */
int   orc_last_log(orc_ctx*, char* buf, int cap);

void      orc_trace_begin(orc_ctx*, long long cap);
long long orc_trace_end(orc_ctx*, float* out, long long cap);
void      orc_counters(orc_ctx*, long long* out4, int reset);   /* trans pops, rot pops, cube evals, corner evals */

void orc_intro_select(float* data, long long n, long long k);   /* jly_sorting.hpp:229-313 */
void orc_svd3_rot(const double* H9, double* Rout9);             /* V*diag(1,1,det)*U^T of H=U W V^T */
void orc_axis_angle(float v1, float v2, float v3, float* R9);   /* jly_goicp.cpp:729-747 */

#ifdef __cplusplus
}
#endif
#endif
