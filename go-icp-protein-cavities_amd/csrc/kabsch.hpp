// kabsch.hpp -- the 3x3 algebra of one ICP step (reference jly_icp3d.hpp:283-307), shared by the host loop (host_goicp.cpp) and the
// device-resident ICP loop (engine.hip: k_icp_loop). Plain IEEE double arithmetic (+, -, *, /, sqrt; built without FMA contraction), so
// both sides produce the same bits.
#pragma once
#include <math.h>
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define GOICP_HD __host__ __device__
#elif defined(__HIPCC__)
#define GOICP_HD __host__ __device__
#else
#define GOICP_HD
#endif

namespace goicp {

GOICP_HD inline void mul3(const double* A, const double* B, double* Cm) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; Cm[3 * i + j] = s; }
}

// 3x3 SVD (one-sided Jacobi) -> Kabsch rotation with the reference's float determinant (jly_icp3d.hpp:283-303)
GOICP_HD inline void kabsch_rotation(const double H[9], double Rout[9]) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++) A[i] = H[i];
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            double al = 0, be = 0, ga = 0;
            for (int i = 0; i < 3; i++) { al += A[3 * i + p] * A[3 * i + p]; be += A[3 * i + q] * A[3 * i + q]; ga += A[3 * i + p] * A[3 * i + q]; }
            if (ga == 0) continue;
            off += fabs(ga) / sqrt(al * be + 1e-300);
            const double zeta = (be - al) / (2 * ga);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
            const double cs = 1 / sqrt(1 + t * t), sn = cs * t;
            for (int i = 0; i < 3; i++) {
                const double ap = A[3 * i + p], aq = A[3 * i + q]; A[3 * i + p] = cs * ap - sn * aq; A[3 * i + q] = sn * ap + cs * aq;
                const double vp = V[3 * i + p], vq = V[3 * i + q]; V[3 * i + p] = cs * vp - sn * vq; V[3 * i + q] = sn * vp + cs * vq;
            }
        }
        if (off < 1e-15) break;
    }
    double w[3]; int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; j++) w[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (w[ord[b]] > w[ord[a]]) { const int t_ = ord[a]; ord[a] = ord[b]; ord[b] = t_; }
    double U[9], Vs[9], W[3];
    for (int j = 0; j < 3; j++) {
        const int s = ord[j]; W[j] = w[s];
        for (int i = 0; i < 3; i++) { Vs[3 * i + j] = V[3 * i + s]; U[3 * i + j] = w[s] > 0 ? A[3 * i + s] / w[s] : 0; }
    }
    if (!(W[2] > 1e-300 * (W[0] + 1))) {      // rank-deficient H: complete U to an orthonormal basis
        if (!(W[1] > 1e-300 * (W[0] + 1))) {
            double a0 = U[0], a1 = U[3], a2 = U[6];
            if (a0 * a0 + a1 * a1 + a2 * a2 == 0) { U[0] = 1; U[3] = 0; U[6] = 0; a0 = 1; a1 = a2 = 0; }
            double bx = fabs(a0) < 0.9 ? 1 : 0, by = fabs(a0) < 0.9 ? 0 : 1, bz = 0;
            const double dp = bx * a0 + by * a1 + bz * a2; bx -= dp * a0; by -= dp * a1; bz -= dp * a2;
            const double nb = sqrt(bx * bx + by * by + bz * bz); U[1] = bx / nb; U[4] = by / nb; U[7] = bz / nb;
        }
        U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    }
    double Ut[9], R0[9], D[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, VD[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ut[3 * i + j] = U[3 * j + i];
    mul3(Vs, Ut, R0);
    const float a = (float)(R0[0] * (R0[4] * R0[8] - R0[5] * R0[7]));
    const float b = (float)(-R0[1] * (R0[3] * R0[8] - R0[5] * R0[6]));
    const float c = (float)(R0[2] * (R0[3] * R0[7] - R0[4] * R0[6]));
    const float det = a + b + c;
    D[8] = det;
    mul3(Vs, D, VD); mul3(VD, Ut, Rout);
}

}  // namespace goicp
