// Optimal similarity alignment (orthogonal Procrustes with scale) from accumulated moments -- the closed-form core of
// the reference's PA-MPJPE / PA-MPVPE (apps/eval_interhand.py:28-75 / common/utils/intag_eval.py:92-143
// `batch_compute_similarity_transform_torch`: K = X1 X2^T, K = U S V^T, R = V diag(1,1,det) U^T, s = tr(R K)/var1,
// t = mu2 - s R mu1).
//
// R = argmax over proper rotations of tr(R K); instead of a 3x3 SVD plus the reflection fix this takes the dominant
// eigenvector of Horn's symmetric 4x4 matrix N(K) (Horn 1987, "Closed-form solution of absolute orientation using unit
// quaternions"): the unit quaternion q maximising q^T N q is that rotation, and the eigenvalue is tr(R K) itself.  The
// eigen-problem is solved by cyclic Jacobi rotations in double precision (4x4: converges in a handful of sweeps).
//
// Plain C++ usable from host and device code: the same source is compiled into the HIP kernel (rih_metrics.hip) and
// into a host harness by tests/test_cpu_host.py, which checks it against numpy's SVD.
#pragma once
#include <math.h>

#ifndef RIH_HD
#if defined(__HIPCC__)
#define RIH_HD __host__ __device__
#else
#define RIH_HD
#endif
#endif

// Eigen-decomposition of a symmetric 4x4 matrix A (destroyed): eigenvalues on the diagonal of A, eigenvectors in the
// columns of V.
RIH_HD inline void rih_jacobi4(double A[4][4], double V[4][4]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 4; ++i) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < 3; ++p) {
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {       // A <- A J  (columns p, q)
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {       // A <- J^T A  (rows p, q)
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; ++k) {       // V <- V J
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
}

// Moments of N point pairs (x1_n, x2_n): s1 = sum x1, s2 = sum x2, s12[a][b] = sum x1_a x2_b, q1 = sum |x1|^2.
// Output: row-major R (maps x1 into the frame of x2), scale, translation t:  x1_hat = scale * R x1 + t.
RIH_HD inline void rih_similarity_from_moments(int N, const double s1[3], const double s2[3], const double s12[3][3],
                                               double q1, double R[3][3], double* scale, double t[3]) {
    double mu1[3], mu2[3], K[3][3];
    for (int a = 0; a < 3; ++a) { mu1[a] = s1[a] / N; mu2[a] = s2[a] / N; }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) K[a][b] = s12[a][b] - N * mu1[a] * mu2[b];       // X1 X2^T
    const double var1 = q1 - N * (mu1[0] * mu1[0] + mu1[1] * mu1[1] + mu1[2] * mu1[2]);
    double Nm[4][4] = {
        {K[0][0] + K[1][1] + K[2][2], K[1][2] - K[2][1], K[2][0] - K[0][2], K[0][1] - K[1][0]},
        {K[1][2] - K[2][1], K[0][0] - K[1][1] - K[2][2], K[0][1] + K[1][0], K[2][0] + K[0][2]},
        {K[2][0] - K[0][2], K[0][1] + K[1][0], -K[0][0] + K[1][1] - K[2][2], K[1][2] + K[2][1]},
        {K[0][1] - K[1][0], K[2][0] + K[0][2], K[1][2] + K[2][1], -K[0][0] - K[1][1] + K[2][2]}};
    double V[4][4];
    rih_jacobi4(Nm, V);
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (Nm[i][i] > Nm[best][best]) best = i;
    const double lam = Nm[best][best];                       // = tr(R K)
    double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    const double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z);     R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y);     R[2][1] = 2 * (y * z + w * x);     R[2][2] = 1 - 2 * (x * x + y * y);
    *scale = lam / var1;
    for (int a = 0; a < 3; ++a)
        t[a] = mu2[a] - (*scale) * (R[a][0] * mu1[0] + R[a][1] * mu1[1] + R[a][2] * mu1[2]);
}
