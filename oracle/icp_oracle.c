/* oracle/icp_oracle.c -- TEST INFRASTRUCTURE ONLY (see icp_oracle.h).
 *
 * Every function restates one piece of the reference CPU path with the same floating-point operation
 * order, so that results are bit-identical to the compiled reference (x86-64, no FMA contraction:
 * build with -ffp-contract=off).  Citations are to files under the reference tree.
 */
#include "icp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* CPUMatrix::mean(0): src/libalg/CPUMatrix.cpp:53-59 -> mean_axises src/libalg/mean.cpp:115-132
 * -> sum_axises mean.cpp:80-113: r[j] += m[i*dim1+j] sequentially in i, then r[j] /= (size_t)dim0. */
void oracle_mean0(const double *A, size_t n, double out[3])
{
    double r0 = 0, r1 = 0, r2 = 0;
    for (size_t i = 0; i < n; ++i) {
        r0 += A[i * 3 + 0];
        r1 += A[i * 3 + 1];
        r2 += A[i * 3 + 2];
    }
    out[0] = r0 / (double)n;
    out[1] = r1 / (double)n;
    out[2] = r2 / (double)n;
}

/* CPUMatrix::operator-(1x3 row): CPUMatrix.cpp:166-173 -> element_wise_op(subtract) src/libalg/alg.cpp:49-76 */
void oracle_sub_row(const double *A, size_t n, const double c[3], double *out)
{
    for (size_t i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j)
            out[i * 3 + j] = A[i * 3 + j] - c[j];
}

/* element_wise_reduce(a, b, 1,3,1,3, squared_norm_2, add, add): alg.cpp:12-32 with
 * squared_norm_2(a,b) = pow(a-b,2) (src/libalg/basic_operations.cpp:5-8; folded to (a-b)*(a-b) at -O3). */
static inline double sqdist3(const double *a, const double *b)
{
    double tmp = 0;
    double d;
    d = a[0] - b[0]; tmp = tmp + d * d;
    d = a[1] - b[1]; tmp = tmp + d * d;
    d = a[2] - b[2]; tmp = tmp + d * d;
    double r = 0;
    r = r + tmp;
    return r;
}

/* get_correspondence_indices(CPUMatrix&, CPUMatrix&): src/cpu/icp.cpp:72-93.
 * dist = std::sqrt(p.euclidianDistance(q)) where euclidianDistance already takes a sqrt
 * (CPUMatrix.cpp:291-295): the comparison key is the FOURTH root of d^2; first minimum wins (strict <). */
void oracle_correspondences(const double *Pz, size_t n, const double *Qz, size_t m, int32_t *idx, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (long long i = 0; i < (long long)n; ++i) {
        const double *p = Pz + (size_t)i * 3;
        double min_dist = DBL_MAX;
        int chosen = -1;
        for (size_t j = 0; j < m; ++j) {
            double dist = sqrt(sqrt(sqdist3(p, Qz + j * 3)));
            if (dist < min_dist) {
                min_dist = dist;
                chosen = (int)j;
            }
        }
        idx[i] = chosen;
    }
}

/* compute_cross_variance: src/cpu/icp.cpp:108-131.  Per point: doted = q^T . p via dot_product
 * (alg.cpp:78-103: tmp = 0; tmp += a*b), doted *= weight (weight == 1), cov += doted. */
void oracle_cross_cov(const double *Pz, size_t n, const double *Qz, const int32_t *idx, double cov[9])
{
    for (int k = 0; k < 9; ++k) cov[k] = 0;
    for (size_t i = 0; i < n; ++i) {
        const double *q = Qz + (size_t)idx[i] * 3;
        const double *p = Pz + i * 3;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double tmp = 0;
                tmp += q[a] * p[b];
                tmp = tmp * 1.0;
                cov[a * 3 + b] = cov[a * 3 + b] + tmp;
            }
    }
}

/* Stand-in for CPUMatrix::svd (CPUMatrix.cpp:297-321 -> svd.cpp:36-73 -> LAPACK dgesvd_, third party,
 * unpinned).  One-sided (Hestenes) Jacobi: A*V = U*diag(S), S sorted descending, A = U diag(S) VT.
 * Exactly-zero singular values get their U column completed so that det(U) == det(V)
 * (=> U*V^T is a proper rotation on the null space; reproduces LAPACK's R[2][2] = +1 on the planar
 * 30-point fixture).  Factor signs are otherwise arbitrary, as in LAPACK; only U*VT is pinned. */
void oracle_svd3(const double Ain[9], double U[9], double S[3], double VT[9])
{
    double a[3][3], v[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            a[i][j] = Ain[i * 3 + j];
            v[i][j] = (i == j) ? 1.0 : 0.0;
        }
    static const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int k = 0; k < 3; ++k) {
            int p = pr[k][0], q = pr[k][1];
            double alpha = 0, beta = 0, gamma = 0;
            for (int i = 0; i < 3; ++i) {
                alpha += a[i][p] * a[i][p];
                beta += a[i][q] * a[i][q];
                gamma += a[i][p] * a[i][q];
            }
            if (gamma == 0.0 || fabs(gamma) <= 4.5e-16 * sqrt(alpha * beta))
                continue;
            rotated = 1;
            double zeta = (beta - alpha) / (2.0 * gamma);
            double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            double c = 1.0 / sqrt(1.0 + tt * tt);
            double s = c * tt;
            for (int i = 0; i < 3; ++i) {
                double x = a[i][p], y = a[i][q];
                a[i][p] = c * x - s * y;
                a[i][q] = s * x + c * y;
                x = v[i][p]; y = v[i][q];
                v[i][p] = c * x - s * y;
                v[i][q] = s * x + c * y;
            }
        }
        if (!rotated) break;
    }
    double sg[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j)
        sg[j] = sqrt(a[0][j] * a[0][j] + a[1][j] * a[1][j] + a[2][j] * a[2][j]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (sg[ord[j]] < sg[ord[j + 1]]) { int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double u[3][3], vv[3][3];
    /* singular values at roundoff level (<= 1e-13 * S[0]) carry no direction: treated as zero */
    const double tol = 1e-13 * sg[ord[0]];
    int rank = 0;
    for (int j = 0; j < 3; ++j) {
        int c = ord[j];
        S[j] = sg[c];
        for (int i = 0; i < 3; ++i) vv[i][j] = v[i][c];
        if (sg[c] > 0.0 && sg[c] > tol && rank == j) {
            for (int i = 0; i < 3; ++i) u[i][j] = a[i][c] / sg[c];
            rank = j + 1;
        } else {
            for (int i = 0; i < 3; ++i) u[i][j] = 0.0;
        }
    }
    double detv = vv[0][0] * (vv[1][1] * vv[2][2] - vv[1][2] * vv[2][1])
                - vv[0][1] * (vv[1][0] * vv[2][2] - vv[1][2] * vv[2][0])
                + vv[0][2] * (vv[1][0] * vv[2][1] - vv[1][1] * vv[2][0]);
    double sgn = detv < 0 ? -1.0 : 1.0;
    if (rank == 0) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) u[i][j] = vv[i][j];
    } else {
        if (rank == 1) {
            /* any unit vector orthogonal to u0: cross with the axis of its smallest component */
            int m = 0;
            if (fabs(u[1][0]) < fabs(u[m][0])) m = 1;
            if (fabs(u[2][0]) < fabs(u[m][0])) m = 2;
            double e[3] = {0, 0, 0};
            e[m] = 1.0;
            u[0][1] = u[1][0] * e[2] - u[2][0] * e[1];
            u[1][1] = u[2][0] * e[0] - u[0][0] * e[2];
            u[2][1] = u[0][0] * e[1] - u[1][0] * e[0];
        }
        /* Gram-Schmidt clean-up of u1 against u0 (a no-op to rounding when S[1] is well above tol) */
        double d01 = u[0][0] * u[0][1] + u[1][0] * u[1][1] + u[2][0] * u[2][1];
        for (int i = 0; i < 3; ++i) u[i][1] -= d01 * u[i][0];
        double n1 = sqrt(u[0][1] * u[0][1] + u[1][1] * u[1][1] + u[2][1] * u[2][1]);
        for (int i = 0; i < 3; ++i) u[i][1] /= n1;
        /* third column: +-(u0 x u1); the sign follows the data when S[2] is significant, otherwise
         * det(U) = det(V) (proper rotation on the null space) */
        double w[3] = {u[1][0] * u[2][1] - u[2][0] * u[1][1],
                       u[2][0] * u[0][1] - u[0][0] * u[2][1],
                       u[0][0] * u[1][1] - u[1][0] * u[0][1]};
        double s3 = sgn;
        if (rank == 3)
            s3 = (w[0] * u[0][2] + w[1] * u[1][2] + w[2] * u[2][2]) < 0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) u[i][2] = s3 * w[i];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            U[i * 3 + j] = u[i][j];
            VT[i * 3 + j] = vv[j][i];
        }
}

/* R = U.dot(V_T) (icp.cpp:221) and t = Q_center - P_center.dot(R.transpose()) (icp.cpp:223);
 * dot_product accumulates tmp = 0; tmp += a[i][m]*b[m][j] in m order (alg.cpp:91-101). */
void oracle_rt(const double U[9], const double VT[9], const double Pc[3], const double Qc[3], double R[9], double t[3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double tmp = 0;
            for (int m = 0; m < 3; ++m) tmp += U[i * 3 + m] * VT[m * 3 + j];
            R[i * 3 + j] = tmp;
        }
    for (int j = 0; j < 3; ++j) {
        double tmp = 0;
        for (int m = 0; m < 3; ++m) tmp += Pc[m] * R[j * 3 + m];
        t[j] = Qc[j] - tmp;
    }
}

/* P_copy = P_copy.dot(R.transpose()) + t (icp.cpp:225) */
void oracle_apply(double *P, size_t n, const double R[9], const double t[3])
{
    for (size_t i = 0; i < n; ++i) {
        double x = P[i * 3 + 0], y = P[i * 3 + 1], z = P[i * 3 + 2];
        for (int j = 0; j < 3; ++j) {
            double tmp = 0;
            tmp += x * R[j * 3 + 0];
            tmp += y * R[j * 3 + 1];
            tmp += z * R[j * 3 + 2];
            P[i * 3 + j] = tmp + t[j];
        }
    }
}

/* CPUMatrix::euclidianDistance on two n x 3 clouds (CPUMatrix.cpp:291-295, alg.cpp:12-32):
 * per row tmp = sum_j (a-b)^2, r += tmp, sqrt(r).  Printed by both mains (main.cpp:36, main.cu:79). */
double oracle_euclidian_distance(const double *A, const double *B, size_t n)
{
    double r = 0;
    for (size_t i = 0; i < n; ++i) {
        double tmp = 0;
        for (int j = 0; j < 3; ++j) {
            double d = A[i * 3 + j] - B[i * 3 + j];
            tmp = tmp + d * d;
        }
        r = r + tmp;
    }
    return sqrt(r);
}

/* icp(): src/cpu/icp.cpp:194-229 (lines 213 and 215 produce unused values and are omitted). */
int oracle_icp(const double *P, size_t n, const double *Q, size_t m, unsigned iters,
               double *P_out, double *Qc_out, oracle_trace *tr, int nthreads)
{
    if (n == 0 || m == 0) return -1;
    double Qc[3], Pc[3], cov[9], U[9], S[3], VT[9], R[9], t[3];
    double *Qz = (double *)malloc(m * 3 * sizeof(double));
    double *Pz = (double *)malloc(n * 3 * sizeof(double));
    double *Pcur = (double *)malloc(n * 3 * sizeof(double));
    int32_t *idx = (int32_t *)malloc(n * sizeof(int32_t));
    if (!Qz || !Pz || !Pcur || !idx) { free(Qz); free(Pz); free(Pcur); free(idx); return -2; }
    oracle_mean0(Q, m, Qc);                 /* icp.cpp:197 */
    oracle_sub_row(Q, m, Qc, Qz);           /* icp.cpp:198 */
    if (Qc_out) memcpy(Qc_out, Qc, sizeof Qc);
    memcpy(Pcur, P, n * 3 * sizeof(double)); /* icp.cpp:203 */
    for (unsigned it = 0; it < iters; ++it) {
        if (tr && tr->P_in) memcpy(tr->P_in + (size_t)it * n * 3, Pcur, n * 3 * sizeof(double));
        oracle_mean0(Pcur, n, Pc);          /* icp.cpp:206 */
        oracle_sub_row(Pcur, n, Pc, Pz);    /* icp.cpp:208 */
        oracle_correspondences(Pz, n, Qz, m, idx, nthreads); /* icp.cpp:210 */
        oracle_cross_cov(Pz, n, Qz, idx, cov);               /* icp.cpp:214 */
        oracle_svd3(cov, U, S, VT);         /* icp.cpp:218 */
        oracle_rt(U, VT, Pc, Qc, R, t);     /* icp.cpp:221-223 */
        oracle_apply(Pcur, n, R, t);        /* icp.cpp:225 */
        if (tr) {
            if (tr->Pc) memcpy(tr->Pc + it * 3, Pc, sizeof Pc);
            if (tr->idx) memcpy(tr->idx + (size_t)it * n, idx, n * sizeof(int32_t));
            if (tr->cov) memcpy(tr->cov + it * 9, cov, sizeof cov);
            if (tr->S) memcpy(tr->S + it * 3, S, sizeof S);
            if (tr->R) memcpy(tr->R + it * 9, R, sizeof R);
            if (tr->t) memcpy(tr->t + it * 3, t, sizeof t);
        }
    }
    memcpy(P_out, Pcur, n * 3 * sizeof(double));
    free(Qz); free(Pz); free(Pcur); free(idx);
    return 0;
}
