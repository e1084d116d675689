/* oracle/icp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, tight loops, identical operation order) of the reference's CPU ICP path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (icp_amd/, include/icp_amd.h) never links, loads or calls it.
 *
 * Parity status: PINNED -- validated bit-for-bit (idx, Pc, cov) and to <=1e-12 (R, t, P_out) against the
 * unmodified reference sources compiled into oracle/_ref/ref_trace (see tests/golden/make_golden.py,
 * tests/test_oracle_golden.py).  The 3x3 SVD of the reference lives in LAPACK dgesvd_ (not vendored,
 * not version-pinned by the reference: CMakeLists.txt:26,34); it is restated as a one-sided Jacobi SVD
 * and pinned through R = U*V^T, the only quantity the hot path consumes.
 *
 * All clouds are row-major n x 3 doubles, exactly like the reference's CPUMatrix.
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-iteration trace, every pointer optional (NULL = not recorded). Sizes are per iteration. */
typedef struct {
    double  *P_in;  /* iters * n*3 : running cloud at the start of each iteration      */
    double  *Pc;    /* iters * 3                                                        */
    int32_t *idx;   /* iters * n                                                        */
    double  *cov;   /* iters * 9                                                        */
    double  *S;     /* iters * 3                                                        */
    double  *R;     /* iters * 9                                                        */
    double  *t;     /* iters * 3                                                        */
} oracle_trace;

void   oracle_mean0(const double *A, size_t n, double out[3]);
void   oracle_sub_row(const double *A, size_t n, const double c[3], double *out);
void   oracle_correspondences(const double *Pz, size_t n, const double *Qz, size_t m, int32_t *idx, int nthreads);
void   oracle_cross_cov(const double *Pz, size_t n, const double *Qz, const int32_t *idx, double cov[9]);
void   oracle_svd3(const double A[9], double U[9], double S[3], double VT[9]);
void   oracle_rt(const double U[9], const double VT[9], const double Pc[3], const double Qc[3], double R[9], double t[3]);
void   oracle_apply(double *P, size_t n, const double R[9], const double t[3]);
double oracle_euclidian_distance(const double *A, const double *B, size_t n);
/* Whole loop. P, Q are not modified. Qc_out (3, optional). Returns 0, or -1 on empty input. */
int    oracle_icp(const double *P, size_t n, const double *Q, size_t m, unsigned iters,
                  double *P_out, double *Qc_out, oracle_trace *trace, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
