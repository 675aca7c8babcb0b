/*
 * ccp_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's Gauss-Seidel hot path:
 *   slack-CSR storage + ingest, at(), gaussSeidel(), applyToVector(), the vector
 *   helpers, and the closed-form 5-point Poisson assembly of SolveChannel.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The shipped product (libccp_gs.so) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against the
 * compiled reference header (oracle/_ref, built by oracle/Makefile from
 * /root/reference/.../sparse-matrix.h where it lies) by tests/golden/gen_golden.py
 * and tests/test_oracle_vs_ref.py, and against the committed fixtures in
 * tests/golden/ (which include the reference's own 4x4 known-answer test,
 * labs/lab3/src/OpenCVHW1/main6.cc:238-249).
 *
 * All reference citations are relative to /root/reference/.
 */
#ifndef CCP_ORACLE_H
#define CCP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Slack-CSR storage: project/src/PhotoMontage/sparse-matrix.h:670-676.
 * row_begin has n_rows entries (NOT n_rows+1). */
typedef struct orc_matrix {
    double  *values;
    int32_t *col_offset;
    int32_t *row_begin;
    int32_t *row_num_nze;
    int32_t *row_space_left;
    int32_t  n_rows;
    int32_t  n_cols;
    int64_t  n_values;      /* allocated length of values / col_offset */
} orc_matrix;

void orc_matrix_free(orc_matrix *m);

/* sparse-matrix.h:537-620 (initializeFromEigenRowMajor). non_zeros may be NULL. */
int orc_from_eigen_row_major(orc_matrix *m,
                             const double *values, int32_t n_values,
                             const int32_t *row_offset, int32_t n_row_offset,
                             const int32_t *col_offset, int32_t n_col_offset,
                             const int32_t *non_zeros, int32_t n_non_zeros);

/* sparse-matrix.h:265-319 (initializeFromVector): row-sorted COO, explicit zeros become slack. */
int orc_from_vector(orc_matrix *m, const int32_t *rows, const int32_t *cols,
                    const double *vals, int64_t count);

/* sparse-matrix.h:162-173 + 627-645 (at / getNearestIndex). */
double orc_at(const orc_matrix *m, int32_t row, int32_t col);

/* sparse-matrix.h:350-380 (gaussSeidel).  x0 == NULL reproduces the reference start
 * vector (all 1.0, :352); a non-NULL x0 is the documented `init` extension
 * (mirrors conjugateGradient's 4th argument, :396,401-403).
 * Outputs: x (n_cols entries), iterations executed, last L1 step (eps). */
int orc_gauss_seidel(const orc_matrix *m, const double *b, const double *x0,
                     double epsilon, int max_iteration,
                     double *x, int *iters_done, double *last_eps);

/* sparse-matrix.h:396-434 (conjugateGradient with optional initial guess; the solver the blend
 * call sites use today, PhotoMontage.cpp:613 / hw8_pa.cc:972).  init == NULL starts from 0.
 * Outputs x (n_cols entries) and the number of completed iterations (`cnt`). */
int orc_conjugate_gradient(const orc_matrix *m, const double *b, const double *init,
                           double epsilon, int max_iteration, double *x, int *iters_done);

/* sparse-matrix.h:494-535 (conjugateGradientEigen: Jacobi-preconditioned, x0 = 0; used by RunTest,
 * utils.cc:99) with extractDiagnolColInv (:472-491). */
int orc_conjugate_gradient_jacobi(const orc_matrix *m, const double *b, double epsilon, int max_iteration,
                                  double *x, int *iters_done);

/* sparse-matrix.h:382-393 (applyToVector). */
void orc_apply_to_vector(const orc_matrix *m, const double *in, double *out);

/* sparse-matrix.h:45-105 vector helpers (serial left-to-right order, which is what the
 * serial PSTL backend executes in this container). */
double orc_manhatton_dist(const double *a, const double *b, int64_t n);
double orc_veclen2(const double *a, int64_t n);
double orc_dot_prod(const double *a, const double *b, int64_t n);
void   orc_vecsub(const double *a, const double *b, double *out, int64_t n);
void   orc_vecadd_scaled(const double *a, const double *b, double scale_b, double *out, int64_t n);

/* ||b - A x||_2 / ||b||_2 with applyToVector + vecsub + veclen2 (SURVEY §8d metric). */
double orc_rel_residual(const orc_matrix *m, const double *b, const double *x);

/* ---- Poisson assembly: project/src/PhotoMontage/PhotoMontage.cpp:541-597 ------------
 * Closed form of ATA = A^T A for the forward-difference + pin system.
 * Writes compressed CSR (row_offset has n+1 entries, Eigen outerIndexPtr style).
 * Returns nnz; call with NULL outputs to query the size. */
int64_t orc_poisson_csr(int32_t W, int32_t H, double *values, int32_t *col_offset,
                        int32_t *row_offset);
/* Band forms of the same closed form (rows [y0, y1) of the image), for threaded set-up of large systems. */
int64_t orc_poisson_csr_band(int32_t W, int32_t H, int32_t y0, int32_t y1, int64_t k0, double *values,
                             int32_t *col_offset, int32_t *row_offset);
int64_t orc_poisson_csr_band64(int32_t W, int32_t H, int32_t y0, int32_t y1, int64_t k0, double *values,
                               int64_t *col_offset, int64_t *row_offset);
void orc_poisson_row_starts(int32_t W, int32_t H, int64_t *starts);
void orc_poisson_apply_band(int32_t W, int32_t H, int32_t y0, int32_t y1, const double *v, double *out);


/* ATb for one channel.  gx/gy: H x W x channels float32, row stride in BYTES
 * (cv::Mat CV_32FC3 layout).  Summation order is Eigen's row-major sparse*dense
 * product: ascending row index of A (gy above, gx left, -gx here, -gy here, pin). */
void orc_poisson_rhs(int32_t W, int32_t H, const float *gx, const float *gy,
                     int64_t row_stride_bytes, int32_t channels, int32_t channel,
                     int32_t constraint, double *atb);

/* PhotoMontage.cpp:399-408,419-425 (GradientAt loop): images[k] is H x W x 3 u8
 * (stride bytes), label H x W u8.  Only y<H-1, x<W-1 are written. */
void orc_gradient_field(int32_t W, int32_t H, const uint8_t *const *images,
                        int64_t image_stride_bytes, const uint8_t *label,
                        int64_t label_stride_bytes, float *gx, float *gy,
                        int64_t grad_stride_bytes);

/* PhotoMontage.cpp:617-626: out(y,x)[c] = uchar(max(min(sol,255),0)). */
void orc_clamp_store_u8(int32_t W, int32_t H, const double *sol, uint8_t *out,
                        int64_t out_stride_bytes, int32_t channels, int32_t channel);

/* PhotoMontage.cpp:599-610: composite initial guess. */
void orc_composite_init(int32_t W, int32_t H, const uint8_t *const *images,
                        int64_t image_stride_bytes, const uint8_t *label,
                        int64_t label_stride_bytes, int32_t channel, double *init);

/* ---- symmetric permutation (colour-major ordering) ----------------------------------
 * perm[new] = old.  Produces compressed CSR of P A P^T with columns re-sorted per row
 * (the reference's at() requires sorted columns, sparse-matrix.h:627-645).  The
 * reference gaussSeidel on this matrix IS a multi-colour sweep when perm lists the
 * colours one after another. */
int orc_permute_csr(int32_t n, const double *values, const int32_t *col_offset,
                    const int32_t *row_offset /* n+1 */, const int32_t *perm,
                    double *p_values, int32_t *p_col_offset, int32_t *p_row_offset);

#ifdef __cplusplus
}
#endif
#endif
