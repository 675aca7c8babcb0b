/*
 * ccp_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See ccp_oracle.h.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/).  Arithmetic order is kept identical to the reference so results are
 * bit-identical to the compiled reference header (verified: tests/test_oracle_vs_ref.py).
 */
#include "ccp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_matrix_free(orc_matrix *m)
{
    if (!m) return;
    free(m->values);
    free(m->col_offset);
    free(m->row_begin);
    free(m->row_num_nze);
    free(m->row_space_left);
    memset(m, 0, sizeof(*m));
}

static void *dup_bytes(const void *src, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (p && bytes) memcpy(p, src, bytes);
    return p;
}

/* project/src/PhotoMontage/sparse-matrix.h:537-620 */
int orc_from_eigen_row_major(orc_matrix *m,
                             const double *values, int32_t n_values,
                             const int32_t *row_offset, int32_t n_row_offset,
                             const int32_t *col_offset, int32_t n_col_offset,
                             const int32_t *non_zeros, int32_t n_non_zeros)
{
    memset(m, 0, sizeof(*m));
    const int32_t nr = n_row_offset;
    m->n_rows = nr;                       /* :549 */
    m->n_cols = n_col_offset;             /* :550 */
    m->n_values = n_values;
    m->values = (double *)dup_bytes(values, sizeof(double) * (size_t)n_values);        /* :552 */
    m->row_begin = (int32_t *)dup_bytes(row_offset, sizeof(int32_t) * (size_t)nr);     /* :553 */
    m->col_offset = (int32_t *)dup_bytes(col_offset, sizeof(int32_t) * (size_t)n_values); /* :554 */
    m->row_space_left = (int32_t *)calloc((size_t)(nr ? nr : 1), sizeof(int32_t));     /* :556 */
    if (!m->values || !m->row_begin || !m->col_offset || !m->row_space_left) return -1;

    if (non_zeros) {                      /* :560-589, rows carry slack ("holes") */
        m->row_num_nze = (int32_t *)dup_bytes(non_zeros, sizeof(int32_t) * (size_t)n_non_zeros);
        if (!m->row_num_nze) return -1;
        int32_t i = 0;
        while (i < nr && m->row_begin[i] != n_values) ++i;   /* first trailing-empty row */
        int32_t last = 0;
        if (i > 0) last = m->row_begin[i - 1] + m->row_num_nze[i - 1];
        for (; i < nr; ++i) m->row_begin[i] = last;
        for (int32_t r = 0; r + 1 < nr; ++r)
            m->row_space_left[r] = m->row_begin[r + 1] - m->row_begin[r] - m->row_num_nze[r];
        if (nr >= 2)                      /* :588 uses row n-2's start for the last row */
            m->row_space_left[nr - 1] = n_values - m->row_begin[nr - 2] - m->row_num_nze[nr - 1];
    } else {                              /* :592-619, compressed input */
        m->row_num_nze = (int32_t *)calloc((size_t)(nr ? nr : 1), sizeof(int32_t));
        if (!m->row_num_nze) return -1;
        int32_t i = 0;
        for (; i < nr - 1; ++i) {
            m->row_num_nze[i] = m->row_begin[i + 1] - m->row_begin[i];
            if (m->row_begin[i] == n_values) break;
        }
        if (m->row_begin[i] == n_values) {            /* :608-614 trailing all-zero rows */
            for (; i < nr; ++i) m->row_begin[i] -= 1;
        } else {
            m->row_num_nze[i] = n_values - m->row_begin[i];
        }
    }
    return 0;
}

/* project/src/PhotoMontage/sparse-matrix.h:265-319 (lab3: 209-255) */
int orc_from_vector(orc_matrix *m, const int32_t *rows, const int32_t *cols,
                    const double *vals, int64_t count)
{
    memset(m, 0, sizeof(*m));
    if (count <= 0) return -1;
    m->n_values = count;
    m->values = (double *)dup_bytes(vals, sizeof(double) * (size_t)count);
    m->col_offset = (int32_t *)dup_bytes(cols, sizeof(int32_t) * (size_t)count);
    if (!m->values || !m->col_offset) return -1;

    m->n_rows = rows[count - 1] + 1;                  /* :270 */
    int32_t maxc = 0;                                 /* :271-275 (starts from 0) */
    for (int64_t k = 0; k < count; ++k)
        if (cols[k] > maxc) maxc = cols[k];
    m->n_cols = maxc + 1;

    const int32_t nr = m->n_rows;
    m->row_begin = (int32_t *)calloc((size_t)nr, sizeof(int32_t));
    m->row_space_left = (int32_t *)calloc((size_t)nr, sizeof(int32_t));
    m->row_num_nze = (int32_t *)calloc((size_t)nr, sizeof(int32_t));
    if (!m->row_begin || !m->row_space_left || !m->row_num_nze) return -1;

    /* :283-306 compaction: non-zeros of a row slide to the row's front, zeros become slack */
    int64_t wr = 0;
    int32_t last_row = -1;
    for (int64_t rd = 0; rd < count; ++rd) {
        const int32_t r = rows[rd];
        if (r != last_row) { last_row = r; wr = rd; }
        if (m->values[rd] == 0) {
            m->row_space_left[r] += 1;
        } else {
            m->row_begin[r] += 1;                     /* used as the nnz counter first */
            m->values[wr] = m->values[rd];
            m->col_offset[wr] = m->col_offset[rd];
            ++wr;
        }
    }
    memcpy(m->row_num_nze, m->row_begin, sizeof(int32_t) * (size_t)nr);   /* :308 */
    int32_t run = 0;                                  /* :310-318 exclusive scan incl. slack */
    for (int32_t r = 0; r < nr; ++r) {
        const int32_t start = run;
        run += m->row_begin[r] + m->row_space_left[r];
        m->row_begin[r] = start;
    }
    return 0;
}

/* sparse-matrix.h:627-645 */
static int32_t nearest_index(const orc_matrix *m, int32_t row, int32_t col)
{
    int32_t lo = m->row_begin[row];
    int32_t hi = lo + m->row_num_nze[row] - 1;
    if (m->col_offset[lo] == col) return lo;
    while (hi > lo) {
        const int32_t mid = (hi + lo) / 2;
        if (m->col_offset[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

/* sparse-matrix.h:162-173 */
double orc_at(const orc_matrix *m, int32_t row, int32_t col)
{
    if (!m->row_num_nze[row]) return 0.0;
    const int32_t k = nearest_index(m, row, col);
    return m->col_offset[k] == col ? m->values[k] : 0.0;
}

/* sparse-matrix.h:45-49: serial left-to-right transform_reduce, init 0.0 */
double orc_manhatton_dist(const double *a, const double *b, int64_t n)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc = acc + fabs(a[i] - b[i]);
    return acc;
}

/* sparse-matrix.h:51-55 */
double orc_veclen2(const double *a, int64_t n)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc = acc + a[i] * a[i];
    return acc;
}

/* sparse-matrix.h:58-63 */
double orc_dot_prod(const double *a, const double *b, int64_t n)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc = acc + a[i] * b[i];
    return acc;
}

/* sparse-matrix.h:75-79 */
void orc_vecsub(const double *a, const double *b, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = a[i] - b[i];
}

/* sparse-matrix.h:81-85 */
void orc_vecadd_scaled(const double *a, const double *b, double scale_b, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = a[i] + scale_b * b[i];
}

/* sparse-matrix.h:350-380 */
int orc_gauss_seidel(const orc_matrix *m, const double *b, const double *x0,
                     double epsilon, int max_iteration,
                     double *x, int *iters_done, double *last_eps)
{
    const int64_t n = m->n_cols;          /* x is sized from b, which must match n_cols (:351-352) */
    double *prev = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    if (!prev) return -1;
    for (int64_t i = 0; i < n; ++i) x[i] = x0 ? x0[i] : 1.0;   /* :352 (1.0f widened) */
    double eps = 10;                      /* :354 */
    int cnt = 0;
    while (eps > epsilon && cnt < max_iteration) {
        memcpy(prev, x, sizeof(double) * (size_t)n);           /* :358 */
        for (int32_t i = 0; i < m->n_rows; ++i) {
            const double a_ii = orc_at(m, i, i);               /* :360 */
            if (a_ii == 0) continue;                           /* :361-363 */
            double sigma = 0;
            int32_t k = m->row_begin[i];
            for (int32_t j = 0; j < m->row_num_nze[i]; ++j, ++k) {
                const int32_t col = m->col_offset[k];
                if (col != i) sigma += m->values[k] * x[col];  /* :368-370 */
            }
            x[i] = (b[i] - sigma) / a_ii;                      /* :373 */
        }
        eps = orc_manhatton_dist(x, prev, n);                  /* :376 */
        ++cnt;
    }
    free(prev);
    if (iters_done) *iters_done = cnt;
    if (last_eps) *last_eps = eps;
    return 0;
}

/* sparse-matrix.h:382-393 */
void orc_apply_to_vector(const orc_matrix *m, const double *in, double *out)
{
    for (int32_t i = 0; i < m->n_rows; ++i) {
        int32_t k = m->row_begin[i];
        double sum = 0;
        for (int32_t j = 0; j < m->row_num_nze[i]; ++j, ++k)
            sum += m->values[k] * in[m->col_offset[k]];
        out[i] = sum;
    }
}

/* sparse-matrix.h:396-434 */
int orc_conjugate_gradient(const orc_matrix *m, const double *b, const double *init,
                           double epsilon, int max_iteration, double *x, int *iters_done)
{
    const int64_t n = m->n_cols;
    double *r = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    double *r1 = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    double *p = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    double *ap = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    if (!r || !r1 || !p || !ap) { free(r); free(r1); free(p); free(ap); return -1; }
    for (int64_t i = 0; i < n; ++i) x[i] = init ? init[i] : 0.0;        /* :397,401-403 */
    orc_apply_to_vector(m, x, r);                                       /* :406 r = A x   */
    orc_vecsub(b, r, r, n);                                             /* :407 r = b - r */
    memcpy(p, r, sizeof(double) * (size_t)n);                           /* :410 */
    int cnt = 0;
    while (cnt < max_iteration) {
        const double rlen = orc_veclen2(r, n);                          /* :418 */
        orc_apply_to_vector(m, p, ap);                                  /* :419 */
        const double alpha = rlen / orc_dot_prod(p, ap, n);             /* :420 */
        orc_vecadd_scaled(x, p, alpha, x, n);                           /* :421 */
        orc_vecadd_scaled(r, ap, -alpha, r1, n);                        /* :422 */
        const double r1len = orc_veclen2(r1, n);                        /* :423 */
        if (sqrt(r1len) < epsilon) break;                               /* :425 */
        const double beta = r1len / rlen;                               /* :426 */
        orc_vecadd_scaled(r1, p, beta, p, n);                           /* :427 */
        double *t = r1; r1 = r; r = t;                                  /* :429 swap */
        ++cnt;
    }
    free(r); free(r1); free(p); free(ap);
    if (iters_done) *iters_done = cnt;
    return 0;
}

/* sparse-matrix.h:494-535 (conjugateGradientEigen) with extractDiagnolColInv (:472-491) */
int orc_conjugate_gradient_jacobi(const orc_matrix *m, const double *b, double epsilon, int max_iteration,
                                  double *x, int *iters_done)
{
    const int64_t n = m->n_cols;
    const size_t bytes = sizeof(double) * (size_t)(n ? n : 1);
    double *r = (double *)malloc(bytes), *z = (double *)malloc(bytes), *p = (double *)malloc(bytes);
    double *ap = (double *)malloc(bytes), *inv = (double *)malloc(bytes);
    if (!r || !z || !p || !ap || !inv) { free(r); free(z); free(p); free(ap); free(inv); return -1; }
    for (int64_t i = 0; i < n; ++i) inv[i] = 1.0;                       /* :474 res(cols(), T(1)) */
    for (int32_t i = 0; i < m->n_rows; ++i) {                           /* :476-489 */
        int32_t idx = m->row_begin[i];
        for (int32_t j = 0; j < m->row_num_nze[i]; ++j, ++idx)
            if (m->col_offset[idx] == i) {
                if (m->values[idx] != 0.0) inv[i] = 1.0 / m->values[idx];
                break;
            }
    }
    for (int64_t i = 0; i < n; ++i) x[i] = 0.0;                         /* :495 */
    orc_apply_to_vector(m, x, r);                                       /* :500 */
    orc_vecsub(b, r, r, n);                                             /* :501 */
    for (int64_t i = 0; i < n; ++i) p[i] = r[i] * inv[i];               /* :503 vecmul */
    double olddist = orc_dot_prod(p, r, n);                             /* :508 */
    int cnt = 0;
    while (cnt < max_iteration) {
        orc_apply_to_vector(m, p, ap);                                  /* :516 */
        const double alpha = olddist / orc_dot_prod(p, ap, n);          /* :517 */
        orc_vecadd_scaled(x, p, alpha, x, n);                           /* :518 */
        orc_vecadd_scaled(r, ap, -alpha, r, n);                         /* :519 */
        const double error = orc_veclen2(r, n);                         /* :520 */
        if (sqrt(error) < epsilon) break;                               /* :521 */
        for (int64_t i = 0; i < n; ++i) z[i] = r[i] * inv[i];           /* :522 */
        const double newdist = orc_dot_prod(z, r, n);                   /* :523 */
        const double beta = newdist / olddist;                          /* :524 */
        olddist = newdist;                                              /* :525 */
        orc_vecadd_scaled(z, p, beta, p, n);                            /* :526 */
        ++cnt;                                                          /* :528 */
    }
    free(r); free(z); free(p); free(ap); free(inv);
    if (iters_done) *iters_done = cnt;
    return 0;
}

double orc_rel_residual(const orc_matrix *m, const double *b, const double *x)
{
    const int64_t n = m->n_rows;
    double *r = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    if (!r) return NAN;
    orc_apply_to_vector(m, x, r);
    orc_vecsub(b, r, r, n);
    const double rr = orc_veclen2(r, n);
    const double bb = orc_veclen2(b, n);
    free(r);
    return sqrt(rr) / sqrt(bb);
}

/* ---- Poisson assembly --------------------------------------------------------------- */

/* cell(x,y): a forward-difference pair of rows exists in A for this pixel
 * (PhotoMontage.cpp:551-554 loop bounds). */
static int has_cell(int32_t x, int32_t y, int32_t W, int32_t H)
{
    return x >= 0 && y >= 0 && x < W - 1 && y < H - 1;
}

/* PhotoMontage.cpp:541-592, closed form of A^T A. */
int64_t orc_poisson_csr(int32_t W, int32_t H, double *values, int32_t *col_offset,
                        int32_t *row_offset)
{
    int64_t k = 0;
    for (int32_t y = 0; y < H; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            const int32_t i = y * W + x;
            const int up = has_cell(x, y - 1, W, H);
            const int left = has_cell(x - 1, y, W, H);
            const int here = has_cell(x, y, W, H);
            const int diag = up + left + 2 * here + (i == 0 ? 1 : 0);
            if (row_offset) row_offset[i] = (int32_t)k;
            if (up)   { if (values) { values[k] = -1.0; col_offset[k] = i - W; } ++k; }
            if (left) { if (values) { values[k] = -1.0; col_offset[k] = i - 1; } ++k; }
            if (diag) { if (values) { values[k] = (double)diag; col_offset[k] = i; } ++k; }
            if (here) {
                if (values) { values[k] = -1.0; col_offset[k] = i + 1; } ++k;
                if (values) { values[k] = -1.0; col_offset[k] = i + W; } ++k;
            }
        }
    }
    if (row_offset) row_offset[(int64_t)W * H] = (int32_t)k;
    return k;
}

/* The same closed form for the image rows [y0, y1) only, entries written from position k0 on (= the number of
 * entries of the rows above, orc_poisson_row_starts): lets bench.py's cpu_baseline leg build the 1.34e9-entry
 * system of the 16384^2 workload on several host threads.  Same bytes as orc_poisson_csr (tests). */
int64_t orc_poisson_csr_band(int32_t W, int32_t H, int32_t y0, int32_t y1, int64_t k0, double *values,
                             int32_t *col_offset, int32_t *row_offset)
{
    int64_t k = k0;
    for (int32_t y = y0; y < y1; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            const int32_t i = y * W + x;
            const int up = has_cell(x, y - 1, W, H);
            const int left = has_cell(x - 1, y, W, H);
            const int here = has_cell(x, y, W, H);
            const int diag = up + left + 2 * here + (i == 0 ? 1 : 0);
            row_offset[i] = (int32_t)k;
            if (up)   { values[k] = -1.0; col_offset[k] = i - W; ++k; }
            if (left) { values[k] = -1.0; col_offset[k] = i - 1; ++k; }
            if (diag) { values[k] = (double)diag; col_offset[k] = i; ++k; }
            if (here) {
                values[k] = -1.0; col_offset[k] = i + 1; ++k;
                values[k] = -1.0; col_offset[k] = i + W; ++k;
            }
        }
    }
    if (y1 == H) row_offset[(int64_t)W * H] = (int32_t)k;
    return k;
}

/* The same with 64-bit indices: the arrays SparseMatrix<double, int64_t> ingests.  The reference's default IndexType
 * (int) cannot hold the 16384^2 system: getNearestIndex computes (end + idx) / 2 in Index arithmetic
 * (sparse-matrix.h:636), which overflows once a row's entries lie beyond position 2^30 — the last fifth of that
 * matrix's 1.34e9 entries — so bench.py's cpu_baseline leg instantiates the header with a 64-bit IndexType there. */
int64_t orc_poisson_csr_band64(int32_t W, int32_t H, int32_t y0, int32_t y1, int64_t k0, double *values,
                               int64_t *col_offset, int64_t *row_offset)
{
    int64_t k = k0;
    for (int32_t y = y0; y < y1; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            const int64_t i = (int64_t)y * W + x;
            const int up = has_cell(x, y - 1, W, H);
            const int left = has_cell(x - 1, y, W, H);
            const int here = has_cell(x, y, W, H);
            const int diag = up + left + 2 * here + (i == 0 ? 1 : 0);
            row_offset[i] = k;
            if (up)   { values[k] = -1.0; col_offset[k] = i - W; ++k; }
            if (left) { values[k] = -1.0; col_offset[k] = i - 1; ++k; }
            if (diag) { values[k] = (double)diag; col_offset[k] = i; ++k; }
            if (here) {
                values[k] = -1.0; col_offset[k] = i + 1; ++k;
                values[k] = -1.0; col_offset[k] = i + W; ++k;
            }
        }
    }
    if (y1 == H) row_offset[(int64_t)W * H] = k;
    return k;
}

/* starts[y] = entries of the image rows above y, y = 0..H (starts[H] = nnz). */
void orc_poisson_row_starts(int32_t W, int32_t H, int64_t *starts)
{
    int64_t k = 0;
    for (int32_t y = 0; y < H; ++y) {
        starts[y] = k;
        for (int32_t x = 0; x < W; ++x) {
            const int up = has_cell(x, y - 1, W, H), left = has_cell(x - 1, y, W, H), here = has_cell(x, y, W, H);
            k += up + left + 2 * here + ((up + left + 2 * here + (y == 0 && x == 0)) != 0);
        }
    }
    starts[H] = k;
}

/* out = A v for the rows [y0, y1) of that matrix, in applyToVector's accumulation order (sparse-matrix.h:382-393:
 * storage order = up, left, diagonal, right, down, from 0.0). */
void orc_poisson_apply_band(int32_t W, int32_t H, int32_t y0, int32_t y1, const double *v, double *out)
{
    for (int32_t y = y0; y < y1; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            const int64_t i = (int64_t)y * W + x;
            const int up = has_cell(x, y - 1, W, H);
            const int left = has_cell(x - 1, y, W, H);
            const int here = has_cell(x, y, W, H);
            const int diag = up + left + 2 * here + (i == 0 ? 1 : 0);
            double acc = 0.0;
            if (up)   acc += -1.0 * v[i - W];
            if (left) acc += -1.0 * v[i - 1];
            if (diag) acc += (double)diag * v[i];
            if (here) {
                acc += -1.0 * v[i + 1];
                acc += -1.0 * v[i + W];
            }
            out[i] = acc;
        }
    }
}

static const float *grad_px(const float *base, int64_t stride_bytes, int32_t channels,
                            int32_t x, int32_t y)
{
    return (const float *)((const char *)base + (int64_t)y * stride_bytes) + (int64_t)x * channels;
}

/* PhotoMontage.cpp:563-572 (b entries), :579-581 (pin), :592 (ATb = A^T b). */
void orc_poisson_rhs(int32_t W, int32_t H, const float *gx, const float *gy,
                     int64_t row_stride_bytes, int32_t channels, int32_t channel,
                     int32_t constraint, double *atb)
{
    for (int32_t y = 0; y < H; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            const int64_t i = (int64_t)y * W + x;
            double acc = 0.0;
            if (has_cell(x, y - 1, W, H))   /* row 2(i-W)+1: +v(x,y) in the gy equation above */
                acc += 1.0 * (double)grad_px(gy, row_stride_bytes, channels, x, y - 1)[channel];
            if (has_cell(x - 1, y, W, H))   /* row 2(i-1): +v(x,y) in the gx equation to the left */
                acc += 1.0 * (double)grad_px(gx, row_stride_bytes, channels, x - 1, y)[channel];
            if (has_cell(x, y, W, H)) {     /* rows 2i, 2i+1: -v(x,y) */
                acc += -1.0 * (double)grad_px(gx, row_stride_bytes, channels, x, y)[channel];
                acc += -1.0 * (double)grad_px(gy, row_stride_bytes, channels, x, y)[channel];
            }
            if (i == 0) acc += 1.0 * (double)constraint;   /* row 2WH */
            atb[i] = acc;
        }
    }
}

/* PhotoMontage.cpp:399-408, 419-425 */
void orc_gradient_field(int32_t W, int32_t H, const uint8_t *const *images,
                        int64_t image_stride_bytes, const uint8_t *label,
                        int64_t label_stride_bytes, float *gx, float *gy,
                        int64_t grad_stride_bytes)
{
    for (int32_t y = 0; y < H - 1; ++y) {
        for (int32_t x = 0; x < W - 1; ++x) {
            const uint8_t *img = images[label[(int64_t)y * label_stride_bytes + x]];
            const uint8_t *p0 = img + (int64_t)y * image_stride_bytes + 3 * (int64_t)x;
            const uint8_t *px = p0 + 3;
            const uint8_t *py = p0 + image_stride_bytes;
            float *ox = (float *)((char *)gx + (int64_t)y * grad_stride_bytes) + 3 * (int64_t)x;
            float *oy = (float *)((char *)gy + (int64_t)y * grad_stride_bytes) + 3 * (int64_t)x;
            for (int c = 0; c < 3; ++c) {
                ox[c] = (float)((int)px[c] - (int)p0[c]);
                oy[c] = (float)((int)py[c] - (int)p0[c]);
            }
        }
    }
}

/* PhotoMontage.cpp:617-626 */
void orc_clamp_store_u8(int32_t W, int32_t H, const double *sol, uint8_t *out,
                        int64_t out_stride_bytes, int32_t channels, int32_t channel)
{
    for (int32_t y = 0; y < H; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            double v = sol[(int64_t)y * W + x];
            v = v < 255.0 ? v : 255.0;     /* std::min(sol, 255.0) */
            v = v > 0.0 ? v : 0.0;         /* std::max(.., 0.0)    */
            out[(int64_t)y * out_stride_bytes + (int64_t)x * channels + channel] = (uint8_t)v;
        }
    }
}

/* PhotoMontage.cpp:599-610 */
void orc_composite_init(int32_t W, int32_t H, const uint8_t *const *images,
                        int64_t image_stride_bytes, const uint8_t *label,
                        int64_t label_stride_bytes, int32_t channel, double *init)
{
    for (int32_t y = 0; y < H; ++y)
        for (int32_t x = 0; x < W; ++x) {
            const uint8_t *img = images[label[(int64_t)y * label_stride_bytes + x]];
            init[(int64_t)y * W + x] = (double)img[(int64_t)y * image_stride_bytes + 3 * (int64_t)x + channel];
        }
}

/* ---- symmetric permutation ---------------------------------------------------------- */
int orc_permute_csr(int32_t n, const double *values, const int32_t *col_offset,
                    const int32_t *row_offset, const int32_t *perm,
                    double *p_values, int32_t *p_col_offset, int32_t *p_row_offset)
{
    int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    if (!inv) return -1;
    for (int32_t r = 0; r < n; ++r) inv[perm[r]] = r;
    int32_t k = 0;
    for (int32_t r = 0; r < n; ++r) {
        const int32_t old = perm[r];
        const int32_t start = k;
        p_row_offset[r] = k;
        for (int32_t j = row_offset[old]; j < row_offset[old + 1]; ++j, ++k) {
            /* insertion sort by new column index (rows are short) */
            const int32_t c = inv[col_offset[j]];
            const double v = values[j];
            int32_t pos = k;
            while (pos > start && p_col_offset[pos - 1] > c) {
                p_col_offset[pos] = p_col_offset[pos - 1];
                p_values[pos] = p_values[pos - 1];
                --pos;
            }
            p_col_offset[pos] = c;
            p_values[pos] = v;
        }
    }
    p_row_offset[n] = k;
    free(inv);
    return 0;
}
