/* oracle_asan.c — exercises oracle/ccp_oracle.c under AddressSanitizer + UBSan (CPU build only;
 * GPU sanitizers are not available on this pool).  Built and run by tests/test_sanitizers.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "ccp_oracle.h"

int main(void)
{
    const int W = 17, H = 13, n = W * H;
    const long nnz = orc_poisson_csr(W, H, NULL, NULL, NULL);
    double *v = malloc(sizeof(double) * nnz), *pv = malloc(sizeof(double) * nnz);
    int32_t *c = malloc(sizeof(int32_t) * nnz), *pc = malloc(sizeof(int32_t) * nnz);
    int32_t *r = malloc(sizeof(int32_t) * (n + 1)), *pr = malloc(sizeof(int32_t) * (n + 1));
    int32_t *perm = malloc(sizeof(int32_t) * n);
    double *xt = malloc(sizeof(double) * n), *b = malloc(sizeof(double) * n), *x = malloc(sizeof(double) * n);
    if (orc_poisson_csr(W, H, v, c, r) != nnz || nnz != (n - 1) + 4 * (W - 1) * (H - 1)) return 1;
    orc_matrix m;
    if (orc_from_eigen_row_major(&m, v, (int32_t)nnz, r, n, c, n, NULL, 0)) return 2;
    for (int i = 0; i < n; ++i) xt[i] = (i * 37 % 255) + 0.25;
    orc_apply_to_vector(&m, xt, b);
    int it = 0; double eps = 0;
    if (orc_gauss_seidel(&m, b, NULL, 0.0, 50, x, &it, &eps) || it != 50) return 3;
    if (x[n - 1] != 1.0 || orc_at(&m, 0, 0) != 3.0) return 4;
    if (orc_conjugate_gradient(&m, b, NULL, 1e-9, 2000, x, &it)) return 5;
    for (int i = 0; i < n - 1; ++i) if (fabs(x[i] - xt[i]) > 1e-6) return 6;
    if (!(orc_rel_residual(&m, b, x) < 1e-9)) return 7;
    /* red-first permutation */
    int k = 0;
    for (int col = 0; col < 2; ++col)
        for (int i = 0; i < n; ++i) if ((((i % W) + (i / W)) & 1) == col) perm[k++] = i;
    if (orc_permute_csr(n, v, c, r, perm, pv, pc, pr) || pr[n] != nnz) return 8;
    orc_matrix mp;
    if (orc_from_eigen_row_major(&mp, pv, (int32_t)nnz, pr, n, pc, n, NULL, 0)) return 9;
    for (int i = 0; i < n; ++i) x[i] = b[perm[i]];
    double *xp = malloc(sizeof(double) * n);
    if (orc_gauss_seidel(&mp, x, NULL, 0.0, 10, xp, &it, &eps)) return 10;
    /* COO ingest with an explicit zero and the 4x4 known answer */
    const int32_t rows[] = {0,0,0,0, 1,1,1,1, 2,2,2,2, 3,3,3,3}, cols[] = {0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3};
    const double vals[] = {10,-1,2,0, -1,11,-1,3, 2,-1,10,-1, 0,3,-1,8}, bb[] = {6, 25, -11, 15};
    orc_matrix k4; double x4[4];
    if (orc_from_vector(&k4, rows, cols, vals, 16)) return 11;
    if (orc_gauss_seidel(&k4, bb, NULL, 1e-6, 1000, x4, &it, &eps) || it != 8) return 12;
    if (fabs(x4[0] - 1) > 1e-6 || fabs(x4[1] - 2) > 1e-6 || fabs(x4[2] + 1) > 1e-6 || fabs(x4[3] - 1) > 1e-6) return 13;
    orc_matrix_free(&m); orc_matrix_free(&mp); orc_matrix_free(&k4);
    free(v); free(pv); free(c); free(pc); free(r); free(pr); free(perm); free(xt); free(b); free(x); free(xp);
    puts("oracle asan OK");
    return 0;
}
