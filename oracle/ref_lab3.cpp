// ref_lab3.cpp — thin extern "C" driver around the UNMODIFIED lab3 reference header
//   /root/reference/labs/lab3/src/OpenCVHW1/sparse-matrix.h  (SparseMatrix<T>, T=int/double)
// compiled where it lies (see oracle/Makefile; output goes to oracle/_ref/ only).
// Test infrastructure only.  Built as its own shared object because the lab3 and project
// headers define same-named inline helpers.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sparse-matrix.h"   // -I/root/reference/labs/lab3/src/OpenCVHW1

extern "C" {

// The reference's only gaussSeidel test (labs/lab3/src/OpenCVHW1/main6.cc:238-249):
// SparseMatrix<int> 4x4 from an initializer list, b = (6, 25, -11, 15).
// The matrix/RHS literals below are that test's input data.
int ref_lab3_known_answer(double epsilon, int max_iteration, double *x_gs, double *x_cg)
{
    SparseMatrix<int> sp2;
    sp2.initialize(4, 4, {
        10, -1,  2,  0,
        -1, 11, -1,  3,
         2, -1, 10, -1,
         0,  3, -1,  8 });
    std::vector<double> b = { 6, 25, -11, 15 };
    std::vector<double> x = sp2.gaussSeidel(b, epsilon, max_iteration);
    std::memcpy(x_gs, x.data(), sizeof(double) * 4);
    if (x_cg) {
        x = sp2.conjugateGradient(b);
        std::memcpy(x_cg, x.data(), sizeof(double) * 4);
    }
    return 0;
}

// Generic T=int path: initializeFromVector + inserts + dense scan (main6.cc:193-231 shape).
int ref_lab3_int_insert_scenario(const int *rows, const int *cols, const int *vals, int count,
                                 const int *op_row, const int *op_col, const int *op_val,
                                 int n_ops, int n_rows, int n_cols, int *dense_steps)
{
    SparseMatrix<int> m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count), v(vals, vals + count);
    m.initializeFromVector(r, std::move(c), std::move(v));
    auto snap = [&](int step) {
        for (int i = 0; i < n_rows; ++i)
            for (int j = 0; j < n_cols; ++j)
                dense_steps[((size_t)step * n_rows + i) * n_cols + j] = m.at(i, j);
    };
    snap(0);
    for (int k = 0; k < n_ops; ++k) {
        m.insert(op_val[k], op_row[k], op_col[k]);
        snap(k + 1);
    }
    return 0;
}

int ref_lab3_gs_vector_int(const int *rows, const int *cols, const int *vals, int count,
                           const double *b, int n, double epsilon, int max_iteration,
                           double *x_out)
{
    SparseMatrix<int> m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count), v(vals, vals + count);
    m.initializeFromVector(r, std::move(c), std::move(v));
    std::vector<double> bv(b, b + n);
    std::vector<double> x = m.gaussSeidel(bv, epsilon, max_iteration);
    std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return (int)x.size();
}

int ref_lab3_gs_vector_double(const int *rows, const int *cols, const double *vals, int count,
                              const double *b, int n, double epsilon, int max_iteration,
                              double *x_out)
{
    SparseMatrix<double> m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count);
    std::vector<double> v(vals, vals + count);
    m.initializeFromVector(r, std::move(c), std::move(v));
    std::vector<double> bv(b, b + n);
    std::vector<double> x = m.gaussSeidel(bv, epsilon, max_iteration);
    std::memcpy(x_out, x.data(), sizeof(double) * x.size());
    return (int)x.size();
}

int ref_lab3_spmv_vector_double(const int *rows, const int *cols, const double *vals, int count,
                                const double *in, int n_rows, double *out)
{
    SparseMatrix<double> m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count);
    std::vector<double> v(vals, vals + count);
    m.initializeFromVector(r, std::move(c), std::move(v));
    std::vector<double> iv(in, in + m.cols()), ov(n_rows, 0.0);
    m.applyToVector(iv, ov);
    std::memcpy(out, ov.data(), sizeof(double) * ov.size());
    return 0;
}

// The reference's modify benchmark (labs/lab3/src/OpenCVHW1/main6.cc:92-187) without its Eigen half:
// the scenario arrives from the caller (the same arrays the device-side benchmark uses), the lab3
// SparseMatrix<int> ingests it (initializeFromVector), every edit is insert(0, row, col), and the dense
// scan its own CheckEqual performs (main6.cc:19-33) is summed into `checksum`.  Returns milliseconds.
int ref_lab3_modify_bench(const int *rows, const int *cols, const int *vals, int count,
                          const int *op_row, const int *op_col, int n_ops, int size,
                          double *ms_init, double *ms_modify, long long *checksum)
{
    using clk = std::chrono::steady_clock;
    SparseMatrix<int> m;
    std::vector<int> r(rows, rows + count), c(cols, cols + count), v(vals, vals + count);
    auto t0 = clk::now();
    m.initializeFromVector(r, std::move(c), std::move(v));
    *ms_init = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    t0 = clk::now();
    for (int k = 0; k < n_ops; ++k) m.insert(0, op_row[k], op_col[k]);
    *ms_modify = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    long long sum = 0;
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) sum += m.at(i, j);
    *checksum = sum;
    return 0;
}

}  // extern "C"
