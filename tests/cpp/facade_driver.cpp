// facade_driver.cpp — exercises include/ccp/sparse-matrix.h the way the reference's own test
// program does (labs/lab3/src/OpenCVHW1/main6.cc:192-253), plus file-driven solves that the
// pytest wrappers compare with the golden fixtures.
//
//   facade_driver host                 host-side checks only (insert cases vs a dense mirror)
//   facade_driver known                the 4x4 Gauss-Seidel known-answer flow on the GPU
//   facade_driver gs <in.bin> <out.bin> CSR solve: header {n, nnz, max_it, ordering, has_colour},
//                                      eps, values, cols, row_offset[n+1], b, [colour]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "ccp/sparse-matrix.h"
#include "ccp/photomontage.h"

using Dense = std::vector<std::vector<double>>;

template <typename M>
static bool same_as_mirror(const M &sp, const Dense &mat)
{
    for (size_t r = 0; r < mat.size(); ++r)
        for (size_t c = 0; c < mat[r].size(); ++c)
            if ((double)sp.at((int)r, (int)c) != mat[r][c]) {
                std::fprintf(stderr, "mismatch at (%zu,%zu): %g vs %g\n", r, c, (double)sp.at((int)r, (int)c), mat[r][c]);
                return false;
            }
    return true;
}

static int host_checks()
{
    // the five insert cases of main6.cc:193-231 (input data of the reference test)
    SparseMatrix<int> spi;
    Dense mat = {{1, 0, 0, 1, 0}, {0, 0, 0, 0, 0}, {8, 0, 1, 0, 0}};
    std::vector<int> vals = {1, 1, 0, 8, 1}, cols = {0, 3, 4, 0, 2}, rows = {0, 0, 0, 2, 2};
    spi.initializeFromVector(rows, std::move(cols), std::move(vals));
    if (spi.rows() != 3 || spi.cols() != 5 || !same_as_mirror(spi, mat)) return 1;
    struct Op { int v, r, c; };
    const Op ops[] = {{0, 1, 0}, {0, 0, 0}, {1, 2, 2}, {8, 0, 0}, {9, 1, 1}};
    for (const auto &o : ops) {
        spi.insert(o.v, o.r, o.c);
        mat[o.r][o.c] = o.v;
        if (!same_as_mirror(spi, mat)) return 2;
    }
    // pass-by-value copy (lab3 flavour) keeps the content
    SparseMatrix<int> copy = spi;
    if (!same_as_mirror(copy, mat)) return 3;
    // seeded random scenario against the mirror (the reference's insert diverges from its own
    // mirror criterion here; the facade must not)
    std::mt19937 gen(77);
    const int nr = 9, nc = 11;
    Dense m2(nr, std::vector<double>(nc, 0.0));
    std::vector<int> r2, c2;
    std::vector<double> v2;
    for (int r = 0; r < nr; ++r)
        for (int c = 0; c < nc; ++c) {
            const unsigned u = gen() % 100;
            if (u < 30 || (r == nr - 1 && c == nc - 1)) { m2[r][c] = 1 + gen() % 8; r2.push_back(r); c2.push_back(c); v2.push_back(m2[r][c]); }
            else if (u < 40) { r2.push_back(r); c2.push_back(c); v2.push_back(0.0); }
        }
    SparseMatrix<double> sd;
    sd.initializeFromVector(r2, std::move(c2), std::move(v2));
    if (!same_as_mirror(sd, m2)) return 4;
    for (int k = 0; k < 400; ++k) {
        const int r = gen() % nr, c = gen() % nc;
        const double v = (gen() % 3 == 0) ? 0.0 : double(1 + gen() % 9);
        sd.insert(v, r, c);
        m2[r][c] = v;
        if (!same_as_mirror(sd, m2)) { std::fprintf(stderr, "random scenario step %d\n", k); return 5; }
    }
    SparseMatrix<double> moved = std::move(sd);
    if (!same_as_mirror(moved, m2)) return 6;
    // vector helpers
    std::vector<double> v1 = {1.0, 2.0, 3.0, 10.0}, w1 = {2.0, 1.0, 3.0, 8.0};
    if (manhattonDist(v1, w1) != 4.0 || veclen2(v1) != 114.0 || dotProd(v1, w1) != 93.0) return 7;
    std::printf("host OK\n");
    return 0;
}

static int known_answer()
{
    SparseMatrix<int> sp2;
    sp2.initialize(4, 4, {10, -1, 2, 0, -1, 11, -1, 3, 2, -1, 10, -1, 0, 3, -1, 8});   // main6.cc:238-244
    std::vector<double> b = {6, 25, -11, 15};
    auto vec = sp2.gaussSeidel(b);                                                        // main6.cc:247
    std::printf("By Guass-Seidel %.17g %.17g %.17g %.17g iterations %d\n", vec[0], vec[1], vec[2], vec[3],
                sp2.lastReport().iterations);
    std::vector<double> out(4);
    sp2.applyToVector(vec, out);
    std::printf("A*x %.17g %.17g %.17g %.17g\n", out[0], out[1], out[2], out[3]);
    auto fast = sp2.gaussSeidel(b, 1e-9, 1000, {}, ccp::Ordering::MultiColour);
    std::printf("multicolour %.17g %.17g %.17g %.17g\n", fast[0], fast[1], fast[2], fast[3]);
    return 0;
}

template <typename T>
static bool rd(FILE *f, std::vector<T> &v, size_t n)
{
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

static int solve_file(const char *in, const char *out)
{
    FILE *f = std::fopen(in, "rb");
    if (!f) return 10;
    int hdr[5];
    double eps;
    if (std::fread(hdr, sizeof(int), 5, f) != 5 || std::fread(&eps, sizeof(double), 1, f) != 1) return 11;
    const int n = hdr[0], nnz = hdr[1], max_it = hdr[2], ordering = hdr[3], has_colour = hdr[4];
    std::vector<double> values, b;
    std::vector<int> cols, rowp, colour;
    if (!rd(f, values, nnz) || !rd(f, cols, nnz) || !rd(f, rowp, n + 1) || !rd(f, b, n)) return 12;
    if (has_colour && !rd(f, colour, n)) return 13;
    std::fclose(f);
    SparseMatrix<double> m;      // the ConvertFromEigen hand-off (utils.cc:5-15), compressed input
    m.initializeFromEigenRowMajor(values.data(), nnz, rowp.data(), n, cols.data(), n, nullptr, n);
    if (has_colour) m.setColouring(colour, *std::max_element(colour.begin(), colour.end()) + 1);
    auto x = m.gaussSeidel(b, eps, max_it, {}, ordering ? ccp::Ordering::MultiColour : ccp::Ordering::Lexicographic);
    std::vector<double> ax(n);
    m.applyToVector(x, ax);
    const double rel = m.relativeResidual(b, x);
    FILE *o = std::fopen(out, "wb");
    if (!o) return 14;
    const int it = m.lastReport().iterations;
    std::fwrite(&it, sizeof(int), 1, o);
    std::fwrite(&rel, sizeof(double), 1, o);
    std::fwrite(x.data(), sizeof(double), n, o);
    std::fwrite(ax.data(), sizeof(double), n, o);
    std::fclose(o);
    return 0;
}

// blend <in.bin> <out.bin>: header {W, H, K, iterations, fast_init}, K images HxWx3 u8, label HxW u8.
// Route 1: ccp::BuildSolveGradientFusion (everything on device).  Route 2: the reference's own
// structure — GradientAt loop on the host (PhotoMontage.cpp:419-425) then ccp::SolveChannel per
// channel (:429-433).  Both must give the same image; route 1 is written to out.bin.
static int blend_file(const char *in, const char *out)
{
    FILE *f = std::fopen(in, "rb");
    if (!f) return 20;
    int hdr[5];
    if (std::fread(hdr, sizeof(int), 5, f) != 5) return 21;
    const int W = hdr[0], H = hdr[1], K = hdr[2], iters = hdr[3], fast = hdr[4];
    std::vector<std::vector<uint8_t>> imgs(K, std::vector<uint8_t>((size_t)W * H * 3));
    for (auto &im : imgs)
        if (std::fread(im.data(), 1, im.size(), f) != im.size()) return 22;
    std::vector<uint8_t> label((size_t)W * H);
    if (std::fread(label.data(), 1, label.size(), f) != label.size()) return 23;
    std::fclose(f);
    std::vector<ccp::ImageView> views;
    for (auto &im : imgs) views.push_back(ccp::ImageView{im.data(), H, W, 3, (size_t)W * 3});
    ccp::ImageView lab{label.data(), H, W, 1, (size_t)W};
    std::vector<uint8_t> res1((size_t)W * H * 3, 0), res2((size_t)W * H * 3, 0);
    ccp::ImageView r1{res1.data(), H, W, 3, (size_t)W * 3}, r2{res2.data(), H, W, 3, (size_t)W * 3};
    ccp::BuildSolveGradientFusion(views, lab, r1, iters, fast != 0);
    // route 2
    std::vector<float> gx((size_t)W * H * 3, 0.f), gy((size_t)W * H * 3, 0.f);
    std::vector<uint8_t> composite((size_t)W * H * 3);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t *im = imgs[label[(size_t)y * W + x]].data();
            for (int c = 0; c < 3; ++c) {
                composite[((size_t)y * W + x) * 3 + c] = im[((size_t)y * W + x) * 3 + c];
                if (y < H - 1 && x < W - 1) {
                    gx[((size_t)y * W + x) * 3 + c] = (float)((int)im[((size_t)y * W + x + 1) * 3 + c] - (int)im[((size_t)y * W + x) * 3 + c]);
                    gy[((size_t)y * W + x) * 3 + c] = (float)((int)im[((size_t)(y + 1) * W + x) * 3 + c] - (int)im[((size_t)y * W + x) * 3 + c]);
                }
            }
        }
    ccp::ImageView vgx{gx.data(), H, W, 3, (size_t)W * 3 * sizeof(float)}, vgy{gy.data(), H, W, 3, (size_t)W * 3 * sizeof(float)};
    ccp::ImageView comp{composite.data(), H, W, 3, (size_t)W * 3};
    for (int c = 0; c < 3; ++c)
        ccp::SolveChannel(c, imgs[0][c], vgx, vgy, r2, iters, fast ? &comp : nullptr);
    if (res1 != res2) { std::fprintf(stderr, "SolveChannel route differs from BuildSolveGradientFusion\n"); return 24; }
    // route 3: the reference's member-function shape (PhotoMontage.h:24): SolveChannel(channel_idx, constraint, gx, gy,
    // output, Images) on an object that carries iterations_, fast_init_value and result_label_ (PhotoMontage.cpp:428-434)
    std::vector<uint8_t> res3((size_t)W * H * 3, 0);
    ccp::ImageView r3{res3.data(), H, W, 3, (size_t)W * 3};
    ccp::PhotoMontage pm;
    pm.iterations_ = iters;
    pm.fast_init_value = fast;
    pm.result_label_ = lab;
    for (int c = 0; c < 3; ++c) pm.SolveChannel(c, imgs[0][c], vgx, vgy, r3, views);
    if (res1 != res3) { std::fprintf(stderr, "PhotoMontage::SolveChannel differs from BuildSolveGradientFusion\n"); return 26; }
    FILE *o = std::fopen(out, "wb");
    if (!o) return 25;
    std::fwrite(res1.data(), 1, res1.size(), o);
    std::fclose(o);
    return 0;
}

// insert <in.bin> <out.bin>: header {nr, nc, count, n_ops, as_int}, rows, cols (int32), vals (f64),
// ops (n_ops x {val, row, col} f64).  Replays initializeFromVector + insert() (main6.cc:193-231) on
// SparseMatrix<int> or SparseMatrix<double> and writes the dense scan (at() of every cell, as
// CheckEqual reads it, main6.cc:19-33) after the ingest and after every op: (n_ops+1) x nr x nc f64.
template <typename T>
static int replay_inserts(int nr, int nc, const std::vector<int> &rows, const std::vector<int> &cols,
                          const std::vector<double> &vals, const std::vector<double> &ops, FILE *o)
{
    SparseMatrix<T> m;
    std::vector<int> c(cols);
    std::vector<T> v(vals.begin(), vals.end());
    m.initializeFromVector(rows, std::move(c), std::move(v));
    std::vector<double> dense((size_t)nr * nc);
    auto scan = [&]() {
        for (int r = 0; r < nr; ++r)
            for (int q = 0; q < nc; ++q) dense[(size_t)r * nc + q] = (double)m.at(r, q);
        std::fwrite(dense.data(), sizeof(double), dense.size(), o);
    };
    scan();
    for (size_t k = 0; k + 2 < ops.size(); k += 3) {
        m.insert((T)ops[k], (int)ops[k + 1], (int)ops[k + 2]);
        scan();
    }
    return 0;
}

static int insert_file(const char *in, const char *out)
{
    FILE *f = std::fopen(in, "rb");
    if (!f) return 30;
    int hdr[5];
    if (std::fread(hdr, sizeof(int), 5, f) != 5) return 31;
    const int nr = hdr[0], nc = hdr[1], count = hdr[2], n_ops = hdr[3], as_int = hdr[4];
    std::vector<int> rows, cols;
    std::vector<double> vals, ops;
    if (!rd(f, rows, count) || !rd(f, cols, count) || !rd(f, vals, count) || !rd(f, ops, (size_t)n_ops * 3)) return 32;
    std::fclose(f);
    FILE *o = std::fopen(out, "wb");
    if (!o) return 33;
    const int st = as_int ? replay_inserts<int>(nr, nc, rows, cols, vals, ops, o) : replay_inserts<double>(nr, nc, rows, cols, vals, ops, o);
    std::fclose(o);
    return st;
}

// edit <in.bin> <out.bin>: header {n, nnz, n_edits, iterations, ordering}, values, cols, row_offset[n+1], b,
// edits (n_edits x {val, row, col} f64).  Route 1: solve once (uploads the matrix), apply the edits with
// insert() — forwarded to the device copy — and solve again.  Route 2: apply the same edits to a fresh
// matrix that has never been on the device and solve.  Output: x of route 1, x of route 2, A*x of route 1,
// then the device edit statistics of route 1 as 5 f64.
static int edit_file(const char *in, const char *out)
{
    FILE *f = std::fopen(in, "rb");
    if (!f) return 50;
    int hdr[5];
    if (std::fread(hdr, sizeof(int), 5, f) != 5) return 51;
    const int n = hdr[0], nnz = hdr[1], n_edits = hdr[2], iters = hdr[3], ordering = hdr[4];
    std::vector<double> values, b, edits;
    std::vector<int> cols, rowp;
    if (!rd(f, values, nnz) || !rd(f, cols, nnz) || !rd(f, rowp, n + 1) || !rd(f, b, n) || !rd(f, edits, (size_t)n_edits * 3)) return 52;
    std::fclose(f);
    const auto ord = ordering ? ccp::Ordering::MultiColour : ccp::Ordering::Lexicographic;
    SparseMatrix<double> m1, m2;
    m1.initializeFromEigenRowMajor(values.data(), nnz, rowp.data(), n, cols.data(), n, nullptr, n);
    m2.initializeFromEigenRowMajor(values.data(), nnz, rowp.data(), n, cols.data(), n, nullptr, n);
    (void)m1.gaussSeidel(b, 0.0, 2, {}, ord);                       // the matrix goes to the device here
    std::vector<double> tmp(n);
    m1.applyToVector(b, tmp);
    for (int k = 0; k < n_edits; ++k) {
        m1.insert(edits[3 * k], (int)edits[3 * k + 1], (int)edits[3 * k + 2]);
        m2.insert(edits[3 * k], (int)edits[3 * k + 1], (int)edits[3 * k + 2]);
    }
    auto x1 = m1.gaussSeidel(b, 0.0, iters, {}, ord);
    std::vector<double> ax1(n);
    m1.applyToVector(x1, ax1);
    auto x2 = m2.gaussSeidel(b, 0.0, iters, {}, ord);
    const auto st = m1.deviceEditStats();
    const double stats[5] = {(double)st.edits, (double)st.image_uploads, (double)st.rows_patched, (double)st.slices_relocated,
                             (double)st.image_rebuilds};
    FILE *o = std::fopen(out, "wb");
    if (!o) return 53;
    std::fwrite(x1.data(), sizeof(double), n, o);
    std::fwrite(x2.data(), sizeof(double), n, o);
    std::fwrite(ax1.data(), sizeof(double), n, o);
    std::fwrite(stats, sizeof(double), 5, o);
    std::fclose(o);
    return 0;
}

// sizes: the facade refuses vectors shorter than the matrix before anything touches the device
static int size_checks()
{
    SparseMatrix<double> m;
    m.initialize(3, 3, {4, -1, 0, -1, 4, -1, 0, -1, 4});
    m.setDevice(0);
    m.setColouring({0, 1, 0}, 2);
    SparseMatrix<double> copy(m);                     // keeps device and colouring (checked via a solve elsewhere)
    int caught = 0;
    std::vector<double> b2 = {1, 2}, b3 = {1, 2, 3}, out2(2), out3(3);
    try { m.gaussSeidel(b2); } catch (const std::invalid_argument &) { ++caught; }
    try { m.gaussSeidel(b3, 1e-6, 10, b2); } catch (const std::invalid_argument &) { ++caught; }
    try { m.conjugateGradient(b2); } catch (const std::invalid_argument &) { ++caught; }
    try { m.conjugateGradientEigen(b2); } catch (const std::invalid_argument &) { ++caught; }
    try { m.applyToVector(b2, out3); } catch (const std::invalid_argument &) { ++caught; }
    try { m.applyToVector(b3, out2); } catch (const std::invalid_argument &) { ++caught; }
    try { m.relativeResidual(b2, b3); } catch (const std::invalid_argument &) { ++caught; }
    if (caught != 7) { std::fprintf(stderr, "only %d of 7 short vectors rejected\n", caught); return 40; }
    std::printf("sizes OK\n");
    return 0;
}

int main(int argc, char **argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "host";
        if (mode == "host") return host_checks();
        if (mode == "known") return known_answer();
        if (mode == "sizes") return size_checks();
        if (mode == "gs" && argc == 4) return solve_file(argv[2], argv[3]);
        if (mode == "blend" && argc == 4) return blend_file(argv[2], argv[3]);
        if (mode == "insert" && argc == 4) return insert_file(argv[2], argv[3]);
        if (mode == "edit" && argc == 4) return edit_file(argv[2], argv[3]);
        std::fprintf(stderr, "usage: facade_driver host|known|gs in out\n");
        return 64;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 70;
    }
}
