// ccp/photomontage.h — the image side of the path: drop-in counterparts of
// PhotoMontage::SolveChannel (project/src/PhotoMontage/PhotoMontage.cpp:535-628, same maths in
// labs/lab8/src/OpenCVHW1/hw8_pa.cc:902-986) and of the three-channel driver
// BuildSolveGradientFusion (PhotoMontage.cpp:410-436), on top of the C ABI (ccp_gs.h).
//
// OpenCV is not required: ccp::ImageView is a POD with cv::Mat's continuous row-major interleaved
// layout {data, rows, cols, channels, step}; `ccp::view(mat)` style adapters are one line
// (`ImageView{m.data, m.rows, m.cols, m.channels(), m.step}`).
//
// The solver is Gauss-Seidel (red-black, fixed iteration count — the slot where the reference
// calls conjugateGradient, PhotoMontage.cpp:613) or conjugate gradient; the matrix is never built.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ccp_gs.h"

namespace ccp {

struct ImageView {
    void *data;
    int rows, cols, channels;
    std::size_t step;          // bytes per image row
};

// GaussSeidel: red-black order (the fast path).  GaussSeidelReferenceOrder: the reference's index-order
// sweep, iterates bit-identical to SparseMatrix::gaussSeidel on the matrix SolveChannel builds.
enum class Solver { GaussSeidel, ConjugateGradient, GaussSeidelReferenceOrder };

namespace detail {
inline void check(int status, const char *what)
{
    if (status != CCP_OK) throw std::runtime_error(std::string(what) + ": " + ccp_status_string(status));
}
struct GridHandle {
    ccp_grid *g = nullptr;
    GridHandle(int W, int H, int C, int device)
    {
        ccp_grid_desc d{W, H, C, 0, H, 0, device, 0};
        check(ccp_grid_create(&d, &g), "ccp_grid_create");
    }
    ~GridHandle() { ccp_grid_destroy(g); }
    GridHandle(const GridHandle &) = delete;
    GridHandle &operator=(const GridHandle &) = delete;
};
inline void solve(ccp_grid *g, Solver solver, int iterations, int channels)
{
    std::vector<ccp_gs_report> rep(channels);
    if (solver == Solver::GaussSeidel)
        check(ccp_grid_gauss_seidel(g, 1e-10, iterations, /*check_every=*/0, rep.data()), "ccp_grid_gauss_seidel");
    else if (solver == Solver::GaussSeidelReferenceOrder)
        check(ccp_grid_gauss_seidel_lexicographic(g, 1e-10, iterations, /*check_every=*/0, rep.data()),
              "ccp_grid_gauss_seidel_lexicographic");
    else
        check(ccp_grid_conjugate_gradient(g, 1e-10, iterations, rep.data()), "ccp_grid_conjugate_gradient");
}
}  // namespace detail

// SolveChannel(channel_idx, constraint, color_gradient_x, color_gradient_y, output, ...)
// (PhotoMontage.h:24).  gx, gy: CV_32FC3 views (only y<H-1, x<W-1 are read), output: CV_8UC3 view
// whose channel `channel_idx` is written (clamped, PhotoMontage.cpp:617-626).  `init` (optional):
// CV_8UC3 composite image used as the start vector (fast_init_value, :599-610); without it Gauss-
// Seidel starts from 1.0 (sparse-matrix.h:352) and CG from 0 (:397), as the reference solvers do.
inline void SolveChannel(int channel_idx, int constraint, const ImageView &gx, const ImageView &gy, ImageView &output,
                         int iterations, const ImageView *init = nullptr, Solver solver = Solver::GaussSeidel,
                         int device = 0)
{
    const int W = gx.cols, H = gx.rows, C = gx.channels;
    if (gy.cols != W || gy.rows != H || gy.channels != C || output.cols != W || output.rows != H)
        throw std::invalid_argument("SolveChannel: image shapes differ");
    if (channel_idx < 0 || channel_idx >= C || channel_idx >= output.channels)
        throw std::invalid_argument("SolveChannel: bad channel index");
    // one-channel system: pick the channel's plane out of the interleaved gradients
    std::vector<float> px((std::size_t)W * H, 0.f), py((std::size_t)W * H, 0.f);
    for (int y = 0; y + 1 < H; ++y) {
        const float *rx = reinterpret_cast<const float *>(static_cast<const char *>(gx.data) + y * gx.step);
        const float *ry = reinterpret_cast<const float *>(static_cast<const char *>(gy.data) + y * gy.step);
        for (int x = 0; x + 1 < W; ++x) {
            px[(std::size_t)y * W + x] = rx[(std::size_t)x * C + channel_idx];
            py[(std::size_t)y * W + x] = ry[(std::size_t)x * C + channel_idx];
        }
    }
    detail::GridHandle h(W, H, 1, device);
    const int32_t pin = constraint;
    detail::check(ccp_grid_assemble_rhs(h.g, px.data(), py.data(), (int64_t)W * sizeof(float), &pin), "ccp_grid_assemble_rhs");
    if (init) {
        std::vector<uint8_t> plane((std::size_t)W * H);
        for (int y = 0; y < H; ++y) {
            const uint8_t *r = static_cast<const uint8_t *>(init->data) + y * init->step;
            for (int x = 0; x < W; ++x) plane[(std::size_t)y * W + x] = r[(std::size_t)x * init->channels + channel_idx];
        }
        detail::check(ccp_grid_set_x_u8(h.g, plane.data(), W), "ccp_grid_set_x_u8");
    } else {
        detail::check(ccp_grid_fill_x(h.g, solver == Solver::ConjugateGradient ? 0.0 : 1.0), "ccp_grid_fill_x");
    }
    detail::solve(h.g, solver, iterations, 1);
    std::vector<uint8_t> out((std::size_t)W * H);
    detail::check(ccp_grid_store_u8(h.g, out.data(), W), "ccp_grid_store_u8");
    for (int y = 0; y < H; ++y) {
        uint8_t *r = static_cast<uint8_t *>(output.data) + y * output.step;
        for (int x = 0; x < W; ++x) r[(std::size_t)x * output.channels + channel_idx] = out[(std::size_t)y * W + x];
    }
}

// The reference's own member-function shape (PhotoMontage.h:24):
//     void SolveChannel(int channel_idx, int constraint, const cv::Mat &color_gradient_x, const cv::Mat &color_gradient_y,
//                       cv::Mat &output, const std::vector<cv::Mat> &Images);
// with the state it reads from its object (PhotoMontage.h:40,59; PhotoMontage.cpp:599-613): `iterations_`,
// `fast_init_value` and `result_label_` — when fast_init_value is set the start vector is the composite
// Images[result_label_(y, x)](y, x)[channel_idx].  A call site written against the reference class compiles against
// this one with cv::Mat replaced by ccp::ImageView; `solver` picks what runs in the slot of the reference's
// conjugateGradient call (:613).
class PhotoMontage {
public:
    int fast_init_value = 0;                 // PhotoMontage.h:40
    int iterations_ = 50;                    // PhotoMontage.h:59 (set by Run, PhotoMontage.cpp:248)
    ImageView result_label_{nullptr, 0, 0, 1, 0};   // CV_8UC1 (BuildSolveMRF's labelling; PhotoMontage.cpp:391)
    Solver solver = Solver::GaussSeidel;
    int device = 0;

    void SolveChannel(int channel_idx, int constraint, const ImageView &color_gradient_x, const ImageView &color_gradient_y,
                      ImageView &output, const std::vector<ImageView> &Images)
    {
        const int W = color_gradient_x.cols, H = color_gradient_x.rows;
        if (!fast_init_value) {
            ccp::SolveChannel(channel_idx, constraint, color_gradient_x, color_gradient_y, output, iterations_, nullptr, solver, device);
            return;
        }
        if (result_label_.data == nullptr || result_label_.cols != W || result_label_.rows != H)
            throw std::invalid_argument("PhotoMontage::SolveChannel: fast_init_value needs result_label_ of the gradients' shape");
        // the composite image (PhotoMontage.cpp:599-610): channel `channel_idx` of an interleaved image of the
        // gradients' channel count, which is the form the free function takes its start vector in
        const int C = color_gradient_x.channels;
        std::vector<uint8_t> comp((std::size_t)W * H * C, 0);
        for (int y = 0; y < H; ++y) {
            const uint8_t *lab = static_cast<const uint8_t *>(result_label_.data) + y * result_label_.step;
            for (int x = 0; x < W; ++x) {
                const std::size_t k = lab[(std::size_t)x * result_label_.channels];
                if (k >= Images.size()) throw std::invalid_argument("PhotoMontage::SolveChannel: label out of range");
                const ImageView &im = Images[k];
                if (im.cols != W || im.rows != H || channel_idx >= im.channels)
                    throw std::invalid_argument("PhotoMontage::SolveChannel: image shapes differ");
                comp[((std::size_t)y * W + x) * C + channel_idx] =
                    (static_cast<const uint8_t *>(im.data) + y * im.step)[(std::size_t)x * im.channels + channel_idx];
            }
        }
        const ImageView init{comp.data(), H, W, C, (std::size_t)W * C};
        ccp::SolveChannel(channel_idx, constraint, color_gradient_x, color_gradient_y, output, iterations_, &init, solver, device);
    }
};

// BuildSolveGradientFusion(Images, ResultLabel) (PhotoMontage.cpp:410-436): gradient field of the
// label-selected images, three channel solves, clamped CV_8UC3 result — all three channels in one
// device pass each.  images[k]: CV_8UC3 views of equal shape; label: CV_8UC1.
inline void BuildSolveGradientFusion(const std::vector<ImageView> &images, const ImageView &label, ImageView &result,
                                     int iterations, bool fast_init_value = true, Solver solver = Solver::GaussSeidel,
                                     int device = 0)
{
    if (images.empty()) throw std::invalid_argument("BuildSolveGradientFusion: no images");
    const int W = label.cols, H = label.rows;
    std::vector<const uint8_t *> ptrs;
    for (const auto &im : images) {
        if (im.cols != W || im.rows != H || im.channels != 3 || im.step != images[0].step)
            throw std::invalid_argument("BuildSolveGradientFusion: images must be CV_8UC3 of the label's shape");
        ptrs.push_back(static_cast<const uint8_t *>(im.data));
    }
    detail::GridHandle h(W, H, 3, device);
    detail::check(ccp_grid_assemble_from_images(h.g, ptrs.data(), (int32_t)ptrs.size(), (int64_t)images[0].step,
                                                static_cast<const uint8_t *>(label.data), (int64_t)label.step,
                                                fast_init_value ? 1 : 0),
                  "ccp_grid_assemble_from_images");
    if (!fast_init_value) detail::check(ccp_grid_fill_x(h.g, solver == Solver::ConjugateGradient ? 0.0 : 1.0), "ccp_grid_fill_x");
    detail::solve(h.g, solver, iterations, 3);
    detail::check(ccp_grid_store_u8(h.g, static_cast<uint8_t *>(result.data), (int64_t)result.step), "ccp_grid_store_u8");
}

}  // namespace ccp
