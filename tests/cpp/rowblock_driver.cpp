// rowblock_driver.cpp — a C++ host driving the row-blocked multi-GPU Gauss-Seidel path through the C ABI
// only (include/ccp_gs.h): what the reference's call site — the per-channel solve of
// BuildSolveGradientFusion, project/src/PhotoMontage/PhotoMontage.cpp:428-434 — would do to spread one
// Poisson system over the GPUs of a node.  One process per GPU:
//
//   rowblock_driver <world> <rank> <id-file> <W> <H> <ghost> <iterations> [check_every] [epsilon]
//
// Rank 0 creates the communicator id (ccp_comm_unique_id) and publishes it through <id-file>; the other
// ranks wait for the file.  Every rank owns a contiguous block of image rows, builds the synthetic
// system on its device (x_true -> b = A x_true, x0 = 1 as sparse-matrix.h:352), runs
// ccp_grid_gauss_seidel_rowblocked and prints one line:
//   rank R rows [a,b) iterations K converged C l1 <step> rr <sum> bb <sum> abs <sum over owned rows of |x|>
// The sums rr, bb are global (all-reduced inside the library); abs is local.
//
// rank = -1: ALL ranks as threads of this one process (one handle pair per host thread, as the ABI asks) —
// with CCP_GS_RCCL_LIB=tests/cpp/libfake_rccl.so that is how a one-GPU test box runs the multi-rank path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ccp_gs.h"

#define OK(call)                                                                              \
    do {                                                                                      \
        int st_ = (call);                                                                     \
        if (st_ != CCP_OK) {                                                                  \
            std::fprintf(stderr, "%s: %s\n", #call, ccp_status_string(st_));                  \
            return 10 + st_;                                                                  \
        }                                                                                     \
    } while (0)

struct Args {
    int world, W, H, ghost, iters, check_every;
    double epsilon;
};

static int run_rank(const Args &a, int rank, int device, const uint8_t *id);

int main(int argc, char **argv)
{
    if (argc < 8) {
        std::fprintf(stderr, "usage: rowblock_driver world rank|-1 id-file W H ghost iterations [check_every] [epsilon]\n");
        return 64;
    }
    const int world = std::atoi(argv[1]), rank = std::atoi(argv[2]);
    const std::string id_file = argv[3];
    const int W = std::atoi(argv[4]), H = std::atoi(argv[5]), ghost = std::atoi(argv[6]), iters = std::atoi(argv[7]);
    const int check_every = argc > 8 ? std::atoi(argv[8]) : 0;
    const double epsilon = argc > 9 ? std::atof(argv[9]) : 0.0;
    const Args args{world, W, H, ghost, iters, check_every, epsilon};
    const int n_dev = ccp_device_count();
    if (n_dev < 1) {
        std::fprintf(stderr, "no HIP device: this path has no CPU fallback\n");
        return 2;
    }

    uint8_t id[CCP_COMM_ID_BYTES];
    if (rank < 0) {
        // every rank a thread of this process
        OK(ccp_comm_unique_id(id));
        std::vector<int> rc((size_t)world, 0);
        std::vector<std::thread> ts;
        for (int r = 0; r < world; ++r) ts.emplace_back([&, r] { rc[(size_t)r] = run_rank(args, r, r % n_dev, id); });
        for (auto &t : ts) t.join();
        for (int r = 0; r < world; ++r)
            if (rc[(size_t)r] != 0) return rc[(size_t)r];
        return 0;
    }
    if (rank == 0) {
        OK(ccp_comm_unique_id(id));
        const std::string tmp = id_file + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 3;
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) return 3;
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(id_file.c_str(), "rb")); ++tries)
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) return 4;
        std::fclose(f);
    }
    return run_rank(args, rank, rank % n_dev, id);
}

static int run_rank(const Args &a, int rank, int device, const uint8_t *id)
{
    const int world = a.world, W = a.W, H = a.H, ghost = a.ghost, iters = a.iters, check_every = a.check_every;
    const double epsilon = a.epsilon;
    ccp_comm *comm = nullptr;
    OK(ccp_comm_create(id, rank, world, device, &comm));

    // contiguous row blocks, the first H % world ranks one row taller
    const int base = H / world, extra = H % world;
    const int row_begin = rank * base + (rank < extra ? rank : extra);
    const int row_count = base + (rank < extra ? 1 : 0);
    ccp_grid_desc d{W, H, 1, row_begin, row_count, world > 1 ? ghost : 0, device, 0};
    ccp_grid *g = nullptr;
    OK(ccp_grid_create(&d, &g));
    OK(ccp_grid_randomize_x(g, 1234, 0.0, 255.0));         // x_true: a function of (seed, x, y) only
    OK(ccp_grid_b_from_x(g));
    OK(ccp_grid_fill_x(g, 1.0));
    OK(ccp_grid_attach_comm(g, comm));
    OK(ccp_grid_exchange_halos(g));
    ccp_gs_report rep{};
    OK(ccp_grid_gauss_seidel_rowblocked(g, epsilon, iters, check_every, &rep));
    double rr_bb[2] = {0, 0}, abs_sum = 0;
    OK(ccp_grid_residual_norm2_global(g, rr_bb));
    OK(ccp_grid_abs_sum(g, &abs_sum));
    int64_t exchanges = 0;
    int32_t wait_mode = 0, up = 0, down = 0;
    OK(ccp_grid_comm_stats(g, &exchanges, &wait_mode, &up, &down));
    std::printf("rank %d rows [%d,%d) iterations %d converged %d l1 %.17g rr %.17g bb %.17g abs %.17g exchanges %lld wait_mode %d\n", rank,
                row_begin, row_begin + row_count, rep.iterations, rep.converged, rep.last_l1_step, rr_bb[0], rr_bb[1], abs_sum,
                (long long)exchanges, wait_mode);
    OK(ccp_grid_attach_comm(g, nullptr));
    OK(ccp_grid_destroy(g));
    OK(ccp_comm_destroy(comm));
    return 0;
}
