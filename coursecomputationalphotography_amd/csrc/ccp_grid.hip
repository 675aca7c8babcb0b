// ccp_grid.hip — C ABI of the structured (matrix-free) Poisson grid path.  See include/ccp_gs.h.
#include "ccp_grid_kernels.hpp"
#include "ccp_grid_fused.hpp"
#include "ccp_grid_lex.hpp"
#include "ccp_cg.hpp"
#include "ccp_grid_cg.hpp"
#include "ccp_comm.hpp"

#include <algorithm>
#include <functional>
#include <cstring>
#include <chrono>
#include <vector>

using namespace ccp;

constexpr int kEdgeRing = 64;        // edge-counter slots (one per edge epoch, cleared together every kEdgeRing epochs)

struct ccp_grid {
    ccp_grid_desc desc{};
    Geom geom{};
    int ghost_top = 0, ghost_bottom = 0;
    // a side "shrinks" when the local block stops short of the image border there: its
    // outermost ghost row has no neighbour row, so validity recedes one row per half-sweep
    bool shrink_top = false, shrink_bottom = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // border tiles of a fused pass run here, beside the ordinary ones
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    // Row blocks with neighbours.  The pass that uses up the ghost rows finishes the owned rows the
    // neighbours need FIRST, inside the one launch (short edge chunks dispatched first): its waves count
    // themselves in *edge_counter and the last one publishes edge_epoch in *edge_flag (signal memory),
    // which the stream of the halo exchange waits for (ccp_grid_fused.hpp: fused_signal_edge).
    unsigned long long *edge_counter = nullptr;   // device memory: kEdgeRing counters, the pass of epoch e uses slot e % kEdgeRing
    unsigned long long *edge_flag = nullptr;      // hipMallocSignalMemory
    unsigned long long edge_epoch = 0;
    // Bounded device-side waits record giving up here (host-mapped memory): the polling kernel of wait_mode 1, which
    // gives up after edge_timeout_ticks (it must not hold a queue for ever when something — a profiler serialising
    // dispatches — keeps the pass from running beside it) and so lets the exchange send rows that may not be final;
    // and a wave of k_fused_multi whose input tiles never completed.  Every later call on the handle, and every call
    // that hands results to the host, then fails with CCP_ERR_STATE: a lost hand-off is an error, never a silently
    // wrong row.
    unsigned *edge_timeout = nullptr;
    unsigned long long edge_timeout_ticks = 200000000ull;   // 100 MHz constant clock: 2 s (CCP_GS_EDGE_TIMEOUT_TICKS)
    int wait_mode = 0;                   // 0: hipStreamWaitValue64, 1: a one-wave polling kernel (CCP_GS_EDGE_WAIT=spin, or no wait-value support)
    bool edge_signal = true;             // CCP_GS_EDGE_SIGNAL=0: the flag is only published after the whole pass
    // RCCL (ccp_grid_attach_comm): neighbour ranks, the rows their ghost zones take, the stream the
    // messages are issued on and the event the sweeps wait for
    ccp_comm *comm = nullptr;
    int up_rank = -1, down_rank = -1;
    int send_up = 0, send_down = 0;
    hipStream_t stream_comm = nullptr;
    hipEvent_t ev_comm = nullptr, ev_ready = nullptr;
    bool overlap = true;                 // exchange beside the rest of the last pass (ccp_grid_set_overlap)
    long exchanges = 0;                  // halo exchanges issued (statistics)
    DevBuf<double> x, b;
    DevBuf<double> cg_r, cg_p, cg_p2, cg_ap;   // conjugate-gradient work vectors, one channel each (p double-buffered: fused loop)
    DevBuf<CgState> cg_state;
    DevBuf<double> x_alt;        // ping-pong partner of x for the temporally blocked sweep
    // The CSR entry point's grid twins (ccp_csr.hip) ask for their layout at every solve: for them a run of passes may
    // end in either buffer — x and x_alt then swap roles — instead of being planned as an EVEN number of launches
    // (50 iterations at depth 8: seven launches instead of eight).  Handles whose x_dev a caller may hold keep the parity.
    bool allow_swap = false;
    // Dirichlet-mask grid (CCP_GRID_DIRICHLET_MASK): uniform 5-point stencil on the pixels whose mask byte
    // is 1, everything else fixed at zero.  maskp: one byte per pixel in the layout of x (one channel);
    // tile_live: which tiles of the current tiling hold any unknown (k_masked_tile_census), cached per tiling.
    bool masked = false;
    DevBuf<unsigned char> maskp, tile_live;
    DevBuf<int> tile_rows;       // ... and which of a live tile's own rows do (two ints per tile)
    int live_T = -1, live_R = -1, live_lo = -1, live_hi = -1;
    long unknowns = 0;           // mask bytes set (owned rows), for the statistics
    bool fuse = true;            // use k_fused_sweep for unchecked sweeps
    int multi = 0;               // CCP_GS_MULTI=1: consecutive passes of equal depth as ONE launch (k_fused_multi)
    DevBuf<unsigned> multi_words; // its ticket and per-tile completion counters
    bool xcd_swizzle = false;    // CCP_GS_XCD=1: tiles of a pass in XCD-contiguous runs (measured 2-4 % slower at 16384^2: off)
    const char *trace_file = nullptr;   // CCP_GS_TRACE_FILE: per-wave start/end stamps of every fused pass are appended here (diagnostics; syncs)
    DevBuf<unsigned long long> trace;
    bool short_edges = true;     // chunk rows at an image edge are short (CCP_GS_SHORT_EDGES=0 turns it off)
    int side_rows_override = 0;  // CCP_GS_SIDE_ROWS
    int fuse_tmax = kFusedMaxT;  // iterations fused per launch (<= kFusedMaxT)
    int rows_per_chunk = 128;    // rows a fused wave finalises (plus 4T halo rows); default for every T
    // per-depth launch cost (ms) and chunk rows measured by ccp_grid_tune; index = T
    float tune_ms[kFusedMaxT + 1] = {0};
    int tune_rows[kFusedMaxT + 1] = {0};
    bool tuned = false;
    bool all_border = false;     // debug (CCP_GS_ALL_BORDER): every tile of a pass goes through k_fused_border
    bool force_border = false;   // debug (CCP_GS_FORCE_BORDER): and every trip there takes the border body
    DevBuf<double> partial;      // per-block partial sums (L1 step / residual / checksums)
    long partial_region = 0;     // doubles per colour region of `partial` (L1 step)
    DevBuf<double> small;        // 4*kMaxChannels doubles of reduced results
    DevBuf<double> sweep_sums;   // row blocks: step sums of every sweep of a checked pass (kFusedMaxCheckedT * kMaxChannels)
    DevBuf<SolveState> state;
    DevBuf<int> redo_mask;       // per-channel flags for re-running one channel of a checked pass
    // lexicographic (reference-order) path: diagonal-major copies of x and b, snapshot, step sums
    DevBuf<double> lex_x, lex_b, lex_snap, lex_partial, lex_eps;
    DevBuf<unsigned> lex_progress, lex_ticket;   // strip-wave pipeline: diagonals finished per (channel, sweep, strip); work tickets
    DevBuf<double> lex_edges;    // time-skewed strips: results of each strip's lanes 62/63 per step and sweep (read by the strip to its right)
    int lex_mode = 3;            // 3: time-skewed strips, the T sweeps of a pass on the T waves of a workgroup (k_lex_wg, default);
                                 // CCP_GS_LEX_MODE=planes -> 0: one launch per hyperplane (k_lex_plane, the independent engine)
    int lex_tmax = 8;            // deepest time-skewed pass
                                 // k_lex_wg: ticket -> group * strips + strip, in the order the strips can start:
    unsigned *lex_order_pin = nullptr;   // ... built here, in pinned host memory the kernel reads directly (one word per strip, long before
    unsigned *lex_order_dev = nullptr;   //     the strip's gate opens): copying 34 KB to the device cost 7-8 ms of host time inside the timing
    size_t lex_order_pin_cap = 0;        //     of a 256-sweep solve, from pageable and from pinned memory alike (CCP_GS_DEBUG laps, round 4)
    size_t lex_order_count = 0;
    int lex_order_groups = 0, lex_order_strips = 0, lex_order_depth = 0;
    LexGeom lexg{};
    DevBuf<double> stage;        // natural-order staging rows for host transfers
    long stage_rows = 0;
    int half_sweeps_since_refresh = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;   // ccp_grid_region_begin / _end (created on first use)
    long region_launches = 0;    // passes / half-sweep launches since ccp_grid_region_begin
    long region_iterations = 0;  // iterations those passes performed (sum of their depths)
    float last_ms = 0.f;
    int last_launches = 0;
    bool timing_pending = false;
    int cpt = 2;                 // half-columns per thread of the sweep kernel
    int rows_per_block = 32;
};

namespace {

// Did a polling kernel give up on the edge flag (see ccp_grid::edge_timeout)?
int edge_timeout_status(const ccp_grid *g)
{
    if (g->edge_timeout && __atomic_load_n(g->edge_timeout, __ATOMIC_ACQUIRE) != 0) {
        if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] an edge hand-off timed out: ghost rows may hold rows that were not final\n");
        return CCP_ERR_STATE;
    }
    return CCP_OK;
}

int bind(ccp_grid *g)
{
    if (!g) return CCP_ERR_BAD_ARG;
    if (hipSetDevice(g->device) != hipSuccess) return CCP_ERR_NO_DEVICE;
    return edge_timeout_status(g);
}

long sweep_blocks_x(const ccp_grid *g) { return (g->geom.pitch + (long)kBlock * g->cpt - 1) / ((long)kBlock * g->cpt); }

void choose_tiling(ccp_grid *g)
{
    // Aim for >= 4096 workgroups (>> 256 CUs x 8 resident blocks) while keeping the
    // two-row halo re-read per row tile small.
    const long bx = sweep_blocks_x(g);
    long r = (long)g->geom.local_rows * bx * g->desc.channels / 4096;
    r = std::max<long>(4, std::min<long>(64, r));
    g->rows_per_block = (int)r;
}

template <bool L1>
int launch_half_sweep(ccp_grid *g, int c, int l_lo, int l_hi, const int *active)
{
    if (l_hi <= l_lo) return CCP_OK;
    const Geom &geo = g->geom;
    const int rpb = g->rows_per_block;
    dim3 grid((unsigned)sweep_blocks_x(g), (unsigned)((l_hi - l_lo + rpb - 1) / rpb), (unsigned)g->desc.channels);
    dim3 block(kBlock);
    double *part = g->partial.p + (long)c * g->partial_region;
#define CCP_LAUNCH_SWEEP(CPT_, SHFL_)                                                                          \
    hipLaunchKernelGGL((k_half_sweep<CPT_, L1, SHFL_>), grid, block, 0, g->stream, g->x.p, g->x.p, g->b.p, geo, \
                       c, l_lo, l_hi, rpb, part, active, static_cast<const unsigned char *>(nullptr))
    if (g->masked) {
        hipLaunchKernelGGL((k_half_sweep<2, L1, true, true>), grid, block, 0, g->stream, g->x.p, g->x.p, g->b.p, geo, c, l_lo, l_hi,
                           rpb, part, active, g->maskp.p);
    } else {
        CCP_LAUNCH_SWEEP(2, true);
    }
#undef CCP_LAUNCH_SWEEP
    CCP_HIP(hipGetLastError());
    g->last_launches++;
    g->region_launches++;
    return CCP_OK;
}

// Rows a half-sweep may update: everything on a side that is the image border, and on a side
// with ghosts one row fewer per half-sweep since the last halo refresh.
void sweep_range(const ccp_grid *g, int &l_lo, int &l_hi)
{
    const int s = g->half_sweeps_since_refresh;
    l_lo = g->shrink_top ? std::min(s + 1, g->ghost_top) : 0;
    l_hi = g->geom.local_rows - (g->shrink_bottom ? std::min(s + 1, g->ghost_bottom) : 0);
}

long l1_blocks_per_colour_channel(const ccp_grid *g, int l_lo, int l_hi)
{
    const int rpb = g->rows_per_block;
    return sweep_blocks_x(g) * (long)((l_hi - l_lo + rpb - 1) / rpb);
}

// blocks_out[c]: L1 block results per channel the colour-c half-sweep wrote.
int one_iteration(ccp_grid *g, bool l1, const int *active, long *blocks_out)
{
    for (int c = 0; c < 2; ++c) {
        int lo, hi;
        sweep_range(g, lo, hi);
        if ((g->shrink_top || g->shrink_bottom) && g->half_sweeps_since_refresh >= g->desc.ghost)
            return CCP_ERR_STATE;   // ghosts exhausted: refresh the halo first
        if (l1) {
            CCP_TRY(launch_half_sweep<true>(g, c, lo, hi, active));
            if (blocks_out) blocks_out[c] = l1_blocks_per_colour_channel(g, lo, hi);
        } else {
            CCP_TRY(launch_half_sweep<false>(g, c, lo, hi, active));
        }
        if (g->shrink_top || g->shrink_bottom) g->half_sweeps_since_refresh++;
    }
    return CCP_OK;
}

// Which tiles of a pass can touch a pixel with a missing neighbour (same predicates the kernels
// used to evaluate per wave): leading / trailing chunks by rows, leading / trailing strips by columns.
// edge_rows > 0 (row block with neighbours, the pass whose result is exchanged): chunk 0 / the last chunk
// are cut to exactly the owned rows the upper / lower neighbour takes, so they finish — and are
// signalled — early (FusedParams::first_edge / last_edge).
void fused_tile_counts(const ccp_grid *g, int T, FusedParams &P, int edge_rows = 0)
{
    const Geom &geo = g->geom;
    const int HS = 2 * T, R = P.rows_per_chunk;
    // Only IMAGE edges need the border arithmetic.  At the stale edge of a ghost zone the rows next to
    // the edge are invalid by construction (validity recedes one row per half-sweep, they are neither
    // stored nor used), so a tile there is an ordinary tile: the rows beyond the block read as 0.
    // A chunk row at an image edge is cut short: HS + 16 rows, enough for the chunk next to it to be
    // clear of the edge (its halo starts below image row 0 / ends above image row H-1).
    const int rows = P.st_hi - P.st_lo;
    const int edge_short = HS + 16;
    const bool at_top = geo.y0 + P.st_lo - HS <= 0, at_bot = geo.y0 + P.st_hi + HS >= geo.H - 1;
    const bool short_edges = g->short_edges;
    P.first_rows = (short_edges && at_top && rows > 2 * edge_short + R / 2 && R > edge_short) ? edge_short : 0;
    P.last_rows = (short_edges && at_bot && rows > 2 * edge_short + R / 2 && R > edge_short) ? edge_short : 0;
    P.first_edge = P.last_edge = 0;
    if (edge_rows > 0) {
        // rows from the first / last stored row that cover the neighbour's share of the owned rows
        int want_top = (g->shrink_top && !at_top) ? geo.own_lo + std::min(edge_rows, g->send_up > 0 ? g->send_up : edge_rows) - P.st_lo : 0;
        int want_bot = (g->shrink_bottom && !at_bot) ? P.st_hi - (geo.own_hi - std::min(edge_rows, g->send_down > 0 ? g->send_down : edge_rows)) : 0;
        want_top += want_top & 1;
        want_bot += want_bot & 1;
        const int others = P.first_rows + P.last_rows;
        if (want_top > 0 && want_bot > 0 && rows < want_top + want_bot + others + 2) want_top = want_bot = 0;   // too thin a block
        if (want_top > 0 && rows >= want_top + others + 2 && P.first_rows == 0) {
            P.first_rows = want_top;
            P.first_edge = 1;
        }
        if (want_bot > 0 && rows >= want_bot + P.first_rows + 2 && P.last_rows == 0) {
            P.last_rows = want_bot;
            P.last_edge = 1;
        }
    }
    const int mid = rows - P.first_rows - P.last_rows;
    if (P.first_edge || P.last_edge) {
        // the edge chunks take tile slots too: give the middle correspondingly fewer, taller chunks, so the
        // pass needs no more rounds on the chip's wave slots than it would without the hand-off (one
        // workgroup past a whole round costs a round).  (Round 3 tried the alternative — regular chunks of the usual
        // height plus short FOLLOW-UP chunks dispatched last, which take over the slots the edge tiles free after a
        // third of the pass: 0.760 ms per 32-iteration interval of a 2048-row block against 0.752 ms for this scheme and
        // 0.714 ms without the hand-off, profiles/r03_rank_block_edges.jsonl — two more chunk rows of halo cost what
        // the better packing saves.)
        const int n_whole = (rows + R - 1) / R;
        const int n_mid = std::max(1, n_whole - P.first_edge - P.last_edge);
        int r_mid = (mid + n_mid - 1) / n_mid;
        r_mid += r_mid & 1;
        P.rows_per_chunk = std::max(R, r_mid);
    }
    P.n_chunks = (mid + P.rows_per_chunk - 1) / P.rows_per_chunk + (P.first_rows > 0) + (P.last_rows > 0);
    auto top = [&](int c) { int ra, rb; fused_chunk_rows(P, c, ra, rb); return geo.y0 + ra - HS <= 0; };
    auto bot = [&](int c) { int ra, rb; fused_chunk_rows(P, c, ra, rb); return geo.y0 + rb + HS >= geo.H - 1; };
    P.nb_top = 0;
    while (P.nb_top < P.n_chunks && top(P.nb_top)) ++P.nb_top;
    P.nb_bot = 0;
    while (P.nb_top + P.nb_bot < P.n_chunks && bot(P.n_chunks - 1 - P.nb_bot)) ++P.nb_bot;
    const int U = fused_useful_px(T);
    auto left = [&](int s) { return s * U - fused_halo_px(T) <= 0; };
    auto right = [&](int s) { return s * U - fused_halo_px(T) + 2 * kStripLanes >= geo.W - 1; };
    P.ns_left = 0;
    while (P.ns_left < P.n_strips && left(P.ns_left)) ++P.ns_left;
    P.ns_right = 0;
    while (P.ns_left + P.ns_right < P.n_strips && right(P.n_strips - 1 - P.ns_right)) ++P.ns_right;
    if (g->all_border) {                 // debug: every tile through the border launch
        P.nb_top = P.n_chunks;
        P.nb_bot = 0;
    }
}

// Which tiles of a Dirichlet-mask grid hold unknowns: one census per tiling (depth, chunk rows, row range).
int masked_tile_census(ccp_grid *g, const FusedParams &P, int T)
{
    if (g->live_T != T || g->live_R != P.rows_per_chunk || g->live_lo != P.st_lo || g->live_hi != P.st_hi) {
        const size_t tiles = (size_t)P.n_chunks * P.n_strips;
        if (g->tile_live.n < tiles) CCP_TRY(g->tile_live.alloc(tiles));
        if (g->tile_rows.n < 2 * tiles) CCP_TRY(g->tile_rows.alloc(2 * tiles));
        hipLaunchKernelGGL(k_masked_tile_census, dim3((unsigned)P.n_strips, (unsigned)P.n_chunks), dim3(kWave), 0, g->stream, P, T,
                           g->tile_live.p, g->tile_rows.p);
        CCP_HIP(hipGetLastError());
        g->live_T = T;
        g->live_R = P.rows_per_chunk;
        g->live_lo = P.st_lo;
        g->live_hi = P.st_hi;
    }
    return CCP_OK;
}

// One pass of depth T over a Dirichlet-mask grid: every tile is an ordinary tile (zero lies outside the block
// as it does outside the region), tiles without any unknown in reach leave at once.
template <int T>
int launch_fused_masked(ccp_grid *g, FusedParams &P, int l1, long *l1_blocks)
{
    if (T > kMaskedMaxT) return CCP_ERR_BAD_ARG;
    constexpr int TM = T <= kMaskedMaxT ? T : 1;
    const int rows = P.st_hi - P.st_lo;
    P.first_rows = P.last_rows = 0;
    P.first_edge = P.last_edge = 0;
    P.n_chunks = (rows + P.rows_per_chunk - 1) / P.rows_per_chunk;
    P.nb_top = P.nb_bot = P.ns_left = P.ns_right = 0;
    P.side_rows = P.rows_per_chunk;
    P.side_subs = 1;
    P.side_rows_edge = P.rows_per_chunk;
    P.edge_counter = nullptr;
    P.edge_flag = nullptr;
    P.edge_target = P.edge_epoch = 0;
    P.trace = nullptr;
    const int waves = kBlock / kWave;
    dim3 grid((unsigned)((P.n_strips + waves - 1) / waves), (unsigned)P.n_chunks, (unsigned)g->desc.channels);
    if (l1_blocks) {
        l1_blocks[0] = (long)grid.x * grid.y;
        l1_blocks[1] = 0;
    }
    CCP_TRY(masked_tile_census(g, P, T));
    static const bool clip_rows = !(getenv("CCP_GS_MASK_ROWS") && atoi(getenv("CCP_GS_MASK_ROWS")) == 0);   // A/B: march every row of a live tile
    const int *rows_p = clip_rows ? g->tile_rows.p : nullptr;
    if (l1 == 2 && T > kMaskedMaxCheckedT) return CCP_ERR_BAD_ARG;
    constexpr int TMC = T <= kMaskedMaxCheckedT ? T : 1;
    if (l1 == 2) hipLaunchKernelGGL((k_fused_sweep_masked<TMC, 2, kFusedUnroll>), grid, dim3(kBlock), 0, g->stream, P, g->tile_live.p, rows_p);
    else if (l1 == 1) hipLaunchKernelGGL((k_fused_sweep_masked<TM, 1, kFusedUnroll>), grid, dim3(kBlock), 0, g->stream, P, g->tile_live.p, rows_p);
    else hipLaunchKernelGGL((k_fused_sweep_masked<TM, 0, kFusedUnroll>), grid, dim3(kBlock), 0, g->stream, P, g->tile_live.p, rows_p);
    CCP_HIP(hipGetLastError());
    g->last_launches++;
    g->region_launches++;
    g->region_iterations += T;
    return CCP_OK;
}

// One pass of depth T.  l1: 0 none, 1 step of the last sweep, 2 step of every sweep of the pass;
// l1_blocks[0/1]: block results per (sweep, channel) of the ordinary / the border launch.
// edge_rows > 0: the EDGE kernels (edge chunks first, in-launch signal); *signalled tells the caller
// whether the pass will publish g->edge_epoch itself.
template <int T>
int launch_fused_t(ccp_grid *g, const double *xin, double *xout, int st_lo, int st_hi, const int *active,
                   int l1 = 0, long *l1_blocks = nullptr, int rows_override = 0, int edge_rows = 0, bool *signalled = nullptr)
{
    FusedParams P;
    P.xin = xin;
    P.xout = xout;
    P.b = g->b.p;
    P.g = g->geom;
    P.st_lo = st_lo;
    P.st_hi = st_hi;
    P.rows_per_chunk = rows_override > 0 ? rows_override : ((g->tuned && g->tune_rows[T] > 0) ? g->tune_rows[T] : g->rows_per_chunk);
    const int U = fused_useful_px(T);
    P.n_strips = (g->geom.W + U - 1) / U;
    P.partial = g->partial.p;
    P.partial_border = g->partial.p + g->partial_region;
    P.active = active;
    P.xcd_swizzle = g->xcd_swizzle ? 1 : 0;
    const bool want_edge = edge_rows > 0 && l1 == 0 && active == nullptr && g->edge_counter && g->edge_signal;
    P.mask = g->maskp.p;
    if (g->masked) return launch_fused_masked<T>(g, P, l1, l1_blocks);
    fused_tile_counts(g, T, P, want_edge ? edge_rows : 0);
    const int waves = kBlock / kWave;
    const int edge_chunks = std::min(P.nb_top + P.nb_bot, P.n_chunks);
    const int edge_strips = std::min(P.ns_left + P.ns_right, P.n_strips);
    // side-strip tiles: ~2.5x the time per march step of an ordinary tile -> 2/5 of its march length
    {
        const int HS = 2 * T, R = P.rows_per_chunk;               // (possibly raised by fused_tile_counts for an edge pass)
        int sr = std::max(16, (R + 2 * HS) * 2 / 5 - 2 * HS);
        sr += sr & 1;
        if (g->side_rows_override > 0) sr = std::max(2, g->side_rows_override);
        P.side_rows = std::min(sr, R);
        P.side_subs = (R + P.side_rows - 1) / P.side_rows;
        // the side strips of an edge chunk are cut finer (their waves march ~2.5x slower and a sub-tile pays
        // 4T rows of halo march whatever its height: 16 rows end at about half a pass)
        P.side_rows_edge = std::min(P.side_rows, 16);
        if (P.first_edge || P.last_edge)
            P.side_subs = std::max(P.side_subs, (std::max(P.first_rows, P.last_rows) + P.side_rows_edge - 1) / P.side_rows_edge);
    }
    const long n_border = (long)edge_chunks * (P.n_strips - edge_strips) + (long)P.n_chunks * edge_strips * P.side_subs;
    const bool any_plain = edge_chunks < P.n_chunks && edge_strips < P.n_strips;
    dim3 grid((unsigned)((P.n_strips + waves - 1) / waves), (unsigned)P.n_chunks, (unsigned)g->desc.channels);
    dim3 bgrid((unsigned)((n_border + waves - 1) / waves), 1, (unsigned)g->desc.channels);
    if (l1_blocks) {
        l1_blocks[0] = any_plain ? (long)grid.x * grid.y : 0;
        l1_blocks[1] = n_border ? (long)bgrid.x : 0;
    }
    if (l1 == 2 && T > kFusedMaxCheckedT) return CCP_ERR_BAD_ARG;
    // in-launch signal: every wave of an edge chunk (inner strips: one wave per strip in either kernel;
    // edge strips: one wave per non-empty sub-tile) counts itself once per channel
    const bool edge = want_edge && (P.first_edge || P.last_edge);
    P.edge_counter = g->edge_counter;
    P.edge_flag = g->edge_flag;
    P.edge_epoch = g->edge_epoch;
    P.edge_target = 0;
    if (edge) {
        long waves_expected = 0;
        for (int c = 0; c < P.n_chunks; c += std::max(1, P.n_chunks - 1)) {      // chunk 0 and chunk n_chunks-1, once each
            if (!fused_is_edge_chunk(P, c)) continue;
            int ra, rb;
            fused_chunk_rows(P, c, ra, rb);
            const int subs = std::min(P.side_subs, (rb - ra + P.side_rows_edge - 1) / P.side_rows_edge);
            waves_expected += (P.n_strips - edge_strips) + (long)edge_strips * subs;
        }
        P.edge_target = (unsigned long long)waves_expected * g->desc.channels;
        // every edge pass counts from zero in a ring slot of its own (a miscount can cost one overlap, never
        // the next); the whole ring is cleared once per kEdgeRing epochs, off the critical path of a pass
        if (g->edge_epoch % kEdgeRing == 0) CCP_HIP(hipMemsetAsync(g->edge_counter, 0, sizeof(unsigned long long) * kEdgeRing, g->stream));
        P.edge_counter = g->edge_counter + (g->edge_epoch % kEdgeRing);
    }
    if (signalled) *signalled = edge;
    // diagnostics: per-wave time stamps of this pass (ordinary launch first, border launch behind it)
    const size_t trace_plain = (size_t)grid.x * grid.y * grid.z * waves * 4, trace_border = (size_t)bgrid.x * bgrid.z * waves * 4;
    unsigned long long *trace_p = nullptr, *trace_b = nullptr;
    if (g->trace_file) {
        if (g->trace.n < trace_plain + trace_border) CCP_TRY(g->trace.alloc(trace_plain + trace_border));
        CCP_HIP(hipMemsetAsync(g->trace.p, 0, (trace_plain + trace_border) * sizeof(unsigned long long), g->stream));
        trace_p = g->trace.p;
        trace_b = g->trace.p + trace_plain;
    }
    P.trace = trace_b;
    constexpr int TC = T <= kFusedMaxCheckedT ? T : 1;       // per-sweep sums exist up to kFusedMaxCheckedT
    hipStream_t bstream = g->stream2;
    // The border launch sees everything queued on the main stream so far, runs beside the ordinary
    // tiles, and whatever comes next on the main stream waits for it.
    if (n_border) {
        CCP_HIP(hipEventRecord(g->ev_main, g->stream));
        CCP_HIP(hipStreamWaitEvent(bstream, g->ev_main, 0));
    }
    // an EDGE pass issues its border tiles first: the few side-strip waves of the edge chunks must not queue
    // behind a chip full of ordinary tiles, or the hand-off would come at the end of the pass
    if (n_border && edge) {
        hipLaunchKernelGGL((k_fused_border<T, 0, kFusedUnroll, true>), bgrid, dim3(kBlock), 0, bstream, P, g->force_border ? 1 : 0);
    }
    if (any_plain) {
        P.trace = trace_p;
        if (l1 == 2) hipLaunchKernelGGL((k_fused_sweep<TC, 2, kFusedUnroll>), grid, dim3(kBlock), 0, g->stream, P);
        else if (l1 == 1) hipLaunchKernelGGL((k_fused_sweep<T, 1, kFusedUnroll>), grid, dim3(kBlock), 0, g->stream, P);
        else if (edge) hipLaunchKernelGGL((k_fused_sweep<T, 0, kFusedUnroll, true>), grid, dim3(kBlock), 0, g->stream, P);
        else hipLaunchKernelGGL((k_fused_sweep<T, 0, kFusedUnroll>), grid, dim3(kBlock), 0, g->stream, P);
    }
    if (n_border) {
        const int fb = g->force_border ? 1 : 0;
        P.trace = trace_b;
        if (l1 == 2) hipLaunchKernelGGL((k_fused_border<TC, 2, kFusedUnroll>), bgrid, dim3(kBlock), 0, bstream, P, fb);
        else if (l1 == 1) hipLaunchKernelGGL((k_fused_border<T, 1, kFusedUnroll>), bgrid, dim3(kBlock), 0, bstream, P, fb);
        else if (!edge) hipLaunchKernelGGL((k_fused_border<T, 0, kFusedUnroll>), bgrid, dim3(kBlock), 0, bstream, P, fb);
        CCP_HIP(hipEventRecord(g->ev_side, bstream));
        CCP_HIP(hipStreamWaitEvent(g->stream, g->ev_side, 0));
    }
    CCP_HIP(hipGetLastError());
    if (g->trace_file) {
        CCP_HIP(hipStreamSynchronize(g->stream));
        std::vector<unsigned long long> host(trace_plain + trace_border);
        CCP_HIP(hipMemcpy(host.data(), g->trace.p, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(g->trace_file, "ab")) {
            const unsigned long long head[8] = {0x43435054524143ull, (unsigned long long)T, grid.x, grid.y, grid.z, bgrid.x, (unsigned long long)host.size(),
                                                (unsigned long long)P.rows_per_chunk};
            fwrite(head, sizeof(head), 1, f);
            fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
            fclose(f);
        }
    }
    g->last_launches++;
    g->region_launches++;
    g->region_iterations += T;
    return CCP_OK;
}

// `n_passes` passes of depth T as ONE launch (k_fused_multi): xin -> xout -> xin ...; st_lo/st_hi per pass.  Returns
// CCP_ERR_UNSUPPORTED when the shape does not qualify (the caller then issues the passes one by one).
template <int T>
int launch_fused_multi_t(ccp_grid *g, int n_passes, const double *xin, double *xout, const int *st_lo, const int *st_hi)
{
    if (n_passes < 2 || n_passes > kMultiMaxPasses || g->trace_file) return CCP_ERR_UNSUPPORTED;
    if (g->masked && T > kMaskedMaxT) return CCP_ERR_UNSUPPORTED;
    FusedMultiParams M{};
    FusedParams &P = M.P;
    P.xin = xin;
    P.xout = xout;
    P.b = g->b.p;
    P.g = g->geom;
    P.st_lo = st_lo[0];
    P.st_hi = st_hi[0];
    P.rows_per_chunk = (g->tuned && g->tune_rows[T] > 0) ? g->tune_rows[T] : g->rows_per_chunk;
    const int U = fused_useful_px(T);
    P.n_strips = (g->geom.W + U - 1) / U;
    P.partial = g->partial.p;
    P.partial_border = g->partial.p + g->partial_region;
    P.active = nullptr;
    P.xcd_swizzle = 0;
    P.mask = g->maskp.p;
    P.trace = nullptr;
    const int HS = 2 * T;
    if (g->masked) {
        // every tile an ordinary tile, uniform chunks (launch_fused_masked's geometry)
        P.first_rows = P.last_rows = P.first_edge = P.last_edge = 0;
        P.n_chunks = (P.st_hi - P.st_lo + P.rows_per_chunk - 1) / P.rows_per_chunk;
        P.nb_top = P.nb_bot = P.ns_left = P.ns_right = 0;
        P.side_rows = P.side_rows_edge = P.rows_per_chunk;
        P.side_subs = 1;
        CCP_TRY(masked_tile_census(g, P, T));
    } else {
        fused_tile_counts(g, T, P, 0);
        const int R = P.rows_per_chunk;
        int sr = std::max(16, (R + 2 * HS) * 2 / 5 - 2 * HS);
        sr += sr & 1;
        if (g->side_rows_override > 0) sr = std::max(2, g->side_rows_override);
        P.side_rows = std::min(sr, R);
        P.side_subs = (R + P.side_rows - 1) / P.side_rows;
        P.side_rows_edge = P.side_rows;
    }
    P.edge_counter = nullptr;
    P.edge_flag = nullptr;
    P.edge_epoch = P.edge_target = 0;
    // a tile waits for the tiles two chunks up and down: of any two adjacent chunks at least one must be as tall
    // as the halo (2T rows) — true when the regular and the special chunk heights are, whatever the remainder chunk
    if (P.rows_per_chunk < HS || (P.first_rows > 0 && P.first_rows < HS) || (P.last_rows > 0 && P.last_rows < HS)) return CCP_ERR_UNSUPPORTED;
    for (int c = 0; c + 1 < P.n_chunks; ++c) {
        int ra, rb, rc, rd;
        fused_chunk_rows(P, c, ra, rb);
        fused_chunk_rows(P, c + 1, rc, rd);
        if (rb - ra < HS && rd - rc < HS) return CCP_ERR_UNSUPPORTED;
    }
    for (int q = 0; q < n_passes; ++q) {
        if (st_hi[q] <= st_lo[q] || st_lo[q] < st_lo[0] || st_hi[q] > st_hi[0]) return CCP_ERR_UNSUPPORTED;
        M.st_lo[q] = st_lo[q];
        M.st_hi[q] = st_hi[q];
    }
    const int waves = kBlock / kWave;
    const int edge_chunks = std::min(P.nb_top + P.nb_bot, P.n_chunks);
    const int edge_strips = std::min(P.ns_left + P.ns_right, P.n_strips);
    const long n_border = g->masked ? 0 : (long)edge_chunks * (P.n_strips - edge_strips) + (long)P.n_chunks * edge_strips * P.side_subs;
    M.n_passes = n_passes;
    M.gx = (P.n_strips + waves - 1) / waves;
    M.gy = P.n_chunks;
    M.bgx = (int)((n_border + waves - 1) / waves);
    M.channels = g->desc.channels;
    const size_t cells = (size_t)n_passes * M.channels * P.n_chunks * P.n_strips;
    if (g->multi_words.n < cells + 2) CCP_TRY(g->multi_words.alloc(cells + 2));
    CCP_HIP(hipMemsetAsync(g->multi_words.p, 0, (cells + 2) * sizeof(unsigned), g->stream));
    M.ticket = g->multi_words.p;
    M.cells = g->multi_words.p + 2;
    M.error = g->edge_timeout;
    M.tile_live = g->tile_live.p;
    const long total = (long)n_passes * (M.bgx + (long)M.gx * M.gy) * M.channels;
    if (total > 0x7fffffffL) return CCP_ERR_UNSUPPORTED;
    constexpr int TM = T <= kMaskedMaxT ? T : 1;
    if (g->masked) hipLaunchKernelGGL((k_fused_multi<TM, kFusedUnroll, true>), dim3((unsigned)total), dim3(kBlock), 0, g->stream, M);
    else hipLaunchKernelGGL((k_fused_multi<T, kFusedUnroll>), dim3((unsigned)total), dim3(kBlock), 0, g->stream, M);
    CCP_HIP(hipGetLastError());
    g->last_launches += n_passes;
    g->region_launches += n_passes;
    g->region_iterations += (long)T * n_passes;
    return CCP_OK;
}

template <int TMAX>
struct FusedMultiDepth {
    static int launch(int T, ccp_grid *g, int n, const double *xin, double *xout, const int *lo, const int *hi)
    {
        if (T == TMAX) return launch_fused_multi_t<TMAX>(g, n, xin, xout, lo, hi);
        return FusedMultiDepth<TMAX - 1>::launch(T, g, n, xin, xout, lo, hi);
    }
};
template <>
struct FusedMultiDepth<0> {
    static int launch(int, ccp_grid *, int, const double *, double *, const int *, const int *) { return CCP_ERR_UNSUPPORTED; }
};

// run-time depth -> the instantiation of that depth
template <int TMAX>
struct FusedDepth {
    static int launch(int T, ccp_grid *g, const double *xin, double *xout, int st_lo, int st_hi, const int *active,
                      int l1, long *l1_blocks, int rows_override = 0, int edge_rows = 0, bool *signalled = nullptr)
    {
        if (T == TMAX) return launch_fused_t<TMAX>(g, xin, xout, st_lo, st_hi, active, l1, l1_blocks, rows_override, edge_rows, signalled);
        return FusedDepth<TMAX - 1>::launch(T, g, xin, xout, st_lo, st_hi, active, l1, l1_blocks, rows_override, edge_rows, signalled);
    }
};
template <>
struct FusedDepth<0> {
    static int launch(int, ccp_grid *, const double *, double *, int, int, const int *, int, long *, int = 0, int = 0, bool * = nullptr) { return CCP_ERR_BAD_ARG; }
};

__global__ void k_publish_flag(unsigned long long *flag, unsigned long long epoch)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave polling the flag (wait_mode 1).  Bounded, so that it cannot hold a hardware queue for ever; a wait
// that gives up is RECORDED in *timed_out (host-mapped) and turns into CCP_ERR_STATE on the host side — what
// follows it on the stream may have sent rows that were not final.
__global__ void k_wait_flag(const unsigned long long *flag, unsigned long long epoch, unsigned long long ticks, unsigned *timed_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > ticks) {                         // 100 MHz constant clock
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

// A new epoch of the edge hand-off: returns the value waiters of THIS pass look for.
void edge_epoch_begin(ccp_grid *g) { g->edge_epoch++; }

// After the pass: whatever happened inside it, the flag reaches the epoch once the whole pass is done
// (also the only publication when the pass could not signal by itself).
int edge_epoch_publish_after_pass(ccp_grid *g)
{
    if (!g->edge_flag) return CCP_OK;
    if (g->wait_mode == 0) {
        CCP_HIP(hipStreamWriteValue64(g->stream, g->edge_flag, g->edge_epoch, 0));      // a queue packet, no kernel launch
    } else {
        hipLaunchKernelGGL(k_publish_flag, dim3(1), dim3(64), 0, g->stream, g->edge_flag, g->edge_epoch);
        CCP_HIP(hipGetLastError());
    }
    return CCP_OK;
}

// Make `s` wait until the edge rows of the last edge pass are final.
int edge_wait_on_stream(ccp_grid *g, hipStream_t s)
{
    if (!g->edge_flag) return CCP_OK;
    if (g->wait_mode == 0) {
        CCP_HIP(hipStreamWaitValue64(s, g->edge_flag, g->edge_epoch, hipStreamWaitValueGte, ~0ull));
    } else {
        hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, s, g->edge_flag, g->edge_epoch, g->edge_timeout_ticks, g->edge_timeout);
        CCP_HIP(hipGetLastError());
    }
    return CCP_OK;
}

// T fused iterations xin -> xout, with the ghost bookkeeping of 2T half-sweeps.
// edge_rows > 0 (row blocks with neighbours): the owned rows next to each neighbour are finished first
// inside the launch and published through the edge flag (see launch_fused_t); without an in-launch
// signal the flag is published after the pass.  Same tiles, same arithmetic.
int launch_fused(ccp_grid *g, int T, const double *xin, double *xout, const int *active, int l1 = 0,
                 long *l1_blocks = nullptr, int edge_rows = 0)
{
    const bool shrinking = g->shrink_top || g->shrink_bottom;
    const int s = g->half_sweeps_since_refresh;
    if (shrinking && s + 2 * T > g->desc.ghost) return CCP_ERR_STATE;   // ghosts exhausted: refresh first
    const int st_lo = g->shrink_top ? std::min(s + 2 * T, g->ghost_top) : 0;
    const int st_hi = g->geom.local_rows - (g->shrink_bottom ? std::min(s + 2 * T, g->ghost_bottom) : 0);
    if (st_hi > st_lo)
        CCP_TRY(FusedDepth<kFusedMaxT>::launch(T, g, xin, xout, st_lo, st_hi, active, l1, l1_blocks, 0, edge_rows, nullptr));
    if (shrinking) g->half_sweeps_since_refresh += 2 * T;
    return CCP_OK;
}

// `iterations` unchecked sweeps: an even number of fused launches (so the result lands back
// in g->x), a lone leftover iteration through the in-place half-sweep kernels.
// l1_last: the last launch also accumulates the L1 step of the final iteration (fused check).
// edge_rows > 0: a new edge epoch — the last launch hands the neighbours' rows over early.
int run_unchecked(ccp_grid *g, int iterations, const int *active = nullptr, bool l1_last = false,
                  long *l1_blocks = nullptr, int edge_rows = 0)
{
    if (edge_rows > 0) edge_epoch_begin(g);
    if (!g->fuse || iterations < 2) {
        if (l1_last) return CCP_ERR_STATE;
        for (int k = 0; k < iterations; ++k) CCP_TRY(one_iteration(g, false, active, nullptr));
        if (edge_rows > 0) CCP_TRY(edge_epoch_publish_after_pass(g));
        return CCP_OK;
    }
    if (!g->x_alt.p) {
        const size_t elems = (size_t)g->geom.ch_stride * g->desc.channels;
        CCP_TRY(g->x_alt.alloc(elems));
        CCP_HIP(hipMemsetAsync(g->x_alt.p, 0, elems * sizeof(double), g->stream));
    }
    // Split `iterations` into an EVEN number of launches of depth <= tmax with the least total
    // cost: measured per-depth launch times when the handle was tuned, otherwise "fewer, deeper
    // launches are cheaper".  f[i][p]: best cost for i iterations with launch-count parity p.
    const int tmax = g->fuse_tmax;
    // depth-1 passes only (a handle tuned for one iteration per exchange, CCP_GS_TMAX=1) cannot cover an odd
    // count in an even number of launches: the odd iteration goes through the in-place kernels
    if (tmax == 1 && (iterations & 1)) {
        if (l1_last) return CCP_ERR_STATE;
        CCP_TRY(run_unchecked(g, iterations - 1, active, false, nullptr, 0));
        CCP_TRY(one_iteration(g, false, active, nullptr));
        if (edge_rows > 0) CCP_TRY(edge_epoch_publish_after_pass(g));
        return CCP_OK;
    }
    auto cost = [&](int T) -> double { return g->tuned && g->tune_ms[T] > 0 ? (double)g->tune_ms[T] : 1.0 + 0.01 * T; };
    const double inf = 1e300;
    std::vector<double> f((size_t)(iterations + 1) * 2, inf);
    std::vector<int> step((size_t)(iterations + 1) * 2, 0);
    f[0] = 0.0;
    for (int i = 1; i <= iterations; ++i)
        for (int p = 0; p < 2; ++p)
            for (int T = 1; T <= tmax && T <= i; ++T) {
                const double c = f[(size_t)(i - T) * 2 + (p ^ 1)] + cost(T);
                if (c < f[(size_t)i * 2 + p]) {
                    f[(size_t)i * 2 + p] = c;
                    step[(size_t)i * 2 + p] = T;
                }
            }
    const bool free_parity = g->allow_swap && !g->ghost_top && !g->ghost_bottom && edge_rows == 0 &&
                             f[(size_t)iterations * 2 + 1] < f[(size_t)iterations * 2];
    std::vector<int> plan;
    for (int i = iterations, p = free_parity ? 1 : 0; i > 0;) {
        const int T = step[(size_t)i * 2 + p];
        if (T == 0) return CCP_ERR_STATE;
        plan.push_back(T);
        i -= T;
        p ^= 1;
    }
    std::sort(plan.begin(), plan.end(), std::greater<int>());
    double *cur = g->x.p, *alt = g->x_alt.p;
    for (size_t k = 0; k < plan.size();) {
        // consecutive passes of one depth as ONE launch (k_fused_multi), where that is switched on and the shape
        // qualifies; the pass that carries the stop rule or the edge hand-off keeps its own kernels
        if (g->multi && active == nullptr) {
            size_t j = k;
            while (j < plan.size() && plan[j] == plan[k] && (int)(j - k) < kMultiMaxPasses &&
                   !(j + 1 == plan.size() && (l1_last || edge_rows > 0)))
                ++j;
            const int n = (int)(j - k), T = plan[k];
            const bool shrinking = g->shrink_top || g->shrink_bottom;
            if (n >= 2 && !(shrinking && g->half_sweeps_since_refresh + 2 * T * n > g->desc.ghost)) {
                int lo[kMultiMaxPasses], hi[kMultiMaxPasses];
                for (int q = 0; q < n; ++q) {
                    const int s = g->half_sweeps_since_refresh + 2 * T * (q + 1);
                    lo[q] = g->shrink_top ? std::min(s, g->ghost_top) : 0;
                    hi[q] = g->geom.local_rows - (g->shrink_bottom ? std::min(s, g->ghost_bottom) : 0);
                }
                const int st = FusedMultiDepth<kFusedMaxT>::launch(T, g, n, cur, alt, lo, hi);
                if (st == CCP_OK) {
                    if (shrinking) g->half_sweeps_since_refresh += 2 * T * n;
                    if (n & 1) std::swap(cur, alt);
                    k = j;
                    continue;
                }
                if (st != CCP_ERR_UNSUPPORTED) return st;
            }
        }
        const bool last = k + 1 == plan.size();
        CCP_TRY(launch_fused(g, plan[k], cur, alt, active, (l1_last && last) ? 1 : 0, l1_blocks, last ? edge_rows : 0));
        std::swap(cur, alt);
        ++k;
    }
    if (cur != g->x.p) {
        // an odd number of passes (free_parity): the result is in the partner buffer, which becomes x
        if (!free_parity) return CCP_ERR_STATE;
        std::swap(g->x.p, g->x_alt.p);
        std::swap(g->x.n, g->x_alt.n);
    }
    if (edge_rows > 0) CCP_TRY(edge_epoch_publish_after_pass(g));
    return CCP_OK;
}

// Dirichlet-mask grids: p := 0 wherever the mask byte is 0 (all channels).  grid = (blocks, 1, channels)
__global__ void __launch_bounds__(kBlock)
k_zero_unmasked(double *__restrict__ p, const unsigned char *__restrict__ mask, long per_channel)
{
    double *__restrict__ q = p + (long)blockIdx.z * per_channel;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < per_channel; i += (long)gridDim.x * kBlock)
        if (mask[i] == 0) q[i] = 0.0;
}

int zero_unmasked(ccp_grid *g, double *plane)
{
    if (!g->masked) return CCP_OK;
    hipLaunchKernelGGL(k_zero_unmasked, dim3(2048, 1, (unsigned)g->desc.channels), dim3(kBlock), 0, g->stream, plane, g->maskp.p,
                       g->geom.ch_stride);
    CCP_HIP(hipGetLastError());
    return CCP_OK;
}

int begin_timing(ccp_grid *g)
{
    g->last_launches = 0;
    g->timing_pending = false;
    CCP_HIP(hipEventRecord(g->ev0, g->stream));
    return CCP_OK;
}

int end_timing(ccp_grid *g)
{
    CCP_HIP(hipEventRecord(g->ev1, g->stream));
    g->timing_pending = true;
    return CCP_OK;
}

// Host <-> device rows in natural order through the staging buffer.
template <bool TO_DEVICE>
int transfer_rows(ccp_grid *g, double *dev_base, int channel, double *rows, int first_row, int n_rows)
{
    CCP_TRY(bind(g));
    if (!rows || channel < 0 || channel >= g->desc.channels || n_rows < 0) return CCP_ERR_BAD_ARG;
    const int img_lo = g->geom.y0, img_hi = g->geom.y0 + g->geom.local_rows;
    if (first_row < img_lo || first_row + n_rows > img_hi) return CCP_ERR_BAD_ARG;
    const int W = g->desc.width;
    int done = 0;
    while (done < n_rows) {
        const int chunk = (int)std::min<long>(g->stage_rows, n_rows - done);
        const int l_first = first_row + done - g->geom.y0;
        dim3 grid((unsigned)((W + kBlock - 1) / kBlock), (unsigned)chunk);
        const size_t bytes = (size_t)chunk * W * sizeof(double);
        double *host = rows + (size_t)done * W;
        if (TO_DEVICE) {
            CCP_HIP(hipMemcpyAsync(g->stage.p, host, bytes, hipMemcpyHostToDevice, g->stream));
            hipLaunchKernelGGL((k_convert<true>), grid, dim3(kBlock), 0, g->stream, dev_base, g->stage.p,
                               g->geom, channel, l_first, W);
            CCP_HIP(hipGetLastError());
        } else {
            hipLaunchKernelGGL((k_convert<false>), grid, dim3(kBlock), 0, g->stream, dev_base, g->stage.p,
                               g->geom, channel, l_first, W);
            CCP_HIP(hipGetLastError());
            CCP_HIP(hipMemcpyAsync(host, g->stage.p, bytes, hipMemcpyDeviceToHost, g->stream));
        }
        CCP_HIP(hipStreamSynchronize(g->stream));
        done += chunk;
    }
    return edge_timeout_status(g);
}

}  // namespace

extern "C" {

int ccp_grid_create(const ccp_grid_desc *d, ccp_grid **out)
try {
    if (!d || !out) return CCP_ERR_BAD_ARG;
    *out = nullptr;
    if (d->width < 1 || d->height < 1 || d->channels < 1 || d->channels > kMaxChannels) return CCP_ERR_BAD_ARG;
    if (d->row_begin < 0 || d->row_count < 1 || d->row_begin + d->row_count > d->height || d->ghost < 0)
        return CCP_ERR_BAD_ARG;
    if ((long)d->width * d->height > 0x7fffffffL) return CCP_ERR_BAD_ARG;   // int32 indices, as the reference
    CCP_TRY(select_device(d->device));
    ccp_grid *g = new (std::nothrow) ccp_grid();
    if (!g) return CCP_ERR_ALLOC;
    g->desc = *d;
    g->device = d->device;
    g->ghost_top = d->row_begin > 0 ? std::min(d->ghost, d->row_begin) : 0;
    g->ghost_bottom = (d->row_begin + d->row_count < d->height) ? std::min(d->ghost, d->height - d->row_begin - d->row_count) : 0;
    Geom &geo = g->geom;
    geo.W = d->width;
    geo.H = d->height;
    geo.y0 = d->row_begin - g->ghost_top;
    g->shrink_top = geo.y0 > 0;
    g->shrink_bottom = d->row_begin + d->row_count + g->ghost_bottom < d->height;
    geo.local_rows = g->ghost_top + d->row_count + g->ghost_bottom;
    geo.own_lo = g->ghost_top;
    geo.own_hi = g->ghost_top + d->row_count;
    geo.pitch = (((long)d->width + 1) / 2 + 15) / 16 * 16;
    geo.ch_stride = (long)geo.local_rows * 2 * geo.pitch;
    g->cpt = 2;
    if (const char *e = getenv("CCP_GS_FUSE")) g->fuse = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_SHORT_EDGES")) g->short_edges = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_XCD")) g->xcd_swizzle = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_MULTI")) g->multi = atoi(e);
    if (const char *e = getenv("CCP_GS_TRACE_FILE")) g->trace_file = e[0] ? e : nullptr;
    if (const char *e = getenv("CCP_GS_SIDE_ROWS")) g->side_rows_override = atoi(e);
    if (const char *e = getenv("CCP_GS_LEX_MODE"))
        g->lex_mode = strcmp(e, "planes") == 0 ? 0 : 3;
    if (const char *e = getenv("CCP_GS_TMAX")) g->fuse_tmax = std::max(1, std::min(kFusedMaxT, atoi(e)));
    if (const char *e = getenv("CCP_GS_ALL_BORDER")) g->all_border = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_FORCE_BORDER")) g->force_border = atoi(e) != 0;
    if (const char *e = getenv("CCP_GS_CHUNK")) g->rows_per_chunk = std::max(1, atoi(e));
    choose_tiling(g);
    g->masked = (d->flags & CCP_GRID_DIRICHLET_MASK) != 0;
    if (g->masked) {
        // (a row block of a masked grid is a masked grid of its own rows plus ghosts: what lies beyond the ghosts is read
        // as zero like everything outside the region, and is as stale as any ghost — the trapezoid argument covers it)
        g->fuse_tmax = std::min(g->fuse_tmax, kMaskedMaxT);
        if (!getenv("CCP_GS_CHUNK")) g->rows_per_chunk = 160;   // what the tuner picks on region canvases of 4-80 M pixels (untuned handles)
    }

    const size_t elems = (size_t)geo.ch_stride * d->channels;
    int st = g->x.alloc(elems);
    if (st == CCP_OK && g->masked) {
        st = g->maskp.alloc((size_t)geo.ch_stride);
        if (st == CCP_OK && hipMemset(g->maskp.p, 0, (size_t)geo.ch_stride) != hipSuccess) st = CCP_ERR_HIP;
    }
    if (st == CCP_OK) st = g->b.alloc(elems);
    // partial sums: the finest launch is one block per (x tile, row, channel*2) with 2 doubles
    const size_t part = (size_t)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)) * geo.local_rows * d->channels * 2 * 2 + 64;
    g->partial_region = (long)(part / 2);
    if (st == CCP_OK) st = g->partial.alloc(part);
    if (st == CCP_OK) st = g->small.alloc(4 * kMaxChannels);
    if (st == CCP_OK) st = g->state.alloc(1);
    g->stage_rows = std::max<long>(1, std::min<long>(geo.local_rows, (8L << 20) / d->width));
    if (st == CCP_OK) st = g->stage.alloc((size_t)g->stage_rows * d->width);
    if (st == CCP_OK && (hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess)) st = CCP_ERR_HIP;
    if (st == CCP_OK && (hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking) != hipSuccess ||
                         hipEventCreateWithFlags(&g->ev_main, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&g->ev_side, hipEventDisableTiming) != hipSuccess))
        st = CCP_ERR_HIP;
    if (st == CCP_OK && (g->ghost_top || g->ghost_bottom)) {
        // edge hand-off of a row block with neighbours: counter in device memory, flag in signal memory
        int can_wait = 0;
        (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, g->device);
        g->wait_mode = can_wait ? 0 : 1;
        if (const char *e = getenv("CCP_GS_EDGE_WAIT")) g->wait_mode = (strcmp(e, "spin") == 0) ? 1 : (can_wait ? 0 : 1);
        if (const char *e = getenv("CCP_GS_EDGE_SIGNAL")) g->edge_signal = atoi(e) != 0;
        if (hipMalloc(reinterpret_cast<void **>(&g->edge_counter), sizeof(unsigned long long) * kEdgeRing) != hipSuccess ||
            hipExtMallocWithFlags(reinterpret_cast<void **>(&g->edge_flag), sizeof(unsigned long long), hipMallocSignalMemory) != hipSuccess ||
            hipMemset(g->edge_counter, 0, sizeof(unsigned long long) * kEdgeRing) != hipSuccess ||
            hipMemset(g->edge_flag, 0, sizeof(unsigned long long)) != hipSuccess)
            st = CCP_ERR_HIP;
        if (const char *e = getenv("CCP_GS_EDGE_TIMEOUT_TICKS")) g->edge_timeout_ticks = strtoull(e, nullptr, 10);
    }
    // the word in which a kernel records a bounded wait that gave up (k_wait_flag, k_fused_multi): host-mapped
    if (st == CCP_OK && hipHostMalloc(reinterpret_cast<void **>(&g->edge_timeout), sizeof(unsigned), hipHostMallocMapped) != hipSuccess) st = CCP_ERR_HIP;
    if (g->edge_timeout) *g->edge_timeout = 0;
    if (st == CCP_OK && (hipMemset(g->x.p, 0, elems * sizeof(double)) != hipSuccess ||
                         hipMemset(g->b.p, 0, elems * sizeof(double)) != hipSuccess))
        st = CCP_ERR_HIP;
    if (st != CCP_OK) {
        ccp_grid_destroy(g);
        return st;
    }
    *out = g;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_destroy(ccp_grid *g)
try {
    if (!g) return CCP_OK;
    (void)hipSetDevice(g->device);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->ev_r0) (void)hipEventDestroy(g->ev_r0);
    if (g->ev_r1) (void)hipEventDestroy(g->ev_r1);
    if (g->ev_main) (void)hipEventDestroy(g->ev_main);
    if (g->ev_side) (void)hipEventDestroy(g->ev_side);
    if (g->stream2) {
        (void)hipStreamSynchronize(g->stream2);
        (void)hipStreamDestroy(g->stream2);
    }
    if (g->stream_comm) {
        (void)hipStreamSynchronize(g->stream_comm);
        (void)hipStreamDestroy(g->stream_comm);
    }
    if (g->ev_comm) (void)hipEventDestroy(g->ev_comm);
    if (g->ev_ready) (void)hipEventDestroy(g->ev_ready);
    if (g->edge_counter) (void)hipFree(g->edge_counter);
    if (g->edge_flag) (void)hipFree(g->edge_flag);
    if (g->edge_timeout) (void)hipHostFree(g->edge_timeout);
    if (g->lex_order_pin) (void)hipHostFree(g->lex_order_pin);
    delete g;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_get_layout(ccp_grid *g, ccp_grid_layout *out)
try {
    if (!g || !out) return CCP_ERR_BAD_ARG;
    out->x_dev = g->x.p;
    out->b_dev = g->b.p;
    out->pitch = g->geom.pitch;
    out->local_rows = g->geom.local_rows;
    out->ghost_top = g->ghost_top;
    out->ghost_bottom = g->ghost_bottom;
    out->channels = g->desc.channels;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_set_stream(ccp_grid *g, void *hip_stream)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    g->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_synchronize(ccp_grid *g)
try {
    CCP_TRY(bind(g));
    CCP_HIP(hipStreamSynchronize(g->stream));
    return edge_timeout_status(g);
} CCP_ABI_CATCH

int ccp_grid_set_b_host(ccp_grid *g, int32_t channel, const double *rows, int32_t first_row, int32_t n_rows)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    CCP_TRY(transfer_rows<true>(g, g->b.p, channel, const_cast<double *>(rows), first_row, n_rows));
    return zero_unmasked(g, g->b.p);
} CCP_ABI_CATCH

int ccp_grid_set_x_host(ccp_grid *g, int32_t channel, const double *rows, int32_t first_row, int32_t n_rows)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    CCP_TRY(transfer_rows<true>(g, g->x.p, channel, const_cast<double *>(rows), first_row, n_rows));
    return zero_unmasked(g, g->x.p);
} CCP_ABI_CATCH

int ccp_grid_get_x_host(ccp_grid *g, int32_t channel, double *rows, int32_t first_row, int32_t n_rows)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    return transfer_rows<false>(g, g->x.p, channel, rows, first_row, n_rows);
} CCP_ABI_CATCH

int ccp_grid_get_b_host(ccp_grid *g, int32_t channel, double *rows, int32_t first_row, int32_t n_rows)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    return transfer_rows<false>(g, g->b.p, channel, rows, first_row, n_rows);
} CCP_ABI_CATCH

int ccp_grid_set_mask_host(ccp_grid *g, const uint8_t *mask, int64_t row_stride_bytes)
try {
    CCP_TRY(bind(g));
    if (!g->masked) return CCP_ERR_STATE;
    if (!mask || row_stride_bytes < g->desc.width) return CCP_ERR_BAD_ARG;
    const Geom &geo = g->geom;
    std::vector<unsigned char> split((size_t)geo.ch_stride, 0);
    long count = 0;
    for (int l = 0; l < geo.local_rows; ++l) {
        const int y = geo.y0 + l;
        const uint8_t *row = mask + (size_t)y * (size_t)row_stride_bytes;
        for (int x = 0; x < geo.W; ++x) {
            if (!row[x]) continue;
            split[(size_t)(((long)l * 2 + ((x + y) & 1)) * geo.pitch + (x >> 1))] = 1;
            if (l >= geo.own_lo && l < geo.own_hi) ++count;
        }
    }
    CCP_HIP(hipMemcpyAsync(g->maskp.p, split.data(), split.size(), hipMemcpyHostToDevice, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    g->unknowns = count;
    g->live_T = -1;                                      // the tile census belongs to the old mask
    CCP_TRY(zero_unmasked(g, g->x.p));
    CCP_TRY(zero_unmasked(g, g->b.p));
    if (g->x_alt.p) CCP_TRY(zero_unmasked(g, g->x_alt.p));
    return CCP_OK;
} CCP_ABI_CATCH

}  // extern "C"

// Library-internal twin of ccp_grid_set_mask_host for a mask that is already on the device in the grid's layout
// (the region recognition builds it there: ccp_csr.hip).  Asynchronous on the handle's stream.
int ccp::grid_set_allow_swap(ccp_grid *g, bool on)
{
    if (!g) return CCP_ERR_BAD_ARG;
    g->allow_swap = on;
    return CCP_OK;
}

int ccp::grid_set_mask_split_device(ccp_grid *g, const unsigned char *split_mask_dev, long unknowns)
{
    CCP_TRY(bind(g));
    if (!g->masked) return CCP_ERR_STATE;
    if (!split_mask_dev) return CCP_ERR_BAD_ARG;
    CCP_HIP(hipMemcpyAsync(g->maskp.p, split_mask_dev, (size_t)g->geom.ch_stride, hipMemcpyDeviceToDevice, g->stream));
    g->unknowns = unknowns;
    g->live_T = -1;                                      // the tile census belongs to the old mask
    CCP_TRY(zero_unmasked(g, g->x.p));
    CCP_TRY(zero_unmasked(g, g->b.p));
    if (g->x_alt.p) CCP_TRY(zero_unmasked(g, g->x_alt.p));
    CCP_HIP(hipStreamSynchronize(g->stream));            // the caller's buffer may go
    return CCP_OK;
}

extern "C" {

int ccp_grid_fill_x(ccp_grid *g, double value)
try {
    CCP_TRY(bind(g));
    const long n = g->geom.ch_stride * g->desc.channels;
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(kBlock), 0, g->stream, g->x.p, n, value);
    CCP_HIP(hipGetLastError());
    CCP_TRY(zero_unmasked(g, g->x.p));
    g->half_sweeps_since_refresh = 0;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_randomize_x(ccp_grid *g, uint64_t seed, double lo, double hi)
try {
    CCP_TRY(bind(g));
    dim3 grid((unsigned)((g->geom.pitch + kBlock - 1) / kBlock), (unsigned)g->geom.local_rows, (unsigned)g->desc.channels * 2);
    hipLaunchKernelGGL(k_randomize, grid, dim3(kBlock), 0, g->stream, g->x.p, g->geom, seed, lo, hi);
    CCP_HIP(hipGetLastError());
    CCP_TRY(zero_unmasked(g, g->x.p));
    g->half_sweeps_since_refresh = 0;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_b_from_x(ccp_grid *g)
try {
    CCP_TRY(bind(g));
    if ((g->shrink_top || g->shrink_bottom) && g->half_sweeps_since_refresh >= g->desc.ghost) return CCP_ERR_STATE;
    const Geom &geo = g->geom;
    // every local row whose neighbour rows are local too: the ghost rows need their b as well
    // (they are recomputed between halo exchanges); only the outermost ghost row cannot get one
    const int l_lo = g->shrink_top ? 1 : 0;
    const int l_hi = geo.local_rows - (g->shrink_bottom ? 1 : 0);
    dim3 grid((unsigned)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)), (unsigned)(l_hi - l_lo), (unsigned)g->desc.channels * 2);
    if (g->masked)
        hipLaunchKernelGGL((k_apply<2, 0, true>), grid, dim3(kBlock), 0, g->stream, g->x.p, g->b.p, g->b.p, geo, l_lo, g->partial.p, g->maskp.p);
    else
        hipLaunchKernelGGL((k_apply<2, 0>), grid, dim3(kBlock), 0, g->stream, g->x.p, g->b.p, g->b.p, geo, l_lo, g->partial.p,
                           static_cast<const unsigned char *>(nullptr));
    CCP_HIP(hipGetLastError());
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_halo_refreshed(ccp_grid *g)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    g->half_sweeps_since_refresh = 0;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_sweep(ccp_grid *g, int32_t iterations)
try {
    CCP_TRY(bind(g));
    if (iterations < 0) return CCP_ERR_BAD_ARG;
    CCP_TRY(begin_timing(g));
    CCP_TRY(run_unchecked(g, iterations));
    CCP_TRY(end_timing(g));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_sweep_edges_first(ccp_grid *g, int32_t iterations, int32_t edge_rows)
try {
    CCP_TRY(bind(g));
    if (iterations < 0 || edge_rows < 0) return CCP_ERR_BAD_ARG;
    if (!g->edge_flag) return ccp_grid_sweep(g, iterations);          // no neighbour blocks: nothing to hand over early
    CCP_TRY(begin_timing(g));
    CCP_TRY(run_unchecked(g, iterations, nullptr, false, nullptr, std::max(1, edge_rows)));
    CCP_TRY(end_timing(g));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_stream_wait_edges(ccp_grid *g, void *hip_stream)
try {
    CCP_TRY(bind(g));
    return edge_wait_on_stream(g, reinterpret_cast<hipStream_t>(hip_stream));
} CCP_ABI_CATCH

int ccp_grid_tune(ccp_grid *g, int32_t max_t, int32_t *chosen_t, int32_t *chosen_rows_per_chunk, float *ms_per_iteration)
try {
    CCP_TRY(bind(g));
    if (max_t < 1) return CCP_ERR_BAD_ARG;
    max_t = std::min<int>(max_t, g->masked ? kMaskedMaxT : kFusedMaxT);
    if (!g->x_alt.p) {
        const size_t elems = (size_t)g->geom.ch_stride * g->desc.channels;
        CCP_TRY(g->x_alt.alloc(elems));
        CCP_HIP(hipMemsetAsync(g->x_alt.p, 0, elems * sizeof(double), g->stream));
    }
    const int saved_chunk = g->rows_per_chunk, saved_launches = g->last_launches;
    const bool saved_tuned = g->tuned;
    g->tuned = false;                                  // candidates below set rows_per_chunk directly
    float tab_ms[kFusedMaxT + 1] = {0};
    int tab_rows[kFusedMaxT + 1] = {0};
    const int rows = g->geom.local_rows;
    const int fixed_candidates[] = {32, 48, 64, 80, 96, 112, 128, 160, 192, 256};
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g->device);
    float best = 1e30f;
    int best_t = 1, best_r = saved_chunk;
    hipEvent_t e0, e1;
    CCP_HIP(hipEventCreate(&e0));
    CCP_HIP(hipEventCreate(&e1));
    int status = CCP_OK;
    for (int T = 1; T <= max_t && status == CCP_OK; ++T) {
        // plus the chunk heights that fill the chip's wave slots in exactly 1..4 rounds: a short
        // row block has few tiles, and one tile past a whole round costs a round
        std::vector<int> chunk_candidates(std::begin(fixed_candidates), std::end(fixed_candidates));
        {
            const int U = fused_useful_px(T);
            const long blocks_x = ((g->geom.W + U - 1) / U + kBlock / kWave - 1) / (kBlock / kWave);
            const long slots = (long)cus * (g->masked ? masked_waves_per_simd(T) : fused_waves_per_simd(T));       // resident workgroups
            // chunk rows at an image edge are short ones of their own (fused_tile_counts)
            const int n_short = (g->geom.y0 - 2 * T <= 0) + (g->geom.y0 + rows + 2 * T >= g->geom.H - 1);
            const int short_rows = n_short * (2 * T + 16);
            for (int rounds = 1; rounds <= 4; ++rounds) {
                const long chunks = rounds * slots / (blocks_x * g->desc.channels) - n_short;
                if (chunks < 1 || rows <= short_rows) continue;
                int R = (int)((rows - short_rows + chunks - 1) / chunks);
                R += R & 1;
                if (R >= 16 && R <= 1024 && std::find(chunk_candidates.begin(), chunk_candidates.end(), R) == chunk_candidates.end())
                    chunk_candidates.push_back(R);
            }
        }
        for (int R : chunk_candidates) {
            if (R > rows && R != chunk_candidates[0]) continue;
            g->rows_per_chunk = R;
            auto once = [&]() -> int {
                return FusedDepth<kFusedMaxT>::launch(T, g, g->x.p, g->x_alt.p, 0, rows, nullptr, 0, nullptr);
            };
            status = once();                                   // warm (code, TLB)
            if (status != CCP_OK) break;
            (void)hipEventRecord(e0, g->stream);
            status = once();
            if (status == CCP_OK) status = once();
            (void)hipEventRecord(e1, g->stream);
            if (status != CCP_OK) break;
            if (hipEventSynchronize(e1) != hipSuccess) { status = CCP_ERR_HIP; break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const float per_launch = ms / 2.0f;
            if (getenv("CCP_GS_DEBUG"))
                fprintf(stderr, "[ccp_gs] tune T=%d rows/chunk=%d: %.4f ms/launch, %.5f ms/iteration\n", T, R, per_launch, per_launch / T);
            if (tab_ms[T] == 0.f || per_launch < tab_ms[T]) {
                tab_ms[T] = per_launch;
                tab_rows[T] = R;
            }
            const float per_iter = per_launch / T;
            if (per_iter < best) {
                best = per_iter;
                best_t = T;
                best_r = R;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    g->last_launches = saved_launches;
    g->rows_per_chunk = saved_chunk;
    if (status != CCP_OK) {
        g->tuned = saved_tuned;
        return status;
    }
    for (int T = 1; T <= kFusedMaxT; ++T) {
        g->tune_ms[T] = tab_ms[T];
        g->tune_rows[T] = tab_rows[T];
    }
    g->tuned = true;
    g->fuse_tmax = max_t;
    // Several passes in ONE launch (k_fused_multi) pay where a pass is short against the drain and fill between two
    // dependent launches: +5 % at 4096^2 x 3, nothing at 16384^2 (NOTES.md).  Decided here by timing four passes of the
    // chosen depth both ways on this shape (whole-image handles; CCP_GS_MULTI overrides).  x is put back afterwards.
    if (!getenv("CCP_GS_MULTI") && !g->shrink_top && !g->shrink_bottom && !g->trace_file && best_t >= 2) {
        const size_t elems = (size_t)g->geom.ch_stride * g->desc.channels;
        DevBuf<double> keep;
        if (keep.alloc(elems) == CCP_OK) {
            CCP_HIP(hipMemcpyAsync(keep.p, g->x.p, elems * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
            hipEvent_t t0, t1, t2;
            CCP_HIP(hipEventCreate(&t0));
            CCP_HIP(hipEventCreate(&t1));
            CCP_HIP(hipEventCreate(&t2));
            int lo[kMultiMaxPasses], hi[kMultiMaxPasses];
            for (int q = 0; q < kMultiMaxPasses; ++q) {
                lo[q] = 0;
                hi[q] = rows;
            }
            int st = CCP_OK;
            auto singles = [&]() {
                for (int q = 0; q < 4 && st == CCP_OK; ++q)
                    st = FusedDepth<kFusedMaxT>::launch(best_t, g, (q & 1) ? g->x_alt.p : g->x.p, (q & 1) ? g->x.p : g->x_alt.p, 0, rows, nullptr, 0, nullptr);
            };
            auto multi = [&]() { if (st == CCP_OK) st = FusedMultiDepth<kFusedMaxT>::launch(best_t, g, 4, g->x.p, g->x_alt.p, lo, hi); };
            singles();                                          // warm both
            multi();
            (void)hipEventRecord(t0, g->stream);
            singles();
            singles();
            (void)hipEventRecord(t1, g->stream);
            multi();
            multi();
            (void)hipEventRecord(t2, g->stream);
            float ms_single = 0.f, ms_multi = 0.f;
            if (st == CCP_OK && hipEventSynchronize(t2) == hipSuccess) {
                (void)hipEventElapsedTime(&ms_single, t0, t1);
                (void)hipEventElapsedTime(&ms_multi, t1, t2);
                g->multi = ms_multi < 0.98f * ms_single ? 1 : 0;
                if (getenv("CCP_GS_DEBUG"))
                    fprintf(stderr, "[ccp_gs] tune: 8 passes of depth %d one by one %.3f ms, four per launch %.3f ms -> %s\n", best_t, ms_single, ms_multi,
                            g->multi ? "several passes per launch" : "one pass per launch");
            }
            (void)hipEventDestroy(t0);
            (void)hipEventDestroy(t1);
            (void)hipEventDestroy(t2);
            CCP_HIP(hipMemcpyAsync(g->x.p, keep.p, elems * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
            CCP_HIP(hipStreamSynchronize(g->stream));
            g->last_launches = saved_launches;
            if (st != CCP_OK && st != CCP_ERR_UNSUPPORTED) return st;
        }
    }
    if (chosen_t) *chosen_t = best_t;
    if (chosen_rows_per_chunk) *chosen_rows_per_chunk = best_r;
    if (ms_per_iteration) *ms_per_iteration = best;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_set_fused(ccp_grid *g, int32_t on)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    g->fuse = on != 0;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_set_tiling(ccp_grid *g, int32_t max_t, int32_t rows_per_chunk)
try {
    if (!g || max_t < 1 || rows_per_chunk < 2) return CCP_ERR_BAD_ARG;
    g->fuse_tmax = std::min<int>(max_t, kFusedMaxT);
    g->rows_per_chunk = rows_per_chunk + (rows_per_chunk & 1);       // the march advances two rows per trip
    g->tuned = false;                                                // a tuning table would override the request
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_get_tiling(ccp_grid *g, int32_t *max_t, int32_t *rows_per_chunk, int32_t *tuned)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    const int T = g->fuse_tmax;
    if (max_t) *max_t = T;
    if (rows_per_chunk) *rows_per_chunk = (g->tuned && g->tune_rows[T] > 0) ? g->tune_rows[T] : g->rows_per_chunk;
    if (tuned) *tuned = g->tuned ? 1 : 0;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_sweep_l1(ccp_grid *g, double *l1_per_channel)
try {
    CCP_TRY(bind(g));
    if (!l1_per_channel) return CCP_ERR_BAD_ARG;
    const int C = g->desc.channels;
    CCP_TRY(begin_timing(g));
    long blocks[2] = {0, 0};
    CCP_TRY(one_iteration(g, true, nullptr, blocks));
    hipLaunchKernelGGL(k_check, dim3((unsigned)C), dim3(kBlock), 0, g->stream, g->partial.p, blocks[0],
                       g->partial.p + g->partial_region, blocks[1], 0.0, 0, static_cast<SolveState *>(nullptr), g->small.p);
    CCP_HIP(hipGetLastError());
    CCP_TRY(end_timing(g));
    CCP_HIP(hipMemcpyAsync(l1_per_channel, g->small.p, sizeof(double) * C, hipMemcpyDeviceToHost, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_gauss_seidel(ccp_grid *g, double epsilon, int32_t max_iteration, int32_t check_every, ccp_gs_report *report)
try {
    CCP_TRY(bind(g));
    if (g->ghost_top || g->ghost_bottom) return CCP_ERR_STATE;   // row blocks are driven by the caller (halo exchange)
    if (check_every < 0) return CCP_ERR_BAD_ARG;
    const int C = g->desc.channels;
    SolveState host{};
    for (int ch = 0; ch < C; ++ch) {
        host.active[ch] = 1;
        host.last_eps[ch] = 10.0;            // sparse-matrix.h:354
    }
    CCP_HIP(hipMemcpyAsync(g->state.p, &host, sizeof(host), hipMemcpyHostToDevice, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    CCP_TRY(begin_timing(g));
    int issued = 0;
    // `while (eps > epsilon && cnt < max_iteration)`: eps starts at 10
    const bool enter = (10.0 > epsilon);
    bool any_active = enter && max_iteration > 0;
    const int *active = reinterpret_cast<const int *>(g->state.p);
    if (any_active && check_every == 0) {
        // fixed count, no stop test: the temporally blocked sweep
        CCP_TRY(run_unchecked(g, max_iteration, nullptr));
        issued = max_iteration;
        any_active = false;
    }
    if (any_active && check_every >= 1 && g->fuse) {
        // The reference tests its stop rule after EVERY sweep (check_every = 1; k > 1 tests every k-th
        // sweep with the same machinery).  A temporally blocked pass knows the
        // previous level of every pixel it updates, so it reports the step of each of its T sweeps
        // (L1 = 2) at no extra traffic; k_decide_sums finds the first sweep that meets the rule.  If
        // that sweep is inside the pass, the channel is re-run from the pass's input buffer (still
        // intact: passes ping-pong) for exactly the missing sweeps — once per solve.
        const size_t elems = (size_t)g->geom.ch_stride * C;
        if (!g->x_alt.p) {
            CCP_TRY(g->x_alt.alloc(elems));
            CCP_HIP(hipMemsetAsync(g->x_alt.p, 0, elems * sizeof(double), g->stream));
        }
        if (!g->redo_mask.p) CCP_TRY(g->redo_mask.alloc(kMaxChannels));
        const size_t plane = (size_t)g->geom.ch_stride * sizeof(double);
        double *cur = g->x.p, *alt = g->x_alt.p;
        int was_active[kMaxChannels];
        for (int ch = 0; ch < C; ++ch) was_active[ch] = 1;
        // The host runs ONE PASS AHEAD of what it has looked at: pass n+1 is queued before the state after pass n is read
        // back, so the chip does not idle through a host round trip per pass (a pass of 4096^2 x 3 takes 0.3 ms; the
        // round trip was a tenth of it).  That is safe because the rule is applied on the device, in stream order: a
        // channel that stops in pass n is frozen before pass n+1 starts, which then leaves it — and its buffers — alone.
        if (!g->sweep_sums.p) CCP_TRY(g->sweep_sums.alloc((size_t)kFusedMaxCheckedT * kMaxChannels));
        struct Queued {
            SolveState st;
            int k0 = 0, T = 0;
            double *cur = nullptr, *alt = nullptr;     // the pass read cur and wrote alt
            hipEvent_t ev = nullptr;
        } ring[2];
        for (Queued &q : ring) CCP_HIP(hipEventCreateWithFlags(&q.ev, hipEventDisableTiming));
        int k0 = 0, n_queued = 0, n_seen = 0, status = CCP_OK;
        auto queue_pass = [&]() -> int {
            Queued &q = ring[n_queued & 1];
            q.k0 = k0;
            q.T = std::min(g->masked ? kMaskedMaxCheckedT : kFusedMaxCheckedT, max_iteration - k0);
            q.cur = cur;
            q.alt = alt;
            long blocks[2] = {0, 0};
            CCP_TRY(launch_fused(g, q.T, cur, alt, active, 2, blocks));
            // the step of each sweep (a block per channel and sweep), then the rule on them in sweep order
            hipLaunchKernelGGL(k_sweep_sums_wide, dim3((unsigned)C, (unsigned)q.T), dim3(kBlock), 0, g->stream, g->partial.p, blocks[0],
                               g->partial.p + g->partial_region, blocks[1], g->sweep_sums.p);
            hipLaunchKernelGGL(k_decide_sums, dim3(1), dim3(kMaxChannels), 0, g->stream, g->sweep_sums.p, C, q.T, k0 + 1, check_every, epsilon,
                               g->state.p);
            CCP_HIP(hipGetLastError());
            CCP_HIP(hipMemcpyAsync(&q.st, g->state.p, sizeof(q.st), hipMemcpyDeviceToHost, g->stream));
            CCP_HIP(hipEventRecord(q.ev, g->stream));
            k0 += q.T;
            std::swap(cur, alt);
            ++n_queued;
            return CCP_OK;
        };
        auto look_at_pass = [&]() -> int {
            Queued &q = ring[n_seen & 1];
            CCP_HIP(hipEventSynchronize(q.ev));
            host = q.st;
            any_active = false;
            for (int ch = 0; ch < C; ++ch) {
                any_active |= host.active[ch] != 0;
                if (!was_active[ch] || host.active[ch]) continue;
                was_active[ch] = 0;                                   // stopped inside this pass
                const int m = host.iterations[ch] - q.k0;             // sweeps of the pass it wanted: 1..T
                double *have = q.alt;                                 // where the channel's x_k is
                if (m < q.T) {
                    int mask[kMaxChannels] = {0};
                    mask[ch] = 1;
                    CCP_HIP(hipMemcpyAsync(g->redo_mask.p, mask, sizeof(mask), hipMemcpyHostToDevice, g->stream));
                    double *p = q.cur, *r = q.alt;
                    for (int left = m; left > 0;) {
                        const int t = std::min(left, g->masked ? kMaskedMaxT : kFusedMaxT);
                        CCP_TRY(launch_fused(g, t, p, r, g->redo_mask.p));
                        std::swap(p, r);
                        left -= t;
                    }
                    CCP_HIP(hipStreamSynchronize(g->stream));          // `mask` lives on this stack frame
                    have = p;
                }
                // a frozen channel is never touched again: keep its result in BOTH buffers
                double *other = (have == q.cur) ? q.alt : q.cur;
                CCP_HIP(hipMemcpyAsync(other + (size_t)ch * g->geom.ch_stride, have + (size_t)ch * g->geom.ch_stride, plane,
                                       hipMemcpyDeviceToDevice, g->stream));
            }
            ++n_seen;
            return CCP_OK;
        };
        while (status == CCP_OK && any_active && (n_seen < n_queued || k0 < max_iteration)) {
            // keep two passes queued while there are sweeps left, then look at the older one
            while (status == CCP_OK && n_queued - n_seen < 2 && k0 < max_iteration) status = queue_pass();
            if (status == CCP_OK) status = look_at_pass();
        }
        // (passes queued beyond the one that stopped the last channel find every channel frozen: they leave at once)
        if (n_seen < n_queued) (void)hipStreamSynchronize(g->stream);      // (a copy into `ring` may still be in flight)
        for (Queued &q : ring) (void)hipEventDestroy(q.ev);
        CCP_TRY(status);
        if (cur != g->x.p)
            CCP_HIP(hipMemcpyAsync(g->x.p, cur, elems * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        issued = k0;
        any_active = false;
    }
    const int batch_checks = 8;             // checked sweeps enqueued between two host polls
    while (any_active && issued < max_iteration) {
        int checks = 0;
        while (issued < max_iteration && checks < batch_checks) {
            if (g->fuse && check_every >= 2 && max_iteration - issued >= check_every) {
                // a whole check period in fused launches; the last one accumulates the L1 step of
                // the period's final sweep, so checking costs no extra pass over the grid
                long blocks[2] = {0, 0};
                CCP_TRY(run_unchecked(g, check_every, active, true, blocks));
                issued += check_every;
                hipLaunchKernelGGL(k_check, dim3((unsigned)C), dim3(kBlock), 0, g->stream, g->partial.p, blocks[0],
                                   g->partial.p + g->partial_region, blocks[1], epsilon, issued, g->state.p,
                                   static_cast<double *>(nullptr));
                CCP_HIP(hipGetLastError());
                ++checks;
                continue;
            }
            // check_every-1 unchecked sweeps (fused), then one sweep that accumulates the L1 step
            const int plain = std::min(check_every - 1, max_iteration - issued);
            if (plain > 0) {
                CCP_TRY(run_unchecked(g, plain, active));
                issued += plain;
            }
            if (issued >= max_iteration) break;
            long blocks[2] = {0, 0};
            CCP_TRY(one_iteration(g, true, active, blocks));
            ++issued;
            hipLaunchKernelGGL(k_check, dim3((unsigned)C), dim3(kBlock), 0, g->stream, g->partial.p, blocks[0],
                               g->partial.p + g->partial_region, blocks[1], epsilon, issued, g->state.p,
                               static_cast<double *>(nullptr));
            CCP_HIP(hipGetLastError());
            ++checks;
        }
        CCP_HIP(hipMemcpyAsync(&host, g->state.p, sizeof(host), hipMemcpyDeviceToHost, g->stream));
        CCP_HIP(hipStreamSynchronize(g->stream));
        any_active = false;
        for (int ch = 0; ch < C; ++ch) any_active |= host.active[ch] != 0;
    }
    CCP_TRY(end_timing(g));
    CCP_HIP(hipMemcpyAsync(&host, g->state.p, sizeof(host), hipMemcpyDeviceToHost, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    float ms = 0.f;
    CCP_HIP(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    g->last_ms = ms;
    g->timing_pending = false;
    if (report) {
        for (int ch = 0; ch < C; ++ch) {
            report[ch].converged = host.converged[ch];
            report[ch].iterations = host.converged[ch] ? host.iterations[ch] : issued;
            report[ch].last_l1_step = host.last_eps[ch];
            report[ch].seconds = ms * 1e-3;
        }
    }
    return CCP_OK;
} CCP_ABI_CATCH

namespace {

// `iterations` lexicographic sweeps of the channels in `mask`, pipelined over the hyperplanes
// tau = x + y + 2k (ccp_grid_lex.hpp).  partial != nullptr: per-sweep step sums are written too.
// partial sums one checked sweep writes per channel
// (time-skewed strips: the strip count depends on the depth; the partial layout uses the largest, depth 8's —
// slots a shallower launch does not write must read as zero, so the buffer is cleared per batch)
// Tickets of a launch of `groups` groups of T sweeps on an image W pixels wide, in the order the strips can start: (k, s)
// follows its left neighbour (k, s-1) by a strip lag (the 62 diagonals its first pixel lies further on + the 16 steps of
// edge values it wants to see published + the publication's own lag) and the same strip of the group before by a group
// lag; a workgroup that is resident but waiting keeps a slot from one that could run.  Everything a strip waits for —
// (k, s-1), (k-1, s), the last reader of its edge buffer (k - kLexEdgeSets, s+1): lex_wg_body — holds a smaller ticket, and a
// ticket's holder never waits for a larger one: no deadlock, whatever the residency (tests/test_lex_tickets.py checks
// exactly that through ccp_debug_lex_tickets).  order[ticket] = k * strips + s, strips = lex_strip_count(W, T, groups).
void lex_ticket_order(int W, int T, int groups, std::vector<unsigned> &order)
{
    const int S = lex_strip_count(W, T, groups);
    const long strip_lag = kLexSkewCols + 34, group_lag = 19 + 4 * (T - 1) + 16;
    struct Ticket { long start; int k, s; };
    std::vector<long> start((size_t)groups * S, -1);
    std::vector<Ticket> tickets;
    for (int k = 0; k < groups; ++k)
        for (int st = lex_strip_first(T, k); st <= lex_strip_last(W, T, k); ++st) {
            long t = 0;
            if (lex_strip_exists(W, T, k, st - 1)) t = std::max(t, start[(size_t)k * S + st - 1] + strip_lag);
            if (lex_strip_exists(W, T, k - 1, st)) t = std::max(t, start[(size_t)(k - 1) * S + st] + group_lag);
            if (lex_strip_exists(W, T, k - kLexEdgeSets, st + 1)) t = std::max(t, start[(size_t)(k - kLexEdgeSets) * S + st + 1] + 1);
            start[(size_t)k * S + st] = t;
            tickets.push_back({t, k, st});
        }
    std::stable_sort(tickets.begin(), tickets.end(), [](const Ticket &a, const Ticket &b) { return a.start < b.start; });
    order.clear();
    for (const Ticket &t : tickets) order.push_back((unsigned)((long)t.k * S + t.s));
}

// the diagonal-major arrays behind their front rows (kLexFrontRows: what k_lex_wg's loader prefetches for a strip that
// starts left of the image lies up to 64 diagonals before the first one; never used)
double *lex_xd(const ccp_grid *g) { return g->lex_x.p + (size_t)kLexFrontRows * g->lexg.P; }
double *lex_bd(const ccp_grid *g) { return g->lex_b.p + (size_t)kLexFrontRows * g->lexg.P; }
constexpr int kLexBatchSweeps = 128;     // sweeps in flight between two looks at the stop rule: the first batch (later ones grow while the rule is far off)
constexpr int kLexLaunchSweeps = 1024;   // most sweeps one launch of k_lex_wg carries (its strips move 2 columns left per sweep: lex_strip_count)
long lex_partials_per_sweep(const ccp_grid *g)
{
    if (g->lex_mode == 3) return (long)lex_strip_count(g->desc.width, 1, kLexLaunchSweeps);   // (the groups of a checked batch shift by 2 x its sweeps at most)
    return (long)g->lexg.n_diag * g->lexg.nbx;
}

// `iterations` lexicographic sweeps as time-skewed strips: groups of T sweeps per pass through memory (k_lex_wg),
// T = 8, 4, 2, 1 for what is left over; every depth is one launch holding all its groups.
extern "C++" {
template <int T>
int lex_launch_skew(ccp_grid *g, int groups, unsigned mask, double *partial, int t_last = T)
{
    const LexGeom &lg = g->lexg;
    const int C = g->desc.channels;
    if ((long)groups * T > kLexLaunchSweeps) return CCP_ERR_STATE;
    const bool dbg = getenv("CCP_GS_DEBUG") != nullptr;
    const auto t_in = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (dbg) fprintf(stderr, "[ccp_gs] lex_launch_skew %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
    };
    const int S = lex_strip_count(lg.W, T, groups);                      // strip slots per group (not every group has all of them)
    const int S_cap = lex_strip_count(lg.W, 1, kLexLaunchSweeps);       // ... of the largest launch, whatever its depth
    const long edge_steps = kWave + lg.H + 2 * (T - 1);
    // persistent workgroups: as many as are resident at once (k_lex_wg's comment), each taking strips from the ticket counter
    int per_cu = 0, cus = 0;
    if (g->masked) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_wg_masked<T, false>, (T + 2) * kWave, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lex_wg<T, false>, (T + 2) * kWave, 0);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g->device);
    const long resident = (long)std::max(per_cu, 1) * std::max(cus, 1);
    lap("occupancy known");
    // Buffers are sized for the largest launch a solve can issue (kLexLaunchSweeps), not for this one: a call with more
    // sweeps than the call before must not pay a 0.5 GB reallocation inside its own timing (the kernel trace of round 4
    // showed 25 ms of it between the layout conversion and the launch).
    const size_t need = (size_t)C * groups * S * kLexWordStride;
    const size_t need_cap = (size_t)C * std::max(groups, kLexLaunchSweeps / 8) * S_cap * kLexWordStride;
    const size_t edges = (size_t)kLexEdgeSets * C * S * edge_steps * 2 * T + (size_t)C * resident * kLexScratch;   // (+ the storers' scratch slots, one set per resident workgroup)
    const size_t edges_cap = (size_t)kLexEdgeSets * C * S_cap * (kWave + lg.H + 14) * 16 + (size_t)C * resident * kLexScratch;   // (what depth 8 needs)
    if (g->lex_progress.n < need) CCP_TRY(g->lex_progress.alloc(std::max(need, need_cap)));
    if (g->lex_order_groups != groups || g->lex_order_strips != S || g->lex_order_depth != T) {
        // Tickets in the order the strips can start: (k, s) follows its left neighbour (k, s-1) by a strip lag (the 62
        // diagonals its first pixel lies further on + the 16 steps of edge values it wants to see published + the
        // publication's own lag) and the same strip of the group before by a group lag; a workgroup that is resident but
        // waiting keeps a slot from one that could run.  Everything a strip waits for holds a smaller ticket — a ticket's
        // holder never waits for a larger one: no deadlock, whatever the residency.
        std::vector<unsigned> tickets;
        lex_ticket_order(lg.W, T, groups, tickets);
        if (tickets.empty()) return CCP_ERR_STATE;
        lap("tickets sorted");
        const size_t order_cap = std::max(tickets.size(), (size_t)(kLexLaunchSweeps / 8) * S_cap);
        CCP_HIP(hipStreamSynchronize(g->stream));               // (an earlier copy from the pinned buffer may still be in flight)
        lap("stream drained");
        if (g->lex_order_pin_cap < tickets.size()) {
            if (g->lex_order_pin) (void)hipHostFree(g->lex_order_pin);
            g->lex_order_pin = g->lex_order_dev = nullptr;
            g->lex_order_pin_cap = 0;
            CCP_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->lex_order_pin), order_cap * sizeof(unsigned), hipHostMallocMapped));
            CCP_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&g->lex_order_dev), g->lex_order_pin, 0));
            g->lex_order_pin_cap = order_cap;
        }
        for (size_t i = 0; i < tickets.size(); ++i) g->lex_order_pin[i] = tickets[i];
        __atomic_thread_fence(__ATOMIC_RELEASE);
        g->lex_order_count = tickets.size();
        g->lex_order_groups = groups;
        g->lex_order_strips = S;
        g->lex_order_depth = T;
        lap("order queued");
    }
    const unsigned n_tickets = (unsigned)g->lex_order_count;
    const long wgs = std::max<long>(1, std::min<long>((long)n_tickets, (resident + C - 1) / C));
    if (g->lex_edges.n < edges) CCP_TRY(g->lex_edges.alloc(std::max(edges, edges_cap)));
    if (!g->lex_ticket.p) CCP_TRY(g->lex_ticket.alloc(kMaxChannels));
    CCP_HIP(hipMemsetAsync(g->lex_progress.p, 0, need * sizeof(unsigned), g->stream));
    CCP_HIP(hipMemsetAsync(g->lex_ticket.p, 0, kMaxChannels * sizeof(unsigned), g->stream));
    dim3 grid((unsigned)wgs, (unsigned)C);
    lap("buffers cleared");
    if (getenv("CCP_GS_DEBUG"))
        fprintf(stderr, "[ccp_gs] k_lex_wg<%d>: %d workgroups per CU on %d CUs, %ld persistent workgroups per channel for %d groups x %d strips\n",
                T, per_cu, cus, wgs, groups, S);
    const dim3 block((T + 2) * kWave);
    double *nop = nullptr;
    unsigned long long *trace = nullptr;
    const size_t trace_words = (size_t)4 * C * groups * S;
    if (g->trace_file) {
        if (g->trace.n < trace_words) CCP_TRY(g->trace.alloc(trace_words));
        CCP_HIP(hipMemsetAsync(g->trace.p, 0, trace_words * sizeof(unsigned long long), g->stream));
        trace = g->trace.p;
    }
#define CCP_LEX_WG(KERNEL, CHECK, P, STRIDE)                                                                                        \
    hipLaunchKernelGGL((KERNEL<T, CHECK>), grid, block, 0, g->stream,                                                                \
                       LexWgArgs{lex_xd(g), lex_bd(g), g->geom, lg, groups, S, n_tickets, g->lex_progress.p, g->lex_ticket.p, g->lex_order_dev,  \
                                 g->lex_edges.p, edge_steps, mask, P, STRIDE, trace, t_last})
    if (g->masked) {
        if (partial) CCP_LEX_WG(k_lex_wg_masked, true, partial, lex_partials_per_sweep(g));
        else CCP_LEX_WG(k_lex_wg_masked, false, nop, 0L);
    } else {
        if (partial) CCP_LEX_WG(k_lex_wg, true, partial, lex_partials_per_sweep(g));
        else CCP_LEX_WG(k_lex_wg, false, nop, 0L);
    }
#undef CCP_LEX_WG
    CCP_HIP(hipGetLastError());
    if (trace) {
        CCP_HIP(hipStreamSynchronize(g->stream));
        std::vector<unsigned long long> host(trace_words);
        CCP_HIP(hipMemcpy(host.data(), trace, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(g->trace_file, "ab")) {
            // (the same 8-word header as the fused passes' records, tag "LEXTRAC")
            const unsigned long long head[8] = {0x4341525458454cull, (unsigned long long)T, (unsigned long long)groups, (unsigned long long)S,
                                                (unsigned long long)C, (unsigned long long)lg.H, (unsigned long long)host.size(), (unsigned long long)lg.W};
            fwrite(head, sizeof(head), 1, f);
            fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
            fclose(f);
        }
    }
    return CCP_OK;
}
}  // extern "C++"

int lex_run_skew(ccp_grid *g, int iterations, unsigned mask, double *partial)
{
    const int C = g->desc.channels;
    const long per_sweep = (long)C * lex_partials_per_sweep(g);          // partial doubles per sweep (all channels)
    int left = iterations, done = 0;
    if (g->lex_tmax >= 8 && left >= 8 && left % 8 != 0) {
        // a count that is not a multiple of 8: ONE launch of depth-8 groups whose last group passes the sweeps it does
        // not have through (lex_wg_pass_through) instead of remainder launches of depth 4, 2, 1, each a pipeline of its
        // own to fill and drain.  (With the stop rule too: the passed-through sweeps leave step sums of 0 in slots
        // beyond the batch's last sweep, which nobody reads; the partial buffer holds whole groups.)
        const int groups = (left + 7) / 8;
        return lex_launch_skew<8>(g, groups, mask, partial, left - 8 * (groups - 1));
    }
    for (int T = 8; T >= 1; T >>= 1) {
        if (T > g->lex_tmax || left < T) continue;
        const int groups = left / T;
        double *p = partial ? partial + (long)done * per_sweep : nullptr;
        if (T == 8) CCP_TRY(lex_launch_skew<8>(g, groups, mask, p));
        else if (T == 4) CCP_TRY(lex_launch_skew<4>(g, groups, mask, p));
        else if (T == 2) CCP_TRY(lex_launch_skew<2>(g, groups, mask, p));
        else CCP_TRY(lex_launch_skew<1>(g, groups, mask, p));
        done += groups * T;
        left -= groups * T;
    }
    return CCP_OK;
}

int lex_run(ccp_grid *g, int iterations, unsigned mask, double *partial)
{
    if (g->lex_mode == 3 && iterations > 0) return lex_run_skew(g, iterations, mask, partial);
    const LexGeom &lg = g->lexg;
    const int d_max = lg.n_diag - 1;
    const int C = g->desc.channels;
    for (int tau = 0; tau <= d_max + 2 * (iterations - 1); ++tau) {
        const int k_lo = std::max(0, (tau - d_max + 1) / 2);          // smallest k with tau - 2k <= d_max
        const int k_hi = std::min(iterations - 1, tau / 2);           // largest k with tau - 2k >= 0
        if (k_hi < k_lo) continue;
        dim3 grid((unsigned)lg.nbx, (unsigned)(k_hi - k_lo + 1), (unsigned)C);
        if (partial)
            hipLaunchKernelGGL((k_lex_plane<true>), grid, dim3(kBlock), 0, g->stream, lex_xd(g), lex_bd(g), g->geom, lg,
                               tau, k_lo, mask, partial);
        else
            hipLaunchKernelGGL((k_lex_plane<false>), grid, dim3(kBlock), 0, g->stream, lex_xd(g), lex_bd(g), g->geom, lg,
                               tau, k_lo, mask, static_cast<double *>(nullptr));
    }
    CCP_HIP(hipGetLastError());
    return CCP_OK;
}

}  // namespace

int ccp_grid_gauss_seidel_lexicographic(ccp_grid *g, double epsilon, int32_t max_iteration, int32_t check_every,
                                        ccp_gs_report *report)
try {
    CCP_TRY(bind(g));
    if (g->ghost_top || g->ghost_bottom || g->desc.row_count != g->desc.height) return CCP_ERR_STATE;   // whole image only
    if (g->masked && g->lex_mode != 3) return CCP_ERR_UNSUPPORTED;  // Dirichlet masks: k_lex_wg only
    if (max_iteration < 0 || check_every < 0) return CCP_ERR_BAD_ARG;
    const int C = g->desc.channels, W = g->desc.width, H = g->desc.height;
    LexGeom &lg = g->lexg;
    lg.W = W;
    lg.H = H;
    lg.P = ((long)W + 15) / 16 * 16;
    lg.n_diag = W + H - 1;
    lg.plane = (long)lg.n_diag * lg.P;
    lg.nbx = (std::min(W, H) + kLexTile - 1) / kLexTile;
    const size_t elems = (size_t)lg.plane * C;
    const size_t slack = (size_t)kLexSlackRows * lg.P;       // k_lex_wg prefetches some diagonals past the last one (never used)
    const size_t front = (size_t)kLexFrontRows * lg.P;       // ... and a strip that starts left of the image some diagonals before the first
    if (g->lex_x.n != front + elems + slack) {
        CCP_TRY(g->lex_x.alloc(front + elems + slack));
        CCP_TRY(g->lex_b.alloc(front + elems + slack));
        CCP_HIP(hipMemsetAsync(g->lex_x.p, 0, front * sizeof(double), g->stream));
        CCP_HIP(hipMemsetAsync(g->lex_b.p, 0, front * sizeof(double), g->stream));
        CCP_HIP(hipMemsetAsync(lex_xd(g) + elems, 0, slack * sizeof(double), g->stream));
        CCP_HIP(hipMemsetAsync(lex_bd(g) + elems, 0, slack * sizeof(double), g->stream));
    }
    CCP_TRY(begin_timing(g));
    dim3 cgrid((unsigned)((W + kBlock - 1) / kBlock), (unsigned)H, (unsigned)C);
    dim3 tgrid((unsigned)((W + kLexCT - 1) / kLexCT), (unsigned)((H + kLexCT - 1) / kLexCT), (unsigned)C);
    dim3 sgrid((unsigned)((W + kLexCT - 1) / kLexCT), (unsigned)((lg.n_diag + kLexCT - 1) / kLexCT), (unsigned)C);
    hipLaunchKernelGGL(k_lex_to_diag_sheared, sgrid, dim3(kBlock), 0, g->stream, g->x.p, lex_xd(g), g->geom, lg);
    if (g->masked) hipLaunchKernelGGL(k_lex_convert_b_masked, cgrid, dim3(kBlock), 0, g->stream, g->b.p, g->maskp.p, lex_bd(g), g->geom, lg);
    else hipLaunchKernelGGL(k_lex_to_diag_sheared, sgrid, dim3(kBlock), 0, g->stream, g->b.p, lex_bd(g), g->geom, lg);
    CCP_HIP(hipGetLastError());

    const unsigned all = (C >= 32) ? 0xffffffffu : ((1u << C) - 1u);
    int iterations_of[kMaxChannels], converged[kMaxChannels];
    double last_eps[kMaxChannels];
    for (int ch = 0; ch < C; ++ch) {
        iterations_of[ch] = max_iteration;
        converged[ch] = 0;
        last_eps[ch] = 10.0;                                              // `double eps = 10` (sparse-matrix.h:354)
    }
    if (check_every == 0 || !(10.0 > epsilon)) {
        // fixed count — or the reference loop never starts (eps = 10 <= epsilon)
        const int n = check_every == 0 ? max_iteration : 0;
        for (int done = 0; done < n;) {                    // gridDim.y carries the sweeps in flight: keep it small
            const int kb = std::min(g->lex_mode != 0 ? kLexLaunchSweeps : 32768, n - done);   // (k_lex_wg: one progress word per group and strip)
            CCP_TRY(lex_run(g, kb, all, nullptr));
            done += kb;
        }
        for (int ch = 0; ch < C; ++ch) iterations_of[ch] = n;
    } else {
        // Sweeps in flight between two looks at the rule: kLexBatchSweeps to begin with, doubled (up to one launch's worth)
        // while every channel's step, at the rate it has been shrinking, is more than four batches away from epsilon.  A
        // batch is one pipeline to fill and drain and one snapshot; a channel that stops inside a batch is redone from the
        // snapshot for exactly the sweeps the reference would have made — the result does not depend on the batching.  (The
        // reference's defaults, epsilon 1e-6 and 1,000 sweeps, never stop early on an image-sized system: 8 batches of 128
        // were 7.0 ms at 512^2 where one pipeline of 1,000 sweeps is 2.5 ms.)
        const int batch_cap = g->lex_mode == 3 ? kLexLaunchSweeps : kLexBatchSweeps;
        int batch = kLexBatchSweeps;
        const long per = lex_partials_per_sweep(g);                         // partials per (iteration, channel)
        if (g->lex_partial.n != (size_t)per * batch_cap * C) CCP_TRY(g->lex_partial.alloc((size_t)per * batch_cap * C));
        if (g->lex_eps.n != (size_t)batch_cap * C) CCP_TRY(g->lex_eps.alloc((size_t)batch_cap * C));
        if (g->lex_snap.n != elems) CCP_TRY(g->lex_snap.alloc(elems));
        std::vector<double> eps_vec((size_t)batch_cap * C);
        double *eps_host = eps_vec.data();
        unsigned mask = all;
        int done = 0;
        while (mask && done < max_iteration) {
            const int kb = std::min(batch, max_iteration - done);
            CCP_HIP(hipMemcpyAsync(g->lex_snap.p, lex_xd(g), elems * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
            if (g->lex_mode == 3) CCP_HIP(hipMemsetAsync(g->lex_partial.p, 0, sizeof(double) * (size_t)per * kb * C, g->stream));
            CCP_TRY(lex_run(g, kb, mask, g->lex_partial.p));
            hipLaunchKernelGGL(k_lex_reduce, dim3((unsigned)kb, (unsigned)C), dim3(kBlock), 0, g->stream, g->lex_partial.p, per,
                               g->lex_eps.p);
            CCP_HIP(hipGetLastError());
            CCP_HIP(hipMemcpyAsync(eps_host, g->lex_eps.p, sizeof(double) * kb * C, hipMemcpyDeviceToHost, g->stream));
            CCP_HIP(hipStreamSynchronize(g->stream));
            // how far the rule is, in sweeps, for the channel nearest to it (the step of the batch's last sweep against
            // the one half a batch earlier)
            double nearest = 1e300;
            for (int ch = 0; ch < C && kb >= 2; ++ch) {
                if (!((mask >> ch) & 1u)) continue;
                const double e1 = eps_host[(size_t)(kb - 1) * C + ch], e0 = eps_host[(size_t)(kb / 2 - 1) * C + ch];
                double left = 1e300;
                if (!(e1 > epsilon)) left = 0.0;
                else if (e1 < e0 && epsilon > 0.0) left = std::log(epsilon / e1) / (std::log(e1 / e0) / (double)(kb - kb / 2));
                nearest = std::min(nearest, left);
            }
            if (nearest > 4.0 * batch) batch = std::min(batch * 2, batch_cap);
            for (int ch = 0; ch < C; ++ch) {
                if (!((mask >> ch) & 1u)) continue;
                int stop = -1;
                for (int k = 0; k < kb; ++k) {
                    if ((done + k + 1) % check_every != 0) continue;
                    last_eps[ch] = eps_host[(size_t)k * C + ch];
                    if (!(last_eps[ch] > epsilon)) {
                        stop = k;
                        break;
                    }
                }
                if (stop < 0) continue;
                converged[ch] = 1;
                iterations_of[ch] = done + stop + 1;
                mask &= ~(1u << ch);
                if (stop < kb - 1) {
                    // the pipeline ran past the sweep the rule stops at: redo exactly stop+1 sweeps of
                    // this channel from the snapshot taken before the batch
                    CCP_HIP(hipMemcpyAsync(lex_xd(g) + (size_t)ch * lg.plane, g->lex_snap.p + (size_t)ch * lg.plane,
                                           (size_t)lg.plane * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
                    CCP_TRY(lex_run(g, stop + 1, 1u << ch, nullptr));
                }
            }
            done += kb;
        }
        for (int ch = 0; ch < C; ++ch)
            if (!converged[ch]) iterations_of[ch] = done;
    }
    hipLaunchKernelGGL((k_lex_convert_tiled<false>), tgrid, dim3(kBlock), 0, g->stream, g->x.p, lex_xd(g), g->geom, lg);
    CCP_HIP(hipGetLastError());
    CCP_TRY(end_timing(g));
    CCP_HIP(hipStreamSynchronize(g->stream));
    float ms = 0.f;
    CCP_HIP(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    g->last_ms = ms;
    g->timing_pending = false;
    if (report) {
        for (int ch = 0; ch < C; ++ch) {
            report[ch].converged = converged[ch];
            report[ch].iterations = iterations_of[ch];
            report[ch].last_l1_step = last_eps[ch];
            report[ch].seconds = ms * 1e-3;
        }
    }
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_debug_lex_tickets(int32_t width, int32_t depth, int32_t groups, uint32_t *order, int64_t capacity, int32_t *strips, int64_t *count)
try {
    if (width < 1 || groups < 1 || (depth != 1 && depth != 2 && depth != 4 && depth != 8) || (long)groups * depth > kLexLaunchSweeps || !strips || !count)
        return CCP_ERR_BAD_ARG;
    std::vector<unsigned> t;
    lex_ticket_order(width, depth, groups, t);
    *strips = lex_strip_count(width, depth, groups);
    *count = (int64_t)t.size();
    if (order) {
        if (capacity < (int64_t)t.size()) return CCP_ERR_BAD_ARG;
        std::copy(t.begin(), t.end(), order);
    }
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_conjugate_gradient(ccp_grid *g, double epsilon, int32_t max_iteration, ccp_gs_report *report)
try {
    CCP_TRY(bind(g));
    if (g->ghost_top || g->ghost_bottom || g->desc.row_count != g->desc.height) return CCP_ERR_STATE;
    const Geom &geo = g->geom;
    const long n = geo.ch_stride;                       // one channel incl. pads (pads stay 0 in b, r, p, Ap)
    if (!g->cg_r.p) {
        CCP_TRY(g->cg_r.alloc((size_t)n));
        CCP_TRY(g->cg_p.alloc((size_t)n));
        CCP_TRY(g->cg_p2.alloc((size_t)n));
        CCP_TRY(g->cg_ap.alloc((size_t)n));
        CCP_TRY(g->cg_state.alloc(1));
    }
    hipStream_t s = g->stream;
    dim3 grid((unsigned)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)), (unsigned)geo.local_rows, 2);
    for (int ch = 0; ch < g->desc.channels; ++ch) {
        CCP_HIP(hipMemsetAsync(g->cg_r.p, 0, sizeof(double) * n, s));
        CCP_HIP(hipMemsetAsync(g->cg_ap.p, 0, sizeof(double) * n, s));
        // matrix-free A*v on one channel: the kernel sees channel 0 of the offset pointers
        auto spmv = [&](const double *in, double *out) -> int {
            if (g->masked) hipLaunchKernelGGL((k_apply<2, 0, true>), grid, dim3(kBlock), 0, s, in, out, out, geo, 0, g->partial.p, g->maskp.p);
            else hipLaunchKernelGGL((k_apply<2, 0>), grid, dim3(kBlock), 0, s, in, out, out, geo, 0, g->partial.p, static_cast<const unsigned char *>(nullptr));
            return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
        };
        auto spmv_dot = [&](const double *in, double *out, int *n_partials) -> int {
            if (g->masked) hipLaunchKernelGGL((k_apply<2, 2, true>), grid, dim3(kBlock), 0, s, in, out, out, geo, 0, g->partial.p, g->maskp.p);
            else hipLaunchKernelGGL((k_apply<2, 2>), grid, dim3(kBlock), 0, s, in, out, out, geo, 0, g->partial.p, static_cast<const unsigned char *>(nullptr));
            *n_partials = (int)(grid.x * grid.y * grid.z);
            return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
        };
        // CCP_GS_CG_FUSED: 0 the three-pass loop (88 B), 2 the fused loop with the row-per-block pass A (its iterates are
        // the three-pass loop's bit for bit), otherwise the fused loop with the marching pass A (default)
        const int fused_mode = getenv("CCP_GS_CG_FUSED") ? atoi(getenv("CCP_GS_CG_FUSED")) : 1;
        const bool fused = fused_mode != 0;
        if (!fused) {
            CCP_TRY(cg_solve(spmv, spmv_dot, g->b.p + (long)ch * n, g->x.p + (long)ch * n, g->cg_r.p, g->cg_p.p, g->cg_ap.p, n, epsilon,
                             max_iteration, g->cg_state.p, g->partial.p, s, g->ev0, g->ev1, report ? report + ch : nullptr));
            continue;
        }
        // fused loop: 72 B per unknown and iteration (ccp_cg.hpp); same iterates bit for bit
        CCP_HIP(hipMemsetAsync(g->cg_p2.p, 0, sizeof(double) * n, s));
        const int march_rows = 32;
        const dim3 mgrid((unsigned)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)), (unsigned)((geo.local_rows + march_rows - 1) / march_rows));
        auto apply = [&](double *xv, const double *rv, const double *p_in, double *p_out, double *apv, int *n_partials) -> int {
            if (fused_mode != 2) {
                if (g->masked)
                    hipLaunchKernelGGL((k_cg_apply_march<true>), mgrid, dim3(kBlock), 0, s, xv, rv, p_in, p_out, apv, geo, march_rows, g->partial.p, g->maskp.p, g->cg_state.p,
                                       0, geo.local_rows);
                else
                    hipLaunchKernelGGL((k_cg_apply_march<false>), mgrid, dim3(kBlock), 0, s, xv, rv, p_in, p_out, apv, geo, march_rows, g->partial.p,
                                       static_cast<const unsigned char *>(nullptr), g->cg_state.p, 0, geo.local_rows);
                *n_partials = (int)(mgrid.x * mgrid.y);
                return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
            }
            if (g->masked)
                hipLaunchKernelGGL((k_cg_apply_fused<2, true>), grid, dim3(kBlock), 0, s, xv, rv, p_in, p_out, apv, geo, g->partial.p, g->maskp.p, g->cg_state.p);
            else
                hipLaunchKernelGGL((k_cg_apply_fused<2, false>), grid, dim3(kBlock), 0, s, xv, rv, p_in, p_out, apv, geo, g->partial.p,
                                   static_cast<const unsigned char *>(nullptr), g->cg_state.p);
            *n_partials = (int)(grid.x * grid.y * grid.z);
            return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
        };
        CCP_TRY(cg_solve_fused(spmv, apply, g->b.p + (long)ch * n, g->x.p + (long)ch * n, g->cg_r.p, g->cg_p.p, g->cg_p2.p, g->cg_ap.p, n,
                               epsilon, max_iteration, g->cg_state.p, g->partial.p, s, g->ev0, g->ev1, report ? report + ch : nullptr));
    }
    return CCP_OK;
} CCP_ABI_CATCH

namespace {

// g->small.p[2*ch] = sum (b - A x)^2, [2*ch+1] = sum b^2 over the OWNED rows (device, on g->stream)
int residual_to_small(ccp_grid *g)
{
    if ((g->shrink_top || g->shrink_bottom) && g->half_sweeps_since_refresh >= g->desc.ghost) return CCP_ERR_STATE;
    const Geom &geo = g->geom;
    const int C = g->desc.channels;
    dim3 grid((unsigned)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)), (unsigned)(geo.own_hi - geo.own_lo), (unsigned)C * 2);
    if (g->masked)
        hipLaunchKernelGGL((k_apply<2, 1, true>), grid, dim3(kBlock), 0, g->stream, g->x.p, g->b.p, g->b.p, geo, geo.own_lo, g->partial.p, g->maskp.p);
    else
        hipLaunchKernelGGL((k_apply<2, 1>), grid, dim3(kBlock), 0, g->stream, g->x.p, g->b.p, g->b.p, geo, geo.own_lo, g->partial.p,
                           static_cast<const unsigned char *>(nullptr));
    CCP_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_pair_reduce, dim3((unsigned)C), dim3(kBlock), 0, g->stream, g->partial.p, (long)grid.x * grid.y, g->small.p);
    CCP_HIP(hipGetLastError());
    return CCP_OK;
}

int small_to_rr_bb(ccp_grid *g, double *rr_bb)
{
    const int C = g->desc.channels;
    double host[2 * kMaxChannels];
    CCP_HIP(hipMemcpyAsync(host, g->small.p, sizeof(double) * 2 * C, hipMemcpyDeviceToHost, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    for (int ch = 0; ch < C; ++ch) {
        rr_bb[ch] = host[2 * ch];
        rr_bb[C + ch] = host[2 * ch + 1];
    }
    return edge_timeout_status(g);
}

}  // namespace

int ccp_grid_residual_norm2(ccp_grid *g, double *rr_bb)
try {
    CCP_TRY(bind(g));
    if (!rr_bb) return CCP_ERR_BAD_ARG;
    CCP_TRY(residual_to_small(g));
    return small_to_rr_bb(g, rr_bb);
} CCP_ABI_CATCH

int ccp_grid_abs_sum(ccp_grid *g, double *per_channel)
try {
    CCP_TRY(bind(g));
    if (!per_channel) return CCP_ERR_BAD_ARG;
    const int C = g->desc.channels;
    const unsigned blocks = 1024;
    hipLaunchKernelGGL(k_abs_sum, dim3(blocks, 1, (unsigned)C), dim3(kBlock), 0, g->stream, g->x.p, g->geom, g->partial.p);
    CCP_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sum_reduce, dim3((unsigned)C), dim3(kBlock), 0, g->stream, g->partial.p, (long)blocks, g->small.p);
    CCP_HIP(hipGetLastError());
    CCP_HIP(hipMemcpyAsync(per_channel, g->small.p, sizeof(double) * C, hipMemcpyDeviceToHost, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    return edge_timeout_status(g);
} CCP_ABI_CATCH

int ccp_grid_assemble_rhs(ccp_grid *g, const float *gx, const float *gy, int64_t row_stride_bytes, const int32_t *constraint)
try {
    CCP_TRY(bind(g));
    if (!gx || !gy || !constraint) return CCP_ERR_BAD_ARG;
    if (g->masked) return CCP_ERR_UNSUPPORTED;            // SolveChannel's right-hand side belongs to SolveChannel's matrix
    if (g->ghost_top || g->ghost_bottom || g->desc.row_count != g->desc.height) return CCP_ERR_STATE;
    const int W = g->desc.width, H = g->desc.height, C = g->desc.channels;
    const size_t row_bytes = (size_t)W * C * sizeof(float);
    if (row_stride_bytes < (int64_t)row_bytes) return CCP_ERR_BAD_ARG;
    DevBuf<float> dgx, dgy;
    DevBuf<int> dcons;
    CCP_TRY(dgx.alloc((size_t)W * H * C));
    CCP_TRY(dgy.alloc((size_t)W * H * C));
    CCP_TRY(dcons.alloc(C));
    // only rows y < H-1 and columns x < W-1 are defined in the reference (PhotoMontage.cpp:416-425);
    // the kernel never reads the rest, but the copy must not touch it on the host either.
    const size_t copy_rows = H > 1 ? (size_t)(H - 1) : 0;
    CCP_HIP(hipMemsetAsync(dgx.p, 0, (size_t)W * H * C * sizeof(float), g->stream));
    CCP_HIP(hipMemsetAsync(dgy.p, 0, (size_t)W * H * C * sizeof(float), g->stream));
    if (copy_rows && W > 1) {
        const size_t width_bytes = (size_t)(W - 1) * C * sizeof(float);
        CCP_HIP(hipMemcpy2DAsync(dgx.p, row_bytes, gx, (size_t)row_stride_bytes, width_bytes, copy_rows, hipMemcpyHostToDevice, g->stream));
        CCP_HIP(hipMemcpy2DAsync(dgy.p, row_bytes, gy, (size_t)row_stride_bytes, width_bytes, copy_rows, hipMemcpyHostToDevice, g->stream));
    }
    CCP_HIP(hipMemcpyAsync(dcons.p, constraint, sizeof(int) * C, hipMemcpyHostToDevice, g->stream));
    dim3 grid((unsigned)((W + kBlock - 1) / kBlock), (unsigned)H, (unsigned)C);
    hipLaunchKernelGGL(k_assemble_rhs, grid, dim3(kBlock), 0, g->stream, g->b.p, g->geom, dgx.p, dgy.p, C, dcons.p);
    CCP_HIP(hipGetLastError());
    CCP_HIP(hipStreamSynchronize(g->stream));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_assemble_from_images(ccp_grid *g, const uint8_t *const *images, int32_t n_images,
                                  int64_t image_stride_bytes, const uint8_t *label, int64_t label_stride_bytes,
                                  int32_t init_x_from_composite)
try {
    CCP_TRY(bind(g));
    if (!images || !label || n_images < 1 || n_images > 256) return CCP_ERR_BAD_ARG;
    if (g->desc.channels != 3 || g->masked) return CCP_ERR_UNSUPPORTED;      // BGR images; SolveChannel's matrix
    if (g->ghost_top || g->ghost_bottom || g->desc.row_count != g->desc.height) return CCP_ERR_STATE;
    const int W = g->desc.width, H = g->desc.height;
    if (image_stride_bytes < (int64_t)W * 3 || label_stride_bytes < W) return CCP_ERR_BAD_ARG;
    for (int k = 0; k < n_images; ++k)
        if (!images[k]) return CCP_ERR_BAD_ARG;
    // labels must select existing images (the reference indexes Images[label] unchecked)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if (label[(size_t)y * label_stride_bytes + x] >= n_images) return CCP_ERR_BAD_ARG;
    DevBuf<uint8_t> dimg, dlab;
    const size_t plane = (size_t)W * H * 3;
    CCP_TRY(dimg.alloc(plane * n_images));
    CCP_TRY(dlab.alloc((size_t)W * H));
    for (int k = 0; k < n_images; ++k)
        CCP_HIP(hipMemcpy2DAsync(dimg.p + plane * k, (size_t)W * 3, images[k], (size_t)image_stride_bytes, (size_t)W * 3,
                                 (size_t)H, hipMemcpyHostToDevice, g->stream));
    CCP_HIP(hipMemcpy2DAsync(dlab.p, (size_t)W, label, (size_t)label_stride_bytes, (size_t)W, (size_t)H,
                             hipMemcpyHostToDevice, g->stream));
    dim3 grid((unsigned)((W + kBlock - 1) / kBlock), (unsigned)H, 3);
    if (init_x_from_composite)
        hipLaunchKernelGGL((k_assemble_from_images<true>), grid, dim3(kBlock), 0, g->stream, g->b.p, g->x.p, g->geom, dimg.p, dlab.p);
    else
        hipLaunchKernelGGL((k_assemble_from_images<false>), grid, dim3(kBlock), 0, g->stream, g->b.p, g->x.p, g->geom, dimg.p, dlab.p);
    CCP_HIP(hipGetLastError());
    CCP_HIP(hipStreamSynchronize(g->stream));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_store_u8(ccp_grid *g, uint8_t *out, int64_t row_stride_bytes)
try {
    CCP_TRY(bind(g));
    if (!out) return CCP_ERR_BAD_ARG;
    if (g->ghost_top || g->ghost_bottom || g->desc.row_count != g->desc.height) return CCP_ERR_STATE;
    const int W = g->desc.width, H = g->desc.height, C = g->desc.channels;
    if (row_stride_bytes < (int64_t)W * C) return CCP_ERR_BAD_ARG;
    DevBuf<uint8_t> d;
    CCP_TRY(d.alloc((size_t)W * H * C));
    dim3 grid((unsigned)((W + kBlock - 1) / kBlock), (unsigned)H, (unsigned)C);
    hipLaunchKernelGGL(k_store_u8, grid, dim3(kBlock), 0, g->stream, g->x.p, g->geom, d.p, C);
    CCP_HIP(hipGetLastError());
    CCP_HIP(hipMemcpy2DAsync(out, (size_t)row_stride_bytes, d.p, (size_t)W * C, (size_t)W * C, (size_t)H, hipMemcpyDeviceToHost, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_set_x_u8(ccp_grid *g, const uint8_t *image, int64_t row_stride_bytes)
try {
    CCP_TRY(bind(g));
    if (!image) return CCP_ERR_BAD_ARG;
    if (g->ghost_top || g->ghost_bottom || g->desc.row_count != g->desc.height) return CCP_ERR_STATE;
    const int W = g->desc.width, H = g->desc.height, C = g->desc.channels;
    if (row_stride_bytes < (int64_t)W * C) return CCP_ERR_BAD_ARG;
    DevBuf<uint8_t> d;
    CCP_TRY(d.alloc((size_t)W * H * C));
    CCP_HIP(hipMemcpy2DAsync(d.p, (size_t)W * C, image, (size_t)row_stride_bytes, (size_t)W * C, (size_t)H, hipMemcpyHostToDevice, g->stream));
    dim3 grid((unsigned)((W + kBlock - 1) / kBlock), (unsigned)H, (unsigned)C);
    hipLaunchKernelGGL(k_load_u8, grid, dim3(kBlock), 0, g->stream, g->x.p, g->geom, d.p, C);
    CCP_HIP(hipGetLastError());
    CCP_TRY(zero_unmasked(g, g->x.p));
    CCP_HIP(hipStreamSynchronize(g->stream));
    return CCP_OK;
} CCP_ABI_CATCH

// ============================================================================================
// Row blocks over RCCL: neighbour halo exchange and all-reduced norms behind the C ABI.
// ============================================================================================
namespace {

bool has_neighbours(const ccp_grid *g) { return g->up_rank >= 0 || g->down_rank >= 0; }

// The messages of one exchange, all in one RCCL group on `s`: the outermost owned rows go to the
// neighbours' ghost rows, theirs arrive in ours.  One image row of one channel is 2*pitch contiguous
// doubles, so a block of rows is one message per channel and direction.
int issue_exchange(ccp_grid *g, hipStream_t s, double *base = nullptr)
{
    const RcclApi *api = rccl_api();
    if (!api || !g->comm) return CCP_ERR_STATE;
    const Geom &geo = g->geom;
    const size_t row = (size_t)2 * geo.pitch;
    if (!base) base = g->x.p;                        // (a checked solve exchanges the buffer its iterate is in)
    CCP_RCCL(api->GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int ch = 0; ch < g->desc.channels && r == ncclSuccess; ++ch) {
        double *x = base + (size_t)ch * geo.ch_stride;
        if (g->up_rank >= 0) {
            r = api->Send(x + (size_t)geo.own_lo * row, (size_t)g->send_up * row, ncclDouble, g->up_rank, g->comm->comm, s);
            if (r == ncclSuccess) r = api->Recv(x, (size_t)g->ghost_top * row, ncclDouble, g->up_rank, g->comm->comm, s);
        }
        if (g->down_rank >= 0 && r == ncclSuccess) {
            r = api->Send(x + (size_t)(geo.own_hi - g->send_down) * row, (size_t)g->send_down * row, ncclDouble, g->down_rank,
                          g->comm->comm, s);
            if (r == ncclSuccess) r = api->Recv(x + (size_t)geo.own_hi * row, (size_t)g->ghost_bottom * row, ncclDouble, g->down_rank,
                                               g->comm->comm, s);
        }
    }
    const ncclResult_t e = api->GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "ncclSend/ncclRecv", __FILE__, __LINE__);
    CCP_RCCL(e);
    return CCP_OK;
}

// Refresh the ghost rows.  after_edges: the sweeps were issued with an edge epoch — the messages wait for
// the edge flag only and travel beside the rest of the pass; otherwise they wait for everything queued.
int exchange(ccp_grid *g, bool after_edges, double *base = nullptr)
{
    if (!g->comm) return CCP_ERR_STATE;
    if (has_neighbours(g)) {
        if (after_edges) {
            CCP_TRY(edge_wait_on_stream(g, g->stream_comm));
        } else {
            CCP_HIP(hipEventRecord(g->ev_ready, g->stream));
            CCP_HIP(hipStreamWaitEvent(g->stream_comm, g->ev_ready, 0));
        }
        CCP_TRY(issue_exchange(g, g->stream_comm, base));
        CCP_HIP(hipEventRecord(g->ev_comm, g->stream_comm));
        CCP_HIP(hipStreamWaitEvent(g->stream, g->ev_comm, 0));
        g->exchanges++;
    }
    g->half_sweeps_since_refresh = 0;
    return CCP_OK;
}

// `iterations` sweeps with a halo exchange every ghost/2 iterations.
int sweep_rowblocked(ccp_grid *g, int iterations)
{
    const bool nb = has_neighbours(g);
    const int ipe = std::max(1, g->desc.ghost / 2);
    int left = iterations;
    while (left > 0) {
        int since = g->half_sweeps_since_refresh / 2;
        if (nb && since >= ipe) {
            CCP_TRY(exchange(g, false));
            since = 0;
        }
        const int room = nb ? std::min(left, ipe - since) : left;
        if (nb && g->overlap && since + room == ipe) {
            // these sweeps use up the ghost rows: their last pass hands the neighbours' rows over first and
            // the exchange runs beside the rest of it
            CCP_TRY(run_unchecked(g, room, nullptr, false, nullptr, g->desc.ghost));
            CCP_TRY(exchange(g, true));
        } else {
            CCP_TRY(run_unchecked(g, room));
        }
        left -= room;
    }
    return CCP_OK;
}

}  // namespace

int ccp_grid_attach_comm(ccp_grid *g, ccp_comm *c)
try {
    CCP_TRY(bind(g));
    if (!c) {                                        // detach
        g->comm = nullptr;
        g->up_rank = g->down_rank = -1;
        return CCP_OK;
    }
    if (c->device != g->device) return CCP_ERR_BAD_ARG;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    // every rank learns every block: the partition must be contiguous row blocks of one image, in rank order
    const int me[4] = {g->desc.row_begin, g->desc.row_count, g->desc.ghost,
                       g->desc.width ^ (g->desc.height << 1) ^ (g->desc.channels << 28) ^ ((g->desc.flags & CCP_GRID_DIRICHLET_MASK) << 27)};
    int *dev = reinterpret_cast<int *>(c->scratch.p);
    std::vector<int> all((size_t)4 * c->world);
    CCP_HIP(hipMemcpyAsync(dev + 4 * c->rank, me, sizeof(me), hipMemcpyHostToDevice, g->stream));
    CCP_RCCL(api->AllGather(dev + 4 * c->rank, dev, 4, ncclInt32, c->comm, g->stream));
    CCP_HIP(hipMemcpyAsync(all.data(), dev, sizeof(int) * all.size(), hipMemcpyDeviceToHost, g->stream));
    CCP_HIP(hipStreamSynchronize(g->stream));
    int next = 0;
    for (int r = 0; r < c->world; ++r) {
        if (all[4 * r] != next || all[4 * r + 2] != me[2] || all[4 * r + 3] != me[3]) return CCP_ERR_BAD_ARG;
        next += all[4 * r + 1];
    }
    if (next != g->desc.height) return CCP_ERR_BAD_ARG;
    g->up_rank = (c->rank > 0 && g->ghost_top > 0) ? c->rank - 1 : -1;
    g->down_rank = (c->rank + 1 < c->world && g->ghost_bottom > 0) ? c->rank + 1 : -1;
    if ((c->rank > 0) != (g->up_rank >= 0) || (c->rank + 1 < c->world) != (g->down_rank >= 0)) return CCP_ERR_STATE;   // a neighbour without ghost rows
    // what the neighbours' ghost zones take: their depth, clipped by the image border on their far side
    g->send_up = g->up_rank >= 0 ? std::min(g->desc.ghost, g->desc.height - g->desc.row_begin) : 0;
    g->send_down = g->down_rank >= 0 ? std::min(g->desc.ghost, g->desc.row_begin + g->desc.row_count) : 0;
    if (g->send_up > g->desc.row_count || g->send_down > g->desc.row_count) return CCP_ERR_UNSUPPORTED;   // block thinner than the ghost depth
    if (!g->stream_comm) {
        int lo = 0, hi = 0;                              // hi = greatest priority (numerically lowest)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        CCP_HIP(hipStreamCreateWithPriority(&g->stream_comm, hipStreamNonBlocking, hi));
        CCP_HIP(hipEventCreateWithFlags(&g->ev_comm, hipEventDisableTiming));
        CCP_HIP(hipEventCreateWithFlags(&g->ev_ready, hipEventDisableTiming));
    }
    g->comm = c;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_set_overlap(ccp_grid *g, int32_t on)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    g->overlap = on != 0;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_exchange_halos(ccp_grid *g)
try {
    CCP_TRY(bind(g));
    return exchange(g, false);
} CCP_ABI_CATCH

int ccp_grid_sweep_rowblocked(ccp_grid *g, int32_t iterations)
try {
    CCP_TRY(bind(g));
    if (iterations < 0) return CCP_ERR_BAD_ARG;
    if (!g->comm) return CCP_ERR_STATE;
    CCP_TRY(begin_timing(g));
    CCP_TRY(sweep_rowblocked(g, iterations));
    CCP_TRY(end_timing(g));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_residual_norm2_global(ccp_grid *g, double *rr_bb)
try {
    CCP_TRY(bind(g));
    if (!rr_bb) return CCP_ERR_BAD_ARG;
    if (!g->comm) return CCP_ERR_STATE;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    if (has_neighbours(g) && g->half_sweeps_since_refresh > 0) CCP_TRY(exchange(g, false));   // A x needs current ghost rows
    CCP_TRY(residual_to_small(g));
    CCP_RCCL(api->AllReduce(g->small.p, g->small.p, (size_t)2 * g->desc.channels, ncclDouble, ncclSum, g->comm->comm, g->stream));
    return small_to_rr_bb(g, rr_bb);
} CCP_ABI_CATCH

int ccp_grid_gauss_seidel_rowblocked(ccp_grid *g, double epsilon, int32_t max_iteration, int32_t check_every,
                                     ccp_gs_report *report)
try {
    CCP_TRY(bind(g));
    if (!g->comm) return CCP_ERR_STATE;
    if (max_iteration < 0 || check_every < 0) return CCP_ERR_BAD_ARG;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    const int C = g->desc.channels;
    const bool nb = has_neighbours(g);
    const int ipe = std::max(1, g->desc.ghost / 2);
    double eps[kMaxChannels];
    int stop_at[kMaxChannels];
    for (int ch = 0; ch < C; ++ch) {
        eps[ch] = 10.0;                                   // `double eps = 10` (sparse-matrix.h:354)
        stop_at[ch] = 0;
    }
    CCP_TRY(begin_timing(g));
    int cnt = 0;
    auto any_above = [&]() {
        for (int ch = 0; ch < C; ++ch)
            if (stop_at[ch] == 0 && eps[ch] > epsilon) return true;
        return false;
    };
    if (check_every == 0) {
        if (10.0 > epsilon) {
            CCP_TRY(sweep_rowblocked(g, max_iteration));
            cnt = max_iteration;
        }
    } else if (g->fuse && !(getenv("CCP_GS_ROWBLOCK_CHECKED_FUSED") && atoi(getenv("CCP_GS_ROWBLOCK_CHECKED_FUSED")) == 0)) {
        // The rule after every check_every-th sweep at the speed of the temporally blocked pass: a checked pass reports the
        // step of each of its sweeps (as ccp_grid_gauss_seidel does on one block); the blocks' sums are all-reduced and
        // every rank takes the same decisions.  A channel freezes at the sweep its rule fired at (re-run from the pass's
        // input for the missing sweeps when that sweep lies inside a pass), exactly as on one block.
        SolveState host{};
        for (int ch = 0; ch < C; ++ch) {
            host.active[ch] = 1;
            host.last_eps[ch] = 10.0;
        }
        CCP_HIP(hipMemcpyAsync(g->state.p, &host, sizeof(host), hipMemcpyHostToDevice, g->stream));
        const size_t elems = (size_t)g->geom.ch_stride * C;
        if (!g->x_alt.p) {
            CCP_TRY(g->x_alt.alloc(elems));
            CCP_HIP(hipMemsetAsync(g->x_alt.p, 0, elems * sizeof(double), g->stream));
        }
        if (!g->redo_mask.p) CCP_TRY(g->redo_mask.alloc(kMaxChannels));
        if (!g->sweep_sums.p) CCP_TRY(g->sweep_sums.alloc((size_t)kFusedMaxCheckedT * kMaxChannels));
        const int *active = reinterpret_cast<const int *>(g->state.p);
        const size_t plane = (size_t)g->geom.ch_stride * sizeof(double);
        const bool shrinking = g->shrink_top || g->shrink_bottom;
        const int max_checked = g->masked ? kMaskedMaxCheckedT : kFusedMaxCheckedT;
        double *cur = g->x.p, *alt = g->x_alt.p;
        int was_active[kMaxChannels];
        for (int ch = 0; ch < C; ++ch) was_active[ch] = 1;
        bool any_active = (10.0 > epsilon) && max_iteration > 0;
        int k0 = 0;
        // every rank enters with fresh ghost rows: the depth of a pass follows from the ghost rows left, and the ranks must
        // agree on it (the all-reduce below carries T * C values)
        if (any_active && nb) CCP_TRY(exchange(g, false, cur));
        while (any_active && k0 < max_iteration) {
            // the ghost rows left decide how deep the pass may be; none left: refresh them in the buffer the iterate is in
            if (nb && shrinking && g->half_sweeps_since_refresh + 2 > g->desc.ghost) CCP_TRY(exchange(g, false, cur));
            int T = std::min(max_checked, max_iteration - k0);
            if (shrinking) T = std::min(T, (g->desc.ghost - g->half_sweeps_since_refresh) / 2);
            if (T < 1) return CCP_ERR_STATE;
            const int since0 = g->half_sweeps_since_refresh;
            long blocks[2] = {0, 0};
            CCP_TRY(launch_fused(g, T, cur, alt, active, 2, blocks));
            hipLaunchKernelGGL(k_sweep_sums_wide, dim3((unsigned)C, (unsigned)T), dim3(kBlock), 0, g->stream, g->partial.p, blocks[0],
                               g->partial.p + g->partial_region, blocks[1], g->sweep_sums.p);
            CCP_HIP(hipGetLastError());
            CCP_RCCL(api->AllReduce(g->sweep_sums.p, g->sweep_sums.p, (size_t)T * C, ncclDouble, ncclSum, g->comm->comm, g->stream));
            hipLaunchKernelGGL(k_decide_sums, dim3(1), dim3(64), 0, g->stream, g->sweep_sums.p, C, T, k0 + 1, check_every, epsilon, g->state.p);
            CCP_HIP(hipGetLastError());
            CCP_HIP(hipMemcpyAsync(&host, g->state.p, sizeof(host), hipMemcpyDeviceToHost, g->stream));
            CCP_HIP(hipStreamSynchronize(g->stream));
            const int since1 = g->half_sweeps_since_refresh;
            any_active = false;
            for (int ch = 0; ch < C; ++ch) {
                any_active |= host.active[ch] != 0;
                if (!was_active[ch] || host.active[ch]) continue;
                was_active[ch] = 0;                                   // stopped inside this pass
                const int m = host.iterations[ch] - k0;               // sweeps of the pass it wanted: 1..T
                double *have = alt;
                if (m < T) {
                    int mask[kMaxChannels] = {0};
                    mask[ch] = 1;
                    CCP_HIP(hipMemcpyAsync(g->redo_mask.p, mask, sizeof(mask), hipMemcpyHostToDevice, g->stream));
                    g->half_sweeps_since_refresh = since0;           // the re-run starts where the pass started
                    double *p = cur, *q = alt;
                    for (int left = m; left > 0;) {
                        const int t = std::min(left, g->masked ? kMaskedMaxT : kFusedMaxT);
                        CCP_TRY(launch_fused(g, t, p, q, g->redo_mask.p));
                        std::swap(p, q);
                        left -= t;
                    }
                    CCP_HIP(hipStreamSynchronize(g->stream));          // `mask` lives on this stack frame
                    g->half_sweeps_since_refresh = since1;
                    have = p;
                }
                // a frozen channel is never touched again: keep its result in BOTH buffers
                double *other = (have == cur) ? alt : cur;
                CCP_HIP(hipMemcpyAsync(other + (size_t)ch * g->geom.ch_stride, have + (size_t)ch * g->geom.ch_stride, plane,
                                       hipMemcpyDeviceToDevice, g->stream));
            }
            k0 += T;
            std::swap(cur, alt);
        }
        if (cur != g->x.p) CCP_HIP(hipMemcpyAsync(g->x.p, cur, elems * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        cnt = k0;
        for (int ch = 0; ch < C; ++ch) {
            eps[ch] = host.last_eps[ch];
            stop_at[ch] = host.converged[ch] ? host.iterations[ch] : 0;
        }
    } else {
        // `while (eps > epsilon && cnt < max_iteration)` (sparse-matrix.h:356) with the step summed over all blocks
        while (any_above() && cnt < max_iteration) {
            const int plain = std::min(check_every - 1, max_iteration - cnt);
            if (plain > 0) {
                CCP_TRY(sweep_rowblocked(g, plain));
                cnt += plain;
            }
            if (cnt >= max_iteration) break;
            if (nb && g->half_sweeps_since_refresh / 2 >= ipe) CCP_TRY(exchange(g, false));
            long blocks[2] = {0, 0};
            CCP_TRY(one_iteration(g, true, nullptr, blocks));
            ++cnt;
            hipLaunchKernelGGL(k_check, dim3((unsigned)C), dim3(kBlock), 0, g->stream, g->partial.p, blocks[0],
                               g->partial.p + g->partial_region, blocks[1], 0.0, 0, static_cast<SolveState *>(nullptr), g->small.p);
            CCP_HIP(hipGetLastError());
            CCP_RCCL(api->AllReduce(g->small.p, g->small.p, (size_t)C, ncclDouble, ncclSum, g->comm->comm, g->stream));
            double host[kMaxChannels];
            CCP_HIP(hipMemcpyAsync(host, g->small.p, sizeof(double) * C, hipMemcpyDeviceToHost, g->stream));
            CCP_HIP(hipStreamSynchronize(g->stream));
            for (int ch = 0; ch < C; ++ch) {
                if (stop_at[ch]) continue;
                eps[ch] = host[ch];
                if (!(eps[ch] > epsilon)) stop_at[ch] = cnt;
            }
            if (getenv("CCP_GS_DEBUG")) fprintf(stderr, "[ccp_gs] rowblocked sweep %d: step %.17g (channel 0), epsilon %.17g\n", cnt, eps[0], epsilon);
        }
    }
    CCP_TRY(end_timing(g));
    CCP_HIP(hipStreamSynchronize(g->stream));
    float ms = 0.f;
    CCP_HIP(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    g->last_ms = ms;
    g->timing_pending = false;
    if (report) {
        for (int ch = 0; ch < C; ++ch) {
            report[ch].converged = stop_at[ch] ? 1 : 0;
            report[ch].iterations = stop_at[ch] ? stop_at[ch] : cnt;
            report[ch].last_l1_step = eps[ch];
            report[ch].seconds = ms * 1e-3;
        }
    }
    return CCP_OK;
} CCP_ABI_CATCH

// conjugateGradient (sparse-matrix.h:396-434) on a row block: the three-vector loop of ccp_cg.hpp on the OWNED rows (one
// contiguous range of the planes), A applied to the owned rows with the direction's row above and below them fetched from
// the neighbours before every product (one image row per neighbour), every dot product all-reduced.
int ccp_grid_conjugate_gradient_rowblocked(ccp_grid *g, double epsilon, int32_t max_iteration, ccp_gs_report *report)
try {
    CCP_TRY(bind(g));
    if (!g->comm) return CCP_ERR_STATE;
    if (max_iteration < 0) return CCP_ERR_BAD_ARG;
    const RcclApi *api = rccl_api();
    if (!api) return CCP_ERR_RCCL;
    const Geom &geo = g->geom;
    if ((g->up_rank >= 0 && g->ghost_top < 1) || (g->down_rank >= 0 && g->ghost_bottom < 1)) return CCP_ERR_STATE;
    const long n = geo.ch_stride;
    const size_t row = (size_t)2 * geo.pitch;
    const long off = (long)geo.own_lo * (long)row, n_own = (long)(geo.own_hi - geo.own_lo) * (long)row;
    if (!g->cg_r.p) {
        CCP_TRY(g->cg_r.alloc((size_t)n));
        CCP_TRY(g->cg_p.alloc((size_t)n));
        CCP_TRY(g->cg_p2.alloc((size_t)n));
        CCP_TRY(g->cg_ap.alloc((size_t)n));
        CCP_TRY(g->cg_state.alloc(1));
    }
    hipStream_t s = g->stream;
    dim3 grid((unsigned)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)), (unsigned)(geo.own_hi - geo.own_lo), 2);
    // one row of `plane` to each neighbour's ghost row next to its owned rows, theirs into ours
    auto fetch_rows = [&](double *plane) -> int {
        if (g->up_rank < 0 && g->down_rank < 0) return CCP_OK;
        CCP_RCCL(api->GroupStart());
        ncclResult_t r = ncclSuccess;
        if (g->up_rank >= 0) {
            r = api->Send(plane + (size_t)geo.own_lo * row, row, ncclDouble, g->up_rank, g->comm->comm, s);
            if (r == ncclSuccess) r = api->Recv(plane + (size_t)(geo.own_lo - 1) * row, row, ncclDouble, g->up_rank, g->comm->comm, s);
        }
        if (g->down_rank >= 0 && r == ncclSuccess) {
            r = api->Send(plane + (size_t)(geo.own_hi - 1) * row, row, ncclDouble, g->down_rank, g->comm->comm, s);
            if (r == ncclSuccess) r = api->Recv(plane + (size_t)geo.own_hi * row, row, ncclDouble, g->down_rank, g->comm->comm, s);
        }
        const ncclResult_t e = api->GroupEnd();
        if (r != ncclSuccess) return rccl_fail(r, "ncclSend/ncclRecv", __FILE__, __LINE__);
        CCP_RCCL(e);
        g->exchanges++;
        return CCP_OK;
    };
    double *total = reinterpret_cast<double *>(g->comm->scratch.p);
    auto sums = [&](double *partial, int *count, const double **sum_at, int slot) -> int {
        hipLaunchKernelGGL(k_reduce_to_one, dim3(1), dim3(kBlock), 0, s, partial, (long)*count, total + slot);
        CCP_HIP(hipGetLastError());
        CCP_RCCL(api->AllReduce(total + slot, total + slot, 1, ncclDouble, ncclSum, g->comm->comm, s));
        *count = 1;
        *sum_at = total + slot;
        return CCP_OK;
    };
    for (int ch = 0; ch < g->desc.channels; ++ch) {
        CCP_HIP(hipMemsetAsync(g->cg_r.p, 0, sizeof(double) * n, s));
        CCP_HIP(hipMemsetAsync(g->cg_p.p, 0, sizeof(double) * n, s));
        CCP_HIP(hipMemsetAsync(g->cg_ap.p, 0, sizeof(double) * n, s));
        // the loop sees the owned range of every vector; the product goes back to the planes' bases
        auto spmv = [&](const double *in_own, double *out_own) -> int {
            double *in = const_cast<double *>(in_own) - off;
            double *out = out_own - off;
            CCP_TRY(fetch_rows(in));
            if (g->masked) hipLaunchKernelGGL((k_apply<2, 0, true>), grid, dim3(kBlock), 0, s, in, out, out, geo, geo.own_lo, g->partial.p, g->maskp.p);
            else hipLaunchKernelGGL((k_apply<2, 0>), grid, dim3(kBlock), 0, s, in, out, out, geo, geo.own_lo, g->partial.p, static_cast<const unsigned char *>(nullptr));
            return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
        };
        auto spmv_dot = [&](const double *in_own, double *out_own, int *n_partials) -> int {
            *n_partials = 0;                             // (the loop runs its own dot pass over the owned range)
            return spmv(in_own, out_own);
        };
        const int fused_mode = getenv("CCP_GS_CG_FUSED") ? atoi(getenv("CCP_GS_CG_FUSED")) : 1;
        if (fused_mode == 0) {
            CCP_TRY(cg_solve(spmv, spmv_dot, g->b.p + (long)ch * n + off, g->x.p + (long)ch * n + off, g->cg_r.p + off, g->cg_p.p + off,
                             g->cg_ap.p + off, n_own, epsilon, max_iteration, g->cg_state.p, g->partial.p, s, g->ev0, g->ev1,
                             report ? report + ch : nullptr, sums, true));
            continue;
        }
        // the fused loop (72 B per unknown and iteration, ccp_cg.hpp): pass A recomputes the direction of a row's
        // neighbours from r and the previous direction, so BOTH travel — one row of each per neighbour and iteration
        CCP_HIP(hipMemsetAsync(g->cg_p2.p, 0, sizeof(double) * n, s));
        const int march_rows = 32;
        const int own_rows = geo.own_hi - geo.own_lo;
        const dim3 mgrid((unsigned)((geo.pitch + 2L * kBlock - 1) / (2L * kBlock)), (unsigned)((own_rows + march_rows - 1) / march_rows));
        auto apply = [&](double *xv_own, const double *rv_own, const double *p_in_own, double *p_out_own, double *apv_own, int *n_partials) -> int {
            double *rv = const_cast<double *>(rv_own) - off, *p_in = const_cast<double *>(p_in_own) - off;
            CCP_TRY(fetch_rows(rv));
            CCP_TRY(fetch_rows(p_in));
            if (g->masked)
                hipLaunchKernelGGL((k_cg_apply_march<true>), mgrid, dim3(kBlock), 0, s, xv_own - off, rv, p_in, p_out_own - off, apv_own - off, geo, march_rows,
                                   g->partial.p, g->maskp.p, g->cg_state.p, geo.own_lo, geo.own_hi);
            else
                hipLaunchKernelGGL((k_cg_apply_march<false>), mgrid, dim3(kBlock), 0, s, xv_own - off, rv, p_in, p_out_own - off, apv_own - off, geo, march_rows,
                                   g->partial.p, static_cast<const unsigned char *>(nullptr), g->cg_state.p, geo.own_lo, geo.own_hi);
            *n_partials = (int)(mgrid.x * mgrid.y);
            return hipGetLastError() == hipSuccess ? CCP_OK : CCP_ERR_HIP;
        };
        CCP_TRY(cg_solve_fused(spmv, apply, g->b.p + (long)ch * n + off, g->x.p + (long)ch * n + off, g->cg_r.p + off, g->cg_p.p + off,
                               g->cg_p2.p + off, g->cg_ap.p + off, n_own, epsilon, max_iteration, g->cg_state.p, g->partial.p, s, g->ev0,
                               g->ev1, report ? report + ch : nullptr, sums, true));
    }
    // the ghost rows of x are stale now: the next sweep needs an exchange first
    g->half_sweeps_since_refresh = g->desc.ghost;
    return edge_timeout_status(g);
} CCP_ABI_CATCH

int ccp_grid_comm_stats(ccp_grid *g, int64_t *exchanges, int32_t *wait_mode, int32_t *send_up_rows, int32_t *send_down_rows)
try {
    if (!g) return CCP_ERR_BAD_ARG;
    if (exchanges) *exchanges = g->exchanges;
    if (wait_mode) *wait_mode = g->edge_flag ? g->wait_mode : -1;
    if (send_up_rows) *send_up_rows = g->send_up;
    if (send_down_rows) *send_down_rows = g->send_down;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_region_begin(ccp_grid *g)
try {
    CCP_TRY(bind(g));
    if (!g->ev_r0) {
        CCP_HIP(hipEventCreate(&g->ev_r0));
        CCP_HIP(hipEventCreate(&g->ev_r1));
    }
    g->region_launches = 0;
    g->region_iterations = 0;
    CCP_HIP(hipEventRecord(g->ev_r0, g->stream));
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_region_end(ccp_grid *g, float *milliseconds, int64_t *sweep_launches, int64_t *pass_iterations)
try {
    CCP_TRY(bind(g));
    if (!g->ev_r0) return CCP_ERR_STATE;
    CCP_HIP(hipEventRecord(g->ev_r1, g->stream));
    CCP_HIP(hipEventSynchronize(g->ev_r1));
    float ms = 0.f;
    CCP_HIP(hipEventElapsedTime(&ms, g->ev_r0, g->ev_r1));
    if (milliseconds) *milliseconds = ms;
    if (sweep_launches) *sweep_launches = g->region_launches;
    if (pass_iterations) *pass_iterations = g->region_iterations;
    return CCP_OK;
} CCP_ABI_CATCH

int ccp_grid_last_timing(ccp_grid *g, float *milliseconds, int32_t *kernel_launches)
try {
    CCP_TRY(bind(g));
    if (g->timing_pending) {
        CCP_HIP(hipEventSynchronize(g->ev1));
        float ms = 0.f;
        CCP_HIP(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        g->last_ms = ms;
        g->timing_pending = false;
    }
    if (milliseconds) *milliseconds = g->last_ms;
    if (kernel_launches) *kernel_launches = g->last_launches;
    return CCP_OK;
} CCP_ABI_CATCH

}  // extern "C"
