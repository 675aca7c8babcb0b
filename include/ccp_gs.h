/*
 * ccp_gs.h — C ABI of libccp_gs.so: MI355X (gfx950) Gauss-Seidel / SpMV / Poisson-assembly
 * path of linwe2012/CourseComputationalPhotography.
 *
 * The reference has no FFI; its boundary for this path is the member-function surface of the
 * header-only template `SparseMatrix<T,IndexType>` plus `PhotoMontage::SolveChannel`
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces
 * (paths relative to the reference repo root).  The C++ facade `include/ccp/sparse-matrix.h`
 * keeps the reference's own names and signatures on top of this ABI; INTEGRATION.md shows the
 * binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns a ccp_status (0 = ok) and never throws;
 *   - "host" pointers are caller-owned host buffers (never retained);
 *     "dev" pointers are device addresses on the handle's GPU;
 *   - one handle per host thread; a handle is bound to one HIP device and one stream;
 *   - values are IEEE fp64, indices int32 (as the reference: SparseMatrix<double,int>);
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point
 *     returns CCP_ERR_NO_DEVICE.
 */
#ifndef CCP_GS_H
#define CCP_GS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCP_GS_ABI_VERSION 6

typedef enum ccp_status {
    CCP_OK = 0,
    CCP_ERR_BAD_ARG = 1,      /* null pointer, negative size, inconsistent arrays            */
    CCP_ERR_NO_DEVICE = 2,    /* no HIP device / hipSetDevice failed                         */
    CCP_ERR_HIP = 3,          /* a HIP runtime call or kernel launch failed                  */
    CCP_ERR_ALLOC = 4,        /* host or device allocation failed                            */
    CCP_ERR_STATE = 5,        /* call sequence error (e.g. solve before upload)              */
    CCP_ERR_UNSUPPORTED = 6,  /* e.g. colouring that is not a proper colouring of the matrix */
    CCP_ERR_RCCL = 7          /* RCCL could not be loaded, or a communicator / collective call failed */
} ccp_status;

/* Sweep ordering of the Gauss-Seidel solve.
 *  LEXICOGRAPHIC: the reference's row-index order (sparse-matrix.h:359-374), executed on the
 *                 GPU by level scheduling; bit-identical to the reference iterates, slow.
 *  MULTICOLOUR  : rows grouped by colour, colours swept one after another (red-black for the
 *                 5-point grid).  Identical to the reference gaussSeidel applied to P A P^T
 *                 with the rows listed colour by colour.  The fast path. */
typedef enum ccp_ordering {
    CCP_ORDER_LEXICOGRAPHIC = 0,
    CCP_ORDER_MULTICOLOUR = 1
} ccp_ordering;

/* What a solve reports (the reference returns none of this; SURVEY.md §5 "metrics"). */
typedef struct ccp_gs_report {
    int32_t iterations;       /* sweeps executed (the reference's `cnt`, sparse-matrix.h:355) */
    int32_t converged;        /* 1 if the L1-step rule stopped the loop before max_iteration  */
    double  last_l1_step;     /* sum|x_k - x_{k-1}| of the last CHECKED sweep (`eps`, :376)   */
    double  seconds;          /* device time of the sweep loop (HIP events), excl. transfers  */
} ccp_gs_report;

const char *ccp_status_string(int status);
int ccp_abi_version(void);
/* Number of visible HIP devices (0 when there is none); never fails. */
int ccp_device_count(void);

/* ========================================================================================
 * 1. General slack-CSR matrix  —  replaces SparseMatrix<double,int> storage + solvers
 *    (project/src/PhotoMontage/sparse-matrix.h:107-121, 670-676).
 * ====================================================================================== */
typedef struct ccp_csr ccp_csr;

int ccp_csr_create(int device, ccp_csr **out);
int ccp_csr_destroy(ccp_csr *m);

/* Hand the five reference arrays to the device.  Replaces the storage hand-off that
 * initializeFromEigenRowMajor / initializeFromVector end in (sparse-matrix.h:537-620,
 * 265-319): values_/col_offset_ have n_values entries, row_begin_/row_num_nze_ have n_rows
 * entries (row_begin is NOT n_rows+1 long); live entries of a row are sorted by column. */
int ccp_csr_upload(ccp_csr *m, int32_t n_rows, int32_t n_cols, int64_t n_values,
                   const double *values, const int32_t *col_offset,
                   const int32_t *row_begin, const int32_t *row_num_nze);

/* Optional colouring for CCP_ORDER_MULTICOLOUR: colour[i] in [0,n_colours), rows of one
 * colour must not reference each other (checked: CCP_ERR_UNSUPPORTED otherwise).
 * Without it the library colours the rows itself: a matrix recognised as the Laplacian of a raster region takes
 * the parity of its reconstructed pixel coordinates (a proper 2-colouring however the pieces of the region merge, and
 * the sweep runs on the region grid); any other matrix is coloured greedily in row order.  ccp_csr_get_colouring
 * exports whichever it is. */
int ccp_csr_set_colouring(ccp_csr *m, const int32_t *colour, int32_t n_colours);

/* Which kernels the last ccp_csr_gauss_seidel ran on.  The general path stores the matrix (sliced ELL);
 * two matrix shapes are recognised at the first solve and swept matrix-free with identical results:
 * SolveChannel's W x H Poisson matrix (PhotoMontage.cpp:541-597) and — multi-colour order — the 5-point
 * Laplacian of a raster REGION with zero Dirichlet values around it (diagonal 4, -1 to the 4-neighbours
 * inside, unknowns in raster order: a blend restricted to a brush / label region), whose pixel coordinates
 * are reconstructed from the couplings and verified against every row.  canvas_*: the grid it ran on;
 * sweep_launches: kernel passes of the last solve on the region grid (each several iterations deep). */
#define CCP_PATH_SLICED_ELL 0
#define CCP_PATH_POISSON_GRID 1
#define CCP_PATH_REGION_GRID 2
int ccp_csr_last_path(ccp_csr *m, int32_t *path, int32_t *canvas_width, int32_t *canvas_height, int64_t *sweep_launches);
/* The recognition alone, on the host (no device needed; diagnostics and tests): compressed CSR (row_offset has
 * n+1 entries) plus a 2-colouring in, *recognised and — when 1 — the canvas size and the pixel coordinates of
 * every unknown out (outputs after `recognised` may be NULL). */
int ccp_csr_embed_region_host(int32_t n, const int32_t *row_offset, const int32_t *col, const double *val, const int32_t *colour,
                              int32_t *recognised, int32_t *canvas_width, int32_t *canvas_height, int32_t *x_out, int32_t *y_out);

/* SparseMatrix::insert(val, row, col) (sparse-matrix.h:183-247) on the uploaded matrix: val == 0 removes the
 * entry (insertZero: it becomes slack), an existing entry is overwritten, a new one is inserted in column
 * order (insertNoneZero).  The edit is applied to the device images of the matrix incrementally before the
 * next solve: the rows touched are re-laid in their slices by one small kernel (a slice has spare entry
 * columns; one that outgrows them moves to a reserve at the end of the arrays) — no re-upload and no new
 * schedule.  Only a new coupling that breaks the ordering a schedule promises (two rows of one colour, a
 * level out of order) drops THAT image, which is rebuilt at the next solve that needs it.  row/col must be
 * inside the uploaded shape.  ccp_csr_edit_stats: edits received, whole images built and uploaded, rows
 * patched, slices relocated, images dropped for a rebuild (outputs may be NULL). */
int ccp_csr_insert(ccp_csr *m, int32_t row, int32_t col, double val);
/* The same for a batch of edits, applied in order (a brush stroke; initializeFromTriplets, :256-263). */
int ccp_csr_insert_many(ccp_csr *m, int64_t count, const int32_t *rows, const int32_t *cols, const double *vals);
int ccp_csr_edit_stats(ccp_csr *m, int64_t *edits, int64_t *image_uploads, int64_t *rows_patched, int64_t *slices_relocated,
                       int64_t *image_rebuilds);
/* Device memory the handle holds for a copy of the STORED matrix (row offsets, columns, values; bytes): ccp_csr_upload
 * starts copying the structure in the background for the recognition of the first solve — best effort: when the device
 * has no room the upload still succeeds (eager_copies_skipped counts that) and the copy is made when something needs it —
 * and the copy is given back (0 bytes) once a grid twin sweeps the matrix.  Waits for the background copy.  Outputs may
 * be NULL. */
int ccp_csr_device_footprint(ccp_csr *m, int64_t *stored_matrix_bytes, int64_t *eager_copies_skipped);

/* The colouring the multi-colour sweep uses (the caller's, or the library's greedy one): colour[i]
 * for every row (n_rows entries; may be NULL to ask for the count only).  With it a caller can hand
 * the reference gaussSeidel the same permuted matrix P A P^T and compare iterate for iterate. */
int ccp_csr_get_colouring(ccp_csr *m, int32_t *colour, int32_t *n_colours);

/* SparseMatrix::gaussSeidel(b, epsilon, max_iteration) (sparse-matrix.h:350-380).
 * x0 == NULL starts from all-ones as the reference does (:352); a non-NULL x0 is the `init`
 * extension mirroring conjugateGradient's 4th argument (:396).  b, x_out: n_cols entries.
 * check_every: the L1-step stop rule (:356,376) is evaluated every check_every-th sweep
 * (1 = the reference's behaviour; 0 = never: exactly max_iteration sweeps).
 * With CCP_ORDER_MULTICOLOUR a matrix that is exactly the W x H Poisson matrix SolveChannel
 * assembles (PhotoMontage.cpp:541-597) — and whose colouring, if given, is (x+y)&1 — is recognised
 * at the first solve and swept by the matrix-free grid kernels: identical bits, far faster. */
int ccp_csr_gauss_seidel(ccp_csr *m, const double *b, const double *x0, double *x_out,
                         double epsilon, int32_t max_iteration, int32_t check_every,
                         int32_t ordering, ccp_gs_report *report);

/* SparseMatrix::conjugateGradient(b, epsilon, max_iteration, initialize) (sparse-matrix.h:396-434)
 * — the solver the blend call sites use today (PhotoMontage.cpp:613, hw8_pa.cc:972).
 * init == NULL starts from 0 (:397).  Stops when sqrt(r'r) < epsilon (:425) or after
 * max_iteration iterations.  report->last_l1_step carries sqrt(r'r) of the last update.
 * Reductions are deterministic but tree-ordered: iterates match the reference to rounding. */
int ccp_csr_conjugate_gradient(ccp_csr *m, const double *b, const double *init, double *x_out,
                               double epsilon, int32_t max_iteration, ccp_gs_report *report);

/* SparseMatrix::conjugateGradientEigen(b, epsilon, max_iteration) (sparse-matrix.h:494-535; RunTest,
 * utils.cc:99): Jacobi-preconditioned conjugate gradient from x0 = 0 with extractDiagnolColInv()
 * (:472-491) as the preconditioner.  report as for ccp_csr_conjugate_gradient. */
int ccp_csr_conjugate_gradient_jacobi(ccp_csr *m, const double *b, double *x_out, double epsilon,
                                      int32_t max_iteration, ccp_gs_report *report);

/* SparseMatrix::applyToVector(in, out) (sparse-matrix.h:382-393). in: n_cols, out: n_rows.  SolveChannel's matrix is
 * applied matrix-free on the grid kernels (the row's products in the stored order: same bits; no image of the matrix is
 * built for it); so is its residual below. */
int ccp_csr_apply_to_vector(ccp_csr *m, const double *in, double *out);

/* sum (b - A x)^2 and sum b^2 (applyToVector + vecsub + veclen2, sparse-matrix.h:51-55,75-79). */
int ccp_csr_residual_norm2(ccp_csr *m, const double *b, const double *x, double *rr, double *bb);

/* ----------------------------------------------------------------------------------------
 * Row block of a matrix distributed over the ranks of a communicator (section 4: ccp_comm) — SURVEY.md §8e,
 * BASELINE configs[4] on several GPUs: "row-block by unknown index with a halo index list".  No reference
 * counterpart (the reference is one process, sparse-matrix.h:350-380); the sweep each rank runs is the
 * multi-colour ccp_csr_gauss_seidel above, and the iterates are the one-GPU iterates bit for bit.
 *
 * ccp_csr_upload_rows — COLLECTIVE over `comm`.  This rank owns rows [first_row, first_row + n_rows) of an
 * n_global x n_global matrix; the blocks of ranks 0, 1, ... follow one another and cover it.  The five arrays
 * are the reference's for those rows (row_begin / row_num_nze have n_rows entries), column indices are
 * GLOBAL.  colour[i] in [0, n_colours) for the owned rows: a proper colouring of the whole matrix (two coupled
 * rows never share a colour — checked on every rank against its own rows, CCP_ERR_UNSUPPORTED otherwise), the
 * same n_colours everywhere.  The ranks exchange their halo index lists (the columns a block references outside
 * itself) and the colours of those rows once, here.  If any rank refuses its arguments every rank returns an
 * error (its own, or CCP_ERR_STATE for "a peer failed") and the handle holds no matrix.
 * Afterwards, with b / x0 / x_out / in / out holding the block's OWN rows (n_rows entries):
 *   ccp_csr_gauss_seidel(CCP_ORDER_MULTICOLOUR)  collective: after every colour the new values other blocks
 *        reference travel to them (one ncclSend/ncclRecv pair per neighbouring block, all in one group); the
 *        stop rule (:376) uses the all-reduced step sum, so every rank stops at the same sweep;
 *   ccp_csr_apply_to_vector, ccp_csr_residual_norm2   collective (rr, bb: sums over the whole matrix);
 *   ccp_csr_conjugate_gradient, ccp_csr_conjugate_gradient_jacobi   collective: the ghosts of the direction travel
 *        before every product, every dot product is all-reduced (iterates equal to the one-GPU loop's to rounding, the
 *        same stop iteration on every rank);
 *   ccp_csr_get_colouring   the block's own rows.
 * The reference-order sweep, ccp_csr_insert and ccp_csr_set_colouring return CCP_ERR_UNSUPPORTED on a row block.  ccp_csr_upload returns the handle to the one-GPU form.
 * `comm` must stay alive for as long as the handle is used as a row block.  What the ranks agree on are refused
 * ARGUMENTS; a HIP or RCCL failure in the middle of a collective call (CCP_ERR_HIP / CCP_ERR_RCCL) is local to the rank
 * it happens on — treat it as fatal for the communicator, as with any RCCL error.
 * The sweep overlaps the messages with arithmetic: per colour the 64-row slices that hold a row some peer
 * references are swept first and their values travel on a second stream while the other slices of the colour
 * are swept (off when those slices are more than a quarter of the block, or with CCP_GS_ROWS_OVERLAP=0).
 * ccp_csr_rows_info: the block, its ghosts (columns owned by peers), the peers it exchanges with, the slices swept
 * first (0: no overlap), and the values sent / grouped exchanges issued since the upload (outputs may be NULL). */
typedef struct ccp_comm ccp_comm;
int ccp_csr_upload_rows(ccp_csr *m, ccp_comm *comm, int32_t first_row, int32_t n_rows, int32_t n_global, int64_t n_values,
                        const double *values, const int32_t *col_offset, const int32_t *row_begin, const int32_t *row_num_nze,
                        const int32_t *colour, int32_t n_colours);
int ccp_csr_rows_info(ccp_csr *m, int32_t *first_row, int32_t *n_rows, int32_t *n_ghost, int32_t *n_peers, int32_t *edge_slices,
                      int64_t *values_sent, int64_t *exchanges);

/* ========================================================================================
 * 2. Structured Poisson grid  —  matrix-free form of the system SolveChannel assembles
 *    (project/src/PhotoMontage/PhotoMontage.cpp:541-597; labs/lab8/.../hw8_pa.cc:911-967).
 *    The matrix is never stored: degree and neighbours follow from (x,y).
 *    A grid handle may own only a block of image rows [row_begin, row_begin+row_count) plus
 *    `ghost` rows on each side that has a neighbour block (row-blocked multi-GPU).
 * ====================================================================================== */
typedef struct ccp_grid ccp_grid;

typedef struct ccp_grid_desc {
    int32_t width;        /* W: unknowns per image row                                       */
    int32_t height;       /* H: image rows of the WHOLE system                               */
    int32_t channels;     /* right-hand sides sharing the matrix (3 for a BGR blend)         */
    int32_t row_begin;    /* first image row this handle owns (0 on a single GPU)            */
    int32_t row_count;    /* rows owned (H on a single GPU)                                  */
    int32_t ghost;        /* ghost rows kept above/below where another block exists          */
    int32_t device;       /* HIP device ordinal                                              */
    int32_t flags;        /* CCP_GRID_* bits                                                 */
} ccp_grid_desc;

/* flags: a Dirichlet-mask grid.  Instead of SolveChannel's matrix the handle carries the 5-point Laplacian of
 * an arbitrary pixel REGION of the W x H canvas: diagonal 4, -1 to every 4-neighbour inside the region,
 * nothing outside (zero Dirichlet values around it) — the matrix a gradient-domain blend restricted to a
 * brush / label region solves (BASELINE configs[4]).  The region is one byte per pixel (ccp_grid_set_mask_host);
 * pixels outside it are fixed at 0 in x and b.  Sweeps (red-black, fixed count or stop rule), b := A x, the
 * residual and conjugate gradient honour the mask, and so does the sweep in the reference's order (raster order of
 * the region); SolveChannel's assembly entry points return CCP_ERR_UNSUPPORTED.  Row blocks work as for the plain
 * grid (ghost rows, ccp_grid_attach_comm, ccp_grid_sweep_rowblocked, ...: the region of BASELINE configs[4] on
 * several GPUs, every block handed the whole mask and keeping its own rows of it).  ccp_csr_gauss_seidel reaches
 * this form by itself when the uploaded matrix is such a Laplacian of a raster region (ccp_csr_last_path). */
#define CCP_GRID_DIRICHLET_MASK 1

/* Device layout, for callers that move halos themselves (torch.distributed / RCCL).
 * Element (channel ch, local row l, colour c, half-column j) of x or b lives at
 *   base + (((ch*local_rows + l)*2 + c)*pitch + j) * 8 bytes,
 * local row l = image row (row_begin - ghost_top + l), colour c = (x+y)&1, j = x>>1.
 * One image row of one channel is therefore 2*pitch contiguous doubles. */
typedef struct ccp_grid_layout {
    void   *x_dev;
    void   *b_dev;
    int64_t pitch;        /* doubles per colour half-row (>= ceil(W/2), multiple of 16)      */
    int32_t local_rows;   /* ghost_top + row_count + ghost_bottom                            */
    int32_t ghost_top;
    int32_t ghost_bottom;
    int32_t channels;
} ccp_grid_layout;

int ccp_grid_create(const ccp_grid_desc *desc, ccp_grid **out);
int ccp_grid_destroy(ccp_grid *g);
int ccp_grid_get_layout(ccp_grid *g, ccp_grid_layout *out);
/* All device work of this handle is enqueued on `hip_stream` (a hipStream_t; NULL = the
 * null stream). */
int ccp_grid_set_stream(ccp_grid *g, void *hip_stream);
int ccp_grid_synchronize(ccp_grid *g);

/* Host hand-off in natural raster order (what the reference's std::vector<double> holds):
 * `n_rows` image rows starting at image row `first_row`, W doubles each.  Rows may cover the
 * ghosts.  These calls synchronise the stream. */
int ccp_grid_set_b_host(ccp_grid *g, int32_t channel, const double *rows, int32_t first_row, int32_t n_rows);
int ccp_grid_set_x_host(ccp_grid *g, int32_t channel, const double *rows, int32_t first_row, int32_t n_rows);
int ccp_grid_get_x_host(ccp_grid *g, int32_t channel, double *rows, int32_t first_row, int32_t n_rows);
int ccp_grid_get_b_host(ccp_grid *g, int32_t channel, double *rows, int32_t first_row, int32_t n_rows);
/* Dirichlet-mask grids: the region, H x W bytes (non-zero = unknown), row stride in bytes.  Synchronises. */
int ccp_grid_set_mask_host(ccp_grid *g, const uint8_t *mask, int64_t row_stride_bytes);
/* x := value everywhere (the reference start vector is 1.0, sparse-matrix.h:352). Async. */
int ccp_grid_fill_x(ccp_grid *g, double value);
/* b := A x (applyToVector order, sparse-matrix.h:382-393) on every local row whose neighbour
 * rows are local too (owned rows and all ghost rows but the outermost one, whose b is never
 * used); needs valid ghost rows of x.  Used to build b = A x_true on device. Async. */
int ccp_grid_b_from_x(ccp_grid *g);
/* x := uniform [lo,hi) pseudo-random field depending only on (seed, channel, image x, y):
 * identical across any row partition.  Fills owned + ghost rows.  Async. */
int ccp_grid_randomize_x(ccp_grid *g, uint64_t seed, double lo, double hi);

/* `iterations` red-black Gauss-Seidel sweeps (red = colour 0 first) over all local rows that
 * have their neighbours locally; no convergence check, no host synchronisation. Async.
 * With ghosts: each sweep invalidates two more ghost rows, so at most ghost/2 iterations may
 * run between two halo refreshes (enforced: CCP_ERR_STATE). */
int ccp_grid_sweep(ccp_grid *g, int32_t iterations);

/* The same sweeps for a row block with neighbour blocks whose halo exchange the CALLER performs (SURVEY
 * §8e; the RCCL path inside the library is ccp_grid_sweep_rowblocked): the LAST pass finalises the
 * `edge_rows` owned rows next to each neighbour first — short edge chunks dispatched first inside the one
 * launch, whose waves count themselves and publish an epoch in a signal-memory flag — so the exchange can
 * overlap the rest of the pass.  Results are identical to ccp_grid_sweep.  ccp_grid_stream_wait_edges makes
 * `hip_stream` (the stream the caller's send/receive is issued on) wait until those rows are final
 * (hipStreamWaitValue64, or a one-wave polling kernel where that is unsupported); the caller must make the
 * handle's stream wait for its receives before the next sweep.  A block without ghost rows: plain
 * ccp_grid_sweep / no-op. */
int ccp_grid_sweep_edges_first(ccp_grid *g, int32_t iterations, int32_t edge_rows);
int ccp_grid_stream_wait_edges(ccp_grid *g, void *hip_stream);
/* Pick the temporal-blocking depth (iterations fused per kernel pass, <= max_t) and the rows a
 * wave finalises per pass by timing the candidates on this handle's actual shape (a few dozen
 * launches into the scratch buffer; x, b and the ghost bookkeeping are left untouched).  Results
 * never depend on the choice — only speed does.  Outputs may be NULL.  Synchronises. */
int ccp_grid_tune(ccp_grid *g, int32_t max_t, int32_t *chosen_t, int32_t *chosen_rows_per_chunk,
                  float *ms_per_iteration);
/* Unchecked sweeps through the temporally blocked pass (on, the default) or the in-place half-sweep
 * kernels (off): two independent implementations of the same arithmetic — results are bit-identical,
 * which is what bench.py's in-run parity check compares. */
int ccp_grid_set_fused(ccp_grid *g, int32_t on);
/* Fix the temporal-blocking depth limit and the rows a wave finalises per pass instead of tuning
 * (discards a tuning table); ccp_grid_get_tiling reports what the next sweep will use at depth max_t. */
int ccp_grid_set_tiling(ccp_grid *g, int32_t max_t, int32_t rows_per_chunk);
int ccp_grid_get_tiling(ccp_grid *g, int32_t *max_t, int32_t *rows_per_chunk, int32_t *tuned);
/* One more sweep that also returns, per channel, sum|x_new - x_old| over the OWNED rows (the
 * local share of the reference's manhattonDist step, sparse-matrix.h:376) to a host array of
 * `channels` doubles; row-blocked callers all-reduce it.  Synchronises. */
int ccp_grid_sweep_l1(ccp_grid *g, double *l1_per_channel);
/* Tell the handle its ghost rows were just refreshed (by the caller's halo exchange). */
int ccp_grid_halo_refreshed(ccp_grid *g);

/* SparseMatrix::gaussSeidel loop (sparse-matrix.h:350-380) on the resident system, all
 * channels at once, each channel with its own stop test exactly as three separate reference
 * calls would (PhotoMontage.cpp:429-433): a channel that met `eps <= epsilon` is frozen.
 * report: array of `channels` entries (may be NULL).  Single-block handles only.
 * check_every: 1 = test after every sweep, exactly as the reference does — still temporally
 * blocked: a pass reports the step of each of its sweeps and a channel that met the rule inside a
 * pass is re-run to precisely that sweep; k >= 2 = test after every k-th sweep; 0 = no test,
 * exactly max_iteration sweeps (fastest). */
int ccp_grid_gauss_seidel(ccp_grid *g, double epsilon, int32_t max_iteration,
                          int32_t check_every, ccp_gs_report *report);

/* The same loop in the reference's OWN sweep order (index order, sparse-matrix.h:357-370): iterates,
 * stop sweep and result are those of SparseMatrix::gaussSeidel on the unpermuted matrix, bit for bit
 * (eps to summation order).  The sweeps are pipelined over the hyperplanes x + y + 2k (pixel, iteration):
 * time-skewed strips, 8 sweeps per pass through memory, one launch per pass depth, on a diagonal-major copy
 * of x and b (2x the image's memory while it runs).  Dirichlet-mask handles too: the unknowns are swept in
 * raster order (what SparseMatrix::gaussSeidel does with a region matrix numbered in raster order).
 * Whole-image handles only (no ghost rows: CCP_ERR_STATE).  check_every as above. */
int ccp_grid_gauss_seidel_lexicographic(ccp_grid *g, double epsilon, int32_t max_iteration,
                                        int32_t check_every, ccp_gs_report *report);

/* Diagnostics (host only, no device needed): the order in which a launch of the reference-order sweep hands its strips
 * to the persistent workgroups — `groups` groups of `depth` (1, 2, 4, 8) sweeps on an image `width` pixels wide,
 * groups * depth <= 1024.  order[ticket] = group * *strips + strip for the *count strips that hold a pixel of the image
 * (order may be NULL to ask for the sizes).  A strip waits only for strips with smaller tickets: its left neighbour,
 * the same strip of the group before, and strip + 1 of two groups before (tests/test_lex_tickets.py). */
int ccp_debug_lex_tickets(int32_t width, int32_t depth, int32_t groups, uint32_t *order, int64_t capacity, int32_t *strips,
                          int64_t *count);

/* conjugateGradient (sparse-matrix.h:396-434) on the resident system, matrix-free, channel by
 * channel; the resident x is the initial guess (ccp_grid_fill_x(g, 0) for the reference default,
 * ccp_grid_set_x_u8 for the composite start of PhotoMontage.cpp:599-610).  report: `channels`
 * entries (may be NULL).  Single-block handles only. */
int ccp_grid_conjugate_gradient(ccp_grid *g, double epsilon, int32_t max_iteration, ccp_gs_report *report);

/* ----------------------------------------------------------------------------------------
 * Row blocks across the GPUs of one node (BASELINE configs[3]).  One process (or host
 * thread) per GPU; each creates a communicator rank and one grid handle owning a contiguous block of
 * image rows plus `ghost` rows per neighbour side.  The reference has no counterpart — its solver is a
 * single-threaded loop (sparse-matrix.h:350-380); the call site that would drive this is the per-channel
 * solve of BuildSolveGradientFusion (PhotoMontage.cpp:428-434).
 *   id:   rank 0 calls ccp_comm_unique_id and hands the CCP_COMM_ID_BYTES bytes to the other ranks by
 *         any host channel (file, socket, MPI, torch.distributed ...);
 *   comm: every rank calls ccp_comm_create(id, rank, world, device) — collective (ncclCommInitRank);
 *   grid: ccp_grid_create with its row block, ccp_grid_attach_comm (collective: the blocks are checked
 *         to be the rank-ordered contiguous partition of one image), then the *_rowblocked calls.
 * Halo exchange: the `ghost` outermost owned rows per side travel to the neighbour's ghost rows once per
 * ghost/2 iterations (ncclSend/ncclRecv, one group, a stream of their own); in between the ghost rows are
 * recomputed redundantly, so the owned rows are bit-identical to the single-GPU red-black sweep.  With
 * overlap on (default) the pass that uses up the ghost rows finishes the rows the neighbours need first,
 * inside its one launch, and the messages leave while the rest of the pass is still running.
 * RCCL is bound at run time (librccl.so.1); CCP_ERR_RCCL reports a missing library or a failed call. */
#define CCP_COMM_ID_BYTES 128
typedef struct ccp_comm ccp_comm;
/* Can this process take part in a communicator on `device` (RCCL loadable — exactly one copy of it and of the
 * HIP runtime mapped in the process — and the device selectable)?  Not collective: call it on every rank and let
 * the ranks agree before the collective ccp_comm_create.  CCP_OK / CCP_ERR_RCCL / CCP_ERR_NO_DEVICE. */
int ccp_comm_probe(int32_t device);
int ccp_comm_unique_id(uint8_t *id_out /* CCP_COMM_ID_BYTES */);
int ccp_comm_create(const uint8_t *id /* CCP_COMM_ID_BYTES */, int32_t rank, int32_t world, int32_t device, ccp_comm **out);
int ccp_comm_destroy(ccp_comm *c);
int ccp_comm_info(ccp_comm *c, int32_t *rank, int32_t *world, int32_t *device, int32_t *rccl_version);
/* In-place sum / maximum over all ranks of `count` (<= 64) host doubles; synchronises. */
int ccp_comm_all_reduce_sum(ccp_comm *c, double *values, int32_t count);
int ccp_comm_all_reduce_max(ccp_comm *c, double *values, int32_t count);

/* Collective over the communicator.  c == NULL detaches.  The communicator must outlive the attachment. */
int ccp_grid_attach_comm(ccp_grid *g, ccp_comm *c);
int ccp_grid_set_overlap(ccp_grid *g, int32_t on);
/* Refresh the ghost rows now (after the caller wrote x, or before reading the ghost rows).  Async. */
int ccp_grid_exchange_halos(ccp_grid *g);
/* `iterations` red-black sweeps with the exchanges they need; no stop rule, no host sync.  Precondition:
 * ghost rows valid (fresh handle after fill/randomize on every rank, or ccp_grid_exchange_halos). */
int ccp_grid_sweep_rowblocked(ccp_grid *g, int32_t iterations);
/* SparseMatrix::gaussSeidel's loop (sparse-matrix.h:350-380) on the partitioned system: the L1 step of a
 * checked sweep is all-reduced, so every rank takes the same decision.  check_every as for
 * ccp_grid_gauss_seidel, and at its speed: checked temporally blocked passes report the step of each of their
 * sweeps, the blocks' sums are all-reduced once per pass, and a channel freezes at the sweep ITS rule fired at
 * (every channel's result equals the one-block solve's).  CCP_GS_ROWBLOCK_CHECKED_FUSED=0: the round-2 loop (one
 * in-place sweep per check; all channels run until the last one has met the rule). */
int ccp_grid_gauss_seidel_rowblocked(ccp_grid *g, double epsilon, int32_t max_iteration, int32_t check_every,
                                     ccp_gs_report *report);
/* SparseMatrix::conjugateGradient (sparse-matrix.h:396-434; the solver the blend call sites use) on the partitioned
 * system, from the x the blocks hold: every rank runs the loop on its owned rows; before every product with A the
 * direction's rows next to the block come from the neighbours (one image row each way; in the default fused loop one row
 * of the residual as well), every dot product is all-reduced.  Iterates equal the one-block loop's to rounding; a solve that stops stops at the same iteration on
 * every rank.  Needs ghost >= 1; works for Dirichlet-mask grids too.  Afterwards the ghost rows of x are stale (the
 * next rowblocked sweep refreshes them).  report: one per channel, as ccp_grid_conjugate_gradient.  Collective. */
int ccp_grid_conjugate_gradient_rowblocked(ccp_grid *g, double epsilon, int32_t max_iteration, ccp_gs_report *report);
/* ccp_grid_residual_norm2 summed over all blocks (refreshes stale ghost rows first).  Collective. */
int ccp_grid_residual_norm2_global(ccp_grid *g, double *rr_bb);
/* Statistics: exchanges issued; how the exchange waits for the edge rows (0 hipStreamWaitValue64, 1 polling
 * kernel, -1 no neighbours); rows sent up / down per exchange.  Outputs may be NULL.
 * The polling kernel (mode 1) is bounded (2 s of device time): should it ever give up, the rows it was
 * guarding may have travelled before they were final, and EVERY later call on the handle — in particular
 * every call that hands results to the host — returns CCP_ERR_STATE.  A lost hand-off is an error, never a
 * silently wrong ghost row. */
int ccp_grid_comm_stats(ccp_grid *g, int64_t *exchanges, int32_t *wait_mode, int32_t *send_up_rows, int32_t *send_down_rows);

/* Per-channel sums over the OWNED rows: rr = sum (b - A x)^2, bb = sum b^2 (2*channels
 * doubles: rr[0..ch), bb[0..ch)).  Synchronises. */
int ccp_grid_residual_norm2(ccp_grid *g, double *rr_bb);
/* Per-channel L1 distance sum|x - other| against a host vector is not needed: the L1 step of
 * the last checked sweep is in the report.  Sum over owned rows of |x| (checksum helper): */
int ccp_grid_abs_sum(ccp_grid *g, double *per_channel);

/* Poisson right-hand side on device: ATb for every channel from the gradient fields
 * (PhotoMontage.cpp:563-572,579-581,592).  gx, gy: host, H x W x channels float32
 * interleaved, row stride in bytes (cv::Mat CV_32FC3 layout); only y<H-1, x<W-1 are read.
 * constraint[ch] = the pin value v(0,0).  Single-block handles only. */
int ccp_grid_assemble_rhs(ccp_grid *g, const float *gx, const float *gy, int64_t row_stride_bytes,
                          const int32_t *constraint);
/* The whole front half of BuildSolveGradientFusion on device (PhotoMontage.cpp:399-433): the
 * gradient field of the label-selected images (GradientAt) and from it ATb of all three
 * channels, pin = Images[0](0,0)[ch]; optionally the composite image as the start vector
 * (PhotoMontage.cpp:599-610).  images[k]: H x W x 3 u8 interleaved (cv::Mat CV_8UC3), label:
 * H x W u8 (CV_8UC1), strides in bytes.  3-channel single-block handles only. */
int ccp_grid_assemble_from_images(ccp_grid *g, const uint8_t *const *images, int32_t n_images,
                                  int64_t image_stride_bytes, const uint8_t *label,
                                  int64_t label_stride_bytes, int32_t init_x_from_composite);
/* Solve epilogue (PhotoMontage.cpp:617-626): out(y,x)[ch] = uchar(max(min(x,255),0)) for all
 * channels into an interleaved H x W x channels u8 host image. */
int ccp_grid_store_u8(ccp_grid *g, uint8_t *out, int64_t row_stride_bytes);
/* Composite initial guess (PhotoMontage.cpp:599-610): x(y,x)[ch] = image(y,x)[ch] from an
 * interleaved u8 host image (the `fast_init_value` extension for GS). */
int ccp_grid_set_x_u8(ccp_grid *g, const uint8_t *image, int64_t row_stride_bytes);

/* Device time of the last ccp_grid_sweep / ccp_grid_gauss_seidel in milliseconds and the
 * number of half-sweep kernel launches it issued (HIP events on the handle's stream). */
int ccp_grid_last_timing(ccp_grid *g, float *milliseconds, int32_t *kernel_launches);
/* The same over a caller-chosen region spanning many calls: _begin records a HIP event on the handle's
 * stream, _end records a second one, waits for it and returns the device time between them and the
 * number of sweep launches issued in between (a temporally blocked pass — its ordinary and its border
 * kernel run side by side — counts once; an in-place half-sweep counts once) and the iterations the
 * temporally blocked passes among them performed (sum of their depths).  bench.py's roofline figure is
 * this time / this count, over exactly the timed steps. */
int ccp_grid_region_begin(ccp_grid *g);
int ccp_grid_region_end(ccp_grid *g, float *milliseconds, int64_t *sweep_launches, int64_t *pass_iterations);

#ifdef __cplusplus
}
#endif
#endif /* CCP_GS_H */
