// ccp_csr_region.hpp — the O(n) parts of the raster-region recognition (ccp_csr.hip: detect_region) as device kernels.
//
// The question (see embed_region in ccp_csr.hip, the host statement of the same algorithm): is the uploaded matrix the
// 5-point Laplacian of a pixel region — diagonal 4, -1 to every 4-neighbour inside the region, unknowns numbered in raster
// order — and if so, where does every unknown sit on a canvas?  At BASELINE configs[4]'s 41.75 M unknowns the host version
// costs 0.3 s for the embedding plus 0.2 s for the mask and index arrays and their upload (256 host cores, 32 threads):
// two hundred times the 2 ms the 50 sweeps then take.  Here the device does everything that is per unknown —
//   classify   which couplings are "the pixel above" / "the pixel to the left" (k_region_classify)
//   runs       horizontal runs of consecutive unknowns: an inclusive scan of the run-start flags (hipcub)
//   links      ONE vertical coupling per pair of runs, compacted with what the host needs to know about it
//   place      canvas coordinates of every unknown from its run's origin; identity canvas, `where`, mask bytes
//   verify     every coupling a 4-neighbour pair on the canvas, every 4-neighbour pair a coupling
// — and the host keeps only what is per RUN (a few per canvas row and connected piece): the weighted union-find that
// turns the links into relative positions, the bounding boxes of the pieces and their shelf layout.
#pragma once

#include "ccp_common.hpp"

namespace ccp {

// up[i] = the unknown above i (or -1), has_left[i] = i-1 is its left neighbour.  A row that cannot be a region row
// (more than two earlier / two later couplings, no diagonal, a second earlier coupling that is not i-1) raises *bad.
// Values are checked by the caller (4 on the diagonal, -1 elsewhere).
__global__ void __launch_bounds__(kBlock)
k_region_classify(const long *__restrict__ row_ptr, const int *__restrict__ col, int n, int *__restrict__ up,
                  unsigned char *__restrict__ has_left, int *__restrict__ run_flag, int *__restrict__ bad)
{
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int lower0 = -1, lower1 = -1, n_lower = 0, n_upper = 0;
    bool diag = false, ok = true;
    for (long k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        const int c = col[k];
        if (c == (int)i) {
            ok &= !diag;
            diag = true;
        } else if (c < (int)i) {
            if (n_lower == 0) lower0 = c;
            else if (n_lower == 1) lower1 = c;
            ++n_lower;
        } else {
            ++n_upper;
        }
    }
    ok &= diag && n_upper <= 2 && n_lower <= 2;
    int u = -1;
    unsigned char hl = 0;
    if (ok && n_lower == 2) {
        ok = lower1 == (int)i - 1 && lower0 < (int)i - 1;
        u = lower0;
        hl = 1;
    } else if (ok && n_lower == 1) {
        if (lower0 == (int)i - 1) hl = 1;                 // taken as the pixel to the left (embed_region has the argument)
        else u = lower0;
    }
    up[i] = u;
    has_left[i] = hl;
    run_flag[i] = hl ? 0 : 1;
    if (!ok) atomicOr(bad, 1);
}

// run_id = inclusive scan of run_flag (run of i = run_id[i] - 1): first unknown and colour of every run
__global__ void __launch_bounds__(kBlock)
k_region_run_starts(const int *__restrict__ run_id, const unsigned char *__restrict__ has_left, const int *__restrict__ colour, int n,
                    int *__restrict__ run_start, int *__restrict__ run_colour)
{
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || has_left[i]) return;
    const int r = run_id[i] - 1;
    run_start[r] = (int)i;
    run_colour[r] = colour[i];
}

// One constraint per pair of runs: a vertical coupling parallel to its left neighbour's says nothing new.
// link[4p..4p+3] = run of the upper unknown, run of the lower one, offsets of the two inside their runs; link_at[p] = i.
__global__ void __launch_bounds__(kBlock)
k_region_links(const int *__restrict__ up, const unsigned char *__restrict__ has_left, const int *__restrict__ run_id,
               const int *__restrict__ run_start, int n, int *__restrict__ counter, int cap, int *__restrict__ link, int *__restrict__ link_at)
{
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int u = up[i];
    if (u < 0) return;
    if (has_left[i] && u >= 1 && up[i - 1] == u - 1 && has_left[u]) return;
    const int p = atomicAdd(counter, 1);
    if (p >= cap) return;
    const int A = run_id[u] - 1, B = run_id[i] - 1;
    link[4 * (long)p] = A;
    link[4 * (long)p + 1] = B;
    link[4 * (long)p + 2] = u - run_start[A];
    link[4 * (long)p + 3] = (int)i - run_start[B];
    link_at[p] = (int)i;
}

// Canvas position of every unknown from its run's origin (x0, y0): identity canvas, index into the grid's colour-split
// planes, mask byte.  Off the canvas interior or of the wrong colour parity: *bad.
// free_parity: no colouring came with the matrix — the canvas parity BECOMES the colouring (written to colour[i]).
__global__ void __launch_bounds__(kBlock)
k_region_place(const int *__restrict__ run_id, const int *__restrict__ run_start, const int *__restrict__ x0, const int *__restrict__ y0,
               int *__restrict__ colour, int n, int W, int H, long pitch, int *__restrict__ ident, long *__restrict__ where,
               unsigned char *__restrict__ mask_split, int *__restrict__ bad, int free_parity)
{
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int r = run_id[i] - 1;
    const int x = x0[r] + ((int)i - run_start[r]), y = y0[r];
    if (free_parity) colour[i] = (x + y) & 1;
    if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1 || ((x + y) & 1) != (colour[i] & 1)) {
        atomicOr(bad, 1);
        where[i] = 0;
        return;
    }
    ident[(long)y * W + x] = (int)i;
    const long w = ((long)y * 2 + ((x + y) & 1)) * pitch + (x >> 1);
    where[i] = w;
    mask_split[w] = 1;
}

// Every coupling of row i is one of its four canvas neighbours and every canvas neighbour that is an unknown is coupled.
__global__ void __launch_bounds__(kBlock)
k_region_verify(const long *__restrict__ row_ptr, const int *__restrict__ col, const int *__restrict__ run_id, const int *__restrict__ run_start,
                const int *__restrict__ x0, const int *__restrict__ y0, int n, int W, const int *__restrict__ ident, int *__restrict__ bad)
{
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int r = run_id[i] - 1;
    const int x = x0[r] + ((int)i - run_start[r]), y = y0[r];
    const long at = (long)y * W + x;
    bool ok = ident[at] == (int)i;
    const int nb0 = ident[at - W], nb1 = ident[at - 1], nb2 = ident[at + 1], nb3 = ident[at + W];
    const int present = (nb0 >= 0) + (nb1 >= 0) + (nb2 >= 0) + (nb3 >= 0);
    int matched = 0;
    long off_diag = 0;
    for (long k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        const int c = col[k];
        if (c == (int)i) continue;
        ++off_diag;
        matched += (c == nb0) + (c == nb1) + (c == nb2) + (c == nb3);
    }
    ok &= matched == present && off_diag == present;
    if (!ok) atomicOr(bad, 1);
}

__global__ void __launch_bounds__(kBlock)
k_fill_int(int *__restrict__ p, long n, int v)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) p[i] = v;
}

}  // namespace ccp

// ---- construction of a sliced-ELL image on the device (ccp_csr.hip: build_schedule_device) ---------------------------
namespace ccp {

__global__ void __launch_bounds__(kBlock)
k_iota(int *__restrict__ p, long n)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) p[i] = (int)i;
}

// inv[perm[r]] = r
__global__ void __launch_bounds__(kBlock)
k_invert_perm(const int *__restrict__ perm, int *__restrict__ inv, long n)
{
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < n; r += (long)gridDim.x * kBlock) inv[perm[r]] = (int)r;
}

// One wave per slice: the widest row of the slice, and the entries its slab takes ((width + slack) * 64).
__global__ void __launch_bounds__(kBlock)
k_slice_widths(const long *__restrict__ row_ptr, const int *__restrict__ perm, const int *__restrict__ slice_row0,
               const int *__restrict__ slice_rows, int n_slices, int slack, int *__restrict__ width, long *__restrict__ cells)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int s = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (s >= n_slices) return;
    int len = 0;
    if (lane < slice_rows[s]) {
        const int old = perm[slice_row0[s] + lane];
        len = (int)(row_ptr[old + 1] - row_ptr[old]);
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, kWave));
    if (lane == 0) {
        width[s] = len;
        cells[s] = (long)(len + slack) * kWave;
    }
}

// One wave per slice: padding (column -1, value 0) over the whole slab, then lane t lays row perm[row0 + t] out —
// entry k at off + k*64 + t, columns mapped through inv (square part only) — and, SORTED, orders its entries by the
// mapped column with the stable insertion sort of the host statement (rows are a handful of entries).
template <bool SORTED>
__global__ void __launch_bounds__(kBlock)
k_slice_fill(const long *__restrict__ row_ptr, const int *__restrict__ col, const double *__restrict__ val, const int *__restrict__ perm,
             const int *__restrict__ inv, int n, const int *__restrict__ slice_row0, const int *__restrict__ slice_rows,
             const int *__restrict__ width, const long *__restrict__ slice_off, int n_slices, int slack, int *__restrict__ cols,
             double *__restrict__ vals)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int s = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (s >= n_slices) return;
    const long base = slice_off[s];
    const int cap = width[s] + slack;
    for (int k = 0; k < cap; ++k) {
        cols[base + (long)k * kWave + lane] = -1;
        vals[base + (long)k * kWave + lane] = 0.0;
    }
    if (lane >= slice_rows[s]) return;
    const int old = perm[slice_row0[s] + lane];
    const long a = row_ptr[old];
    const int len = (int)(row_ptr[old + 1] - a);
    for (int k = 0; k < len; ++k) {
        const int c = col[a + k];
        const int mc = (c >= 0 && c < n) ? inv[c] : c;
        const double v = val[a + k];
        int b = k;
        if (SORTED) {
            // (entries 0..k-1 of this lane are in place and sorted: shift the larger ones up)
            while (b > 0 && cols[base + (long)(b - 1) * kWave + lane] > mc) {
                cols[base + (long)b * kWave + lane] = cols[base + (long)(b - 1) * kWave + lane];
                vals[base + (long)b * kWave + lane] = vals[base + (long)(b - 1) * kWave + lane];
                --b;
            }
        }
        cols[base + (long)b * kWave + lane] = mc;
        vals[base + (long)b * kWave + lane] = v;
    }
}

}  // namespace ccp
