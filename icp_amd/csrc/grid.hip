// icp_amd/csrc/grid.hip -- exact nearest-neighbour search over a uniform grid of the target cloud.
//
// Same result as the brute-force kernels (kernels.hip) and therefore as the oracle's scan
// (reference src/cpu/icp.cpp:72-93): idx[i] = the lowest index among the targets that minimise
// key = sqrt(sqrt(d^2)), d^2 evaluated in the oracle's un-fused fp64 order.  Only the set of targets that gets
// evaluated shrinks -- from all M to the ones in the grid cells a query's search box touches -- and the argument
// that nothing outside the box can win uses nothing but the monotonicity of IEEE rounding:
//
//  * cell_of() is a composition of monotone rounded operations, and the SAME function files the targets into cells
//    and maps the corners of a query box [a - r, a + r] to the cell range that is visited.  A target outside the
//    visited range therefore lies strictly outside the box along at least one axis, e.g. t.x < xlo = fl(a.x - r).
//  * its computed difference fl(a.x - t.x) is then >= g = fl(a.x - xlo), and since squares and sums of non-negative
//    terms round monotonically its computed d^2 is >= fl(g * g).  B = the smallest such bound over the box faces
//    that have cells beyond them.
//  * the search stops once B > tie_band(dmin), dmin = smallest d^2 evaluated so far: no unvisited target can reach
//    the key of the best one, not even as a tie.  Otherwise the box grows and is searched again.
//
// Targets are visited in cell order, not index order, so every candidate inside the tie band of the running
// minimum goes through the lexicographic (key, index) comparison -- a total order, independent of visiting order.
// One lane per source point; the previous iteration's correspondence (seed) makes the first box the final one.
#include "kernels.h"
#include "exact.h"

#include <math.h>
#include <string.h>

namespace icpk {

__device__ __forceinline__ int cell_of(double x, double lo, double inv, int dim)
{
    double t = (x - lo) * inv;                                        // monotone in x (inv >= 0)
    t = __builtin_fmin(__builtin_fmax(t, 0.0), (double)(dim - 1));   // NaN -> 0
    return (int)t;
}

// doubles order like these unsigned integers (NaNs end up beyond the infinities, so they surface in the result)
__device__ __forceinline__ unsigned long long ordered_bits(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static double from_ordered_bits(unsigned long long o)
{
    const unsigned long long b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    double v;
    memcpy(&v, &b, sizeof v);
    return v;
}

// out[0..2] = min x,y,z; out[3..5] = max x,y,z (ordered bit patterns; preset to ~0 / 0)
__global__ __launch_bounds__(kWG) void grid_bbox_kernel(const double *__restrict__ qx, const double *__restrict__ qy,
                                                        const double *__restrict__ qz, size_t m,
                                                        unsigned long long *__restrict__ out)
{
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0, 0, 0};
    for (size_t j = (size_t)blockIdx.x * kWG + threadIdx.x; j < m; j += (size_t)gridDim.x * kWG) {
        const unsigned long long b[3] = {ordered_bits(qx[j]), ordered_bits(qy[j]), ordered_bits(qz[j])};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = b[k] < lo[k] ? b[k] : lo[k];
            hi[k] = b[k] > hi[k] ? b[k] : hi[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo[k], o, 64), h2 = __shfl_xor(hi[k], o, 64);
            lo[k] = l2 < lo[k] ? l2 : lo[k];
            hi[k] = h2 > hi[k] ? h2 : hi[k];
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&out[k], lo[k]);
            atomicMax(&out[3 + k], hi[k]);
        }
    }
}

__global__ __launch_bounds__(kWG) void grid_count_kernel(const double *__restrict__ qx, const double *__restrict__ qy,
                                                         const double *__restrict__ qz, size_t m, GridDesc g,
                                                         uint32_t *__restrict__ cell, uint32_t *__restrict__ counts)
{
    const size_t j = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= m) return;
    const int cx = cell_of(qx[j], g.lo[0], g.inv[0], g.dim[0]);
    const int cy = cell_of(qy[j], g.lo[1], g.inv[1], g.dim[1]);
    const int cz = cell_of(qz[j], g.lo[2], g.inv[2], g.dim[2]);
    const uint32_t c = ((uint32_t)cz * (uint32_t)g.dim[1] + (uint32_t)cy) * (uint32_t)g.dim[0] + (uint32_t)cx;
    cell[j] = c;
    atomicAdd(&counts[c], 1u);
}

// exclusive prefix sum of counts[cells] into start[cells + 1] (and a copy in cursor[cells]); one workgroup,
// run once per target cloud
__global__ __launch_bounds__(1024) void grid_scan_kernel(const uint32_t *__restrict__ counts, size_t cells,
                                                         uint32_t *__restrict__ start, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t part[1024];
    const size_t per = (cells + 1023) / 1024;
    const size_t b = (size_t)threadIdx.x * per;
    const size_t e = b + per < cells ? b + per : cells;
    uint32_t sum = 0;
    for (size_t k = b; k < e; ++k) sum += counts[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (size_t k = b; k < e; ++k) {
        start[k] = run;
        cursor[k] = run;
        run += counts[k];
    }
    if (threadIdx.x == 1023) start[cells] = part[1023];
}

// targets sorted by cell: coordinates (SoA) and original index.  The order inside a cell is arbitrary (atomic
// cursor); the search does not depend on it.
__global__ __launch_bounds__(kWG) void grid_scatter_kernel(const double *__restrict__ qx, const double *__restrict__ qy,
                                                           const double *__restrict__ qz, size_t m,
                                                           const uint32_t *__restrict__ cell, uint32_t *__restrict__ cursor,
                                                           double *__restrict__ sx, double *__restrict__ sy,
                                                           double *__restrict__ sz, uint32_t *__restrict__ perm)
{
    const size_t j = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= m) return;
    const uint32_t pos = atomicAdd(&cursor[cell[j]], 1u);
    sx[pos] = qx[j];
    sy[pos] = qy[j];
    sz[pos] = qz[j];
    perm[pos] = (uint32_t)j;
}

__global__ __launch_bounds__(kWG) void search_grid_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                          const double *__restrict__ pz, const double *__restrict__ pc, size_t n,
                                                          GridDesc g, const uint32_t *__restrict__ start,
                                                          const double *__restrict__ sx, const double *__restrict__ sy,
                                                          const double *__restrict__ sz, const uint32_t *__restrict__ perm,
                                                          const double *__restrict__ qx, const double *__restrict__ qy,
                                                          const double *__restrict__ qz, unsigned m,
                                                          const uint32_t *__restrict__ idx_prev, double *__restrict__ out_key,
                                                          uint32_t *__restrict__ out_idx)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n) return;
    const double ax = px[i] - pc[0], ay = py[i] - pc[1], az = pz[i] - pc[2];
    double bk = INFINITY, dmin = INFINITY;   // oracle: min_dist = DBL_MAX on the key, idx = -1
    uint32_t bj = 0xFFFFFFFFu;
    const double amax = __builtin_fmax(__builtin_fmax(__builtin_fabs(ax), __builtin_fabs(ay)), __builtin_fabs(az));
    if (!(amax < INFINITY) || ax != ax || ay != ay || az != az) {   // every d^2 is inf or NaN: no correspondence
        out_key[i] = bk;
        out_idx[i] = bj;
        return;
    }
    if (idx_prev) {
        const uint32_t js = idx_prev[i];
        if (js < m) {
            const double d = d2_exact(ax, ay, az, qx[js], qy[js], qz[js]);
            if (d < INFINITY) {
                dmin = d;
                bk = key_of(d);
                bj = js;
            }
        }
    }
    double edge = dmin < INFINITY ? tie_band(dmin) : INFINITY;
    double r;
    if (edge < INFINITY) {
        r = sqrt(edge) * (1.0 + 0x1p-10);
    } else {   // no candidate yet: reach the bounding box of the targets, plus half a cell
        const double ex = __builtin_fmax(__builtin_fmax(g.lo[0] - ax, ax - g.hi[0]), 0.0);
        const double ey = __builtin_fmax(__builtin_fmax(g.lo[1] - ay, ay - g.hi[1]), 0.0);
        const double ez = __builtin_fmax(__builtin_fmax(g.lo[2] - az, az - g.hi[2]), 0.0);
        r = __builtin_fmax(__builtin_fmax(ex, ey), ez) + g.r0;
    }
    r = __builtin_fmax(__builtin_fmax(r, 0x1p-30 * amax), 1e-150);   // keep a - r distinguishable from a
    for (;;) {
        const double xlo = ax - r, xhi = ax + r, ylo = ay - r, yhi = ay + r, zlo = az - r, zhi = az + r;
        const int cx0 = cell_of(xlo, g.lo[0], g.inv[0], g.dim[0]), cx1 = cell_of(xhi, g.lo[0], g.inv[0], g.dim[0]);
        const int cy0 = cell_of(ylo, g.lo[1], g.inv[1], g.dim[1]), cy1 = cell_of(yhi, g.lo[1], g.inv[1], g.dim[1]);
        const int cz0 = cell_of(zlo, g.lo[2], g.inv[2], g.dim[2]), cz1 = cell_of(zhi, g.lo[2], g.inv[2], g.dim[2]);
        // lower bound on the computed d^2 of every target outside the visited cells
        double B = INFINITY;
        if (cx0 > 0) { const double t = ax - xlo; B = __builtin_fmin(B, t * t); }
        if (cx1 < g.dim[0] - 1) { const double t = xhi - ax; B = __builtin_fmin(B, t * t); }
        if (cy0 > 0) { const double t = ay - ylo; B = __builtin_fmin(B, t * t); }
        if (cy1 < g.dim[1] - 1) { const double t = yhi - ay; B = __builtin_fmin(B, t * t); }
        if (cz0 > 0) { const double t = az - zlo; B = __builtin_fmin(B, t * t); }
        if (cz1 < g.dim[2] - 1) { const double t = zhi - az; B = __builtin_fmin(B, t * t); }
        const bool all = cx0 == 0 && cy0 == 0 && cz0 == 0 && cx1 == g.dim[0] - 1 && cy1 == g.dim[1] - 1 && cz1 == g.dim[2] - 1;
        for (int cz = cz0; cz <= cz1; ++cz) {
            for (int cy = cy0; cy <= cy1; ++cy) {
                // the cells cx0..cx1 of one (cy, cz) row are adjacent in the sorted arrays
                const size_t row = ((size_t)cz * (size_t)g.dim[1] + (size_t)cy) * (size_t)g.dim[0];
                const uint32_t k0 = start[row + cx0], k1 = start[row + cx1 + 1];
                for (uint32_t k = k0; k < k1; ++k) {
                    const double d = d2_exact(ax, ay, az, sx[k], sy[k], sz[k]);
                    if (d <= edge && d < INFINITY) {
                        const double key = key_of(d);
                        const uint32_t j = perm[k];
                        if (key < bk || (key == bk && j < bj)) {
                            bk = key;
                            bj = j;
                        }
                        if (d < dmin) {
                            dmin = d;
                            edge = tie_band(d);
                        }
                    }
                }
            }
        }
        if (all || B > edge) break;
        r = (edge < INFINITY) ? __builtin_fmax(sqrt(edge) * (1.0 + 0x1p-10), r * 1.0625) : 2.0 * r;
    }
    out_key[i] = bk;
    out_idx[i] = bj;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void grid_free(Grid *g)
{
    if (!g) return;
    if (g->start) (void)hipFree(g->start);
    if (g->perm) (void)hipFree(g->perm);
    if (g->store) (void)hipFree(g->store);
    *g = Grid{};
}

// Cell size: about four targets per cell of the occupied volume (area / length when the cloud is flat along some
// axes), at most 1024 cells per axis and 2^22 cells in all.
static void choose_cells(const double lo[3], const double hi[3], size_t m, GridDesc *d)
{
    double ext[3];
    int active = 0;
    double vol = 1.0;
    for (int k = 0; k < 3; ++k) {
        ext[k] = hi[k] - lo[k];
        if (ext[k] > 0.0) { ++active; vol *= ext[k]; }
    }
    double want = (double)m / 4.0;
    if (want < 1.0) want = 1.0;
    if (want > 4194304.0) want = 4194304.0;
    double h = active ? pow(vol / want, 1.0 / active) : 1.0;
    if (!(h > 0.0) || !(h < INFINITY)) {   // under/overflow of the volume: fall back to the longest axis
        const double longest = fmax(fmax(ext[0], ext[1]), ext[2]);
        h = longest > 0.0 ? longest / cbrt(want) : 1.0;
    }
    for (;;) {
        double cells = 1.0;
        for (int k = 0; k < 3; ++k) {
            int dim = 1;
            if (ext[k] > 0.0) {
                const double c = ceil(ext[k] / h);
                dim = c >= 1024.0 ? 1024 : (c < 1.0 ? 1 : (int)c);
            }
            d->dim[k] = dim;
            cells *= dim;
        }
        if (cells <= 4194304.0) break;
        h *= 1.26;
    }
    double hmin = INFINITY;
    for (int k = 0; k < 3; ++k) {
        d->lo[k] = lo[k];
        d->hi[k] = hi[k];
        d->inv[k] = ext[k] > 0.0 ? (double)d->dim[k] / ext[k] : 0.0;
        if (!(d->inv[k] < INFINITY)) { d->inv[k] = 0.0; d->dim[k] = 1; }   // extent so small that dim/ext overflows
        if (d->dim[k] > 1) hmin = fmin(hmin, ext[k] / d->dim[k]);
    }
    d->r0 = hmin < INFINITY ? 0.5 * hmin : 1.0;
}

hipError_t grid_build(hipStream_t s, Soa3 Qz, size_t m, Grid *g)
{
    grid_free(g);
    if (m == 0 || m > 0xFFFFFFF0u) return hipErrorInvalidValue;
    hipError_t e;
    unsigned long long *bb = nullptr;
    uint32_t *cell = nullptr, *counts = nullptr, *cursor = nullptr;
    unsigned long long host_bb[6];
#define GRID_TRY(expr) do { e = (expr); if (e != hipSuccess) goto done; } while (0)
    GRID_TRY(hipMalloc(&bb, 6 * sizeof(unsigned long long)));
    GRID_TRY(hipMemsetAsync(bb, 0xFF, 3 * sizeof(unsigned long long), s));
    GRID_TRY(hipMemsetAsync(bb + 3, 0x00, 3 * sizeof(unsigned long long), s));
    {
        size_t blocks = (m + kWG - 1) / kWG;
        if (blocks > 2048) blocks = 2048;
        grid_bbox_kernel<<<(unsigned)blocks, kWG, 0, s>>>(Qz.x, Qz.y, Qz.z, m, bb);
    }
    GRID_TRY(hipGetLastError());
    GRID_TRY(hipMemcpyAsync(host_bb, bb, sizeof host_bb, hipMemcpyDeviceToHost, s));
    GRID_TRY(hipStreamSynchronize(s));
    {
        double lo[3], hi[3];
        bool finite = true;
        for (int k = 0; k < 3; ++k) {
            lo[k] = from_ordered_bits(host_bb[k]);
            hi[k] = from_ordered_bits(host_bb[3 + k]);
            finite = finite && isfinite(lo[k]) && isfinite(hi[k]) && isfinite(hi[k] - lo[k]);
        }
        if (!finite) {   // NaN / inf coordinates: no grid; the caller uses a brute-force variant
            g->ok = false;
            e = hipSuccess;
            goto done;
        }
        choose_cells(lo, hi, m, &g->d);
    }
    g->cells = (size_t)g->d.dim[0] * (size_t)g->d.dim[1] * (size_t)g->d.dim[2];
    GRID_TRY(hipMalloc(&g->start, (g->cells + 1) * sizeof(uint32_t)));
    GRID_TRY(hipMalloc(&g->perm, m * sizeof(uint32_t)));
    GRID_TRY(hipMalloc(&g->store, 3 * m * sizeof(double)));
    g->sx = g->store;
    g->sy = g->store + m;
    g->sz = g->store + 2 * m;
    GRID_TRY(hipMalloc(&cell, m * sizeof(uint32_t)));
    GRID_TRY(hipMalloc(&counts, g->cells * sizeof(uint32_t)));
    GRID_TRY(hipMalloc(&cursor, g->cells * sizeof(uint32_t)));
    GRID_TRY(hipMemsetAsync(counts, 0, g->cells * sizeof(uint32_t), s));
    grid_count_kernel<<<(unsigned)((m + kWG - 1) / kWG), kWG, 0, s>>>(Qz.x, Qz.y, Qz.z, m, g->d, cell, counts);
    GRID_TRY(hipGetLastError());
    grid_scan_kernel<<<1, 1024, 0, s>>>(counts, g->cells, g->start, cursor);
    GRID_TRY(hipGetLastError());
    grid_scatter_kernel<<<(unsigned)((m + kWG - 1) / kWG), kWG, 0, s>>>(Qz.x, Qz.y, Qz.z, m, cell, cursor, g->sx, g->sy, g->sz,
                                                                       g->perm);
    GRID_TRY(hipGetLastError());
    GRID_TRY(hipStreamSynchronize(s));   // the temporaries are freed below
    g->ok = true;
#undef GRID_TRY
done:
    if (bb) (void)hipFree(bb);
    if (cell) (void)hipFree(cell);
    if (counts) (void)hipFree(counts);
    if (cursor) (void)hipFree(cursor);
    if (e != hipSuccess) grid_free(g);
    return e;
}

hipError_t search_grid(hipStream_t s, Soa3 P, size_t n, const double *pc, const Grid &g, Soa3 Qz, size_t m,
                       const uint32_t *idx_prev, double *out_key, uint32_t *out_idx)
{
    if (!g.ok || m == 0 || m > 0xFFFFFFF0u) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    search_grid_kernel<<<(unsigned)((n + kWG - 1) / kWG), kWG, 0, s>>>(P.x, P.y, P.z, pc, n, g.d, g.start, g.sx, g.sy, g.sz, g.perm,
                                                                      Qz.x, Qz.y, Qz.z, (unsigned)m, idx_prev, out_key, out_idx);
    return hipGetLastError();
}

}  // namespace icpk
