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
//    visited range therefore lies strictly outside the box along at least one axis, e.g. t.x < xlo = fl(a.x - r)
//    (and even below any xe < xlo that cell_of() still maps into the visited range: reach_lo / reach_hi take xe
//    next to the outer edge of the visited cells, so the bound counts the whole of the cells that were searched).
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
#include "reduce.h"
#include "exact.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
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
    __shared__ unsigned long long s_lo[3][kWG / 64], s_hi[3][kWG / 64];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo[k], o, 64), h2 = __shfl_xor(hi[k], o, 64);
            lo[k] = l2 < lo[k] ? l2 : lo[k];
            hi[k] = h2 > hi[k] ? h2 : hi[k];
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[k][threadIdx.x >> 6] = lo[k];
            s_hi[k][threadIdx.x >> 6] = hi[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {   // six atomics per workgroup on the six result words (they serialise: 77 us with one set per wave)
        const int k = threadIdx.x % 3;
        const bool high = threadIdx.x >= 3;
        unsigned long long v = high ? s_hi[k][0] : s_lo[k][0];
#pragma unroll
        for (int w = 1; w < kWG / 64; ++w) {
            const unsigned long long o = high ? s_hi[k][w] : s_lo[k][w];
            v = high ? (o > v ? o : v) : (o < v ? o : v);
        }
        if (high) atomicMax(&out[3 + k], v);
        else atomicMin(&out[k], v);
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

// The same prefix sum in three launches for large grids: per-block sums (4096 cells per workgroup), their scan by
// grid_scan_kernel, then the local scans.
constexpr int kScanPerThread = 16;
constexpr size_t kScanPerBlock = (size_t)kWG * kScanPerThread;

__global__ __launch_bounds__(kWG) void grid_scan_sums_kernel(const uint32_t *__restrict__ counts, size_t cells,
                                                             uint32_t *__restrict__ block_sums)
{
    __shared__ uint32_t part[kWG / 64];
    const size_t b = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * kScanPerThread;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
        if (b + k < cells) sum += counts[b + k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(kWG) void grid_scan_apply_kernel(const uint32_t *__restrict__ counts, size_t cells,
                                                              const uint32_t *__restrict__ block_off, uint32_t *__restrict__ start,
                                                              uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t part[kWG];
    const size_t b = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * kScanPerThread;
    uint32_t c[kScanPerThread], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        c[k] = (b + k < cells) ? counts[b + k] : 0u;
        sum += c[k];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < kWG; o <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = block_off[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0u);
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        if (b + k < cells) {
            start[b + k] = run;
            cursor[b + k] = run;
        }
        run += c[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kWG - 1) start[cells] = run;
}

// targets sorted by cell: one 32-byte record {x, y, z, original index} each (one load per candidate), plus the index alone.  The order inside a cell is arbitrary (atomic
// cursor); the search does not depend on it.
__global__ __launch_bounds__(kWG) void grid_scatter_kernel(const double *__restrict__ qx, const double *__restrict__ qy,
                                                           const double *__restrict__ qz, size_t m,
                                                           const uint32_t *__restrict__ cell, uint32_t *__restrict__ cursor,
                                                           double4 *__restrict__ rec, uint32_t *__restrict__ perm)
{
    const size_t j = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= m) return;
    const uint32_t pos = atomicAdd(&cursor[cell[j]], 1u);
    if (rec) rec[pos] = make_double4(qx[j], qy[j], qz[j], __longlong_as_double((long long)j));   // index as a bit pattern
    perm[pos] = (uint32_t)j;
}

// The cells a box [a - r, a + r]^3 touches, and B = lower bound on the computed d^2 of every target in the other cells.
struct Box {
    int cx0, cx1, cy0, cy1, cz0, cz1;
    double B;
    bool all;   // the box covers the whole grid
};
// How far the VISITED CELLS reach beyond the box along one axis.  Every target of the cells c0..c1 is evaluated, not only
// those inside the box, so the unvisited targets lie beyond the outer edges of those cells, which is further out than the
// box faces (half a cell on average -- as much again as the distance to the match: a much stronger bound for the same
// work, which ends searches earlier and lets certificates outlive more motion).  The edge is not trusted as arithmetic:
// xe = a point 2^-20 of a cell inside the edge, accepted only if cell_of() -- the function that filed the targets -- maps
// it into the visited range.  Then an unvisited target t below the range has cell_of(t) < c0 <= cell_of(xe), hence t < xe
// by monotonicity, exactly the property the box face xlo has (and falls back to: cell_of(xlo) = c0 by construction).
__device__ __forceinline__ double reach_lo(double xlo, int c0, double lo, double inv, double h, int dim)
{
    const double xe = lo + ((double)c0 + 0x1p-20) * h;
    return (xe < xlo && cell_of(xe, lo, inv, dim) >= c0) ? xe : xlo;
}
__device__ __forceinline__ double reach_hi(double xhi, int c1, double lo, double inv, double h, int dim)
{
    const double xe = lo + ((double)(c1 + 1) - 0x1p-20) * h;
    return (xe > xhi && cell_of(xe, lo, inv, dim) <= c1) ? xe : xhi;
}
__device__ __forceinline__ Box make_box(const GridDesc &g, double ax, double ay, double az, double r)
{
    Box b;
    const double xlo = ax - r, xhi = ax + r, ylo = ay - r, yhi = ay + r, zlo = az - r, zhi = az + r;
    b.cx0 = cell_of(xlo, g.lo[0], g.inv[0], g.dim[0]);
    b.cx1 = cell_of(xhi, g.lo[0], g.inv[0], g.dim[0]);
    b.cy0 = cell_of(ylo, g.lo[1], g.inv[1], g.dim[1]);
    b.cy1 = cell_of(yhi, g.lo[1], g.inv[1], g.dim[1]);
    b.cz0 = cell_of(zlo, g.lo[2], g.inv[2], g.dim[2]);
    b.cz1 = cell_of(zhi, g.lo[2], g.inv[2], g.dim[2]);
    double B = INFINITY;
    if (b.cx0 > 0) { const double t = ax - reach_lo(xlo, b.cx0, g.lo[0], g.inv[0], g.h[0], g.dim[0]); B = __builtin_fmin(B, t * t); }
    if (b.cx1 < g.dim[0] - 1) { const double t = reach_hi(xhi, b.cx1, g.lo[0], g.inv[0], g.h[0], g.dim[0]) - ax; B = __builtin_fmin(B, t * t); }
    if (b.cy0 > 0) { const double t = ay - reach_lo(ylo, b.cy0, g.lo[1], g.inv[1], g.h[1], g.dim[1]); B = __builtin_fmin(B, t * t); }
    if (b.cy1 < g.dim[1] - 1) { const double t = reach_hi(yhi, b.cy1, g.lo[1], g.inv[1], g.h[1], g.dim[1]) - ay; B = __builtin_fmin(B, t * t); }
    if (b.cz0 > 0) { const double t = az - reach_lo(zlo, b.cz0, g.lo[2], g.inv[2], g.h[2], g.dim[2]); B = __builtin_fmin(B, t * t); }
    if (b.cz1 < g.dim[2] - 1) { const double t = reach_hi(zhi, b.cz1, g.lo[2], g.inv[2], g.h[2], g.dim[2]) - az; B = __builtin_fmin(B, t * t); }
    b.B = B;
    b.all = b.cx0 == 0 && b.cy0 == 0 && b.cz0 == 0 && b.cx1 == g.dim[0] - 1 && b.cy1 == g.dim[1] - 1 && b.cz1 == g.dim[2] - 1;
    return b;
}
// First radius: the tie band of the seed's distance, or (no seed) up to the targets' bounding box plus half a cell.
__device__ __forceinline__ double first_radius(const GridDesc &g, double ax, double ay, double az, double amax, double edge)
{
    double r;
    if (edge < INFINITY) {
        r = sqrt(edge) * (1.0 + 0x1p-10);
    } else {
        const double ex = __builtin_fmax(__builtin_fmax(g.lo[0] - ax, ax - g.hi[0]), 0.0);
        const double ey = __builtin_fmax(__builtin_fmax(g.lo[1] - ay, ay - g.hi[1]), 0.0);
        const double ez = __builtin_fmax(__builtin_fmax(g.lo[2] - az, az - g.hi[2]), 0.0);
        r = __builtin_fmax(__builtin_fmax(ex, ey), ez) + g.r0;
    }
    return __builtin_fmax(__builtin_fmax(r, 0x1p-30 * amax), 1e-150);   // keep a - r distinguishable from a
}
__device__ __forceinline__ double next_radius(double edge, double r)
{
    return (edge < INFINITY) ? __builtin_fmax(sqrt(edge) * (1.0 + 0x1p-10), r * 1.0625) : 2.0 * r;
}
// One evaluated target: lexicographic (key, index) best among the candidates inside the tie band of the running minimum.
__device__ __forceinline__ bool consider(double d, uint32_t j, double &dmin, double &edge, double &bk, uint32_t &bj)
{
    bool took = false;
    if (d <= edge && d < INFINITY) {
        const double key = key_of(d);
        if (key < bk || (key == bk && j < bj)) {
            bk = key;
            bj = j;
            took = true;
        }
        if (d < dmin) {
            dmin = d;
            edge = tie_band(d);
        }
    }
    return took;
}

__device__ __forceinline__ double seed_d2(double ax, double ay, double az, uint32_t js, const double *__restrict__ qx,
                                          const double *__restrict__ qy, const double *__restrict__ qz,
                                          const double4 *__restrict__ qrec4)
{
    if (qrec4) {
        const double4 v = qrec4[js];
        return d2_exact(ax, ay, az, v.x, v.y, v.z);
    }
    return d2_exact(ax, ay, az, qx[js], qy[js], qz[js]);
}
__device__ __forceinline__ void consider_rec(double ax, double ay, double az, const double4 v, double &dmin, double &edge,
                                             double &bk, uint32_t &bj)
{
    consider(d2_exact(ax, ay, az, v.x, v.y, v.z), (uint32_t)__double_as_longlong(v.w), dmin, edge, bk, bj);
}
// same, also remembering the coordinates of the best target (they go back into the source row's record)
__device__ __forceinline__ void consider_rec(double ax, double ay, double az, const double4 v, double &dmin, double &edge,
                                             double &bk, uint32_t &bj, double &bx, double &by, double &bz)
{
    if (consider(d2_exact(ax, ay, az, v.x, v.y, v.z), (uint32_t)__double_as_longlong(v.w), dmin, edge, bk, bj)) {
        bx = v.x;
        by = v.y;
        bz = v.z;
    }
}

// Certificate carried from one iteration to the next (grid search with source records): s = a DISTANCE such that the computed
// d^2 of EVERY target other than the matched one is >= s^2, at the row's centred position a of that iteration.  The next
// iteration moves the row to a' with |a' - a| <= delta (bounded from the rigid motion the engine applied), so every other
// target then has a computed d^2 >= (s - delta)^2 up to a few ulp; if that still clears the tie band of the matched target's
// new d^2, the match stands -- not even a tie is possible -- and no cell is visited.  Stored as a float rounded DOWN in the
// upper half of the row's B record's last word (the lower half is the index); 0 = no certificate.  (A distance rather than
// its square, which round 2 stored: the check that every row runs every iteration then needs no square root for it.)
__device__ __forceinline__ float float_below(double v)
{
    if (!(v > 0.0)) return 0.0f;
    float f = (float)v;
    if ((double)f > v) f = __uint_as_float(__float_as_uint(f) - 1u);   // positive: the next float towards zero (inf -> FLT_MAX)
    return f;
}
__device__ __forceinline__ double pack_idx_cert(uint32_t j, float cert)
{
    return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(cert) << 32) | (unsigned long long)j));
}

// number of non-empty cells (added to *out)
__global__ __launch_bounds__(kWG) void grid_occupied_kernel(const uint32_t *__restrict__ counts, size_t cells,
                                                            unsigned long long *__restrict__ out)
{
    unsigned long long mine = 0;
    for (size_t c = (size_t)blockIdx.x * kWG + threadIdx.x; c < cells; c += (size_t)gridDim.x * kWG) mine += counts[c] != 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// The source centroid of a search launch: read from pc_io, or (cen_part != nullptr, small clouds) finished here by every
// workgroup for itself -- the fixed-order sum of the chunk partials that finalize_centroid_kernel would run as a launch of its
// own (3.7 us of a 30 us iteration); identical code on identical inputs, so every workgroup holds the same bits, and
// workgroup 0 publishes them for the kernels that follow.  Every thread of the workgroup must call it (barriers).
template <int BATCH = kSumGroup>   // loads in flight per thread while the partials are summed (registers)
__device__ __forceinline__ void grid_launch_centroid(double *__restrict__ pc_io, const double *__restrict__ cen_part, unsigned cen_chunks,
                                                     double cen_denom, double pc[3])
{
    if (cen_part) {
        __shared__ double s_stage[kWG];
        __shared__ double s_pc[3];
        const double acc = ordered_sum<kCenW, kWG, BATCH>(cen_part, cen_chunks, s_stage);
        if (threadIdx.x < 3) {
            s_pc[threadIdx.x] = acc / cen_denom;
            if (blockIdx.x == 0) pc_io[threadIdx.x] = acc / cen_denom;
        }
        __syncthreads();
        pc[0] = s_pc[0]; pc[1] = s_pc[1]; pc[2] = s_pc[2];
    } else {
        pc[0] = pc_io[0]; pc[1] = pc_io[1]; pc[2] = pc_io[2];
    }
}
// "The registration has settled": no point of the target's bounding volume moved by more than order_switch cell half-edges in
// the motion that mot records (the solve record of the last update; pc = the centroid after it).  Decided from the motion
// record alone, so every lane of a launch -- and the host, one launch later -- sees the same answer.
__device__ __forceinline__ bool grid_settled(const GridDesc &g, const double *__restrict__ mot, const double pc[3], double order_switch)
{
    const double o0 = mot[25], o1 = mot[26], o2 = mot[27];
    const double c0 = ((mot[12] * o0 + mot[13] * o1) + mot[14] * o2) + mot[21] - pc[0];
    const double c1 = ((mot[15] * o0 + mot[16] * o1) + mot[17] * o2) + mot[22] - pc[1];
    const double c2 = ((mot[18] * o0 + mot[19] * o1) + mot[20] * o2) + mot[23] - pc[2];
    double rot = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double e = mot[12 + k] - ((k % 4 == 0) ? 1.0 : 0.0);
        rot += e * e;
    }
    const double ext = (g.hi[0] - g.lo[0]) + (g.hi[1] - g.lo[1]) + (g.hi[2] - g.lo[2]);
    const double moved = sqrt(rot) * ext + ((__builtin_fabs(c0) + __builtin_fabs(c1)) + __builtin_fabs(c2));
    return moved <= order_switch * g.r0;   // NaN: not settled
}

// One lane per source point.  A lane whose box is large (more than row_cap rows of cells or eval_cap targets -- an
// outlier far from every target, or an un-seeded first pass) does not walk it alone while the other 63 lanes of its
// wave wait: it records what it has found so far and queues the point for search_grid_overflow_kernel.
// What every lane of a grid search needs besides its own row: the launch's arguments and the source centroid.
struct GridCtx {
    const double *px, *py, *pz;
    GridDesc g;
    const uint32_t *start;
    const double4 *rec;
    const double *qx, *qy, *qz;
    const double4 *qrec4;
    unsigned m;
    const uint32_t *idx_prev;
    double4 *prec;
    size_t prec_stride;
    int row_cap;
    unsigned eval_cap;
    int defer_always;
    int issue_cert;   // this search leaves a certificate with every row it walks (whether it may CHECK the previous search's is `mot`)
    uint32_t *ovf_count, *ovf_list;
    double2 *out_rec;
    const double *mot;
    unsigned long long *stat;
    double pc[3];
};
// One source row on its way through a search: where it is (centred), its seed / match so far, the smallest d^2 known.
struct GridRow {
    size_t i;
    double ax, ay, az, sqx, sqy, sqz, dmin, edge, amax;
    uint32_t js, bj;
};
#define GRID_CTX_LOCALS(c)                                                                                             \
    const double *__restrict__ px = (c).px, *__restrict__ py = (c).py, *__restrict__ pz = (c).pz, *pc = (c).pc;      \
    const GridDesc &g = (c).g;                                                                                         \
    const uint32_t *__restrict__ start = (c).start;                                                                    \
    const double4 *__restrict__ rec = (c).rec;                                                                         \
    const double *__restrict__ qx = (c).qx, *__restrict__ qy = (c).qy, *__restrict__ qz = (c).qz;                     \
    const double4 *__restrict__ qrec4 = (c).qrec4;                                                                     \
    const unsigned m = (c).m;                                                                                          \
    const uint32_t *__restrict__ idx_prev = (c).idx_prev;                                                              \
    double4 *__restrict__ prec = (c).prec;                                                                             \
    const size_t prec_stride = (c).prec_stride;                                                                        \
    const int row_cap = (c).row_cap, defer_always = (c).defer_always;                                                  \
    const unsigned eval_cap = (c).eval_cap;                                                                            \
    uint32_t *__restrict__ ovf_count = (c).ovf_count, *__restrict__ ovf_list = (c).ovf_list;                          \
    double2 *__restrict__ out_rec = (c).out_rec;                                                                       \
    const double *__restrict__ mot = (c).mot;                                                                          \
    unsigned long long *__restrict__ stat = (c).stat;                                                                  \
    (void)px; (void)py; (void)pz; (void)pc; (void)g; (void)start; (void)rec; (void)qx; (void)qy; (void)qz; (void)qrec4; (void)m; \
    (void)idx_prev; (void)prec; (void)prec_stride; (void)row_cap; (void)defer_always; (void)eval_cap; (void)ovf_count;         \
    (void)ovf_list; (void)out_rec; (void)mot; (void)stat
#define GRID_ROW_LOCALS(r)                                                                                            \
    const size_t i = (r).i;                                                                                            \
    double &ax = (r).ax, &ay = (r).ay, &az = (r).az, &sqx = (r).sqx, &sqy = (r).sqy, &sqz = (r).sqz, &dmin = (r).dmin, \
           &edge = (r).edge, &amax = (r).amax;                                                                         \
    uint32_t &js = (r).js, &bj = (r).bj;                                                                                \
    (void)js

// Everything that precedes the walk over the cells, for row r.i: true = the row has to walk (its state is in r), false = its
// result is written (certificate held, or no finite coordinates).  With source records the caller passes the row's two
// halves, v0 = A[i] and v1 = B[i].
__device__ __forceinline__ bool grid_row_setup(const GridCtx &ctx, GridRow &row, const double4 v0, const double4 v1)
{
    GRID_CTX_LOCALS(ctx);
    GRID_ROW_LOCALS(row);
    double cert = 0.0, pl1 = 0.0, delta = 0.0;   // delta: bound on how far the row moved (centred coordinates) since the previous search; 0 = unknown
    (void)delta;
    sqx = sqy = sqz = 0.0;
    js = 0xFFFFFFFFu;
    cert = 0.0; pl1 = 0.0;   // last iteration's certificate for this row; |p|_1 of the un-centred row
    if (prec) {
        // v0 = A[i]: the row; v1 = B[i]: its match, index | certificate (loaded by the caller)
        ax = v0.x - pc[0]; ay = v0.y - pc[1]; az = v0.z - pc[2];
        const unsigned long long w1 = (unsigned long long)__double_as_longlong(v1.w);
        if (idx_prev) js = (uint32_t)w1;   // (B is kept current by whatever searched last: this kernel or merge_cov)
        sqx = v1.x; sqy = v1.y; sqz = v1.z;
        if (idx_prev && mot) {
            cert = (double)__uint_as_float((uint32_t)(w1 >> 32));   // issued with this index by the search that wrote B
            pl1 = (__builtin_fabs(v0.x) + __builtin_fabs(v0.y)) + __builtin_fabs(v0.z);
        }
    } else {
        ax = px[i] - pc[0]; ay = py[i] - pc[1]; az = pz[i] - pc[2];
        if (idx_prev) js = idx_prev[i];
    }
    dmin = INFINITY;   // oracle: min_dist = DBL_MAX on the key, idx = -1
    bj = 0xFFFFFFFFu;
    amax = __builtin_fmax(__builtin_fmax(__builtin_fabs(ax), __builtin_fabs(ay)), __builtin_fabs(az));
    if (!(amax < INFINITY) || ax != ax || ay != ay || az != az) {   // every d^2 is inf or NaN: no correspondence
        if (prec) prec[prec_stride + i] = make_double4(0.0, 0.0, 0.0, __longlong_as_double((long long)bj));
        else out_rec[i] = make_double2(INFINITY, __longlong_as_double((long long)bj));
        return false;
    }
    if (!prec && js < m) {   // the coordinates of the seed
        sqx = qrec4 ? qrec4[js].x : qx[js];
        sqy = qrec4 ? qrec4[js].y : qy[js];
        sqz = qrec4 ? qrec4[js].z : qz[js];
    }
    if (js < m) {
        const double d = d2_exact(ax, ay, az, sqx, sqy, sqz);
        if (d < INFINITY) {
            dmin = d;
            bj = js;
        }
    }
    edge = dmin < INFINITY ? tie_band(dmin) : INFINITY;
    delta = 0.0;
    if (mot && idx_prev && bj == js && js < m) {
        // The engine moved every row by p' = fl(R p + t) and re-centred: a' = R a + c + e with c = R Pc + t - Pc' and e the
        // rounding of the transform and of the two centrings (<= 8u of the magnitudes involved; 2^-48 = 32u is charged).
        // R is orthogonal up to the defect the solve kernel measured (mot[kSolveW + 3] = |R^T R - I|_F, required < 2^-42), so
        // a = R^T (a' - c - e) up to 2^-40 (|a'| + |c|), and the row moved by  a' - a = (I - R^T) a' + R^T (c + e):
        //     |a' - a| <= |(I - R^T) a'| + |c| + eps.
        // mot = the solve record of that motion: R at [12..20], t at [21..23], the old Pc at [25..27]; pc = the new centroid.
        const double o0 = mot[25], o1 = mot[26], o2 = mot[27], t0 = mot[21], t1 = mot[22], t2 = mot[23];
        const double c0 = ((mot[12] * o0 + mot[13] * o1) + mot[14] * o2) + t0 - pc[0];
        const double c1 = ((mot[15] * o0 + mot[16] * o1) + mot[17] * o2) + t1 - pc[1];
        const double c2 = ((mot[18] * o0 + mot[19] * o1) + mot[20] * o2) + t2 - pc[2];
        const double cn = (__builtin_fabs(c0) + __builtin_fabs(c1)) + __builtin_fabs(c2);          // >= |c|_2
        const double mag = pl1 + ((__builtin_fabs(t0) + __builtin_fabs(t1)) + __builtin_fabs(t2)) +
                           ((__builtin_fabs(o0) + __builtin_fabs(o1)) + __builtin_fabs(o2)) +
                           ((__builtin_fabs(pc[0]) + __builtin_fabs(pc[1])) + __builtin_fabs(pc[2]));
        const double al1 = (__builtin_fabs(ax) + __builtin_fabs(ay)) + __builtin_fabs(az);          // >= |a'|_2
        // (I - R^T) a': row k of R^T is column k of R
        const double m0 = ax - ((mot[12] * ax + mot[15] * ay) + mot[18] * az);
        const double m1 = ay - ((mot[13] * ax + mot[16] * ay) + mot[19] * az);
        const double m2 = az - ((mot[14] * ax + mot[17] * ay) + mot[20] * az);
        if (mot[kSolveW + 3] < 0x1p-42) {
            // |m|_2 from above: the 1-norm always is; the hardware's approximate square root (v_sqrt_f64: 2^29 ulp, i.e. 2^-23
            // relative, by its specification; 2^-18 is charged) of the squared norm is the better bound wherever that sum is
            // far from underflow.  No correctly rounded root is needed for a bound, and this one costs one instruction where
            // the library's costs fifty -- on every row, every iteration.
            const double mm = (m0 * m0 + m1 * m1) + m2 * m2;
            const double m_l1 = (__builtin_fabs(m0) + __builtin_fabs(m1)) + __builtin_fabs(m2);
            const double m_l2 = __builtin_amdgcn_sqrt(mm) * (1.0 + 0x1p-18);
            const double mn = mm > 0x1p-900 ? __builtin_fmin(m_l1, m_l2) : m_l1;
            delta = (mn + cn) * (1.0 + 0x1p-40) + 0x1p-40 * (al1 + cn) + 0x1p-48 * mag;
            if (cert > 0.0) {
                const double room = cert * (1.0 - 0x1p-40) - delta;   // every other target is at least this far from a'
                const double lower = room > 0.0 ? (room * room) * (1.0 - 0x1p-40) : 0.0;   // ... so its new computed d^2 is above this
                if (lower > edge) {
                    prec[prec_stride + i] = make_double4(sqx, sqy, sqz, pack_idx_cert(js, float_below(room * (1.0 - 0x1p-40))));
                    return false;
                }
            }
        }
    }
    return true;
}

// The walk over the cells for a row that grid_row_setup() left to walk; ends with the row's result written.
template <int ROWS, int FLIGHT>   // rows of cells per group (their runs are walked as one sequence); record loads in flight per lane
__device__ __forceinline__ void grid_row_walk(const GridCtx &ctx, GridRow &row)
{
    GRID_CTX_LOCALS(ctx);
    GRID_ROW_LOCALS(row);
    double r = first_radius(g, ax, ay, az, amax, edge);
    unsigned evals = 0;
    bool defer = false;
    // Keys (two square roots each) are not needed to find the winner of a pass: d1 / d2 = the smallest and the second
    // smallest d^2 met, k1 = where the smallest sits.  With d2 outside the tie band of d1 that target has the strictly
    // smallest key; only otherwise (near-ties, duplicated targets, lattices) the box is walked once more, comparing
    // (key, index) among the targets inside the band -- which are the only ones that can hold the smallest key.
    double d1 = INFINITY, d2 = INFINITY, Bfinal = 0.0;
    uint32_t k1 = 0;
    Box b;
    for (;;) {
        b = make_box(g, ax, ay, az, r);
        d1 = INFINITY;     // every pass meets every target of its box exactly once, the seed included
        d2 = INFINITY;
        Bfinal = b.all ? INFINITY : b.B;
        if (ovf_list) {
            // A large box is handed over when it would keep this lane busy while most of its wave idles (an outlier);
            // when most lanes of the wave have large boxes (an un-seeded pass over a scan, whose cells are mostly empty)
            // they all walk their own, up to 16 times the caps.
            const long rows = (long)(b.cy1 - b.cy0 + 1) * (long)(b.cz1 - b.cz0 + 1);
            const bool big = rows > (long)row_cap;
            const int n_big = __popcll(__ballot(big)), n_act = __popcll(__ballot(true));
            if (big && (defer_always || rows > 16L * row_cap || 4 * n_big < n_act)) {
                defer = true;
                break;
            }
        }
        // The cells cx0..cx1 of one (cy, cz) row are adjacent in the sorted arrays: a row is one run [k0, k1) of target
        // records.  Rows are taken four at a time -- their eight cell-start loads go out together -- and the runs of a group
        // are walked as ONE flattened sequence, four record loads in flight per lane: the wave then iterates max-over-lanes
        // of (candidates / 4) times instead of once per candidate of the longest run of every row, which is what bounds
        // this latency-bound kernel.  The visiting order changes nothing.
        int cy = b.cy0, cz = b.cz0;
        while (cz <= b.cz1 && !defer) {
            uint32_t kb[ROWS], ce[ROWS], total = 0;
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                uint32_t a0 = 0, a1 = 0;
                if (cz <= b.cz1) {
                    const size_t row = ((size_t)cz * (size_t)g.dim[1] + (size_t)cy) * (size_t)g.dim[0];
                    a0 = start[row + b.cx0];
                    a1 = start[row + b.cx1 + 1];
                    if (++cy > b.cy1) { cy = b.cy0; ++cz; }
                }
                kb[u] = a0;
                total += a1 - a0;
                ce[u] = total;
            }
            evals += total;
            if (ovf_list && evals > eval_cap) {
                const int n_big = __popcll(__ballot(true));   // lanes of the wave that are past the cap here, roughly
                if (defer_always || evals > 16u * eval_cap || 4 * n_big < 64) {
                    defer = true;
                    break;
                }
            }
            // position t of the flattened sequence -> sorted position: t + (offset of the run that holds t); selects, no branches
            uint32_t o[ROWS];
            o[0] = kb[0];
#pragma unroll
            for (int u = 1; u < ROWS; ++u) o[u] = kb[u] - ce[u - 1];
            for (uint32_t t = 0; t < total; t += FLIGHT) {
                double2 vxy[FLIGHT];   // the coordinates only (24 of the record's 32 bytes): the index is read for the winner alone
                double vz[FLIGHT];
                uint32_t kk[FLIGHT];
#pragma unroll
                for (int w = 0; w < FLIGHT; ++w) {
                    const uint32_t tt = (t + w < total) ? t + w : t;   // past the end: a record that is met anyway (masked below)
                    uint32_t off = o[ROWS - 1];
#pragma unroll
                    for (int u = ROWS - 2; u >= 0; --u) off = tt < ce[u] ? o[u] : off;
                    kk[w] = tt + off;
                    const double *rp = (const double *)&rec[kk[w]];
                    vxy[w] = *(const double2 *)rp;
                    vz[w] = rp[2];
                }
#pragma unroll
                for (int w = 0; w < FLIGHT; ++w) {
                    const double dw = d2_exact(ax, ay, az, vxy[w].x, vxy[w].y, vz[w]);
                    const double d = (w == 0 || t + w < total) ? dw : INFINITY;
                    d2 = __builtin_fmin(d2, __builtin_fmax(d, d1));
                    k1 = d < d1 ? kk[w] : k1;
                    d1 = __builtin_fmin(d1, d);
                }
            }
        }
        if (defer) break;
        if (d1 < INFINITY) {
            dmin = d1;
            edge = tie_band(d1);
        }
        if (b.all || b.B > edge) break;
        r = next_radius(edge, r);
    }
    float c = 0.0f;   // the certificate of this search (header of float_below): none for near-ties and deferred rows
    if (!defer && d1 < INFINITY) {
        if (d2 > edge) {
            const double4 v = rec[k1];   // just read: the unique minimum
            bj = (uint32_t)__double_as_longlong(v.w);
            sqx = v.x; sqy = v.y; sqz = v.z;
            if (ctx.issue_cert) c = float_below(sqrt(__builtin_fmin(d2, Bfinal)) * (1.0 - 0x1p-50));   // a distance: s^2 <= every other target's computed d^2
        } else {
            // near-ties: (key, index) among the targets inside the band, over the same box
            double bk = INFINITY;
            bj = 0xFFFFFFFFu;
            for (int cz = b.cz0; cz <= b.cz1; ++cz)
                for (int cy = b.cy0; cy <= b.cy1; ++cy) {
                    const size_t row = ((size_t)cz * (size_t)g.dim[1] + (size_t)cy) * (size_t)g.dim[0];
                    const uint32_t k0 = start[row + b.cx0], ke = start[row + b.cx1 + 1];
                    for (uint32_t k = k0; k < ke; ++k) {
                        const double4 v = rec[k];
                        const double d = d2_exact(ax, ay, az, v.x, v.y, v.z);
                        if (d <= edge) {
                            const double key = key_of(d);
                            const uint32_t j = (uint32_t)__double_as_longlong(v.w);
                            if (key < bk || (key == bk && j < bj)) {
                                bk = key;
                                bj = j;
                                sqx = v.x; sqy = v.y; sqz = v.z;
                            }
                        }
                    }
                }
        }
    }
    // the result (deferred: the seed, the overflow kernel starts from it): with source records, the matched target goes
    // into the second half of this row's own line -- the covariance kernel and the next search read it from there
    if (prec) prec[prec_stride + i] = make_double4(sqx, sqy, sqz, pack_idx_cert(bj, c));
    else out_rec[i] = make_double2(dmin, __longlong_as_double((long long)bj));
    if (defer) ovf_list[atomicAdd(ovf_count, 1u)] = (uint32_t)i;
    if (stat) atomicAdd(&stat[1], (unsigned long long)evals);
}

template <int ROWS, int FLIGHT, int WAVES>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void search_grid_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                          const double *__restrict__ pz, double *__restrict__ pc_io, size_t n,
                                                          GridDesc g, const uint32_t *__restrict__ start,
                                                          const double4 *__restrict__ rec,
                                                          const double *__restrict__ qx, const double *__restrict__ qy,
                                                          const double *__restrict__ qz, const double4 *__restrict__ qrec4,
                                                          unsigned m,
                                                          const uint32_t *__restrict__ idx_prev,
                                                          double4 *__restrict__ prec, size_t prec_stride,
                                                          const uint32_t *__restrict__ order,
                                                          int row_cap, unsigned eval_cap, int defer_always, int issue_cert,
                                                          uint32_t *__restrict__ ovf_count,
                                                          uint32_t *__restrict__ ovf_list, double2 *__restrict__ out_rec,
                                                          const double *__restrict__ mot, double order_switch,
                                                          unsigned long long *__restrict__ stat,
                                                          const double *__restrict__ cen_part, unsigned cen_chunks, double cen_denom,
                                                          int xcd_map, uint32_t *__restrict__ settled_hint)
{
    // Workgroups go to the eight XCDs round robin (workgroup b runs on XCD b % 8), and every XCD has an L2 of its own.  The
    // launch is a multiple of eight workgroups and XCD x takes the x-th EIGHTH of the rows -- in the spatial order a compact
    // slab of the cloud -- so the target records and cell starts an XCD's rows walk are an eighth of the target instead of
    // all of it (-2 % on an iteration while the clouds move).  xcd_map = 0 (ICP_AMD_GRID_XCD=0): workgroup b takes rows [256 b, 256 b + 256).
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned vblock = xcd_map ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    const size_t l = (size_t)vblock * kWG + threadIdx.x;
    if ((size_t)vblock * kWG >= n) return;   // the launch is rounded up to a multiple of eight (workgroup 0 always has rows)
    double pc[3];
    grid_launch_centroid(pc_io, cen_part, cen_chunks, cen_denom, pc);
    // order: source rows sorted by position, so that the lanes of a wave walk the same few cells (cache lines shared).
    // The rows themselves stay where they are (everything a lane needs of its row is its two record halves, when the caller
    // keeps records).  Once the registration has settled, most rows keep their match by certificate without visiting a cell,
    // and the kernel is a stream over the row records: the rows are then taken in storage order (coalesced).  Either order
    // gives the same results; this only picks the faster one.
    const bool settled = mot && idx_prev && grid_settled(g, mot, pc, order_switch);
    const bool spatial = order != nullptr && !settled;
    if (settled_hint && blockIdx.x == 0 && threadIdx.x == 0) *settled_hint = settled ? 1u : 0u;   // for the host's next launch
    const size_t i = spatial ? order[l < n ? l : n - 1] : l;
    GridCtx ctx;
    ctx.px = px; ctx.py = py; ctx.pz = pz; ctx.g = g; ctx.start = start; ctx.rec = rec; ctx.qx = qx; ctx.qy = qy; ctx.qz = qz;
    ctx.qrec4 = qrec4; ctx.m = m; ctx.idx_prev = idx_prev; ctx.prec = prec; ctx.prec_stride = prec_stride; ctx.row_cap = row_cap;
    ctx.eval_cap = eval_cap; ctx.defer_always = defer_always; ctx.issue_cert = issue_cert; ctx.ovf_count = ovf_count; ctx.ovf_list = ovf_list; ctx.out_rec = out_rec;
    ctx.mot = mot; ctx.stat = stat; ctx.pc[0] = pc[0]; ctx.pc[1] = pc[1]; ctx.pc[2] = pc[2];
    // Rows that walk are rare once certificates hold, but nearly every wave has a few, and a wave is as slow as its slowest
    // lane: the workgroup gathers its walking rows -- with the state their lane has already computed: reading the record and
    // deciding again was 4 us of an iteration while rows walk -- and packs them into as few waves as they need; the other
    // waves leave.  (Packing them in the order of their search radius, so that a wave's rows have boxes of about one size,
    // measured the same: gpurun r3q.)
    __shared__ GridRow s_row[kWG];
    __shared__ uint32_t s_walkers;
    if (threadIdx.x == 0) s_walkers = 0;
    __syncthreads();
    GridRow row;
    row.i = i;
    bool need = false;
    if (l < n) {
        double4 v0 = make_double4(0.0, 0.0, 0.0, 0.0), v1 = v0;
        if (prec) {
            v0 = prec[row.i];
            v1 = prec[prec_stride + row.i];
        }
        need = grid_row_setup(ctx, row, v0, v1);
    }
    if (need) s_row[atomicAdd(&s_walkers, 1u)] = row;
    __syncthreads();
    const uint32_t walkers = s_walkers;
    if (stat && threadIdx.x == 0) atomicAdd(&stat[0], (unsigned long long)walkers);
    const size_t left = n - (size_t)vblock * kWG;
    if (walkers + 64 <= (left < (size_t)kWG ? left : (size_t)kWG)) {   // packing frees at least one wave
        if (threadIdx.x >= walkers) return;
        row = s_row[threadIdx.x];   // (walking rows have written nothing yet)
        need = true;
    }
    if (need) grid_row_walk<ROWS, FLIGHT>(ctx, row);
}

// One row's contribution to the chunk's covariance sums, exactly as cov_rec_kernel (kernels.hip) forms it; stores the row's index.
__device__ __forceinline__ void cov_row_sums(unsigned m, double reject_d2, double p0, double p1, double p2, double q0, double q1, double q2,
                                             uint32_t j, uint32_t *__restrict__ idx_slot, double (&acc)[kCovW])
{
    *idx_slot = j;
    if (j < m) {
        const double d = d2_exact(p0, p1, p2, q0, q1, q2);
        if (d <= reject_d2) {
            acc[0] += q0 * p0; acc[1] += q0 * p1; acc[2] += q0 * p2;
            acc[3] += q1 * p0; acc[4] += q1 * p1; acc[5] += q1 * p2;
            acc[6] += q2 * p0; acc[7] += q2 * p1; acc[8] += q2 * p2;
            acc[9] += d;
            acc[10] += 1.0;
        }
    }
}

// The settled registration: (nearly) every row keeps its match by certificate and the search is a stream over the row
// records -- from whose registers the iteration's covariance can be had as well.  One workgroup per 1024-row chunk in
// storage order, four rows per lane with all eight record loads in flight, exactly the shape of cov_rec_kernel
// (kernels.hip), whose outputs this kernel writes too: idx_out and the chunk's partial sums, same arithmetic in the same
// order, bit for bit.  A launch of this kernel therefore replaces search_grid_kernel AND cov_rec_kernel (64 of an
// iteration's 228 bytes per row, and a launch) -- and, when cen_part is given, finalize_centroid_kernel as well.
// 128 registers = 4 waves per SIMD: the 977 workgroups of a 1M-row cloud are resident at once.  Rows whose certificate does not hold walk the cells as in
// search_grid_kernel (the workgroup's lanes share them out; nothing is handed to the overflow kernel: ctx.ovf_list is null),
// and the lanes that own those rows read the results back before the sums are formed.  The host launches it when the previous
// search reported a settled registration (a mapped word, possibly a launch old): the choice never changes a result.
template <int ROWS, int FLIGHT>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(4, 4))) void search_grid_settled_kernel(GridCtx ctx_in, double *__restrict__ pc_io, size_t n,
                                                                  const double *__restrict__ cen_part, unsigned cen_chunks, double cen_denom,
                                                                  double order_switch, uint32_t *__restrict__ settled_hint,
                                                                  uint32_t *__restrict__ idx_out, double reject_d2,
                                                                  double *__restrict__ partial)
{
    __shared__ double smem[4 * kCovW];
    __shared__ uint32_t s_walk[kChunk];
    __shared__ uint32_t s_walkers;
    GridCtx ctx = ctx_in;
    double4 *__restrict__ prec = ctx.prec;
    const size_t stride = ctx.prec_stride;
    const size_t base = (size_t)blockIdx.x * kChunk + threadIdx.x;
    constexpr int PER = kChunk / kWG;
    // The rows first: their eight loads are on their way while the workgroup finishes the centroid (a sum over partials that
    // sit in the L2 and two barriers -- 2 us that used to precede the first row load, or a launch of its own before this one).
    double4 v0[PER], v1[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const size_t i = base + (size_t)k * kWG, ic = i < n ? i : n - 1;   // rows past the end: a valid address, masked below
        v0[k] = prec[ic];
        v1[k] = prec[stride + ic];
    }
    if (threadIdx.x == 0) s_walkers = 0;
    grid_launch_centroid<16>(pc_io, cen_part, cen_chunks, cen_denom, ctx.pc);
    if (settled_hint && blockIdx.x == 0 && threadIdx.x == 0)
        *settled_hint = grid_settled(ctx.g, ctx.mot, ctx.pc, order_switch) ? 1u : 0u;
    __syncthreads();
    // cov_rec_kernel's sums, in its order, formed as the rows are decided: p = A - pc as cov_rec_kernel forms it, q = the
    // matched target, j = its index (certificate held: the seed; no finite coordinates: none).  Should a row of the workgroup
    // have to walk, these sums are dropped and formed again from the chunk as it is after the walk (below) -- keeping the
    // rows' operands in registers instead cost a wave per SIMD (148 registers against 128: 977 workgroups at 1M rows are then
    // two rounds over the 256 CUs instead of one).
    double acc[kCovW];
#pragma unroll
    for (int k = 0; k < kCovW; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        GridRow row;
        row.i = base + (size_t)k * kWG;
        if (row.i < n) {
            if (grid_row_setup(ctx, row, v0[k], v1[k])) s_walk[atomicAdd(&s_walkers, 1u)] = (uint32_t)row.i;
            cov_row_sums(ctx.m, reject_d2, row.ax, row.ay, row.az, row.sqx, row.sqy, row.sqz, row.bj, idx_out + row.i, acc);
        }
    }
    __syncthreads();
    const uint32_t walkers = s_walkers;   // the same in every lane
    if (ctx.stat && threadIdx.x == 0) atomicAdd(&ctx.stat[0], (unsigned long long)walkers);
    if (walkers) {
        for (uint32_t w = threadIdx.x; w < walkers; w += kWG) {
            GridRow wr;
            wr.i = s_walk[w];
            if (grid_row_setup(ctx, wr, prec[wr.i], prec[stride + wr.i])) grid_row_walk<ROWS, FLIGHT>(ctx, wr);
        }
        // The walked rows' results were written by lanes of THIS workgroup: the barrier (a release / acquire pair at workgroup
        // scope) orders them before the reads below.  No device-scope fence: on this part that is a write-back of the XCD's
        // dirty L2 lines -- every row the launch has written so far -- per workgroup that walked (measured: HISTORY.md, round 3).
        __syncthreads();
        // Nothing of the rows was kept in registers across the walk: a workgroup that had to walk reads its chunk again.
#pragma unroll
        for (int k = 0; k < kCovW; ++k) acc[k] = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const size_t i = base + (size_t)k * kWG;
            if (i < n) {
                const double4 a = prec[i];
                const double *b = (const double *)&prec[stride + i];
                const double q0 = __hip_atomic_load(b + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const double q1 = __hip_atomic_load(b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const double q2 = __hip_atomic_load(b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t j = (uint32_t)__hip_atomic_load((const unsigned long long *)(b + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                cov_row_sums(ctx.m, reject_d2, a.x - ctx.pc[0], a.y - ctx.pc[1], a.z - ctx.pc[2], q0, q1, q2, j, idx_out + i, acc);
            }
        }
    }
    const double tot = block_sum<kCovW>(acc, smem);
    if (threadIdx.x < kCovW) partial[(size_t)blockIdx.x * kCovW + threadIdx.x] = tot;
}

// The queued points, one workgroup each: the 256 lanes walk the box together (waves take rows of cells, lanes stride
// over a row's targets).  Every lane keeps the lexicographic best of its own candidates inside the tie band of its
// own running minimum; the target that wins overall is inside every lane's band when its lane meets it, so the
// lexicographic minimum over the lanes' results is the overall one.
__global__ __launch_bounds__(kWG) void search_grid_overflow_kernel(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz, const double *__restrict__ pc,
    GridDesc g, const uint32_t *__restrict__ start, const double4 *__restrict__ rec, const double *__restrict__ qx,
    const double *__restrict__ qy, const double *__restrict__ qz, const double4 *__restrict__ qrec4, unsigned m,
    const uint32_t *__restrict__ ovf_count,
    uint32_t *__restrict__ ovf_count_next, const uint32_t *__restrict__ ovf_list, uint32_t *__restrict__ host_count,
    double4 *__restrict__ prec, size_t prec_stride, double2 *__restrict__ out_rec)
{
    __shared__ double s_d[kWG / 64], s_k[kWG / 64];
    __shared__ uint32_t s_j[kWG / 64];
    const unsigned count = *ovf_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *ovf_count_next = 0;                  // the counter the next launch of search_grid_kernel appends to
        if (host_count) *host_count = count;  // a hint for the host's choice of variant, read without synchronisation
    }
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (unsigned q = blockIdx.x; q < count; q += gridDim.x) {
        const size_t i = ovf_list[q];
        // the row: from its record where there is one (the running cloud of that path; the SoA arrays may be behind)
        const double rx = prec ? prec[i].x : px[i], ry = prec ? prec[i].y : py[i], rz = prec ? prec[i].z : pz[i];
        const double ax = rx - pc[0], ay = ry - pc[1], az = rz - pc[2];
        const double amax = __builtin_fmax(__builtin_fmax(__builtin_fabs(ax), __builtin_fabs(ay)), __builtin_fabs(az));
        double bk = INFINITY, dmin = INFINITY;
        uint32_t bj = 0xFFFFFFFFu;
        const uint32_t js = (uint32_t)__double_as_longlong(prec ? prec[prec_stride + i].w : out_rec[i].y);
        if (js < m) {
            const double d = seed_d2(ax, ay, az, js, qx, qy, qz, qrec4);
            if (d < INFINITY) {
                dmin = d;
                bk = key_of(d);
                bj = js;
            }
        }
        double edge = dmin < INFINITY ? tie_band(dmin) : INFINITY;
        double r = first_radius(g, ax, ay, az, amax, edge);
        for (;;) {   // every quantity that steers this loop is uniform over the workgroup
            const Box b = make_box(g, ax, ay, az, r);
            const long ny = b.cy1 - b.cy0 + 1, rows = ny * (long)(b.cz1 - b.cz0 + 1);
            for (long rr = wave; rr < rows; rr += kWG / 64) {
                const int cz = b.cz0 + (int)(rr / ny), cy = b.cy0 + (int)(rr % ny);
                const size_t row = ((size_t)cz * (size_t)g.dim[1] + (size_t)cy) * (size_t)g.dim[0];
                const uint32_t k0 = start[row + b.cx0], k1 = start[row + b.cx1 + 1];
                for (uint32_t k = k0 + lane; k < k1; k += 64) consider_rec(ax, ay, az, rec[k], dmin, edge, bk, bj);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double d2 = __shfl_xor(dmin, o, 64), k2 = __shfl_xor(bk, o, 64);
                const uint32_t j2 = __shfl_xor(bj, o, 64);
                dmin = __builtin_fmin(dmin, d2);
                if (k2 < bk || (k2 == bk && j2 < bj)) { bk = k2; bj = j2; }
            }
            __syncthreads();   // the previous pass's reads of s_* are done
            if (lane == 0) { s_d[wave] = dmin; s_k[wave] = bk; s_j[wave] = bj; }
            __syncthreads();
            dmin = s_d[0]; bk = s_k[0]; bj = s_j[0];
#pragma unroll
            for (int w = 1; w < kWG / 64; ++w) {
                dmin = __builtin_fmin(dmin, s_d[w]);
                if (s_k[w] < bk || (s_k[w] == bk && s_j[w] < bj)) { bk = s_k[w]; bj = s_j[w]; }
            }
            edge = dmin < INFINITY ? tie_band(dmin) : INFINITY;
            if (b.all || b.B > edge) break;
            r = next_radius(edge, r);
        }
        if (threadIdx.x == 0) {
            if (prec) {
                double4 v = make_double4(0.0, 0.0, 0.0, 0.0);
                if (bj < m) v = qrec4 ? qrec4[bj] : make_double4(qx[bj], qy[bj], qz[bj], 0.0);
                prec[prec_stride + i] = make_double4(v.x, v.y, v.z, __longlong_as_double((long long)(unsigned long long)bj));
            } else {
                out_rec[i] = make_double2(bk, __longlong_as_double((long long)bj));
            }
        }
        __syncthreads();   // the row's B record / out_rec[i] may be read again if the queue lists the row twice (it does not), and
                           // the shared scratch is reused by the next queued row
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
__global__ void grid_warmup_kernel() {}
hipError_t grid_warmup(hipStream_t s)
{
    return launch(grid_warmup_kernel, dim3(1), dim3(64), 0, s);
}

void grid_free(Grid *g)
{
    if (!g) return;
    if (g->on_stream) {
        if (g->start) (void)dfree_on(g->start, g->stream);
        if (g->perm) (void)dfree_on(g->perm, g->stream);
        if (g->rec) (void)dfree_on(g->rec, g->stream);
    } else {
        if (g->start) (void)dfree(g->start);
        if (g->perm) (void)dfree(g->perm);
        if (g->rec) (void)dfree(g->rec);
    }
    *g = Grid{};
}

// Cell size: about per_cell points per cell of the bounding volume (area / length when the cloud is flat along some
// axes) times h_scale, at most 1024 cells per axis and kMaxCells cells in all.
constexpr double kMaxCells = 16777216.0;   // 2^24: 64 MB of cell starts
static void choose_cells(const double lo[3], const double hi[3], size_t m, double per_cell, double h_scale, GridDesc *d)
{
    double ext[3];
    int active = 0;
    double vol = 1.0;
    for (int k = 0; k < 3; ++k) {
        ext[k] = hi[k] - lo[k];
        if (ext[k] > 0.0) { ++active; vol *= ext[k]; }
    }
    double want = (double)m / per_cell;
    if (want < 1.0) want = 1.0;
    if (want > kMaxCells) want = kMaxCells;
    double h = active ? pow(vol / want, 1.0 / active) : 1.0;
    if (!(h > 0.0) || !(h < INFINITY)) {   // under/overflow of the volume: fall back to the longest axis
        const double longest = fmax(fmax(ext[0], ext[1]), ext[2]);
        h = longest > 0.0 ? longest / cbrt(want) : 1.0;
    }
    h *= h_scale;
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
        if (cells <= kMaxCells) break;
        h *= 1.26;
    }
    double hmin = INFINITY;
    for (int k = 0; k < 3; ++k) {
        d->lo[k] = lo[k];
        d->hi[k] = hi[k];
        d->inv[k] = ext[k] > 0.0 ? (double)d->dim[k] / ext[k] : 0.0;
        if (!(d->inv[k] < INFINITY)) { d->inv[k] = 0.0; d->dim[k] = 1; }   // extent so small that dim/ext overflows
        d->h[k] = d->dim[k] > 1 ? ext[k] / d->dim[k] : 0.0;
        if (d->dim[k] > 1) hmin = fmin(hmin, ext[k] / d->dim[k]);
    }
    d->r0 = hmin < INFINITY ? 0.5 * hmin : 1.0;
}

// Files the points into cells: g->start, g->perm and (with_coords) the sorted coordinates.
static hipError_t grid_sort(hipStream_t s, Soa3 Qz, size_t m, double per_cell, bool with_coords, bool refine, Grid *g)
{
    grid_free(g);
    if (m == 0 || m > 0xFFFFFFF0u) return hipErrorInvalidValue;
    g->stream = s;   // everything that touches this grid is enqueued on s: releasing it waits for s only
    g->on_stream = true;
    hipError_t e;
    unsigned long long *bb = nullptr;
    uint32_t *cell = nullptr, *counts = nullptr, *cursor = nullptr, *block_sums = nullptr;
    unsigned long long host_bb[6];
#define GRID_TRY(expr) do { e = (expr); if (e != hipSuccess) goto done; } while (0)
    GRID_TRY(dmalloc(&bb, 6 * sizeof(unsigned long long)));
    GRID_TRY(hipMemsetAsync(bb, 0xFF, 3 * sizeof(unsigned long long), s));
    GRID_TRY(hipMemsetAsync(bb + 3, 0x00, 3 * sizeof(unsigned long long), s));
    {
        size_t blocks = (m + kWG - 1) / kWG;
        if (blocks > 512) blocks = 512;   // every workgroup ends in six atomics on the same six words, and those serialise
        GRID_TRY(launch(grid_bbox_kernel, dim3((unsigned)blocks), dim3(kWG), 0, s, Qz.x, Qz.y, Qz.z, m, bb));
    }
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
        if (!finite) {   // NaN / inf coordinates: no grid
            g->ok = false;
            e = hipSuccess;
            goto done;
        }
        // Scans are surfaces: most cells of the bounding volume stay empty and the occupied ones are fuller than
        // planned.  Count the occupied cells and shrink the cell edge until they hold about per_cell points on average
        // (assuming the occupied count grows like 1/h^2; volume-filling clouds pass at once), within the cell budget.
        GRID_TRY(dmalloc(&cell, m * sizeof(uint32_t)));
        double h_scale = 1.0;
        for (int round = 0;; ++round) {
            choose_cells(lo, hi, m, per_cell, h_scale, &g->d);
            g->cells = (size_t)g->d.dim[0] * (size_t)g->d.dim[1] * (size_t)g->d.dim[2];
            GRID_TRY(dmalloc(&counts, g->cells * sizeof(uint32_t)));
            GRID_TRY(hipMemsetAsync(counts, 0, g->cells * sizeof(uint32_t), s));
            GRID_TRY(launch(grid_count_kernel, dim3((unsigned)((m + kWG - 1) / kWG)), dim3(kWG), 0, s, Qz.x, Qz.y, Qz.z, m, g->d, cell, counts));
            if (!refine || round == 3 || (double)g->cells > kMaxCells / 2.0) break;
            unsigned long long occupied = 0;
            GRID_TRY(hipMemsetAsync(bb, 0, sizeof(unsigned long long), s));
            GRID_TRY(launch(grid_occupied_kernel, dim3(256), dim3(kWG), 0, s, counts, g->cells, bb));
            GRID_TRY(hipMemcpyAsync(&occupied, bb, sizeof occupied, hipMemcpyDeviceToHost, s));
            GRID_TRY(hipStreamSynchronize(s));
            const double avg = (double)m / (double)(occupied ? occupied : 1);
            if (avg <= 2.0 * per_cell) break;
            double f = sqrt(per_cell / avg);
            h_scale *= f < 0.25 ? 0.25 : f;
            (void)dfree_on(counts, s);
            counts = nullptr;
        }
    }
    GRID_TRY(dmalloc(&g->start, (g->cells + 1) * sizeof(uint32_t)));
    GRID_TRY(dmalloc(&g->perm, m * sizeof(uint32_t)));
    if (with_coords) GRID_TRY(dmalloc(&g->rec, m * sizeof(double4)));
    GRID_TRY(dmalloc(&cursor, g->cells * sizeof(uint32_t)));
    if (g->cells <= 65536) {
        GRID_TRY(launch(grid_scan_kernel, dim3(1), dim3(1024), 0, s, counts, g->cells, g->start, cursor));
    } else {
        const size_t blocks = (g->cells + kScanPerBlock - 1) / kScanPerBlock;   // <= 4096
        GRID_TRY(dmalloc(&block_sums, 3 * (blocks + 1) * sizeof(uint32_t)));
        uint32_t *block_off = block_sums + blocks + 1, *unused_copy = block_off + blocks + 1;
        GRID_TRY(launch(grid_scan_sums_kernel, dim3((unsigned)blocks), dim3(kWG), 0, s, counts, g->cells, block_sums));
        GRID_TRY(launch(grid_scan_kernel, dim3(1), dim3(1024), 0, s, block_sums, blocks, block_off, unused_copy));
        GRID_TRY(launch(grid_scan_apply_kernel, dim3((unsigned)blocks), dim3(kWG), 0, s, counts, g->cells, block_off, g->start, cursor));
    }
    GRID_TRY(launch(grid_scatter_kernel, dim3((unsigned)((m + kWG - 1) / kWG)), dim3(kWG), 0, s, Qz.x, Qz.y, Qz.z, m, cell, cursor, g->rec, g->perm));
    GRID_TRY(hipStreamSynchronize(s));   // the temporaries are freed below
    g->ok = true;
#undef GRID_TRY
done:
    if (bb) (void)dfree_on(bb, s);
    if (cell) (void)dfree_on(cell, s);
    if (counts) (void)dfree_on(counts, s);
    if (cursor) (void)dfree_on(cursor, s);
    if (block_sums) (void)dfree_on(block_sums, s);
    if (e != hipSuccess) grid_free(g);
    return e;
}

static double per_cell_setting(double dflt)
{
    const double x = knob_double("grid_per_cell", 0.0);
    return x >= 0.25 && x <= 65536.0 ? x : dflt;
}

hipError_t grid_build(hipStream_t s, Soa3 Qz, size_t m, Grid *g)
{
    return grid_sort(s, Qz, m, per_cell_setting(2.0), true, knob_int("grid_refine", 1) != 0, g);
}

__global__ __launch_bounds__(kWG) void iota_kernel(uint32_t *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

// order[0, n) = the source rows sorted by the cell of a coarse grid over the source cloud itself (about two waves of
// points per cell, cells in x-fastest order).  Rigid motions keep neighbours together, so the order computed once
// stays good for every iteration.  Rows with non-finite coordinates anywhere: identity order.
// One wave per cell: the cell's rows into ascending index (each finds its rank by counting the smaller ones).  Cells of more
// than kOrderRunMax rows are left as they are and reported.
constexpr unsigned kOrderRunMax = 4096;
__global__ __launch_bounds__(64) void grid_order_stable_kernel(const uint32_t *__restrict__ start, uint32_t *__restrict__ perm,
                                                               uint32_t *__restrict__ too_long)
{
    __shared__ uint32_t v[kOrderRunMax];
    const uint32_t s0 = start[blockIdx.x], k = start[blockIdx.x + 1] - s0;
    if (k < 2) return;
    if (k > kOrderRunMax) {
        if (threadIdx.x == 0) *too_long = 1u;   // (every writer stores the same value)
        return;
    }
    for (uint32_t i = threadIdx.x; i < k; i += 64) v[i] = perm[s0 + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < k; i += 64) {
        const uint32_t mine = v[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < k; ++j) rank += v[j] < mine ? 1u : 0u;
        perm[s0 + rank] = mine;
    }
}

hipError_t grid_order(hipStream_t s, Soa3 P, size_t n, uint32_t *order, bool *stable)
{
    if (stable) *stable = false;
    if (n == 0) return hipSuccess;
    Grid g;
    hipError_t e = grid_sort(s, P, n, 128.0, false, false, &g);
    if (e != hipSuccess) return e;
    uint32_t *flag = nullptr;
    uint32_t host_flag = 1;
    if (g.ok) {
        if (stable && g.cells <= 0x7FFFFFFFu) {
            e = dmalloc(&flag, sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(uint32_t), s);
            if (e == hipSuccess) e = launch(grid_order_stable_kernel, dim3((unsigned)g.cells), dim3(64), 0, s, g.start, g.perm, flag);
            if (e == hipSuccess) e = hipMemcpyAsync(&host_flag, flag, sizeof host_flag, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(order, g.perm, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
    } else {
        e = launch(iota_kernel, dim3((unsigned)((n + kWG - 1) / kWG)), dim3(kWG), 0, s, order, n);
        host_flag = 0;   // the identity is as stable as it gets
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && stable) *stable = host_flag == 0;
    if (flag) (void)dfree_on(flag, s);
    grid_free(&g);
    return e;
}

constexpr unsigned kStatSlots = 4096;
hipError_t search_grid(hipStream_t s, Soa3 P, size_t n, double *pc, const Grid &g, Soa3 Qz, size_t m,
                       const uint32_t *idx_prev, const uint32_t *order, GridScratch *sc, double2 *out_rec, const double *motion,
                       const double *cen_part, size_t cen_chunks, double cen_denom, GridFused *fused)
{
    if (fused) fused->ran = false;
    if (!g.ok || m == 0 || m > 0xFFFFFFF0u) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    const double order_switch = sc ? sc->order_switch : 0.125;   // read from the environment once, when the scratch was allocated
    if (sc && sc->stat && sc->stat_slot >= kStatSlots) {         // a diagnostics slot that is used again starts from zero
        hipError_t ez = hipMemsetAsync(sc->stat + 2 * (sc->stat_slot % kStatSlots), 0, 2 * sizeof(unsigned long long), s);
        if (ez != hipSuccess) return ez;
    }
    GridScratch *all = sc;   // the hints and the diagnostics do not depend on whether this launch may hand rows over
    if (sc && !sc->enabled) sc = nullptr;
    const int row_cap = sc ? sc->row_cap : 64, eval_cap = sc ? sc->eval_cap : 768;
    uint32_t *cnt = sc ? sc->count2 + sc->parity : nullptr, *cnt_next = sc ? sc->count2 + (sc->parity ^ 1) : nullptr;
    unsigned long long *stat = (all && all->stat) ? all->stat + 2 * (all->stat_slot++ % kStatSlots) : nullptr;
    uint32_t *hint = (all && all->host_count) ? all->host_count + 1 : nullptr;
    if (!P.rec) motion = nullptr;   // certificates live in the source records
    // A search may CHECK the certificates of the previous one only when it knows the motion in between (`motion`); it ISSUES
    // them whenever the caller keeps row records and has not switched them off -- also the first, un-seeded search of a
    // source cloud, so that the second one already finds three quarters of its rows certified.
    const int issue = (P.rec && all && all->issue_cert) ? 1 : 0;
    // Rows per group x loads in flight: 4 x 6 at 4 waves per SIMD.  Round 2 measured 4 x 4 (5 waves) against 4 x 8, 8 x 4 and 8 x 8
    // with whole 32-byte records in flight (profiles/r2_v2_grid_shape_ab.txt): eight loads were 9-12 % slower everywhere
    // (registers: 4 or 3 waves per SIMD, more masked slots), eight rows per group equal on volumes.  With only the 24 bytes of
    // coordinates in flight per candidate (round 3), six loads fit 110 registers and are 2-3.5 % faster than four while rows
    // walk; eight (126 registers) measure between the two (profiles/r3_flight_ab.txt).  ICP_AMD_DEBUG=grid_flight=4|6|8.
    const bool settled_launch = fused && fused->idx_out && fused->partial && all && all->fuse_cov && hint && all->host_mem[1] == 1u &&
                                P.rec && idx_prev && motion;
    // cen_part: the launch is asked to finish the centroid itself.  The settled kernel does that behind its row loads for any
    // cloud whose partials one pass of the sum covers; search_grid_kernel, whose workgroups are a quarter of the rows and
    // four times as many, only for small clouds -- otherwise the sum runs as the launch of its own that it would have been.
    if (cen_part && cen_chunks > (settled_launch ? kGridFuseCentroidChunksSettled : kGridFuseCentroidChunks)) {
        hipError_t ec = finalize_centroid(s, cen_part, cen_chunks, cen_denom, pc);
        if (ec != hipSuccess) return ec;
        cen_part = nullptr;
    }
    if (settled_launch) {
        GridCtx c{};
        c.px = P.x; c.py = P.y; c.pz = P.z; c.g = g.d; c.start = g.start; c.rec = g.rec; c.qx = Qz.x; c.qy = Qz.y; c.qz = Qz.z;
        c.qrec4 = Qz.rec; c.m = (unsigned)m; c.idx_prev = idx_prev; c.prec = P.rec; c.prec_stride = P.rec_stride; c.row_cap = row_cap;
        c.eval_cap = (unsigned)eval_cap; c.defer_always = 0; c.issue_cert = 1; c.ovf_count = nullptr; c.ovf_list = nullptr; c.out_rec = out_rec;
        c.mot = motion; c.stat = stat;
        hipError_t e = launch(search_grid_settled_kernel<4, 4>, dim3((unsigned)((n + kChunk - 1) / kChunk)), dim3(kWG), 0, s, c, pc, n,
                              cen_part, (unsigned)cen_chunks, cen_denom, order_switch, hint, fused->idx_out, fused->reject_d2,
                              fused->partial);
        fused->ran = e == hipSuccess;
        return e;   // nothing was handed to the overflow kernel: its counters are as the last launch left them (zero)
    }
    static const int xcd_map = knob_int("grid_xcd", 1);
    const size_t wgs_rows = (n + kWG - 1) / kWG;
    const int flight = all ? all->flight : 6;   // read when the scratch was allocated (grid_flight)
    auto kern = flight >= 8 ? search_grid_kernel<4, 8, 4> : flight >= 6 ? search_grid_kernel<4, 6, 4> : search_grid_kernel<4, 4, 5>;
    hipError_t e = launch(kern, dim3((unsigned)(xcd_map ? (wgs_rows + 7) / 8 * 8 : wgs_rows)), dim3(kWG), 0, s, P.x, P.y,
                          P.z, pc, n, g.d, g.start, g.rec, Qz.x, Qz.y, Qz.z, Qz.rec, (unsigned)m, idx_prev, P.rec, P.rec_stride, order,
                          row_cap, (unsigned)eval_cap, sc ? sc->defer_always : 0, issue, cnt, sc ? sc->list : nullptr, out_rec, motion,
                          order_switch, stat, cen_part, (unsigned)cen_chunks, cen_denom, xcd_map, hint);
    if (e != hipSuccess || !sc) return e;
    size_t wgs = n < 512 ? n : 512;   // two per CU walk the queue; an empty queue (the usual case) costs the launch only
    e = launch(search_grid_overflow_kernel, dim3((unsigned)wgs), dim3(kWG), 0, s, P.x, P.y, P.z, pc, g.d, g.start, g.rec, Qz.x, Qz.y, Qz.z,
               Qz.rec, (unsigned)m, cnt, cnt_next, sc->list, sc->host_count, P.rec, P.rec_stride, out_rec);
    sc->parity ^= 1;
    return e;
}

hipError_t grid_scratch_alloc(hipStream_t s, size_t n, GridScratch *sc)
{
    grid_scratch_free(sc);
    // what one lane may walk on its own before it hands its point to the overflow kernel (tunable for tests)
    sc->row_cap = knob_int("grid_rowcap", 64);
    sc->eval_cap = knob_int("grid_evalcap", 768);
    sc->defer_always = knob("grid_defer") && !strcmp(knob("grid_defer"), "always");
    sc->order_switch = knob_double("grid_order_switch", 0.125);
    hipError_t e = dmalloc(&sc->count2, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(sc->count2, 0, 2 * sizeof(uint32_t), s);   // s: the stream the searches will run on
    if (e == hipSuccess) e = dmalloc(&sc->list, (n ? n : 1) * sizeof(uint32_t));
    if (e == hipSuccess && knob_int("grid_stats", 0)) {
        e = dmalloc(&sc->stat, 2 * kStatSlots * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(sc->stat, 0, 2 * kStatSlots * sizeof(unsigned long long), s);
    }
    sc->fuse_cov = knob_int("grid_fuse_cov", 1);
    sc->flight = knob_int("grid_flight", 6);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sc->host_mem, 2 * sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        sc->host_mem[0] = 0;
        sc->host_mem[1] = 0;
        e = hipHostGetDevicePointer((void **)&sc->host_count, (void *)sc->host_mem, 0);
    }
    if (e != hipSuccess) grid_scratch_free(sc);
    return e;
}

void grid_scratch_free(GridScratch *sc)
{
    if (!sc) return;
    if (sc->stat) {
        const unsigned used = sc->stat_slot < kStatSlots ? sc->stat_slot : kStatSlots;
        unsigned long long *h = (unsigned long long *)malloc(2 * kStatSlots * sizeof(unsigned long long));
        if (h && hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(h, sc->stat, 2 * kStatSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess)
            for (unsigned k = 0; k < used; ++k) fprintf(stderr, "grid search %u: %llu rows walked, %llu targets evaluated\n", k, h[2 * k], h[2 * k + 1]);
        free(h);
        (void)dfree(sc->stat);
    }
    if (sc->count2) (void)dfree(sc->count2);
    if (sc->list) (void)dfree(sc->list);
    if (sc->host_mem) (void)hipHostFree((void *)sc->host_mem);
    *sc = GridScratch{};
}

}  // namespace icpk
