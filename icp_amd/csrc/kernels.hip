// icp_amd/csrc/kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ICP hot path.
//
// Built with -ffp-contract=off: the reference CPU path (the parity oracle) evaluates every product and
// sum with its own rounding (x86-64, no FMA), and exact ties between candidates only stay ties if the
// device does the same.  Fusion is used only where it is asked for explicitly (__builtin_fma in the
// pre-filter of the search kernel, whose candidates are always re-evaluated un-fused).
//
// Data layout: clouds live in HBM as SoA (x[], y[], z[]), fp64, source rows padded to a multiple of 1024
// (pad rows are finite and never counted), target rows centred once and padded with +inf to a multiple
// of the LDS tile (a pad target has d^2 = +inf and can never win).
//
// Determinism: no floating-point atomics; every reduction is a fixed-shape tree inside a 1024-row chunk
// followed by an in-index-order sum over chunks, so results do not depend on the grid, on the number of
// target splits, or on how many GPUs share the source rows.
#include "kernels.h"
#include "reduce.h"
#include "exact.h"
#include "search_common.h"

#include <math.h>
#include <stdlib.h>

namespace icpk {

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
// wave_sum / block_sum: reduce.h (shared with grid.hip, whose settled search forms the covariance partials itself)
__global__ __launch_bounds__(kWG) void colsum_partials_kernel(const double *__restrict__ aos, const double *__restrict__ x,
                                                              const double *__restrict__ y, const double *__restrict__ z,
                                                              size_t n, double *__restrict__ partial)
{
    __shared__ double smem[4 * 3];
    double acc[3] = {0.0, 0.0, 0.0};
    const size_t base = (size_t)blockIdx.x * kChunk + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kChunk / kWG; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) {
            if (aos) {
                acc[0] += aos[i * 3 + 0];
                acc[1] += aos[i * 3 + 1];
                acc[2] += aos[i * 3 + 2];
            } else {
                acc[0] += x[i];
                acc[1] += y[i];
                acc[2] += z[i];
            }
        }
    }
    const double tot = block_sum<3>(acc, smem);
    if (threadIdx.x < 3) partial[(size_t)blockIdx.x * kCenW + threadIdx.x] = tot;
}

__global__ __launch_bounds__(kWG) void finalize_centroid_kernel(const double *__restrict__ partial, size_t count, double denom,
                                                                double *__restrict__ out3)
{
    __shared__ double stage[kWG];
    const double acc = ordered_sum<kCenW, kWG>(partial, count, stage);
    if (threadIdx.x < 3) out3[threadIdx.x] = acc / denom;
}

__global__ __launch_bounds__(kWG) void aos_to_soa_kernel(const double *__restrict__ aos, size_t n, size_t n_pad,
                                                         const double *__restrict__ c3, double pad, double *__restrict__ x,
                                                         double *__restrict__ y, double *__restrict__ z)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n_pad) return;
    double vx = pad, vy = pad, vz = pad;
    if (i < n) {
        vx = aos[i * 3 + 0];
        vy = aos[i * 3 + 1];
        vz = aos[i * 3 + 2];
        if (c3) {  // Q -= Q_center (reference src/cpu/icp.cpp:198)
            vx = vx - c3[0];
            vy = vy - c3[1];
            vz = vz - c3[2];
        }
    }
    x[i] = vx;
    y[i] = vy;
    z[i] = vz;
}

__global__ __launch_bounds__(kWG) void soa_to_aos_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ z, size_t n, double *__restrict__ aos)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n) return;
    aos[i * 3 + 0] = x[i];
    aos[i * 3 + 1] = y[i];
    aos[i * 3 + 2] = z[i];
}

// A source stored in spatial order (engine option sort_source): row k of the engine's arrays is row perm[k] of the caller's.
// The cloud goes in through a gather and comes out through a scatter; so do the correspondence indices.
__global__ __launch_bounds__(kWG) void aos_gather_to_soa_kernel(const double *__restrict__ aos, const uint32_t *__restrict__ perm, size_t n,
                                                                size_t n_pad, double pad, double *__restrict__ x, double *__restrict__ y,
                                                                double *__restrict__ z)
{
    const size_t k = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (k >= n_pad) return;
    const size_t i = k < n ? perm[k] : 0;
    x[k] = k < n ? aos[i * 3 + 0] : pad;
    y[k] = k < n ? aos[i * 3 + 1] : pad;
    z[k] = k < n ? aos[i * 3 + 2] : pad;
}
__global__ __launch_bounds__(kWG) void soa_scatter_to_aos_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                                 const double *__restrict__ z, const uint32_t *__restrict__ perm, size_t n,
                                                                 double *__restrict__ aos)
{
    const size_t k = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (k >= n) return;
    const size_t i = perm[k];
    aos[i * 3 + 0] = x[k];
    aos[i * 3 + 1] = y[k];
    aos[i * 3 + 2] = z[k];
}
__global__ __launch_bounds__(kWG) void scatter_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, size_t n,
                                                          uint32_t *__restrict__ dst)
{
    const size_t k = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (k < n) dst[perm[k]] = src[k];
}

// ------------------------------------------------------------------------------------------------
// correspondence search
// ------------------------------------------------------------------------------------------------

// grid = (n_pad / (256*PPL), q_splits).  Every lane owns PPL source points (kept in registers, centred
// on load); the workgroup streams its target range through a 24 KB SoA LDS tile, and all lanes of a wave
// read the same target (LDS broadcast), so the inner loop is pure fp64 VALU: 8 ops per pair for the
// distance (6 in the fused pre-filter) + 1 v_min_f64.  Indices are not tracked in the fast path at all:
// once per CB targets each lane checks whether its running minimum moved, and only then does the wave
// take the slow path above.
//
// idx_prev (optional): last iteration's correspondence of every source point.  It is a legitimate
// candidate, so seeding (bk, bj) with it changes nothing in the result (the order on (key, index) is
// total) -- it only makes the running minimum tight from the first target on, so that almost no block
// takes the slow path, in every target split.
// Per-lane search state for PPL source points.
template <int PPL>
struct LaneSearch {
    double ax[PPL], ay[PPL], az[PPL];  // centred source points: P = P_copy - P_center (src/cpu/icp.cpp:208)
    uint32_t mh[PPL];                  // running minimum of the HIGH WORDS of the filter values, starts each block at thr_hi + 1
    uint32_t thr_hi[PPL];              // high word of thr
    double thr[PPL];                   // filter-space image of bd
    double bd[PPL], bk[PPL];
    uint32_t bj[PPL];
    uint32_t sblk[PPL];                // block (global target index / CB) already evaluated by seed_init, or ~0
};

template <int MODE>
__device__ __forceinline__ double filter_threshold(double bd)
{
    // fused and un-fused d^2 agree within (1+u)^6: any target with exact d^2 < bd has a fused d^2 below
    // bd*(1+2^-49) (+ a subnormal-range guard)
    return (MODE == kFma) ? __builtin_fma(bd, 1.0 + 0x1p-49, 1e-300) : bd;
}

// The fast path tracks no fp64 minimum at all.  A squared distance is a non-negative double, and non-negative doubles order
// like their bit patterns: a target can be below the threshold only if the HIGH 32 bits of its d^2 (sign, exponent, twenty
// mantissa bits -- the upper register of the pair, no instruction to get at it) are <= those of the threshold.  One
// v_min3_u32 folds the high words of TWO targets into the running minimum: half an instruction per pair where v_min_f64
// took one (9.08 -> 8.58 per pair in EXACT, 7.08 -> 6.58 in FMA).  A block is re-scanned exactly when its minimum high word
// is <= the threshold's: every target that could beat or tie the threshold is in such a block, and a block that only shares
// the threshold's high word (a 2^-20 band) costs a re-scan that changes nothing.  NaN and inf have high words above every
// finite threshold, like the fmin they replace.
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;   // (written out: left to itself the compiler re-associates the minima into trees of v_min_u32, 0.7 per pair)
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void set_threshold(double thr, double &out_thr, uint32_t &out_hi, uint32_t &out_mh)
{
    out_thr = thr;
    // (no threshold yet, thr = +inf: every FINITE d^2 is below it and an infinite one -- a row or a target out of range -- is
    // not, so the word to compare with is the largest finite one; the exact high word would re-scan every block of such a row)
    out_hi = thr < INFINITY ? (uint32_t)__double2hiint(thr) : 0x7FEFFFFFu;
    out_mh = out_hi + 1u;   // "nothing at or below the threshold's high word yet" (thr >= 0: no wrap)
}

// Scan `count` (multiple of CB) targets held in LDS as SoA; jbase = global index of target 0 of the tile.
template <int PPL, int MODE, int CB>
__device__ __forceinline__ void scan_tile(LaneSearch<PPL> &L, const double *sx, const double *sy, const double *sz, int count,
                                          unsigned jbase)
{
    for (int j0 = 0; j0 < count; j0 += CB) {
        static_assert(CB % 2 == 0, "targets are taken two at a time");
#pragma unroll
        for (int u = 0; u < CB; u += 2) {
            const double tx0 = sx[j0 + u], ty0 = sy[j0 + u], tz0 = sz[j0 + u];
            const double tx1 = sx[j0 + u + 1], ty1 = sy[j0 + u + 1], tz1 = sz[j0 + u + 1];
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const double d0 = (MODE == kFma) ? d2_fused(L.ax[s], L.ay[s], L.az[s], tx0, ty0, tz0)
                                                 : d2_exact(L.ax[s], L.ay[s], L.az[s], tx0, ty0, tz0);
                const double d1 = (MODE == kFma) ? d2_fused(L.ax[s], L.ay[s], L.az[s], tx1, ty1, tz1)
                                                 : d2_exact(L.ax[s], L.ay[s], L.az[s], tx1, ty1, tz1);
                L.mh[s] = umin3(L.mh[s], (uint32_t)__double2hiint(d0), (uint32_t)__double2hiint(d1));
            }
        }
        const unsigned blk = (jbase + (unsigned)j0) / CB;
        bool ev = false;
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            if (blk == L.sblk[s]) L.mh[s] = L.thr_hi[s] + 1u;   // the seed's own block was evaluated exactly by seed_init
            ev |= (L.mh[s] <= L.thr_hi[s]);
        }
        if (ev) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                if (L.mh[s] <= L.thr_hi[s]) {
                    double dbest;
                    rescan_block<CB>(L.ax[s], L.ay[s], L.az[s], sx, sy, sz, j0, jbase + (unsigned)j0, L.bd[s], L.bk[s], L.bj[s], dbest);
                    set_threshold(filter_threshold<MODE>(L.bd[s]), L.thr[s], L.thr_hi[s], L.mh[s]);
                }
            }
        }
    }
}

template <int PPL, int MODE, int CB>
__global__ __launch_bounds__(kWG) void search_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                     const double *__restrict__ pz, const double *__restrict__ pc,
                                                     const double *__restrict__ qx, const double *__restrict__ qy,
                                                     const double *__restrict__ qz, unsigned n_pad, unsigned m, unsigned m_scan,
                                                     unsigned q_per_split, const uint32_t *__restrict__ idx_prev,
                                                     double *__restrict__ out_key, uint32_t *__restrict__ out_idx)
{
    __shared__ double sq[3][kTileQ];
    const unsigned tid = threadIdx.x;
    const unsigned base = blockIdx.x * (kWG * PPL) + tid;
    const double c0 = pc[0], c1 = pc[1], c2 = pc[2];
    // this workgroup's target range [q0, q1): m_scan = m rounded up to kSplitGran (the rest of the padding is never read)
    const unsigned q0 = blockIdx.y * q_per_split;
    const unsigned q1 = (q0 + q_per_split < m_scan) ? (q0 + q_per_split) : m_scan;

    LaneSearch<PPL> L;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        L.ax[s] = px[base + s * kWG] - c0;
        L.ay[s] = py[base + s * kWG] - c1;
        L.az[s] = pz[base + s * kWG] - c2;
        const uint32_t js = idx_prev ? idx_prev[base + s * kWG] : 0xFFFFFFFFu;
        L.bd[s] = seed_init<CB>(L.ax[s], L.ay[s], L.az[s], qx, qy, qz, js, m, q0, q1, L.bk[s], L.bj[s], L.sblk[s]);
        set_threshold(filter_threshold<MODE>(L.bd[s]), L.thr[s], L.thr_hi[s], L.mh[s]);
    }

    for (unsigned t0 = q0; t0 < q1; t0 += kTileQ) {
        const int cnt = (q1 - t0 < (unsigned)kTileQ) ? (int)(q1 - t0) : kTileQ;   // multiple of kSplitGran
        __syncthreads();
        for (int k = tid; k < cnt; k += kWG) {
            sq[0][k] = qx[t0 + k];
            sq[1][k] = qy[t0 + k];
            sq[2][k] = qz[t0 + k];
        }
        __syncthreads();
        scan_tile<PPL, MODE, CB>(L, sq[0], sq[1], sq[2], cnt, t0);
    }
    const size_t o = (size_t)blockIdx.y * n_pad + base;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        out_key[o + s * kWG] = L.bk[s];
        out_idx[o + s * kWG] = L.bj[s];
    }
}

// ------------------------------------------------------------------------------------------------
// correspondence search with an fp32 pre-filter (MODE f32)
// ------------------------------------------------------------------------------------------------
// Same structure and same exact slow path as search_kernel, but the fast path works on
//     s_j = |q_j|^2 - 2 p.q_j   ( = d^2(p, q_j) - |p|^2, so argmin_j s_j = argmin_j d^2 )
// in fp32: three v_fma_f32 per pair on a 16-byte target record {-2qx, -2qy, -2qz, |q|^2} and one v_min3_f32 per two
// pairs -- 3.5 two-cycle instructions per pair instead of 7.1 four-cycle ones.  It is only a filter: with
// eps = 2^-24, |s_j(fp32) - s_j(real)| <= eps (10.1 |p| |q|max + 4.1 |q|max^2) =: E (five roundings on the products,
// four on |q|^2), hence every target whose exact fp64 d^2 is below the running exact minimum bd satisfies
//     s_j(fp32) < bd (1+2^-50) - |p|^2 (1-2^-50) + E,
// which is the event threshold (rounded up to fp32).  Blocks that trip it are re-evaluated in the oracle's own
// un-fused fp64 sequence, exactly as in the fp64 modes, so the indices are bit-identical; a cloud for which E is
// large compared with the nearest-neighbour distances only takes the slow path more often.
constexpr int kTileQF = 2048;

__global__ __launch_bounds__(kWG) void build_qrec_kernel(const double *__restrict__ qx, const double *__restrict__ qy,
                                                         const double *__restrict__ qz, size_t m, size_t m_pad,
                                                         QRec *__restrict__ rec, unsigned long long *__restrict__ qmax2_bits)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    double n2 = 0.0;
    if (i < m_pad) {
        QRec r{0.0f, 0.0f, 0.0f, INFINITY};   // pad: s = +inf, never an event
        if (i < m) {
            const double x = qx[i], y = qy[i], z = qz[i];
            n2 = x * x + y * y + z * z;
            r = QRec{(float)(-2.0 * x), (float)(-2.0 * y), (float)(-2.0 * z), (float)n2};
        }
        float *o = (float *)rec + (i >> 1) * 8 + (i & 1);   // pair-interleaved layout
        o[0] = r.x;
        o[2] = r.y;
        o[4] = r.z;
        o[6] = r.w;
    }
    // max |q|^2 over the cloud (positive doubles order like their bit patterns; NaN/inf end up above every finite value)
    unsigned long long b = (unsigned long long)__double_as_longlong(n2);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(b, o, 64);
        b = other > b ? other : b;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(qmax2_bits, b);
}

template <int PPL, int CB, bool PK>
__global__ __launch_bounds__(kWG) void search_f32_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                         const double *__restrict__ pz, const double *__restrict__ pc,
                                                         const double *__restrict__ qx, const double *__restrict__ qy,
                                                         const double *__restrict__ qz, const QRec *__restrict__ qrec, double qmax,
                                                         unsigned n_pad, unsigned m, unsigned m_scan, unsigned q_per_split,
                                                         const uint32_t *__restrict__ idx_prev, double *__restrict__ out_key,
                                                         uint32_t *__restrict__ out_idx)
{
    __shared__ QRec st[kTileQF];
    const unsigned tid = threadIdx.x;
    const unsigned base = blockIdx.x * (kWG * PPL) + tid;
    const double c0 = pc[0], c1 = pc[1], c2 = pc[2];

    float fx[PPL], fy[PPL], fz[PPL], ms[PPL], thr[PPL];
    double off[PPL];   // E - |p|^2 (1 - 2^-50): threshold = bd (1+2^-50) + off
    double bd[PPL], bk[PPL];
    uint32_t bj[PPL], sblk[PPL];
    const unsigned q0 = blockIdx.y * q_per_split;
    const unsigned q1 = (q0 + q_per_split < m_scan) ? (q0 + q_per_split) : m_scan;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const double ax = px[base + s * kWG] - c0, ay = py[base + s * kWG] - c1, az = pz[base + s * kWG] - c2;
        const double pn = (ax * ax + ay * ay) + az * az;
        const double E = 0x1p-24 * (11.0 * sqrt(pn) * qmax + 5.0 * qmax * qmax) + 1e-35;
        // outside the range where the fp32 bound is meaningful the lane always takes the exact path: its threshold is +inf
        // and its filter operands are zero, so that every filter value is the finite |q|^2 (a huge coordinate would
        // overflow the products to inf / NaN, which v_min3_f32 drops -- the block would be skipped)
        const bool usable = pn < 1e30;
        fx[s] = usable ? (float)ax : 0.0f;
        fy[s] = usable ? (float)ay : 0.0f;
        fz[s] = usable ? (float)az : 0.0f;
        off[s] = usable ? (E - pn) + pn * 0x1p-50 : INFINITY;
        const uint32_t js = idx_prev ? idx_prev[base + s * kWG] : 0xFFFFFFFFu;
        bd[s] = seed_init<CB>(ax, ay, az, qx, qy, qz, js, m, q0, q1, bk[s], bj[s], sblk[s]);
        thr[s] = f32_above(__builtin_fma(bd[s], 1.0 + 0x1p-50, off[s]));
        ms[s] = thr[s];
    }

    for (unsigned t0 = q0; t0 < q1; t0 += kTileQF) {
        const int cnt = (q1 - t0 < (unsigned)kTileQF) ? (int)(q1 - t0) : kTileQF;   // multiple of kSplitGran
        __syncthreads();
        for (int k = tid; k < cnt; k += kWG) st[k] = qrec[t0 + k];
        __syncthreads();
        for (int j0 = 0; j0 < cnt; j0 += CB) {
            // software-pipelined over pairs of targets: the LDS reads of pair u+1 are in flight while pair u is computed
            const f32x2 *pr = (const f32x2 *)&st[j0];
            f32x2 X = pr[0], Y = pr[1], Z = pr[2], W = pr[3];
#pragma unroll
            for (int u = 0; u < CB; u += 2) {
                f32x2 Xn = X, Yn = Y, Zn = Z, Wn = W;
                if (u + 2 < CB) {
                    Xn = pr[(u + 2) * 2 + 0];
                    Yn = pr[(u + 2) * 2 + 1];
                    Zn = pr[(u + 2) * 2 + 2];
                    Wn = pr[(u + 2) * 2 + 3];
                }
                if (PK) {
                    // stage by stage over the PPL independent chains: a dependent v_pk_fma_f32 right behind its producer
                    // costs a wait state
                    f32x2 t[PPL];
#pragma unroll
                    for (int s = 0; s < PPL; ++s) t[s] = __builtin_elementwise_fma((f32x2){fx[s], fx[s]}, X, W);
#pragma unroll
                    for (int s = 0; s < PPL; ++s) t[s] = __builtin_elementwise_fma((f32x2){fy[s], fy[s]}, Y, t[s]);
#pragma unroll
                    for (int s = 0; s < PPL; ++s) t[s] = __builtin_elementwise_fma((f32x2){fz[s], fz[s]}, Z, t[s]);
#pragma unroll
                    for (int s = 0; s < PPL; ++s) ms[s] = vmin3f(ms[s], t[s].x, t[s].y);
                } else {
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        const float sa = __builtin_fmaf(fz[s], Z.x, __builtin_fmaf(fy[s], Y.x, __builtin_fmaf(fx[s], X.x, W.x)));
                        const float sb = __builtin_fmaf(fz[s], Z.y, __builtin_fmaf(fy[s], Y.y, __builtin_fmaf(fx[s], X.y, W.y)));
                        ms[s] = vmin3f(ms[s], sa, sb);
                    }
                }
                X = Xn; Y = Yn; Z = Zn; W = Wn;
            }
            const unsigned blk = (t0 + (unsigned)j0) / CB;
            bool ev = false;
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                if (blk == sblk[s]) ms[s] = thr[s];   // the seed's own block was evaluated exactly by seed_init
                ev |= (ms[s] < thr[s]);
            }
            if (ev) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    if (ms[s] < thr[s]) {
                        // exact re-evaluation of the block in fp64 from the source arrays (rare: reloaded, not kept in registers)
                        const double ax = px[base + s * kWG] - c0, ay = py[base + s * kWG] - c1, az = pz[base + s * kWG] - c2;
                        const unsigned jg = t0 + (unsigned)j0;
                        double dbest;
                        rescan_block<CB>(ax, ay, az, qx + jg, qy + jg, qz + jg, 0, jg, bd[s], bk[s], bj[s], dbest);
                        thr[s] = f32_above(__builtin_fma(bd[s], 1.0 + 0x1p-50, off[s]));
                        ms[s] = thr[s];
                    }
                }
            }
        }
    }
    const size_t o = (size_t)blockIdx.y * n_pad + base;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        out_key[o + s * kWG] = bk[s];
        out_idx[o + s * kWG] = bj[s];
    }
}

// ------------------------------------------------------------------------------------------------
// merge of the target splits + cross-covariance partials
// ------------------------------------------------------------------------------------------------
// cov[a][b] = sum_i Qz[idx[i]][a] * Pz[i][b]  (reference src/cpu/icp.cpp:108-131: q^T . p per point, each
// product rounded once, summed).  MERGE: idx comes from the per-split (key, idx) records; otherwise from idx_in.
template <bool MERGE>
__global__ __launch_bounds__(kWG) void cov_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                  const double *__restrict__ pz, size_t n, size_t n_pad,
                                                  const double *__restrict__ pc, const double *__restrict__ qx,
                                                  const double *__restrict__ qy, const double *__restrict__ qz,
                                                  const double4 *__restrict__ qrec4, size_t m,
                                                  int q_splits, const double *__restrict__ keys,
                                                  const uint32_t *__restrict__ idxs, const double2 *__restrict__ grid_rec,
                                                  const uint32_t *__restrict__ idx_in, uint32_t *__restrict__ idx_out,
                                                  double4 *__restrict__ prec, size_t prec_stride, double reject_d2,
                                                  double *__restrict__ partial)
{
    __shared__ double smem[4 * kCovW];
    const double c0 = pc[0], c1 = pc[1], c2 = pc[2];
    double acc[kCovW];
#pragma unroll
    for (int k = 0; k < kCovW; ++k) acc[k] = 0.0;
    const size_t base = (size_t)blockIdx.x * kChunk + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kChunk / kWG; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) {
            uint32_t j;
            double q0 = 0.0, q1 = 0.0, q2 = 0.0;
            if (MERGE && grid_rec) {
                j = (uint32_t)__double_as_longlong(grid_rec[i].y);
                idx_out[i] = j;
            } else if (MERGE) {
                double bk = keys[i];
                j = idxs[i];
                for (int y = 1; y < q_splits; ++y) {  // lexicographic (key, index): a split may report the seed
                    const double ky = keys[(size_t)y * n_pad + i];
                    const uint32_t jy = idxs[(size_t)y * n_pad + i];
                    if (ky < bk || (ky == bk && jy < j)) {
                        bk = ky;
                        j = jy;
                    }
                }
                idx_out[i] = j;
            } else {
                j = idx_in[i];
            }
            if (MERGE && prec && !(j < m)) prec[prec_stride + i] = make_double4(0.0, 0.0, 0.0, __longlong_as_double((long long)(unsigned long long)j));
            if (j < m) {
                const double p0 = px[i] - c0, p1 = py[i] - c1, p2 = pz[i] - c2;
                if (qrec4) {   // one 32-byte record per gathered target
                    const double4 v = qrec4[j];
                    q0 = v.x; q1 = v.y; q2 = v.z;
                } else {
                    q0 = qx[j]; q1 = qy[j]; q2 = qz[j];
                }
                // the next grid search's seed: the match and its index (no certificate: this search proved no bound)
                if (MERGE && prec) prec[prec_stride + i] = make_double4(q0, q1, q2, __longlong_as_double((long long)(unsigned long long)j));
                // the weight hook of compute_cross_variance (src/cpu/icp.cpp:108-131: weight = kernel(p - q), a constant 1
                // there) as a hard threshold on the pair's distance: weight 1 up to reject_d2 (+inf = the reference), else 0
                const double d = d2_exact(p0, p1, p2, q0, q1, q2);
                if (d <= reject_d2) {
                    acc[0] += q0 * p0; acc[1] += q0 * p1; acc[2] += q0 * p2;
                    acc[3] += q1 * p0; acc[4] += q1 * p1; acc[5] += q1 * p2;
                    acc[6] += q2 * p0; acc[7] += q2 * p1; acc[8] += q2 * p2;
                    acc[9] += d;
                    acc[10] += 1.0;   // pairs that entered the sums
                }
            }
        }
    }
    const double tot = block_sum<kCovW>(acc, smem);
    if (threadIdx.x < kCovW) partial[(size_t)blockIdx.x * kCovW + threadIdx.x] = tot;
}

// The same partial sums on the grid path with source records: everything a row needs -- its coordinates, the matched
// target's coordinates and index -- sits in the row's own 64-byte record (search_grid_kernel wrote the second half), so a
// workgroup streams 64 KB of whole lines, all of its loads in flight before the first use, and gathers nothing.  Same
// arithmetic in the same order as cov_kernel<true>: bit-identical partials.
__global__ __launch_bounds__(kWG) void cov_rec_kernel(const double4 *__restrict__ prec, size_t prec_stride, size_t n, const double *__restrict__ pc,
                                                      size_t m, uint32_t *__restrict__ idx_out, double reject_d2,
                                                      double *__restrict__ partial)
{
    __shared__ double smem[4 * kCovW];
    const double c0 = pc[0], c1 = pc[1], c2 = pc[2];
    double acc[kCovW];
#pragma unroll
    for (int k = 0; k < kCovW; ++k) acc[k] = 0.0;
    const size_t base = (size_t)blockIdx.x * kChunk + threadIdx.x;
    double4 a[kChunk / kWG], b[kChunk / kWG];
#pragma unroll
    for (int k = 0; k < kChunk / kWG; ++k) {
        const size_t i = base + (size_t)k * kWG;
        const size_t ic = i < n ? i : n - 1;   // rows past the end: a valid address, masked below
        a[k] = prec[ic];
        b[k] = prec[prec_stride + ic];
    }
#pragma unroll
    for (int k = 0; k < kChunk / kWG; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) {
            const uint32_t j = (uint32_t)__double_as_longlong(b[k].w);
            idx_out[i] = j;
            if (j < m) {
                const double p0 = a[k].x - c0, p1 = a[k].y - c1, p2 = a[k].z - c2;
                const double q0 = b[k].x, q1 = b[k].y, q2 = b[k].z;
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
    }
    const double tot = block_sum<kCovW>(acc, smem);
    if (threadIdx.x < kCovW) partial[(size_t)blockIdx.x * kCovW + threadIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------
// 3x3 solve
// ------------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi SVD of the row-major 3x3 A: A V = U diag(S).  Replaces cusolverDnDgesvd
// (reference src/libgpualg/svd.cu:19-115) / LAPACK dgesvd_ (src/libalg/svd.cpp:36-73).  Only R = U V^T is
// consumed by the hot path; it is the orthogonal polar factor of A and does not depend on factor signs.
// Directions of singular values at roundoff level (<= 1e-13 S[0]) are completed so that det(U) = det(V).
//
// One lane runs this while its launch waits, and a settled grid iteration is three such waits long, so the length of the
// DEPENDENT chain is what counts: a lone wave issues a dependent fp64 instruction every ~20 cycles (measured: 8 us of a 22 us
// iteration on a 2 903-point cloud is this kernel).  Inverse square roots therefore come from the hardware's seed instruction
// (v_rsq_f64: 2^-23 relative by its specification) plus two Newton steps -- about an ulp -- instead of a correctly rounded
// root followed by a correctly rounded division: 30 dependent instructions a rotation where the textbook formulas need 87, and
// the convergence test compares squares.  What the result's accuracy rests on: c^2 + s^2 = 1 in every rotation (both from one
// inverse root, to an ulp), the convergence test, and unit columns of U (norms from refined inverse roots, a Gram-Schmidt pass
// and a cross product).  The angle of a rotation only decides how much of its off-diagonal element is left for the next
// sweep.  The matrix is scaled by a power of two to a largest element in [1/2, 1) first (exact), so that no square under- or
// overflows for any input scale.
__device__ __forceinline__ double rsqrt_2nr(double x)   // x in [1/4, 8]
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}
__device__ __forceinline__ double rsqrt_1nr(double x)   // one step: 2^-45 relative, for a quantity that only steers an iteration
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

// One rotation of columns P and Q of a (and of v): false = they are orthogonal already.
constexpr double kJacobiOff = 4.5e-16 * 4.5e-16;   // converged: |a_p . a_q| <= 4.5e-16 |a_p| |a_q|, compared as squares
template <int P, int Q>
__device__ __forceinline__ bool jacobi_pair(double (&a)[9], double (&v)[9])
{
    const double alpha = (a[P] * a[P] + a[3 + P] * a[3 + P]) + a[6 + P] * a[6 + P];
    const double beta = (a[Q] * a[Q] + a[3 + Q] * a[3 + Q]) + a[6 + Q] * a[6 + Q];
    const double gamma = (a[P] * a[Q] + a[3 + P] * a[3 + Q]) + a[6 + P] * a[6 + Q];
    if (gamma == 0.0 || gamma * gamma <= kJacobiOff * (alpha * beta)) return false;
    // t = tan of the rotation angle, the smaller root: sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with
    // zeta = (beta - alpha) / (2 gamma), written without the division by gamma: t = sign(d) g / (|d| + sqrt(d^2 + g^2)),
    // d = beta - alpha, g = 2 gamma -- a ratio, so d and g are first scaled by a power of two to max(|d|, |g|) in [1/2, 1):
    // the sum of squares is then in [1/4, 2) whatever the norms of the two columns are.
    const double d0 = beta - alpha, g0 = gamma + gamma;
    const int e2 = __builtin_amdgcn_frexp_exp(__builtin_fmax(__builtin_fabs(d0), __builtin_fabs(g0)));
    const double d = __builtin_amdgcn_ldexp(d0, -e2), g2 = __builtin_amdgcn_ldexp(g0, -e2);
    const double x = d * d + g2 * g2;
    const double h = x * rsqrt_1nr(x);              // |(d, g)|: an error here only leaves a little of gamma for the next sweep
    const double y = __builtin_fabs(d) + h;         // t = sign(d) g / y, so  c = 1 / sqrt(1 + t^2) = y / sqrt(y^2 + g^2),  s = c t:
    const double r = rsqrt_2nr(y * y + g2 * g2);    // one inverse root (of a value in [1/4, 8)) gives both, to about an ulp --
    const double c = y * r;                         // c^2 + s^2 = 1 is what keeps V orthogonal
    const double s = (d >= 0.0 ? g2 : -g2) * r;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double x0 = a[i * 3 + P], y0 = a[i * 3 + Q];
        a[i * 3 + P] = c * x0 - s * y0;
        a[i * 3 + Q] = s * x0 + c * y0;
        x0 = v[i * 3 + P];
        y0 = v[i * 3 + Q];
        v[i * 3 + P] = c * x0 - s * y0;
        v[i * 3 + Q] = s * x0 + c * y0;
    }
    return true;
}

__device__ __forceinline__ void svd3(const double *A, double *U, double *S, double *VT)
{
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double amax = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) amax = __builtin_fmax(amax, __builtin_fabs(A[k]));   // (NaN elements are skipped here and poison the sweeps)
    int ex = 0;
    if (amax > 0.0 && amax < INFINITY) ex = __builtin_amdgcn_frexp_exp(amax);   // amax = f * 2^ex, f in [1/2, 1)
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = __builtin_amdgcn_ldexp(A[k], -ex);
    for (int sweep = 0; sweep < 30; ++sweep) {
        const bool r01 = jacobi_pair<0, 1>(a, v);
        const bool r02 = jacobi_pair<0, 2>(a, v);
        const bool r12 = jacobi_pair<1, 2>(a, v);
        if (!(r01 || r02 || r12)) break;
    }
    // column norms of the scaled matrix and their reciprocals (one refined inverse root each; a column that vanishes next to
    // the largest -- it is dropped below -- takes the library root)
    double sg[3], rs[3];
    int ord[3] = {0, 1, 2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double ss = (a[j] * a[j] + a[3 + j] * a[3 + j]) + a[6 + j] * a[6 + j];
        if (ss >= 0x1p-600 && ss < INFINITY) {
            rs[j] = rsqrt_2nr(ss);
            sg[j] = ss * rs[j];
        } else {
            rs[j] = 0.0;
            sg[j] = sqrt(ss);
        }
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (sg[ord[j]] < sg[ord[j + 1]]) {
                const int t = ord[j];
                ord[j] = ord[j + 1];
                ord[j + 1] = t;
            }
    double u[9], vv[9];
    const double tol = 1e-13 * sg[ord[0]];
    int rank = 0;
    for (int j = 0; j < 3; ++j) {
        const int c = ord[j];
        S[j] = __builtin_amdgcn_ldexp(sg[c], ex);
        for (int i = 0; i < 3; ++i) vv[i * 3 + j] = v[i * 3 + c];
        const bool ok = sg[c] > 0.0 && sg[c] > tol && rank == j;
        for (int i = 0; i < 3; ++i) u[i * 3 + j] = ok ? a[i * 3 + c] * rs[c] : 0.0;   // (ok: sg > 1e-13 of the largest, so rs is set)
        if (ok) rank = j + 1;
    }
    const double detv = vv[0] * (vv[4] * vv[8] - vv[5] * vv[7]) - vv[1] * (vv[3] * vv[8] - vv[5] * vv[6]) +
                        vv[2] * (vv[3] * vv[7] - vv[4] * vv[6]);
    if (rank == 0) {
        for (int k = 0; k < 9; ++k) u[k] = vv[k];
    } else {
        if (rank == 1) {  // any unit vector orthogonal to u0
            int mm = 0;
            if (fabs(u[3]) < fabs(u[mm * 3])) mm = 1;
            if (fabs(u[6]) < fabs(u[mm * 3])) mm = 2;
            double e[3] = {0, 0, 0};
            e[mm] = 1.0;
            u[1] = u[3] * e[2] - u[6] * e[1];
            u[4] = u[6] * e[0] - u[0] * e[2];
            u[7] = u[0] * e[1] - u[3] * e[0];
        }
        const double d01 = u[0] * u[1] + u[3] * u[4] + u[6] * u[7];
        for (int i = 0; i < 3; ++i) u[i * 3 + 1] -= d01 * u[i * 3];
        const double n1 = rsqrt_2nr((u[1] * u[1] + u[4] * u[4]) + u[7] * u[7]);   // the squared norm is ~1 (rank 1: >= 2/3)
        for (int i = 0; i < 3; ++i) u[i * 3 + 1] *= n1;
        const double w[3] = {u[3] * u[7] - u[6] * u[4], u[6] * u[1] - u[0] * u[7], u[0] * u[4] - u[3] * u[1]};
        double s3 = detv < 0 ? -1.0 : 1.0;
        if (rank == 3) s3 = (w[0] * u[2] + w[1] * u[5] + w[2] * u[8]) < 0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) u[i * 3 + 2] = s3 * w[i];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            U[i * 3 + j] = u[i * 3 + j];
            VT[i * 3 + j] = vv[j * 3 + i];
        }
}

__device__ __forceinline__ void solve_record(const double *tot, const double *pc, const double *qc, double n_global, double *out);

// One lane, after solve_record: pairs used; how far this iteration's motion is from the identity (convergence test of the
// caller); how far R is from orthogonal.  out = the solve record, pc = the centroid the motion was solved about.
__device__ __forceinline__ void solve_extra(const double *out, const double *pc, double pairs, double *extra)
{
    double dr = 0.0, dt = 0.0;
    for (int k = 0; k < 9; ++k) dr = fmax(dr, fabs(out[12 + k] - ((k % 4 == 0) ? 1.0 : 0.0)));
    // the motion of the cloud is x -> R x + t; about its centroid Pc the translation part is R Pc + t - Pc
    for (int j = 0; j < 3; ++j) {
        const double moved = ((out[12 + j * 3] * pc[0] + out[12 + j * 3 + 1] * pc[1]) + out[12 + j * 3 + 2] * pc[2]) + out[21 + j];
        dt = fmax(dt, fabs(moved - pc[j]));
    }
    extra[0] = pairs;
    extra[1] = dr;
    extra[2] = dt;
    // |R^T R - I|_F: how far R is from orthogonal (a few ulp; NaN for a NaN covariance) -- the grid search's motion
    // certificate inverts R by its transpose and asks for < 2^-42
    double defect = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double g = 0.0;
            for (int k = 0; k < 3; ++k) g += out[12 + k * 3 + a] * out[12 + k * 3 + b];
            g -= (a == b) ? 1.0 : 0.0;
            defect += g * g;
        }
    extra[3] = __builtin_amdgcn_sqrt(defect);   // compared with a threshold only: the hardware's approximate root (2^-23 relative)
}

// One workgroup of kSolveWG threads: the fixed-order sum of all chunk partials (one pass of ordered_sum up to
// 32 * (kSolveWG / kCovW) chunks = 2.7M rows: a single memory latency), then one lane solves.
constexpr int kSolveWG = 1024;
__global__ __launch_bounds__(kSolveWG) void solve_kernel(const double *__restrict__ partial, size_t count,
                                                         const double *__restrict__ pc, const double *__restrict__ qc,
                                                         double n_global, double *__restrict__ out, double *__restrict__ extra)
{
    __shared__ double stage[kSolveWG];
    __shared__ double tot[kCovW];
    const double acc = ordered_sum<kCovW, kSolveWG>(partial, count, stage);
    if (threadIdx.x < kCovW) tot[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    // n_global <= 0: the rms is taken over the pairs that entered the sums (distance rejection active)
    solve_record(tot, pc, qc, n_global > 0.0 ? n_global : (tot[10] > 0.0 ? tot[10] : 1.0), out);
    if (extra) solve_extra(out, pc, tot[10], extra);
}

// One lane: cov -> SVD -> R, t; writes the 28-double record (layout in kernels.h).
__device__ __forceinline__ void solve_record(const double *tot, const double *pc, const double *qc, double n_global, double *out)
{
    double cov[9], U[9], S[3], VT[9], R[9], t[3];
    for (int k = 0; k < 9; ++k) cov[k] = tot[k];
    svd3(cov, U, S, VT);
    // R = U.dot(V_T) (src/cpu/icp.cpp:221); dot_product sums tmp += a*b in m order (src/libalg/alg.cpp:91-101)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double tmp = 0.0;
            for (int mm = 0; mm < 3; ++mm) tmp += U[i * 3 + mm] * VT[mm * 3 + j];
            R[i * 3 + j] = tmp;
        }
    // t = Q_center - P_center.dot(R.transpose()) (src/cpu/icp.cpp:223)
    for (int j = 0; j < 3; ++j) {
        double tmp = 0.0;
        for (int mm = 0; mm < 3; ++mm) tmp += pc[mm] * R[j * 3 + mm];
        t[j] = qc[j] - tmp;
    }
    for (int k = 0; k < 9; ++k) out[k] = cov[k];
    for (int k = 0; k < 3; ++k) out[9 + k] = S[k];
    for (int k = 0; k < 9; ++k) out[12 + k] = R[k];
    for (int k = 0; k < 3; ++k) out[21 + k] = t[k];
    out[24] = sqrt(tot[9] / n_global);
    for (int k = 0; k < 3; ++k) out[25 + k] = pc[k];
}

// ------------------------------------------------------------------------------------------------
// transform apply + centroid partials of the new cloud
// ------------------------------------------------------------------------------------------------
// P_copy = P_copy.dot(R.transpose()) + t (src/cpu/icp.cpp:225): per row and output column j,
// ((0 + x R[j][0]) + y R[j][1]) + z R[j][2], then + t[j]; un-fused.
// REC: the running cloud is the A half of the source records (grid path): 32 bytes in, 32 bytes out per row, one stream
// each way; the SoA arrays are not touched (they are caught up on demand, source_rec_to_soa).
template <bool REC>
__global__ __launch_bounds__(kWG) void apply_kernel(double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
                                                    size_t n, const double *__restrict__ rec, double4 *__restrict__ prec,
                                                    double *__restrict__ cen_partial)
{
    __shared__ double smem[4 * 3];
    const size_t base = (size_t)blockIdx.x * kChunk + threadIdx.x;
    // every load of the workgroup's four row groups goes out before the first dependent use (HBM bound: bytes in flight)
    double lx[kChunk / kWG], ly[kChunk / kWG], lz[kChunk / kWG], lw[kChunk / kWG];
#pragma unroll
    for (int k = 0; k < kChunk / kWG; ++k) {
        const size_t i = base + (size_t)k * kWG;
        const bool in = i < n;
        if (REC) {
            const double4 v = prec[in ? i : n - 1];
            lx[k] = v.x; ly[k] = v.y; lz[k] = v.z; lw[k] = v.w;
        } else {
            lx[k] = in ? px[i] : 0.0;
            ly[k] = in ? py[i] : 0.0;
            lz[k] = in ? pz[i] : 0.0;
            lw[k] = 0.0;
        }
    }
    const double *R = rec + 12, *t = rec + 21;
    const double r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
    const double t0 = t[0], t1 = t[1], t2 = t[2];
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kChunk / kWG; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) {
            const double x = lx[k], y = ly[k], z = lz[k];
            const double nx = ((x * r00 + y * r01) + z * r02) + t0;
            const double ny = ((x * r10 + y * r11) + z * r12) + t1;
            const double nz = ((x * r20 + y * r21) + z * r22) + t2;
            if (REC) {
                prec[i] = make_double4(nx, ny, nz, lw[k]);
            } else {
                px[i] = nx;
                py[i] = ny;
                pz[i] = nz;
            }
            acc[0] += nx;
            acc[1] += ny;
            acc[2] += nz;
        }
    }
    const double tot = block_sum<3>(acc, smem);
    if (threadIdx.x < 3) cen_partial[(size_t)blockIdx.x * kCenW + threadIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------
// small-target search (whole target cloud in LDS, <= 4096 points): index carried in the mantissa
// ------------------------------------------------------------------------------------------------
// With only ~1000 targets and 256 source points per wave, "some lane's minimum moved" is true in every block,
// so the event scheme of search_kernel degenerates.  Here the fast path needs no events at all: the low 12
// mantissa bits of every (fused) d^2 are replaced by the target index (one v_and_or_b32), after which a plain
// v_min_f64 tracks value and index together (lowest index wins among equal truncated values), and a
// v_max/v_min pair tracks the runner-up.  If the runner-up is more than 2^-38 (relative) above the minimum, no
// other target can have an oracle key <= the winner's (truncation 2^-40, fusion 6u, and a key gap of 2^-42 is far
// above the rounding of the two square roots), so the embedded index is the oracle's answer.  Otherwise -- true
// near-ties, duplicates, lattice data -- that source point is re-scanned exactly.
constexpr double kSmallPad = 1e150;            // pad target coordinate: d^2 ~ 3e300, finite (a NaN would poison the runner-up)
constexpr unsigned kSmallIdxMask = 0xFFFFF000u;

template <int MODE>
__device__ __forceinline__ double packed_d2(double ax, double ay, double az, double tx, double ty, double tz, unsigned j)
{
    const double d = (MODE == kFma) ? d2_fused(ax, ay, az, tx, ty, tz) : d2_exact(ax, ay, az, tx, ty, tz);
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    unsigned lo;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)b), "s"(kSmallIdxMask), "v"(j));   // (lo & mask) | j
    b = (b & 0xFFFFFFFF00000000ull) | lo;
    return __longlong_as_double((long long)b);
}

// v_min_f64 / v_max_f64 on values known to be finite, non-NaN doubles with integer-edited mantissas: spelled as
// instructions so that the compiler does not add a canonicalising v_max_f64 x,x in front of each of them.
__device__ __forceinline__ double vmin64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Exact (oracle-order) scan of all m targets for one source point: first minimum of sqrt(sqrt(d^2)).
__device__ __noinline__ uint32_t exact_scan_small(double ax, double ay, double az, const double *sx, const double *sy, const double *sz,
                                                  unsigned m)
{
    double bd = INFINITY, bk = INFINITY;
    uint32_t bj = 0xFFFFFFFFu;
    for (unsigned j = 0; j < m; ++j) {
        const double d = d2_exact(ax, ay, az, sx[j], sy[j], sz[j]);
        if (d < bd) {
            bd = d;
            const double key = key_of(d);
            if (key < bk) {
                bk = key;
                bj = j;
            }
        }
    }
    return bj;
}

template <int PPL, int MODE>
__device__ __forceinline__ void scan_small(const double (&ax)[PPL], const double (&ay)[PPL], const double (&az)[PPL], const double *sx,
                                           const double *sy, const double *sz, unsigned m, unsigned m_pad, uint32_t (&bj)[PPL])
{
    double m1[PPL], m2[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) m1[s] = m2[s] = INFINITY;
    for (unsigned j0 = 0; j0 < m_pad; j0 += 8) {
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) {
            const double tx = sx[j0 + u], ty = sy[j0 + u], tz = sz[j0 + u];
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const double t = packed_d2<MODE>(ax[s], ay[s], az[s], tx, ty, tz, j0 + u);
                m2[s] = vmin64(m2[s], vmax64(m1[s], t));
                m1[s] = vmin64(m1[s], t);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const unsigned j = (unsigned)__double_as_longlong(m1[s]) & ~kSmallIdxMask;
        const bool unique = (m2[s] > m1[s] * (1.0 + 0x1p-38)) && (m1[s] > 1e-290) && (m1[s] < 1e290) && (j < m);
        bj[s] = j;
        if (!unique) bj[s] = exact_scan_small(ax[s], ay[s], az[s], sx, sy, sz, m);
    }
}

// --- expanded form (batch default): the fast path works on s_j = |q_j|^2 - 2 p.q_j (= d^2 - |p|^2, same argmin) with the
// target as one 32-byte LDS record {-2qx, -2qy, -2qz, |q|^2}: three v_fma_f64 instead of three subtractions, a multiply
// and two FMAs -- 7 instructions per pair with the index embedding and the runner-up tracking instead of 10.
// |s_j(computed) - s_j| <= E = 2^-52 (11 |p| |q|max + 5 |q|max^2) (the bound of search_f32_kernel with fp64's epsilon),
// so if the runner-up is more than 2E + 2^-38 (|s| + |p|^2) above the minimum (index truncation 2^-40 |s| on both, tie
// band 2^-48 d^2) no other target can reach the winner's key; otherwise the point is re-scanned exactly.
constexpr double kExpPadW = 1e300;   // pad record: s = 1e300, finite

__device__ __forceinline__ double embed_index(double v, unsigned j)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    unsigned lo;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)b), "s"(kSmallIdxMask), "v"(j));
    b = (b & 0xFFFFFFFF00000000ull) | lo;
    return __longlong_as_double((long long)b);
}

// Exact (oracle-order) scan from the records: q = -X/2 exactly (a power-of-two scaling).
__device__ __noinline__ uint32_t exact_scan_small_rec(double ax, double ay, double az, const double4 *rec, unsigned m)
{
    double bd = INFINITY, bk = INFINITY;
    uint32_t bj = 0xFFFFFFFFu;
    for (unsigned j = 0; j < m; ++j) {
        const double4 v = rec[j];
        const double d = d2_exact(ax, ay, az, -0.5 * v.x, -0.5 * v.y, -0.5 * v.z);
        if (d < bd) {
            bd = d;
            const double key = key_of(d);
            if (key < bk) {
                bk = key;
                bj = j;
            }
        }
    }
    return bj;
}

template <int PPL>
__device__ __forceinline__ void scan_small_expanded(const double (&ax)[PPL], const double (&ay)[PPL], const double (&az)[PPL],
                                                    const double4 *rec, unsigned m, unsigned m_pad, double qmax2, uint32_t (&bj)[PPL])
{
    double m1[PPL], m2[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) m1[s] = m2[s] = INFINITY;
    for (unsigned j0 = 0; j0 < m_pad; j0 += 4) {
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) {
            const double4 v = rec[j0 + u];
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const double t = embed_index(__builtin_fma(az[s], v.z, __builtin_fma(ay[s], v.y, __builtin_fma(ax[s], v.x, v.w))), j0 + u);
                m2[s] = vmin64(m2[s], vmax64(m1[s], t));
                m1[s] = vmin64(m1[s], t);
            }
        }
    }
    const bool range_ok = qmax2 < 1e280;   // false also for NaN
    const double qmax = sqrt(qmax2);
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const unsigned j = (unsigned)__double_as_longlong(m1[s]) & ~kSmallIdxMask;
        const double pn = (ax[s] * ax[s] + ay[s] * ay[s]) + az[s] * az[s];
        const double E = 0x1p-52 * (11.0 * sqrt(pn) * qmax + 5.0 * qmax2);
        const double tol = 2.0 * E + 0x1p-38 * (__builtin_fabs(m1[s]) + pn) + 1e-300;
        const bool unique = range_ok && pn < 1e280 && (m2[s] - m1[s] > tol) && __builtin_fabs(m1[s]) < 1e290 && j < m;
        bj[s] = j;
        if (!unique) bj[s] = exact_scan_small_rec(ax[s], ay[s], az[s], rec, m);
    }
}

// --- seeded form (batch default from the second iteration on): the previous iteration's match is a candidate, so a row
// knows a bound bd on its answer's d^2 before it looks at a single target, and the engine's event scheme (search_kernel) works
// here too -- no index is embedded and no runner-up is tracked.  The fast path is the three FMAs of the expanded form on
// records whose fourth word is |q|^2 + C, and half a v_min3_u32 on the HIGH words of two such values: 3.5 instructions per
// pair instead of 7.  C (a power of two >= 4 max(|a|^2, |q|^2), fixed when the second iteration starts) keeps every value
// positive, so that they order like their high words:
//     f_j = fl(|q_j|^2 + C - 2 a.q_j),   F_j = d^2_j - |a|^2 + C exactly,   |f_j - F_j| <= E = 2^-52 (11 |a| qmax + 5 qmax^2 + 8 C)
// (the expanded form's bound with the roundings of the offset: the record's word, three FMAs on values up to ~2C, and the
// threshold's own arithmetic).  A target that can beat or tie the row's best has an oracle d^2 <= bd, hence
//     f_j <= bd (1 + 2^-50) - |a|^2 (1 - 2^-50) + C + E =: thr,
// and a block of kBatchCB targets whose smallest high word is <= thr's is re-scanned in the oracle's arithmetic, in target
// order, with the engine's own (key, index) rules (rescan_block): the indices are the oracle's.  f_j >= C - |a|^2 - E > 0 is
// checked per row (C > 2 |a|^2 + 4 E); a row that fails it -- or a registration whose range defeats the bound -- is scanned
// exactly, as the expanded form does.  While the clouds still move a tenth of the rows re-scan a block or two per iteration.
constexpr int kBatchCB = 8;

// rescan_block (search_common.h) on the records: q = -X/2 exactly.  The empty asm ties the loads to this call: without it the
// branch-free first pass over the block -- whose records the caller has just loaded -- was hoisted in front of the event
// test and ran for every block, doubling the fast path.  (As a real call it cost 0.3 ms of 13.5 and 760 MB of register
// spills per launch around the call sites.)
// Records from m on are padding (coordinates 0, a huge fourth word): no candidates.
__device__ __forceinline__ void rescan_block_rec(double ax, double ay, double az, const double4 *rec, unsigned j0, unsigned m, double &bd,
                                              double &bk, uint32_t &bj, double &dbest)
{
    asm volatile("" : "+v"(j0));
    double dm = INFINITY, d2nd = INFINITY;
    int um = 0;
#pragma unroll
    for (int u = 0; u < kBatchCB; ++u) {
        const double4 v = rec[j0 + u];
        const double d = (j0 + u < m) ? d2_exact(ax, ay, az, -0.5 * v.x, -0.5 * v.y, -0.5 * v.z) : INFINITY;
        d2nd = __builtin_fmin(d2nd, __builtin_fmax(d, dm));
        um = (d < dm) ? u : um;
        dm = __builtin_fmin(dm, d);
    }
    if (!(dm < bd)) return;
    if (d2nd > __builtin_fma(dm, 1.0 + 0x1p-48, 1e-300)) {
        bd = dm;
        const double key = key_of(dm);
        const uint32_t j = j0 + (unsigned)um;
        if (key < bk || (key == bk && j < bj)) {
            bk = key;
            bj = j;
            dbest = dm;
        }
        return;
    }
#pragma unroll 1
    for (int u = 0; u < kBatchCB; ++u) {
        const double4 v = rec[j0 + u];
        const double d = (j0 + u < m) ? d2_exact(ax, ay, az, -0.5 * v.x, -0.5 * v.y, -0.5 * v.z) : INFINITY;
        if (d < bd) {
            bd = d;
            const double key = key_of(d);
            const uint32_t j = j0 + (unsigned)u;
            if (key < bk || (key == bk && j < bj)) {
                bk = key;
                bj = j;
                dbest = d;
            }
        }
    }
}

// bj: in, the previous iteration's matches (the seeds); out, this iteration's.  rec[j].w = |q_j|^2 + C.
template <int PPL>
__device__ __forceinline__ void scan_small_seeded(const double (&ax)[PPL], const double (&ay)[PPL], const double (&az)[PPL],
                                                  const double4 *rec, unsigned m, unsigned m_pad, double qmax2, double C, uint32_t (&bj)[PPL])
{
    double bd[PPL], bk[PPL], off[PPL];
    uint32_t bn[PPL], sblk[PPL], thr1[PPL], mh[PPL];   // thr1 = the threshold's high word + 1: an event is mh < thr1 (0: never)
    bool ok[PPL];
    const bool range_ok = qmax2 < 1e280 && C < 1e290;   // false also for NaN
    const double qmax = sqrt(qmax2);
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const double pn = (ax[s] * ax[s] + ay[s] * ay[s]) + az[s] * az[s];
        const double E = 0x1p-52 * (11.0 * sqrt(pn) * qmax + 5.0 * qmax2 + 8.0 * C);
        ok[s] = range_ok && pn < 1e280 && C > 2.0 * pn + 4.0 * E;   // false also for NaN
        off[s] = (C - pn * (1.0 - 0x1p-50)) + E;
        // the seed, and its own block up front (seed_init, search_common.h)
        bk[s] = INFINITY;
        bn[s] = 0xFFFFFFFFu;
        sblk[s] = 0xFFFFFFFFu;
        bd[s] = INFINITY;
        const uint32_t js = bj[s];
        if (ok[s] && js < m) {
            const double4 v = rec[js];
            const double d = d2_exact(ax[s], ay[s], az[s], -0.5 * v.x, -0.5 * v.y, -0.5 * v.z);
            if (d < INFINITY) {
                bk[s] = key_of(d);
                bn[s] = js;
                double dbest = d, b0 = __builtin_fma(d, 1.0 + 0x1p-48, 1e-300);
                rescan_block_rec(ax[s], ay[s], az[s], rec, (js / kBatchCB) * kBatchCB, m, b0, bk[s], bn[s], dbest);
                sblk[s] = js / kBatchCB;
                bd[s] = __builtin_fma(dbest, 1.0 + 0x1p-48, 1e-300);
            }
        }
        const double thr = bd[s] < INFINITY ? __builtin_fma(bd[s], 1.0 + 0x1p-50, off[s]) * (1.0 + 0x1p-50) : INFINITY;
        thr1[s] = !ok[s] ? 0u : (thr < INFINITY ? (uint32_t)__double2hiint(thr) + 1u : 0x7FF00000u);
        mh[s] = thr1[s];
    }
    for (unsigned j0 = 0; j0 < m_pad; j0 += kBatchCB) {
#pragma unroll
        for (unsigned u = 0; u < (unsigned)kBatchCB; u += 2) {
            const double4 v0 = rec[j0 + u], v1 = rec[j0 + u + 1];
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const double f0 = __builtin_fma(az[s], v0.z, __builtin_fma(ay[s], v0.y, __builtin_fma(ax[s], v0.x, v0.w)));
                const double f1 = __builtin_fma(az[s], v1.z, __builtin_fma(ay[s], v1.y, __builtin_fma(ax[s], v1.x, v1.w)));
                mh[s] = umin3(mh[s], (uint32_t)__double2hiint(f0), (uint32_t)__double2hiint(f1));
            }
        }
        const unsigned blk = j0 / kBatchCB;
        bool ev = false;
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            if (blk == sblk[s]) mh[s] = thr1[s];
            ev |= (mh[s] < thr1[s]);
        }
        if (ev) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                if (mh[s] < thr1[s]) {
                    double dbest;
                    rescan_block_rec(ax[s], ay[s], az[s], rec, j0, m, bd[s], bk[s], bn[s], dbest);
                    const double thr = bd[s] < INFINITY ? __builtin_fma(bd[s], 1.0 + 0x1p-50, off[s]) * (1.0 + 0x1p-50) : INFINITY;
                    thr1[s] = thr < INFINITY ? (uint32_t)__double2hiint(thr) + 1u : 0x7FF00000u;
                    mh[s] = thr1[s];
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < PPL; ++s) bj[s] = ok[s] ? bn[s] : exact_scan_small_rec(ax[s], ay[s], az[s], rec, m);
}

// ------------------------------------------------------------------------------------------------
// multi-registration mode: one workgroup = one whole registration, every iteration on-chip
// ------------------------------------------------------------------------------------------------
// BASELINE config 5 (thousands of small independent registrations).  The centred target lives in LDS (SoA),
// the source points in registers (PPL per lane), and the loop centroid -> search (scan_small) -> covariance ->
// solve -> apply runs `iters` times without touching HBM.  All reductions have the shape the engine uses (1024-row chunks:
// 4 rows per lane in order, wave butterfly, waves in order; chunks in order), so a registration computed here is
// bit-identical to the same registration run through icp_amd_run.
// STRICT (ICP_AMD_STRICT_ORDER=1; difference form on SoA targets only): the three centroids' and the nine covariance sums run
// sequentially over the rows, one lane per output, as the reference's do (src/libalg/mean.cpp:106-131, src/cpu/icp.cpp:115-129)
// and as the engine's strict order does (strict_centroid_kernel / strict_cov_kernel): Qc, Pc and cov carry the reference's
// bits.  The running cloud and the indices are staged in LDS for that (3 n doubles + n words behind the solve record).
template <int PPL, int MODE, bool STRICT = false>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(PPL <= 2 ? 4 : PPL == 4 ? 3 : 2))) void batch_icp_kernel(const double *__restrict__ P_in, const double *__restrict__ Q_in, unsigned n,
                                                        unsigned m, unsigned m_pad, unsigned iters, double *__restrict__ P_out,
                                                        double *__restrict__ rec_out, uint32_t *__restrict__ idx_out)
{
    extern __shared__ double lds[];
    constexpr bool EXP = MODE == kExpanded || MODE == kSeeded;   // targets as {-2q, |q|^2} records (4 doubles) instead of SoA (3 doubles)
    constexpr bool SEEDED = MODE == kSeeded;                     // from the second iteration on: scan_small_seeded
    double *sx = lds, *sy = lds + m_pad, *sz = lds + 2 * (size_t)m_pad;
    double4 *trec = (double4 *)lds;
    double *red = lds + (EXP ? 4 : 3) * (size_t)m_pad;  // 4*12: scratch of block_sum
    double *bc = red + 48;                  // [0..2] Pc, [4..15] cov sums, [16..18] Qc, [20..47] solve record
    constexpr int NCH = (PPL + 3) / 4;      // 1024-row chunks covered by the PPL*256 slots
    constexpr int KPC = PPL < 4 ? PPL : 4;  // slots per chunk: clouds of up to 256 / 512 rows keep 1 / 2 rows per lane -- the rows a
                                            // 4-slot lane would add to them do not exist, so the chunk sum has the same terms in the same order
    const unsigned tid = threadIdx.x;
    static_assert(!STRICT || MODE == kExact, "strict order reads the targets from the SoA layout");
    const unsigned np = (n + 1u) & ~1u;     // STRICT: rows of the staged cloud (sp[k * np + i]), then the staged indices
    double *sp = bc + 48;
    uint32_t *sj = (uint32_t *)(sp + 3 * (size_t)np);
    (void)sp; (void)sj;
    const double *Pb = P_in + (size_t)blockIdx.x * n * 3;
    const double *Qb = Q_in + (size_t)blockIdx.x * m * 3;

    // Q_center = Q.mean(0); Q -= Q_center (src/cpu/icp.cpp:197-198)
    double tq = 0.0;
    if constexpr (STRICT) {
        if (tid < 3) {
            double r = 0.0;
            for (unsigned j = 0; j < m; ++j) r += Qb[(size_t)j * 3 + tid];
            bc[16 + tid] = r / (double)m;
        }
    } else
    for (unsigned c0 = 0; c0 < m; c0 += kChunk) {
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned i = c0 + k * kWG + tid;
            if (i < m) {
                acc[0] += Qb[(size_t)i * 3 + 0];
                acc[1] += Qb[(size_t)i * 3 + 1];
                acc[2] += Qb[(size_t)i * 3 + 2];
            }
        }
        __syncthreads();
        const double t = block_sum<3>(acc, red);
        if (tid < 3) tq += t;
    }
    if (!STRICT && tid < 3) bc[16 + tid] = tq / (double)m;
    __syncthreads();
    const double qc0 = bc[16], qc1 = bc[17], qc2 = bc[18];
    double qn_max = 0.0;
    for (unsigned j = tid; j < m_pad; j += kWG) {
        const bool in = j < m;
        if constexpr (EXP) {
            double4 r = make_double4(0.0, 0.0, 0.0, kExpPadW);
            if (in) {
                const double x = Qb[(size_t)j * 3 + 0] - qc0, y = Qb[(size_t)j * 3 + 1] - qc1, z = Qb[(size_t)j * 3 + 2] - qc2;
                const double w = (x * x + y * y) + z * z;
                r = make_double4(-2.0 * x, -2.0 * y, -2.0 * z, w);
                qn_max = (w > qn_max || w != w) ? w : qn_max;   // a NaN sticks: the range test then fails
            }
            trec[j] = r;
        } else {
            sx[j] = in ? Qb[(size_t)j * 3 + 0] - qc0 : kSmallPad;
            sy[j] = in ? Qb[(size_t)j * 3 + 1] - qc1 : kSmallPad;
            sz[j] = in ? Qb[(size_t)j * 3 + 2] - qc2 : kSmallPad;
        }
    }
    if constexpr (EXP) {   // max |q|^2 of this registration's centred target -> bc[3]
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double other = __shfl_xor(qn_max, o, 64);
            qn_max = (other > qn_max || other != other) ? other : qn_max;
        }
        __syncthreads();   // red is free again (block_sum above)
        if ((tid & 63) == 0) red[tid >> 6] = qn_max;
        __syncthreads();
        if (tid == 0) {
            double v = red[0];
            for (int w = 1; w < kWG / 64; ++w) v = (red[w] > v || red[w] != red[w]) ? red[w] : v;
            bc[3] = v;
        }
    }
    double px[PPL], py[PPL], pz[PPL];
    uint32_t bj[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const unsigned i = s * kWG + tid;
        const bool in = i < n;
        px[s] = in ? Pb[(size_t)i * 3 + 0] : 0.0;
        py[s] = in ? Pb[(size_t)i * 3 + 1] : 0.0;
        pz[s] = in ? Pb[(size_t)i * 3 + 2] : 0.0;
        bj[s] = 0xFFFFFFFFu;
    }
    __syncthreads();

    for (unsigned it = 0; it < iters; ++it) {
        // P_center = P_copy.mean(0) (src/cpu/icp.cpp:206)
        double tc = 0.0;
        if constexpr (STRICT) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const unsigned i = s * kWG + tid;
                if (i < n) { sp[i] = px[s]; sp[np + i] = py[s]; sp[2 * (size_t)np + i] = pz[s]; }
            }
            __syncthreads();
            if (tid < 3) {
                double r = 0.0;
                for (unsigned i = 0; i < n; ++i) r += sp[tid * (size_t)np + i];
                bc[tid] = r / (double)n;
            }
        } else
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < KPC; ++k) {
                const int s = c * 4 + k;
                if (s * kWG + tid < n) {
                    acc[0] += px[s];
                    acc[1] += py[s];
                    acc[2] += pz[s];
                }
            }
            __syncthreads();
            const double t = block_sum<3>(acc, red);
            if (tid < 3) tc += t;
        }
        if (!STRICT && tid < 3) bc[tid] = tc / (double)n;
        __syncthreads();
        const double c0 = bc[0], c1 = bc[1], c2 = bc[2];

        // correspondences (src/cpu/icp.cpp:210)
        double ax[PPL], ay[PPL], az[PPL];
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            ax[s] = px[s] - c0;
            ay[s] = py[s] - c1;
            az[s] = pz[s] - c2;
        }
        if constexpr (SEEDED) {
            if (it == 1) {
                // C = a power of two >= 4 max(|a|^2 over the rows, |q|^2 over the targets), added to every record's fourth word
                // (the first iteration, which read the plain |q|^2, is over; nothing else reads that word)
                double pm = 0.0;
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    const double pn = (ax[s] * ax[s] + ay[s] * ay[s]) + az[s] * az[s];
                    pm = (s * kWG + tid < n && (pn > pm || pn != pn)) ? pn : pm;   // a NaN sticks
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const double other = __shfl_xor(pm, o, 64);
                    pm = (other > pm || other != other) ? other : pm;
                }
                if ((tid & 63) == 0) red[tid >> 6] = pm;
                __syncthreads();
                if (tid == 0) {
                    double v = bc[3];
                    for (int w = 0; w < kWG / 64; ++w) v = (red[w] > v || red[w] != red[w]) ? red[w] : v;
                    bc[19] = (v > 0.0 && v < 1e280) ? __builtin_amdgcn_ldexp(1.0, __builtin_amdgcn_frexp_exp(v) + 2) : INFINITY;
                }
                __syncthreads();
                const double C = bc[19];
                if (C < INFINITY)
                    for (unsigned j = tid; j < m; j += kWG) trec[j].w += C;
                __syncthreads();
            }
            if (it == 0) scan_small_expanded<PPL>(ax, ay, az, trec, m, m_pad, bc[3], bj);
            else if (bc[19] < INFINITY) scan_small_seeded<PPL>(ax, ay, az, trec, m, m_pad, bc[3], bc[19], bj);
            else {   // no usable offset (a degenerate or out-of-range registration): every row exactly
#pragma unroll
                for (int s = 0; s < PPL; ++s) bj[s] = exact_scan_small_rec(ax[s], ay[s], az[s], trec, m);
            }
        } else if constexpr (EXP)
            scan_small_expanded<PPL>(ax, ay, az, trec, m, m_pad, bc[3], bj);
        else
            scan_small<PPL, MODE>(ax, ay, az, sx, sy, sz, m, m_pad, bj);

        // cross-covariance (src/cpu/icp.cpp:214)
        double tv = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double acc[kCovW];
#pragma unroll
            for (int k = 0; k < kCovW; ++k) acc[k] = 0.0;
#pragma unroll
            for (int k = 0; k < KPC; ++k) {
                const int s = c * 4 + k;
                const uint32_t j = bj[s];
                if (s * kWG + tid < n && j < m) {
                    const double p0 = ax[s], p1 = ay[s], p2 = az[s];
                    double q0, q1, q2;
                    if constexpr (EXP) {
                        const double4 v = trec[j];
                        q0 = -0.5 * v.x; q1 = -0.5 * v.y; q2 = -0.5 * v.z;   // exact: power-of-two scaling
                    } else {
                        q0 = sx[j]; q1 = sy[j]; q2 = sz[j];
                    }
                    acc[0] += q0 * p0; acc[1] += q0 * p1; acc[2] += q0 * p2;
                    acc[3] += q1 * p0; acc[4] += q1 * p1; acc[5] += q1 * p2;
                    acc[6] += q2 * p0; acc[7] += q2 * p1; acc[8] += q2 * p2;
                    acc[9] += d2_exact(p0, p1, p2, q0, q1, q2);
                }
            }
            __syncthreads();
            const double t = block_sum<kCovW>(acc, red);
            if (tid < kCovW) tv += t;
        }
        if (tid < kCovW) bc[4 + tid] = tv;
        if constexpr (STRICT) {   // the nine sums again, row by row (slot 9, the sum of d^2 behind the rms, keeps the tree's value)
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const unsigned i = s * kWG + tid;
                if (i < n) sj[i] = bj[s];
            }
            __syncthreads();
            if (tid < 9) {
                const unsigned a = tid / 3, b = tid % 3;
                const double cb = bc[b];
                const double *qa = a == 0 ? sx : (a == 1 ? sy : sz);
                double cov = 0.0;
                for (unsigned i = 0; i < n; ++i) {
                    const uint32_t j = sj[i];
                    if (!(j < m)) continue;
                    const double p = sp[b * (size_t)np + i] - cb;
                    cov = cov + qa[j] * p;
                }
                bc[4 + tid] = cov;
            }
        }
        __syncthreads();
        // svd, R, t (src/cpu/icp.cpp:218-223)
        if (tid == 0) solve_record(bc + 4, bc, bc + 16, (double)n, bc + 20);
        __syncthreads();
        // P_copy = P_copy.dot(R.transpose()) + t (src/cpu/icp.cpp:225)
        const double *R = bc + 20 + 12, *t = bc + 20 + 21;
        const double r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
        const double t0 = t[0], t1 = t[1], t2 = t[2];
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            const double x = px[s], y = py[s], z = pz[s];
            px[s] = ((x * r00 + y * r01) + z * r02) + t0;
            py[s] = ((x * r10 + y * r11) + z * r12) + t1;
            pz[s] = ((x * r20 + y * r21) + z * r22) + t2;
        }
        __syncthreads();  // bc is rewritten by the next iteration
    }
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const unsigned i = s * kWG + tid;
        if (i < n) {
            double *o = P_out + ((size_t)blockIdx.x * n + i) * 3;
            o[0] = px[s];
            o[1] = py[s];
            o[2] = pz[s];
            if (idx_out) idx_out[(size_t)blockIdx.x * n + i] = bj[s];
        }
    }
    if (rec_out && tid < kSolveW) rec_out[(size_t)blockIdx.x * kSolveW + tid] = iters ? bc[20 + tid] : 0.0;
}

__global__ __launch_bounds__(kWG) void fourth_root_kernel(const double *__restrict__ d2, size_t n, double *__restrict__ key)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i < n) key[i] = key_of(d2[i]);
}

// ------------------------------------------------------------------------------------------------
// launchers (shapes are validated by the engine before any launch)
// ------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(size_t n, size_t per) { return (unsigned)((n + per - 1) / per); }

hipError_t colsum_partials(hipStream_t s, const double *aos, Soa3 soa, size_t n, double *partial)
{
    if (n == 0) return hipSuccess;
    return launch(colsum_partials_kernel, dim3(blocks_for(n, kChunk)), dim3(kWG), 0, s, aos, soa.x, soa.y, soa.z, n, partial);
}

hipError_t finalize_centroid(hipStream_t s, const double *partial, size_t count, double denom, double *out3)
{
    return launch(finalize_centroid_kernel, dim3(1), dim3(kWG), 0, s, partial, count, denom, out3);
}

hipError_t aos_to_soa(hipStream_t s, const double *aos, size_t n, size_t n_pad, const double *c3, double pad, Soa3 out)
{
    if (n_pad == 0) return hipSuccess;
    return launch(aos_to_soa_kernel, dim3(blocks_for(n_pad, kWG)), dim3(kWG), 0, s, aos, n, n_pad, c3, pad, out.x, out.y, out.z);
}

__global__ __launch_bounds__(kWG) void build_rec_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ z, size_t rows, double4 *__restrict__ rec)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i < rows) rec[i] = make_double4(x[i], y[i], z[i], 0.0);
}

hipError_t build_rec(hipStream_t s, Soa3 soa, size_t rows)
{
    if (rows == 0 || !soa.rec) return hipErrorInvalidValue;
    return launch(build_rec_kernel, dim3(blocks_for(rows, kWG)), dim3(kWG), 0, s, soa.x, soa.y, soa.z, rows, soa.rec);
}

__global__ __launch_bounds__(kWG) void build_source_rec_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                               const double *__restrict__ z, size_t rows, double4 *__restrict__ rec,
                                                               size_t stride)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= rows) return;
    rec[i] = make_double4(x[i], y[i], z[i], 0.0);
    rec[stride + i] = make_double4(0.0, 0.0, 0.0, __longlong_as_double(0xFFFFFFFFll));   // no match yet, no certificate
}

__global__ __launch_bounds__(kWG) void source_rec_to_soa_kernel(const double4 *__restrict__ rec, size_t rows, double *__restrict__ x,
                                                                double *__restrict__ y, double *__restrict__ z)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= rows) return;
    const double4 v = rec[i];
    x[i] = v.x;
    y[i] = v.y;
    z[i] = v.z;
}

hipError_t source_rec_to_soa(hipStream_t s, Soa3 P, size_t rows)
{
    if (rows == 0) return hipSuccess;
    if (!P.rec) return hipErrorInvalidValue;
    return launch(source_rec_to_soa_kernel, dim3(blocks_for(rows, kWG)), dim3(kWG), 0, s, P.rec, rows, P.x, P.y, P.z);
}

hipError_t build_source_rec(hipStream_t s, Soa3 P, size_t rows)
{
    if (rows == 0 || !P.rec || P.rec_stride < rows) return hipErrorInvalidValue;
    return launch(build_source_rec_kernel, dim3(blocks_for(rows, kWG)), dim3(kWG), 0, s, P.x, P.y, P.z, rows, P.rec, P.rec_stride);
}

hipError_t soa_to_aos(hipStream_t s, Soa3 in, size_t n, double *aos)
{
    if (n == 0) return hipSuccess;
    return launch(soa_to_aos_kernel, dim3(blocks_for(n, kWG)), dim3(kWG), 0, s, in.x, in.y, in.z, n, aos);
}
hipError_t aos_gather_to_soa(hipStream_t s, const double *aos, const uint32_t *perm, size_t n, size_t n_pad, double pad, Soa3 out)
{
    if (n_pad == 0) return hipSuccess;
    return launch(aos_gather_to_soa_kernel, dim3(blocks_for(n_pad, kWG)), dim3(kWG), 0, s, aos, perm, n, n_pad, pad, out.x, out.y, out.z);
}
hipError_t soa_scatter_to_aos(hipStream_t s, Soa3 in, const uint32_t *perm, size_t n, double *aos)
{
    if (n == 0) return hipSuccess;
    return launch(soa_scatter_to_aos_kernel, dim3(blocks_for(n, kWG)), dim3(kWG), 0, s, in.x, in.y, in.z, perm, n, aos);
}
hipError_t scatter_u32(hipStream_t s, const uint32_t *src, const uint32_t *perm, size_t n, uint32_t *dst)
{
    if (n == 0) return hipSuccess;
    return launch(scatter_u32_kernel, dim3(blocks_for(n, kWG)), dim3(kWG), 0, s, src, perm, n, dst);
}

// Split geometry shared by both search kernels: the scanned range is m rounded up to kSplitGran targets (pads inside
// it compare as +inf), cut into q_splits pieces of a whole number of kSplitGran-blocks; trailing pieces may be empty.
size_t search_scan_rows(size_t m) { return (m + kSplitGran - 1) / kSplitGran * kSplitGran; }
size_t search_rows_per_split(size_t m, int q_splits)
{
    const size_t blocks = search_scan_rows(m) / kSplitGran;
    return (blocks + (size_t)q_splits - 1) / (size_t)q_splits * kSplitGran;
}

template <int PPL, int CB>
static hipError_t search_launch(hipStream_t s, int variant, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, size_t m,
                                size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key, uint32_t *out_idx)
{
    const unsigned m_scan = (unsigned)search_scan_rows(m);
    const unsigned q_per_split = (unsigned)search_rows_per_split(m, q_splits);
    dim3 grid((unsigned)(n_pad / (kWG * PPL)), (unsigned)q_splits);
    if (variant == kFma)
        return launch(search_kernel<PPL, kFma, CB>, grid, dim3(kWG), 0, s, P.x, P.y, P.z, pc, Qz.x, Qz.y, Qz.z, (unsigned)n_pad,
                      (unsigned)m, m_scan, q_per_split, idx_prev, out_key, out_idx);
    return launch(search_kernel<PPL, kExact, CB>, grid, dim3(kWG), 0, s, P.x, P.y, P.z, pc, Qz.x, Qz.y, Qz.z, (unsigned)n_pad,
                  (unsigned)m, m_scan, q_per_split, idx_prev, out_key, out_idx);
}

hipError_t search(hipStream_t s, int variant, int ppl, int cb, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, size_t m,
                  size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key, uint32_t *out_idx)
{
    if (n_pad == 0 || m_pad == 0 || n_pad % kChunk || m_pad % kTileQ || q_splits < 1 || m > m_pad) return hipErrorInvalidValue;
#define ICPK_CASE(PPL_, CB_) \
    if (ppl == PPL_ && cb == CB_) return search_launch<PPL_, CB_>(s, variant, P, n_pad, pc, Qz, m, m_pad, q_splits, idx_prev, out_key, out_idx);
    ICPK_CASE(1, 8) ICPK_CASE(1, 16) ICPK_CASE(1, 32)
    ICPK_CASE(2, 8) ICPK_CASE(2, 16) ICPK_CASE(2, 32)
    ICPK_CASE(4, 8) ICPK_CASE(4, 16) ICPK_CASE(4, 32)
#undef ICPK_CASE
    return hipErrorInvalidValue;
}

hipError_t build_qrec(hipStream_t s, Soa3 Qz, size_t m, size_t m_pad, void *qrec, unsigned long long *qmax2_bits)
{
    if (m_pad == 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(qmax2_bits, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    return launch(build_qrec_kernel, dim3(blocks_for(m_pad, kWG)), dim3(kWG), 0, s, Qz.x, Qz.y, Qz.z, m, m_pad, (QRec *)qrec, qmax2_bits);
}

template <int PPL, int CB, bool PK>
static hipError_t search_f32_launch(hipStream_t s, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, const void *qrec, double qmax,
                                    size_t m, size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key, uint32_t *out_idx)
{
    const unsigned m_scan = (unsigned)search_scan_rows(m);
    const unsigned q_per_split = (unsigned)search_rows_per_split(m, q_splits);
    dim3 grid((unsigned)(n_pad / (kWG * PPL)), (unsigned)q_splits);
    return launch(search_f32_kernel<PPL, CB, PK>, dim3(grid), dim3(kWG), 0, s, P.x, P.y, P.z, pc, Qz.x, Qz.y, Qz.z, (const QRec *)qrec, qmax, (unsigned)n_pad,
                                                    (unsigned)m, m_scan, q_per_split, idx_prev, out_key, out_idx);
}

hipError_t search_f32(hipStream_t s, int ppl, int cb, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, const void *qrec, double qmax,
                      size_t m, size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key, uint32_t *out_idx)
{
    if (n_pad == 0 || m_pad == 0 || n_pad % (kWG * 8) || m_pad % kTileQF || q_splits < 1 || m > m_pad || !(qmax >= 0.0))
        return hipErrorInvalidValue;
    static const bool pk = knob_int("pk", 1) != 0;
#define ICPK_CASE(PPL_, CB_) \
    if (ppl == PPL_ && cb == CB_) \
        return pk ? search_f32_launch<PPL_, CB_, true>(s, P, n_pad, pc, Qz, qrec, qmax, m, m_pad, q_splits, idx_prev, out_key, out_idx) \
                  : search_f32_launch<PPL_, CB_, false>(s, P, n_pad, pc, Qz, qrec, qmax, m, m_pad, q_splits, idx_prev, out_key, out_idx);
    ICPK_CASE(2, 32) ICPK_CASE(4, 16) ICPK_CASE(4, 32) ICPK_CASE(8, 16) ICPK_CASE(8, 32)
#undef ICPK_CASE
    return hipErrorInvalidValue;
}

hipError_t merge_cov(hipStream_t s, Soa3 P, size_t n, size_t n_pad, const double *pc, Soa3 Qz, size_t m, int q_splits,
                     const double *keys, const uint32_t *idxs, const double2 *grid_rec, bool result_in_rec, uint32_t *idx_out,
                     double reject_d2, double *partial)
{
    if (n == 0) return hipSuccess;
    if (result_in_rec && !P.rec) return hipErrorInvalidValue;
    if (result_in_rec) {
        return launch(cov_rec_kernel, dim3(blocks_for(n, kChunk)), dim3(kWG), 0, s, P.rec, P.rec_stride, n, pc, m, idx_out, reject_d2, partial);
    }
    return launch(cov_kernel<true>, dim3(blocks_for(n, kChunk)), dim3(kWG), 0, s, P.x, P.y, P.z, n, n_pad, pc, Qz.x, Qz.y, Qz.z, Qz.rec, m, q_splits, keys,
                                                           idxs, grid_rec, nullptr, idx_out, P.rec, P.rec_stride, reject_d2, partial);
}

hipError_t cov_from_idx(hipStream_t s, Soa3 P, size_t n, const double *pc, Soa3 Qz, size_t m, const uint32_t *idx,
                        double *partial)
{
    if (n == 0) return hipSuccess;
    return launch(cov_kernel<false>, dim3(blocks_for(n, kChunk)), dim3(kWG), 0, s, P.x, P.y, P.z, n, 0, pc, Qz.x, Qz.y, Qz.z, Qz.rec, m, 1, nullptr, nullptr,
                                                            nullptr, idx, nullptr, nullptr, (size_t)0, INFINITY, partial);
}

// ---- strict order: the reference's sequential reductions, one lane per output ----
__global__ __launch_bounds__(64) void strict_centroid_kernel(const double *__restrict__ aos, const double *__restrict__ x,
                                                             const double *__restrict__ y, const double *__restrict__ z,
                                                             const double4 *__restrict__ rec, size_t n, double *__restrict__ out3)
{
    const int k = threadIdx.x;
    if (k >= 3) return;
    double r = 0.0;
    if (aos) {
        for (size_t i = 0; i < n; ++i) r += aos[i * 3 + k];
    } else if (rec) {
        for (size_t i = 0; i < n; ++i) {
            const double4 v = rec[i];
            r += k == 0 ? v.x : (k == 1 ? v.y : v.z);
        }
    } else {
        const double *c = k == 0 ? x : (k == 1 ? y : z);
        for (size_t i = 0; i < n; ++i) r += c[i];
    }
    out3[k] = r / (double)n;
}

__global__ __launch_bounds__(64) void strict_cov_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                        const double *__restrict__ pz, const double4 *__restrict__ prec, size_t n,
                                                        const double *__restrict__ pc, const double *__restrict__ qx,
                                                        const double *__restrict__ qy, const double *__restrict__ qz, size_t m,
                                                        const uint32_t *__restrict__ idx, double *__restrict__ partial, size_t chunks)
{
    const int k = threadIdx.x;
    if (k >= 9) return;
    const int a = k / 3, b = k % 3;
    const double cb = pc[b];
    const double *qa = a == 0 ? qx : (a == 1 ? qy : qz);
    const double *pb = b == 0 ? px : (b == 1 ? py : pz);
    double cov = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t j = idx[i];
        if (!(j < m)) continue;   // (the reference has a match for every row of a finite cloud)
        double p;
        if (prec) {
            const double4 v = prec[i];
            p = (b == 0 ? v.x : (b == 1 ? v.y : v.z)) - cb;
        } else {
            p = pb[i] - cb;
        }
        cov = cov + qa[j] * p;   // tmp = 0 + q_a p_b; tmp *= 1.0; cov += tmp
    }
    partial[k] = cov;
    for (size_t c = 1; c < chunks; ++c) partial[c * kCovW + k] = 0.0;
}

hipError_t strict_centroid(hipStream_t s, const double *aos, Soa3 P, size_t n, double *out3)
{
    if (n == 0) return hipErrorInvalidValue;
    return launch(strict_centroid_kernel, dim3(1), dim3(64), 0, s, aos, P.x, P.y, P.z, aos ? nullptr : P.rec, n, out3);
}

hipError_t strict_cov(hipStream_t s, Soa3 P, size_t n, const double *pc, Soa3 Qz, size_t m, const uint32_t *idx, double *partial,
                      size_t chunks)
{
    if (n == 0 || chunks == 0) return hipErrorInvalidValue;
    return launch(strict_cov_kernel, dim3(1), dim3(64), 0, s, P.x, P.y, P.z, P.rec, n, pc, Qz.x, Qz.y, Qz.z, m, idx, partial, chunks);
}

hipError_t solve(hipStream_t s, const double *partial, size_t count, const double *pc, const double *qc, double n_global,
                 double *out, double *extra)
{
    return launch(solve_kernel, dim3(1), dim3(kSolveWG), 0, s, partial, count, pc, qc, n_global, out, extra);
}

hipError_t apply(hipStream_t s, Soa3 P, size_t n, const double *solve_rec, double *cen_partial)
{
    if (n == 0) return hipSuccess;
    if (P.rec)
        return launch(apply_kernel<true>, dim3(blocks_for(n, kChunk)), dim3(kWG), 0, s, P.x, P.y, P.z, n, solve_rec, P.rec, cen_partial);
    return launch(apply_kernel<false>, dim3(blocks_for(n, kChunk)), dim3(kWG), 0, s, P.x, P.y, P.z, n, solve_rec, P.rec, cen_partial);
}

size_t batch_lds_bytes(size_t m) { return (4 * ((m + 31) / 32 * 32) + 48 + 48) * sizeof(double); }   // the larger layout (records)

template <int PPL, int MODE, bool STRICT>
static hipError_t batch_launch(hipStream_t s, const double *P_in, const double *Q_in, size_t batch, size_t n, size_t m, unsigned iters,
                               double *P_out, double *rec_out, uint32_t *idx_out)
{
    size_t lds = (((MODE == kExpanded || MODE == kSeeded) ? 4 : 3) * ((m + 31) / 32 * 32) + 48 + 48) * sizeof(double);
    if (STRICT) lds += 3 * ((n + 1) & ~(size_t)1) * sizeof(double) + ((n + 1) & ~(size_t)1) * sizeof(uint32_t);
    auto kern = batch_icp_kernel<PPL, MODE, STRICT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return launch(kern, dim3((unsigned)batch), dim3(kWG), lds, s, P_in, Q_in, (unsigned)n, (unsigned)m, (unsigned)((m + 31) / 32 * 32), iters, P_out, rec_out,
                                           idx_out);
}

hipError_t batch_icp(hipStream_t s, int variant, bool strict, const double *P_in, const double *Q_in, size_t batch, size_t n, size_t m,
                     unsigned iters, double *P_out, double *rec_out, uint32_t *idx_out)
{
    if (batch == 0 || n == 0 || m == 0 || n > kBatchMaxN || m > kBatchMaxM || batch > 0x7FFFFFFFu) return hipErrorInvalidValue;
    if (strict) {   // the reference's reduction order: one instantiation per source size class (the indices do not depend on the variant)
        if (n <= 1024) return batch_launch<4, kExact, true>(s, P_in, Q_in, batch, n, m, iters, P_out, rec_out, idx_out);
        return batch_launch<8, kExact, true>(s, P_in, Q_in, batch, n, m, iters, P_out, rec_out, idx_out);
    }
#define ICPK_BATCH(PPL_) \
    return variant == kSeeded   ? batch_launch<PPL_, kSeeded, false>(s, P_in, Q_in, batch, n, m, iters, P_out, rec_out, idx_out) \
           : variant == kExpanded ? batch_launch<PPL_, kExpanded, false>(s, P_in, Q_in, batch, n, m, iters, P_out, rec_out, idx_out) \
           : variant == kFma    ? batch_launch<PPL_, kFma, false>(s, P_in, Q_in, batch, n, m, iters, P_out, rec_out, idx_out) \
                                : batch_launch<PPL_, kExact, false>(s, P_in, Q_in, batch, n, m, iters, P_out, rec_out, idx_out)
    if (n <= 256) ICPK_BATCH(1);
    if (n <= 512) ICPK_BATCH(2);
    if (n <= 1024) ICPK_BATCH(4);
    ICPK_BATCH(8);
#undef ICPK_BATCH
}

hipError_t fourth_root(hipStream_t s, const double *d2, size_t n, double *key)
{
    if (n == 0) return hipSuccess;
    return launch(fourth_root_kernel, dim3(blocks_for(n, kWG)), dim3(kWG), 0, s, d2, n, key);
}

}  // namespace icpk
