// icp_amd/csrc/search_common.h -- device helpers shared by the brute-force search kernels (kernels.hip, search_mfma.hip):
// the exact slow path, the seeding pre-pass and the fp32 record type of the pre-filters.
#pragma once
#include "exact.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace icpk {

// Slow path, entered by the lanes whose running minimum dropped inside the last block of CB targets:
// re-evaluates the block un-fused and updates the lexicographic best (key, index) exactly as a scan in target order
// would.  Invariant on exit: every target not yet scanned that can still beat (bk, bj) has d^2 < bd
// (key is a monotone function of d^2; targets arrive in ascending index order; a seed taken from the
// previous iteration may have a higher index than targets still to come, hence the index compare).
//
// The in-order scan (one branch and possibly two square roots per target, loads not overlapped) is kept for the
// rare hard case only.  First a branch-free pass finds the smallest d^2 of the block (dm, first index um) and the
// second smallest (d2nd) with the loads of eight targets in flight.  dm >= bd: a false alarm of the pre-filter,
// nothing changes.  d2nd outside the tie band of dm (the fourth root merges fewer than 16 neighbouring doubles, so a
// d^2 more than 2^-48 above dm has a strictly larger key): every other target of the block loses to um on the key,
// and the in-order scan would end with best = lexmin(best, (key(dm), um)), bd = dm -- computed directly.
template <int CB>
__device__ __forceinline__ void rescan_block(double ax, double ay, double az, const double *sx, const double *sy,
                                             const double *sz, int j0, unsigned jglobal, double &bd, double &bk, uint32_t &bj,
                                             double &dbest)
{
    double dm = INFINITY, d2nd = INFINITY;
    int um = 0;
#pragma unroll 8
    for (int u = 0; u < CB; ++u) {
        const double d = d2_exact(ax, ay, az, sx[j0 + u], sy[j0 + u], sz[j0 + u]);
        d2nd = __builtin_fmin(d2nd, __builtin_fmax(d, dm));
        um = (d < dm) ? u : um;
        dm = __builtin_fmin(dm, d);
    }
    if (!(dm < bd)) return;
    if (d2nd > __builtin_fma(dm, 1.0 + 0x1p-48, 1e-300)) {
        bd = dm;
        const double key = key_of(dm);
        const uint32_t j = jglobal + (unsigned)um;
        if (key < bk || (key == bk && j < bj)) {
            bk = key;
            bj = j;
            dbest = dm;   // exact d^2 of the current best
        }
        return;
    }
#pragma unroll 1
    for (int u = 0; u < CB; ++u) {
        const double d = d2_exact(ax, ay, az, sx[j0 + u], sy[j0 + u], sz[j0 + u]);
        if (d < bd) {
            bd = d;
            const double key = key_of(d);
            const uint32_t j = jglobal + (unsigned)u;
            if (key < bk || (key == bk && j < bj)) {
                bk = key;
                bj = j;
                dbest = d;
            }
        }
    }
}

// Seeding (both search kernels).  The previous iteration's correspondence js is a legitimate candidate; starting from
// it makes the running minimum tight from the first target on.  Two details keep that exact and cheap:
//  * targets with the SAME key and a lower index must still get through: their d^2 lies within a few ulp of the seed's
//    (the fourth root merges at most ~8 neighbouring doubles), so the running bound carries a 2^-48 margin;
//  * with that margin the seed itself would trip the event check of its own block -- once per source point per
//    launch, i.e. in nearly every block of a wave when the target range is short.  So the seed's block (if it lies in
//    this workgroup's target range) is evaluated exactly up front, by every lane for its own seed in parallel, and
//    skipped by the event check later (sblk).
// Returns the bound bd to start from; bk/bj are set to the best of {seed, seed's block}.
template <int CB>
__device__ __forceinline__ double seed_init(double ax, double ay, double az, const double *__restrict__ qx,
                                            const double *__restrict__ qy, const double *__restrict__ qz, uint32_t js, unsigned m,
                                            unsigned q0, unsigned q1, double &bk, uint32_t &bj, uint32_t &sblk)
{
    bk = INFINITY;  // oracle: min_dist = DBL_MAX on the key, i.e. any finite d^2 is accepted
    bj = 0xFFFFFFFFu;
    sblk = 0xFFFFFFFFu;
    if (js >= m) return INFINITY;
    const double d = d2_exact(ax, ay, az, qx[js], qy[js], qz[js]);
    if (!(d < INFINITY)) return INFINITY;
    bk = key_of(d);
    bj = js;
    double dbest = d;
    if (js >= q0 && js < q1) {
        const unsigned jb = (js / CB) * CB;
        double bd = __builtin_fma(d, 1.0 + 0x1p-48, 1e-300);
        rescan_block<CB>(ax, ay, az, qx + jb, qy + jb, qz + jb, 0, jb, bd, bk, bj, dbest);
        sblk = js / CB;
    }
    return __builtin_fma(dbest, 1.0 + 0x1p-48, 1e-300);
}

// Two consecutive targets a, b share one 32-byte record {a.x,b.x, a.y,b.y, a.z,b.z, a.w,b.w} (w = |q|^2), the
// operand shape of v_pk_fma_f32: one packed FMA advances both targets for one source point.
struct __attribute__((aligned(16))) QRec { float x, y, z, w; };   // half a pair record (16 bytes)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// v_min3_f32 (the compiler fuses the two fminf; an asm statement here would pin the schedule of the packed FMAs)
__device__ __forceinline__ float vmin3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

__device__ __forceinline__ float f32_above(double x)
{
    const float f = (float)x;
    return __builtin_fmaf(__builtin_fabsf(f), 0x1p-22f, f) + 1e-37f;   // at least one fp32 ulp above x
}

}  // namespace icpk
