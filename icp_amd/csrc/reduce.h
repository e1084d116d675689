// icp_amd/csrc/reduce.h -- the fixed-order sum of chunk partials, shared by the kernels that finish a reduction
// (finalize_centroid_kernel, solve_kernel in kernels.hip; the grid search when it finishes the centroid itself, grid.hip).
#pragma once
#include "kernels.h"

namespace icpk {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum NV per-thread values over the 256-thread workgroup; thread k < NV returns the total of value k.
// smem holds 4*NV doubles.  Order: butterfly inside each wave, then waves 0..3 in order.
template <int NV>
__device__ __forceinline__ double block_sum(double (&v)[NV], double *smem)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double r = wave_sum(v[k]);
        if (lane == 0) smem[w * NV + k] = r;
    }
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < NV) {
        const int k = threadIdx.x;
        tot = ((smem[0 * NV + k] + smem[1 * NV + k]) + smem[2 * NV + k]) + smem[3 * NV + k];
    }
    return tot;
}


// Fixed-order sum of `count` partial records of W doubles; thread k < W ends with the total of column k.
// Order (independent of the launch geometry and of how the records were distributed over GPUs): groups of kSumGroup
// consecutive records are summed in index order, then the group sums are summed in index order.  Up to kSumGroup
// records that is the plain in-order sum.  The groups of one pass are summed by different threads at the same time, so
// the dependent add chain is kSumGroup + count / kSumGroup long instead of count.
constexpr int kSumGroup = 32;
template <int W, int TB, int BATCH = kSumGroup>   // TB = threads of the workgroup (stage holds TB doubles); BATCH = loads in flight per thread
__device__ __forceinline__ double ordered_sum(const double *__restrict__ partial, size_t count, double *stage)
{
    constexpr int GP = TB / W;   // groups per pass
    const int col = threadIdx.x % W, gl = threadIdx.x / W;
    const size_t groups = (count + kSumGroup - 1) / kSumGroup;
    double acc = 0.0;
    for (size_t g0 = 0; g0 < groups; g0 += GP) {
        double s = 0.0;
        const size_t g = g0 + (size_t)gl;
        if (gl < GP && g < groups) {
            const size_t r0 = g * kSumGroup;
            const int recs = (int)((count - r0 < (size_t)kSumGroup) ? (count - r0) : (size_t)kSumGroup);
            const double *src = partial + r0 * W + col;
            if (recs == kSumGroup) {   // the loads of the group in flight before the first add (BATCH at a time: registers); the adds in index order
#pragma unroll
                for (int r0b = 0; r0b < kSumGroup; r0b += BATCH) {
                    double v[BATCH];
#pragma unroll
                    for (int r = 0; r < BATCH; ++r) v[r] = src[(size_t)(r0b + r) * W];
#pragma unroll
                    for (int r = 0; r < BATCH; ++r) s += v[r];
                }
            } else {
#pragma unroll 8
                for (int r = 0; r < recs; ++r) s += src[(size_t)r * W];
            }
        }
        __syncthreads();
        if (gl < GP) stage[gl * W + col] = s;
        __syncthreads();
        if (threadIdx.x < W) {
            const int ng = (int)((groups - g0 < (size_t)GP) ? (groups - g0) : (size_t)GP);
#pragma unroll 8
            for (int k = 0; k < ng; ++k) acc += stage[k * W + threadIdx.x];
        }
    }
    return acc;
}

}  // namespace icpk
