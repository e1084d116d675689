// icp_amd/csrc/exact.h -- internal: the oracle's per-pair arithmetic as device functions, shared by every search kernel.
// Translation units that include this header must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace icpk {

// The oracle's squared distance (reference src/libalg/alg.cpp:22-30 with squared_norm_2/add/add,
// basic_operations.cpp:5-13): tmp = ((0 + dx*dx) + dy*dy) + dz*dz, r = 0 + tmp; every op rounded.
__device__ __forceinline__ double d2_exact(double ax, double ay, double az, double qx, double qy, double qz)
{
    const double dx = ax - qx, dy = ay - qy, dz = az - qz;
    return (dx * dx + dy * dy) + dz * dz;
}
// Same differences, products fused into the sums: within (1+u)^6 of d2_exact; pre-filter only.
__device__ __forceinline__ double d2_fused(double ax, double ay, double az, double qx, double qy, double qz)
{
    const double dx = ax - qx, dy = ay - qy, dz = az - qz;
    return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
}
// The oracle's comparison key (src/cpu/icp.cpp:83 on top of CPUMatrix.cpp:291-295): sqrt applied twice.
__device__ __forceinline__ double key_of(double d2) { return sqrt(sqrt(d2)); }

// Running bound that still admits every target whose key can tie with a best of squared distance d: the fourth root
// merges fewer than 16 neighbouring doubles, so anything more than 2^-48 (relative) above d has a strictly larger key.
__device__ __forceinline__ double tie_band(double d) { return __builtin_fma(d, 1.0 + 0x1p-48, 1e-300); }

}  // namespace icpk
