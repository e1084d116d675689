// tools/ubench_valu.hip -- instruction-rate microbenchmark for the fp64 VALU ops of the search kernel.
// Measures cycles per wave-instruction per SIMD (s_memtime) and the clock the chip holds (s_memtime /
// s_memrealtime) for several ops and occupancies.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 20000;

#define OPS8(STR) \
    asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));

template <int OP>
__global__ void k(double *out, unsigned long long *cyc, unsigned long long *rt, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double c = seed * 0.5 + 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) OPS8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8")
        if (OP == 1) OPS8("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8")
        if (OP == 2) OPS8("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8")
        if (OP == 3) OPS8("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8")
        if (OP == 4) { asm volatile("v_cmp_lt_f64 vcc, %0, %8\n v_cmp_lt_f64 vcc, %1, %8\n v_cmp_lt_f64 vcc, %2, %8\n v_cmp_lt_f64 vcc, %3, %8\n v_cmp_lt_f64 vcc, %4, %8\n v_cmp_lt_f64 vcc, %5, %8\n v_cmp_lt_f64 vcc, %6, %8\n v_cmp_lt_f64 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc"); }
        if (OP == 5) { float f0 = (float)a0; asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(f0)); a0 = f0; }
        // the search kernel's own mix for one pair: 3 sub, 3 mul, 2 add, 1 min (dependent chain inside, 8 chains would need 8 copies; here 1 chain x8 regs reuse)
        if (OP == 6) OPS8("v_add_f64 %0, %0, -%8\n v_add_f64 %1, %1, -%8\n v_add_f64 %2, %2, -%8\n v_mul_f64 %3, %0, %0\n v_mul_f64 %4, %1, %1\n v_mul_f64 %5, %2, %2\n v_add_f64 %6, %3, %4\n v_add_f64 %7, %6, %5")
        if (OP == 7) { unsigned u0 = (unsigned)a0, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3; asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc"); a0 = u0 + u1 + u2 + u3; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char *name, int waves_per_simd, int cus)
{
    const int threads = 256;                       // 4 waves = 1 per SIMD
    const int blocks = cus * waves_per_simd;
    double *out; unsigned long long *cyc, *rt;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
    CHECK(hipMalloc(&cyc, blocks * 8)); CHECK(hipMalloc(&rt, blocks * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<OP><<<blocks, threads>>>(out, cyc, rt, 1.0);  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, threads>>>(out, cyc, rt, 1.0);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    std::vector<unsigned long long> hc(blocks), hr(blocks);
    CHECK(hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    double cmed = hc[blocks / 2], rmed = hr[blocks / 2];
    double insts = (double)ITERS * 8;               // per wave
    // waves_per_simd waves share one SIMD: SIMD issues waves_per_simd*insts in cmed cycles
    printf("%-10s waves/SIMD=%d  wall %.3f ms  cycles/wave-inst/SIMD %.2f  clock %.3f GHz  (wall-derived inst/s/SIMD %.3e)\n", name,
           waves_per_simd, ms, cmed / (insts * waves_per_simd), cmed / rmed * 0.1, insts * waves_per_simd / (ms * 1e-3));
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    for (int w : {1, 2, 4}) {
        run<0>("add_f64", w, cus); run<1>("mul_f64", w, cus); run<2>("fma_f64", w, cus); run<3>("min_f64", w, cus);
        run<4>("cmp_f64", w, cus); run<5>("add_f32", w, cus); run<6>("pairmix", w, cus); run<7>("cndmask", w, cus);
    }
    return 0;
}
