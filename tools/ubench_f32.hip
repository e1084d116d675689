// tools/ubench_f32.hip -- fp32 VALU instruction-rate microbenchmark (v_fma_f32 / v_pk_fma_f32 / v_min3_f32 mixes), long enough to
// reach the clock the chip sustains.  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f32 tools/ubench_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, unsigned long long *rt, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 0.5f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 q = {b, c}, r = {c, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0)  // fmac form: acc = b * c + acc   (2 source registers + accumulator)
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (OP == 1)  // 3 distinct sources (VOP3)
            asm volatile("v_fma_f32 %0, %1, %8, %9\n v_fma_f32 %1, %2, %8, %9\n v_fma_f32 %2, %3, %8, %9\n v_fma_f32 %3, %4, %8, %9\n v_fma_f32 %4, %5, %8, %9\n v_fma_f32 %5, %6, %8, %9\n v_fma_f32 %6, %7, %8, %9\n v_fma_f32 %7, %0, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (OP == 2)  // packed fma, accumulate in place
            asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q), "v"(r));
        if (OP == 3)  // min3
            asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (OP == 4)  // the search kernel's mix: 6 pk_fma + 2 min3 (two slots, two targets)
            asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %0, %9, %8, %0\n v_pk_fma_f32 %1, %9, %8, %1\n v_pk_fma_f32 %0, %8, %8, %0\n v_pk_fma_f32 %1, %9, %9, %1\n v_min3_f32 %2, %2, %3, %4\n v_min3_f32 %5, %5, %6, %7"
                         : "+v"(p0), "+v"(p1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q), "v"(r));
        if (OP == 5)  // v_fma_f32 with an SGPR operand
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed), "v"(c));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int OP>
void run(const char *name, int wps, int cus, int iters)
{
    const int blocks = cus * wps;
    float *out; unsigned long long *cyc, *rt;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CHECK(hipMalloc(&cyc, blocks * 8)); CHECK(hipMalloc(&rt, blocks * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<OP><<<blocks, 256>>>(out, cyc, rt, 1.0f, iters / 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<OP><<<blocks, 256>>>(out, cyc, rt, 1.0f, iters);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> hc(blocks), hr(blocks);
    CHECK(hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    const double insts = (double)iters * 8 * wps;      // wave-instructions per SIMD
    const double clk = (double)hc[blocks / 2] / hr[blocks / 2] * 0.1;
    printf("%-12s waves/SIMD=%d  wall %.2f ms  clock %.3f GHz  cycles/inst (wall*clock) %.3f  inst/s/SIMD %.3e\n", name, wps, ms, clk,
           ms * 1e-3 * clk * 1e9 / insts, insts / (ms * 1e-3));
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 400000;
    for (int w : {2, 4, 8}) {
        run<0>("fmac_f32", w, cus, iters); run<1>("fma_f32_3src", w, cus, iters); run<2>("pk_fma_f32", w, cus, iters);
        run<3>("min3_f32", w, cus, iters); run<4>("search_mix", w, cus, iters); run<5>("fma_f32_sgpr", w, cus, iters);
    }
    return 0;
}
