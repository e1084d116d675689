// tools/ubench_mfma.hip -- what the filter loop of search_mfma16_kernel can reach: v_mfma_f32_32x32x16_f16 (operands from
// LDS / registers, random data) with the per-pair reduction work beside it, in the forms tried for the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma tools/ubench_mfma.hip && tools/ubench_mfma
// Every mode runs the loop shape of the kernel: per iteration 2 target blocks (a0, a1 from LDS) x G source groups = 2G MFMAs.
// Reported: cycles per MFMA per SIMD = wall time x in-kernel clock / (MFMAs issued per SIMD), and the clock the chip held.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((aligned(16))) Half8 { _Float16 v[8]; };

__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

enum { CHAIN8 = 0, CHAIN4 = 1, CHAIN1 = 2, MIN2x16 = 3, TREE8 = 4, MFMA_ONLY = 5, VALU_ONLY = 6, OR3 = 7, CHAIN8_CHECK = 8, CHAIN8_CHECK128 = 9, MAX3ABS = 10 };

template <int MODE>
__device__ __forceinline__ float reduce(float v, const f32x16 &acc)
{
    if (MODE == CHAIN8 || MODE == VALU_ONLY || MODE == CHAIN8_CHECK || MODE == CHAIN8_CHECK128) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) v = min3f(v, acc[r], acc[r + 1]);
    } else if (MODE == CHAIN4) {
#pragma unroll
        for (int r = 0; r < 8; r += 2) v = min3f(v, acc[r], acc[r + 1]);
    } else if (MODE == CHAIN1) {
        v = min3f(v, acc[0], acc[1]);
    } else if (MODE == MIN2x16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v = __builtin_fminf(v, acc[r]);
            asm volatile("" : "+v"(v));   // keeps the compiler from fusing two v_min_f32 into one v_min3_f32
        }
    } else if (MODE == TREE8) {
        const float t0 = min3f(acc[0], acc[1], acc[2]), t1 = min3f(acc[3], acc[4], acc[5]), t2 = min3f(acc[6], acc[7], acc[8]);
        const float t3 = min3f(acc[9], acc[10], acc[11]), t4 = min3f(acc[12], acc[13], acc[14]);
        const float u0 = min3f(t0, t1, t2), u1 = min3f(t3, t4, acc[15]);
        v = min3f(v, u0, u1);
    } else if (MODE == OR3) {
        unsigned u = __float_as_uint(v);
#pragma unroll
        for (int r = 0; r < 16; r += 2) u = u | __float_as_uint(acc[r]) | __float_as_uint(acc[r + 1]);
        v = __uint_as_float(u);
    }
    return v;
}

template <int MODE, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k(float *out, unsigned long long *cyc, unsigned long long *rt, const Half8 *__restrict__ ain,
                                         const Half8 *__restrict__ bin, int iters)
{
    __shared__ Half8 st[2][1024];
    const unsigned tid = threadIdx.x, lane = tid & 63u, col = lane & 31u, half = lane >> 5;
    for (int kk = tid; kk < 1024; kk += 256) {
        st[0][kk] = ain[kk];
        st[1][kk] = ain[1024 + kk];
    }
    __syncthreads();
    const Half8 *sa = st[half];
    f16x8 b[G];
    float ms[G], thr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        b[g] = *(const f16x8 *)&bin[(blockIdx.x * 256 + tid + g * 64) & 4095];
        thr[g] = -1e30f - (float)g;
        ms[g] = 1e30f;
    }
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = zero;
    unsigned events = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 a_next = *(const f16x8 *)&sa[col];
    for (int i = 0; i < iters; ++i) {
        const int j0 = (i * 64) & 1023;
        const f16x8 a0 = a_next;
        const f16x8 a1 = *(const f16x8 *)&sa[j0 + 32 + col];
        a_next = *(const f16x8 *)&sa[((j0 + 64) & 1023) + col];
        if (MODE == MFMA_ONLY) {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[g], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[g], acc[g], 0, 0, 0);
        } else if (MODE == VALU_ONLY) {
            // the same reduction work on registers nobody produces in the loop (made opaque so that it is not hoisted)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                asm volatile("" : "+v"(acc[g]));
                ms[g] = reduce<MODE>(ms[g], acc[g]);
                asm volatile("" : "+v"(acc[g]));
                ms[g] = reduce<MODE>(ms[g], acc[g]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[g], zero, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                ms[g] = reduce<MODE>(ms[g], acc[g]);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[g], zero, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) ms[g] = reduce<MODE>(ms[g], acc[g]);
            if (MODE == CHAIN8_CHECK || (MODE == CHAIN8_CHECK128 && (i & 1))) {
                bool ev = false;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if ((unsigned)i == (unsigned)(tid + g)) ms[g] = thr[g];   // the seed-block test of the kernel
                    ev |= ms[g] < thr[g];
                }
                if (ev) {   // never taken (thr is far below every value); the kernel's slow path would sit here
                    ++events;
#pragma unroll
                    for (int g = 0; g < G; ++g) ms[g] = thr[g];
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        cyc[blockIdx.x] = t1 - t0;
        rt[blockIdx.x] = r1 - r0;
    }
    float s = (float)events;
#pragma unroll
    for (int g = 0; g < G; ++g) s += ms[g] + acc[g][0] + acc[g][7] + acc[g][15];
    out[blockIdx.x * 256 + tid] = s;
}

static std::vector<Half8> random_halves(size_t n, bool zeros)
{
    std::vector<Half8> v(n);
    for (size_t i = 0; i < n; ++i)
        for (int kk = 0; kk < 8; ++kk) v[i].v[kk] = zeros ? (_Float16)0.0f : (_Float16)((float)(rand() % 20001 - 10000) * 0.01f);
    return v;
}

// Operands shaped like the kernel's (round 3, the two leads of DESIGN section 9): target records {Txh,Txh,Txl,0,Tyh,Tyh,Tyl,0} /
// {Tzh,Tzh,Tzl,0,W0,W1,0,0} with T = -2 sigma q in hi + lo fp16 pieces and W = sigma^2 |q|^2 in two, source records
// {xh,xl,xh,0,yh,yl,yh,0} / {zh,zl,zh,0,1,1,0,0}; coordinates uniform in the scaled range (largest 2^7).
//   pattern 1: as the kernel builds them          pattern 2: the target's lo pieces (Txl, Tyl, Tzl, W1) zero
//   pattern 3: both clouds shifted so that every coordinate is positive (one sign for T, one for P)      pattern 4: 2 and 3
static int g_pattern = 0;
static void split16h(double v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)(float)v;
    lo = (_Float16)(float)(v - (double)(float)hi);
}
static double urand(double lo, double hi) { return lo + (hi - lo) * ((double)rand() / ((double)RAND_MAX + 1.0)); }
static std::vector<Half8> kernel_halves(size_t n, bool target, int pattern)
{
    // first half of the array = "half 0" records (x, y), second half = "half 1" records (z, W / ones), like rec16
    std::vector<Half8> v(n);
    const bool nolo = pattern == 2 || pattern == 4, onesign = pattern == 3 || pattern == 4;
    const _Float16 zero = (_Float16)0.0f, one = (_Float16)1.0f;
    for (size_t i = 0; i < n / 2; ++i) {
        const double lo = onesign ? 0.0 : -128.0;
        const double x = urand(lo, 128.0), y = urand(lo, 128.0), z = urand(lo, 128.0);
        _Float16 xh, xl, yh, yl, zh, zl;
        Half8 a, b;
        if (target) {
            split16h(-2.0 * x, xh, xl); split16h(-2.0 * y, yh, yl); split16h(-2.0 * z, zh, zl);
            if (nolo) xl = yl = zl = zero;
            const double w = (x * x + y * y) + z * z;
            _Float16 w0, w1;
            split16h(w, w0, w1);
            if (nolo) w1 = zero;
            a = Half8{{xh, xh, xl, zero, yh, yh, yl, zero}};
            b = Half8{{zh, zh, zl, zero, w0, w1, zero, zero}};
        } else {
            split16h(x, xh, xl); split16h(y, yh, yl); split16h(z, zh, zl);
            a = Half8{{xh, xl, xh, zero, yh, yl, yh, zero}};
            b = Half8{{zh, zl, zh, zero, one, one, zero, zero}};
        }
        v[i] = a;
        v[n / 2 + i] = b;
    }
    return v;
}

template <int MODE, int G>
void run(const char *name, int wps, int cus, int iters, bool zeros)
{
    const int blocks = cus * wps;
    float *out;
    unsigned long long *cyc, *rt;
    Half8 *a, *b;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&cyc, blocks * 8));
    CHECK(hipMalloc(&rt, blocks * 8));
    CHECK(hipMalloc(&a, 2048 * sizeof(Half8)));
    CHECK(hipMalloc(&b, 4096 * sizeof(Half8)));
    // (the kernel reads ain[0..1023] as half 0 and ain[1024..2047] as half 1; a lane's B operand is half 0 or half 1 by lane)
    const std::vector<Half8> ha = g_pattern ? kernel_halves(2048, true, g_pattern) : random_halves(2048, zeros);
    std::vector<Half8> hb = g_pattern ? kernel_halves(4096, false, g_pattern) : random_halves(4096, zeros);
    if (g_pattern) {   // lanes 0..31 of a wave take half-0 records, lanes 32..63 half-1 records (index = ... + tid + 64 g, mod 4096)
        const std::vector<Half8> src = hb;
        for (size_t i = 0; i < 4096; ++i) hb[i] = ((i & 63) < 32) ? src[i % 2048] : src[2048 + i % 2048];
    }
    CHECK(hipMemcpy(a, ha.data(), ha.size() * sizeof(Half8), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b, hb.data(), hb.size() * sizeof(Half8), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<MODE, G><<<blocks, 256>>>(out, cyc, rt, a, b, iters / 4);   // warm-up, also of the clock governor
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE, G><<<blocks, 256>>>(out, cyc, rt, a, b, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(blocks), hr(blocks);
    CHECK(hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end());
    std::sort(hr.begin(), hr.end());
    const double clk = (double)hc[blocks / 2] / (double)hr[blocks / 2] * 0.1;   // GHz: s_memrealtime ticks at 100 MHz
    const double slots = (double)iters * 2 * G * wps;                             // MFMA slots per SIMD
    if (g_pattern) printf("pattern %d: ", g_pattern);
    printf("%-16s G=%d waves/SIMD=%d %s wall %8.3f ms  clock %.3f GHz  cycles/MFMA-slot/SIMD %6.2f  (kernel cycles/iter/wave %7.1f)\n", name, G, wps,
           zeros ? "zeros " : "random", ms, clk, ms * 1e-3 * clk * 1e9 / slots, (double)hc[blocks / 2] / iters);
    fflush(stdout);
    CHECK(hipFree(out)); CHECK(hipFree(cyc)); CHECK(hipFree(rt)); CHECK(hipFree(a)); CHECK(hipFree(b));
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = argc > 1 ? atoi(argv[1]) : 200000;
    printf("device %s  CUs %d\n", p.name, cus);
    if (argc > 2) {   // operand patterns of the kernel: ubench_mfma <iters> patterns
        for (int rep = 0; rep < 2; ++rep)
            for (int pat : {0, 1, 2, 3, 4}) {
                g_pattern = pat;
                run<CHAIN8_CHECK, 4>("chain8+check64", 4, cus, iters, false);
                run<MFMA_ONLY, 4>("mfma_only", 4, cus, iters, false);
            }
        return 0;
    }
    for (int w : {4, 2, 1}) {
        run<MFMA_ONLY, 4>("mfma_only", w, cus, iters, false);
        run<CHAIN1, 4>("chain1", w, cus, iters, false);
        run<CHAIN4, 4>("chain4", w, cus, iters, false);
        run<CHAIN8, 4>("chain8", w, cus, iters, false);
        run<TREE8, 4>("tree8", w, cus, iters, false);
        run<MIN2x16, 4>("min2x16", w, cus, iters, false);
        run<OR3, 4>("or3x8", w, cus, iters, false);
        run<CHAIN8_CHECK, 4>("chain8+check64", w, cus, iters, false);
        run<CHAIN8_CHECK128, 4>("chain8+check128", w, cus, iters, false);
        run<VALU_ONLY, 4>("valu_only", w, cus, iters, false);
    }
    run<CHAIN8, 2>("chain8", 4, cus, iters, false);
    run<MFMA_ONLY, 4>("mfma_only", 4, cus, iters, true);
    run<CHAIN8, 4>("chain8", 4, cus, iters, true);
    return 0;
}
