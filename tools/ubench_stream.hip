// tools/ubench_stream.hip -- what the memory system of the box delivers for the access shapes of the grid iteration's
// streaming kernels ("HBM bandwidth ... to be confirmed on the box with a copy kernel").  One figure per
// shape and size: bytes moved / hipEvent time of 50 back-to-back launches after 5 warm-up launches.
//   copy        : read 32 B, write 32 B per row (double4 -> double4), the textbook copy
//   apply-soa   : read 24 + 4, write 24 + 32 per row  (apply_kernel on the grid path before round 3: SoA in and out + record half)
//   apply-rec   : read 32, write 32 per row, in place    (apply_kernel on the grid path now: the record half only)
//   cov-rec     : read 64, write 4 per row               (cov_rec_kernel)
//   search      : read 64, write 32 per row              (search_grid_kernel once every row holds its certificate)
//   2r1w        : read 32 + 32, write 32 per row          (two read streams, one write stream)
//   search perm / p-split : the search's accesses with the rows taken in a scattered order (+ 4 B of the order itself): one
//                 64-byte record per row against two arrays of 32-byte halves
// NT = the writes as non-temporal stores (__builtin_nontemporal_store).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_stream tools/ubench_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kWG = 256, kPer = 4;   // four rows per lane, all loads in flight before the first use (as the kernels do)

template <bool NT> __device__ __forceinline__ void put(double4 *p, double4 v)
{
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

template <bool NT> __global__ __launch_bounds__(kWG) void copy_kernel(const double4 *__restrict__ a, double4 *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * kWG * kPer + threadIdx.x;
    double4 v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const size_t i = base + (size_t)k * kWG; v[k] = a[i < n ? i : n - 1]; }
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const size_t i = base + (size_t)k * kWG; if (i < n) put<NT>(&b[i], v[k]); }
}

template <bool NT> __global__ __launch_bounds__(kWG) void apply_rec_kernel(double4 *__restrict__ rec, size_t n)
{
    const size_t base = (size_t)blockIdx.x * kWG * kPer + threadIdx.x;
    double4 v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const size_t i = base + (size_t)k * kWG; v[k] = rec[2 * (i < n ? i : n - 1)]; }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) put<NT>(&rec[2 * i], make_double4(v[k].x + 1.0, v[k].y + 1.0, v[k].z + 1.0, v[k].w));
    }
}

template <bool NT> __global__ __launch_bounds__(kWG) void apply_soa_kernel(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                                                           const unsigned *__restrict__ idx, double4 *__restrict__ rec, size_t n)
{
    const size_t base = (size_t)blockIdx.x * kWG * kPer + threadIdx.x;
    double a[kPer], b[kPer], c[kPer];
    unsigned j[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const size_t i = base + (size_t)k * kWG, ic = i < n ? i : n - 1;
        a[k] = x[ic]; b[k] = y[ic]; c[k] = z[ic]; j[k] = idx[ic];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) {
            x[i] = a[k] + 1.0; y[i] = b[k] + 1.0; z[i] = c[k] + 1.0;
            put<NT>(&rec[2 * i], make_double4(a[k] + 1.0, b[k] + 1.0, c[k] + 1.0, __longlong_as_double((long long)j[k])));
        }
    }
}

__global__ __launch_bounds__(kWG) void cov_rec_like_kernel(const double4 *__restrict__ rec, unsigned *__restrict__ idx, double *__restrict__ sink, size_t n)
{
    const size_t base = (size_t)blockIdx.x * kWG * kPer + threadIdx.x;
    double4 u[kPer], v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const size_t i = base + (size_t)k * kWG, ic = i < n ? i : n - 1; u[k] = rec[2 * ic]; v[k] = rec[2 * ic + 1]; }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) { idx[i] = (unsigned)__double_as_longlong(v[k].w); acc += u[k].x * v[k].x + u[k].y * v[k].y + u[k].z * v[k].z; }
    }
    if (acc == 12345.678) sink[0] = acc;
}

template <bool NT> __global__ __launch_bounds__(kWG) void search_like_kernel(double4 *__restrict__ rec, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;   // one row per lane, like the search
    if (i >= n) return;
    const double4 u = rec[2 * i], v = rec[2 * i + 1];
    put<NT>(&rec[2 * i + 1], make_double4(v.x, v.y, v.z, u.w + v.w));
}

template <bool NT> __global__ __launch_bounds__(kWG) void two_read_kernel(const double4 *__restrict__ a, const double4 *__restrict__ b,
                                                                          double4 *__restrict__ c, size_t n)
{
    const size_t base = (size_t)blockIdx.x * kWG * kPer + threadIdx.x;
    double4 u[kPer], v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const size_t i = base + (size_t)k * kWG, ic = i < n ? i : n - 1; u[k] = a[ic]; v[k] = b[ic]; }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const size_t i = base + (size_t)k * kWG;
        if (i < n) put<NT>(&c[i], make_double4(u[k].x + v[k].x, u[k].y + v[k].y, u[k].z + v[k].z, u[k].w));
    }
}

// the same in a permuted row order (the spatial order of the search while the clouds still move): one row per lane
template <bool SPLIT> __global__ __launch_bounds__(kWG) void search_perm_kernel(double4 *__restrict__ rec, double4 *__restrict__ a,
                                                                                double4 *__restrict__ b, const unsigned *__restrict__ perm, size_t n)
{
    const size_t l = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (l >= n) return;
    const size_t i = perm[l];
    if (SPLIT) {
        const double4 u = a[i], v = b[i];
        b[i] = make_double4(v.x, v.y, v.z, u.w + v.w);
    } else {
        const double4 u = rec[2 * i], v = rec[2 * i + 1];
        rec[2 * i + 1] = make_double4(v.x, v.y, v.z, u.w + v.w);
    }
}
__global__ void make_perm_kernel(unsigned *perm, size_t n, unsigned cell)   // rows of a "cell" of `cell` consecutive slots come from all over the array
{
    const size_t l = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (l >= n) return;
    const size_t cells = (n + cell - 1) / cell;
    size_t i = (l % cell) * cells + l / cell;   // a transpose: consecutive l are `cells` rows apart
    perm[l] = (unsigned)(i < n ? i : l);
}

template <class F> static double time_ms(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 5; ++k) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, nullptr));
    for (int k = 0; k < 50; ++k) launch();
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / 50.0;
}

int main()
{
    const size_t sizes[] = {1000000, 4000000, 16000000};
    printf("%-12s %10s %12s %10s %10s\n", "shape", "rows", "bytes/row", "us", "TB/s");
    for (size_t n : sizes) {
        double4 *a, *b, *c, *rec;
        double *x, *y, *z, *sink;
        unsigned *idx;
        CHECK(hipMalloc(&a, n * 32)); CHECK(hipMalloc(&b, n * 32)); CHECK(hipMalloc(&c, n * 32)); CHECK(hipMalloc(&rec, n * 64));
        CHECK(hipMalloc(&x, n * 8)); CHECK(hipMalloc(&y, n * 8)); CHECK(hipMalloc(&z, n * 8)); CHECK(hipMalloc(&idx, n * 4));
        CHECK(hipMalloc(&sink, 8));
        CHECK(hipMemset(a, 0, n * 32)); CHECK(hipMemset(b, 0, n * 32)); CHECK(hipMemset(c, 0, n * 32)); CHECK(hipMemset(rec, 0, n * 64));
        CHECK(hipMemset(x, 0, n * 8)); CHECK(hipMemset(y, 0, n * 8)); CHECK(hipMemset(z, 0, n * 8)); CHECK(hipMemset(idx, 0, n * 4));
        const unsigned g4 = (unsigned)((n + kWG * kPer - 1) / (kWG * kPer)), g1 = (unsigned)((n + kWG - 1) / kWG);
        auto report = [&](const char *name, double bytes_per_row, double ms) {
            printf("%-12s %10zu %12.0f %10.2f %10.2f\n", name, n, bytes_per_row, ms * 1e3, bytes_per_row * (double)n / (ms * 1e-3) / 1e12);
        };
        report("copy", 64, time_ms([&] { copy_kernel<false><<<g4, kWG>>>(a, b, n); }));
        report("copy NT", 64, time_ms([&] { copy_kernel<true><<<g4, kWG>>>(a, b, n); }));
        report("apply-soa", 84, time_ms([&] { apply_soa_kernel<false><<<g4, kWG>>>(x, y, z, idx, rec, n); }));
        report("apply-rec", 64, time_ms([&] { apply_rec_kernel<false><<<g4, kWG>>>(rec, n); }));
        report("apply-rec NT", 64, time_ms([&] { apply_rec_kernel<true><<<g4, kWG>>>(rec, n); }));
        report("cov-rec", 68, time_ms([&] { cov_rec_like_kernel<<<g4, kWG>>>(rec, idx, sink, n); }));
        report("search", 96, time_ms([&] { search_like_kernel<false><<<g1, kWG>>>(rec, n); }));
        report("search NT", 96, time_ms([&] { search_like_kernel<true><<<g1, kWG>>>(rec, n); }));
        unsigned *perm;
        CHECK(hipMalloc(&perm, n * 4));
        make_perm_kernel<<<g1, kWG>>>(perm, n, 128);
        CHECK(hipDeviceSynchronize());
        report("search perm", 100, time_ms([&] { search_perm_kernel<false><<<g1, kWG>>>(rec, a, b, perm, n); }));
        report("search p-split", 100, time_ms([&] { search_perm_kernel<true><<<g1, kWG>>>(rec, a, b, perm, n); }));
        CHECK(hipFree(perm));
        report("2r1w", 96, time_ms([&] { two_read_kernel<false><<<g4, kWG>>>(a, b, c, n); }));
        report("2r1w NT", 96, time_ms([&] { two_read_kernel<true><<<g4, kWG>>>(a, b, c, n); }));
        CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(c)); CHECK(hipFree(rec)); CHECK(hipFree(x)); CHECK(hipFree(y)); CHECK(hipFree(z));
        CHECK(hipFree(idx)); CHECK(hipFree(sink));
    }
    return 0;
}
