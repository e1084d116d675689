// tools/ubench_h2d.hip -- wall time of a host-to-device copy of a pageable buffer (what set_target / set_source receive) by
// size, against the same bytes staged through a pinned buffer by the caller.   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_h2d tools/ubench_h2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s;
    (void)hipStreamCreate(&s);
    const size_t maxb = 64u << 20;
    char *dev, *pin;
    (void)hipMalloc(&dev, maxb);
    (void)hipHostMalloc(&pin, maxb, hipHostMallocDefault);
    printf("%12s %12s %12s %12s %12s\n", "rows(x24B)", "bytes", "pageable ms", "staged ms", "d2h page ms");
    for (size_t rows : {2903ul, 10000ul, 20000ul, 30000ul, 40000ul, 48485ul, 60000ul, 80000ul, 100000ul, 200000ul, 300000ul, 1000000ul}) {
        const size_t b = rows * 24;
        char *h = (char *)malloc(b);
        memset(h, 1, b);
        double best = 1e9, best2 = 1e9, best3 = 1e9;
        for (int r = 0; r < 7; ++r) {
            double t0 = now();
            (void)hipMemcpyAsync(dev, h, b, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            double t1 = now();
            memcpy(pin, h, b);
            (void)hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            double t2 = now();
            (void)hipMemcpyAsync(h, dev, b, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            double t3 = now();
            if (t1 - t0 < best) best = t1 - t0;
            if (t2 - t1 < best2) best2 = t2 - t1;
            if (t3 - t2 < best3) best3 = t3 - t2;
        }
        printf("%12zu %12zu %12.3f %12.3f %12.3f\n", rows, b, best, best2, best3);
        free(h);
    }
    for (size_t b : {8ul, 4096ul, 1ul << 20, 8ul << 20}) {   // what a pinned (mapped) allocation costs, e.g. the grid scratch's two hint words
        double best = 1e9, bestf = 1e9;
        for (int r = 0; r < 5; ++r) {
            void *p = nullptr;
            double t0 = now();
            (void)hipHostMalloc(&p, b, hipHostMallocMapped);
            double t1 = now();
            (void)hipHostFree(p);
            double t2 = now();
            if (t1 - t0 < best) best = t1 - t0;
            if (t2 - t1 < bestf) bestf = t2 - t1;
        }
        printf("hipHostMalloc(%zu, mapped) %.3f ms, hipHostFree %.3f ms\n", b, best, bestf);
    }
    return 0;
}
