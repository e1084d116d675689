// tools/ubench_bank.hip -- does the VGPR bank (register index mod 4) of the three operands of v_fmac_f32 / v_fma_f32 /
// v_pk_fma_f32 change the issue rate?  Explicit registers via asm clobbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 0.5\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n"
                 "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"
                 "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if (OP == 0)  // all three operands in distinct banks: acc bank0/1/2/3 rotating, src banks differ
            asm volatile("v_fmac_f32 v48, v41, v46\n v_fmac_f32 v49, v42, v47\n v_fmac_f32 v50, v43, v44\n v_fmac_f32 v51, v40, v45\n"
                         "v_fmac_f32 v52, v41, v46\n v_fmac_f32 v53, v42, v47\n v_fmac_f32 v54, v43, v44\n v_fmac_f32 v55, v40, v45\n" ::: CLOB);
        if (OP == 1)  // all three operands in the SAME bank
            asm volatile("v_fmac_f32 v48, v40, v44\n v_fmac_f32 v49, v41, v45\n v_fmac_f32 v50, v42, v46\n v_fmac_f32 v51, v43, v47\n"
                         "v_fmac_f32 v52, v40, v44\n v_fmac_f32 v53, v41, v45\n v_fmac_f32 v54, v42, v46\n v_fmac_f32 v55, v43, v47\n" ::: CLOB);
        if (OP == 2)  // two sources same bank, acc different
            asm volatile("v_fmac_f32 v49, v40, v44\n v_fmac_f32 v50, v41, v45\n v_fmac_f32 v51, v42, v46\n v_fmac_f32 v48, v43, v47\n"
                         "v_fmac_f32 v53, v40, v44\n v_fmac_f32 v54, v41, v45\n v_fmac_f32 v55, v42, v46\n v_fmac_f32 v52, v43, v47\n" ::: CLOB);
        if (OP == 3)  // pk_fma, operand pairs in distinct bank pairs: dst/acc v[48:49] (banks 0,1), src0 v[42:43] (2,3), src1 v[44:45](0,1) -> conflict on acc/src1
            asm volatile("v_pk_fma_f32 v[48:49], v[42:43], v[44:45], v[48:49]\n v_pk_fma_f32 v[50:51], v[40:41], v[46:47], v[50:51]\n v_pk_fma_f32 v[52:53], v[42:43], v[44:45], v[52:53]\n v_pk_fma_f32 v[54:55], v[40:41], v[46:47], v[54:55]\n"
                         "v_pk_fma_f32 v[56:57], v[42:43], v[44:45], v[56:57]\n v_pk_fma_f32 v[58:59], v[40:41], v[46:47], v[58:59]\n v_pk_fma_f32 v[60:61], v[42:43], v[44:45], v[60:61]\n v_pk_fma_f32 v[62:63], v[40:41], v[46:47], v[62:63]\n" ::: CLOB);
        if (OP == 4)  // v_add_f32 2 sources distinct banks
            asm volatile("v_add_f32 v48, v41, v46\n v_add_f32 v49, v42, v47\n v_add_f32 v50, v43, v44\n v_add_f32 v51, v40, v45\n"
                         "v_add_f32 v52, v41, v46\n v_add_f32 v53, v42, v47\n v_add_f32 v54, v43, v44\n v_add_f32 v55, v40, v45\n" ::: CLOB);
        if (OP == 5)  // v_fma_f32 VOP3, distinct banks
            asm volatile("v_fma_f32 v48, v41, v46, v48\n v_fma_f32 v49, v42, v47, v49\n v_fma_f32 v50, v43, v44, v50\n v_fma_f32 v51, v40, v45, v51\n"
                         "v_fma_f32 v52, v41, v46, v52\n v_fma_f32 v53, v42, v47, v53\n v_fma_f32 v54, v43, v44, v54\n v_fma_f32 v55, v40, v45, v55\n" ::: CLOB);
        if (OP == 6)  // v_mul_f32 (2 sources, no accumulate)
            asm volatile("v_mul_f32 v48, v41, v46\n v_mul_f32 v49, v42, v47\n v_mul_f32 v50, v43, v44\n v_mul_f32 v51, v40, v45\n"
                         "v_mul_f32 v52, v41, v46\n v_mul_f32 v53, v42, v47\n v_mul_f32 v54, v43, v44\n v_mul_f32 v55, v40, v45\n" ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v48, v49\n v_add_f32 %0, %0, v50\n v_add_f32 %0, %0, v56" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, int wps, int cus, int iters)
{
    const int blocks = cus * wps;
    float *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<OP><<<blocks, 256>>>(out, iters / 10); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); k<OP><<<blocks, 256>>>(out, iters); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s waves/SIMD=%d  wall %.2f ms  inst/s/SIMD %.3e\n", name, wps, ms, (double)iters * 8 * wps / (ms * 1e-3));
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 300000;
    for (int w : {2, 4}) {
        run<0>("fmac distinct banks", w, cus, iters); run<1>("fmac same bank x3", w, cus, iters); run<2>("fmac srcs same bank", w, cus, iters);
        run<3>("pk_fma", w, cus, iters); run<4>("add distinct banks", w, cus, iters); run<5>("fma VOP3 distinct banks", w, cus, iters);
        run<6>("mul distinct banks", w, cus, iters);
    }
    return 0;
}
