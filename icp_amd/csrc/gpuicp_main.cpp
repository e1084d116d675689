// GPUICP -- command-line drop-in for the reference's GPU executable (reference src/gpu/main.cu:58-81):
//   ./GPUICP file1 file2 nbiters method={-loop|-batch|-shared|-shared-loop}
//   ./GPUICP file1 file2 nbiters -batch count      (additive: file1/file2 hold `count` concatenated clouds of equal
//                                                   size, registered independently -- BASELINE config 5)
//   ./GPUICP file1 file2 nbiters method --bench    (additive: the same run through the persistent engine with the loop
//                                                   timed on the device; one JSON line on stderr -- iterations/s, pairs/s,
//                                                   the search kernel's time and roofline fraction, GPU and host core counts;
//                                                   stdout and point_cloud_result are unchanged)
// Same arguments, same stdout line, same output file; built on the C ABI of libicp_amd.so only.
// Argument / shape errors throw an uncaught std::runtime_error like the reference's runtime_assert
// (include/error.hpp:6-15); device errors print and exit(EXIT_FAILURE) like cudaCheckError
// (include/error.cuh:3-9).
#include <cstdio>
#include <cstdlib>
#include <future>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/icp_amd.h"

static void runtime_assert(bool cond, const std::string &msg)
{
    if (!cond) throw std::runtime_error(msg);
}

static std::vector<double> load(const char *path, size_t &rows, size_t &cols)
{
    double *a = nullptr;
    const int rc = icp_amd_read_cloud(path, &a, &rows, &cols, nullptr, 0);   // CSV like readCSV; *.npy / *.f64 by name
    runtime_assert(rc == ICP_AMD_OK, icp_amd_last_error());
    std::vector<double> v(a, a + rows * cols);
    icp_amd_free(a);
    return v;
}

// --bench: the registration through the persistent engine, the iteration loop bracketed by device events
static int run_bench(const std::vector<double> &P, size_t Pl, const std::vector<double> &Q, size_t Ql, unsigned nbiters, double *P_res)
{
    icp_amd_options opt;
    icp_amd_default_options(&opt);
    opt.kernel_timing = 1;
    icp_amd_engine *e = nullptr;
    int rc = icp_amd_create(&e, &opt);
    double loop_ms = 0.0, k_avg = 0.0, k_tot = 0.0;
    unsigned launches = 0;
    int ppl = 0, splits = 0, wgs = 0, variant = 0;
    if (rc == ICP_AMD_OK) rc = icp_amd_set_target(e, Q.data(), Ql);
    if (rc == ICP_AMD_OK) rc = icp_amd_set_source(e, P.data(), Pl);
    if (rc == ICP_AMD_OK) rc = icp_amd_iterate_timed(e, nbiters, &loop_ms);
    if (rc == ICP_AMD_OK) rc = icp_amd_kernel_ms(e, &k_avg, &k_tot, &launches);
    if (rc == ICP_AMD_OK) rc = icp_amd_search_geometry(e, &ppl, &splits, &wgs, &variant);
    if (rc == ICP_AMD_OK) rc = icp_amd_get_source(e, P_res);
    icp_amd_destroy(e);
    if (rc != ICP_AMD_OK) return rc;
    const double pairs = (double)Pl * (double)Ql, secs = loop_ms * 1e-3;
    const char *vname = variant == ICP_AMD_SEARCH_GRID ? "grid" : variant == ICP_AMD_SEARCH_F32 ? "f32" : variant == ICP_AMD_SEARCH_FMA ? "fma" : "exact";
    // roofline of the search kernel: the grid search streams the row records (HBM, ~8 TB/s, 100 bytes per row once settled);
    // the brute-force variants evaluate every pair (8 un-fused fp64 flops each: 39.3 TFLOP/s fp64 VALU, or the 2.5 PFLOP/s
    // f16 matrix pipe the default filter runs on)
    double frac = 0.0;
    const char *bound = "hbm";
    if (k_avg > 0.0) {
        if (variant == ICP_AMD_SEARCH_GRID) frac = 100.0 * (double)Pl / (k_avg * 1e-3) / 8e12;
        else if (variant == ICP_AMD_SEARCH_F32) { frac = 8.0 * pairs / (k_avg * 1e-3) / 2.5e15; bound = "mfma_f16"; }
        else { frac = 8.0 * pairs / (k_avg * 1e-3) / 39.3e12; bound = "valu_fp64"; }
    }
    fprintf(stderr,
            "{\"n\": %zu, \"m\": %zu, \"iters\": %u, \"loop_ms\": %.6f, \"iters_per_sec\": %.3f, \"pairs_per_sec\": %.6e, "
            "\"search\": {\"variant\": \"%s\", \"avg_launch_ms\": %.6f, \"launches\": %u, \"workgroups\": %d, \"q_splits\": %d}, "
            "\"roofline\": {\"bound\": \"%s\", \"frac\": %.4f}, \"gpus\": 1, \"cpu_cores\": %u, "
            "\"note\": \"pairs_per_sec = n x m x iterations / loop time: the pairs an exhaustive scan evaluates (the grid search returns the same "
            "correspondences from far fewer evaluations)\"}\n",
            Pl, Ql, nbiters, loop_ms, secs > 0 ? nbiters / secs : 0.0, secs > 0 ? pairs * nbiters / secs : 0.0, vname, k_avg, launches, wgs,
            splits, bound, frac, std::thread::hardware_concurrency());
    return ICP_AMD_OK;
}

static int run(int argc, char **argv, std::future<int> &warm)
{
    size_t Pl, Pc, Ql, Qc;
    // the two files are independent: parse them at the same time (errors surface in file order, like the reference)
    std::future<std::vector<double>> Qf = std::async(std::launch::async, [&] { return load(argv[2], Ql, Qc); });
    std::vector<double> P;
    try {
        P = load(argv[1], Pl, Pc);
    } catch (...) {
        Qf.wait();
        throw;
    }
    std::vector<double> Q = Qf.get();
    const unsigned nbiters = (unsigned)std::stoi(argv[3]);
    const std::string method{argv[4]};
    runtime_assert(method == "-batch" || method == "-loop" || method == "-shared" || method == "-shared-loop",
                   "Invalid method, must be: -shared|-loop|-shared-loop|-batch");
    const int m = method == "-batch" ? ICP_AMD_METHOD_BATCH
                  : method == "-loop" ? ICP_AMD_METHOD_LOOP
                  : method == "-shared" ? ICP_AMD_METHOD_SHARED : ICP_AMD_METHOD_SHARED_LOOP;
    // the reference hard-wires 3 columns in every kernel; other shapes end in a broadcasting error
    runtime_assert(Pc == 3 && Qc == 3, "element_wise_reduce requires arrays to be broadcastable");
    runtime_assert(Pl > 0 && Ql > 0, "empty point cloud");
    std::vector<double> P_res(Pl * 3);
    (void)warm.get();
    int rc;
    if (argc == 6 && std::string(argv[5]) == "--bench") {
        rc = run_bench(P, Pl, Q, Ql, nbiters, P_res.data());
    } else if (argc == 6) {
        const size_t count = (size_t)std::stoul(argv[5]);
        runtime_assert(count > 0 && Pl % count == 0 && Ql % count == 0, "batch count must divide the row counts of both files");
        rc = icp_amd_run_batch(P.data(), Q.data(), count, Pl / count, Ql / count, nbiters, P_res.data(), nullptr, nullptr, nullptr, nullptr);
    } else if (const char *g = std::getenv("ICP_AMD_GPUS"); g && std::atoi(g) > 1) {
        rc = icp_amd_run_multi(P.data(), Pl, Q.data(), Ql, nbiters, std::atoi(g), nullptr, P_res.data(), nullptr);   // additive
    } else {
        rc = icp_amd_run(P.data(), Pl, Q.data(), Ql, nbiters, m, P_res.data(), nullptr);
    }
    if (rc == ICP_AMD_EINVAL) throw std::runtime_error(icp_amd_last_error());
    if (rc != ICP_AMD_OK) {
        printf("HIP failure: %s\n", icp_amd_last_error());
        exit(EXIT_FAILURE);
    }
    // Q.euclidianDistance(P_res) needs equal shapes in the reference (it throws otherwise)
    runtime_assert(Pl == Ql, "element_wise_reduce requires arrays to be broadcastable");
    std::cout << "Squared actual mean diff: " << icp_amd_euclidian_distance(Q.data(), P_res.data(), Ql, 3) << std::endl;
    runtime_assert(icp_amd_write_cloud("point_cloud_result", P_res.data(), Pl, 3) == ICP_AMD_OK, "could not write point_cloud_result");
    if (const char *bin = getenv("ICP_AMD_RESULT_FILE"))   // additive: the result in full precision (*.npy / *.f64)
        runtime_assert(icp_amd_write_cloud_binary(bin, P_res.data(), Pl, 3) == ICP_AMD_OK, icp_amd_last_error());
    return 0;
}

int main(int argc, char **argv)
{
    runtime_assert(argc == 5 || (argc == 6 && (std::string(argv[4]) == "-batch" || std::string(argv[5]) == "--bench")),
                   "Usage: ./GPUICP file1 file2 nbiters method={-loop|-batch|-shared|-shared-loop}");
    // the HIP context and the code objects (0.15 s in a fresh process) come up while the input files are parsed; a failure
    // here is reported by the run itself
    std::future<int> warm = std::async(std::launch::async, [] { return icp_amd_init(-1); });
    try {
        return run(argc, argv, warm);
    } catch (...) {
        // an argument / shape / parse error ends the process like the reference's uncaught runtime_error -- but never while
        // the warm-up thread is still inside HIP context creation or a code-object load
        if (warm.valid()) warm.wait();
        throw;
    }
}
