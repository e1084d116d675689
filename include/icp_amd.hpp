// include/icp_amd.hpp -- header-only C++ face of the C ABI, shaped like the reference's own entry points
// (reference include/gpu/icp.cuh:38-39):
//
//     CPUMatrix icp_gpu_optimized(CPUMatrix &P, CPUMatrix &Q, unsigned iterations, std::string &method);
//     CPUMatrix icp_gpu(CPUMatrix &P, CPUMatrix &Q, unsigned iterations);
//
// The functions are templates over the matrix type so that this header does not depend on the reference's
// headers: any type with getArray(), getDim0(), getDim1() and a (double*, size_t, size_t) constructor that takes
// ownership of a malloc'd row-major buffer works -- the reference's CPUMatrix (include/libalg/CPUMatrix.hpp:9-93)
// does.  Ownership follows the reference: inputs borrowed and not modified, result owns a malloc'd n x 3 buffer
// (src/gpu/icp.cu:630,651).  Errors follow the reference too: argument errors throw std::runtime_error
// (include/error.hpp), device errors print and exit(EXIT_FAILURE) (include/error.cuh:3-9).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "icp_amd.h"

namespace icp_amd {

inline int method_code(const std::string &method)
{
    if (method == "-shared-loop") return ICP_AMD_METHOD_SHARED_LOOP;
    if (method == "-shared") return ICP_AMD_METHOD_SHARED;
    if (method == "-loop") return ICP_AMD_METHOD_LOOP;
    if (method == "-batch") return ICP_AMD_METHOD_BATCH;
    throw std::runtime_error("Invalid method, must be: -shared|-loop|-shared-loop|-batch");
}

template <class Matrix>
Matrix icp_gpu_optimized(Matrix &P, Matrix &Q, unsigned iterations, const std::string &method)
{
    const int m = method_code(method);
    if (P.getDim1() != 3 || Q.getDim1() != 3) throw std::runtime_error("element_wise_reduce requires arrays to be broadcastable");
    const size_t n = P.getDim0();
    double *out = static_cast<double *>(std::malloc((n ? n : 1) * 3 * sizeof(double)));
    if (!out) throw std::runtime_error("Alloc error !");
    const int rc = icp_amd_run(P.getArray(), n, Q.getArray(), Q.getDim0(), iterations, m, out, nullptr);
    if (rc == ICP_AMD_EINVAL) {
        std::free(out);
        throw std::runtime_error(icp_amd_last_error());
    }
    if (rc != ICP_AMD_OK) {
        std::printf("HIP failure: %s\n", icp_amd_last_error());
        std::exit(EXIT_FAILURE);
    }
    return Matrix(out, n, 3);
}

template <class Matrix>
Matrix icp_gpu(Matrix &P, Matrix &Q, unsigned iterations)
{
    return icp_gpu_optimized(P, Q, iterations, std::string("-batch"));
}

}  // namespace icp_amd
