// oracle/ref_trace.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A tracing driver around the *unmodified reference CPU sources* (compiled where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/; nothing from the reference is copied here).
// It re-walks the body of the reference's icp() (reference src/cpu/icp.cpp:194-229) by calling the
// reference's own functions in the same order, and dumps every intermediate the public icp() hides
// (per-iteration centroid, correspondence indices, covariance, SVD factors, R, t) so that
// tests/golden/make_golden.py can turn them into committed golden vectors.
//
// The two statements of icp() whose results are unused (icp.cpp:213 norm_values and icp.cpp:215
// cross_var_other) are skipped: they only add an N==M requirement and change no result.
//
// Usage: ref_trace P.csv Q.csv iters out.bin [store_p_each_iter=0|1]
//        ref_trace --corr Pz.csv Qz.csv out.bin      (get_correspondence_indices only, no centring;
//                                                     out.bin = idx[N] as i32)
//
// Trace layout (little endian): "ICPTRACE" u64 N, M, iters, flags | Qc[3] |
//   per iteration { [P_k[N*3] if flags&1] Pc[3] idx[N] (i32) cov[9] S[3] U[9] VT[9] R[9] t[3] } | P_out[N*3]
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>
#include <tuple>

#include "libCSV/csv.hpp"
#include "libalg/CPUMatrix.hpp"
#include "libalg/CPUView.hpp"
#include "cpu/icp.hpp"
#include "error.hpp"

static void put(FILE *f, const void *p, size_t n) { if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(2); } }
static void putm(FILE *f, const CPUMatrix &m) { put(f, m.getArray(), m.getDim0() * m.getDim1() * sizeof(double)); }

int main(int argc, char **argv)
{
    if (argc == 5 && std::string(argv[1]) == "--corr")
    {
        std::string hh;
        size_t pl, pc, ql, qc;
        double *pt = readCSV(argv[2], hh, pl, pc);
        CPUMatrix Pz(pt, pl, pc);
        double *qt = readCSV(argv[3], hh, ql, qc);
        CPUMatrix Qz(qt, ql, qc);
        auto corresps = get_correspondence_indices(Pz, Qz); // reference src/cpu/icp.cpp:72-93
        std::vector<int32_t> id(pl);
        for (size_t i = 0; i < pl; ++i) id[i] = std::get<1>(corresps[i]);
        FILE *fo = fopen(argv[4], "wb");
        if (!fo) { perror("fopen"); return 2; }
        put(fo, id.data(), pl * sizeof(int32_t));
        fclose(fo);
        return 0;
    }
    if (argc < 5) { fprintf(stderr, "usage: %s P.csv Q.csv iters out.bin [store_p]\n", argv[0]); return 2; }
    unsigned iters = (unsigned)std::stoi(argv[3]);
    bool store_p = argc > 5 && std::stoi(argv[5]) != 0;
    std::string h;
    size_t Pl, Pc_, Ql, Qc_;
    double *Pt = readCSV(argv[1], h, Pl, Pc_);            // reference main.cpp:25
    CPUMatrix P(Pt, Pl, Pc_);
    double *Qt = readCSV(argv[2], h, Ql, Qc_);            // reference main.cpp:28
    CPUMatrix Q(Qt, Ql, Qc_);
    FILE *f = fopen(argv[4], "wb");
    if (!f) { perror("fopen"); return 2; }
    uint64_t hdr[4] = {Pl, Ql, iters, store_p ? 1u : 0u};
    put(f, "ICPTRACE", 8);
    put(f, hdr, sizeof hdr);

    auto Q_center = Q.mean(0);                            // icp.cpp:197
    Q -= Q_center;                                        // icp.cpp:198
    putm(f, Q_center);
    CPUMatrix P_copy;
    P_copy = P;                                           // icp.cpp:203
    std::vector<int32_t> idx(Pl);
    for (unsigned it = 0; it < iters; ++it)
    {
        if (store_p) putm(f, P_copy);
        auto P_center = P_copy.mean(0);                   // icp.cpp:206
        P = P_copy - P_center;                            // icp.cpp:208
        auto corresps = get_correspondence_indices(P, Q); // icp.cpp:210
        auto cross_var = compute_cross_variance(P, Q, corresps, default_kernel); // icp.cpp:214
        CPUMatrix cov;
        cov = std::get<0>(cross_var);                     // keep a copy: dgesvd overwrites its input
        auto [U, S, V_T] = std::get<0>(cross_var).svd();  // icp.cpp:218
        auto R = U.dot(V_T);                              // icp.cpp:221
        auto t = Q_center - P_center.dot(R.transpose());  // icp.cpp:223
        P_copy = P_copy.dot(R.transpose()) + t;           // icp.cpp:225
        for (size_t i = 0; i < Pl; ++i) idx[i] = std::get<1>(corresps[i]);
        putm(f, P_center);
        put(f, idx.data(), Pl * sizeof(int32_t));
        putm(f, cov); putm(f, S); putm(f, U); putm(f, V_T); putm(f, R); putm(f, t);
    }
    putm(f, P_copy);
    fclose(f);
    return 0;
}
