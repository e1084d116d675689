// icp_amd/csrc/kernels.h -- internal: launchers of the gfx950 kernels (kernels.hip), used by engine.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <tuple>
#include <utility>

namespace icpk {

// Launches `kernel` and returns THAT launch's status (what hipLaunchKernel itself returns).  `k<<<...>>>(...)` followed by
// hipGetLastError() reports the host thread's sticky error word instead, which every HIP user of the process shares: a
// stale error left there by another library (torch, RCCL) would be taken for a failed launch of ours, and clearing the
// word before every call would swallow the host application's own errors.  This helper neither reads nor clears it.
template <typename... KArgs, size_t... I>
inline hipError_t launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s,
                              std::tuple<KArgs...> &vals, std::index_sequence<I...>)
{
    void *ptrs[] = {(void *)&std::get<I>(vals)..., nullptr};
    return hipLaunchKernel((const void *)kernel, grid, block, ptrs, lds, s);
}
template <typename... KArgs, typename... Args>
inline hipError_t launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args)
{
    static_assert(sizeof...(KArgs) == sizeof...(Args), "launch(): argument count differs from the kernel's");
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};
    return launch_impl(kernel, grid, block, lds, s, vals, std::index_sequence_for<KArgs...>{});
}

// Geometry shared by every kernel.
constexpr int kWG      = 256;    // threads per workgroup (4 wave64)
constexpr int kChunk   = 1024;   // source rows per reduction leaf / per cov-apply workgroup
constexpr int kTileQ   = 1024;   // target points per LDS tile; Q arrays are padded to a multiple with +inf
constexpr int kCovW    = 12;     // doubles per cov partial: 9 cov sums, sum d^2, 2 spare
constexpr int kCenW    = 3;      // doubles per centroid partial

enum SearchVariant { kExact = 1, kFma = 2, kF32 = 3, kGrid = 4, kExpanded = 5 /* batch mode only */, kSeeded = 6 /* batch mode: kExpanded, seeded from the second iteration on */ };
constexpr size_t kPadRows = 2048;  // clouds are padded to this many rows (covers every kernel's tile)
constexpr size_t kSplitGran = 128; // target splits are whole multiples of this many targets (>= every kernel's event-check block)

// Debugging / test switches (devmem.cpp).  The variables a user sets have names of their own -- ICP_AMD_DEVICE, ICP_AMD_SEARCH,
// ICP_AMD_GPUS, ICP_AMD_CACHE_MB, ICP_AMD_CSV_THREADS, ICP_AMD_RESULT_FILE, ICP_AMD_MULTI_EXCHANGE, ICP_AMD_BATCH_WORKERS; every
// other switch is an entry of ONE list, ICP_AMD_DEBUG="grid_cert=0,mfma=0,qsplits=5" (lower-case names, INTEGRATION.md has
// the table).  The older spelling of a switch as a variable of its own, ICP_AMD_<NAME>, is still honoured (the list wins).
// knob() returns the value's text or nullptr; none of the switches changes a result.
const char *knob(const char *name);
int knob_int(const char *name, int dflt);
double knob_double(const char *name, double dflt);

// Device memory through a small cache (devmem.cpp): same contracts as hipMalloc / hipFree, but released blocks are kept
// (ICP_AMD_CACHE_MB, default 2048) and reused by later requests of about the same size.
hipError_t dmalloc_bytes(void **p, size_t bytes);
template <class T> inline hipError_t dmalloc(T **p, size_t bytes) { return dmalloc_bytes((void **)p, bytes); }
hipError_t dfree(void *p);
// same, for a block that only work enqueued on stream `s` has touched: waits for that stream instead of the whole device,
// so engines on other streams keep running (a batch of registrations on several host threads)
hipError_t dfree_on(void *p, hipStream_t s);
// a pinned host buffer of kPinBytes (kept and reused like the device blocks); release only when no copy through it is in flight
constexpr size_t kPinBytes = (size_t)8 << 20;
hipError_t pin_acquire(void **p);
void pin_release(void *p);
void dcache_trim();      // give every kept block back to the driver
size_t dcache_held();    // bytes kept

struct Soa3 {            // device SoA view of a cloud
    double *x, *y, *z;
    // optional row-major copy for random access by index, rec[j] = {x, y, z, unused}: one 32-byte record (one cache
    // line touched) per gathered point instead of three
    double4 *rec = nullptr;
    // source clouds on the grid path: rec holds TWO arrays of 32-byte halves, A[i] = rec[i] = {x, y, z, unused} and
    // B[i] = rec[rec_stride + i] = {matched target's coordinates, idx | certificate} (grid.hip).  Two arrays rather than one
    // 64-byte record per row: the transform apply streams A alone and the settled search reads A, B and writes B as three
    // contiguous streams -- 11 and 14 us at 1M rows where the interleaved record measured 20 and 21 (tools/ubench_stream.hip).
    size_t rec_stride = 0;
};
// rec[j] = {x[j], y[j], z[j], 0} for j < rows
hipError_t build_rec(hipStream_t s, Soa3 soa, size_t rows);

// Sum of each coordinate over 1024-row chunks: partial[chunk][3] (fixed reduction order).
// src is row-major AoS (n x 3) when aos != nullptr, else the SoA view.
hipError_t colsum_partials(hipStream_t s, const double *aos, Soa3 soa, size_t n, double *partial);
// out[k] = (fixed-order sum over `count` partials: groups of 32 in index order, then the group sums in order)[k] / denom
hipError_t finalize_centroid(hipStream_t s, const double *partial, size_t count, double denom, double *out3);
// AoS -> SoA, padded rows [n, n_pad) set to `pad`.  If c3 != nullptr subtracts c3[k] (the centring of Q).
hipError_t aos_to_soa(hipStream_t s, const double *aos, size_t n, size_t n_pad, const double *c3, double pad, Soa3 out);
hipError_t soa_to_aos(hipStream_t s, Soa3 in, size_t n, double *aos);
// the same through a row permutation (a source stored in spatial order): out row k = aos row perm[k]; aos row perm[k] = in row k;
// dst[perm[k]] = src[k]
hipError_t aos_gather_to_soa(hipStream_t s, const double *aos, const uint32_t *perm, size_t n, size_t n_pad, double pad, Soa3 out);
hipError_t soa_scatter_to_aos(hipStream_t s, Soa3 in, const uint32_t *perm, size_t n, double *aos);
hipError_t scatter_u32(hipStream_t s, const uint32_t *src, const uint32_t *perm, size_t n, uint32_t *dst);

// Split geometry of both search kernels: targets [0, search_scan_rows(m)) in q_splits pieces of
// search_rows_per_split(m, q_splits) targets each (trailing pieces may be empty; their records read (inf, ~0) or the seed).
size_t search_scan_rows(size_t m);
size_t search_rows_per_split(size_t m, int q_splits);

// Brute-force nearest neighbour: for source row i (centred on the fly with pc) and target split y,
// (key, idx) of the lexicographically smallest (sqrt(sqrt(d^2)), j) over the split's target range.
// out_key / out_idx are [q_splits][n_pad].  n_pad multiple of 256*ppl, m_pad multiple of kTileQ.
// ppl in {1,2,4} source points per lane, cb in {8,16,32} targets between event checks.
// idx_prev: optional seed (previous iteration's correspondences, n_pad entries, >= m means "none").
hipError_t search(hipStream_t s, int variant, int ppl, int cb, Soa3 P, size_t n_pad, const double *pc,
                  Soa3 Qz, size_t m, size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key,
                  uint32_t *out_idx);

// fp32 pre-filter variant.  qrec: m_pad 16-byte records {-2qx,-2qy,-2qz,|q|^2} built by build_qrec, which also
// leaves the bit pattern of max |q|^2 in *qmax2_bits; qmax = sqrt of that.  n_pad and m_pad multiples of 2048.
// ppl in {2,4,8}, cb in {16,32}.
hipError_t build_qrec(hipStream_t s, Soa3 Qz, size_t m, size_t m_pad, void *qrec, unsigned long long *qmax2_bits);
hipError_t search_f32(hipStream_t s, int ppl, int cb, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, const void *qrec,
                      double qmax, size_t m, size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key,
                      uint32_t *out_idx);

// source rows per workgroup of the matrix-pipe filter: 128 * groups (groups = 32-row source groups per wave, 2 or 4)
int search_mfma_rows_per_workgroup(int groups);
// The filter on v_mfma_f32_32x32x16_f16 (search_mfma.hip): every operand in two fp16 pieces after a power-of-two scaling.
// rec16: 32 bytes per target (2 * m_pad 16-byte halves) built by build_qrec16 from the centred target and qmax.
hipError_t build_qrec16(hipStream_t s, Soa3 Qz, size_t m, size_t m_pad, double qmax, void *rec16);
hipError_t search_mfma16(hipStream_t s, int groups, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, const void *rec16, double qmax,
                         size_t m, size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key, uint32_t *out_idx);

// test hook: max over (source, target) pairs of |filter value - exact| / bound; targets in runs of 32 identical points
hipError_t mfma16_probe(hipStream_t s, Soa3 P, size_t n32, Soa3 Qz, size_t m32, size_t m_pad, double qmax, const void *rec16,
                        unsigned long long *worst_bits);

// Uniform grid over the centred target cloud (grid.hip): same correspondences as the brute-force kernels, found by
// evaluating only the targets in the cells a query's search box touches.
// one empty launch from grid.hip: loads that translation unit's code object (icp_amd_init)
hipError_t grid_warmup(hipStream_t s);
struct GridDesc {
    double lo[3], hi[3];   // bounding box of the targets
    double inv[3];         // cells per unit length along each axis (0 for an axis of zero extent)
    double h[3];           // cell edge along each axis (about 1 / inv; only ever used through a check by cell_of())
    int dim[3];            // cells per axis
    double r0;             // first search radius of an un-seeded query that lies inside the box (half a cell)
};
struct Grid {
    GridDesc d{};
    size_t cells = 0;
    uint32_t *start = nullptr;   // cells + 1: first sorted position of each cell, cell id = (cz*dim[1] + cy)*dim[0] + cx
    uint32_t *perm = nullptr;    // m: original index of each sorted target
    double4 *rec = nullptr;      // m: sorted targets as {x, y, z, original index (integer bit pattern)}
    bool ok = false;             // false: the cloud has non-finite coordinates, use a brute-force variant
    hipStream_t stream = nullptr;   // the stream the grid was built on and is searched on (grid_free waits for it)
    bool on_stream = false;
};
// Builds the grid of Qz[0, m) (allocates; synchronises the stream).  Non-finite coordinates: success with ok = false.
hipError_t grid_build(hipStream_t s, Soa3 Qz, size_t m, Grid *g);
void grid_free(Grid *g);
// out_key / out_idx: one (key, idx) record per source row i < n, laid out like one target split of `search`.
// Queue of the source points whose search box is too large for one lane (they are finished by a second kernel, one
// workgroup each).  count2: two counters used alternately (each launch zeroes the other one); host_mem: the last
// launch's queue length, written by the device into mapped host memory (a hint, read without synchronisation).
struct GridScratch {
    uint32_t *count2 = nullptr;
    uint32_t *list = nullptr;               // n entries
    volatile uint32_t *host_mem = nullptr;  // host address: [0] the last overflow queue length, [1] 1 = the last search found the
                                            // registration settled (hints for the host's next launch, read without synchronisation)
    uint32_t *host_count = nullptr;         // device address of the same two words
    int issue_cert = 1;                     // searches over row records leave motion certificates (the engine clears it with grid_cert=0)
    int fuse_cov = 1;                       // ICP_AMD_GRID_FUSE_COV=0: never the settled search that also forms the covariance sums
    int flight = 6;                         // candidate loads in flight per walking lane of search_grid_kernel (grid_flight=4|6|8)
    int parity = 0;
    int row_cap = 64, eval_cap = 768;       // rows of cells / targets one lane walks alone (read from the environment at allocation)
    int defer_always = 0;                   // 1: hand over every box above the caps (tests); 0: only when most of the wave would idle
    double order_switch = 0.125;            // motion (cell half-edges) below which rows are taken in storage order (ICP_AMD_GRID_ORDER_SWITCH)
    unsigned long long *stat = nullptr;     // ICP_AMD_GRID_STATS=1 (diagnostics): per launch {rows that walked cells, targets evaluated},
    unsigned stat_slot = 0;                 // printed to stderr when the scratch is released
    bool enabled = true;                    // false: this launch hands nothing over (every lane finishes its own point) and the
                                            // overflow kernel is not launched -- the caller's choice while the queue stays empty
};
hipError_t grid_scratch_alloc(hipStream_t s, size_t n, GridScratch *sc);   // s: the stream the searches will run on
void grid_scratch_free(GridScratch *sc);
// order: optional processing order of the source rows (grid_order), a permutation of [0, n); results do not depend on it.
// sc == nullptr: every lane finishes its own point, however large its box.
// out_rec: one {key, idx as an integer bit pattern} record per source row i < n (merge_cov's grid_rec) -- written only
// when P.rec is not set; with P.rec the result goes into P.rec[2i+1] = {matched target's coordinates, idx}.
// P.rec, when set, holds two 32-byte halves per source row (Soa3::rec_stride): A = {x, y, z, unused} -- the RUNNING cloud on
// this path: apply updates A only, the SoA arrays are brought up to date by source_rec_to_soa when something needs them --
// and B = {coordinates of the matched target, idx | certificate}, kept current by the search and by merge_cov.
// motion (optional, only used with P.rec and idx_prev): the solve record (kSolveW doubles) of the rigid motion that took the
// source rows from their position at the previous search to the present one, pc being the centroid of the moved cloud.
// With it a row whose previous search left a certificate wide enough for that motion keeps its match without visiting a
// cell (grid.hip); nullptr = every row is searched.
// cen_part != nullptr: pc is an OUTPUT -- the launch finishes the centroid itself from cen_chunks chunk partials
// (pc = fixed-order sum / cen_denom, the result of finalize_centroid) and workgroup 0 stores it in pc for the kernels that
// follow.  search_grid() decides: the search kernel proper does so for clouds of up to kGridFuseCentroidChunks chunks (a launch
// saved for a 3 KB re-read per workgroup), the settled kernel -- whose row loads are in flight meanwhile -- up to
// kGridFuseCentroidChunksSettled (one pass of the fixed-order sum: 32 * (256 / 3) chunks); beyond that search_grid() runs
// finalize_centroid as a launch of its own first.
constexpr size_t kGridFuseCentroidChunks = 128;
constexpr size_t kGridFuseCentroidChunksSettled = 32 * (kWG / kCenW);
// fused (optional): the caller offers the outputs of the merge that would follow (merge_cov with result_in_rec): with source
// records, certificates in play and a registration that the previous search found settled, one launch then does both --
// fused->ran says whether it did (the caller then skips merge_cov).
struct GridFused {
    uint32_t *idx_out = nullptr;
    double reject_d2 = 0.0;
    double *partial = nullptr;
    bool ran = false;
};
hipError_t search_grid(hipStream_t s, Soa3 P, size_t n, double *pc, const Grid &g, Soa3 Qz, size_t m,
                       const uint32_t *idx_prev, const uint32_t *order, GridScratch *sc, double2 *out_rec,
                       const double *motion = nullptr, const double *cen_part = nullptr, size_t cen_chunks = 0,
                       double cen_denom = 1.0, GridFused *fused = nullptr);
// A[i] = {x, y, z, 0}, B[i] = {0, 0, 0, "no index"} for i < rows
hipError_t build_source_rec(hipStream_t s, Soa3 P, size_t rows);
// x, y, z [i] = A[i] for i < rows: the SoA arrays caught up with the running cloud of the records
hipError_t source_rec_to_soa(hipStream_t s, Soa3 P, size_t rows);
// stable (optional): the rows of a cell in ascending index, so that the order is a function of the cloud alone (the counting
// sort leaves them in the order its atomics were served); *stable = false when a cell holds too many rows for that (the order
// is then still a valid permutation).
hipError_t grid_order(hipStream_t s, Soa3 P, size_t n, uint32_t *order, bool *stable = nullptr);

// The reference's own reduction order (icp_amd_options::strict_order), one lane per output.  Source of the rows: aos (n x 3
// row-major) if given, else the records' A half if P.rec is set, else the SoA arrays.
// out3[k] = (((0 + a[0][k]) + a[1][k]) + ...) / n                                     (src/libalg/mean.cpp:106-131)
hipError_t strict_centroid(hipStream_t s, const double *aos, Soa3 P, size_t n, double *out3);
// partial[0][a*3+b] = sum over i, in order, of Qz[idx[i]][a] * (P[i][b] - pc[b])      (src/cpu/icp.cpp:115-129); the same
// nine slots of the other `chunks - 1` partial records are zeroed, so that the fixed-order sum over all records that
// solve() runs returns exactly these nine numbers.  Slots 9.. (sum d^2, pairs) of every record are left as they are.
hipError_t strict_cov(hipStream_t s, Soa3 P, size_t n, const double *pc, Soa3 Qz, size_t m, const uint32_t *idx, double *partial,
                      size_t chunks);

// Merge the splits (lowest index among equal keys), write idx[n], and reduce q (x) p per 1024-row chunk
// into partial[chunk][12].
hipError_t merge_cov(hipStream_t s, Soa3 P, size_t n, size_t n_pad, const double *pc, Soa3 Qz, size_t m,
                     int q_splits, const double *keys, const uint32_t *idxs, const double2 *grid_rec, bool result_in_rec,
                     uint32_t *idx_out, double reject_d2, double *partial);
// result_in_rec: search_grid ran with P.rec set and left {matched target, idx} in P.rec[2i+1]; nothing is gathered.
// grid_rec != nullptr: the (key, idx) records of search_grid instead of keys / idxs (q_splits is then 1).
// P.rec != nullptr: the matched target's coordinates are stored in P.rec[2i+1] (next iteration's seed).
// reject_d2: pairs farther apart than this (squared, the oracle's un-fused d^2) get weight 0 in the sums; +inf = none.
// partial slots: [0..8] cov, [9] sum of d^2, [10] number of pairs that entered the sums, [11] spare.
// cov partials from a given idx (stage-level entry).
hipError_t cov_from_idx(hipStream_t s, Soa3 P, size_t n, const double *pc, Soa3 Qz, size_t m, const uint32_t *idx,
                        double *partial);

// Sum `count` cov partials in the same fixed order, 3x3 SVD, R = U V^T, t = Qc - Pc R^T.
// out: [0..8] cov, [9..11] S, [12..20] R, [21..23] t, [24] rms, [25..27] Pc copy  (28 doubles)
constexpr int kSolveW = 28;
// n_global <= 0: rms over the pairs counted in slot 10.  extra (optional, 4 doubles): pairs used, max |R - I|,
// max |R Pc + t - Pc| -- how far this iteration's motion is from the identity.
hipError_t solve(hipStream_t s, const double *partial, size_t count, const double *pc, const double *qc,
                 double n_global, double *out, double *extra);
// P <- P R^T + t (R, t read from the solve record) and per-chunk sums of the new P.
// P.rec != nullptr: the cloud that is read and written is the records' A half, and ONLY that (the SoA arrays go stale).
hipError_t apply(hipStream_t s, Soa3 P, size_t n, const double *solve_rec, double *cen_partial);

// Multi-registration mode: `batch` independent registrations of n x m points, one workgroup each, all iterations
// on-chip.  P_in [batch][n][3], Q_in [batch][m][3] row-major; P_out same shape; rec_out [batch][kSolveW] = the
// last iteration's solve record; idx_out [batch][n] (both optional).  n <= kBatchMaxN, m <= kBatchMaxM.
constexpr size_t kBatchMaxN = 2048;
constexpr size_t kBatchMaxM = 4096;
size_t batch_lds_bytes(size_t m);
hipError_t batch_icp(hipStream_t s, int variant, bool strict, const double *P_in, const double *Q_in, size_t batch, size_t n, size_t m,
                     unsigned iters, double *P_out, double *rec_out, uint32_t *idx_out);

// test hook: key[i] = sqrt(sqrt(d2[i])) computed on the device
hipError_t fourth_root(hipStream_t s, const double *d2, size_t n, double *key);

}  // namespace icpk
