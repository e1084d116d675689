/* include/icp_amd.h -- C ABI of the MI355X-native ICP engine (libicp_amd.so).
 *
 * Drop-in boundary for the GPU path of FanatoniQ/ICP.  The reference has no FFI: its GPU executable calls
 * two C++ functions (reference include/gpu/icp.cuh:38-39) that in turn call the kernel wrappers of
 * libgpuicp / libgpualg.  This header replaces that seam with plain-C entry points (pointers + sizes, no
 * C++ or torch types): the whole loop (icp_amd_run), a persistent device-resident engine that can be
 * driven phase by phase (used for P-sharding across GPUs), and stage-level calls that mirror the
 * reference's kernel-wrapper level for parity tests.
 *
 * Conventions
 *   - Host clouds are row-major n x 3 doubles, like the reference's CPUMatrix (include/libalg/CPUMatrix.hpp:9-14).
 *   - Every function returns ICP_AMD_OK (0) or a negative ICP_AMD_E* code; icp_amd_last_error() gives the
 *     text (thread-local).  Nothing exits, aborts or throws across this boundary (the reference exit(1)s
 *     on device errors, include/error.cuh:3-9, and throws std::runtime_error on argument errors,
 *     include/error.hpp:6-15; the GPUICP executable built on this ABI restores those behaviours).
 *   - There is no CPU fallback: without a HIP device every compute entry point fails with ICP_AMD_ENODEV.
 *   - Results follow the reference CPU path (src/cpu/icp.cpp:194-229): correspondence = first minimum of
 *     sqrt(sqrt(d^2)) with d^2 accumulated without FMA, cov = sum q (x) p, R = U V^T without reflection
 *     fix, t = Qc - Pc R^T, P <- P R^T + t.
 */
#ifndef ICP_AMD_H
#define ICP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICP_AMD_OK       0
#define ICP_AMD_EINVAL  (-1)  /* bad argument, empty cloud, wrong column count, size over the limits   */
#define ICP_AMD_ENODEV  (-2)  /* no usable HIP device                                                   */
#define ICP_AMD_EHIP    (-3)  /* a HIP runtime call failed (text in icp_amd_last_error)                 */
#define ICP_AMD_ENOMEM  (-4)  /* host or device allocation failed                                       */
#define ICP_AMD_EIO     (-5)  /* file could not be opened / parsed                                      */
#define ICP_AMD_ESTATE  (-6)  /* call sequence error (e.g. iterate before set_source)                   */

/* The four method strings of the reference command line (src/gpu/main.cu:74).  All are served by the one
 * engine (they only chose between CUDA kernel variants in the reference). */
#define ICP_AMD_METHOD_SHARED_LOOP 0
#define ICP_AMD_METHOD_SHARED      1
#define ICP_AMD_METHOD_LOOP        2
#define ICP_AMD_METHOD_BATCH       3

/* Correspondence-search kernel variants (engine option `search`); every variant returns identical indices.
 * An explicit variant is final.  Only AUTO consults the environment: ICP_AMD_SEARCH=exact|fma|f32|grid|auto then decides
 * what AUTO runs (a debugging aid; icp_amd_search_geometry reports the variant that was planned). */
#define ICP_AMD_SEARCH_AUTO   0   /* GRID from 1024 targets on (F32 while most search boxes are huge), else F32;
                                     F32 itself degrades to FMA when the clouds' range defeats the fp32 bound  */
#define ICP_AMD_SEARCH_EXACT  1   /* 8 un-fused fp64 ops per pair, the oracle's own sequence                */
#define ICP_AMD_SEARCH_FMA    2   /* fused fp64 pre-filter + exact re-evaluation of every candidate         */
#define ICP_AMD_SEARCH_F32    3   /* reduced-precision pre-filter with a rigorous error bound + the same exact re-evaluation:
                                     on the fp16 matrix pipe (operands in two fp16 pieces, fp32 accumulation) by default,
                                     on the packed fp32 VALU with ICP_AMD_MFMA=0                                        */
#define ICP_AMD_SEARCH_GRID   4   /* uniform grid over the target: evaluates only the targets near each source point,
                                     with a rounding-safe stop rule; same indices, O(N) instead of O(N M) per iteration */

typedef struct icp_amd_engine icp_amd_engine;

/* Per-iteration outputs, every pointer optional (NULL = not wanted). Host memory owned by the caller. */
typedef struct icp_amd_trace {
    double  *Pc;   /* iters x 3   centroid of the running source cloud                                    */
    int32_t *idx;  /* iters x n   correspondence indices (reference: corresps, src/cpu/icp.cpp:210)        */
    double  *cov;  /* iters x 9   cross-covariance, row = target axis (src/cpu/icp.cpp:108-131)            */
    double  *S;    /* iters x 3   singular values, descending                                              */
    double  *R;    /* iters x 9   rotation U V^T (src/cpu/icp.cpp:221)                                     */
    double  *t;    /* iters x 3   translation (src/cpu/icp.cpp:223)                                        */
    double  *rms;  /* iters       sqrt(mean d^2 of matched pairs), a convergence trace (not in reference)  */
} icp_amd_trace;

typedef struct icp_amd_options {
    int    device;        /* HIP device ordinal                                                            */
    void  *stream;        /* hipStream_t to enqueue on; NULL = the engine creates its own                 */
    int    search;        /* ICP_AMD_SEARCH_*                                                              */
    int    q_splits;      /* target-range splits of the search grid; 0 = heuristic                        */
    int    kernel_timing; /* 1 = record hipEvents around every search launch (icp_amd_kernel_ms)          */
    /* P-sharding (one engine per GPU).  world = 1: single GPU.  Rank r owns the source rows
     * [r*rows_per_rank, min(n_global, (r+1)*rows_per_rank)), rows_per_rank a multiple of 1024. */
    int    rank, world;
    size_t n_global;      /* 0 = the n given to set_source                                                 */
    /* world > 1: device buffers that the CALLER all-gathers between phases (see icp_amd_step_*).
     * Each holds world * chunks_per_rank * width doubles, rank-major; width 12 (cov) / 3 (centroid). */
    double *cov_gather;
    double *cen_gather;
    /* 1 (or ICP_AMD_STRICT_ORDER=1 in the environment): both reductions of an iteration in the reference's own order -- the
     * centroid as a sequential sum over the rows followed by one division (src/libalg/mean.cpp:106-131), the cross-covariance
     * as cov += q_a * p_b row by row (src/cpu/icp.cpp:115-129) -- by one lane per output, so that Pc and cov are BIT-identical
     * to the reference's given the same cloud (the default, fixed-order trees, agrees to ~1e-15).  For clouds of up to 4096
     * source rows (1 Mi target rows for the target's centroid) on one GPU; larger clouds, sharded runs and the rejection hook
     * keep the trees.  Indices are identical either way.  SURVEY.md 7.3-3c. */
    int    strict_order;
    /* 1 (or ICP_AMD_SORT_SOURCE=1 in the environment): the engine keeps a large source cloud (2^18 rows or more, one GPU, not
     * strict order) in a spatial order of its own -- rows sorted by a coarse cell, ascending index inside a cell: a function
     * of the cloud alone -- instead of reaching the caller's rows through a permutation.  The grid search's row accesses are
     * then contiguous: 10 % off a whole 1M x 1M registration.  Clouds and indices go in and come out in the CALLER's order
     * (gather on the way in, scatter on the way out).  What changes: the reduction leaves are 1024-row chunks of the STORED
     * order, so Pc and cov are summed in another order than without the option and differ in the last bits (same bar against
     * the reference, indices exact for the cloud at hand; still the same bits from run to run and across search variants).
     * Off by default: results are then also independent of the GPU count, which a per-rank sort cannot give.  No counterpart
     * in the reference. */
    int    sort_source;
    int    reserved[6];
} icp_amd_options;

const char *icp_amd_last_error(void);
int         icp_amd_version(void);
int         icp_amd_device_count(void);
void        icp_amd_default_options(icp_amd_options *opt);
/* Optional warm-up (no counterpart in the reference, whose first cudaMalloc pays for it): creates the HIP context on
 * `device` (-1: ICP_AMD_DEVICE, default 0) and loads the library's code objects, about 0.15 s in a fresh process.  A
 * caller with input files to parse can run it on another thread meanwhile, as GPUICP does; every other entry point
 * initialises on demand. */
int         icp_amd_init(int device);
/* Device memory released by the library (icp_amd_destroy, a cloud of another size, the one-shot calls) is kept for the
 * next request of about the same size instead of going back to the driver: hipFree costs 0.1 ms per block, a third of a
 * small one-shot registration.  ICP_AMD_CACHE_MB bounds what is kept (default 2048, 0 = keep nothing);
 * icp_amd_cache_trim returns it all to the driver now and reports how many bytes that was.  Likewise the 8 MB pinned host
 * buffers through which clouds of up to that size are copied (at most eight are kept; none with ICP_AMD_CACHE_MB=0; not
 * counted in the bytes reported).  No reference counterpart. */
size_t      icp_amd_cache_trim(void);
/* rows per rank and 1024-row chunks per rank for a P-sharded run (pure host arithmetic). */
size_t      icp_amd_rows_per_rank(size_t n_global, int world);
size_t      icp_amd_chunks_per_rank(size_t n_global, int world);

/* ---- whole path: replaces icp_gpu_optimized / icp_gpu (reference include/gpu/icp.cuh:38-39,
 *      src/gpu/icp.cu:469-652, 275-445).  P, Q borrowed and not modified; P_out caller-owned n x 3. */
int icp_amd_run(const double *P_xyz, size_t n, const double *Q_xyz, size_t m, unsigned iters,
                int method, double *P_out, icp_amd_trace *trace);

/* ---- whole path on several GPUs of one node, single process (no counterpart in the reference, which is
 *      single-device).  The source rows are sharded over the `n_gpus` devices (devices == NULL: ordinals 0..n_gpus-1),
 *      the target is replicated; per iteration the per-chunk partial sums are combined with two RCCL all-gathers over
 *      xGMI (librccl.so is loaded at run time, only by this call), every device then sums all chunks in index order and
 *      solves R, t redundantly.  Results are bit-identical to icp_amd_run for every n_gpus.  trace->idx: n per iteration.
 *      An explicit `devices` list may name a device more than once (several shards per GPU); RCCL cannot form a communicator
 *      over such a list, so the partial sums are then exchanged with peer copies ordered by events -- as they are, on any
 *      list, when ICP_AMD_MULTI_EXCHANGE=copy is set (e.g. where librccl.so is not installed). */
int icp_amd_run_multi(const double *P_xyz, size_t n, const double *Q_xyz, size_t m, unsigned iters,
                      int n_gpus, const int *devices, double *P_out, icp_amd_trace *trace);

/* ---- persistent engine (replaces the allocation / H2D / loop / D2H plan of src/gpu/icp.cu:511-648) */
int icp_amd_create(icp_amd_engine **out, const icp_amd_options *opt);
int icp_amd_destroy(icp_amd_engine *e);
/* Upload + centre the target once (src/gpu/icp.cu:527,537-545; oracle src/cpu/icp.cpp:197-198). */
int icp_amd_set_target(icp_amd_engine *e, const double *Q_xyz, size_t m);
/* Upload this rank's source rows (src/gpu/icp.cu:528). n = local rows. */
int icp_amd_set_source(icp_amd_engine *e, const double *P_xyz, size_t n);
/* Enqueue `iters` iterations (src/gpu/icp.cu:548-628); no host synchronisation.  world must be 1. */
int icp_amd_iterate(icp_amd_engine *e, unsigned iters);
/* ---- hooks the reference declares but leaves inert (weight kernel + exclude_indices of compute_cross_variance,
 *      src/cpu/icp.cpp:108-124; the per-iteration norm it computes and drops, :213).  All off by default, in which case
 *      results equal the reference's.
 * Pairs farther apart than dmax get weight 0 in the cross-covariance (a hard-threshold `kernel`); dmax <= 0 or inf
 * switches the rejection off.  Centroids, and so t = Qc - Pc R^T, stay over all points as in the reference. */
int icp_amd_set_reject_distance(icp_amd_engine *e, double dmax);
/* Iterates until one iteration's motion is within (rot_tol, trans_tol) of the identity -- max |R - I| and
 * max |R Pc + t - Pc|, the displacement of the cloud's centroid -- or max_iters is reached; the test costs one host
 * synchronisation every `check_every` iterations.  world must be 1. */
int icp_amd_iterate_until(icp_amd_engine *e, unsigned max_iters, double rot_tol, double trans_tol, unsigned check_every,
                          unsigned *iters_done);
/* Of the most recent iteration: pairs that entered the covariance, max |R - I|, max |R Pc + t - Pc|.  Synchronises. */
int icp_amd_get_last_motion(icp_amd_engine *e, double *pairs_used, double *delta_rot, double *delta_trans);
/* Same, bracketed by hipEvents on the engine's stream; returns the elapsed device time. */
int icp_amd_iterate_timed(icp_amd_engine *e, unsigned iters, double *elapsed_ms);
/* P-sharded iteration in two phases; the caller all-gathers cov_gather after step_search and cen_gather
 * after step_update (e.g. torch.distributed / RCCL all_gather on the same stream).
 * With world == 1 the two calls are equivalent to icp_amd_iterate(e, 1). */
int icp_amd_step_begin(icp_amd_engine *e);   /* centroid partials of the local rows -> cen_gather slot   */
int icp_amd_step_search(icp_amd_engine *e);  /* centroid, search, cov partials -> cov_gather slot         */
int icp_amd_step_update(icp_amd_engine *e);  /* solve R,t; apply; new centroid partials -> cen_gather slot */
int icp_amd_sync(icp_amd_engine *e);
/* Download the running source rows (src/gpu/icp.cu:629-631). Synchronises. */
int icp_amd_get_source(icp_amd_engine *e, double *P_out);
/* Outputs of the most recent iteration (synchronises). Any pointer may be NULL. idx: local rows. */
int icp_amd_get_last(icp_amd_engine *e, double *Pc, int32_t *idx, double *cov, double *S, double *R,
                     double *t, double *rms);
int icp_amd_get_target_centroid(icp_amd_engine *e, double Qc[3]);
/* Average / total device time of the search launches since create (kernel_timing = 1). Synchronises. */
int icp_amd_kernel_ms(icp_amd_engine *e, double *search_avg_ms, double *search_total_ms, unsigned *launches);
/* Launch geometry chosen for the search kernel (for reports). */
int icp_amd_search_geometry(icp_amd_engine *e, int *points_per_lane, int *q_splits, int *workgroups, int *variant);
/* Search launches so far by kind, and the number of source points the most recent finished grid launch handed to its
 * overflow kernel (for reports; does not synchronise). */
int icp_amd_grid_stats(icp_amd_engine *e, unsigned *grid_launches, unsigned *brute_force_launches, unsigned *last_overflow);

/* ---- multi-registration mode (BASELINE config 5: thousands of small independent registrations).
 *      No counterpart in the reference: its `-batch` flag strip-mines ONE registration 16 source rows at a time
 *      (src/libgpuicp/batchcovs.cu:28-75).  Here a batch is `batch` independent (P_b, Q_b) pairs of n x 3 and m x 3
 *      points, concatenated; each runs `iters` iterations inside one workgroup without leaving the chip, and its
 *      result is bit-identical to icp_amd_run on the same pair.  Limits: n <= 2048, m <= 4096. */
typedef struct icp_amd_batch icp_amd_batch;
int icp_amd_batch_create(icp_amd_batch **out, int device, size_t batch, size_t n, size_t m);
int icp_amd_batch_upload(icp_amd_batch *b, const double *P /* batch*n*3 */, const double *Q /* batch*m*3 */);
/* Runs `iters` iterations from the uploaded clouds (every call starts again from them). elapsed_ms optional.
 * `search`: ICP_AMD_SEARCH_EXACT / _FMA = the all-fp64 difference forms; anything else = the default (a filter on
 * {-2q, |q|^2} records, seeded by the previous iteration's match from the second iteration on; every candidate
 * re-evaluated in the oracle's arithmetic).  All give the same bits. */
int icp_amd_batch_run(icp_amd_batch *b, unsigned iters, int search, double *elapsed_ms);
/* Any pointer may be NULL. R (batch*9), t (batch*3), cov (batch*9), rms (batch): last iteration's. idx: batch*n. */
int icp_amd_batch_download(icp_amd_batch *b, double *P_out, double *R, double *t, double *cov, int32_t *idx, double *rms);
int icp_amd_batch_destroy(icp_amd_batch *b);
/* One-shot convenience over the five calls above.  Clouds beyond the limits of the one-workgroup-per-registration kernel
 * (n <= 2048, m <= 4096) are not an error here: they are registered through ordinary engines, ICP_AMD_BATCH_WORKERS
 * (default 8) registrations at a time on their own host threads and streams, with the results of icp_amd_run. */
int icp_amd_run_batch(const double *P, const double *Q, size_t batch, size_t n, size_t m, unsigned iters,
                      double *P_out, double *R, double *t, int32_t *idx, double *rms);

/* ---- stage level, host pointers in and out, computed on the device with the engine's kernels.
 *      These mirror the reference's kernel-wrapper level and exist for parity tests. */
/* get_array_correspondences_optimized (include/libgpuicp/corresp_optimized.cuh:23-24) with the centring of
 * P (src/gpu/icp.cu:557) fused in: idx[i] = argmin_j key(P[i]-Pc, Qz[j]).  Qz is used as given. */
int icp_amd_correspondences(const double *P_xyz, size_t n, const double Pc[3], const double *Qz_xyz, size_t m,
                            int search, int q_splits, int32_t *idx);
/* The same as a seeded launch: seed[i] is a candidate correspondence of row i (an entry outside [0, m) means none), as the
 * loop hands one iteration's result to the next (no counterpart in the reference, whose kernels start every scan from
 * scratch, corresp_optimized.cu:42-47).  Same indices as the un-seeded call -- a seed only starts the row's running
 * minimum low -- through the kernels' seeded code paths. */
int icp_amd_correspondences_seeded(const double *P_xyz, size_t n, const double Pc[3], const double *Qz_xyz, size_t m,
                                   int search, int q_splits, const int32_t *seed, int32_t *idx);
/* get_array_cross_covs_flattened + reduce_0(SUM) (include/libgpuicp/crosscov.cuh:112-116, src/gpu/icp.cu:579-584) */
int icp_amd_cross_cov(const double *P_xyz, size_t n, const double Pc[3], const double *Qz_xyz, size_t m,
                      const int32_t *idx, double cov[9]);
/* mean_0 (include/libgpualg/mean.cuh; src/gpu/icp.cu:537,551) */
int icp_amd_centroid(const double *A_xyz, size_t n, double out[3]);
/* icp_amd_centroid for callers that cannot allocate.  NO counterpart in the reference: it exists for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc.  It calls no hipMalloc / hipFree / allocating routine.
 * A_dev: n x 3 row-major doubles ALREADY in device memory on `device`; out_dev: 3 doubles in device memory; both kernels
 * are enqueued on `stream` (a hipStream_t; NULL = the null stream) and the call does not synchronise.  The per-chunk partial
 * sums live in a static device array of the library (384 KiB on every GPU the library runs on, held until the process
 * ends), which limits n to icp_amd_centroid_static_max_rows() = 16 Mi rows: larger n is ICP_AMD_EINVAL.  Every call on a
 * GPU shares that GPU's array, so calls that overlap on one GPU (different streams or host threads) race -- serialise them.
 * Under stream capture: call icp_amd_init(device) BEFORE the capture begins.  The first call on a device otherwise looks up
 * the scratch address (hipGetSymbolAddress) and loads the code objects, which need not be legal while a capture is open;
 * after icp_amd_init the call only enqueues its two kernels on `stream` and is captured like any launch.
 * Result: bit-identical to icp_amd_centroid on the same rows. */
int    icp_amd_centroid_static(const double *A_dev, size_t n, double *out_dev, int device, void *stream);
size_t icp_amd_centroid_static_max_rows(void);
/* svd_gpu + matrixMultiplication + gpuTranspose + matrix_op (src/gpu/icp.cu:588-611) */
int icp_amd_solve(const double cov[9], const double Pc[3], const double Qc[3], double R[9], double t[3], double S[3]);
/* matrixMultiplication(P, R^T) + matrix_op(ADD t) (src/gpu/icp.cu:617-626) */
int icp_amd_apply(const double *P_xyz, size_t n, const double R[9], const double t[3], double *P_out);

/* ---- host I/O kept from the reference */
/* readCSV (include/libCSV/csv.hpp:33, src/libCSV/csv.cpp:94-158): optional header line, strtod parsing.
 * *out is malloc'd (free with icp_amd_free). header: optional buffer for the header line.
 * Row counts are the reference's, quirks included: a file WITHOUT a header line and with two or more lines comes back one
 * row short (csv.cpp:113-114,147-148 take its second line for the header text), and a file without any number is an
 * error (csv.cpp:151-156).  Files the reference would turn into an array that disagrees with nblines x nbcols (ragged
 * rows, a first line mixing numbers and text) are rejected with an error instead. */
int    icp_amd_read_csv(const char *path, double **out, size_t *nblines, size_t *nbcols, char *header, size_t header_cap);
void   icp_amd_free(void *p);
/* CPUMatrix::write_in_file / operator<< (src/libalg/CPUMatrix.cpp:81-94,323-327) */
int    icp_amd_write_cloud(const char *path, const double *A, size_t rows, size_t cols);
/* The same parse as icp_amd_read_csv (identical values, shapes and errors) with the lines spread over `threads` host
 * threads (0 = one per core, at most 16): readCSV costs seconds on the 60 MB files of a 1M-point cloud. */
int    icp_amd_read_csv_mt(const char *path, int threads, double **out, size_t *nblines, size_t *nbcols, char *header,
                           size_t header_cap);
/* Cloud by file name (no counterpart in the reference, whose only format is CSV): *.npy (little-endian float64, C order,
 * 1-D or 2-D), *.f64 (raw little-endian float64, 3 columns), anything else = CSV as above.  ICP_AMD_CSV_THREADS=1 forces
 * the serial parser. */
int    icp_amd_read_cloud(const char *path, double **out, size_t *nblines, size_t *nbcols, char *header, size_t header_cap);
/* Result in full precision: *.npy or *.f64 by file name (icp_amd_write_cloud stays the reference's 6-digit text). */
int    icp_amd_write_cloud_binary(const char *path, const double *A, size_t rows, size_t cols);
/* CPUMatrix::euclidianDistance (src/libalg/CPUMatrix.cpp:291-295) on two rows x cols clouds (host, fp64) */
double icp_amd_euclidian_distance(const double *A, const double *B, size_t rows, size_t cols);

#ifdef __cplusplus
}
#endif
#endif /* ICP_AMD_H */
