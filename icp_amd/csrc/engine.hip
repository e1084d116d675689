// icp_amd/csrc/engine.hip -- host side of libicp_amd.so: the device-resident ICP engine and the C ABI
// declared in include/icp_amd.h.  Replaces the driver of the reference GPU path (src/gpu/icp.cu:469-652):
// buffers are allocated once, every iteration is five stream-ordered launches and no host sync.
#include "../../include/icp_amd.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace icpk;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
extern "C" int icp_amd_fail_(int code, const char *fmt, ...);  // hostio.cpp: sets the thread-local error text
#define fail icp_amd_fail_

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? ICP_AMD_ENOMEM : ICP_AMD_EHIP, "HIP failure %s:%d '%s' in %s", \
                        __FILE__, __LINE__, hipGetErrorString(e_), #expr);                                    \
    } while (0)

#define RCCHK(expr)               \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != ICP_AMD_OK) return rc_; \
    } while (0)

extern "C" int icp_amd_version(void) { return 100; }

extern "C" int icp_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Every entry point selects its device through this.  It does NOT touch the host thread's sticky error word
// (hipGetLastError), which belongs to every HIP user of the process: the launch wrappers take each launch's own return
// code (kernels.h: launch()), so a stale error left by another library is neither mistaken for ours nor cleared.
static hipError_t set_device(int device)
{
    return hipSetDevice(device);
}

static int require_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(ICP_AMD_ENODEV, "no HIP device available (%s); libicp_amd has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(ICP_AMD_EINVAL, "device %d out of range [0,%d)", device, n);
    HIPCHK(set_device(device));
    return ICP_AMD_OK;
}

static int env_int(const char *name, int dflt);
static int static_scratch(int device, double **out);

// Context creation + code-object load, for callers that can overlap it with their own I/O.  One tiny launch per
// translation unit with kernels (kernels.hip, grid.hip): HIP loads a unit's code object at its first launch.
extern "C" int icp_amd_init(int device)
{
    RCCHK(require_device(device < 0 ? env_int("ICP_AMD_DEVICE", 0) : device));
    double *buf = nullptr;
    HIPCHK(dmalloc(&buf, 8 * sizeof(double)));
    hipError_t er = hipMemsetAsync(buf, 0, 8 * sizeof(double), nullptr);
    if (er == hipSuccess) er = finalize_centroid(nullptr, buf, 1, 1.0, buf + 3);
    if (er == hipSuccess) er = grid_warmup(nullptr);
    if (er == hipSuccess) er = hipStreamSynchronize(nullptr);
    (void)dfree(buf);
    HIPCHK(er);
    double *sp = nullptr;
    RCCHK(static_scratch(device < 0 ? env_int("ICP_AMD_DEVICE", 0) : device, &sp));   // icp_amd_centroid_static's scratch on this device
    return ICP_AMD_OK;
}

extern "C" size_t icp_amd_cache_trim(void)
{
    const size_t held = dcache_held();
    dcache_trim();
    return held;
}

// ------------------------------------------------------------------------------------------------
// sharding arithmetic
// ------------------------------------------------------------------------------------------------
static size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

extern "C" size_t icp_amd_chunks_per_rank(size_t n_global, int world)
{
    if (world < 1) world = 1;
    const size_t chunks = (n_global + kChunk - 1) / kChunk;
    return (chunks + (size_t)world - 1) / (size_t)world;
}

extern "C" size_t icp_amd_rows_per_rank(size_t n_global, int world)
{
    return icp_amd_chunks_per_rank(n_global, world) * kChunk;
}

extern "C" void icp_amd_default_options(icp_amd_options *opt)
{
    if (!opt) return;
    memset(opt, 0, sizeof *opt);
    opt->world = 1;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
struct icp_amd_engine {
    icp_amd_options opt;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 256;

    // target
    size_t m = 0, m_pad = 0;
    double *q_store = nullptr;  // 3 * m_pad
    double4 *q_rec = nullptr;   // m_pad row-major {x,y,z,0} copy of Qz for gathers by index
    Soa3 Qz{nullptr, nullptr, nullptr};
    double *qc = nullptr;       // 3
    void *qrec = nullptr;       // m_pad 16-byte fp32 records for the f32 pre-filter
    unsigned long long *qmax2_dev = nullptr;
    double qmax = 0.0;          // sqrt(max |q|^2) of the centred target
    bool f32_ok = false;        // the fp32 error bound is meaningful for this target's range
    int mfma_g = 0;             // F32 on the matrix pipe: 32-row source groups per wave (0: packed-VALU kernel)
    bool mfma16 = false;        // ... on the fp16 K = 16 MFMA (operands in two fp16 pieces), needs qrec16
    void *qrec16 = nullptr;     // 2 * m_pad 16-byte halves of the fp16 target records (built when that filter is selected)
    bool qrec16_ok = false;
    bool fuse_centroid = true;  // the grid search finishes the centroid itself where that pays (search_grid() decides; kernels.h)
    bool qrec_ok = false;       // qrec / qmax / f32_ok describe the current target (built with it, or by the first brute-force launch)
    Grid grid;                  // uniform grid over Qz (search variant GRID); built on first use per target
    bool grid_built = false;
    uint32_t *order = nullptr;  // n_pad: source rows in spatial order (GRID), valid when order_ok; or, when `sorted`, the row of the
    bool order_ok = false;      // caller's cloud that each stored row is
    bool sorted = false;        // the source is STORED in spatial order (options.sort_source): stored row k = caller's row order[k]
    GridScratch gscratch;       // queue of the points the grid search hands to its overflow kernel
    double4 *p_rec = nullptr;   // 2 * n_pad: per source row A = {x,y,z,-} in the first half, B = {matched target, idx | certificate} in the second (GRID)
    bool soa_stale = false;     // the running cloud lives in the records' A half (apply updated only that); P.x/y/z are behind
    double2 *grid_out = nullptr;  // n_pad (key, idx) records written by the grid search
    bool use_grid = false;      // planned: search on the grid (variant/ppl/q_splits then describe the brute-force fallback)
    bool grid_auto = false;     // the grid was chosen by AUTO: fall back to brute force while most points overflow
    unsigned bf_left = 0, bf_backoff = 0;   // AUTO: brute-force iterations left before the grid is tried again
    unsigned grid_launches = 0, bf_launches = 0;
    unsigned grid_since_source = 0;   // grid searches since the source cloud was set (overflow-kernel policy)

    // source (local rows)
    size_t n = 0, n_pad = 0, n_global = 0;
    size_t chunks_local = 0, cpr = 0, total_chunks = 0;
    double *p_store = nullptr;  // 3 * n_pad
    Soa3 P{nullptr, nullptr, nullptr};
    double *pc = nullptr;       // 3

    double *stage = nullptr;    // AoS staging for uploads / downloads
    size_t stage_rows = 0;
    void *pin = nullptr;        // pinned host staging for small copies to and from the caller's (pageable) buffers (copy_in)
    bool pin_busy = false;      // a copy out of pin may still be in flight on the stream

    // search
    int variant = kExact, ppl = 4, q_splits = 1, q_splits_first = 1, cb = 8;
    unsigned updates_since_search = 0;   // step_update calls since the last search (the grid's motion certificate needs exactly 1)
    bool grid_cert = true;      // ICP_AMD_GRID_CERT=0: every row is searched every iteration (tests)
    bool have_prev = false;     // idx holds the previous iteration's correspondences (search seed)
    bool seed_enabled = true;   // ICP_AMD_SEED=0 switches seeding off (read when the search is planned)
    double *keys = nullptr;
    uint32_t *idxs = nullptr;
    size_t split_cap = 0;       // allocated q_splits * n_pad
    uint32_t *idx = nullptr;    // n_pad

    // partials
    double *cov_part = nullptr, *cen_part = nullptr;  // world*cpr*{12,3}; own or caller's gather buffers
    bool own_part = false;
    double *rec = nullptr;      // kSolveW doubles: outputs of the most recent solve; + 4: pairs used, |R - I|, |motion of Pc|, |R^T R - I|_F
    double reject_d2 = std::numeric_limits<double>::infinity();   // f-4: pairs farther apart get weight 0 (inf = reference)

    bool have_target = false, have_source = false, cen_ready = false;
    bool strict = false;        // the reference's reduction order where the cloud is small enough (icp_amd_options::strict_order)


    // timing of the search launches
    std::vector<hipEvent_t> ev;  // pairs
    size_t ev_used = 0;
    double search_ms_total = 0;
    unsigned search_launches = 0;
};

// a block of engine e: only work enqueued on e->stream has touched it, so the release waits for that stream alone and
// engines driven by other host threads keep running
static void free_dev(icp_amd_engine *e, void *p) { if (p) (void)dfree_on(p, e->stream); }

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

extern "C" int icp_amd_create(icp_amd_engine **out, const icp_amd_options *opt_in)
{
    if (!out) return fail(ICP_AMD_EINVAL, "icp_amd_create: out is NULL");
    *out = nullptr;
    icp_amd_options opt;
    if (opt_in) opt = *opt_in; else icp_amd_default_options(&opt);
    if (opt.world < 1) opt.world = 1;
    if (opt.rank < 0 || opt.rank >= opt.world) return fail(ICP_AMD_EINVAL, "rank %d not in [0,%d)", opt.rank, opt.world);
    if (opt.world > 1 && (!opt.cov_gather || !opt.cen_gather || opt.n_global == 0))
        return fail(ICP_AMD_EINVAL, "world > 1 needs cov_gather, cen_gather and n_global");
    RCCHK(require_device(opt.device));
    icp_amd_engine *e = new (std::nothrow) icp_amd_engine();
    if (!e) return fail(ICP_AMD_ENOMEM, "out of host memory");
    e->opt = opt;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, opt.device) == hipSuccess && prop.multiProcessorCount > 0)
        e->cu_count = prop.multiProcessorCount;
    if (opt.stream) {
        e->stream = (hipStream_t)opt.stream;
    } else {
        hipError_t er = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (er != hipSuccess) { delete e; return fail(ICP_AMD_EHIP, "hipStreamCreate: %s", hipGetErrorString(er)); }
        e->own_stream = true;
    }
    hipError_t er = dmalloc(&e->qc, 3 * sizeof(double));
    if (er == hipSuccess) er = dmalloc(&e->pc, 3 * sizeof(double));
    if (er == hipSuccess) er = dmalloc(&e->rec, (kSolveW + 4) * sizeof(double));

    if (er != hipSuccess) { icp_amd_destroy(e); return fail(ICP_AMD_ENOMEM, "hipMalloc: %s", hipGetErrorString(er)); }
    *out = e;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_destroy(icp_amd_engine *e)
{
    if (!e) return ICP_AMD_OK;
    (void)hipSetDevice(e->opt.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    free_dev(e, e->q_store); free_dev(e, e->q_rec); free_dev(e, e->qc); free_dev(e, e->qrec); free_dev(e, e->qmax2_dev); free_dev(e, e->p_store); free_dev(e, e->pc); free_dev(e, e->stage);
    free_dev(e, e->keys); free_dev(e, e->idxs); free_dev(e, e->idx); free_dev(e, e->rec); free_dev(e, e->qrec16);
    grid_free(&e->grid);
    free_dev(e, e->order);
    grid_scratch_free(&e->gscratch);
    free_dev(e, e->p_rec); free_dev(e, e->grid_out);
    if (e->own_part) { free_dev(e, e->cov_part); free_dev(e, e->cen_part); }
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    pin_release(e->pin);   // (the stream was synchronised above)
    delete e;
    return ICP_AMD_OK;
}

static int ensure_stage(icp_amd_engine *e, size_t rows)
{
    if (rows <= e->stage_rows) return ICP_AMD_OK;
    free_dev(e, e->stage);
    e->stage = nullptr;
    e->stage_rows = 0;
    HIPCHK(dmalloc(&e->stage, rows * 3 * sizeof(double)));
    e->stage_rows = rows;
    return ICP_AMD_OK;
}

// Copies between the caller's buffers and the device.  The caller's memory is pageable, and below about half a megabyte the
// runtime moves such a buffer at 3 GB/s (tools/ubench_h2d: 480 KB in 0.165 ms each way, 0.035 ms through pinned memory), up to
// a few megabytes at a speed that depends on where the buffer lies (set_source of 48 485 rows: 0.37 ms, of 100 000: 0.13 ms):
// clouds of up to kPinBytes (8 MB, 349 525 rows) go through a pinned buffer (devmem.cpp keeps them between engines); larger
// ones, which the runtime moves at full speed, directly.  copy_in leaves the copy enqueued on the engine's stream (the
// caller's buffer may be reused at once when staged); copy_out returns with the data in place.
static int copy_in(icp_amd_engine *e, void *dev, const void *host, size_t bytes)
{
    if (bytes == 0) return ICP_AMD_OK;
    if (bytes > kPinBytes) {
        HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, e->stream));
        return ICP_AMD_OK;
    }
    if (!e->pin) HIPCHK(pin_acquire(&e->pin));
    if (e->pin_busy) HIPCHK(hipStreamSynchronize(e->stream));
    memcpy(e->pin, host, bytes);
    HIPCHK(hipMemcpyAsync(dev, e->pin, bytes, hipMemcpyHostToDevice, e->stream));
    e->pin_busy = true;
    return ICP_AMD_OK;
}
static int copy_out(icp_amd_engine *e, void *host, const void *dev, size_t bytes)
{
    if (bytes == 0 || bytes > kPinBytes) {
        if (bytes) HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->pin_busy = false;
        return ICP_AMD_OK;
    }
    if (!e->pin) HIPCHK(pin_acquire(&e->pin));
    HIPCHK(hipMemcpyAsync(e->pin, dev, bytes, hipMemcpyDeviceToHost, e->stream));   // stream order: behind any copy out of pin
    HIPCHK(hipStreamSynchronize(e->stream));
    e->pin_busy = false;
    memcpy(host, e->pin, bytes);
    return ICP_AMD_OK;
}

static int plan_search(icp_amd_engine *e);
static int prepare_grid_source(icp_amd_engine *e);

static const size_t kMaxRows = 0x7FFFFC00u;  // indices are 32-bit; padded sizes must fit too
static const size_t kGridOrderMinRows = (size_t)1 << 18;   // source rows from which the grid search takes them in a spatial order
static const size_t kStrictMaxSource = 4096, kStrictMaxTarget = (size_t)1 << 20;   // one lane sums these sequentially
static const size_t kSortMinRows = (size_t)1 << 18;   // options.sort_source: sources from this many rows on are stored in spatial order (below,
                                                       // computing the order costs a short registration more than it saves)
static bool strict_requested(const icp_amd_options &opt)
{
    if (opt.strict_order) return true;
    const char *v = getenv("ICP_AMD_STRICT_ORDER");
    if (!v) v = getenv("ICP_STRICT_ORDER");   // the name SURVEY.md 7.3 uses
    return v && atoi(v) != 0;
}

static bool sort_requested(const icp_amd_options &opt)
{
    if (opt.sort_source) return true;
    const char *v = getenv("ICP_AMD_SORT_SOURCE");
    return v && atoi(v) != 0;
}

static bool wants_grid(const icp_amd_engine *e);
static void choose_geometry(icp_amd_engine *e);

// fp32 target records + range of the target, for the brute-force kernels; synchronises the stream (one 8-byte read-back)
static int bruteforce_target(icp_amd_engine *e)
{
    if (e->qrec_ok) return ICP_AMD_OK;
    HIPCHK(build_qrec(e->stream, e->Qz, e->m, e->m_pad, e->qrec, e->qmax2_dev));
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, e->qmax2_dev, sizeof bits, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    double qmax2;
    memcpy(&qmax2, &bits, sizeof qmax2);
    e->f32_ok = (qmax2 == qmax2) && qmax2 > 1e-30 && qmax2 < 1e30;
    e->qmax = e->f32_ok ? sqrt(qmax2) : 0.0;
    e->qrec_ok = true;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_set_target(icp_amd_engine *e, const double *Q, size_t m)
{
    if (!e || !Q) return fail(ICP_AMD_EINVAL, "icp_amd_set_target: NULL argument");
    if (m == 0) return fail(ICP_AMD_EINVAL, "target cloud is empty");
    if (m > kMaxRows) return fail(ICP_AMD_EINVAL, "target cloud too large (%zu rows, limit %zu)", m, kMaxRows);
    HIPCHK(set_device(e->opt.device));
    const size_t m_pad = round_up(m, kPadRows);
    if (m_pad != e->m_pad) {
        free_dev(e, e->q_store); free_dev(e, e->qrec); free_dev(e, e->q_rec); free_dev(e, e->qrec16);
        e->q_store = nullptr; e->qrec = nullptr; e->q_rec = nullptr; e->qrec16 = nullptr;
        e->m_pad = 0;
        HIPCHK(dmalloc(&e->q_store, 3 * m_pad * sizeof(double)));
        HIPCHK(dmalloc(&e->q_rec, m_pad * sizeof(double4)));
        HIPCHK(dmalloc(&e->qrec, 16 * m_pad));
        if (!e->qmax2_dev) HIPCHK(dmalloc(&e->qmax2_dev, sizeof(unsigned long long)));
        e->m_pad = m_pad;
        e->Qz = Soa3{e->q_store, e->q_store + m_pad, e->q_store + 2 * m_pad, e->q_rec};
    }
    e->m = m;
    RCCHK(ensure_stage(e, m));
    RCCHK(copy_in(e, e->stage, Q, m * 3 * sizeof(double)));
    // Q_center = Q.mean(0); Q -= Q_center (reference src/cpu/icp.cpp:197-198), then SoA + inf padding
    const size_t chunks = (m + kChunk - 1) / kChunk;
    double *part = nullptr;
    HIPCHK(dmalloc(&part, chunks * kCenW * sizeof(double)));
    e->strict = strict_requested(e->opt) && e->opt.world == 1;
    hipError_t er;
    if (e->strict && m <= kStrictMaxTarget) {
        er = strict_centroid(e->stream, e->stage, Soa3{nullptr, nullptr, nullptr}, m, e->qc);
    } else {
        er = colsum_partials(e->stream, e->stage, Soa3{nullptr, nullptr, nullptr}, m, part);
        if (er == hipSuccess) er = finalize_centroid(e->stream, part, chunks, (double)m, e->qc);
    }
    if (er == hipSuccess)
        er = aos_to_soa(e->stream, e->stage, m, m_pad, e->qc, std::numeric_limits<double>::infinity(), e->Qz);
    if (er == hipSuccess) er = build_rec(e->stream, e->Qz, m_pad);
    (void)dfree_on(part, e->stream);
    HIPCHK(er);
    // The brute-force kernels' own view of the target (fp32 records, the largest |q|, whether the fp32 bound means anything)
    // is needed at once only where brute force is the search; behind the grid it is AUTO's rare fallback and is built by
    // the first launch that needs it (bruteforce_target): 0.2 ms and a host synchronisation less per registration.
    e->qrec_ok = false;
    e->qrec16_ok = false;
    e->f32_ok = true;   // provisional until bruteforce_target() has seen the target's range
    e->qmax = 0.0;
    if (!wants_grid(e)) RCCHK(bruteforce_target(e));
    else {
        HIPCHK(hipStreamSynchronize(e->stream));   // the caller may reuse Q once this returns
        e->pin_busy = false;
    }
    e->have_target = true;
    grid_free(&e->grid);
    e->grid_built = false;
    e->have_prev = false;                       // correspondences into the previous target are no seed for this one
    if (e->have_source) {
        RCCHK(plan_search(e));  // the variant and the splits depend on the target
        RCCHK(prepare_grid_source(e));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->bf_left = e->bf_backoff = 0;
    }
    return ICP_AMD_OK;
}

// GRID evaluates only the targets near each source point (grid.hip); the others are brute force over all pairs.
// Returns 0 for AUTO.  An explicit opt.search wins; the environment variable ICP_AMD_SEARCH only decides what AUTO means.
static int requested_variant(const icp_amd_engine *e)
{
    int variant = e->opt.search == ICP_AMD_SEARCH_EXACT  ? kExact
                  : e->opt.search == ICP_AMD_SEARCH_FMA  ? kFma
                  : e->opt.search == ICP_AMD_SEARCH_F32  ? kF32
                  : e->opt.search == ICP_AMD_SEARCH_GRID ? kGrid
                                                         : 0;
    if (variant != 0) return variant;
    const char *sv = getenv("ICP_AMD_SEARCH");
    if (sv && !strcmp(sv, "fma")) variant = kFma;
    if (sv && !strcmp(sv, "exact")) variant = kExact;
    if (sv && !strcmp(sv, "f32")) variant = kF32;
    if (sv && !strcmp(sv, "grid")) variant = kGrid;
    if (sv && !strcmp(sv, "auto")) variant = 0;
    return variant;
}
// F32's pre-filter runs on the matrix pipe: the fp16 K = 16 MFMA with every operand in two fp16 pieces (search_mfma.hip; 2.5
// times the packed-VALU kernel at 1M x 1M, same results).  ICP_AMD_MFMA=0 selects the packed-VALU kernel, ICP_AMD_MFMA_G the
// 32-row source groups per wave of the matrix kernel.
static int mfma_groups(size_t n_pad)
{
    if (knob_int("mfma", 16) == 0) return 0;
    // 128 source rows x groups per workgroup: small clouds need the workgroups more than the operand reuse
    const int g = knob_int("mfma_g", n_pad < 120000 ? 2 : 4);
    return g == 2 ? 2 : 4;   // four waves per SIMD: eight groups would not fit the register file
}
static bool mfma_fp16() { return knob_int("mfma", 16) != 0; }

// AUTO searches on the grid from this many targets on: the grid saves 18 us per iteration at 1000 targets (36 at 2000)
// and costs about 250 us more to set up (tools/tiny_probe.py), so it pays from roughly 15 iterations on here.
static const size_t kAutoGridMinTargets = 1024;
static bool wants_grid(const icp_amd_engine *e)
{
    const int v = requested_variant(e);
    return v == kGrid || (v == 0 && e->m >= kAutoGridMinTargets);
}

// Launch geometry of the search kernel: enough workgroups to load every CU several times over, without
// cutting the target range into pieces so short that every piece re-learns its running minimum.
static void choose_geometry(icp_amd_engine *e)
{
    // Every variant returns identical indices (each candidate is re-evaluated in the oracle's un-fused fp64 sequence);
    // they differ in what the fast path costs per pair: EXACT 9.1 fp64 VALU instructions, FMA 7.1 fp64, F32 2.0 packed fp32.
    // AUTO = GRID from kAutoGridMinTargets targets on, with F32 as the fallback while the clouds are so far from each
    // other that most search boxes are huge; F32 itself degrades to FMA when the fp32 bound is meaningless.
    int variant = requested_variant(e);
    e->use_grid = wants_grid(e) && e->grid_built && e->grid.ok;   // not ok: non-finite target coordinates
    e->grid_auto = variant == 0;
    if (variant == 0 || variant == kGrid) variant = kF32;
    if (variant == kF32 && !e->f32_ok) variant = kFma;
    e->variant = variant;

    const bool f32 = variant == kF32;
    const size_t cus = (size_t)e->cu_count;
    // Source points per lane: more points share one LDS read of a target (the big clouds are LDS-read bound below 8 for
    // the fp32 filter, below 4 for fp64), fewer give more workgroups to spread a small cloud over the chip.
    int ppl = f32 ? (e->n_pad <= 32768 ? 2 : e->n_pad <= 524288 ? 4 : 8) : (e->n_pad <= 32768 ? 2 : 4);
    const int env_ppl = knob_int("ppl", 0);
    if ((env_ppl == 1 && !f32) || env_ppl == 2 || env_ppl == 4 || (env_ppl == 8 && f32)) ppl = env_ppl;
    // Workgroups: each one pays a fixed price (its seed blocks, its tile loads, one record per source point and split
    // for the merge), while too few leave CUs idle at the tail -- the optimum grows like sqrt(pairs).  Measured on
    // MI355X (tools/sweep_small.py, profiles/r1_sweep_small.txt): 520 workgroups at 20k x 20k, 1k at 48k, 2-4k at
    // 100k, 8k at 300k, 8-16k at 1M x 1M (flat); never fewer than two per CU.
    const size_t scan_blocks = search_scan_rows(e->m) / kSplitGran;
    const double pairs = (double)e->n_pad * (double)search_scan_rows(e->m);
    size_t want = (size_t)(sqrt(pairs) / 40.0 * ((double)cus / 256.0));
    if (want < 2 * cus) want = 2 * cus;
    if (want > 64 * cus) want = 64 * cus;
    e->mfma_g = f32 ? mfma_groups(e->n_pad) : 0;
    e->mfma16 = e->mfma_g && mfma_fp16();
    const size_t xblocks = e->n_pad / (e->mfma_g ? (size_t)search_mfma_rows_per_workgroup(e->mfma_g) : (size_t)kWG * ppl);
    size_t splits = (want + xblocks - 1) / xblocks;
    if (splits > 1024) splits = 1024;
    if (e->opt.q_splits > 0) splits = (size_t)e->opt.q_splits;
    splits = (size_t)knob_int("qsplits", (int)splits);
    if (splits < 1) splits = 1;
    if (splits > scan_blocks) splits = scan_blocks;
    e->ppl = ppl;
    e->q_splits = (int)splits;
    // an un-seeded search re-learns its running minimum in every split: keep the first iteration coarser
    const size_t first_cap = f32 ? 8 : 16;
    e->q_splits_first = (int)(splits > first_cap ? first_cap : splits);
    int cb = knob_int("cb", 32);
    if (cb != 8 && cb != 16 && cb != 32) cb = 32;
    if (f32 && cb == 8) cb = 16;
    e->cb = cb;
}

static int ensure_split_buffers(icp_amd_engine *e)
{
    const size_t need = (size_t)e->q_splits * e->n_pad;
    if (need > e->split_cap) {
        free_dev(e, e->keys); free_dev(e, e->idxs);
        e->keys = nullptr; e->idxs = nullptr; e->split_cap = 0;
        HIPCHK(dmalloc(&e->keys, need * sizeof(double)));
        HIPCHK(dmalloc(&e->idxs, need * sizeof(uint32_t)));
        e->split_cap = need;
    }
    return ICP_AMD_OK;
}

// Launch geometry + the per-split (key, idx) scratch it needs; re-run whenever the source or the target changes.
static int plan_search(icp_amd_engine *e)
{
    if (wants_grid(e) && !e->grid_built) {
        HIPCHK(grid_build(e->stream, e->Qz, e->m, &e->grid));
        e->grid_built = true;
    }
    choose_geometry(e);
    if (!e->use_grid && !e->qrec_ok) {
        // brute force is the planned search after all (a target the grid cannot file): f32_ok / qmax were provisional until
        // now, and the geometry reported and the split buffers sized below must be those of the variant that will run
        RCCHK(bruteforce_target(e));
        choose_geometry(e);
    }
    e->seed_enabled = knob_int("seed", 1) != 0;
    e->grid_cert = knob_int("grid_cert", 1) != 0;
    e->fuse_centroid = knob_int("grid_fuse_centroid", 1) != 0;
    // On the grid path the brute-force kernels are only AUTO's fallback: their scratch (12 bytes per source row and
    // target split, 0.4 GB at 1M x 1M) is then allocated by the first fallback launch instead of here.
    if (!e->use_grid) RCCHK(ensure_split_buffers(e));
    return ICP_AMD_OK;
}

// Per-source state of the grid search (overflow queue, output records, the 64-byte row records, the spatial order);
// built from the source cloud as it is now, whenever the source or the target is replaced.
// On the grid path with source records the transform apply updates the records' A half only; whatever reads the SoA arrays
// (a brute-force fallback launch, the download of the cloud, a rebuild of the records) first brings them up to date.
static int sync_soa(icp_amd_engine *e)
{
    if (e->soa_stale && e->P.rec && e->n) HIPCHK(source_rec_to_soa(e->stream, e->P, e->n));
    e->soa_stale = false;
    return ICP_AMD_OK;
}

static int prepare_grid_source(icp_amd_engine *e)
{
    const size_t n = e->n, n_pad = e->n_pad;
    RCCHK(sync_soa(e));   // the records are rebuilt from the SoA arrays below
    e->order_ok = false;    // (a source stored in spatial order needs none: e->order then maps stored rows to the caller's)
    e->P.rec = nullptr;
    e->P.rec_stride = 0;
    if (n && wants_grid(e)) {
        if (!e->gscratch.list) HIPCHK(grid_scratch_alloc(e->stream, n_pad, &e->gscratch));
        if (!e->grid_out) HIPCHK(dmalloc(&e->grid_out, n_pad * sizeof(double2)));
        if (knob_int("grid_srcrec", 1)) {
            if (!e->p_rec) HIPCHK(dmalloc(&e->p_rec, 2 * n_pad * sizeof(double4)));
            e->P.rec = e->p_rec;
            e->P.rec_stride = n_pad;
            HIPCHK(build_source_rec(e->stream, e->P, n));
        }
        // The spatial order of the rows pays from a few hundred thousand rows on: below, the rows' cells share the L2 anyway, the
        // iterations measure the same or slower with it (tools/order_small_probe.py: 100k rows x 20 iterations 0.86 ms without,
        // 0.91 with; 300k: 1.53 / 1.43) and computing it is 0.1-0.15 ms of set_source -- a sixth of a whole registration of
        // the cow pair.  grid_order=1 / 0 forces it on / off.
        const int order_knob = knob_int("grid_order", -1);
        if (!e->sorted && (order_knob > 0 || (order_knob < 0 && n >= kGridOrderMinRows))) {
            if (!e->order) HIPCHK(dmalloc(&e->order, n_pad * sizeof(uint32_t)));
            HIPCHK(grid_order(e->stream, e->P, n, e->order));
            e->order_ok = true;
        }
    }
    return ICP_AMD_OK;
}

extern "C" int icp_amd_set_source(icp_amd_engine *e, const double *P, size_t n)
{
    if (!e || (!P && n)) return fail(ICP_AMD_EINVAL, "icp_amd_set_source: NULL argument");
    if (!e->have_target) return fail(ICP_AMD_ESTATE, "icp_amd_set_source: call icp_amd_set_target first");
    const int world = e->opt.world;
    const size_t n_global = (world > 1) ? e->opt.n_global : (e->opt.n_global ? e->opt.n_global : n);
    if (n_global == 0) return fail(ICP_AMD_EINVAL, "source cloud is empty");
    if (n_global > kMaxRows) return fail(ICP_AMD_EINVAL, "source cloud too large (%zu rows, limit %zu)", n_global, kMaxRows);
    const size_t cpr = icp_amd_chunks_per_rank(n_global, world);
    const size_t first = (size_t)e->opt.rank * cpr * kChunk;
    const size_t expect = first >= n_global ? 0 : ((n_global - first < cpr * kChunk) ? n_global - first : cpr * kChunk);
    if (n != expect)
        return fail(ICP_AMD_EINVAL, "rank %d of %d must own %zu source rows (got %zu)", e->opt.rank, world, expect, n);
    HIPCHK(set_device(e->opt.device));
    const size_t n_pad = round_up(n ? n : 1, kPadRows);
    if (n_pad != e->n_pad) {
        free_dev(e, e->p_store); free_dev(e, e->idx); free_dev(e, e->order);
        grid_scratch_free(&e->gscratch);
        free_dev(e, e->p_rec); free_dev(e, e->grid_out);
        e->p_rec = nullptr; e->grid_out = nullptr;
        e->p_store = nullptr; e->idx = nullptr; e->order = nullptr; e->n_pad = 0;
        HIPCHK(dmalloc(&e->p_store, 3 * n_pad * sizeof(double)));
        HIPCHK(dmalloc(&e->idx, n_pad * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(e->idx, 0xFF, n_pad * sizeof(uint32_t), e->stream));   // ordered with the kernels, like everything else
        e->n_pad = n_pad;
        e->P = Soa3{e->p_store, e->p_store + n_pad, e->p_store + 2 * n_pad};
    }
    e->n = n;
    e->n_global = n_global;
    e->cpr = cpr;
    e->chunks_local = (n + kChunk - 1) / kChunk;
    const size_t total = cpr * (size_t)world;
    if (e->opt.cov_gather && e->opt.cen_gather) {   // caller-owned exchange buffers (required when world > 1)
        if (e->own_part) { free_dev(e, e->cov_part); free_dev(e, e->cen_part); }
        e->cov_part = e->opt.cov_gather;
        e->cen_part = e->opt.cen_gather;
        e->own_part = false;
    } else if (total != e->total_chunks || !e->own_part) {
        if (e->own_part) { free_dev(e, e->cov_part); free_dev(e, e->cen_part); }
        e->cov_part = e->cen_part = nullptr;
        HIPCHK(dmalloc(&e->cov_part, total * kCovW * sizeof(double)));
        HIPCHK(dmalloc(&e->cen_part, total * kCenW * sizeof(double)));
        e->own_part = true;
    }
    e->total_chunks = total;
    RCCHK(plan_search(e));
    // this rank's slots of the partial arrays: unused trailing records stay zero for the whole run
    double *cov_slot = e->cov_part + (size_t)e->opt.rank * cpr * kCovW;
    double *cen_slot = e->cen_part + (size_t)e->opt.rank * cpr * kCenW;
    HIPCHK(hipMemsetAsync(cov_slot, 0, cpr * kCovW * sizeof(double), e->stream));
    HIPCHK(hipMemsetAsync(cen_slot, 0, cpr * kCenW * sizeof(double), e->stream));
    if (n) {
        RCCHK(ensure_stage(e, n));
        RCCHK(copy_in(e, e->stage, P, n * 3 * sizeof(double)));
    }
    HIPCHK(aos_to_soa(e->stream, e->stage, n, n_pad, nullptr, 0.0, e->P));
    e->soa_stale = false;
    // options.sort_source: the rows sorted by a coarse cell of their own bounding volume, ascending index inside a cell (a
    // function of the cloud alone), and the cloud stored in that order from here on.  One GPU, not under strict order (whose
    // sums follow the caller's rows); a cloud with a cell too full to order (thousands of coincident rows) stays as it is.
    e->sorted = false;
    if (n >= (size_t)knob_int("sort_min", (int)kSortMinRows) && world == 1 && sort_requested(e->opt) &&
        !(e->strict && n <= kStrictMaxSource)) {
        if (!e->order) HIPCHK(dmalloc(&e->order, n_pad * sizeof(uint32_t)));
        bool stable = false;
        HIPCHK(grid_order(e->stream, e->P, n, e->order, &stable));
        if (stable) {
            HIPCHK(aos_gather_to_soa(e->stream, e->stage, e->order, n, n_pad, 0.0, e->P));
            e->sorted = true;
        }
    }
    RCCHK(prepare_grid_source(e));
    e->bf_left = e->bf_backoff = 0;
    HIPCHK(hipStreamSynchronize(e->stream));  // P may be reused by the caller after return
    e->pin_busy = false;
    e->have_source = true;
    e->have_prev = false;
    e->cen_ready = false;
    e->grid_since_source = 0;
    e->updates_since_search = 0;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_step_begin(icp_amd_engine *e)
{
    if (!e || !e->have_source) return fail(ICP_AMD_ESTATE, "icp_amd_step_begin: no source cloud");
    HIPCHK(set_device(e->opt.device));
    double *cen_slot = e->cen_part + (size_t)e->opt.rank * e->cpr * kCenW;
    RCCHK(sync_soa(e));
    HIPCHK(colsum_partials(e->stream, nullptr, e->P, e->n, cen_slot));
    e->cen_ready = true;
    return ICP_AMD_OK;
}

static int time_begin(icp_amd_engine *e, hipEvent_t *a, hipEvent_t *b)
{
    if (e->ev_used + 2 > e->ev.size()) {
        hipEvent_t x, y;
        HIPCHK(hipEventCreate(&x));
        HIPCHK(hipEventCreate(&y));
        e->ev.push_back(x);
        e->ev.push_back(y);
    }
    *a = e->ev[e->ev_used];
    *b = e->ev[e->ev_used + 1];
    e->ev_used += 2;
    HIPCHK(hipEventRecord(*a, e->stream));
    return ICP_AMD_OK;
}

static int drain_events(icp_amd_engine *e)
{
    for (size_t k = 0; k + 1 < e->ev_used; k += 2) {
        HIPCHK(hipEventSynchronize(e->ev[k + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e->ev[k], e->ev[k + 1]));
        e->search_ms_total += ms;
        e->search_launches += 1;
    }
    e->ev_used = 0;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_step_search(icp_amd_engine *e)
{
    if (!e || !e->have_source) return fail(ICP_AMD_ESTATE, "icp_amd_step_search: no source cloud");
    if (!e->cen_ready) return fail(ICP_AMD_ESTATE, "icp_amd_step_search: centroid partials missing (step_begin / step_update first)");
    HIPCHK(set_device(e->opt.device));
    double *cov_slot = e->cov_part + (size_t)e->opt.rank * e->cpr * kCovW;
    // P_center = P_copy.mean(0) (src/cpu/icp.cpp:206): chunk partials of every rank, summed in index order -- by a launch of
    // its own, or (grid search: the settled kernel, or any search of a cloud of few chunks) by the search kernel itself
    if (!e->n) HIPCHK(finalize_centroid(e->stream, e->cen_part, e->total_chunks, (double)e->n_global, e->pc));
    if (e->n) {
        hipEvent_t a = nullptr, b = nullptr;
        const bool seed = e->have_prev && e->seed_enabled;
        bool grid = e->use_grid && e->gscratch.list && e->grid_out;
        if (grid && e->grid_auto) {
            // The queue length of the most recent grid launch that has finished (a mapped host word, read without
            // waiting): when most points overflow, the tiled brute-force kernel is the faster exact search.  The grid
            // is tried again after 1, 2, 4, ... brute-force iterations (the clouds move towards each other).
            if (e->bf_left == 0 && *e->gscratch.host_mem > e->n / 8) {
                e->bf_backoff = e->bf_backoff ? (e->bf_backoff < 64 ? 2 * e->bf_backoff : 64) : 1;
                e->bf_left = e->bf_backoff;
                *e->gscratch.host_mem = 0;
            }
            if (e->bf_left) { --e->bf_left; grid = false; }
        }
        if (!grid) RCCHK(sync_soa(e));   // the brute-force kernels and their merge read the SoA arrays
        if (!grid && !e->qrec_ok) {   // the first brute-force launch on a target that was set up for the grid
            RCCHK(bruteforce_target(e));
            choose_geometry(e);
        }
        // strict order: the centroid as one sequential sum over the cloud as it is now, the covariance likewise (below)
        const bool strict = e->strict && e->n <= kStrictMaxSource && e->reject_d2 == std::numeric_limits<double>::infinity();
        const bool fuse_centroid = !strict && grid && e->fuse_centroid;   // search_grid() finishes the centroid: inside its kernel, or by a launch of its own
        if (strict) {
            if (!e->P.rec) RCCHK(sync_soa(e));
            HIPCHK(strict_centroid(e->stream, nullptr, e->P, e->n, e->pc));
        } else if (!fuse_centroid)
            HIPCHK(finalize_centroid(e->stream, e->cen_part, e->total_chunks, (double)e->n_global, e->pc));
        if (e->opt.kernel_timing) {
            if (e->ev_used >= 4096) RCCHK(drain_events(e));
            RCCHK(time_begin(e, &a, &b));
        }
        // A brute-force launch without a previous correspondence (the first iteration of a run) re-learns every row's
        // running minimum ~ln(M) times on its exact path: 60 ms against 27 ms seeded at 1M x 1M.  Where a grid can be had,
        // its exact answer (0.5 ms) is that seed; the brute-force launch still evaluates every pair.  ICP_AMD_GRID_SEED=0
        // keeps the un-seeded launch (tests of that code path).
        bool grid_seeded = false;
        if (!grid && !seed && e->seed_enabled && e->m >= kAutoGridMinTargets && !e->P.rec && knob_int("grid_seed", 1)) {
            // Best effort: the seed only makes the launch below faster.  If the grid or its scratch cannot be had (e.g. out of
            // memory beside the split buffers) the launch simply runs un-seeded; and an engine whose planned search is brute
            // force gives the grid's memory back as soon as the seed is taken (one use per source cloud).
            hipError_t ge = hipSuccess;
            if (!e->grid_built) {
                ge = grid_build(e->stream, e->Qz, e->m, &e->grid);
                e->grid_built = ge == hipSuccess;
            }
            if (ge == hipSuccess && e->grid.ok) {
                if (!e->gscratch.list) ge = grid_scratch_alloc(e->stream, e->n_pad, &e->gscratch);
                if (ge == hipSuccess && !e->grid_out) ge = dmalloc(&e->grid_out, e->n_pad * sizeof(double2));
                if (ge == hipSuccess) {
                    e->gscratch.enabled = true;
                    ge = search_grid(e->stream, e->P, e->n, e->pc, e->grid, e->Qz, e->m, nullptr, nullptr, &e->gscratch, e->grid_out);
                }
                // idx <- the grid's correspondences (the partial sums this also writes are overwritten below)
                if (ge == hipSuccess)
                    ge = merge_cov(e->stream, e->P, e->n, e->n_pad, e->pc, e->Qz, e->m, 1, nullptr, nullptr, e->grid_out, false, e->idx,
                                   e->reject_d2, cov_slot);
                grid_seeded = ge == hipSuccess;
            }
            if (!wants_grid(e)) {   // explicit EXACT / FMA / F32 engine: nothing else will use these
                grid_free(&e->grid);            // waits for the stream the grid was built and searched on
                e->grid_built = false;
                grid_scratch_free(&e->gscratch);
                free_dev(e, e->grid_out);
                e->grid_out = nullptr;
            }
        }
        const bool seeded = seed || grid_seeded;
        GridFused fused;
        const int splits = grid ? 1 : (seeded ? e->q_splits : e->q_splits_first);
        if (!grid) RCCHK(ensure_split_buffers(e));
        if (grid) ++e->grid_launches; else ++e->bf_launches;
        if (grid) {
            // The overflow kernel (4 us when its queue is empty, which is the rule once a registration is under way) is
            // launched for the first grid searches of a source cloud, while the last queue length seen is not zero (a mapped
            // host word, read without waiting: possibly an iteration old), and every 16th search to refresh that word.
            // Without it a lane finishes even a large box by itself -- slower for an outlier, never wrong.
            e->gscratch.enabled = e->gscratch.defer_always || e->grid_since_source < 4 || *e->gscratch.host_mem > 0 ||
                                  e->grid_since_source % 16 == 0;
            ++e->grid_since_source;
            // e->rec describes the motion since the previous search only if exactly one update ran in between.
            // Once the registration has settled the same launch also forms the covariance sums (grid.hip: settled search).
            e->gscratch.issue_cert = e->grid_cert ? 1 : 0;
            fused.idx_out = e->idx;
            fused.reject_d2 = e->reject_d2;
            fused.partial = cov_slot;
            HIPCHK(search_grid(e->stream, e->P, e->n, e->pc, e->grid, e->Qz, e->m, seed ? e->idx : nullptr,
                               e->order_ok ? e->order : nullptr, &e->gscratch, e->grid_out,
                               (seed && e->updates_since_search == 1 && e->grid_cert) ? e->rec : nullptr,
                               fuse_centroid ? e->cen_part : nullptr, e->total_chunks, (double)e->n_global,
                               e->P.rec ? &fused : nullptr));
        } else if (e->variant == kF32 && e->mfma16) {
            if (!e->qrec16_ok) {   // the fp16 records of this target, on first use
                if (!e->qrec16) HIPCHK(dmalloc(&e->qrec16, 32 * e->m_pad));
                HIPCHK(build_qrec16(e->stream, e->Qz, e->m, e->m_pad, e->qmax, e->qrec16));
                e->qrec16_ok = true;
            }
            HIPCHK(search_mfma16(e->stream, e->mfma_g, e->P, e->n_pad, e->pc, e->Qz, e->qrec16, e->qmax, e->m, e->m_pad, splits,
                                 seeded ? e->idx : nullptr, e->keys, e->idxs));
        } else if (e->variant == kF32)
            HIPCHK(search_f32(e->stream, e->ppl, e->cb, e->P, e->n_pad, e->pc, e->Qz, e->qrec, e->qmax, e->m, e->m_pad, splits,
                              seeded ? e->idx : nullptr, e->keys, e->idxs));
        else
            HIPCHK(search(e->stream, e->variant, e->ppl, e->cb, e->P, e->n_pad, e->pc, e->Qz, e->m, e->m_pad, splits,
                          seeded ? e->idx : nullptr, e->keys, e->idxs));
        if (e->opt.kernel_timing) HIPCHK(hipEventRecord(b, e->stream));
        if (!fused.ran)
            HIPCHK(merge_cov(e->stream, e->P, e->n, e->n_pad, e->pc, e->Qz, e->m, splits, e->keys, e->idxs,
                             grid && !e->P.rec ? e->grid_out : nullptr, grid && e->P.rec, e->idx, e->reject_d2, cov_slot));
        if (strict) HIPCHK(strict_cov(e->stream, e->P, e->n, e->pc, e->Qz, e->m, e->idx, cov_slot, e->chunks_local));
        e->have_prev = true;
    }
    e->updates_since_search = 0;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_step_update(icp_amd_engine *e)
{
    if (!e || !e->have_source) return fail(ICP_AMD_ESTATE, "icp_amd_step_update: no source cloud");
    HIPCHK(set_device(e->opt.device));
    const bool rejecting = e->reject_d2 < std::numeric_limits<double>::infinity();
    double *cen_slot = e->cen_part + (size_t)e->opt.rank * e->cpr * kCenW;
    HIPCHK(solve(e->stream, e->cov_part, e->total_chunks, e->pc, e->qc, rejecting ? -1.0 : (double)e->n_global, e->rec,
                 e->rec + kSolveW));
    HIPCHK(apply(e->stream, e->P, e->n, e->rec, cen_slot));
    if (e->P.rec && e->n) e->soa_stale = true;   // only the records' A half was updated
    ++e->updates_since_search;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_iterate(icp_amd_engine *e, unsigned iters)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_iterate: NULL engine");
    if (e->opt.world != 1) return fail(ICP_AMD_ESTATE, "icp_amd_iterate needs world == 1; use the step_* calls when sharded");
    if (!e->have_source) return fail(ICP_AMD_ESTATE, "icp_amd_iterate: no source cloud");
    if (iters && !e->cen_ready) RCCHK(icp_amd_step_begin(e));
    for (unsigned it = 0; it < iters; ++it) {
        RCCHK(icp_amd_step_search(e));
        RCCHK(icp_amd_step_update(e));
    }
    return ICP_AMD_OK;
}

// ---- f-4: the hooks the reference stubs out (weight kernel / exclude_indices, src/cpu/icp.cpp:108-124; the unused
//      per-iteration norm, :213).  Off by default: results then equal the reference's.
extern "C" int icp_amd_set_reject_distance(icp_amd_engine *e, double dmax)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_set_reject_distance: NULL engine");
    if (dmax != dmax) return fail(ICP_AMD_EINVAL, "icp_amd_set_reject_distance: NaN");
    e->reject_d2 = (dmax > 0.0 && dmax < std::numeric_limits<double>::infinity()) ? dmax * dmax
                                                                                  : std::numeric_limits<double>::infinity();
    return ICP_AMD_OK;
}

extern "C" int icp_amd_get_last_motion(icp_amd_engine *e, double *pairs_used, double *delta_rot, double *delta_trans)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_get_last_motion: NULL engine");
    HIPCHK(set_device(e->opt.device));
    double h[3];
    HIPCHK(hipMemcpyAsync(h, e->rec + kSolveW, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (pairs_used) *pairs_used = h[0];
    if (delta_rot) *delta_rot = h[1];
    if (delta_trans) *delta_trans = h[2];
    return ICP_AMD_OK;
}

extern "C" int icp_amd_iterate_until(icp_amd_engine *e, unsigned max_iters, double rot_tol, double trans_tol,
                                     unsigned check_every, unsigned *iters_done)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_iterate_until: NULL engine");
    if (!(rot_tol >= 0.0) || !(trans_tol >= 0.0)) return fail(ICP_AMD_EINVAL, "icp_amd_iterate_until: tolerances must be >= 0");
    if (check_every == 0) check_every = 1;
    unsigned done = 0;
    int rc = ICP_AMD_OK;
    while (done < max_iters) {
        const unsigned step = (max_iters - done < check_every) ? max_iters - done : check_every;
        rc = icp_amd_iterate(e, step);
        if (rc != ICP_AMD_OK) break;
        done += step;
        double dr = 0.0, dt = 0.0;
        rc = icp_amd_get_last_motion(e, nullptr, &dr, &dt);   // the only host synchronisation of the loop
        if (rc != ICP_AMD_OK) break;
        if (dr <= rot_tol && dt <= trans_tol) break;
    }
    if (iters_done) *iters_done = done;
    return rc;
}

extern "C" int icp_amd_iterate_timed(icp_amd_engine *e, unsigned iters, double *elapsed_ms)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_iterate_timed: NULL engine");
    HIPCHK(set_device(e->opt.device));
    if (iters && !e->cen_ready && e->have_source) RCCHK(icp_amd_step_begin(e));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    int rc = ICP_AMD_OK;
    hipError_t er = hipEventRecord(a, e->stream);
    if (er == hipSuccess) rc = icp_amd_iterate(e, iters);
    if (er == hipSuccess && rc == ICP_AMD_OK) er = hipEventRecord(b, e->stream);
    if (er == hipSuccess && rc == ICP_AMD_OK) er = hipEventSynchronize(b);
    float ms = 0;
    if (er == hipSuccess && rc == ICP_AMD_OK) er = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    RCCHK(rc);
    HIPCHK(er);
    if (elapsed_ms) *elapsed_ms = ms;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_sync(icp_amd_engine *e)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_sync: NULL engine");
    HIPCHK(set_device(e->opt.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_get_source(icp_amd_engine *e, double *P_out)
{
    if (!e || (!P_out && e->n)) return fail(ICP_AMD_EINVAL, "icp_amd_get_source: NULL argument");
    if (!e->have_source) return fail(ICP_AMD_ESTATE, "icp_amd_get_source: no source cloud");
    HIPCHK(set_device(e->opt.device));
    if (e->n) {
        RCCHK(ensure_stage(e, e->n));
        RCCHK(sync_soa(e));
        if (e->sorted) HIPCHK(soa_scatter_to_aos(e->stream, e->P, e->order, e->n, e->stage));   // back into the caller's row order
        else HIPCHK(soa_to_aos(e->stream, e->P, e->n, e->stage));
    }
    RCCHK(copy_out(e, P_out, e->stage, e->n * 3 * sizeof(double)));   // synchronises the stream
    return ICP_AMD_OK;
}

extern "C" int icp_amd_get_last(icp_amd_engine *e, double *Pc, int32_t *idx, double *cov, double *S, double *R, double *t,
                                double *rms)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_get_last: NULL engine");
    HIPCHK(set_device(e->opt.device));
    double rec[kSolveW];
    HIPCHK(hipMemcpyAsync(rec, e->rec, sizeof rec, hipMemcpyDeviceToHost, e->stream));
    if (idx && e->n && e->sorted) {   // the indices of the stored rows, back into the caller's row order
        uint32_t *tmp = nullptr;
        HIPCHK(dmalloc(&tmp, e->n * sizeof(uint32_t)));
        hipError_t er = scatter_u32(e->stream, e->idx, e->order, e->n, tmp);
        int rc = er == hipSuccess ? copy_out(e, idx, tmp, e->n * sizeof(uint32_t)) : ICP_AMD_OK;
        free_dev(e, tmp);
        HIPCHK(er);
        RCCHK(rc);
    } else
        RCCHK(copy_out(e, idx, e->idx, idx ? e->n * sizeof(uint32_t) : 0));   // synchronises the stream
    if (cov) memcpy(cov, rec, 9 * sizeof(double));
    if (S) memcpy(S, rec + 9, 3 * sizeof(double));
    if (R) memcpy(R, rec + 12, 9 * sizeof(double));
    if (t) memcpy(t, rec + 21, 3 * sizeof(double));
    if (rms) *rms = rec[24];
    if (Pc) memcpy(Pc, rec + 25, 3 * sizeof(double));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_get_target_centroid(icp_amd_engine *e, double Qc[3])
{
    if (!e || !Qc || !e->have_target) return fail(ICP_AMD_ESTATE, "icp_amd_get_target_centroid: no target cloud");
    HIPCHK(set_device(e->opt.device));
    HIPCHK(hipMemcpyAsync(Qc, e->qc, 3 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_kernel_ms(icp_amd_engine *e, double *avg_ms, double *total_ms, unsigned *launches)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_kernel_ms: NULL engine");
    HIPCHK(set_device(e->opt.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    RCCHK(drain_events(e));
    if (avg_ms) *avg_ms = e->search_launches ? e->search_ms_total / e->search_launches : 0.0;
    if (total_ms) *total_ms = e->search_ms_total;
    if (launches) *launches = e->search_launches;
    e->search_ms_total = 0;
    e->search_launches = 0;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_search_geometry(icp_amd_engine *e, int *ppl, int *q_splits, int *workgroups, int *variant)
{
    if (!e || !e->have_source) return fail(ICP_AMD_ESTATE, "icp_amd_search_geometry: no source cloud");
    // source rows per workgroup: 256 lanes x ppl, or 128 x groups when the fp32 filter runs on the matrix pipe (two lanes per row)
    const size_t rows_wg = e->mfma_g ? (size_t)search_mfma_rows_per_workgroup(e->mfma_g) : (size_t)kWG * e->ppl;
    if (ppl) *ppl = e->use_grid ? 1 : (int)(rows_wg / kWG ? rows_wg / kWG : 1);
    if (q_splits) *q_splits = e->use_grid ? 1 : e->q_splits;
    if (workgroups) *workgroups = e->use_grid ? (int)((e->n + kWG - 1) / kWG) : (int)(e->n_pad / rows_wg) * e->q_splits;
    if (variant) *variant = e->use_grid ? (int)kGrid : e->variant;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_grid_stats(icp_amd_engine *e, unsigned *grid_launches, unsigned *brute_force_launches,
                                  unsigned *last_overflow)
{
    if (!e) return fail(ICP_AMD_EINVAL, "icp_amd_grid_stats: NULL engine");
    if (grid_launches) *grid_launches = e->grid_launches;
    if (brute_force_launches) *brute_force_launches = e->bf_launches;
    if (last_overflow) *last_overflow = e->gscratch.host_mem ? *e->gscratch.host_mem : 0u;
    return ICP_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// whole path
// ------------------------------------------------------------------------------------------------
extern "C" int icp_amd_run(const double *P, size_t n, const double *Q, size_t m, unsigned iters, int method, double *P_out,
                           icp_amd_trace *tr)
{
    if (!P || !Q || !P_out) return fail(ICP_AMD_EINVAL, "icp_amd_run: NULL argument");
    if (n == 0 || m == 0) return fail(ICP_AMD_EINVAL, "icp_amd_run: empty cloud (n=%zu, m=%zu)", n, m);
    if (method < ICP_AMD_METHOD_SHARED_LOOP || method > ICP_AMD_METHOD_BATCH)
        return fail(ICP_AMD_EINVAL, "Invalid method, must be: -shared|-loop|-shared-loop|-batch");
    icp_amd_options opt;
    icp_amd_default_options(&opt);
    opt.device = env_int("ICP_AMD_DEVICE", 0);
    icp_amd_engine *e = nullptr;
    RCCHK(icp_amd_create(&e, &opt));
    int rc = icp_amd_set_target(e, Q, m);
    if (rc == ICP_AMD_OK) rc = icp_amd_set_source(e, P, n);
    const bool tracing = tr && (tr->Pc || tr->idx || tr->cov || tr->S || tr->R || tr->t || tr->rms);
    if (rc == ICP_AMD_OK) {
        if (!tracing) {
            rc = icp_amd_iterate(e, iters);
        } else {
            for (unsigned it = 0; it < iters && rc == ICP_AMD_OK; ++it) {
                rc = icp_amd_iterate(e, 1);
                if (rc == ICP_AMD_OK)
                    rc = icp_amd_get_last(e, tr->Pc ? tr->Pc + 3 * it : nullptr, tr->idx ? tr->idx + (size_t)it * n : nullptr,
                                          tr->cov ? tr->cov + 9 * it : nullptr, tr->S ? tr->S + 3 * it : nullptr,
                                          tr->R ? tr->R + 9 * it : nullptr, tr->t ? tr->t + 3 * it : nullptr,
                                          tr->rms ? tr->rms + it : nullptr);
            }
        }
    }
    if (rc == ICP_AMD_OK) rc = icp_amd_get_source(e, P_out);
    icp_amd_destroy(e);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// whole path on several GPUs, single process, RCCL over xGMI
// ------------------------------------------------------------------------------------------------
namespace {
// RCCL is loaded at run time so that libicp_amd.so has no link-time dependency on it (single-GPU users never load it).
struct Rccl {
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *handle = nullptr;
    static constexpr int kDouble = 8;  // ncclDouble / ncclFloat64 (rccl.h)
    bool load()
    {
        if (handle) return true;
        handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!handle) handle = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!handle) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
        Broadcast = (decltype(Broadcast))dlsym(handle, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllGather && Broadcast && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;

struct MultiRun {
    std::vector<icp_amd_engine *> eng;
    std::vector<double *> cov, cen;
    std::vector<int> dev;
    std::vector<Rccl::comm_t> comm;
    bool by_copies = false;                 // exchange with peer copies instead of RCCL (a device listed twice; ICP_AMD_MULTI_EXCHANGE=copy)
    std::vector<hipEvent_t> ready, done;    // by_copies: per engine, "my slot is written" / "I have read everybody's slot"
    // RCCL's set-up and tear-down probe devices and peers and leave a stale 'invalid device ordinal' in the host thread's
    // sticky error word even when they succeed; the next HIP user of the process (torch, the host application) would report
    // it as its own failure.  The word was clean when this run began (`clean`): whatever is in it afterwards is the probing's.
    const bool clean = hipPeekAtLastError() == hipSuccess;
    void drop_probe_error() const { if (clean) (void)hipGetLastError(); }
    ~MultiRun()
    {
        for (icp_amd_engine *e : eng) icp_amd_destroy(e);
        for (size_t g = 0; g < dev.size(); ++g) {
            (void)hipSetDevice(dev[g]);
            if (g < cov.size() && cov[g]) (void)dfree(cov[g]);
            if (g < cen.size() && cen[g]) (void)dfree(cen[g]);
            if (g < ready.size() && ready[g]) (void)hipEventDestroy(ready[g]);
            if (g < done.size() && done[g]) (void)hipEventDestroy(done[g]);
        }
        for (Rccl::comm_t c : comm)
            if (c) g_rccl.CommDestroy(c);
        drop_probe_error();
    }
};

// The same exchange without RCCL: every engine copies the other engines' slots into its own buffer (peer copies on its
// own stream, ordered by events).  G (G - 1) small copies per exchange instead of one collective -- the way out when RCCL
// cannot form the communicator (one device listed more than once: several shards per GPU) or is not installed.
int all_gather_by_copies(MultiRun &r, std::vector<double *> &buf, size_t per)
{
    const int G = (int)r.eng.size();
    for (int g = 0; g < G; ++g) {
        HIPCHK(set_device(r.dev[g]));
        HIPCHK(hipEventRecord(r.ready[g], r.eng[g]->stream));
    }
    for (int g = 0; g < G; ++g) {
        HIPCHK(set_device(r.dev[g]));
        for (int h = 0; h < G; ++h) {
            if (h == g) continue;
            HIPCHK(hipStreamWaitEvent(r.eng[g]->stream, r.ready[h], 0));
            HIPCHK(hipMemcpyPeerAsync(buf[g] + (size_t)h * per, r.dev[g], buf[h] + (size_t)h * per, r.dev[h], per * sizeof(double),
                                      r.eng[g]->stream));
        }
        HIPCHK(hipEventRecord(r.done[g], r.eng[g]->stream));
    }
    // nobody overwrites its slot (the next iteration's partials) before every reader has copied it
    for (int h = 0; h < G; ++h) {
        HIPCHK(set_device(r.dev[h]));
        for (int g = 0; g < G; ++g)
            if (g != h) HIPCHK(hipStreamWaitEvent(r.eng[h]->stream, r.done[g], 0));
    }
    return ICP_AMD_OK;
}

// in-place all-gather of the rank-major partial array of every device (slot g of device g's buffer is its contribution)
int all_gather_partials(MultiRun &r, std::vector<double *> &buf, size_t per)
{
    if (r.by_copies) return all_gather_by_copies(r, buf, per);
    const int G = (int)r.eng.size();
    int rc = g_rccl.GroupStart();
    for (int g = 0; g < G && rc == 0; ++g)
        rc = g_rccl.AllGather(buf[g] + (size_t)g * per, buf[g], per, Rccl::kDouble, r.comm[g], r.eng[g]->stream);
    const int rc2 = g_rccl.GroupEnd();
    if (rc == 0) rc = rc2;
    if (rc != 0) return fail(ICP_AMD_EHIP, "RCCL all-gather failed: %s", g_rccl.GetErrorString(rc));
    return ICP_AMD_OK;
}
}  // namespace

extern "C" int icp_amd_run_multi(const double *P, size_t n, const double *Q, size_t m, unsigned iters, int n_gpus, const int *devices,
                                 double *P_out, icp_amd_trace *tr)
{
    if (!P || !Q || !P_out) return fail(ICP_AMD_EINVAL, "icp_amd_run_multi: NULL argument");
    if (n == 0 || m == 0) return fail(ICP_AMD_EINVAL, "icp_amd_run_multi: empty cloud (n=%zu, m=%zu)", n, m);
    const int avail = icp_amd_device_count();
    if (avail <= 0) return fail(ICP_AMD_ENODEV, "no HIP device available; libicp_amd has no CPU fallback");
    // an explicit device list may name a device more than once (several shards per GPU); without a list: one shard per GPU
    if (n_gpus < 1 || n_gpus > 1024 || (!devices && n_gpus > avail))
        return fail(ICP_AMD_EINVAL, "icp_amd_run_multi: n_gpus=%d but %d device(s) are visible", n_gpus, avail);
    const int G = n_gpus;
    if (G == 1 && !devices) return icp_amd_run(P, n, Q, m, iters, ICP_AMD_METHOD_SHARED_LOOP, P_out, tr);
    MultiRun r;
    for (int g = 0; g < G; ++g) r.dev.push_back(devices ? devices[g] : g);
    for (int g = 0; g < G; ++g) {
        if (r.dev[g] < 0 || r.dev[g] >= avail) return fail(ICP_AMD_EINVAL, "icp_amd_run_multi: device %d out of range [0,%d)", r.dev[g], avail);
        for (int h = 0; h < g; ++h) r.by_copies |= r.dev[h] == r.dev[g];
    }
    if (const char *x = getenv("ICP_AMD_MULTI_EXCHANGE")) r.by_copies |= !strcmp(x, "copy");
    if (!r.by_copies && !g_rccl.load()) {
        const char *why = dlerror();   // one call: the message is cleared by reading it
        return fail(ICP_AMD_EHIP, "icp_amd_run_multi: cannot load librccl.so (%s); ICP_AMD_MULTI_EXCHANGE=copy exchanges with peer copies instead",
                    why ? why : "missing symbols");
    }
    const size_t cpr = icp_amd_chunks_per_rank(n, G), rpr = cpr * kChunk;
    r.cov.assign(G, nullptr);
    r.cen.assign(G, nullptr);
    r.comm.assign(G, nullptr);
    for (int g = 0; g < G; ++g) {
        RCCHK(require_device(r.dev[g]));
        HIPCHK(dmalloc(&r.cov[g], (size_t)G * cpr * kCovW * sizeof(double)));
        HIPCHK(dmalloc(&r.cen[g], (size_t)G * cpr * kCenW * sizeof(double)));
        icp_amd_options opt;
        icp_amd_default_options(&opt);
        opt.device = r.dev[g];
        opt.rank = g;
        opt.world = G;
        opt.n_global = n;
        opt.cov_gather = r.cov[g];
        opt.cen_gather = r.cen[g];
        icp_amd_engine *e = nullptr;
        RCCHK(icp_amd_create(&e, &opt));
        r.eng.push_back(e);
        // on the engine's own stream: ordered before everything that engine (and the exchange on its stream) does with them
        HIPCHK(hipMemsetAsync(r.cov[g], 0, (size_t)G * cpr * kCovW * sizeof(double), e->stream));
        HIPCHK(hipMemsetAsync(r.cen[g], 0, (size_t)G * cpr * kCenW * sizeof(double), e->stream));
    }
    if (r.by_copies) {
        r.ready.assign(G, nullptr);
        r.done.assign(G, nullptr);
        for (int g = 0; g < G; ++g) {
            HIPCHK(set_device(r.dev[g]));
            HIPCHK(hipEventCreateWithFlags(&r.ready[g], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&r.done[g], hipEventDisableTiming));
            for (int h = 0; h < G; ++h) {   // peer access where the copies cross devices (already enabled is fine)
                if (r.dev[h] == r.dev[g]) continue;
                int can = 0;
                HIPCHK(hipDeviceCanAccessPeer(&can, r.dev[g], r.dev[h]));
                if (can && hipDeviceEnablePeerAccess(r.dev[h], 0) == hipErrorPeerAccessAlreadyEnabled) r.drop_probe_error();
            }
        }
    } else {
        const int rc = g_rccl.CommInitAll(r.comm.data(), G, r.dev.data());
        if (rc != 0) return fail(ICP_AMD_EHIP, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(rc));
        r.drop_probe_error();
    }
    std::vector<size_t> lo(G), cnt(G);
    for (int g = 0; g < G; ++g) {
        lo[g] = (size_t)g * rpr < n ? (size_t)g * rpr : n;
        cnt[g] = n - lo[g] < rpr ? n - lo[g] : rpr;
    }
    {   // the per-device set-up (uploads, centring, records, grid build: milliseconds each, with host waits) runs on one
        // host thread per device; the error text is thread-local, so it is carried back by hand
        std::vector<int> rcs(G, ICP_AMD_OK);
        std::vector<std::string> errs(G);
        std::vector<std::thread> pool;
        auto setup = [&](int g) {
            int rc = icp_amd_set_target(r.eng[g], Q, m);
            if (rc == ICP_AMD_OK) rc = icp_amd_set_source(r.eng[g], P + lo[g] * 3, cnt[g]);
            rcs[g] = rc;
            if (rc != ICP_AMD_OK) errs[g] = icp_amd_last_error();
        };
        for (int g = 1; g < G; ++g) pool.emplace_back(setup, g);
        setup(0);
        for (std::thread &t : pool) t.join();
        for (int g = 0; g < G; ++g)
            if (rcs[g] != ICP_AMD_OK) return fail(rcs[g], "%s", errs[g].c_str());
    }
    for (int g = 0; g < G; ++g) RCCHK(icp_amd_step_begin(r.eng[g]));
    RCCHK(all_gather_partials(r, r.cen, cpr * kCenW));
    const bool tracing = tr && (tr->Pc || tr->idx || tr->cov || tr->S || tr->R || tr->t || tr->rms);
    for (unsigned it = 0; it < iters; ++it) {
        for (int g = 0; g < G; ++g) RCCHK(icp_amd_step_search(r.eng[g]));
        RCCHK(all_gather_partials(r, r.cov, cpr * kCovW));
        for (int g = 0; g < G; ++g) RCCHK(icp_amd_step_update(r.eng[g]));
        RCCHK(all_gather_partials(r, r.cen, cpr * kCenW));
        if (tracing) {
            RCCHK(icp_amd_get_last(r.eng[0], tr->Pc ? tr->Pc + 3 * it : nullptr, nullptr, tr->cov ? tr->cov + 9 * it : nullptr,
                                   tr->S ? tr->S + 3 * it : nullptr, tr->R ? tr->R + 9 * it : nullptr, tr->t ? tr->t + 3 * it : nullptr,
                                   tr->rms ? tr->rms + it : nullptr));
            if (tr->idx)
                for (int g = 0; g < G; ++g)
                    if (cnt[g])
                        RCCHK(icp_amd_get_last(r.eng[g], nullptr, tr->idx + (size_t)it * n + lo[g], nullptr, nullptr, nullptr, nullptr, nullptr));
        }
    }
    if (iters && knob_int("multi_crosscheck", 0)) {
        // Debug cross-check: every device solved R, t redundantly from the same gathered partials, so no
        // broadcast is needed -- but rank 0's R || t, broadcast (ncclBroadcast, or a peer copy on the RCCL-free exchange),
        // must then equal every device's own, bit for bit.  A mismatch means the exchange delivered different partials.
        std::vector<double *> chk(G, nullptr);
        struct Free { std::vector<double *> &v; std::vector<int> &dev; ~Free() { for (size_t g = 0; g < v.size(); ++g) if (v[g]) { (void)hipSetDevice(dev[g]); (void)dfree(v[g]); } } } fr{chk, r.dev};
        for (int g = 0; g < G; ++g) {
            HIPCHK(set_device(r.dev[g]));
            HIPCHK(dmalloc(&chk[g], 12 * sizeof(double)));
        }
        if (r.by_copies) {
            HIPCHK(set_device(r.dev[0]));
            HIPCHK(hipStreamSynchronize(r.eng[0]->stream));
            for (int g = 0; g < G; ++g) {
                HIPCHK(set_device(r.dev[g]));
                HIPCHK(hipMemcpyPeerAsync(chk[g], r.dev[g], r.eng[0]->rec + 12, r.dev[0], 12 * sizeof(double), r.eng[g]->stream));
            }
        } else {
            int rc = g_rccl.GroupStart();
            for (int g = 0; g < G && rc == 0; ++g)
                // sendbuff is only read on the root; every rank still passes a pointer of its OWN device (pointer checks)
                rc = g_rccl.Broadcast(g == 0 ? r.eng[0]->rec + 12 : chk[g], chk[g], 12, Rccl::kDouble, 0, r.comm[g], r.eng[g]->stream);
            const int rc2 = g_rccl.GroupEnd();
            if (rc == 0) rc = rc2;
            if (rc != 0) return fail(ICP_AMD_EHIP, "RCCL broadcast failed: %s", g_rccl.GetErrorString(rc));
        }
        for (int g = 0; g < G; ++g) {
            double want[12], mine[12];
            HIPCHK(set_device(r.dev[g]));
            HIPCHK(hipMemcpyAsync(want, chk[g], sizeof want, hipMemcpyDeviceToHost, r.eng[g]->stream));
            HIPCHK(hipMemcpyAsync(mine, r.eng[g]->rec + 12, sizeof mine, hipMemcpyDeviceToHost, r.eng[g]->stream));
            HIPCHK(hipStreamSynchronize(r.eng[g]->stream));
            if (memcmp(want, mine, sizeof want) != 0)
                return fail(ICP_AMD_EHIP, "icp_amd_run_multi: cross-check failed, shard %d (device %d) solved a different R, t than shard 0", g, r.dev[g]);
        }
    }
    for (int g = 0; g < G; ++g)
        if (cnt[g]) RCCHK(icp_amd_get_source(r.eng[g], P_out + lo[g] * 3));
        else RCCHK(icp_amd_sync(r.eng[g]));
    return ICP_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-registration mode
// ------------------------------------------------------------------------------------------------
struct icp_amd_batch {
    int device = 0;
    size_t batch = 0, n = 0, m = 0;
    hipStream_t stream = nullptr;
    double *P_in = nullptr, *Q_in = nullptr, *P_out = nullptr, *rec = nullptr;
    uint32_t *idx = nullptr;
    bool uploaded = false, ran = false;
};

extern "C" int icp_amd_batch_destroy(icp_amd_batch *b)
{
    if (!b) return ICP_AMD_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void *p : {(void *)b->P_in, (void *)b->Q_in, (void *)b->P_out, (void *)b->rec, (void *)b->idx})
        if (p) (void)(b->stream ? dfree_on(p, b->stream) : dfree(p));   // only this object's stream has touched them
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_batch_create(icp_amd_batch **out, int device, size_t batch, size_t n, size_t m)
{
    if (!out) return fail(ICP_AMD_EINVAL, "icp_amd_batch_create: out is NULL");
    *out = nullptr;
    if (batch == 0 || n == 0 || m == 0) return fail(ICP_AMD_EINVAL, "icp_amd_batch_create: empty batch or cloud (batch=%zu n=%zu m=%zu)", batch, n, m);
    if (n > kBatchMaxN || m > kBatchMaxM || batch > 0x7FFFFFFFu)
        return fail(ICP_AMD_EINVAL, "batch mode holds one registration per workgroup: n <= %zu, m <= %zu (got n=%zu m=%zu); use icp_amd_run for larger clouds",
                    kBatchMaxN, kBatchMaxM, n, m);
    RCCHK(require_device(device));
    icp_amd_batch *b = new (std::nothrow) icp_amd_batch();
    if (!b) return fail(ICP_AMD_ENOMEM, "out of host memory");
    b->device = device; b->batch = batch; b->n = n; b->m = m;
    hipError_t er = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (er == hipSuccess) er = dmalloc(&b->P_in, batch * n * 3 * sizeof(double));
    if (er == hipSuccess) er = dmalloc(&b->Q_in, batch * m * 3 * sizeof(double));
    if (er == hipSuccess) er = dmalloc(&b->P_out, batch * n * 3 * sizeof(double));
    if (er == hipSuccess) er = dmalloc(&b->rec, batch * kSolveW * sizeof(double));
    if (er == hipSuccess) er = dmalloc(&b->idx, batch * n * sizeof(uint32_t));
    if (er != hipSuccess) {
        icp_amd_batch_destroy(b);
        return fail(er == hipErrorOutOfMemory ? ICP_AMD_ENOMEM : ICP_AMD_EHIP, "icp_amd_batch_create: %s", hipGetErrorString(er));
    }
    *out = b;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_batch_upload(icp_amd_batch *b, const double *P, const double *Q)
{
    if (!b || !P || !Q) return fail(ICP_AMD_EINVAL, "icp_amd_batch_upload: NULL argument");
    HIPCHK(set_device(b->device));
    HIPCHK(hipMemcpyAsync(b->P_in, P, b->batch * b->n * 3 * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->Q_in, Q, b->batch * b->m * 3 * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->uploaded = true;
    b->ran = false;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_batch_run(icp_amd_batch *b, unsigned iters, int search_variant, double *elapsed_ms)
{
    if (!b) return fail(ICP_AMD_EINVAL, "icp_amd_batch_run: NULL handle");
    if (!b->uploaded) return fail(ICP_AMD_ESTATE, "icp_amd_batch_run: upload the clouds first");
    HIPCHK(set_device(b->device));
    // EXACT / FMA: the difference form on SoA targets; anything else (AUTO, F32, GRID): the expanded form on records
    // default: the expanded form, from 384 targets on seeded from the second iteration on (below, the seed blocks cost what the
    // shorter fast path saves: 256 x 256 points 10.2 ms unseeded against 11.1; ICP_AMD_DEBUG batch_seeded=0 / 1 decides for every size)
    int variant = search_variant == ICP_AMD_SEARCH_EXACT ? kExact : search_variant == ICP_AMD_SEARCH_FMA ? kFma
                  : knob_int("batch_seeded", b->m >= 384 ? 1 : 0) ? kSeeded : kExpanded;
    const char *sv = search_variant == ICP_AMD_SEARCH_AUTO ? getenv("ICP_AMD_SEARCH") : nullptr;   // an explicit variant wins
    if (sv && !strcmp(sv, "fma")) variant = kFma;
    if (sv && !strcmp(sv, "exact")) variant = kExact;
    hipEvent_t a = nullptr, z = nullptr;
    if (elapsed_ms) {
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&z));
        HIPCHK(hipEventRecord(a, b->stream));
    }
    icp_amd_options none{};
    const bool strict = strict_requested(none);   // ICP_AMD_STRICT_ORDER=1: sequential centroid / covariance sums (kernels.hip)
    hipError_t er = batch_icp(b->stream, variant, strict, b->P_in, b->Q_in, b->batch, b->n, b->m, iters, b->P_out, b->rec, b->idx);
    if (er == hipSuccess && elapsed_ms) {
        er = hipEventRecord(z, b->stream);
        if (er == hipSuccess) er = hipEventSynchronize(z);
        float ms = 0;
        if (er == hipSuccess) er = hipEventElapsedTime(&ms, a, z);
        *elapsed_ms = ms;
    }
    if (a) (void)hipEventDestroy(a);
    if (z) (void)hipEventDestroy(z);
    HIPCHK(er);
    b->ran = true;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_batch_download(icp_amd_batch *b, double *P_out, double *R, double *t, double *cov, int32_t *idx, double *rms)
{
    if (!b) return fail(ICP_AMD_EINVAL, "icp_amd_batch_download: NULL handle");
    if (!b->ran) return fail(ICP_AMD_ESTATE, "icp_amd_batch_download: nothing has run");
    HIPCHK(set_device(b->device));
    std::vector<double> rec;
    if (P_out) HIPCHK(hipMemcpyAsync(P_out, b->P_out, b->batch * b->n * 3 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (idx) HIPCHK(hipMemcpyAsync(idx, b->idx, b->batch * b->n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    if (R || t || cov || rms) {
        rec.resize(b->batch * kSolveW);
        HIPCHK(hipMemcpyAsync(rec.data(), b->rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    for (size_t k = 0; k < b->batch && !rec.empty(); ++k) {
        const double *r = rec.data() + k * kSolveW;
        if (cov) memcpy(cov + 9 * k, r, 9 * sizeof(double));
        if (R) memcpy(R + 9 * k, r + 12, 9 * sizeof(double));
        if (t) memcpy(t + 3 * k, r + 21, 3 * sizeof(double));
        if (rms) rms[k] = r[24];
    }
    return ICP_AMD_OK;
}

extern "C" int icp_amd_run_batch(const double *P, const double *Q, size_t batch, size_t n, size_t m, unsigned iters, double *P_out,
                                 double *R, double *t, int32_t *idx, double *rms)
{
    if (!P || !Q || !P_out) return fail(ICP_AMD_EINVAL, "icp_amd_run_batch: NULL argument");
    if (batch == 0 || n == 0 || m == 0) return fail(ICP_AMD_EINVAL, "icp_amd_run_batch: empty batch or cloud (batch=%zu n=%zu m=%zu)", batch, n, m);
    if (n > kBatchMaxN || m > kBatchMaxM) {
        // Clouds too large for the one-workgroup-per-registration kernel: the registrations go through ordinary engines
        // instead, a few at a time (one host thread, engine and stream each; the device-memory cache makes their set-up
        // cheap), so that small launches of different registrations overlap on the device.  Same results as icp_amd_run.
        RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
        const size_t workers = std::min<size_t>(batch, (size_t)std::max(1, std::min(env_int("ICP_AMD_BATCH_WORKERS", 8), 64)));
        std::vector<int> rcs(workers, ICP_AMD_OK);
        std::vector<std::string> errs(workers);
        auto work = [&](size_t w) {
            icp_amd_options opt;
            icp_amd_default_options(&opt);
            opt.device = env_int("ICP_AMD_DEVICE", 0);
            icp_amd_engine *e = nullptr;
            int rc = icp_amd_create(&e, &opt);
            for (size_t k = w; k < batch && rc == ICP_AMD_OK; k += workers) {
                rc = icp_amd_set_target(e, Q + k * m * 3, m);
                if (rc == ICP_AMD_OK) rc = icp_amd_set_source(e, P + k * n * 3, n);
                if (rc == ICP_AMD_OK) rc = icp_amd_iterate(e, iters);
                if (rc == ICP_AMD_OK) rc = icp_amd_get_source(e, P_out + k * n * 3);
                if (rc == ICP_AMD_OK && iters && (R || t || idx || rms))
                    rc = icp_amd_get_last(e, nullptr, idx ? idx + k * n : nullptr, nullptr, nullptr, R ? R + 9 * k : nullptr,
                                          t ? t + 3 * k : nullptr, rms ? rms + k : nullptr);
            }
            if (rc != ICP_AMD_OK) errs[w] = icp_amd_last_error();
            rcs[w] = rc;
            icp_amd_destroy(e);
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < workers; ++w) pool.emplace_back(work, w);
        work(0);
        for (std::thread &th : pool) th.join();
        for (size_t w = 0; w < workers; ++w)
            if (rcs[w] != ICP_AMD_OK) return fail(rcs[w], "%s", errs[w].c_str());
        return ICP_AMD_OK;
    }
    icp_amd_batch *b = nullptr;
    RCCHK(icp_amd_batch_create(&b, env_int("ICP_AMD_DEVICE", 0), batch, n, m));
    int rc = icp_amd_batch_upload(b, P, Q);
    if (rc == ICP_AMD_OK) rc = icp_amd_batch_run(b, iters, ICP_AMD_SEARCH_AUTO, nullptr);
    if (rc == ICP_AMD_OK) rc = icp_amd_batch_download(b, P_out, R, t, nullptr, idx, rms);
    icp_amd_batch_destroy(b);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// stage-level entry points (parity tests)
// ------------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)dfree(p); }
    hipError_t alloc(size_t bytes) { return dmalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return (T *)p; }
};

// upload a row-major host cloud into a padded SoA device buffer
int upload_soa(hipStream_t s, const double *host, size_t n, size_t n_pad, double pad, DevBuf &store, DevBuf &stage, Soa3 &out)
{
    HIPCHK(store.alloc(3 * n_pad * sizeof(double)));
    HIPCHK(stage.alloc(n * 3 * sizeof(double)));
    if (n) HIPCHK(hipMemcpyAsync(stage.p, host, n * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    out = Soa3{store.as<double>(), store.as<double>() + n_pad, store.as<double>() + 2 * n_pad};
    HIPCHK(aos_to_soa(s, stage.as<double>(), n, n_pad, nullptr, pad, out));
    return ICP_AMD_OK;
}
}  // namespace

static int correspondences_impl(const double *P, size_t n, const double Pc[3], const double *Qz, size_t m, int search_variant,
                                int q_splits, const int32_t *seed, int32_t *idx);

extern "C" int icp_amd_correspondences(const double *P, size_t n, const double Pc[3], const double *Qz, size_t m, int search_variant,
                                       int q_splits, int32_t *idx)
{
    return correspondences_impl(P, n, Pc, Qz, m, search_variant, q_splits, nullptr, idx);
}

// The same call as a SEEDED launch: seed[i] = a candidate correspondence of row i (what an iteration of the loop hands to the
// next one, src/gpu/icp.cu:548-628 keeps no such state); out-of-range entries mean "none".  Seeds change no result -- they
// only start every row's running minimum low -- but they select the kernels' seeded code paths, which is what the
// iterations after the first run.
extern "C" int icp_amd_correspondences_seeded(const double *P, size_t n, const double Pc[3], const double *Qz, size_t m,
                                              int search_variant, int q_splits, const int32_t *seed, int32_t *idx)
{
    if (!seed) return fail(ICP_AMD_EINVAL, "icp_amd_correspondences_seeded: NULL seed");
    return correspondences_impl(P, n, Pc, Qz, m, search_variant, q_splits, seed, idx);
}

static int correspondences_impl(const double *P, size_t n, const double Pc[3], const double *Qz, size_t m, int search_variant,
                                int q_splits, const int32_t *seed, int32_t *idx)
{
    if (!P || !Pc || !Qz || !idx) return fail(ICP_AMD_EINVAL, "icp_amd_correspondences: NULL argument");
    if (n == 0 || m == 0 || n > kMaxRows || m > kMaxRows) return fail(ICP_AMD_EINVAL, "icp_amd_correspondences: bad sizes");
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    hipStream_t s = nullptr;
    const size_t n_pad = round_up(n, kPadRows), m_pad = round_up(m, kPadRows);
    DevBuf pst, pstage, qst, qstage, pc, keys, idxs, out, part, qrec, qmaxb, grec;
    Soa3 dP, dQ;
    RCCHK(upload_soa(s, P, n, n_pad, 0.0, pst, pstage, dP));
    RCCHK(upload_soa(s, Qz, m, m_pad, std::numeric_limits<double>::infinity(), qst, qstage, dQ));
    HIPCHK(pc.alloc(3 * sizeof(double)));
    HIPCHK(hipMemcpyAsync(pc.p, Pc, 3 * sizeof(double), hipMemcpyHostToDevice, s));
    DevBuf seeds;
    const uint32_t *prev = nullptr;
    if (seed) {   // n_pad entries, "none" beyond the n rows
        HIPCHK(seeds.alloc(n_pad * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(seeds.p, 0xFF, n_pad * sizeof(uint32_t), s));
        HIPCHK(hipMemcpyAsync(seeds.p, seed, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        prev = seeds.as<uint32_t>();
    }
    const bool f32 = search_variant == ICP_AMD_SEARCH_F32;
    const size_t max_splits = search_scan_rows(m) / kSplitGran;
    if (q_splits < 1) q_splits = 1;
    if ((size_t)q_splits > max_splits) q_splits = (int)max_splits;
    int ppl = knob_int("ppl", 4);
    if (ppl != 1 && ppl != 2 && ppl != 4 && ppl != 8) ppl = 4;
    if (f32 && ppl == 1) ppl = 2;
    if (!f32 && ppl == 8) ppl = 4;
    HIPCHK(keys.alloc((size_t)q_splits * n_pad * sizeof(double)));
    HIPCHK(idxs.alloc((size_t)q_splits * n_pad * sizeof(uint32_t)));
    HIPCHK(out.alloc(n_pad * sizeof(uint32_t)));
    HIPCHK(part.alloc((n_pad / kChunk) * kCovW * sizeof(double)));
    const int variant = search_variant == ICP_AMD_SEARCH_FMA ? kFma : kExact;
    int cb = knob_int("cb", 32);
    if (cb != 8 && cb != 16 && cb != 32) cb = 32;
    struct GridHolder { Grid g; ~GridHolder() { grid_free(&g); } } gh;
    bool grid = search_variant == ICP_AMD_SEARCH_GRID;
    if (grid) {
        HIPCHK(grid_build(s, dQ, m, &gh.g));
        grid = gh.g.ok;   // non-finite target coordinates: brute force below, like the engine
    }
    if (!grid && !prev && search_variant != ICP_AMD_SEARCH_GRID && m >= kAutoGridMinTargets && knob_int("grid_seed", 1)) {
        // like the engine's first iteration: the grid's exact answer seeds the brute-force scan (which still evaluates
        // every pair); targets the grid cannot file (non-finite coordinates) leave the launch un-seeded
        // best effort, like the engine: a seed that cannot be had (no memory for the grid) leaves the launch un-seeded
        hipError_t ge = grid_build(s, dQ, m, &gh.g);
        if (ge == hipSuccess && gh.g.ok) {
            struct ScratchHolder { GridScratch sc; ~ScratchHolder() { grid_scratch_free(&sc); } } sh;
            ge = grid_scratch_alloc(s, n, &sh.sc);
            if (ge == hipSuccess) ge = grec.alloc(n_pad * sizeof(double2));
            if (ge == hipSuccess) ge = seeds.alloc(n_pad * sizeof(uint32_t));
            if (ge == hipSuccess) ge = hipMemsetAsync(seeds.p, 0xFF, n_pad * sizeof(uint32_t), s);
            if (ge == hipSuccess) ge = search_grid(s, dP, n, pc.as<double>(), gh.g, dQ, m, nullptr, nullptr, &sh.sc, grec.as<double2>());
            if (ge == hipSuccess)
                ge = merge_cov(s, dP, n, n_pad, pc.as<double>(), dQ, m, 1, nullptr, nullptr, grec.as<double2>(), false, seeds.as<uint32_t>(),
                               std::numeric_limits<double>::infinity(), part.as<double>());
            HIPCHK(hipStreamSynchronize(s));   // the scratch is released on scope exit
            if (ge == hipSuccess) prev = seeds.as<uint32_t>();
        }
    }
    if (grid) {
        q_splits = 1;
        struct ScratchHolder { GridScratch sc; ~ScratchHolder() { grid_scratch_free(&sc); } } sh;
        HIPCHK(grid_scratch_alloc(s, n, &sh.sc));
        HIPCHK(grec.alloc(n_pad * sizeof(double2)));
        HIPCHK(search_grid(s, dP, n, pc.as<double>(), gh.g, dQ, m, prev, nullptr, &sh.sc, grec.as<double2>()));
        HIPCHK(hipStreamSynchronize(s));   // the scratch is released on scope exit
    } else if (f32) {
        if (cb == 8) cb = 16;
        HIPCHK(qrec.alloc(16 * m_pad));
        HIPCHK(qmaxb.alloc(sizeof(unsigned long long)));
        HIPCHK(build_qrec(s, dQ, m, m_pad, qrec.p, qmaxb.as<unsigned long long>()));
        unsigned long long bits = 0;
        HIPCHK(hipMemcpyAsync(&bits, qmaxb.p, sizeof bits, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        double qmax2;
        memcpy(&qmax2, &bits, sizeof qmax2);
        if ((qmax2 == qmax2) && qmax2 > 1e-30 && qmax2 < 1e30) {
            if (mfma_fp16()) {
                DevBuf rec16;
                HIPCHK(rec16.alloc(32 * m_pad));
                HIPCHK(build_qrec16(s, dQ, m, m_pad, sqrt(qmax2), rec16.p));
                HIPCHK(search_mfma16(s, mfma_groups(n_pad), dP, n_pad, pc.as<double>(), dQ, rec16.p, sqrt(qmax2), m, m_pad, q_splits, prev,
                                     keys.as<double>(), idxs.as<uint32_t>()));
                HIPCHK(hipStreamSynchronize(s));   // rec16 is released on scope exit
            } else
                HIPCHK(search_f32(s, ppl, cb, dP, n_pad, pc.as<double>(), dQ, qrec.p, sqrt(qmax2), m, m_pad, q_splits, prev,
                                  keys.as<double>(), idxs.as<uint32_t>()));
        } else {   // range outside the fp32 bound's domain: same degradation as the engine
            if (ppl == 8) ppl = 4;
            HIPCHK(search(s, kFma, ppl, cb, dP, n_pad, pc.as<double>(), dQ, m, m_pad, q_splits, prev, keys.as<double>(),
                          idxs.as<uint32_t>()));
        }
    } else {
        HIPCHK(search(s, variant, ppl, cb, dP, n_pad, pc.as<double>(), dQ, m, m_pad, q_splits, prev, keys.as<double>(),
                      idxs.as<uint32_t>()));
    }
    HIPCHK(merge_cov(s, dP, n, n_pad, pc.as<double>(), dQ, m, q_splits, keys.as<double>(), idxs.as<uint32_t>(),
                     grid ? grec.as<double2>() : nullptr, false, out.as<uint32_t>(), std::numeric_limits<double>::infinity(),
                     part.as<double>()));
    HIPCHK(hipMemcpyAsync(idx, out.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_cross_cov(const double *P, size_t n, const double Pc[3], const double *Qz, size_t m, const int32_t *idx,
                                 double cov[9])
{
    if (!P || !Pc || !Qz || !idx || !cov) return fail(ICP_AMD_EINVAL, "icp_amd_cross_cov: NULL argument");
    if (n == 0 || m == 0 || n > kMaxRows || m > kMaxRows) return fail(ICP_AMD_EINVAL, "icp_amd_cross_cov: bad sizes");
    for (size_t i = 0; i < n; ++i)
        if (idx[i] < 0 || (size_t)idx[i] >= m) return fail(ICP_AMD_EINVAL, "icp_amd_cross_cov: idx[%zu]=%d out of range", i, idx[i]);
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    hipStream_t s = nullptr;
    const size_t n_pad = round_up(n, kChunk), m_pad = round_up(m, kTileQ), chunks = n_pad / kChunk;
    DevBuf pst, pstage, qst, qstage, pc, qc, di, part, rec;
    Soa3 dP, dQ;
    RCCHK(upload_soa(s, P, n, n_pad, 0.0, pst, pstage, dP));
    RCCHK(upload_soa(s, Qz, m, m_pad, 0.0, qst, qstage, dQ));
    HIPCHK(pc.alloc(3 * sizeof(double)));
    HIPCHK(qc.alloc(3 * sizeof(double)));
    HIPCHK(hipMemcpyAsync(pc.p, Pc, 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(qc.p, 0, 3 * sizeof(double), s));
    HIPCHK(di.alloc(n * sizeof(uint32_t)));
    HIPCHK(hipMemcpyAsync(di.p, idx, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(part.alloc(chunks * kCovW * sizeof(double)));
    HIPCHK(rec.alloc(kSolveW * sizeof(double)));
    HIPCHK(cov_from_idx(s, dP, n, pc.as<double>(), dQ, m, di.as<uint32_t>(), part.as<double>()));
    icp_amd_options none{};
    if (strict_requested(none) && n <= kStrictMaxSource)   // ICP_AMD_STRICT_ORDER=1: the reference's sequential sum
        HIPCHK(strict_cov(s, dP, n, pc.as<double>(), dQ, m, di.as<uint32_t>(), part.as<double>(), chunks));
    HIPCHK(solve(s, part.as<double>(), chunks, pc.as<double>(), qc.as<double>(), (double)n, rec.as<double>(), nullptr));
    HIPCHK(hipMemcpyAsync(cov, rec.p, 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_centroid(const double *A, size_t n, double out[3])
{
    if (!A || !out) return fail(ICP_AMD_EINVAL, "icp_amd_centroid: NULL argument");
    if (n == 0) return fail(ICP_AMD_EINVAL, "icp_amd_centroid: empty cloud");
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    hipStream_t s = nullptr;
    const size_t chunks = (n + kChunk - 1) / kChunk;
    DevBuf a, part, o;
    HIPCHK(a.alloc(n * 3 * sizeof(double)));
    HIPCHK(part.alloc(chunks * kCenW * sizeof(double)));
    HIPCHK(o.alloc(3 * sizeof(double)));
    HIPCHK(hipMemcpyAsync(a.p, A, n * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    icp_amd_options none{};
    if (strict_requested(none) && n <= kStrictMaxTarget) {   // ICP_AMD_STRICT_ORDER=1: the reference's sequential sum
        HIPCHK(strict_centroid(s, a.as<double>(), Soa3{nullptr, nullptr, nullptr}, n, o.as<double>()));
    } else {
        HIPCHK(colsum_partials(s, a.as<double>(), Soa3{nullptr, nullptr, nullptr}, n, part.as<double>()));
        HIPCHK(finalize_centroid(s, part.as<double>(), chunks, (double)n, o.as<double>()));
    }
    HIPCHK(hipMemcpyAsync(out, o.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ICP_AMD_OK;
}

// ---- the entry point that allocates nothing (include/icp_amd.h): icp_amd_centroid on device-resident rows, with the chunk
//      partials -- the scratch icp_amd_centroid allocates per call -- in a file-scope device array.  Every GPU the library
//      runs on holds its own copy of the array until the process ends; its address on a device comes from
//      hipGetSymbolAddress (with that device current), once per device, and only that address is ever handed to a kernel
//      (the array's name is a host-side stand-in in host code, not a device address).
static const size_t kStaticCentroidMaxRows = (size_t)16 << 20;                        // 16 Mi rows per call
static const size_t kStaticScratchDoubles = kStaticCentroidMaxRows / kChunk * kCenW;  // 16384 chunks x 3 sums = 384 KiB
static __device__ __align__(256) double scratch[kStaticScratchDoubles];
static const int kMaxDevices = 64;
static double *g_scratch_ptr[kMaxDevices];   // per device, set once under g_scratch_mu
static std::mutex g_scratch_mu;

static int static_scratch(int device, double **out)
{
    if (device < 0 || device >= kMaxDevices) return fail(ICP_AMD_EINVAL, "device %d: the static scratch serves devices 0..%d", device, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        double *p = nullptr;
        HIPCHK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));   // the current device's copy (require_device set it)
        if (!p) return fail(ICP_AMD_EHIP, "hipGetSymbolAddress returned a null address for the static scratch");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return ICP_AMD_OK;
}

extern "C" size_t icp_amd_centroid_static_max_rows(void) { return kStaticCentroidMaxRows; }

extern "C" int icp_amd_centroid_static(const double *A_dev, size_t n, double *out_dev, int device, void *stream)
{
    if (!A_dev || !out_dev) return fail(ICP_AMD_EINVAL, "icp_amd_centroid_static: NULL argument");
    if (n == 0) return fail(ICP_AMD_EINVAL, "icp_amd_centroid_static: empty cloud");
    if (n > kStaticCentroidMaxRows)
        return fail(ICP_AMD_EINVAL, "icp_amd_centroid_static: %zu rows exceed the static scratch (limit %zu rows); use icp_amd_centroid",
                    n, kStaticCentroidMaxRows);
    RCCHK(require_device(device));
    double *part = nullptr;
    RCCHK(static_scratch(device, &part));
    hipStream_t s = (hipStream_t)stream;
    const size_t chunks = (n + kChunk - 1) / kChunk;
    // both kernels overwrite / read exactly `chunks` partials: nothing of an earlier call's contents is read, nothing to clear
    HIPCHK(colsum_partials(s, A_dev, Soa3{nullptr, nullptr, nullptr}, n, part));
    HIPCHK(finalize_centroid(s, part, chunks, (double)n, out_dev));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_solve(const double cov[9], const double Pc[3], const double Qc[3], double R[9], double t[3], double S[3])
{
    if (!cov || !Pc || !Qc) return fail(ICP_AMD_EINVAL, "icp_amd_solve: NULL argument");
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    hipStream_t s = nullptr;
    DevBuf part, pc, qc, rec;
    double h[kCovW] = {0};
    memcpy(h, cov, 9 * sizeof(double));
    HIPCHK(part.alloc(sizeof h));
    HIPCHK(pc.alloc(3 * sizeof(double)));
    HIPCHK(qc.alloc(3 * sizeof(double)));
    HIPCHK(rec.alloc(kSolveW * sizeof(double)));
    HIPCHK(hipMemcpyAsync(part.p, h, sizeof h, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(pc.p, Pc, 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(qc.p, Qc, 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(solve(s, part.as<double>(), 1, pc.as<double>(), qc.as<double>(), 1.0, rec.as<double>(), nullptr));
    double r[kSolveW];
    HIPCHK(hipMemcpyAsync(r, rec.p, sizeof r, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (S) memcpy(S, r + 9, 3 * sizeof(double));
    if (R) memcpy(R, r + 12, 9 * sizeof(double));
    if (t) memcpy(t, r + 21, 3 * sizeof(double));
    return ICP_AMD_OK;
}

extern "C" int icp_amd_apply(const double *P, size_t n, const double R[9], const double t[3], double *P_out)
{
    if (!P || !R || !t || !P_out) return fail(ICP_AMD_EINVAL, "icp_amd_apply: NULL argument");
    if (n == 0) return fail(ICP_AMD_EINVAL, "icp_amd_apply: empty cloud");
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    hipStream_t s = nullptr;
    const size_t n_pad = round_up(n, kChunk);
    DevBuf pst, pstage, rec, cen;
    Soa3 dP;
    RCCHK(upload_soa(s, P, n, n_pad, 0.0, pst, pstage, dP));
    double h[kSolveW] = {0};
    memcpy(h + 12, R, 9 * sizeof(double));
    memcpy(h + 21, t, 3 * sizeof(double));
    HIPCHK(rec.alloc(sizeof h));
    HIPCHK(cen.alloc((n_pad / kChunk) * kCenW * sizeof(double)));
    HIPCHK(hipMemcpyAsync(rec.p, h, sizeof h, hipMemcpyHostToDevice, s));
    HIPCHK(apply(s, dP, n, rec.as<double>(), cen.as<double>()));
    HIPCHK(soa_to_aos(s, dP, n, pstage.as<double>()));
    HIPCHK(hipMemcpyAsync(P_out, pstage.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ICP_AMD_OK;
}

// test hook (not in the public header): the fp16 MFMA filter's values against exact ones, as a fraction of its error bound.
// P (n rows, centred) against Qz (m rows, m a multiple of 32, every 32 consecutive rows identical); n is padded to 32.
extern "C" int icp_amd_test_mfma16_bound(const double *P, size_t n, const double *Qz, size_t m, double *worst_ratio)
{
    if (!P || !Qz || !worst_ratio || n == 0 || m == 0 || m % 32) return fail(ICP_AMD_EINVAL, "icp_amd_test_mfma16_bound: bad argument");
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    hipStream_t s = nullptr;
    const size_t n_pad = round_up(n, kPadRows), m_pad = round_up(m, kPadRows);
    DevBuf pst, pstage, qst, qstage, qrec, qmaxb, rec16, worst;
    Soa3 dP, dQ;
    RCCHK(upload_soa(s, P, n, n_pad, 0.0, pst, pstage, dP));
    RCCHK(upload_soa(s, Qz, m, m_pad, std::numeric_limits<double>::infinity(), qst, qstage, dQ));
    HIPCHK(qrec.alloc(16 * m_pad));
    HIPCHK(qmaxb.alloc(sizeof(unsigned long long)));
    HIPCHK(build_qrec(s, dQ, m, m_pad, qrec.p, qmaxb.as<unsigned long long>()));
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, qmaxb.p, sizeof bits, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    double qmax2;
    memcpy(&qmax2, &bits, sizeof qmax2);
    if (!(qmax2 > 1e-30 && qmax2 < 1e30)) return fail(ICP_AMD_EINVAL, "icp_amd_test_mfma16_bound: target range outside the filter's domain");
    HIPCHK(rec16.alloc(32 * m_pad));
    HIPCHK(build_qrec16(s, dQ, m, m_pad, sqrt(qmax2), rec16.p));
    HIPCHK(worst.alloc(sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(worst.p, 0, sizeof(unsigned long long), s));
    HIPCHK(mfma16_probe(s, dP, round_up(n, 32), dQ, m, m_pad, sqrt(qmax2), rec16.p, worst.as<unsigned long long>()));
    HIPCHK(hipMemcpyAsync(&bits, worst.p, sizeof bits, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(worst_ratio, &bits, sizeof *worst_ratio);
    return ICP_AMD_OK;
}

// test hook (not in the public header): device sqrt(sqrt(x)) must be correctly rounded like the host's
extern "C" int icp_amd_test_fourth_root(const double *d2, size_t n, double *key)
{
    if (!d2 || !key || n == 0) return fail(ICP_AMD_EINVAL, "icp_amd_test_fourth_root: bad argument");
    RCCHK(require_device(env_int("ICP_AMD_DEVICE", 0)));
    DevBuf a, b;
    HIPCHK(a.alloc(n * sizeof(double)));
    HIPCHK(b.alloc(n * sizeof(double)));
    HIPCHK(hipMemcpy(a.p, d2, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(fourth_root(nullptr, a.as<double>(), n, b.as<double>()));
    HIPCHK(hipMemcpy(key, b.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return ICP_AMD_OK;
}
