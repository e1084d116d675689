// icp_amd/csrc/devmem.cpp -- a small cache in front of hipMalloc / hipFree.
//
// A one-shot registration (icp_amd_run: create, set_target, set_source, iterate, get_source, destroy) makes about 25
// device allocations and frees them again; at 1M points the frees alone cost 2.7 ms of an 11 ms call, at 3k points
// allocation and release are most of the call.  Released blocks are therefore kept (up to ICP_AMD_CACHE_MB, default
// 2048 MiB per process; 0 switches the cache off) and handed out again to requests of the same or a slightly smaller
// size -- the next registration of a similar cloud allocates nothing.
//
// dfree() keeps hipFree's contract: it waits for the block's device to go idle before the block can be reused, so a
// caller may release memory that kernels it has launched are still reading, exactly as with hipFree.  dfree_on(p, s) is
// for blocks that only work on stream s has touched (the temporaries of a target set-up, a grid): it waits for s alone,
// so engines driven by other host threads keep running -- a batch of registrations on eight threads runs at twice the
// rate.  (Its first version faulted: the overflow counters of the grid search were cleared with hipMemset, i.e. on the
// null stream, which is not ordered with the engines' non-blocking streams; the device-wide wait of every release had been
// hiding that.  Everything the library clears is now cleared on the stream that uses it.)
#include "kernels.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace icpk {

// ---- debugging / test switches: one list, ICP_AMD_DEBUG="name=value,name=value" (kernels.h) ----
const char *knob(const char *name)
{
    static thread_local char buf[128];
    const size_t len = strlen(name);
    if (const char *list = getenv("ICP_AMD_DEBUG")) {
        for (const char *p = list; *p;) {
            while (*p == ',' || *p == ' ') ++p;
            const char *end = strchr(p, ',');
            const size_t item = end ? (size_t)(end - p) : strlen(p);
            if (item > len && !strncmp(p, name, len) && p[len] == '=') {
                size_t vl = item - len - 1;
                while (vl && p[len + vl] == ' ') --vl;
                if (vl >= sizeof buf) vl = sizeof buf - 1;
                memcpy(buf, p + len + 1, vl);
                buf[vl] = 0;
                return buf;
            }
            p += item;
        }
    }
    char var[96] = "ICP_AMD_";   // the switch as a variable of its own (older spelling)
    size_t k = 8;
    for (size_t c = 0; c < len && k + 1 < sizeof var; ++c) var[k++] = (char)toupper((unsigned char)name[c]);
    var[k] = 0;
    const char *v = getenv(var);
    return (v && *v) ? v : nullptr;
}
int knob_int(const char *name, int dflt)
{
    const char *v = knob(name);
    return v ? atoi(v) : dflt;
}
double knob_double(const char *name, double dflt)
{
    const char *v = knob(name);
    return v ? atof(v) : dflt;
}

namespace {

struct Block { void *p; size_t bytes; int dev; };
std::mutex g_mu;
std::unordered_map<void *, Block> g_live;   // blocks handed out
std::vector<Block> g_pool;                  // blocks released and kept
size_t g_held = 0;

size_t cache_cap()
{
    static const size_t cap = [] {
        const char *v = getenv("ICP_AMD_CACHE_MB");
        const long mb = (v && *v) ? atol(v) : 2048;
        return (size_t)(mb > 0 ? mb : 0) << 20;
    }();
    return cap;
}

// drop every kept block of `dev` (all devices: dev < 0); g_mu held
void trim_locked(int dev)
{
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (size_t k = 0; k < g_pool.size();) {
        if (dev >= 0 && g_pool[k].dev != dev) { ++k; continue; }
        (void)hipSetDevice(g_pool[k].dev);
        (void)hipFree(g_pool[k].p);
        g_held -= g_pool[k].bytes;
        g_pool[k] = g_pool.back();
        g_pool.pop_back();
    }
    (void)hipSetDevice(cur);
}

}  // namespace

// ICP_AMD_CACHE_POISON=1 (tests): every block is handed out filled with the 32-bit pattern 3 -- a small valid-looking
// index, a denormal as a double -- so that code relying on fresh (zeroed) or on stale memory shows up as a wrong result.
static hipError_t poisoned(void *p, size_t bytes)
{
    static const bool on = knob_int("cache_poison", 0) != 0;
    if (!on) return hipSuccess;
    hipError_t e = hipMemsetD32((hipDeviceptr_t)p, 3, bytes / 4);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

static hipError_t dmalloc_raw(void **p, size_t bytes);
hipError_t dmalloc_bytes(void **p, size_t bytes)
{
    hipError_t e = dmalloc_raw(p, bytes);
    if (e == hipSuccess) e = poisoned(*p, bytes ? bytes : 8);
    return e;
}

static hipError_t dmalloc_raw(void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        size_t best = g_pool.size();
        for (size_t k = 0; k < g_pool.size(); ++k) {
            const Block &b = g_pool[k];
            if (b.dev == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
                (best == g_pool.size() || b.bytes < g_pool[best].bytes))
                best = k;
        }
        if (best != g_pool.size()) {
            const Block b = g_pool[best];
            g_pool[best] = g_pool.back();
            g_pool.pop_back();
            g_held -= b.bytes;
            g_live[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {   // give the kept blocks back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(g_mu);
            trim_locked(dev);
        }
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) { *p = nullptr; return e; }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*p] = Block{*p, bytes, dev};
    return hipSuccess;
}

static hipError_t release(void *p, bool whole_device, hipStream_t s);
hipError_t dfree(void *p) { return release(p, true, nullptr); }
hipError_t dfree_on(void *p, hipStream_t s) { return release(p, false, s); }

static hipError_t release(void *p, bool whole_device, hipStream_t s)
{
    if (!p) return hipSuccess;
    Block b{p, 0, -1};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) { b = it->second; g_live.erase(it); }
    }
    if (b.dev < 0) return hipFree(p);   // not ours
    if (b.bytes > cache_cap()) return hipFree(p);
    // hipFree's contract: nothing launched before this call may still touch the block once it is handed out again
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != b.dev) (void)hipSetDevice(b.dev);
    const hipError_t e = whole_device ? hipDeviceSynchronize() : hipStreamSynchronize(s);
    if (cur != b.dev) (void)hipSetDevice(cur);
    if (e != hipSuccess) { (void)hipFree(p); return e; }
    std::lock_guard<std::mutex> lk(g_mu);
    while (g_held + b.bytes > cache_cap() && !g_pool.empty()) {   // make room: oldest kept block first
        const Block old = g_pool.front();
        g_pool.erase(g_pool.begin());
        g_held -= old.bytes;
        int c2 = 0;
        (void)hipGetDevice(&c2);
        (void)hipSetDevice(old.dev);
        (void)hipFree(old.p);
        (void)hipSetDevice(c2);
    }
    g_pool.push_back(b);
    g_held += b.bytes;
    return hipSuccess;
}

// Pinned host buffers of kPinBytes for staging small copies (engine.hip: copy_in / copy_out): pinning is the expensive part,
// so released buffers are kept (at most eight, none with ICP_AMD_CACHE_MB=0) and handed to the next engine.
static std::vector<void *> g_pins;   // under g_mu
hipError_t pin_acquire(void **p)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_pins.empty()) {
            *p = g_pins.back();
            g_pins.pop_back();
            return hipSuccess;
        }
    }
    return hipHostMalloc(p, kPinBytes, hipHostMallocDefault);
}
void pin_release(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (cache_cap() && g_pins.size() < 8) {
            g_pins.push_back(p);
            return;
        }
    }
    (void)hipHostFree(p);
}

void dcache_trim()
{
    std::lock_guard<std::mutex> lk(g_mu);
    trim_locked(-1);
    for (void *p : g_pins) (void)hipHostFree(p);
    g_pins.clear();
}

size_t dcache_held()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_held;
}

}  // namespace icpk
