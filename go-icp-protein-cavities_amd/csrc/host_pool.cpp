// host_pool.cpp -- the process-wide pool behind pool.hpp: device blocks, (stream, event, event, pinned buffer) quadruples, auxiliary
// (high-priority / CU-masked) streams, the polling waits, and the once-per-instantiation dynamic-LDS ceiling. Host-only C++ (HIP runtime API,
// no kernels), so that it also builds against a stubbed runtime under ThreadSanitizer (tests/csrc/pool_tsan.cpp): everything a batch's
// worker threads share inside libgoicp_hip.so lives here.
#include "pool.hpp"
#include "goicp_engine.h"
#include <sys/prctl.h>
#include <time.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------------------------
// Device-memory cache. One engine lives for one pair, so a batch creates and destroys engines all the time; hipMalloc and
// hipFree synchronise the device and serialise the host threads that run registrations side by side. Freed blocks are
// therefore parked per (device, size class) and handed out again; at most GOICP_CACHE_GB (default 16) gigabytes stay parked.
// A block is only parked after its engine's stream has been synchronised, so the next owner may use it on any stream.
// ---------------------------------------------------------------------------------------------
namespace {
struct DevCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void*> parked;
    std::map<void*, std::pair<int, size_t>> live;
    size_t parked_bytes = 0, limit = 0; bool init = false, off = false;
    struct Queue { int device; hipStream_t stream; hipEvent_t ev0, ev1; void* pinned; };   // pinned: PINNED_BYTES of host memory the device writes results into
    std::vector<Queue> queues;                 // idle (stream, event, event) triples of destroyed engines
    struct Aux { int device, kind; hipStream_t stream; };
    std::vector<Aux> aux_streams;              // idle auxiliary streams (cache_stream_aux)
    struct SharedHi { std::vector<hipStream_t> streams; size_t next = 0; };
    std::map<int, SharedHi> shared_hi;         // per device: the high-priority streams all engines share (GOICP_HI_SHARE)
    static size_t size_class(size_t n) {
        if (n <= 256) return 256;
        if (n < ((size_t)1 << 20)) { size_t p2 = 256; while (p2 < n) p2 <<= 1; return p2; }
        return (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    }
    void setup() {
        if (init) return;
        init = true;
        const char* e = getenv("GOICP_CACHE_GB");
        const double gb = e ? atof(e) : 16.0;
        off = gb <= 0.0; limit = (size_t)(gb * 1073741824.0);
    }
};
DevCache g_cache;
}  // namespace
hipError_t cache_malloc(void** out, size_t bytes, int device) {
    const size_t cls = DevCache::size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        g_cache.setup();
        if (!g_cache.off) {
            auto it = g_cache.parked.find({device, cls});
            if (it != g_cache.parked.end()) {
                *out = it->second; g_cache.parked.erase(it); g_cache.parked_bytes -= cls;
                g_cache.live[*out] = {device, cls};
                return hipSuccess;
            }
        }
    }
    const hipError_t e = hipMalloc(out, cls);
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_cache.mu); g_cache.live[*out] = {device, cls}; }
    return e;
}
void cache_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        auto it = g_cache.live.find(p);
        if (it != g_cache.live.end()) {
            const auto key = it->second;
            g_cache.live.erase(it);
            if (!g_cache.off && g_cache.parked_bytes + key.second <= g_cache.limit) { g_cache.parked.emplace(key, p); g_cache.parked_bytes += key.second; return; }
        }
    }
    (void)hipFree(p);
}
hipError_t cache_queue(int device, hipStream_t* st, hipEvent_t* e0, hipEvent_t* e1, void** pinned) {
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        g_cache.setup();
        for (size_t k = 0; k < g_cache.queues.size(); k++) if (g_cache.queues[k].device == device) {
            *st = g_cache.queues[k].stream; *e0 = g_cache.queues[k].ev0; *e1 = g_cache.queues[k].ev1; *pinned = g_cache.queues[k].pinned;
            g_cache.queues.erase(g_cache.queues.begin() + k);
            return hipSuccess;
        }
    }
    // non-blocking: a plain stream would synchronise with the legacy default stream, i.e. with every synchronous hipMemcpy / hipMemset any
    // other host thread issues -- engines that run side by side (batch driver) then wait for each other's kernels
    hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(e0);
    if (e == hipSuccess) e = hipEventCreate(e1);
    if (e == hipSuccess && hipHostMalloc(pinned, PINNED_BYTES) != hipSuccess) *pinned = nullptr;   // optional: callers fall back to copies
    return e;
}
void cache_queue_return(int device, hipStream_t st, hipEvent_t e0, hipEvent_t e1, void* pinned) {
    if (st && e0 && e1) {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        if (!g_cache.off && g_cache.queues.size() < 64) { g_cache.queues.push_back({device, st, e0, e1, pinned}); return; }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (pinned) (void)hipHostFree(pinned);
}
// auxiliary streams of an engine, pooled like the rest. kind 0: high priority, for the small latency-critical kernels (ICP 1-NN, point
// evaluation). kind 1: the bulk stream of cavity-sized engines' search launches (k_inner_bnb, hundreds to thousands of workgroups that hold a
// CU's whole register file two at a time), confined to all but GOICP_CU_RESERVE (default 16) CUs -- beside other engines' launches the small
// kernels otherwise wait for search workgroups to retire (configs[3] batch, 16 registrations in flight: 313 -> 390 pairs/s).
int hi_share_count() { static const int n = [] { const char* e = getenv("GOICP_HI_SHARE"); return e ? std::max(0, atoi(e)) : 0; }(); return n; }
static hipError_t create_priority_stream(hipStream_t* st) {
    int lo = 0, hi = 0;                                        // (numerically lower = higher priority)
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
    if (hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
hipError_t cache_stream_aux(int device, int kind, hipStream_t* st) {
    if (kind == 0 && hi_share_count() > 0) {
        // GOICP_HI_SHARE=P: P high-priority streams per device for ALL engines (round robin) instead of one each. Every priority or CU-masked
        // stream is a hardware queue of its own, and past ~36 queues per process the GPU time-slices them: the configs[3] batch falls from
        // 557 pairs/s with 16 engines to 354 with 18. Engines wait on their own event, not on the stream.
        std::lock_guard<std::mutex> lk(g_cache.mu);
        auto& v = g_cache.shared_hi[device];
        if ((int)v.streams.size() < hi_share_count()) {
            hipStream_t s = nullptr; const hipError_t e = create_priority_stream(&s);
            if (e != hipSuccess) return e;
            v.streams.push_back(s); *st = s; return hipSuccess;
        }
        *st = v.streams[v.next++ % v.streams.size()];
        return hipSuccess;
    }
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        for (size_t k = 0; k < g_cache.aux_streams.size(); k++) if (g_cache.aux_streams[k].device == device && g_cache.aux_streams[k].kind == kind) {
            *st = g_cache.aux_streams[k].stream; g_cache.aux_streams.erase(g_cache.aux_streams.begin() + k); return hipSuccess;
        }
    }
    if (kind == 0) return create_priority_stream(st);
    int reserve = 16, ncu = 0;
    { const char* r = getenv("GOICP_CU_RESERVE"); if (r) reserve = atoi(r); }
    if (reserve > 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && 2 * reserve <= ncu) {
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int c = 0; c < ncu - reserve; c++) mask[(size_t)c / 32] |= 1u << (c % 32);
        if (hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();                               // (a runtime without CU masks: an ordinary stream does the job, only slower in a batch)
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
void cache_stream_aux_return(int device, int kind, hipStream_t st) {
    if (!st) return;
    if (kind == 0 && hi_share_count() > 0) return;           // shared: stays with the pool
    {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        if (!g_cache.off && g_cache.aux_streams.size() < 128) { g_cache.aux_streams.push_back({device, kind, st}); return; }
    }
    (void)hipStreamDestroy(st);
}
// GOICP_WAIT=nap: a host thread that spins on the device burns a CPU of the process's budget for the whole kernel; most waits of a
// cavity-sized registration are for launches of hundreds of microseconds. Poll the stream instead: spin for the first spin_us (short
// kernels return inside it), then sleep nap_us between polls (timer slack lowered once per thread so that a nap is a nap).
template <class Q>
static hipError_t wait_nap_impl(Q query, int spin_us, int nap_us) {
    static thread_local bool slack_set = false;
    if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); slack_set = true; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (us < spin_us) { __builtin_ia32_pause(); continue; }
        struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)nap_us * 1000L;
        (void)nanosleep(&ts, nullptr);
    }
}
hipError_t wait_event_nap(hipEvent_t ev, int spin_us, int nap_us) { return wait_nap_impl([ev] { return hipEventQuery(ev); }, spin_us, nap_us); }
hipError_t wait_stream_nap(hipStream_t st, int spin_us, int nap_us) { return wait_nap_impl([st] { return hipStreamQuery(st); }, spin_us, nap_us); }
extern "C" void goicp_engine_release_cached_memory(void) {
    std::vector<void*> blocks;
    std::vector<DevCache::Queue> qs;
    std::vector<DevCache::Aux> aux;
    { std::lock_guard<std::mutex> lk(g_cache.mu); for (auto& kv : g_cache.parked) blocks.push_back(kv.second); g_cache.parked.clear(); g_cache.parked_bytes = 0; qs.swap(g_cache.queues); aux.swap(g_cache.aux_streams);
      for (auto& kv : g_cache.shared_hi) for (hipStream_t s : kv.second.streams) aux.push_back({kv.first, 0, s});      // (no engine may be alive: see the header)
      g_cache.shared_hi.clear(); }
    for (auto& q : aux) (void)hipStreamDestroy(q.stream);
    for (void* p : blocks) (void)hipFree(p);
    for (auto& q : qs) { (void)hipEventDestroy(q.ev0); (void)hipEventDestroy(q.ev1); (void)hipStreamDestroy(q.stream); if (q.pinned) (void)hipHostFree(q.pinned); }
}

// The dynamic-LDS ceiling of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-process property of the instantiation, and
// engines of different cloud sizes share instantiations while they run side by side on host threads (batch driver): it is therefore
// raised ONCE per instantiation to everything the CU has left beside the kernel's static LDS, never to one engine's own need.
hipError_t set_max_dyn_lds(const void* fn) {
    static std::mutex mu; static std::map<std::pair<int, const void*>, int> done;      // per (device, instantiation)
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const std::pair<int, const void*> key(dev, fn);
    if (done.count(key)) return hipSuccess;
    hipFuncAttributes a;
    e = hipFuncGetAttributes(&a, fn);
    if (e != hipSuccess) return e;
    const int room = 160 * 1024 - (int)a.sharedSizeBytes;
    if (room <= 0) return hipErrorInvalidValue;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, room);
    if (e == hipSuccess) done[key] = room;
    return e;
}
