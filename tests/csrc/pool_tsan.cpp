// pool_tsan.cpp -- ThreadSanitizer harness for the state that a batch's worker threads share inside libgoicp_hip.so
// (csrc/host_pool.cpp: device-block cache, pooled streams / events / pinned buffers, auxiliary streams, the once-per-instantiation
// dynamic-LDS ceiling, the polling waits). Test infrastructure: host_pool.cpp is compiled as it ships, against the STUB HIP runtime
// below (blocks are malloc'ed, streams and events are small heap objects), so that the pool's own locking is what is examined.
//
// Build (tests/test_host_vs_oracle.py does this):
//   g++ -std=c++17 -O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -I<csrc> \
//       tests/csrc/pool_tsan.cpp <csrc>/host_pool.cpp -o pool_tsan -pthread
// Exit code 0 and no "ThreadSanitizer" report = pass. Ownership is checked beside the data-race detector: every pooled object carries a
// relaxed-atomic "in use" flag (relaxed, so that the check adds no happens-before edge that could hide a race), and every block is
// filled with its owner's tag and read back before it is returned.
#include "pool.hpp"
#include "goicp_engine.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

struct StubObj { std::atomic<int> in_use{0}; int kind; };
static std::atomic<long> g_live_blocks{0}, g_live_objs{0}, g_errors{0}, g_attr_sets{0};
static StubObj* as_obj(void* p) { return reinterpret_cast<StubObj*>(p); }
static void fail(const char* what) { fprintf(stderr, "pool_tsan: %s\n", what); g_errors.fetch_add(1, std::memory_order_relaxed); }

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); g_live_blocks.fetch_add(1, std::memory_order_relaxed); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); g_live_blocks.fetch_sub(1, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static hipError_t new_obj(void** out, int kind) { StubObj* o = new StubObj; o->kind = kind; *out = o; g_live_objs.fetch_add(1, std::memory_order_relaxed); return hipSuccess; }
static hipError_t del_obj(void* p) { if (as_obj(p)->in_use.load(std::memory_order_relaxed)) fail("object destroyed while in use"); delete as_obj(p); g_live_objs.fetch_sub(1, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { return new_obj(reinterpret_cast<void**>(s), 1); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned int, int) { return new_obj(reinterpret_cast<void**>(s), 2); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return new_obj(reinterpret_cast<void**>(s), 3); }
hipError_t hipStreamDestroy(hipStream_t s) { return del_obj(s); }
hipError_t hipEventCreate(hipEvent_t* e) { return new_obj(reinterpret_cast<void**>(e), 4); }
hipError_t hipEventDestroy(hipEvent_t e) { return del_obj(e); }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -2; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// a "kernel" is finished the third time it is asked about (the polling waits then go through their spin and nap branches)
static thread_local int t_polls = 0;
hipError_t hipEventQuery(hipEvent_t) { return ++t_polls % 3 ? hipErrorNotReady : hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return ++t_polls % 3 ? hipErrorNotReady : hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { memset(a, 0, sizeof *a); a->sharedSizeBytes = 4096; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { g_attr_sets.fetch_add(1, std::memory_order_relaxed); return hipSuccess; }
}

static void take(void* obj, const char* what) { if (as_obj(obj)->in_use.exchange(1, std::memory_order_relaxed)) fail(what); }
static void give(void* obj) { as_obj(obj)->in_use.store(0, std::memory_order_relaxed); }

static int g_fake_kernels[8];          // addresses stand for kernel instantiations

static void worker(int id, int rounds) {
    unsigned rng = 12345u + 977u * (unsigned)id;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (int r = 0; r < rounds; r++) {
        // what goicp_engine_create does: a queue, the auxiliary streams, a few dozen blocks, the LDS ceilings
        hipStream_t st = nullptr, hi = nullptr, bulk = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; void* pinned = nullptr;
        if (cache_queue(0, &st, &e0, &e1, &pinned) != hipSuccess) { fail("cache_queue"); return; }
        take(st, "stream handed to two engines"); take(e0, "event handed to two engines"); take(e1, "event handed to two engines");
        if (pinned) memset(pinned, id, 64);
        if (cache_stream_aux(0, 0, &hi) != hipSuccess || cache_stream_aux(0, 1, &bulk) != hipSuccess) { fail("cache_stream_aux"); return; }
        const bool shared_hi = hi_share_count() > 0;                   // GOICP_HI_SHARE: the high-priority streams belong to all engines of a device
        if (!shared_hi) take(hi, "aux stream handed to two engines");
        take(bulk, "bulk stream handed to two engines");
        std::vector<std::pair<unsigned char*, size_t>> blocks;
        const int nb = 4 + (int)(rnd() % 12);
        for (int b = 0; b < nb; b++) {
            static const size_t sizes[] = {64, 300, 4096, 70000, (size_t)1 << 20, ((size_t)3 << 20) + 17, (size_t)5 << 20};
            const size_t n = sizes[rnd() % 7];
            void* p = nullptr;
            if (cache_malloc(&p, n, 0) != hipSuccess) { fail("cache_malloc"); return; }
            memset(p, 0x40 + id, n);                                   // a block shared with another owner shows up as a race or a wrong tag
            blocks.push_back({(unsigned char*)p, n});
        }
        for (int k = 0; k < 3; k++) if (set_max_dyn_lds(&g_fake_kernels[rnd() % 8]) != hipSuccess) fail("set_max_dyn_lds");
        if (wait_stream_nap(st, 5, 20) != hipSuccess || wait_event_nap(e0, 0, 20) != hipSuccess) fail("wait");
        std::this_thread::yield();
        for (auto& b : blocks) {
            if (b.first[0] != 0x40 + id || b.first[b.second - 1] != 0x40 + id) fail("block overwritten by another owner");
            cache_free(b.first);
        }
        // goicp_engine_destroy
        if (!shared_hi) give(hi);
        cache_stream_aux_return(0, 0, hi);
        give(bulk); cache_stream_aux_return(0, 1, bulk);
        give(st); give(e0); give(e1);
        cache_queue_return(0, st, e0, e1, pinned);
        // a caller emptying the pool between batches while other threads still register (legal: only idle objects are parked)
        // (not with shared high-priority streams: the header asks for no live engine then)
        if (id == 0 && r % 16 == 15 && !shared_hi) goicp_engine_release_cached_memory();
    }
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 16, rounds = argc > 2 ? atoi(argv[2]) : 200;
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; t++) ts.emplace_back(worker, t, rounds);
    for (auto& t : ts) t.join();
    goicp_engine_release_cached_memory();
    if (g_live_blocks.load() != 0) fail("device blocks leaked");
    if (g_live_objs.load() != 0) fail("streams / events leaked");
    if (g_attr_sets.load() > 8) fail("dynamic-LDS ceiling set more than once per instantiation");
    printf("pool_tsan: %d threads x %d engines, errors %ld, attribute sets %ld\n", threads, rounds, g_errors.load(), g_attr_sets.load());
    return g_errors.load() ? 1 : 0;
}
