// pool.hpp -- process-wide pool of device blocks and (stream, event, event, pinned buffer) quadruples, shared by the translation units of
// libgoicp_hip.so (defined in host_pool.cpp). hipMalloc / hipFree / hipStreamCreate synchronise the device and serialise host threads, and a
// batch creates one engine (and, with dt_mode = 1, one device-built grid) per pair; freed blocks are parked per (device, size class) and
// handed out again. Streams are non-blocking: nothing here synchronises with the legacy default stream.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>

hipError_t cache_malloc(void** out, size_t bytes, int device);
void cache_free(void* p);                                      // the block must be idle (its last user's stream synchronised)
hipError_t cache_queue(int device, hipStream_t* st, hipEvent_t* e0, hipEvent_t* e1, void** pinned);
void cache_queue_return(int device, hipStream_t st, hipEvent_t e0, hipEvent_t e1, void* pinned);

#define PINNED_BYTES ((size_t)1 << 20)
// auxiliary streams of an engine (kind 0: high priority; kind 1: CU-masked bulk stream of cavity-sized search launches)
int hi_share_count();   // GOICP_HI_SHARE: high-priority streams shared by all engines of a device (0 = one per engine)
hipError_t cache_stream_aux(int device, int kind, hipStream_t* st);
void cache_stream_aux_return(int device, int kind, hipStream_t st);
// host waits that poll: spin for spin_us, then nap_us sleeps between polls (GOICP_WAIT=nap)
hipError_t wait_stream_nap(hipStream_t st, int spin_us, int nap_us);
hipError_t wait_event_nap(hipEvent_t ev, int spin_us, int nap_us);
// raise a kernel instantiation's dynamic-LDS ceiling once per (device, instantiation)
hipError_t set_max_dyn_lds(const void* fn);
