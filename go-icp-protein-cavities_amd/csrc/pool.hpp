// pool.hpp -- process-wide pool of device blocks and (stream, event, event, pinned buffer) quadruples, shared by the translation units of
// libgoicp_hip.so (defined in engine.hip). hipMalloc / hipFree / hipStreamCreate synchronise the device and serialise host threads, and a
// batch creates one engine (and, with dt_mode = 1, one device-built grid) per pair; freed blocks are parked per (device, size class) and
// handed out again. Streams are non-blocking: nothing here synchronises with the legacy default stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

hipError_t cache_malloc(void** out, size_t bytes, int device);
void cache_free(void* p);                                      // the block must be idle (its last user's stream synchronised)
hipError_t cache_queue(int device, hipStream_t* st, hipEvent_t* e0, hipEvent_t* e1, void** pinned);
void cache_queue_return(int device, hipStream_t st, hipEvent_t e0, hipEvent_t e1, void* pinned);
