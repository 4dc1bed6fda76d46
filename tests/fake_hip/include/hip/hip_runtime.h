// TEST INFRASTRUCTURE (tests/fake_hip): a stand-in for <hip/hip_runtime.h> that
// lets the HOST side of libembodied_hip.so -- csrc/replay_abi.cpp, index_abi.cpp,
// kernels_abi.cpp with replay_index.h / selectors.h / defer_gate.h /
// stream_order.h -- be compiled by a plain C++ compiler with
// -fsanitize=thread or -fsanitize=address,undefined (tools/run_sanitizers.sh).
// "Device" memory is host memory, a "stream" executes at once, events are
// counters.  Only tests/sanitize links against this; the product never does.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct fakeStream* hipStream_t;
struct fakeEvent { uint64_t recorded; };
typedef fakeEvent* hipEvent_t;
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1, hipHostMallocDefault = 0,
       hipDeviceMallocFinegrained = 1, hipMemcpyHostToDevice = 1 };
enum { hipDeviceAttributeIsLargeBar = 1, hipDeviceAttributeMultiprocessorCount = 2 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "fake hip error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* value, int attr, int) {
  *value = attr == hipDeviceAttributeIsLargeBar ? 1 : 256;
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

inline hipError_t fake_alloc(void** p, size_t bytes) {
  *p = std::malloc(bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipMalloc(void** p, size_t bytes) { return fake_alloc(p, bytes); }
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return fake_alloc(p, bytes); }
inline hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return fake_alloc(p, bytes); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, int, hipStream_t) {
  std::memcpy(dst, src, bytes);
  return hipSuccess;
}

inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new fakeEvent{0}; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->recorded += 1; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = nullptr; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount = 256; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
