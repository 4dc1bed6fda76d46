// What the kernel-argument placement costs a SMALL kernel on the GPU's timeline, and
// whether a kernel that takes NO arguments (its block in a __device__ ring the CPU writes
// through the BAR, its slot from a device-side counter) escapes it.  One dependent chain
// of 1.8 MB writers on one stream; period per launch with the GPU saturated (host far
// ahead) and host cost per launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/kernarg_lab tools/kernarg_lab.hip
//   HIP_FORCE_DEV_KERNARG=0 tools/build/kernarg_lab; HIP_FORCE_DEV_KERNARG=1 tools/build/kernarg_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <x86intrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Args { u32x4* dst; unsigned value; unsigned n; unsigned pad[11]; };   // 64 B
static_assert(sizeof(Args) == 64, "");
constexpr int kSlots = 4096;
__device__ Args g_args[kSlots];
__device__ unsigned g_turn, g_done;

__device__ __forceinline__ void body(const Args& a) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.dst[i] = u32x4{a.value, i, a.value, i};
}
__global__ __launch_bounds__(256) void by_value(const Args a) { body(a); }
__global__ __launch_bounds__(256) void by_pointer(const Args* a) { body(*a); }
__global__ __launch_bounds__(256) void no_args() {
  const unsigned turn = __hip_atomic_load(&g_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const Args a = g_args[turn % kSlots];
  body(a);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned blocks = (a.n + 255) / 256;
    if (__hip_atomic_fetch_add(&g_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1 == blocks) {
      __hip_atomic_store(&g_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g_turn, turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ void empty_kernel() {}

int main() {
  const unsigned n = 1806336 / 16;             // 64 frames of 28 224 B
  const unsigned grid = (n + 255) / 256;
  u32x4* dst;
  CHECK(hipMalloc(&dst, n * 16));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Args* slots;                                  // the ring's own address (device memory; CPU-writable with a large BAR)
  CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&slots), HIP_SYMBOL(g_args)));
  Args* ring;                                   // by_pointer: a fine-grained ring like the library's ArgRing
  CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&ring), kSlots * sizeof(Args), hipDeviceMallocFinegrained));
  Args host{dst, 1u, n, {}};
  // can the CPU write the module's global directly?
  int large = 0;
  CHECK(hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, 0));
  std::printf("HIP_FORCE_DEV_KERNARG=%s large BAR %d\n", std::getenv("HIP_FORCE_DEV_KERNARG") ? std::getenv("HIP_FORCE_DEV_KERNARG") : "(unset)", large);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto&& launch, int iters = 20000) {
    for (int i = 0; i < 500; ++i) launch(i);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) launch(i);
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%-58s host %.2f us / launch, GPU period %.2f us\n", name, host_us, ms * 1e3 / iters);
  };
  run("empty kernel", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); });
  run("1.8 MB writer, 64 B of arguments by value", [&](int i) {
    host.value = i;
    hipLaunchKernelGGL(by_value, dim3(grid), dim3(256), 0, s, host);
  });
  run("1.8 MB writer, pointer to a block in the fine-grained ring", [&](int i) {
    host.value = i;
    Args* slot = ring + i % kSlots;
    std::memcpy(slot, &host, sizeof(host));
    _mm_sfence();
    hipLaunchKernelGGL(by_pointer, dim3(grid), dim3(256), 0, s, slot);
  });
  if (large) {
    unsigned zero = 0, turn = 0;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_turn), &zero, 4));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_done), &zero, 4));
    // Probe: does a CPU store to the symbol's address reach the kernel?
    host.value = 77;
    std::memcpy(slots, &host, sizeof(host));
    _mm_sfence();
    hipLaunchKernelGGL(no_args, dim3(grid), dim3(256), 0, s);
    CHECK(hipStreamSynchronize(s));
    unsigned back[4];
    CHECK(hipMemcpy(back, dst, 16, hipMemcpyDeviceToHost));
    std::printf("no-argument kernel read its block from the CPU-written global: %s\n", back[0] == 77 ? "yes" : "NO");
    turn = 1;
    if (back[0] == 77)
      run("1.8 MB writer, NO arguments (block in a __device__ ring, slot from a device counter)", [&](int i) {
        host.value = i;
        // (the host keeps the same count as the device: one launch = one turn; slots are reused
        //  kSlots launches later -- a real ring waits for the launch that read the slot)
        std::memcpy(slots + turn % kSlots, &host, sizeof(host));
        _mm_sfence();
        ++turn;
        hipLaunchKernelGGL(no_args, dim3(grid), dim3(256), 0, s);
        if (i % 2048 == 2047) CHECK(hipStreamSynchronize(s));   // keep the CPU within the ring
      });
  }
  return 0;
}
