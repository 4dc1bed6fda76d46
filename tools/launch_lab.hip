// Host cost of one kernel launch by API (tiny kernel, same stream, no sync inside
// the loop): hipLaunchKernelGGL, hipExtLaunchKernelGGL (null events),
// hipModuleLaunchKernel with a resolved hipFunction_t, with a 64-byte and a
// 3.8 KB argument block.   Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/launch_lab tools/launch_lab.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Small { float* p; int n; float s; char pad[48]; };
struct Big { float* p; char pad[3800]; };
__global__ void k_small(const Small a) { if (threadIdx.x == 0 && a.n < 0) a.p[0] = a.s; }
__global__ void k_big(const Big a) { if (threadIdx.x == 0 && a.pad[7] == 99) a.p[0] = 1.f; }

__global__ void k_spin(long long ticks, float* p) {     // ~ticks of the 100 MHz wall clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) p[0] = 1.f;
}

template <typename F>
double time_us(F&& fn, int iters = 20000) {
  for (int i = 0; i < 2000; ++i) fn();
  CHECK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) fn();
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  CHECK(hipDeviceSynchronize());
  return us;
}

int main() {
  float* p;
  CHECK(hipMalloc(&p, 4096));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Small small{p, 1, 2.f, {}};
  Big big{p, {}};
  hipFunction_t f_small, f_big;
  CHECK(hipGetFuncBySymbol(&f_small, reinterpret_cast<const void*>(k_small)));
  CHECK(hipGetFuncBySymbol(&f_big, reinterpret_cast<const void*>(k_big)));
  std::printf("HIP_FORCE_DEV_KERNARG=%s\n", std::getenv("HIP_FORCE_DEV_KERNARG") ? std::getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
  std::printf("hipLaunchKernelGGL      64 B : %.2f us\n", time_us([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, small); }));
  std::printf("hipExtLaunchKernelGGL   64 B : %.2f us\n", time_us([&] { hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, small); }));
  std::printf("hipModuleLaunchKernel   64 B : %.2f us\n", time_us([&] {
    void* args[] = {&small};
    (void)hipModuleLaunchKernel(f_small, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr); }));
  std::printf("hipModuleLaunch (buffer) 64 B : %.2f us\n", time_us([&] {
    size_t size = sizeof(small);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &small, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    (void)hipModuleLaunchKernel(f_small, 1, 1, 1, 64, 1, 1, 0, s, nullptr, config); }));
  std::printf("hipLaunchKernelGGL    3.8 KB : %.2f us\n", time_us([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, big); }));
  std::printf("hipExtLaunchKernelGGL 3.8 KB : %.2f us\n", time_us([&] { hipExtLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, big); }));
  std::printf("hipModuleLaunchKernel 3.8 KB : %.2f us\n", time_us([&] {
    void* args[] = {&big};
    (void)hipModuleLaunchKernel(f_big, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr); }));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::printf("hipExtLaunch + 2 stamps 64 B : %.2f us\n", time_us([&] { hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, e0, e1, 0, small); }, 5000));
  std::printf("hipExtLaunch + stop     64 B : %.2f us\n", time_us([&] { hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, nullptr, e1, 0, small); }, 5000));
  // What do the stamps of hipExtLaunchKernelGGL read?  A 30 us kernel behind a
  // 100 us predecessor on the same stream, three ways of stamping it.
  {
    hipEvent_t a, b, c, d;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); CHECK(hipEventCreate(&c)); CHECK(hipEventCreate(&d));
    for (int rep = 0; rep < 3; ++rep) {
      float two = -1.f, stop_only = -1.f, same = -1.f;
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 10000LL, p);
      hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, a, b, 0, 3000LL, p);
      CHECK(hipStreamSynchronize(s));
      hipError_t e1_ = hipEventElapsedTime(&two, a, b);
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 10000LL, p);
      hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, nullptr, c, 0, 3000LL, p);
      CHECK(hipStreamSynchronize(s));
      hipError_t e2_ = hipEventElapsedTime(&stop_only, c, c);
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 10000LL, p);
      hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, d, 0, 3000LL, p);
      CHECK(hipStreamSynchronize(s));
      hipError_t e3_ = hipEventElapsedTime(&same, d, d);
      std::printf("30 us kernel: start+stop events %.2f us (%d) | stop only, elapsed(e,e) %.2f us (%d) | "
                  "same event twice %.2f us (%d)\n", two * 1e3, int(e1_), stop_only * 1e3, int(e2_),
                  same * 1e3, int(e3_));
    }
    std::printf("hipExtLaunch same event twice 64 B : %.2f us\n", time_us([&] { hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, d, 0, small); }, 5000));
  }
  hipEvent_t ev;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  std::printf("hipEventRecord (no timing)   : %.2f us\n", time_us([&] { (void)hipEventRecord(ev, s); }, 5000));
  return 0;
}
