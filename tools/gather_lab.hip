// Stand-alone lab for the Replay.sample gather at BASELINE shape (B sequences
// of L=65 rows of 28 224 B out of a 2.8 GB pool): candidate kernel structures
// side by side with a plain contiguous copy of the same bytes, each timed
//   A  per dispatch (begin/end stamps, what rocprofv3 reports), tight loop
//   B  as back-to-back throughput (events around N unstamped launches)
//   C  per dispatch with one tiny kernel between two gathers (pipeline case)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/gather_lab tools/gather_lab.hip
// Run:   tools/build/gather_lab [B=16] [iters=200]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxSeq = 64;
struct Spans { uint32_t w[3 * kMaxSeq]; };   // {row0, count0, row1} per sequence

template <int NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if (NT & 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if (NT & 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- contiguous copy: the speed of light for this byte count -------------
template <int U, int NT>
__global__ void copy_flat(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint32_t units) {
  const uint32_t base = blockIdx.x * (blockDim.x * U) + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int j = 0; j < U; ++j) if (base + j * blockDim.x < units) v[j] = ld<NT>(src + base + j * blockDim.x);
#pragma unroll
  for (int j = 0; j < U; ++j) if (base + j * blockDim.x < units) st<NT>(dst + base + j * blockDim.x, v[j]);
}

// ---- grid(x = piece of a row, y = t, z = sequence): no divisions ---------
template <int U, int NT>
__global__ void gather_grid(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                            uint32_t upr, const Spans sp) {
  const uint32_t seq = blockIdx.z, t = blockIdx.y, L = gridDim.y;
  const uint32_t row0 = sp.w[3 * seq], n0 = sp.w[3 * seq + 1], row1 = sp.w[3 * seq + 2];
  const uint32_t row = t < n0 ? row0 + t : row1 + (t - n0);
  const u32x4* src = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row) * upr;
  u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq * L + t) * upr;
  const uint32_t base = blockIdx.x * (blockDim.x * U) + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int j = 0; j < U; ++j) if (base + j * blockDim.x < upr) v[j] = ld<NT>(src + base + j * blockDim.x);
#pragma unroll
  for (int j = 0; j < U; ++j) if (base + j * blockDim.x < upr) st<NT>(dst + base + j * blockDim.x, v[j]);
}

// ---- same, span table read from device memory (scalar loads) -------------
template <int U, int NT>
__global__ void gather_grid_ptr(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                                uint32_t upr, const uint32_t* __restrict__ sp) {
  const uint32_t seq = blockIdx.z, t = blockIdx.y, L = gridDim.y;
  const uint32_t row0 = sp[3 * seq], n0 = sp[3 * seq + 1], row1 = sp[3 * seq + 2];
  const uint32_t row = t < n0 ? row0 + t : row1 + (t - n0);
  const u32x4* src = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row) * upr;
  u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq * L + t) * upr;
  const uint32_t base = blockIdx.x * (blockDim.x * U) + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int j = 0; j < U; ++j) if (base + j * blockDim.x < upr) v[j] = ld<NT>(src + base + j * blockDim.x);
#pragma unroll
  for (int j = 0; j < U; ++j) if (base + j * blockDim.x < upr) st<NT>(dst + base + j * blockDim.x, v[j]);
}

// ---- span-contiguous: a sequence is 1-2 contiguous byte ranges; treat the
// whole sequence as a flat run of units (no row structure at all) -----------
template <int U, int NT>
__global__ void gather_span(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                            uint32_t upr, uint32_t L, const Spans sp) {
  const uint32_t seq = blockIdx.y;
  const uint32_t row0 = sp.w[3 * seq], n0 = sp.w[3 * seq + 1], row1 = sp.w[3 * seq + 2];
  const uint32_t split = n0 * upr, total = L * upr;      // units in the first range / in all
  const u32x4* s0 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row0) * upr;
  const u32x4* s1 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row1) * upr - split;
  u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq) * total;
  const uint32_t base = blockIdx.x * (blockDim.x * U) + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint32_t u = base + j * blockDim.x;
    if (u < total) v[j] = ld<NT>((u < split ? s0 : s1) + u);
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint32_t u = base + j * blockDim.x;
    if (u < total) st<NT>(dst + u, v[j]);
  }
}

// ---- persistent: gridDim.x workgroups walk the sequences' unit runs with the
// next tile's loads issued before this tile's stores -----------------------
template <int U, int NT>
__global__ void gather_persist(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                               uint32_t upr, uint32_t L, uint32_t nseq, const Spans sp) {
  const uint32_t total = L * upr;
  const uint32_t tile = blockDim.x * U;
  const uint32_t tiles_per_seq = (total + tile - 1) / tile;
  const uint32_t ntiles = tiles_per_seq * nseq;
  u32x4 cur[U], nxt[U];
  uint32_t i = blockIdx.x;
  auto issue = [&](uint32_t ti, u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    const uint32_t row0 = sp.w[3 * seq], n0 = sp.w[3 * seq + 1], row1 = sp.w[3 * seq + 2];
    const uint32_t split = n0 * upr;
    const u32x4* s0 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row0) * upr;
    const u32x4* s1 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row1) * upr - split;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      if (u < total) v[j] = ld<NT>((u < split ? s0 : s1) + u);
    }
  };
  auto put = [&](uint32_t ti, const u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq) * total;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      if (u < total) st<NT>(dst + u, v[j]);
    }
  };
  if (i >= ntiles) return;
  issue(i, cur);
  for (;;) {
    const uint32_t n = i + gridDim.x;
    if (n < ntiles) issue(n, nxt);
    put(i, cur);
    if (n >= ntiles) break;
#pragma unroll
    for (int j = 0; j < U; ++j) cur[j] = nxt[j];
    i = n;
  }
}

// ---- persistent, branch-free loads: out-of-range lanes re-read the last unit
// (clamped index) so that every load is unconditional and the compiler can wait
// for exactly the older tile (vmcnt(U)) before storing it -------------------
template <int U, int NT>
__global__ void gather_persist2(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                                uint32_t upr, uint32_t L, uint32_t nseq, const Spans sp) {
  const uint32_t total = L * upr;
  const uint32_t tile = blockDim.x * U;
  const uint32_t tiles_per_seq = (total + tile - 1) / tile;
  const uint32_t ntiles = tiles_per_seq * nseq;
  u32x4 cur[U], nxt[U];
  uint32_t i = blockIdx.x;
  auto issue = [&](uint32_t ti, u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    const uint32_t row0 = sp.w[3 * seq], n0 = sp.w[3 * seq + 1], row1 = sp.w[3 * seq + 2];
    const uint32_t split = n0 * upr;
    const u32x4* s0 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row0) * upr;
    const u32x4* s1 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row1) * upr - split;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      u = u < total ? u : total - 1;
      v[j] = ld<NT>((u < split ? s0 : s1) + u);
    }
  };
  auto put = [&](uint32_t ti, const u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq) * total;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      if (u < total) st<NT>(dst + u, v[j]);
    }
  };
  if (i >= ntiles) return;
  issue(i, cur);
  while (i + gridDim.x < ntiles) {           // steady state: the next tile always exists
    issue(i + gridDim.x, nxt);
    put(i, cur);
#pragma unroll
    for (int j = 0; j < U; ++j) cur[j] = nxt[j];
    i += gridDim.x;
  }
  put(i, cur);
}

// ---- the same with unconditional (clamped) stores as well: out-of-range
// lanes rewrite the last unit with the value they re-read ------------------
template <int U, int NT>
__global__ void gather_persist3(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                                uint32_t upr, uint32_t L, uint32_t nseq, const Spans sp) {
  const uint32_t total = L * upr;
  const uint32_t tile = blockDim.x * U;
  const uint32_t tiles_per_seq = (total + tile - 1) / tile;
  const uint32_t ntiles = tiles_per_seq * nseq;
  u32x4 cur[U], nxt[U];
  uint32_t i = blockIdx.x;
  auto issue = [&](uint32_t ti, u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    const uint32_t row0 = sp.w[3 * seq], n0 = sp.w[3 * seq + 1], row1 = sp.w[3 * seq + 2];
    const uint32_t split = n0 * upr;
    const u32x4* s0 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row0) * upr;
    const u32x4* s1 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row1) * upr - split;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      u = u < total ? u : total - 1;
      v[j] = ld<NT>((u < split ? s0 : s1) + u);
    }
  };
  auto put = [&](uint32_t ti, const u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq) * total;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      u = u < total ? u : total - 1;
      st<NT>(dst + u, v[j]);
    }
  };
  if (i >= ntiles) return;
  issue(i, cur);
  while (i + gridDim.x < ntiles) {           // steady state: the next tile always exists
    issue(i + gridDim.x, nxt);
    put(i, cur);
#pragma unroll
    for (int j = 0; j < U; ++j) cur[j] = nxt[j];
    i += gridDim.x;
  }
  put(i, cur);
}

// ---- persistent, two register sets used in turn (no copy between them, so
// nothing forces a wait on the newer tile's loads), loads branch-free --------
template <int U, int NT, bool kClampStores>
__global__ void gather_persist4(const uint8_t* __restrict__ pool, uint8_t* __restrict__ batch,
                                uint32_t upr, uint32_t L, uint32_t nseq, const Spans sp) {
  const uint32_t total = L * upr;
  const uint32_t tile = blockDim.x * U;
  const uint32_t tiles_per_seq = (total + tile - 1) / tile;
  const uint32_t ntiles = tiles_per_seq * nseq;
  const uint32_t stride = gridDim.x;
  u32x4 a[U], b[U];
  auto issue = [&](uint32_t ti, u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    const uint32_t row0 = sp.w[3 * seq], n0 = sp.w[3 * seq + 1], row1 = sp.w[3 * seq + 2];
    const uint32_t split = n0 * upr;
    const u32x4* s0 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row0) * upr;
    const u32x4* s1 = reinterpret_cast<const u32x4*>(pool) + static_cast<uint64_t>(row1) * upr - split;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      u = u < total ? u : total - 1;
      v[j] = ld<NT>((u < split ? s0 : s1) + u);
    }
  };
  auto put = [&](uint32_t ti, const u32x4* v) {
    const uint32_t seq = ti / tiles_per_seq, k = ti - seq * tiles_per_seq;
    u32x4* dst = reinterpret_cast<u32x4*>(batch) + static_cast<uint64_t>(seq) * total;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      uint32_t u = k * tile + j * blockDim.x + threadIdx.x;
      if (kClampStores) { u = u < total ? u : total - 1; st<NT>(dst + u, v[j]); }
      else if (u < total) st<NT>(dst + u, v[j]);
    }
  };
  uint32_t i = blockIdx.x;
  if (i >= ntiles) return;
  issue(i, a);
  for (;;) {
    uint32_t n = i + stride;
    if (n >= ntiles) { put(i, a); return; }
    issue(n, b);
    put(i, a);
    i = n;
    n = i + stride;
    if (n >= ntiles) { put(i, b); return; }
    issue(n, a);
    put(i, b);
    i = n;
  }
}

__global__ void tiny_kernel(float* p) { p[threadIdx.x] += 1.f; }

// ---------------------------------------------------------------------------

struct Launcher {
  std::string name;
  std::function<void(int it, hipStream_t, hipEvent_t, hipEvent_t)> go;
};

int main(int argc, char** argv) {
  const int B = argc > 1 ? std::atoi(argv[1]) : 16;
  const int iters = argc > 2 ? std::atoi(argv[2]) : 200;
  const int ring = argc > 3 ? std::atoi(argv[3]) : 4;
  const uint32_t L = 65, rowbytes = 28224, upr = rowbytes / 16;
  const uint32_t chunk = 1024, nrows_pool = 100352;   // 98 chunks of 1024 rows
  const size_t pool_bytes = static_cast<size_t>(nrows_pool) * rowbytes;
  const size_t batch_bytes = static_cast<size_t>(B) * L * rowbytes;
  if (B > kMaxSeq) { std::fprintf(stderr, "B <= %d\n", kMaxSeq); return 1; }
  uint8_t* pool;
  CHECK(hipMalloc(&pool, pool_bytes));
  CHECK(hipMemset(pool, 1, pool_bytes));
  std::vector<uint8_t*> out(ring);
  for (auto& o : out) { CHECK(hipMalloc(&o, batch_bytes)); CHECK(hipMemset(o, 0, batch_bytes)); }
  float* tiny;
  CHECK(hipMalloc(&tiny, 4096));
  CHECK(hipMemset(tiny, 0, 4096));
  hipStream_t stream;
  CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

  // Random windows like Replay.sample: a start row anywhere in a chunk; windows
  // that cross the chunk end continue at the start of another chunk.
  const int nsets = 64;
  std::mt19937 rng(0);
  std::vector<Spans> sets(nsets);
  uint32_t* sets_dev;
  CHECK(hipMalloc(&sets_dev, sizeof(Spans) * nsets));
  for (auto& s : sets) {
    for (int b = 0; b < B; ++b) {
      const uint32_t c = rng() % (nrows_pool / chunk), i = rng() % chunk;
      const uint32_t n0 = std::min(L, chunk - i);
      s.w[3 * b] = c * chunk + i;
      s.w[3 * b + 1] = n0;
      s.w[3 * b + 2] = (rng() % (nrows_pool / chunk)) * chunk;
    }
  }
  CHECK(hipMemcpy(sets_dev, sets.data(), sizeof(Spans) * nsets, hipMemcpyHostToDevice));

  std::vector<Launcher> ls;
  const size_t nslots = pool_bytes / batch_bytes;
  const uint32_t units = static_cast<uint32_t>(batch_bytes / 16);
#define ADD_COPY(U_, NT_, TH_)                                                                   \
  ls.push_back({"copy_flat U" #U_ " NT" #NT_ " T" #TH_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    const uint8_t* src = pool + (static_cast<size_t>(it * 7919u) % nslots) * batch_bytes; \
    hipExtLaunchKernelGGL((copy_flat<U_, NT_>), dim3((units + TH_ * U_ - 1) / (TH_ * U_)), dim3(TH_), 0, s, a, b, 0, \
                          reinterpret_cast<const u32x4*>(src), reinterpret_cast<u32x4*>(out[it % ring]), units); }})
  ADD_COPY(1, 0, 256); ADD_COPY(2, 0, 256); ADD_COPY(2, 3, 256); ADD_COPY(4, 0, 256); ADD_COPY(4, 3, 256);
  ADD_COPY(8, 3, 256); ADD_COPY(4, 3, 512); ADD_COPY(4, 3, 1024); ADD_COPY(2, 3, 1024);
#define ADD_GRID(U_, NT_, TH_)                                                                   \
  ls.push_back({"gather_grid U" #U_ " NT" #NT_ " T" #TH_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    hipExtLaunchKernelGGL((gather_grid<U_, NT_>), dim3((upr + TH_ * U_ - 1) / (TH_ * U_), L, B), dim3(TH_), 0, s, a, b, 0, \
                          pool, out[it % ring], upr, sets[it % nsets]); }})
  ADD_GRID(1, 3, 256); ADD_GRID(2, 3, 256); ADD_GRID(2, 0, 256); ADD_GRID(4, 3, 256); ADD_GRID(7, 3, 256);
  ADD_GRID(2, 3, 128); ADD_GRID(4, 3, 64); ADD_GRID(7, 3, 64); ADD_GRID(4, 3, 448); ADD_GRID(2, 3, 896);
  ADD_GRID(4, 0, 448); ADD_GRID(7, 0, 256);
#define ADD_GRIDP(U_, NT_, TH_)                                                                  \
  ls.push_back({"gather_grid_ptr U" #U_ " NT" #NT_ " T" #TH_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    hipExtLaunchKernelGGL((gather_grid_ptr<U_, NT_>), dim3((upr + TH_ * U_ - 1) / (TH_ * U_), L, B), dim3(TH_), 0, s, a, b, 0, \
                          pool, out[it % ring], upr, sets_dev + (it % nsets) * (3 * kMaxSeq)); }})
  ADD_GRIDP(2, 3, 256); ADD_GRIDP(4, 3, 448);
  const uint32_t seq_units = L * upr;
#define ADD_SPAN(U_, NT_, TH_)                                                                   \
  ls.push_back({"gather_span U" #U_ " NT" #NT_ " T" #TH_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    hipExtLaunchKernelGGL((gather_span<U_, NT_>), dim3((seq_units + TH_ * U_ - 1) / (TH_ * U_), B), dim3(TH_), 0, s, a, b, 0, \
                          pool, out[it % ring], upr, L, sets[it % nsets]); }})
  ADD_SPAN(1, 3, 256); ADD_SPAN(2, 3, 256); ADD_SPAN(2, 0, 256); ADD_SPAN(4, 3, 256); ADD_SPAN(8, 3, 256);
  ADD_SPAN(4, 3, 512); ADD_SPAN(4, 3, 1024); ADD_SPAN(2, 3, 1024); ADD_SPAN(4, 0, 256); ADD_SPAN(4, 1, 256); ADD_SPAN(4, 2, 256);
#define ADD_PERSIST(U_, NT_, TH_, W_)                                                            \
  ls.push_back({"gather_persist U" #U_ " NT" #NT_ " T" #TH_ " W" #W_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    hipExtLaunchKernelGGL((gather_persist<U_, NT_>), dim3(256 * W_), dim3(TH_), 0, s, a, b, 0, \
                          pool, out[it % ring], upr, L, static_cast<uint32_t>(B), sets[it % nsets]); }})
#define ADD_PERSIST2(K_, U_, NT_, TH_, W_)                                                       \
  ls.push_back({#K_ " U" #U_ " NT" #NT_ " T" #TH_ " W" #W_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    hipExtLaunchKernelGGL((K_<U_, NT_>), dim3(256 * W_), dim3(TH_), 0, s, a, b, 0, \
                          pool, out[it % ring], upr, L, static_cast<uint32_t>(B), sets[it % nsets]); }})
  ADD_PERSIST2(gather_persist2, 4, 3, 512, 2); ADD_PERSIST2(gather_persist2, 4, 3, 256, 4); ADD_PERSIST2(gather_persist2, 2, 3, 256, 8);
  ADD_PERSIST2(gather_persist2, 2, 3, 512, 4); ADD_PERSIST2(gather_persist2, 4, 3, 256, 3); ADD_PERSIST2(gather_persist2, 4, 3, 256, 6);
  ADD_PERSIST2(gather_persist3, 4, 3, 512, 2); ADD_PERSIST2(gather_persist3, 4, 3, 256, 4); ADD_PERSIST2(gather_persist3, 2, 3, 256, 8);
  ADD_PERSIST2(gather_persist3, 8, 3, 256, 2); ADD_PERSIST2(gather_persist3, 8, 3, 512, 1); ADD_PERSIST2(gather_persist3, 4, 3, 1024, 1);
#define ADD_PERSIST4(C_, U_, NT_, TH_, W_)                                                       \
  ls.push_back({"gather_persist4 clamp" #C_ " U" #U_ " NT" #NT_ " T" #TH_ " W" #W_, [=](int it, hipStream_t s, hipEvent_t a, hipEvent_t b) { \
    hipExtLaunchKernelGGL((gather_persist4<U_, NT_, C_>), dim3(256 * W_), dim3(TH_), 0, s, a, b, 0, \
                          pool, out[it % ring], upr, L, static_cast<uint32_t>(B), sets[it % nsets]); }})
  ADD_PERSIST4(false, 4, 3, 512, 2); ADD_PERSIST4(false, 4, 3, 256, 4); ADD_PERSIST4(false, 2, 3, 256, 8); ADD_PERSIST4(false, 2, 3, 256, 4);
  ADD_PERSIST4(true, 4, 3, 512, 2); ADD_PERSIST4(true, 4, 3, 256, 4); ADD_PERSIST4(true, 2, 3, 256, 8); ADD_PERSIST4(true, 2, 3, 256, 4);
  ADD_PERSIST4(true, 4, 3, 256, 2); ADD_PERSIST4(true, 8, 3, 256, 2); ADD_PERSIST4(true, 4, 3, 256, 3); ADD_PERSIST4(true, 2, 3, 512, 4);
  ADD_PERSIST4(true, 4, 3, 128, 8); ADD_PERSIST4(true, 4, 3, 64, 16); ADD_PERSIST4(true, 2, 3, 128, 16); ADD_PERSIST4(true, 4, 3, 1024, 1);
  ADD_PERSIST(2, 3, 256, 4); ADD_PERSIST(2, 3, 256, 8); ADD_PERSIST(4, 3, 256, 2); ADD_PERSIST(4, 3, 256, 4);
  ADD_PERSIST(4, 3, 512, 2); ADD_PERSIST(2, 3, 1024, 1); ADD_PERSIST(4, 3, 1024, 1); ADD_PERSIST(4, 0, 256, 4);

  const int batchn = 50;
  std::vector<hipEvent_t> ev(2 * batchn);
  for (auto& e : ev) CHECK(hipEventCreate(&e));
  hipEvent_t w0, w1;
  CHECK(hipEventCreate(&w0));
  CHECK(hipEventCreate(&w1));
  const double mb = 2.0 * batch_bytes / 1e6;
  std::printf("B=%d L=%u rowbytes=%u  r+w bytes per launch %.2f MB  ring=%d iters=%d\n", B, L, rowbytes, mb, ring, iters);
  std::printf("%-34s %9s %9s %9s | %9s | %9s | %9s\n", "variant", "A mean", "A med", "A min", "B thru", "C mean", "D idle");
  for (auto& l : ls) {
    for (int i = 0; i < 10; ++i) l.go(i, stream, nullptr, nullptr);
    CHECK(hipStreamSynchronize(stream));
    // A: stamped, tight loop
    std::vector<float> ts;
    for (int done = 0; done < iters; done += batchn) {
      for (int i = 0; i < batchn; ++i) l.go(done + i, stream, ev[2 * i], ev[2 * i + 1]);
      CHECK(hipStreamSynchronize(stream));
      for (int i = 0; i < batchn; ++i) { float ms; CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); ts.push_back(ms * 1e3f); }
    }
    std::sort(ts.begin(), ts.end());
    double meanA = 0; for (float t : ts) meanA += t; meanA /= ts.size();
    const float medA = ts[ts.size() / 2], minA = ts[0];
    // B: throughput
    CHECK(hipEventRecord(w0, stream));
    for (int i = 0; i < iters; ++i) l.go(i, stream, nullptr, nullptr);
    CHECK(hipEventRecord(w1, stream));
    CHECK(hipStreamSynchronize(stream));
    float msB; CHECK(hipEventElapsedTime(&msB, w0, w1));
    // C: stamped with a tiny kernel between
    std::vector<float> tc;
    for (int done = 0; done < iters; done += batchn) {
      for (int i = 0; i < batchn; ++i) {
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny);
        l.go(done + i, stream, ev[2 * i], ev[2 * i + 1]);
      }
      CHECK(hipStreamSynchronize(stream));
      for (int i = 0; i < batchn; ++i) { float ms; CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tc.push_back(ms * 1e3f); }
    }
    double meanC = 0; for (float t : tc) meanC += t; meanC /= tc.size();
    // D: stamped, GPU idle between launches (sync + 15 us of host spinning)
    double meanD = 0;
    {
      const int nd = std::min(iters, 100);
      for (int i = 0; i < nd; ++i) {
        l.go(i, stream, ev[0], ev[1]);
        CHECK(hipStreamSynchronize(stream));
        float ms; CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        meanD += ms * 1e3;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 15.0) {}
      }
      meanD /= nd;
    }
    std::printf("%-34s %9.2f %9.2f %9.2f | %9.2f | %9.2f | %9.2f  (A: %.0f GB/s r+w)\n", l.name.c_str(), meanA, medA, minA,
                msB * 1e3 / iters, meanC, meanD, mb / meanA * 1e3);
    std::fflush(stdout);
  }

  // ---- pipeline study: what a kernel BEFORE the gather does to the gather's
  // stamped window.  pre = what runs just before each stamped launch.
  {
    hipStream_t other;
    CHECK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
    hipEvent_t pe0, pe1, marker;
    CHECK(hipEventCreate(&pe0)); CHECK(hipEventCreate(&pe1));
    CHECK(hipEventCreateWithFlags(&marker, hipEventDisableTiming));
    uint8_t* big_a; uint8_t* big_b;
    CHECK(hipMalloc(&big_a, 8 << 20)); CHECK(hipMalloc(&big_b, 8 << 20));
    const char* pres[] = {"none", "tiny", "tiny stamped", "tiny stop-only", "tiny + event record",
                          "tiny on other stream", "8MB copy kernel", "two tiny", "tiny, then gather twice (2nd)"};
    std::printf("\n%-34s", "pipeline study (stamped mean us)");
    for (auto* p : pres) std::printf(" | %s", p);
    std::printf("\n");
    for (auto& l : ls) {
      if (l.name != "copy_flat U4 NT3 T256" && l.name != "gather_persist U4 NT3 T512 W2" &&
          l.name != "gather_span U1 NT3 T256") continue;
      std::printf("%-34s", l.name.c_str());
      for (int pre = 0; pre < 9; ++pre) {
        std::vector<float> tc;
        for (int done = 0; done < iters; done += batchn) {
          for (int i = 0; i < batchn; ++i) {
            switch (pre) {
              case 0: break;
              case 1: hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny); break;
              case 2: hipExtLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, pe0, pe1, 0, tiny); break;
              case 3: hipExtLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, nullptr, pe1, 0, tiny); break;
              case 4: hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny);
                      CHECK(hipEventRecord(marker, stream)); break;
              case 5: hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, other, tiny); break;
              case 6: hipLaunchKernelGGL((copy_flat<4, 3>), dim3((8 << 20) / 16 / 1024), dim3(256), 0, stream,
                                         reinterpret_cast<const u32x4*>(big_a), reinterpret_cast<u32x4*>(big_b), (8u << 20) / 16); break;
              case 7: hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny);
                      hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny); break;
              case 8: hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny);
                      l.go(done + i + 17, stream, nullptr, nullptr); break;
            }
            l.go(done + i, stream, ev[2 * i], ev[2 * i + 1]);
          }
          CHECK(hipStreamSynchronize(stream));
          CHECK(hipStreamSynchronize(other));
          for (int i = 0; i < batchn; ++i) { float ms; CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tc.push_back(ms * 1e3f); }
        }
        double m = 0; for (float t : tc) m += t; m /= tc.size();
        std::printf(" | %6.2f", m);
      }
      std::printf("\n");
      // wall-clock cost per (tiny + gather) pair vs gather alone, unstamped
      for (int pre = 0; pre < 2; ++pre) {
        CHECK(hipEventRecord(w0, stream));
        for (int i = 0; i < iters; ++i) {
          if (pre) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, stream, tiny);
          l.go(i, stream, nullptr, nullptr);
        }
        CHECK(hipEventRecord(w1, stream));
        CHECK(hipStreamSynchronize(stream));
        float msw; CHECK(hipEventElapsedTime(&msw, w0, w1));
        std::printf("    unstamped wall per iteration, %s: %.2f us\n", pre ? "tiny + gather" : "gather only", msw * 1e3 / iters);
      }
    }
  }

  // hipMemcpyAsync D2D of the same bytes (throughput only)
  CHECK(hipEventRecord(w0, stream));
  for (int i = 0; i < iters; ++i) CHECK(hipMemcpyAsync(out[i % ring], pool + (i % nslots) * batch_bytes, batch_bytes, hipMemcpyDeviceToDevice, stream));
  CHECK(hipEventRecord(w1, stream));
  CHECK(hipStreamSynchronize(stream));
  float ms; CHECK(hipEventElapsedTime(&ms, w0, w1));
  std::printf("%-34s %9s %9s %9s | %9.2f |\n", "hipMemcpyAsync D2D", "-", "-", "-", ms * 1e3 / iters);
  return 0;
}
