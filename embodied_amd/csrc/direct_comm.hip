// emb_direct_*: the two collectives of a train step -- the gradient all-reduce
// (embodied/jax/opt.py:52-54) and the DP-slice all-to-all
// (embodied/jax/internal.py:145-152) -- as a DIRECT schedule over xGMI.
//
// xGMI is point to point: every GPU of a node has its own link to every other
// (7 x ~77 GB/s each way on MI355X).  A ring moves 2(n-1)/n * G bytes through ONE
// link per rank; here every rank writes to all n-1 peers AT ONCE, each peer's
// share over its own link:
//
//   all-reduce  = reduce-scatter + all-gather, both "push":
//     1. push:    rank r stores shard p of its buffer into peer p's inbox slot r
//                 (n-1 concurrent peer stores), then raises p's rs flag;
//     2. reduce:  r waits for the n-1 rs flags, sums shard r over the ranks IN
//                 RANK ORDER in f32 (every element is reduced once, by one rank:
//                 all ranks end with the same bits), scales for a mean, and
//                 stores the result into its own buffer and into every peer's
//                 gather slot r, then raises their ag flags;
//     3. collect: r waits for the n-1 ag flags and copies the peers' reduced
//                 shards from its gather slots into its buffer.
//     Link traffic per rank and direction: 2(n-1)/n * G, over n-1 links.
//   all-to-all  = 1. push block p into peer p's slot r + flag; 2. wait, copy the
//                 slots into the receive buffer.
//
// Peers' memory is reached through hipIpc handles of ONE uncached (fine-grained)
// allocation per rank (header with the flags + double-buffered slots), so the
// transport needs nothing but the HIP runtime; flags are 32-bit sequence
// numbers written with system-scope release stores behind a system-scope fence
// and read with acquire loads.  Slot reuse is safe with two buffers: a rank
// that starts operation s has seen every peer's flags of operation s-1, i.e.
// every peer has finished reading the slots of operation s-2.
//
//   all-gather  = the all-to-all with the SAME block pushed to every peer (the
//                 trajectory all-gather of embodied/jax/internal.py:145-152).
//
// Every wait is bounded (timeout_ms of emb_direct_create): a peer that never
// arrives turns into an error word, never into a hung GPU -- and the error is
// FATAL for the communicator: the kernel that gave up writes no result and
// raises no peer's flag (so every rank runs into the same time-out instead of
// reducing half-arrived slots), every later kernel of the communicator returns
// at once, and every later host call fails (the word also lives in host-mapped
// memory that the entry points read without synchronising).
#include "abi_common.h"

#include <hip/hip_runtime.h>

using namespace emb_abi;

namespace {

constexpr int kMaxRanks = 8;
constexpr int kThreads = 256;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct alignas(256) DirectHeader {
  uint32_t rs_flag[kMaxRanks];      // written by the peers
  uint32_t ag_flag[kMaxRanks];
  uint32_t a2a_flag[kMaxRanks];
  uint32_t error;                   // local, sticky: a wait ran into its timeout (the communicator is dead)
  uint32_t counter[3][kMaxRanks];   // local: blocks that finished pushing to peer p (rs, ag, a2a)
};

__device__ __forceinline__ void signal(uint32_t* flag, uint32_t seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Blocks of one launch that push to the same peer: the last one to finish (its
// stores fenced at system scope) raises the peer's flag.
__device__ __forceinline__ void arrive(uint32_t* counter, uint32_t expected, uint32_t* peer_flag, uint32_t seq,
                                       const uint32_t* error) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t seen = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (seen + 1 == expected) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      // (a dead communicator raises no flags: its peers give up as well)
      if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) signal(peer_flag, seq);
    }
  }
}

// Wait until flags[p] has reached `seq` for every peer p != me (wrap-safe), at
// most `timeout` ticks of the 100 MHz wall clock; then make the peers' stores
// visible to every lane of the workgroup.  False (for the whole workgroup) if a
// peer did not arrive in time, now or in an earlier operation: the caller
// writes nothing and signals nobody.
struct ErrorWords {
  uint32_t* device;                   // DirectHeader::error
  uint32_t* host;                     // the same word in host-mapped memory (read by the entry points)
};

__device__ __forceinline__ bool wait_all(const uint32_t* flags, int me, int world, uint32_t seq,
                                         uint64_t timeout, ErrorWords error) {
  int failed = 0;
  if (threadIdx.x < static_cast<unsigned>(world) && static_cast<int>(threadIdx.x) != me) {
    if (__hip_atomic_load(error.device, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      failed = 1;
    } else {
      const uint64_t began = wall_clock64();
      for (;;) {
        const uint32_t seen = __hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (static_cast<int32_t>(seen - seq) >= 0) break;
        if (wall_clock64() - began > timeout) {
          __hip_atomic_store(error.device, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(error.host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  failed = __syncthreads_or(failed);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: drops what this CU had cached of the slots
  return failed == 0;
}

struct PushArgs {
  const uint8_t* src[kMaxRanks];
  uint8_t* dst[kMaxRanks];
  uint32_t* flag[kMaxRanks];          // in the PEER's header
  int64_t bytes[kMaxRanks];
  int32_t n, per_peer;                // peers listed, workgroups per peer
  uint32_t seq;
  uint32_t* counter;                  // local header: counter[which][0..]
  const uint32_t* error;              // local header
};

// Bytes of any alignment: 16-byte body, byte tail.
__device__ __forceinline__ void copy_span(const uint8_t* src, uint8_t* dst, int64_t bytes, int part, int parts) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t vecs = aligned ? bytes >> 4 : 0;
  for (int64_t i = static_cast<int64_t>(part) * kThreads + threadIdx.x; i < vecs;
       i += static_cast<int64_t>(parts) * kThreads)
    reinterpret_cast<u32x4*>(dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + i);
  for (int64_t i = (vecs << 4) + static_cast<int64_t>(part) * kThreads + threadIdx.x; i < bytes;
       i += static_cast<int64_t>(parts) * kThreads)
    dst[i] = src[i];
}

__global__ __launch_bounds__(kThreads) void direct_push_kernel(const PushArgs a) {
  const int p = blockIdx.x / a.per_peer, part = blockIdx.x - p * a.per_peer;
  copy_span(a.src[p], a.dst[p], a.bytes[p], part, a.per_peer);
  arrive(a.counter + p, static_cast<uint32_t>(a.per_peer), a.flag[p], a.seq, a.error);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
struct Bf16 { uint16_t bits; };
struct F16 { _Float16 v; };
template <> __device__ __forceinline__ float to_f32<Bf16>(Bf16 v) { return __uint_as_float(uint32_t{v.bits} << 16); }
template <> __device__ __forceinline__ float to_f32<F16>(F16 v) { return static_cast<float>(v.v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ Bf16 from_f32<Bf16>(float v) {
  // round to nearest even, NaN kept quiet
  const uint32_t u = __float_as_uint(v);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return Bf16{static_cast<uint16_t>((u >> 16) | 0x40u)};
  return Bf16{static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16)};
}
template <> __device__ __forceinline__ F16 from_f32<F16>(float v) { return F16{static_cast<_Float16>(v)}; }

struct ReduceArgs {
  uint8_t* local;                     // this rank's buffer (whole)
  const uint8_t* inbox[kMaxRanks];    // my rs slots, one per source rank (inbox[me] unused)
  uint8_t* gather[kMaxRanks];         // peer p's ag slot for ME (gather[me] = nullptr)
  uint32_t* flag[kMaxRanks];          // peer p's ag flag for me
  const uint32_t* rs_flags;           // my header
  ErrorWords error;
  uint32_t* counter;                  // local header: ag counters
  int64_t shard_off, shard_elems;     // my shard, in elements
  int32_t me, world;
  uint32_t seq;
  float scale;
  uint64_t timeout;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void direct_reduce_kernel(const ReduceArgs a) {
  if (!wait_all(a.rs_flags, a.me, a.world, a.seq, a.timeout, a.error)) return;
  constexpr int V = 16 / static_cast<int>(sizeof(T));      // elements per 16-byte access
  union Pack { u32x4 raw; T e[V]; };
  uint8_t* mine = a.local + a.shard_off * static_cast<int64_t>(sizeof(T));
  const int64_t vecs = a.shard_elems / V;
  // 16 bytes per lane from every rank's copy of the shard, summed IN RANK ORDER in
  // f32 (the same sum whoever computes it), 16 bytes per lane to every rank.
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < vecs;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int p = 0; p < a.world; ++p) {
      Pack in;
      in.raw = reinterpret_cast<const u32x4*>(p == a.me ? mine : a.inbox[p])[i];
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += to_f32<T>(in.e[k]);
    }
    Pack out;
#pragma unroll
    for (int k = 0; k < V; ++k) out.e[k] = from_f32<T>(acc[k] * a.scale);
    reinterpret_cast<u32x4*>(mine)[i] = out.raw;
    for (int p = 0; p < a.world; ++p)
      if (p != a.me) reinterpret_cast<u32x4*>(a.gather[p])[i] = out.raw;
  }
  // (the last shard of a count that is no multiple of 16 bytes)
  for (int64_t i = vecs * V + static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.shard_elems;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    float acc = 0.f;
    for (int p = 0; p < a.world; ++p)
      acc += to_f32<T>(reinterpret_cast<const T*>(p == a.me ? mine : a.inbox[p])[i]);
    const T v = from_f32<T>(acc * a.scale);
    reinterpret_cast<T*>(mine)[i] = v;
    for (int p = 0; p < a.world; ++p)
      if (p != a.me) reinterpret_cast<T*>(a.gather[p])[i] = v;
  }
  // every workgroup has stored to every peer: one counter, one round of flags
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t seen = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (seen + 1 == gridDim.x) {
      __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int p = 0; p < a.world; ++p)
        if (p != a.me) signal(a.flag[p], a.seq);
    }
  }
}

struct CollectArgs {
  const uint8_t* slot[kMaxRanks];     // my slots, one per source rank
  uint8_t* dst[kMaxRanks];            // where source p's bytes go
  int64_t bytes[kMaxRanks];
  const uint8_t* own_src;             // all-to-all: my own block, copied locally (may be null)
  uint8_t* own_dst;
  int64_t own_bytes;
  const uint32_t* flags;
  ErrorWords error;
  int32_t me, world;
  uint32_t seq;
  uint64_t timeout;
};

__global__ __launch_bounds__(kThreads) void direct_collect_kernel(const CollectArgs a) {
  if (a.own_bytes) copy_span(a.own_src, a.own_dst, a.own_bytes, blockIdx.x, gridDim.x);    // needs no peer
  if (!wait_all(a.flags, a.me, a.world, a.seq, a.timeout, a.error)) return;
  for (int p = 0; p < a.world; ++p)
    if (p != a.me) copy_span(a.slot[p], a.dst[p], a.bytes[p], blockIdx.x, gridDim.x);
}

int elem_size(int32_t dtype) { return dtype == EMB_F32 ? 4 : 2; }

}  // namespace

// (a std::exception that is none of guarded()'s named kinds: EMB_ERR_INTERNAL, a
// RuntimeError in Python -- not the ValueError of a bad argument)
struct DeadTransport : std::exception {
  std::string msg;
  explicit DeadTransport(std::string m) : msg(std::move(m)) {}
  const char* what() const noexcept override { return msg.c_str(); }
};

struct emb_direct {
  int32_t rank = 0, world = 1;
  int64_t reduce_cap = 0, a2a_cap = 0;      // bytes: whole gradient buffer / one all-to-all block
  int64_t shard_cap = 0;                    // bytes of one reduce shard slot
  size_t region_bytes = 0;
  uint8_t* region = nullptr;                // mine
  uint8_t* peer[kMaxRanks] = {};            // everybody's (peer[rank] = region)
  bool connected = false;
  uint32_t red_seq = 0, a2a_seq = 0;
  uint64_t timeout_ticks = 0;
  uint32_t* host_error = nullptr;           // host-mapped twin of DirectHeader::error
  uint32_t* host_error_dev = nullptr;       // its device address
  hipStream_t side = nullptr;
  hipEvent_t forked = nullptr, done = nullptr;
  bool in_flight = false;

  DirectHeader* header(int r) const { return reinterpret_cast<DirectHeader*>(peer[r]); }
  ErrorWords errors() const { return ErrorWords{&header(rank)->error, host_error_dev}; }
  // A wait inside one of the kernels gave up on a peer (read without synchronising):
  // nothing this communicator has produced since can be trusted.
  void alive(const char* what) const {
    if (host_error && *static_cast<volatile uint32_t*>(host_error) != 0u)
      throw DeadTransport(std::string(what) + ": the direct transport timed out waiting for a peer; "
                          "this communicator is dead (no result was written by the operation that gave up)");
  }
  // slots of rank r's region: rs[parity][src], ag[parity][src], a2a[parity][src]
  uint8_t* rs_slot(int r, int parity, int src) const {
    return peer[r] + sizeof(DirectHeader) + (static_cast<int64_t>(parity) * world + src) * shard_cap;
  }
  uint8_t* ag_slot(int r, int parity, int src) const {
    return peer[r] + sizeof(DirectHeader) + (static_cast<int64_t>(2 + parity) * world + src) * shard_cap;
  }
  uint8_t* a2a_slot(int r, int parity, int src) const {
    return peer[r] + sizeof(DirectHeader) + 4ll * world * shard_cap + (static_cast<int64_t>(parity) * world + src) * a2a_cap;
  }
};

static int64_t round_up(int64_t v, int64_t to) { return (v + to - 1) / to * to; }

static void direct_allreduce(emb_direct* d, void* buf, int64_t count, int32_t dtype, bool mean, hipStream_t s) {
  need(dtype == EMB_F32 || dtype == EMB_BF16 || dtype == EMB_F16, "direct_allreduce: dtype must be f16, bf16 or f32");
  const int es = elem_size(dtype);
  need(count * es <= d->reduce_cap, "direct_allreduce: buffer larger than the capacity given to emb_direct_create");
  const int n = d->world, me = d->rank;
  if (count == 0 || n == 1) return;
  need(d->connected, "direct_allreduce: call emb_direct_connect first");
  // (the reduce and collect kernels move 16 bytes per lane from / to shard offsets of this buffer)
  need(reinterpret_cast<uintptr_t>(buf) % 16 == 0, "direct_allreduce: the buffer must be 16-byte aligned");
  d->alive("direct_allreduce");
  const uint32_t seq = ++d->red_seq;
  const int parity = seq & 1;
  // shards of a multiple of 16 bytes (the last one may be shorter, or empty)
  const int64_t shard = round_up((count + n - 1) / n, 16 / es);
  auto span = [&](int r, int64_t* off, int64_t* elems) {
    *off = std::min<int64_t>(count, shard * r);
    *elems = std::min<int64_t>(count, shard * (r + 1)) - *off;
  };
  uint8_t* local = static_cast<uint8_t*>(buf);
  DirectHeader* mine = d->header(me);
  // 1. push shard p -> peer p's rs slot [me]
  PushArgs push{};
  int64_t most = 0;
  for (int p = 0; p < n; ++p) {
    if (p == me) continue;
    int64_t off, elems;
    span(p, &off, &elems);
    const int k = push.n++;
    push.src[k] = local + off * es;
    push.dst[k] = d->rs_slot(p, parity, me);
    push.flag[k] = &d->header(p)->rs_flag[me];
    push.bytes[k] = elems * es;
    most = std::max<int64_t>(most, elems * es);
  }
  push.per_peer = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(64, most / (kThreads * 16 * 4))));
  push.seq = seq;
  push.counter = mine->counter[0];
  push.error = &mine->error;
  hipLaunchKernelGGL(direct_push_kernel, dim3(push.n * push.per_peer), dim3(kThreads), 0, s, push);
  // 2. reduce my shard, store it here and into every peer's ag slot [me]
  ReduceArgs red{};
  red.local = local;
  span(me, &red.shard_off, &red.shard_elems);
  for (int p = 0; p < n; ++p) {
    red.inbox[p] = d->rs_slot(me, parity, p);
    red.gather[p] = p == me ? nullptr : d->ag_slot(p, parity, me);
    red.flag[p] = p == me ? nullptr : &d->header(p)->ag_flag[me];
  }
  red.rs_flags = mine->rs_flag;
  red.error = d->errors();
  red.counter = &mine->counter[1][0];
  red.me = me;
  red.world = n;
  red.seq = seq;
  red.scale = mean ? 1.f / static_cast<float>(n) : 1.f;
  red.timeout = d->timeout_ticks;
  const int blocks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(256, red.shard_elems / (kThreads * 4))));
  if (dtype == EMB_F32) hipLaunchKernelGGL(direct_reduce_kernel<float>, dim3(blocks), dim3(kThreads), 0, s, red);
  else if (dtype == EMB_BF16) hipLaunchKernelGGL(direct_reduce_kernel<Bf16>, dim3(blocks), dim3(kThreads), 0, s, red);
  else hipLaunchKernelGGL(direct_reduce_kernel<F16>, dim3(blocks), dim3(kThreads), 0, s, red);
  // 3. collect the peers' reduced shards from my ag slots
  CollectArgs col{};
  for (int p = 0; p < n; ++p) {
    int64_t off, elems;
    span(p, &off, &elems);
    col.slot[p] = d->ag_slot(me, parity, p);
    col.dst[p] = local + off * es;
    col.bytes[p] = elems * es;
  }
  col.flags = mine->ag_flag;
  col.error = d->errors();
  col.me = me;
  col.world = n;
  col.seq = seq;
  col.timeout = d->timeout_ticks;
  hipLaunchKernelGGL(direct_collect_kernel, dim3(std::min(256, std::max(1, blocks))), dim3(kThreads), 0, s, col);
  HIP_OK(hipGetLastError());
}

// all-to-all: block p of `send` goes to peer p (gather = false); all-gather: the one
// block `send` holds goes to every peer (gather = true).  Either way block p of
// `recv` came from rank p.
static void direct_alltoall(emb_direct* d, const void* send, void* recv, int64_t bytes, hipStream_t s,
                            bool gather = false) {
  need(bytes <= d->a2a_cap, "direct_alltoall / allgather: block larger than the capacity given to emb_direct_create");
  const int n = d->world, me = d->rank;
  if (bytes == 0) return;
  const uint8_t* from = static_cast<const uint8_t*>(send);
  uint8_t* to = static_cast<uint8_t*>(recv);
  if (n == 1) {
    if (to != from) HIP_OK(hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, s));
    return;
  }
  need(d->connected, "direct_alltoall: call emb_direct_connect first");
  d->alive("direct_alltoall");
  const int64_t stride = gather ? 0 : bytes;
  const uint32_t seq = ++d->a2a_seq;
  const int parity = seq & 1;
  DirectHeader* mine = d->header(me);
  PushArgs push{};
  for (int p = 0; p < n; ++p) {
    if (p == me) continue;
    const int k = push.n++;
    push.src[k] = from + p * stride;
    push.dst[k] = d->a2a_slot(p, parity, me);
    push.flag[k] = &d->header(p)->a2a_flag[me];
    push.bytes[k] = bytes;
  }
  push.per_peer = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(64, bytes / (kThreads * 16 * 4))));
  push.seq = seq;
  push.counter = mine->counter[2];
  push.error = &mine->error;
  hipLaunchKernelGGL(direct_push_kernel, dim3(push.n * push.per_peer), dim3(kThreads), 0, s, push);
  CollectArgs col{};
  for (int p = 0; p < n; ++p) {
    col.slot[p] = d->a2a_slot(me, parity, p);
    col.dst[p] = to + p * bytes;
    col.bytes[p] = bytes;
  }
  col.own_src = from + me * stride;
  col.own_dst = to + me * bytes;
  col.own_bytes = bytes;
  col.flags = mine->a2a_flag;
  col.error = d->errors();
  col.me = me;
  col.world = n;
  col.seq = seq;
  col.timeout = d->timeout_ticks;
  const int blocks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(256, bytes / (kThreads * 16 * 2))));
  hipLaunchKernelGGL(direct_collect_kernel, dim3(blocks), dim3(kThreads), 0, s, col);
  HIP_OK(hipGetLastError());
}

extern "C" {

int32_t emb_direct_create(int32_t rank, int32_t world, int64_t max_reduce_bytes, int64_t max_block_bytes,
                          int32_t timeout_ms, emb_direct_t** out) {
  return guarded([&] {
    need(out && world >= 1 && world <= kMaxRanks && rank >= 0 && rank < world && max_reduce_bytes >= 0 &&
             max_block_bytes >= 0 && timeout_ms > 0, "direct_create: bad arguments (at most 8 ranks: one node)");
    auto d = std::make_unique<emb_direct>();
    d->rank = rank;
    d->world = world;
    d->reduce_cap = max_reduce_bytes;
    d->a2a_cap = round_up(max_block_bytes, 256);
    d->shard_cap = round_up((max_reduce_bytes + world - 1) / world + 16, 256);
    d->region_bytes = sizeof(DirectHeader) + 4ull * world * d->shard_cap + 2ull * world * d->a2a_cap;
    d->timeout_ticks = static_cast<uint64_t>(timeout_ms) * 100000ull;        // wall_clock64: 100 MHz
    // Uncached (fine-grained and never kept in this GPU's L2): stores of a peer GPU
    // become visible to this GPU's loads without a kernel boundary in between, and
    // no line of a slot read two operations ago can be served stale.  Slots are
    // streamed once per operation, so the L2 had nothing to add.  A runtime
    // without the uncached kind gets fine-grained memory, coherent through the
    // system-scope acquire of wait_all.
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&d->region), d->region_bytes, hipDeviceMallocUncached) !=
        hipSuccess) {
      (void)hipGetLastError();
      HIP_OK(hipExtMallocWithFlags(reinterpret_cast<void**>(&d->region), d->region_bytes,
                                   hipDeviceMallocFinegrained));
    }
    HIP_OK(hipMemset(d->region, 0, sizeof(DirectHeader)));
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&d->host_error), 64, hipHostMallocMapped));
    *d->host_error = 0u;
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d->host_error_dev), d->host_error, 0));
    HIP_OK(hipDeviceSynchronize());
    d->peer[rank] = d->region;
    d->connected = world == 1;
    HIP_OK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&d->forked, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&d->done, hipEventDisableTiming));
    *out = d.release();
  });
}

int32_t emb_direct_handle(emb_direct_t* d, uint8_t* handle_out) {
  return guarded([&] {
    need(d && handle_out, "direct_handle: bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == EMB_DIRECT_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipIpcMemHandle_t handle;
    HIP_OK(hipIpcGetMemHandle(&handle, d->region));
    std::memcpy(handle_out, &handle, sizeof(handle));
  });
}

int32_t emb_direct_connect(emb_direct_t* d, const uint8_t* handles) {
  return guarded([&] {
    need(d && (handles || d->world == 1), "direct_connect: bad arguments");
    if (d->connected) return;
    for (int p = 0; p < d->world; ++p) {
      if (p == d->rank) continue;
      hipIpcMemHandle_t handle;
      std::memcpy(&handle, handles + p * EMB_DIRECT_HANDLE_BYTES, sizeof(handle));
      void* ptr = nullptr;
      HIP_OK(hipIpcOpenMemHandle(&ptr, handle, hipIpcMemLazyEnablePeerAccess));
      d->peer[p] = static_cast<uint8_t*>(ptr);
    }
    d->connected = true;
  });
}

int32_t emb_direct_allreduce(emb_direct_t* d, void* buf, int64_t count, int32_t dtype, int32_t mean, void* stream) {
  return guarded([&] {
    need(d && (buf || count == 0) && count >= 0, "direct_allreduce: bad arguments");
    direct_allreduce(d, buf, count, dtype, mean != 0, static_cast<hipStream_t>(stream));
  });
}

int32_t emb_direct_alltoall(emb_direct_t* d, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  return guarded([&] {
    need(d && bytes_per_rank >= 0 && (bytes_per_rank == 0 || (send && recv)), "direct_alltoall: bad arguments");
    direct_alltoall(d, send, recv, bytes_per_rank, static_cast<hipStream_t>(stream));
  });
}

int32_t emb_direct_allgather(emb_direct_t* d, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  return guarded([&] {
    need(d && bytes_per_rank >= 0 && (bytes_per_rank == 0 || (send && recv)), "direct_allgather: bad arguments");
    direct_alltoall(d, send, recv, bytes_per_rank, static_cast<hipStream_t>(stream), true);
  });
}

static int32_t direct_exchange(emb_direct_t* d, void* after_stream, const void* send, void* recv,
                               int64_t bytes_per_rank, void* grads, int64_t count, int32_t dtype, int32_t mean,
                               bool gather) {
  return guarded([&] {
    need(d && bytes_per_rank >= 0 && count >= 0, "direct_exchange: bad arguments");
    need(bytes_per_rank == 0 || (send && recv), "direct_exchange: null slice / trajectory buffers");
    need(count == 0 || grads, "direct_exchange: null gradient buffer");
    if (bytes_per_rank == 0 && count == 0) return;
    d->alive("direct_exchange");
    HIP_OK(hipEventRecord(d->forked, static_cast<hipStream_t>(after_stream)));
    HIP_OK(hipStreamWaitEvent(d->side, d->forked, 0));
    if (bytes_per_rank) direct_alltoall(d, send, recv, bytes_per_rank, d->side, gather);
    if (count) direct_allreduce(d, grads, count, dtype, mean != 0, d->side);
    HIP_OK(hipEventRecord(d->done, d->side));
    d->in_flight = true;
  });
}

int32_t emb_direct_exchange(emb_direct_t* d, void* after_stream, const void* slices_send, void* slices_recv,
                            int64_t bytes_per_rank, void* grads, int64_t count, int32_t dtype, int32_t mean) {
  return direct_exchange(d, after_stream, slices_send, slices_recv, bytes_per_rank, grads, count, dtype, mean, false);
}

int32_t emb_direct_exchange_gather(emb_direct_t* d, void* after_stream, const void* traj_send, void* traj_recv,
                                   int64_t bytes_per_rank, void* grads, int64_t count, int32_t dtype,
                                   int32_t mean) {
  return direct_exchange(d, after_stream, traj_send, traj_recv, bytes_per_rank, grads, count, dtype, mean, true);
}

int32_t emb_direct_wait(emb_direct_t* d, void* stream) {
  return guarded([&] {
    need(d, "direct_wait: null handle");
    d->alive("direct_wait");
    if (!d->in_flight) return;
    HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), d->done, 0));
    d->in_flight = false;
  });
}

int32_t emb_direct_set_timeout(emb_direct_t* d, int32_t timeout_ms) {
  return guarded([&] {
    need(d && timeout_ms > 0, "direct_set_timeout: bad arguments");
    d->timeout_ticks = static_cast<uint64_t>(timeout_ms) * 100000ull;     // launches from now on
  });
}

int32_t emb_direct_status(emb_direct_t* d, int32_t* timed_out) {
  return guarded([&] {
    need(d && timed_out, "direct_status: bad arguments");
    uint32_t word = 0;
    HIP_OK(hipMemcpy(&word, &d->header(d->rank)->error, sizeof(word), hipMemcpyDeviceToHost));    // (synchronises)
    *timed_out = static_cast<int32_t>(word | *static_cast<volatile uint32_t*>(d->host_error));
  });
}

int32_t emb_direct_destroy(emb_direct_t* d) {
  return guarded([&] {
    if (!d) return;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < d->world; ++p)
      if (p != d->rank && d->peer[p]) (void)hipIpcCloseMemHandle(d->peer[p]);
    if (d->side) {
      (void)hipEventDestroy(d->forked);
      (void)hipEventDestroy(d->done);
      (void)hipStreamDestroy(d->side);
    }
    if (d->region) (void)hipFree(d->region);
    if (d->host_error) (void)hipHostFree(d->host_error);
    delete d;
  });
}

}  // extern "C"
