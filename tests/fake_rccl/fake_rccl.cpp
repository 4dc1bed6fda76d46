// TEST INFRASTRUCTURE -- a loopback stand-in for librccl, never part of the product.
//
// RCCL refuses two ranks on one GPU ("duplicate GPU"), so on a 1-GPU box the
// library's own collective entry points (emb_comm_*, embodied_amd/csrc/abi.cpp)
// could only ever run at world 1, where every collective is the identity.  This
// file exports the ten RCCL symbols abi.cpp binds with dlsym and implements them
// between PROCESSES THAT SHARE ONE GPU: EMB_RCCL_LIB=<this .so> selects it
// (abi.cpp rccl()); tests/test_gpu_native_comm_ranks.py then runs
// emb_comm_exchange / allreduce / alltoall / allgather with 2 and 3 real ranks
// and checks them against numpy on the concatenation of the ranks' inputs.
//
// Transport: one POSIX shared-memory segment per communicator = a header
// (barrier, directory) + one staging slot per rank.  Every collective is
//   stream-synchronise -> D2H of my contribution into my slot -> barrier ->
//   combine the peers' slots on the host -> H2D into the receive buffer ->
//   stream-synchronise -> barrier.
// That is a legal (if slow) implementation of a stream-ordered collective: the
// work queued on the stream before the call has finished before the buffers are
// read, and the result is in place before anything queued after the call runs.
// Host-staged on purpose: no hipIpc handles, no kernel that spins on a flag
// another process must set -- a test transport must not be able to hang the box.
// Every wait has a time-out (FAKE_RCCL_TIMEOUT_S, default 60) that turns into
// ncclSystemError.
//
// Reductions are evaluated in rank order 0..n-1 in float (double for f64) and
// rounded once, so every rank computes bit-identical results.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;

struct Piece { uint64_t offset, bytes; };

struct Header {
  std::atomic<uint32_t> attached, detached;
  std::atomic<uint32_t> count, generation;   // sense-reversing barrier
  std::atomic<uint32_t> failed;              // a rank gave up: everybody leaves
  uint32_t world;
  uint64_t slot_bytes;
  uint64_t rounds[kMaxRanks];                // send/recv groups: rounds each rank needs
  Piece pieces[kMaxRanks][kMaxRanks];        // [sender][receiver] inside the sender's slot
  uint64_t total[kMaxRanks][kMaxRanks];      // [sender][receiver] whole message size
};

struct P2p {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

thread_local int group_depth = 0;
thread_local std::vector<P2p> group_ops;
thread_local struct ncclComm* group_comm = nullptr;
std::atomic<uint64_t> ops_done{0};

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + t.tv_nsec * 1e-9;
}

double timeout_s() {
  const char* v = std::getenv("FAKE_RCCL_TIMEOUT_S");
  return v ? std::atof(v) : 60.0;
}

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: case ncclFloat8e4m3: case ncclFloat8e5m2: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

float bf16_to_float(uint16_t v) {
  uint32_t w = static_cast<uint32_t>(v) << 16;
  float f;
  std::memcpy(&f, &w, 4);
  return f;
}

uint16_t float_to_bf16(float f) {
  uint32_t w;
  std::memcpy(&w, &f, 4);
  if ((w & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((w >> 16) | 0x40);   // NaN
  w += 0x7fffu + ((w >> 16) & 1);                                                        // nearest even
  return static_cast<uint16_t>(w >> 16);
}

float half_to_float(uint16_t v) {
  _Float16 h;
  std::memcpy(&h, &v, 2);
  return static_cast<float>(h);
}

uint16_t float_to_half(float f) {
  _Float16 h = static_cast<_Float16>(f);
  uint16_t v;
  std::memcpy(&v, &h, 2);
  return v;
}

}  // namespace

struct ncclComm {
  int rank = 0, world = 1;
  Header* head = nullptr;
  uint8_t* slots = nullptr;
  size_t mapped = 0;
  std::vector<uint8_t> scratch;
  uint8_t* slot(int r) const { return slots + static_cast<size_t>(r) * head->slot_bytes; }
};

namespace {

bool barrier(ncclComm* c) {
  Header* h = c->head;
  if (h->failed.load()) return false;
  const uint32_t gen = h->generation.load();
  if (h->count.fetch_add(1) + 1 == h->world) {
    h->count.store(0);
    h->generation.fetch_add(1);
    return true;
  }
  const double limit = now_s() + timeout_s();
  for (unsigned spin = 0; h->generation.load() == gen; ++spin) {
    if (h->failed.load()) return false;
    if (spin > 200) { sched_yield(); }
    if ((spin & 0xfff) == 0 && now_s() > limit) {
      h->failed.store(1);
      std::fprintf(stderr, "fake_rccl: rank %d waited %.0f s at a barrier\n", c->rank, timeout_s());
      return false;
    }
  }
  return true;
}

bool d2h(void* dst, const void* src, size_t n, hipStream_t s) {
  if (n == 0) return hipStreamSynchronize(s) == hipSuccess;
  return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipStreamSynchronize(s) == hipSuccess;
}

bool h2d(void* dst, const void* src, size_t n, hipStream_t s) {
  if (n == 0) return true;
  return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipStreamSynchronize(s) == hipSuccess;
}

template <class Acc>
Acc combine(Acc a, Acc b, ncclRedOp_t op) {
  switch (op) {
    case ncclProd: return a * b;
    case ncclMax: return std::max(a, b);
    case ncclMin: return std::min(a, b);
    default: return a + b;       // ncclSum, ncclAvg
  }
}

// out[i] = op over ranks of slot(r)[i], evaluated in rank order.
template <class T, class Acc, class Load, class Store>
void reduce_typed(ncclComm* c, uint8_t* out, size_t n, ncclRedOp_t op, Load load, Store store) {
  for (size_t i = 0; i < n; ++i) {
    Acc acc = load(reinterpret_cast<const T*>(c->slot(0))[i]);
    for (int r = 1; r < c->world; ++r)
      acc = combine<Acc>(acc, load(reinterpret_cast<const T*>(c->slot(r))[i]), op);
    if (op == ncclAvg) acc = acc / static_cast<Acc>(c->world);
    reinterpret_cast<T*>(out)[i] = store(acc);
  }
}

bool reduce(ncclComm* c, uint8_t* out, size_t n, ncclDataType_t t, ncclRedOp_t op) {
  switch (t) {
    case ncclFloat32:
      reduce_typed<float, float>(c, out, n, op, [](float v) { return v; }, [](float v) { return v; });
      return true;
    case ncclFloat64:
      reduce_typed<double, double>(c, out, n, op, [](double v) { return v; }, [](double v) { return v; });
      return true;
    case ncclBfloat16:
      reduce_typed<uint16_t, float>(c, out, n, op, bf16_to_float, float_to_bf16);
      return true;
    case ncclFloat16:
      reduce_typed<uint16_t, float>(c, out, n, op, half_to_float, float_to_half);
      return true;
    case ncclInt32:
      reduce_typed<int32_t, int64_t>(c, out, n, op, [](int32_t v) { return int64_t(v); },
                                     [](int64_t v) { return int32_t(v); });
      return true;
    case ncclInt64:
      reduce_typed<int64_t, int64_t>(c, out, n, op, [](int64_t v) { return v; }, [](int64_t v) { return v; });
      return true;
    case ncclUint8:
      reduce_typed<uint8_t, int64_t>(c, out, n, op, [](uint8_t v) { return int64_t(v); },
                                     [](int64_t v) { return uint8_t(v); });
      return true;
    default:
      return false;
  }
}

ncclResult_t run_group(ncclComm* c, std::vector<P2p>& ops) {
  Header* h = c->head;
  const size_t share = h->slot_bytes / h->world / 16 * 16;     // my slot, cut into one part per receiver
  if (share == 0) return ncclInvalidUsage;
  // One send and one receive per peer at most (what an all-to-all issues).
  const P2p* sends[kMaxRanks] = {};
  const P2p* recvs[kMaxRanks] = {};
  for (const P2p& op : ops) {
    if (op.peer < 0 || op.peer >= c->world) return ncclInvalidArgument;
    const P2p** table = op.send ? sends : recvs;
    if (table[op.peer]) return ncclInvalidUsage;
    table[op.peer] = &op;
  }
  uint64_t rounds = 0;
  for (int p = 0; p < c->world; ++p) {
    h->total[c->rank][p] = sends[p] ? sends[p]->bytes : 0;
    if (sends[p]) rounds = std::max<uint64_t>(rounds, (sends[p]->bytes + share - 1) / share);
  }
  h->rounds[c->rank] = rounds;
  if (!barrier(c)) return ncclSystemError;
  for (int r = 0; r < c->world; ++r) rounds = std::max(rounds, h->rounds[r]);
  // What a peer sends me must be what I expect from it.
  for (int p = 0; p < c->world; ++p) {
    const uint64_t want = recvs[p] ? recvs[p]->bytes : 0;
    if (h->total[p][c->rank] != want) {
      std::fprintf(stderr, "fake_rccl: rank %d expects %llu bytes from rank %d, which sends %llu\n",
                   c->rank, (unsigned long long)want, p, (unsigned long long)h->total[p][c->rank]);
      h->failed.store(1);
      return ncclInvalidUsage;
    }
  }
  for (uint64_t round = 0; round < rounds; ++round) {
    for (int p = 0; p < c->world; ++p) {
      Piece piece{static_cast<uint64_t>(p) * share, 0};
      if (sends[p] && round * share < sends[p]->bytes) {
        piece.bytes = std::min<uint64_t>(share, sends[p]->bytes - round * share);
        if (!d2h(c->slot(c->rank) + piece.offset,
                 static_cast<const uint8_t*>(sends[p]->ptr) + round * share, piece.bytes, sends[p]->stream))
          return ncclUnhandledCudaError;
      } else if (sends[p] && round == 0) {
        if (hipStreamSynchronize(sends[p]->stream) != hipSuccess) return ncclUnhandledCudaError;
      }
      h->pieces[c->rank][p] = piece;
    }
    if (!barrier(c)) return ncclSystemError;
    for (int p = 0; p < c->world; ++p) {
      const Piece piece = h->pieces[p][c->rank];
      if (!recvs[p] || piece.bytes == 0) continue;
      if (hipStreamSynchronize(recvs[p]->stream) != hipSuccess) return ncclUnhandledCudaError;
      if (!h2d(static_cast<uint8_t*>(recvs[p]->ptr) + round * share, c->slot(p) + piece.offset,
               piece.bytes, recvs[p]->stream))
        return ncclUnhandledCudaError;
    }
    if (!barrier(c)) return ncclSystemError;
  }
  // (a group of empty messages still ends on a barrier: the directory is
  // rewritten by the next group)
  if (rounds == 0 && !barrier(c)) return ncclSystemError;
  ops_done.fetch_add(1);
  return ncclSuccess;
}

}  // namespace

extern "C" {

// How many collectives this process has run through the loopback transport:
// lets a test prove which library its emb_comm_* calls went through.
uint64_t fake_rccl_ops_done() { return ops_done.load(); }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  static std::atomic<uint32_t> serial{0};
  std::snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%lld_%u", static_cast<int>(getpid()),
                static_cast<long long>(t.tv_nsec), serial.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  if (std::strncmp(id.internal, "/fake_rccl_", 11) != 0) return ncclInvalidArgument;
  const char* env = std::getenv("FAKE_RCCL_SLOT_BYTES");
  const size_t slot = env ? std::strtoull(env, nullptr, 10) / 4096 * 4096 : (size_t(32) << 20);
  if (slot < 4096) return ncclInvalidArgument;
  const size_t head = (sizeof(Header) + 4095) / 4096 * 4096;
  const size_t total = head + slot * nranks;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, static_cast<off_t>(total)) != 0) { close(fd); return ncclSystemError; }
  void* base = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) return ncclSystemError;
  auto* comm = new ncclComm;
  comm->rank = rank;
  comm->world = nranks;
  comm->head = static_cast<Header*>(base);      // a fresh segment is zero-filled
  comm->slots = static_cast<uint8_t*>(base) + head;
  comm->mapped = total;
  if (rank == 0) {
    comm->head->world = nranks;
    comm->head->slot_bytes = slot;
  }
  comm->head->attached.fetch_add(1);
  const double limit = now_s() + timeout_s();
  while (comm->head->attached.load() < static_cast<uint32_t>(nranks)) {
    sched_yield();
    if (now_s() > limit) {
      shm_unlink(id.internal);
      munmap(base, total);
      delete comm;
      return ncclSystemError;
    }
  }
  // Everybody holds a mapping: the name can go (nothing is left in /dev/shm if
  // a rank dies later).
  if (rank == 0) shm_unlink(id.internal);
  if (!barrier(comm)) { munmap(base, total); delete comm; return ncclSystemError; }
  *out = comm;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  comm->head->detached.fetch_add(1);
  munmap(comm->head, comm->mapped);
  delete comm;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclSystemError: return "system error or time-out (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "error (fake_rccl)";
  }
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type,
                           ncclComm_t c, hipStream_t stream) {
  const size_t size = type_size(type);
  if (!c || !size || (count && (!send || !recv))) return ncclInvalidArgument;
  const size_t bytes = count * size, chunk = c->head->slot_bytes;
  for (size_t off = 0; off < bytes; off += chunk) {
    const size_t n = std::min(chunk, bytes - off);
    if (!d2h(c->slot(c->rank), static_cast<const uint8_t*>(send) + off, n, stream))
      return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r)
      if (!h2d(static_cast<uint8_t*>(recv) + r * bytes + off, c->slot(r), n, stream))
        return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
  }
  ops_done.fetch_add(1);
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type,
                           ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
  const size_t size = type_size(type);
  if (!c || !size || (count && (!send || !recv))) return ncclInvalidArgument;
  if (op != ncclSum && op != ncclProd && op != ncclMax && op != ncclMin && op != ncclAvg)
    return ncclInvalidArgument;
  const size_t per = c->head->slot_bytes / size;
  for (size_t done = 0; done < count; done += per) {
    const size_t n = std::min(per, count - done);
    if (!d2h(c->slot(c->rank), static_cast<const uint8_t*>(send) + done * size, n * size, stream))
      return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    c->scratch.resize(n * size);
    if (!reduce(c, c->scratch.data(), n, type, op)) return ncclInvalidArgument;
    if (!h2d(static_cast<uint8_t*>(recv) + done * size, c->scratch.data(), n * size, stream))
      return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
  }
  ops_done.fetch_add(1);
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (group_depth <= 0) return ncclInvalidUsage;
  if (--group_depth > 0) return ncclSuccess;
  ncclComm* c = group_comm;
  std::vector<P2p> ops;
  ops.swap(group_ops);
  group_comm = nullptr;
  if (!c) return ncclSuccess;
  return run_group(c, ops);
}

static ncclResult_t queue_p2p(bool send, void* ptr, size_t count, ncclDataType_t type, int peer,
                              ncclComm_t c, hipStream_t stream) {
  const size_t size = type_size(type);
  if (!c || !size || (count && !ptr)) return ncclInvalidArgument;
  // Point-to-point transfers pair up across processes; this transport runs them
  // as one collective step, so every rank has to issue its group.
  if (group_depth <= 0) return ncclInvalidUsage;
  if (group_comm && group_comm != c) return ncclInvalidUsage;
  group_comm = c;
  group_ops.push_back({send, ptr, count * size, peer, stream});
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c,
                      hipStream_t stream) {
  return queue_p2p(true, const_cast<void*>(buf), count, type, peer, c, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c,
                      hipStream_t stream) {
  return queue_p2p(false, buf, count, type, peer, c, stream);
}

}  // extern "C"
