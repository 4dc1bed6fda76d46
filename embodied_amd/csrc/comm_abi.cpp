// emb_comm_*: the collectives of the N > 1 path on RCCL (symbols taken with
// dlsym from the copy the process already has).
#include "handles.h"

#include <rccl/rccl.h>   // types only: the symbols are taken with dlsym

#include <dlfcn.h>

// -------------------------------------------------------------- collectives --

namespace {

struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

// One RCCL per process: the copy that is already loaded (torch bundles one with
// the same SONAME) if there is one, else the system's.
const Rccl& rccl() {
  static const Rccl table = [] {
    void* lib = nullptr;
    // EMB_RCCL_LIB=<path>: bind the ten symbols below from that library and no
    // other (a site's own RCCL build; the suite's loopback transport between
    // processes that share one GPU, tests/fake_rccl/).  No fallback: a path
    // that does not load is an error.
    if (const char* chosen = emb::knob("EMB_RCCL_LIB"); chosen && *chosen) {
      lib = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
      if (!lib) throw std::runtime_error(std::string("EMB_RCCL_LIB: cannot load ") + chosen + ": " + dlerror());
    }
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) throw std::runtime_error(std::string("cannot load RCCL: ") + dlerror());
    Rccl t;
    auto sym = [&](const char* name) {
      void* p = dlsym(lib, name);
      if (!p) throw std::runtime_error(std::string("RCCL lacks ") + name);
      return p;
    };
    t.get_unique_id = reinterpret_cast<decltype(t.get_unique_id)>(sym("ncclGetUniqueId"));
    t.comm_init_rank = reinterpret_cast<decltype(t.comm_init_rank)>(sym("ncclCommInitRank"));
    t.comm_destroy = reinterpret_cast<decltype(t.comm_destroy)>(sym("ncclCommDestroy"));
    t.all_gather = reinterpret_cast<decltype(t.all_gather)>(sym("ncclAllGather"));
    t.all_reduce = reinterpret_cast<decltype(t.all_reduce)>(sym("ncclAllReduce"));
    t.send = reinterpret_cast<decltype(t.send)>(sym("ncclSend"));
    t.recv = reinterpret_cast<decltype(t.recv)>(sym("ncclRecv"));
    t.group_start = reinterpret_cast<decltype(t.group_start)>(sym("ncclGroupStart"));
    t.group_end = reinterpret_cast<decltype(t.group_end)>(sym("ncclGroupEnd"));
    t.error_string = reinterpret_cast<decltype(t.error_string)>(sym("ncclGetErrorString"));
    return t;
  }();
  return table;
}

void rccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess)
    throw std::runtime_error(std::string(what) + ": " + rccl().error_string(r));
}

}  // namespace

struct emb_comm {
  ncclComm_t comm = nullptr;
  int32_t rank = 0, world = 1;
  // emb_comm_exchange: the communicator's own stream, so that a train step's
  // collectives overlap whatever the caller's stream does next.
  hipStream_t side = nullptr;
  hipEvent_t forked = nullptr, done = nullptr;
  bool in_flight = false;
};

static void alltoall_on(emb_comm* comm, const void* send, void* recv, int64_t bytes_per_rank,
                        hipStream_t s) {
  const auto* from = static_cast<const uint8_t*>(send);
  auto* to = static_cast<uint8_t*>(recv);
  const size_t n = static_cast<size_t>(bytes_per_rank);
  // One fused group of point-to-point transfers: on xGMI every pair of GPUs
  // has its own link, so the n-1 blocks leave on n-1 links at once.
  rccl_ok(rccl().group_start(), "ncclGroupStart");
  ncclResult_t first = ncclSuccess;
  for (int32_t peer = 0; peer < comm->world && first == ncclSuccess; ++peer) {
    first = rccl().send(from + peer * n, n, ncclUint8, peer, comm->comm, s);
    if (first == ncclSuccess) first = rccl().recv(to + peer * n, n, ncclUint8, peer, comm->comm, s);
  }
  const ncclResult_t closed = rccl().group_end();
  rccl_ok(first, "ncclSend/ncclRecv");
  rccl_ok(closed, "ncclGroupEnd");
}

static ncclDataType_t grad_type(int32_t dtype) {
  switch (dtype) {
    case EMB_F32: return ncclFloat32;
    case EMB_BF16: return ncclBfloat16;
    case EMB_F16: return ncclFloat16;
    case EMB_F64: return ncclFloat64;
    default: need(false, "comm_allreduce_grads: dtype must be f16, bf16, f32 or f64");
  }
  return ncclFloat32;
}

extern "C" {

int32_t emb_comm_unique_id(uint8_t* id_out) {
  return guarded([&] {
    need(id_out, "comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == EMB_COMM_ID_BYTES, "RCCL id size");
    ncclUniqueId id;
    rccl_ok(rccl().get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof(id));
  });
}

int32_t emb_comm_init(const uint8_t* id, int32_t rank, int32_t world, emb_comm_t** out) {
  return guarded([&] {
    need(id && out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    auto comm = std::make_unique<emb_comm>();
    comm->rank = rank;
    comm->world = world;
    rccl_ok(rccl().comm_init_rank(&comm->comm, world, uid, rank), "ncclCommInitRank");
    HIP_OK(hipStreamCreateWithFlags(&comm->side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&comm->forked, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&comm->done, hipEventDisableTiming));
    *out = comm.release();
  });
}

int32_t emb_comm_allgather_traj(emb_comm_t* comm, const void* send, void* recv,
                                int64_t bytes_per_rank, void* stream) {
  return guarded([&] {
    need(comm && send && recv && bytes_per_rank >= 0, "comm_allgather_traj: bad arguments");
    if (bytes_per_rank == 0) return;
    rccl_ok(rccl().all_gather(send, recv, static_cast<size_t>(bytes_per_rank), ncclUint8, comm->comm,
                              static_cast<hipStream_t>(stream)),
            "ncclAllGather");
  });
}

int32_t emb_comm_allgather_returns(emb_comm_t* comm, const void* send, void* recv, int64_t count,
                                   void* stream) {
  return guarded([&] {
    need(comm && send && recv && count >= 0, "comm_allgather_returns: bad arguments");
    if (count == 0) return;
    rccl_ok(rccl().all_gather(send, recv, static_cast<size_t>(count), ncclFloat32, comm->comm,
                              static_cast<hipStream_t>(stream)),
            "ncclAllGather");
  });
}

int32_t emb_comm_pmean_scalars(emb_comm_t* comm, void* values, int64_t count, void* stream) {
  return guarded([&] {
    need(comm && values && count >= 0, "comm_pmean_scalars: bad arguments");
    if (count == 0) return;
    rccl_ok(rccl().all_reduce(values, values, static_cast<size_t>(count), ncclFloat32, ncclAvg,
                              comm->comm, static_cast<hipStream_t>(stream)),
            "ncclAllReduce");
  });
}

int32_t emb_comm_alltoall_slices(emb_comm_t* comm, const void* send, void* recv,
                                 int64_t bytes_per_rank, void* stream) {
  return guarded([&] {
    need(comm && send && recv && bytes_per_rank >= 0, "comm_alltoall_slices: bad arguments");
    if (bytes_per_rank == 0) return;
    alltoall_on(comm, send, recv, bytes_per_rank, static_cast<hipStream_t>(stream));
  });
}

static int32_t allreduce_typed(emb_comm_t* comm, void* buf, int64_t count, int32_t dtype,
                               int32_t mean, void* stream) {
  return guarded([&] {
    need(comm && buf && count >= 0, "comm_allreduce_grads: bad arguments");
    const ncclDataType_t type = grad_type(dtype);
    if (count == 0) return;
    rccl_ok(rccl().all_reduce(buf, buf, static_cast<size_t>(count), type,
                              mean ? ncclAvg : ncclSum, comm->comm, static_cast<hipStream_t>(stream)),
            "ncclAllReduce");
  });
}

int32_t emb_comm_allreduce_grads(emb_comm_t* comm, void* buf, int64_t count, int32_t mean,
                                 void* stream) {
  return allreduce_typed(comm, buf, count, EMB_F32, mean, stream);
}

int32_t emb_comm_allreduce_grads_as(emb_comm_t* comm, void* buf, int64_t count, int32_t dtype,
                                    int32_t mean, void* stream) {
  return allreduce_typed(comm, buf, count, dtype, mean, stream);
}

static int32_t comm_exchange(emb_comm_t* comm, void* after_stream, const void* send, void* recv,
                             int64_t bytes_per_rank, void* grads, int64_t count, int32_t dtype, int32_t mean,
                             bool gather) {
  return guarded([&] {
    need(comm && bytes_per_rank >= 0 && count >= 0, "comm_exchange: bad arguments");
    need(bytes_per_rank == 0 || (send && recv), "comm_exchange: null slice / trajectory buffers");
    need(count == 0 || grads, "comm_exchange: null gradient buffer");
    const ncclDataType_t type = grad_type(count ? dtype : EMB_F32);
    if (bytes_per_rank == 0 && count == 0) return;
    HIP_OK(hipEventRecord(comm->forked, static_cast<hipStream_t>(after_stream)));
    HIP_OK(hipStreamWaitEvent(comm->side, comm->forked, 0));
    if (bytes_per_rank && gather)
      rccl_ok(rccl().all_gather(send, recv, static_cast<size_t>(bytes_per_rank), ncclUint8, comm->comm, comm->side),
              "ncclAllGather");
    else if (bytes_per_rank)
      alltoall_on(comm, send, recv, bytes_per_rank, comm->side);
    if (count)
      rccl_ok(rccl().all_reduce(grads, grads, static_cast<size_t>(count), type,
                                mean ? ncclAvg : ncclSum, comm->comm, comm->side),
              "ncclAllReduce");
    HIP_OK(hipEventRecord(comm->done, comm->side));
    comm->in_flight = true;
  });
}

int32_t emb_comm_exchange(emb_comm_t* comm, void* after_stream, const void* slices_send,
                          void* slices_recv, int64_t bytes_per_rank, void* grads, int64_t count,
                          int32_t dtype, int32_t mean) {
  return comm_exchange(comm, after_stream, slices_send, slices_recv, bytes_per_rank, grads, count, dtype, mean, false);
}

int32_t emb_comm_exchange_gather(emb_comm_t* comm, void* after_stream, const void* traj_send,
                                 void* traj_recv, int64_t bytes_per_rank, void* grads, int64_t count,
                                 int32_t dtype, int32_t mean) {
  return comm_exchange(comm, after_stream, traj_send, traj_recv, bytes_per_rank, grads, count, dtype, mean, true);
}

int32_t emb_comm_wait(emb_comm_t* comm, void* stream) {
  return guarded([&] {
    need(comm, "comm_wait: null communicator");
    if (!comm->in_flight) return;
    HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), comm->done, 0));
    comm->in_flight = false;
  });
}

int32_t emb_comm_destroy(emb_comm_t* comm) {
  return guarded([&] {
    if (!comm) return;
    if (comm->side) {
      (void)hipStreamSynchronize(comm->side);
      (void)hipEventDestroy(comm->forked);
      (void)hipEventDestroy(comm->done);
      (void)hipStreamDestroy(comm->side);
    }
    if (comm->comm) rccl_ok(rccl().comm_destroy(comm->comm), "ncclCommDestroy");
    delete comm;
  });
}

}  // extern "C"
