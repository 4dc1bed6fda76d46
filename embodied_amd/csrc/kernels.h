// Launchers for the gfx950 kernels in kernels.hip (host-callable, no torch).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace emb {

constexpr int kMaxKeys = 16;
constexpr int kInlineWords = 704;  // row table / spans / step ids carried in kernel arguments

// One replay column: its chunk pool in HBM and the batch-side buffer.
struct KeyDesc {
  uint8_t* pool;       // (n_slots * chunksize, rowbytes)
  uint8_t* batch;      // (n_rows, rowbytes) contiguous
  int64_t rowbytes;
};

struct MovePlan {
  int32_t n_keys = 0;
  KeyDesc key[kMaxKeys];
  int32_t n_rows = 0;        // rows moved per key
  int32_t seq_len = 1;       // rows per sequence (annotate uses t = r % seq_len)
  // Gather only: key k's batch side is (n_rows / seq_len, key_len[k], rowbytes)
  // and receives the first key_len[k] steps of every sequence -- a key whose
  // consumer reads only the head of the sampled window (DreamerV3's replay
  // context, dreamerv3/agent.py:322-331).  0 = the whole sequence.
  int32_t key_len[kMaxKeys] = {};
  int32_t key_is_first = -1; // gather only: fuse replay.py:277-292
  int32_t key_is_last = -1;
  const uint8_t* is_first_pool = nullptr;  // pool of the is_first key (may be in another launch)
  // Pool rows, one of:
  //   rows        device-visible int32[n_rows] (-1 = skip)               [any size]
  //   rows_host   the same table on the host: if it fits it travels in the
  //               kernel arguments (no upload, no dependent global load)
  //   spans_host  per sequence {row0, count0, row1}: windows that cross at most
  //               one chunk boundary, n_rows = n_seq * seq_len            [gather]
  const int32_t* rows = nullptr;
  const int32_t* rows_host = nullptr;
  const int32_t* spans_host = nullptr;
  int32_t n_seq = 0;
  // The process keeps kernel arguments in host memory (HIP_FORCE_DEV_KERNARG=0):
  // prefer the mover with the fewest argument readers.
  bool args_in_host_memory = false;
  // Gather only: batch side cut into groups of `group` sequences whose starts
  // are `group_stride` bytes apart (every key.batch is then the key's offset
  // inside group 0); 0 = dense.
  int32_t group = 0;
  int64_t group_stride = 0;
  // Scatter only: the batch bytes of this key (4-byte multiple rows, e.g. the
  // 20-byte step ids) come from host memory through the kernel arguments.
  int32_t inline_key = -1;
  const uint8_t* inline_bytes = nullptr;
  // Scatter only (Driver mask fused into the insert, driver.py:72-74): keys in
  // mask_bits are written as value * !mask_flags[r] in mask_dtype[k] (a DType
  // code), to the pool row and, if mask_out[k] is set, to that (n_rows,
  // rowbytes) buffer as well (the masked actions the next env step receives).
  uint32_t mask_bits = 0;
  int8_t mask_dtype[kMaxKeys] = {};
  uint8_t* mask_out[kMaxKeys] = {};
  const uint8_t* mask_flags = nullptr;
  // Scatter only: the rows are rows of the workers' OPEN chunks (an insert), not
  // rows of existing items (a write-back): what cross-stream ordering it needs
  // (abi.cpp StreamOrder).
  bool fresh_rows = false;
  // The launch's stream may use this many compute units only (a stream made by
  // emb_stream_create_on_cus; 0 = all): the persistent mover sizes its grid by it.
  int32_t cu_limit = 0;
};

// True if this plan's tables fit the kernel-argument block (else the caller
// must provide `rows` in device-visible memory).
bool plan_fits_inline(const MovePlan& plan);

// A prepared launch: the kernel's argument block (opaque) and its grid.
constexpr size_t kMoveArgsBytes = 4096;
struct MoveLaunch {
  alignas(16) unsigned char args[kMoveArgsBytes];
  uint32_t blocks = 0;
  uint32_t threads = 0;
  bool span = false;       // persistent span mover (wide keys of a span table)
  bool stage_tables = false;   // arguments in host memory: by-value flat movers stage their tables in LDS
  int nt = 3;              // span mover: non-temporal hints of this launch (bit 0 loads, bit 1 stores)
};
hipError_t prepare_move(const MovePlan& plan, MoveLaunch* out, bool gather = true);
size_t move_args_bytes();
// An empty one-wave kernel whose completion is stamped with `stop`: put in front
// of a timed launch it takes the previous (unstamped) kernel's end-of-kernel
// release into ITS window instead of the timed one's.
hipError_t launch_marker(hipStream_t stream, hipEvent_t stop);
// Copies launch.args into device memory with a one-workgroup kernel (its own
// arguments may live in host memory); `stop` (optional) stamps its completion.
hipError_t launch_args_writer(const MoveLaunch& launch, void* device_dst, hipStream_t stream,
                              hipEvent_t stop = nullptr);
// device_args == nullptr: arguments by value.  Otherwise a device-visible copy
// of launch.args[0 .. move_args_bytes()) that the kernel reads through a pointer.
hipError_t launch_move(const MoveLaunch& launch, bool gather, const void* device_args,
                       hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

// Name of the kernel launch_move runs for `launch` (as rocprofv3 prints it,
// without namespaces and parameters); valid until the thread's next call.
const char* move_kernel_name(const MoveLaunch& launch, bool gather, bool indirect);

// pool[rows[r]] -> batch[r]   (Replay.sample: replay.py:255-292 on device)
// start/stop (optional): events stamped with the dispatch's own begin/end
// (hipExtLaunchKernelGGL), i.e. the kernel duration rocprofv3 reports.
hipError_t launch_gather(const MovePlan& plan, hipStream_t stream,
                         hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// batch[r] -> pool[rows[r]]   (Replay.add / Replay.update: chunk.py:41-58)
hipError_t launch_scatter(const MovePlan& plan, hipStream_t stream);

// (B, total, rowbytes) -> (B, count, rowbytes) window starting at `start`
// (streams.py:133-138).
// dst[0 .. bytes) = src[0 .. bytes) by a kernel (src may be pinned host memory).
hipError_t launch_copy_bytes(const void* src, void* dst, int64_t bytes, hipStream_t stream);
hipError_t launch_window(const uint8_t* src, uint8_t* dst, int64_t batch,
                         int64_t total, int64_t start, int64_t count,
                         int64_t rowbytes, hipStream_t stream);

enum ObsLayout { kLayoutSame = 0, kLayoutChannelsFirst = 1 };
enum DType {
  kU8 = 0, kI8 = 1, kI16 = 2, kI32 = 3, kI64 = 4,
  kF16 = 5, kBF16 = 6, kF32 = 7, kF64 = 8, kBool = 9,
};

// Per-env uint8 frames (N, P, C) [env e at src + env_ids[e] * P*C, or e if
// env_ids is null] -> policy batch (N, P, C) or (N, C, P) in u8/f16/bf16/f32,
// value * scale + offset for float outputs (driver.py:65 + jax/agent.py:230).
hipError_t launch_obs_stack(const uint8_t* src, const int32_t* env_ids, void* dst,
                            int64_t n, int64_t pixels, int64_t channels,
                            int layout, int out_dtype, float scale, float offset,
                            hipStream_t stream);

// out[n, :] = act[n, :] * !is_last[n] in the action's own dtype (out may be act) (driver.py:72-74,84-87).
// flag (optional, pinned host memory) receives `seq` when the whole launch has stored;
// counter: a zeroed device word of the caller's.
hipError_t launch_mask_rows(const void* act, void* out, int64_t n, int64_t row_elems,
                            int dtype, const uint8_t* is_last, hipStream_t stream,
                            uint32_t* counter = nullptr, uint32_t* flag = nullptr, uint32_t seq = 0);

// Return scans (float32).  Batch-major (B, T) unless stated.
// group > 0: rew / last / term are keys of a grouped packed batch (row b starts
// (b / group) * group_stride_bytes + (b % group) * T * itemsize into its key).
hipError_t launch_gae(const float* rew, const float* val, const uint8_t* last,
                      const uint8_t* term, int64_t B, int64_t T, float live_scale,
                      float lam, float* adv, float* tar, hipStream_t stream,
                      int64_t group = 0, int64_t group_stride_bytes = 0);
hipError_t launch_lambda_return(const uint8_t* last, const uint8_t* term,
                                const float* rew, const float* boot, int64_t B,
                                int64_t T, float disc, float lam, float* ret,
                                hipStream_t stream);

// One lambda-return problem of emb_scan_lambda_multi (same meaning as the
// arguments of launch_lambda_return).
struct LambdaProblem {
  const uint8_t* last; const uint8_t* term; const float* rew; const float* boot; float* ret;
  int64_t B, T; float disc, lam;
};
// All problems in one launch when every row has at most 257 steps (else one
// launch each).
hipError_t launch_lambda_return_multi(int n_problems, const LambdaProblem* problems, hipStream_t stream);
// Time-major: rew (T-1, B), cont/value (T, B) -> ret (T-1, B).
hipError_t launch_director_score(const float* rew, const float* cont,
                                 const float* value, int64_t T, int64_t B,
                                 float discount, float lam, float* ret,
                                 hipStream_t stream);

// Director manager-step abstraction (director/hierarchy.py:240-256), time-major:
// reward (T-1, B), cont (T, B) -> reward_out (T/k - 1, B), cont_out (T/k, B).
hipError_t launch_abstract_traj(const float* reward, const float* cont, int64_t T, int64_t B,
                                int k, float* reward_out, float* cont_out, hipStream_t stream);

// Synthetic vector env (bench/test input): counter-hash frames written straight
// into HBM, SURVEY.md 8d.
// `counters` = int32[2][2n]: generation `turn` is read, the other one written.
hipError_t launch_synth_env(uint8_t* image, float* reward, uint8_t* is_first,
                            uint8_t* is_last, uint8_t* is_terminal, int64_t n,
                            int64_t frame_bytes, int64_t env0, int64_t episode_len,
                            const uint8_t* reset, int32_t* counters, int turn,
                            hipStream_t stream);

// Obs stack fused with the early part of Replay.add (driver.py:65 +
// jax/agent.py:230 + chunk.py:41-50): every frame is read once and written to
// the policy batch AND to the pool row its step will occupy; up to kPreNarrow
// narrow observation keys, the step ids and a device copy of the row table ride
// in the same launch.  Rows, step ids and the narrow keys' descriptors reach
// the kernel through `table_dev`: prewrite_table_bytes(n) bytes of DEVICE
// memory filled with prewrite_fill_table (directly through the BAR, or a
// staging buffer + copy).
constexpr int kPreNarrow = 8;
constexpr int kStepBytes = 20;
struct PrewritePlan {
  const uint8_t* frames = nullptr;   // (n, pixels, channels) u8
  void* dst = nullptr;               // policy batch
  uint8_t* frame_pool = nullptr;     // pool of the frame key (rows of pixels * channels bytes)
  int64_t pixels = 0, channels = 0;
  int layout = kLayoutSame, out_dtype = kU8;
  float scale = 1.f, offset = 0.f;
  int32_t n = 0;
  int32_t n_narrow = 0;
  struct { const uint8_t* src; uint8_t* pool; int64_t rowbytes; } narrow[kPreNarrow] = {};
  uint8_t* stepid_pool = nullptr;
  const void* table_dev = nullptr;
  int32_t* rows_out = nullptr;       // device int32[n] for launch_publish_one (optional)
  // Carried publish: the PREVIOUS step's masked key (value * !flag in
  // carry_dtype) written to rows carry_rows[e] of carry_pool by this launch.
  // carry_flags = the replay's is_last pool (1-byte rows): env e's flag is read at
  // carry_rows[e], where that step's own early insert stored it.
  const uint8_t* carry_src = nullptr;
  uint8_t* carry_pool = nullptr;
  const uint8_t* carry_flags = nullptr;
  int64_t carry_rowbytes = 0;
  int carry_dtype = kU8;
  const int32_t* carry_rows = nullptr;   // host int32[n]
};
// A key the early-insert launch can carry: one element per lane of a workgroup.
bool carry_supported(int64_t rowbytes, int dtype);
bool prewrite_supported(const PrewritePlan& plan);
size_t prewrite_table_bytes(int64_t n);
void prewrite_fill_table(void* dst, const PrewritePlan& plan, const int32_t* rows, const uint8_t* stepids);
hipError_t launch_obs_stack_insert(const PrewritePlan& plan, hipStream_t stream, hipEvent_t stop = nullptr);
// One key of n rows to the pool rows `rows_dev` (device int32[n], -1 = skip):
// value * !flags[r] in `dtype` when flags is set (and to `out` as well), a plain
// copy otherwise.  flags_by_row: `flags` is a pool of 1-byte rows and row r's
// flag is flags[rows_dev[r]].
hipError_t launch_publish_one(const void* src, void* pool, void* out, const int32_t* rows_dev,
                              const uint8_t* flags, int64_t n, int64_t rowbytes, int dtype,
                              hipStream_t stream, hipEvent_t stop = nullptr, bool flags_by_row = false);

}  // namespace emb
