// gfx950 (CDNA4, wave64) kernels for the Driver / Replay / return-scan hot path.
//
// Everything here is HBM-bound byte movement or a short recurrence: no MFMA.
// What matters (cdna_hip_programming.md G2, G11, G13): 16-byte accesses per
// lane with consecutive lanes on consecutive addresses, several independent
// loads in flight per lane, and far more than 256 workgroups per launch.
#include "kernels.h"
#include "knobs.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace emb {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one dwordx4

// Payload pointers that reach a kernel through memory (the indirect-argument
// movers read their KeyDescs with loads) have no known address space, and the
// compiler then emits flat_load / flat_store — slower than global_load /
// global_store and tied to the LDS counter.  Everything the movers touch is
// device global memory: say so.
typedef __attribute__((address_space(1))) u32x4 gu32x4;
template <bool kNonTemporal>
__device__ __forceinline__ u32x4 load16(const u32x4* p) {
  const gu32x4* g = (const gu32x4*)p;
  if constexpr (kNonTemporal) return __builtin_nontemporal_load(g);
  else return *g;
}
template <bool kNonTemporal>
__device__ __forceinline__ void store16(u32x4* p, u32x4 v) {
  gu32x4* g = (gu32x4*)p;
  if constexpr (kNonTemporal) __builtin_nontemporal_store(v, g);
  else *g = v;
}

constexpr int kThreads = 256;

// What a persistent span-mover workgroup needs before its first load, in one
// contiguous 160-byte block at the very start of the arguments: a wave copies
// it into registers with three independent scalar loads (one memory latency)
// instead of walking four or five dependent ones through the full plan — which
// matters most when the plan is read through a pointer (host-resident kernel
// arguments: the device copy starts out cold in every XCD's L2).
constexpr int kSpanKeys = 4;     // wide keys per span launch (more: flat mover)
struct alignas(16) SpanHead {
  int32_t wide_workers, n_wide, seq_len, group;
  int64_t group_stride;
  uint32_t ntiles, pad_;
  uint32_t tile0[kSpanKeys];          // first tile of wide key k; 0xFFFFFFFF beyond n_wide
  uint32_t tiles_per_seq[kSpanKeys];
  // 16-byte units of one sequence of wide key k on the batch side: its head
  // length (MovePlan::key_len, seq_len unless the key is a context-only one)
  // times rowbytes / 16.
  uint32_t units_per_seq[kSpanKeys];
  KeyDesc key[kSpanKeys];
};
static_assert(sizeof(SpanHead) == 176, "SpanHead is loaded as 44 dwords");

// Per-launch plan in kernel-argument memory (< 4 KiB).
// Everything of a launch plan except the head and the inline words: which key
// owns which virtual blocks, the keys themselves, how rows are resolved.  One
// contiguous block so that the by-value movers can bring it into LDS with one
// load per lane when the arguments live in host memory (see stage_tables).
struct alignas(16) MoveTables {
  KeyDesc key[kMaxKeys];
  int32_t first_block[kMaxKeys + 1];
  int32_t unit[kMaxKeys];           // 0: 16-byte flat path; else bytes per lane
  // Steps of a sequence that key k moves (its head: seq_len, or fewer for a
  // context-only key of a gather) and its batch rows in this launch
  // (n_seq * key_len[k]; n_rows when every key moves whole sequences).
  int32_t key_len[kMaxKeys];
  int32_t key_rows[kMaxKeys];
  int32_t n_keys, n_rows, seq_len, key_is_first, key_is_last;
  int32_t rows_mode;                // 0 device table, 1 inline rows, 2 inline spans
  int32_t inline_key, inline_key_word0;
  const uint8_t* is_first_pool;
  const int32_t* rows;
  uint32_t mask_bits;               // scatter: keys written as value * !mask_flags[r]
  int8_t mask_dtype[kMaxKeys];
  uint8_t* mask_out[kMaxKeys];
  const uint8_t* mask_flags;
  // Gather only: the batch side in groups of `group` sequences, `group_stride`
  // bytes apart (0 = one dense (n_rows, rowbytes) array per key).  Sequence s of
  // key k starts at key.batch + (s / group) * group_stride + (s % group) * L *
  // rowbytes: the layout of a packed batch cut into per-destination-rank blocks
  // (distributed.py, DP-slice exchange).
  int32_t group;
  int64_t group_stride;
};
static_assert(sizeof(MoveTables) % 16 == 0 && sizeof(MoveTables) / 16 <= 64,
              "the tables are staged as one 16-byte load per lane of one wave");

// Per-launch plan in kernel-argument memory (< 4 KiB).  Span mode (rows_mode 2):
// the wide keys are moved by `head.wide_workers` persistent workgroups walking
// `head.ntiles` tiles (see move_wide_spans); those keys own no virtual blocks.
struct alignas(16) MoveArgs {
  SpanHead head;
  // Row table / span table / step ids carried in the arguments.  Directly behind
  // the head: the span mover stages head + the first spans with ONE load per lane.
  uint32_t inline_words[kInlineWords];
  MoveTables t;
};
static_assert(sizeof(MoveArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// What the span mover stages through LDS before anything else: the head and the
// first spans, contiguous in the arguments.
constexpr int kStagedSeqs = 70;      // (176 + 70 * 12 + 8) / 16 = 64 lanes: one wave, one load each
struct StagedSpans {
  SpanHead head;
  uint32_t spans[3 * kStagedSeqs];
  uint32_t tail_[2];                  // (two words of span 70: never read from here)
};
static_assert(sizeof(StagedSpans) == 64 * 16, "one 16-byte load per lane of one wave");
static_assert(offsetof(MoveArgs, inline_words) == sizeof(SpanHead), "spans follow the head");


__device__ __forceinline__ int find_key(const MoveTables& tb, int block) {
  int k = 0;
  while (k + 1 < tb.n_keys && block >= tb.first_block[k + 1]) ++k;
  return k;
}

// Pool row of step t of sequence seq.  (A launch's row table is n_seq
// sequences of tb.seq_len steps; a key's batch side holds the first klen of
// them, so batch row r of that key is (seq, t) = (r / klen, r % klen).)
// kLds: the launch's first kFlatStagedSeqs spans were staged into LDS (`lds`)
// together with the tables -- the indirect flat movers, whose argument block
// lives in uncached device memory (see flat_move_kernel_indirect).
constexpr int kFlatStagedSeqs = 64;      // 192 words = 48 lanes, one 16-byte load each
template <bool kLds = false>
__device__ __forceinline__ int32_t row_at(const MoveArgs& a, const MoveTables& tb, uint32_t seq, uint32_t t,
                                          const uint32_t* lds = nullptr) {
  if (tb.rows_mode == 2) {
    if (kLds && seq < static_cast<uint32_t>(kFlatStagedSeqs)) {
      const uint32_t row0 = lds[3 * seq], n0 = lds[3 * seq + 1];
      return static_cast<int32_t>(t < n0 ? row0 + t : lds[3 * seq + 2] + (t - n0));
    }
    const uint32_t row0 = a.inline_words[3 * seq], n0 = a.inline_words[3 * seq + 1];
    return static_cast<int32_t>(t < n0 ? row0 + t : a.inline_words[3 * seq + 2] + (t - n0));
  }
  const uint32_t r = seq * static_cast<uint32_t>(tb.seq_len) + t;
  if (tb.rows_mode == 1) return static_cast<int32_t>(a.inline_words[r]);
  return tb.rows[r];
}

// Byte offset of step t of sequence seq of `key` on the batch side (klen steps
// per sequence there; see MoveArgs::group).
__device__ __forceinline__ int64_t batch_offset(const MoveTables& tb, const KeyDesc& key, uint32_t klen,
                                                uint32_t seq, uint32_t t) {
  if (tb.group == 0) return static_cast<int64_t>(seq * klen + t) * key.rowbytes;
  const uint32_t g = static_cast<uint32_t>(tb.group);
  const uint32_t grp = seq / g, j = seq - grp * g;
  return static_cast<int64_t>(grp) * tb.group_stride + static_cast<int64_t>(j * klen + t) * key.rowbytes;
}

template <typename T>
__device__ __forceinline__ T gload(const void* p) {
  return *(const __attribute__((address_space(1))) T*)p;
}
template <typename T>
__device__ __forceinline__ void gstore(void* p, T v) {
  *(__attribute__((address_space(1))) T*)p = v;
}

template <typename T>
__device__ __forceinline__ void copy_unit(const uint8_t* s, uint8_t* d) {
  gstore<T>(d, gload<T>(s));
}

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void copy_bytes(const uint8_t* s, uint8_t* d, int unit) {
  switch (unit) {
    case 16: copy_unit<u32x4>(s, d); break;
    case 8: copy_unit<u32x2_t>(s, d); break;
    case 4: copy_unit<uint32_t>(s, d); break;
    case 2: copy_unit<uint16_t>(s, d); break;
    default: copy_unit<uint8_t>(s, d); break;
  }
}

// A key's rows are moved as a flat sequence of 16-byte units: unit u belongs to
// batch row u / upr.  A workgroup owns blockDim.x * U consecutive units; lane j
// takes units j, j + blockDim.x, ...: consecutive lanes touch consecutive
// 16 bytes on both sides, every lane has U independent row lookups and then U
// independent loads in flight before its first store, and no lane waits on a
// per-workgroup scalar dependency chain.
template <bool kGather, int U, int NT, bool kLds = false>
__device__ __forceinline__ void move_wide(const MoveArgs& a, const MoveTables& tb, const KeyDesc& key, int k,
                                          int local, const uint32_t* lds = nullptr) {
  const uint32_t upr = static_cast<uint32_t>(key.rowbytes >> 4);
  const uint32_t klen = static_cast<uint32_t>(tb.key_len[k]);
  const uint32_t nrows = static_cast<uint32_t>(tb.key_rows[k]);
  const uint32_t total = upr * nrows;
  // (Workgroup b runs on XCD b % 8 and takes tile b: every frame is spread over
  // all XCDs' memory channels, which a streaming copy prefers -- giving each XCD
  // one contiguous eighth of the batch instead measured 2-3 % slower.)
  // One division per wave for the workgroup's first unit; lanes step from it
  // with adds and compares (a 32-bit divide costs ~40 VALU instructions).
  const uint32_t base = static_cast<uint32_t>(local) * (blockDim.x * U);
  const uint32_t r0 = base / upr;
  const uint32_t off0 = base - r0 * upr;
  const uint32_t seq0 = r0 / klen, t0 = r0 - seq0 * klen;
  // The workgroup's units span at most kRows consecutive batch rows when rows
  // are long (the usual case: one 28 KB frame = 1764 units): resolve those rows
  // ONCE per wave with wave-uniform (scalar) reads of the inline span table and
  // let the lanes select, instead of every lane reading the table.
  constexpr int kRows = 3;
  const bool few_rows = tb.rows_mode == 2 &&
                        (off0 + blockDim.x * U - 1) / upr < static_cast<uint32_t>(kRows);
  int32_t row_tab[kRows];
  if (few_rows) {
    uint32_t seq = seq0, t = t0;
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      // (rows_mode 2 spelled out, not row_at: its other modes read through
      // tb.rows, and a wave-uniform choice between that pointer and one into the
      // by-value argument block makes the compiler copy the whole block -- 3.8 KB
      // per lane -- to scratch)
      if (r0 + i < nrows) {
        if (kLds && seq < static_cast<uint32_t>(kFlatStagedSeqs)) {
          const uint32_t start = lds[3 * seq], n0 = lds[3 * seq + 1];
          row_tab[i] = static_cast<int32_t>(t < n0 ? start + t : lds[3 * seq + 2] + (t - n0));
        } else {
          const uint32_t start = a.inline_words[3 * seq], n0 = a.inline_words[3 * seq + 1];
          row_tab[i] = static_cast<int32_t>(t < n0 ? start + t : a.inline_words[3 * seq + 2] + (t - n0));
        }
      } else {
        row_tab[i] = -1;
      }
      if (++t >= klen) { t = 0; ++seq; }
    }
  }
  uint32_t sq[U], tt[U], off[U];
  int32_t row[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    uint32_t x = off0 + j * blockDim.x + threadIdx.x, dr = 0;
    while (x >= upr) { x -= upr; ++dr; }
    uint32_t seq = seq0, t = t0 + dr;
    while (t >= klen) { t -= klen; ++seq; }
    sq[j] = seq;
    tt[j] = t;
    off[j] = x;
    if (base + j * blockDim.x + threadIdx.x >= total) {
      row[j] = -1;
    } else if (few_rows) {
      row[j] = dr == 0 ? row_tab[0] : dr == 1 ? row_tab[1] : row_tab[2];
    } else {
      row[j] = row_at<kLds>(a, tb, seq, t, lds);
    }
  }
  u32x4 buf[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    if (row[j] < 0) continue;
    const uint8_t* pool = key.pool + static_cast<int64_t>(row[j]) * key.rowbytes;
    const uint8_t* batch = key.batch + batch_offset(tb, key, klen, sq[j], tt[j]);
    const u32x4* src = reinterpret_cast<const u32x4*>(kGather ? pool : batch) + off[j];
    buf[j] = load16<(NT & 1) != 0>(src);
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    if (row[j] < 0) continue;
    uint8_t* pool = key.pool + static_cast<int64_t>(row[j]) * key.rowbytes;
    uint8_t* batch = key.batch + batch_offset(tb, key, klen, sq[j], tt[j]);
    u32x4* dst = reinterpret_cast<u32x4*>(kGather ? batch : pool) + off[j];
    store16<(NT & 2) != 0>(dst, buf[j]);
  }
}

// Span mode: every sequence is one or two contiguous runs of pool rows
// ({row0, n0, row1}: rows row0..row0+n0-1, then row1..), so per wide key a
// sequence is a flat run of L * rowbytes/16 units with ONE split point and no
// row structure at all.  `wide_workers` workgroups (a couple per CU) walk the
// tiles of blockDim.x * U units with a grid stride, and every lane issues the
// NEXT tile's loads before it stores the current one: loads and stores of one
// workgroup overlap and the launch has a single ramp instead of one per
// 8 KB workgroup.  Measured on MI355X (tools/gather_lab.hip, B=16, L=65,
// 28 224-byte rows, cold 2.8 GB pool): 10.0-10.2 us against 10.1 us for a
// plain contiguous copy of the same bytes and 13.8 us for the flat
// one-tile-per-workgroup mover above.
template <bool kGather, int U, int NT>
__device__ __forceinline__ void move_wide_spans(const MoveArgs& a, const StagedSpans& staged) {
  const SpanHead& h = staged.head;
  const uint32_t tile = blockDim.x * U;
  const uint32_t ntiles = h.ntiles;
  const uint32_t stride = static_cast<uint32_t>(h.wide_workers);
  struct Where {
    const u32x4* p0; const u32x4* p1;   // pool runs, both indexed by the unit number
    const u32x4* b;                     // batch side of the sequence
    uint32_t split, total, u0;
  };
  auto locate = [&](uint32_t ti) {
    // Which wide key: tile0[] beyond n_wide is 0xFFFFFFFF.
    const int k = (ti >= h.tile0[1]) + (ti >= h.tile0[2]) + (ti >= h.tile0[3]);
    const KeyDesc key = h.key[k];
    const uint32_t tps = h.tiles_per_seq[k];
    const uint32_t local = ti - h.tile0[k];
    const uint32_t seq = local / tps, piece = local - seq * tps;
    const uint32_t upr = static_cast<uint32_t>(key.rowbytes >> 4);
    // The first kStagedSeqs spans came with the head (LDS); later ones are read
    // from the argument block.
    // (No pointer into `a` here: a by-value argument block whose address
    // escapes is copied to scratch, 3.7 KB per lane.)
    uint32_t row0, n0, row1;
    if (seq < kStagedSeqs) {
      row0 = staged.spans[3 * seq], n0 = staged.spans[3 * seq + 1], row1 = staged.spans[3 * seq + 2];
    } else {
      row0 = a.inline_words[3 * seq], n0 = a.inline_words[3 * seq + 1], row1 = a.inline_words[3 * seq + 2];
    }
    Where w;
    w.split = n0 * upr;
    w.total = h.units_per_seq[k];     // (a context-only key: fewer than L * upr, all of them maybe left of the split)
    w.u0 = piece * tile + threadIdx.x;
    w.p0 = reinterpret_cast<const u32x4*>(key.pool) + static_cast<uint64_t>(row0) * upr;
    w.p1 = reinterpret_cast<const u32x4*>(key.pool) + static_cast<uint64_t>(row1) * upr - w.split;
    if (h.group == 0) {
      w.b = reinterpret_cast<const u32x4*>(key.batch) + static_cast<uint64_t>(seq) * w.total;
    } else {
      const uint32_t g = static_cast<uint32_t>(h.group), grp = seq / g, j = seq - grp * g;
      w.b = reinterpret_cast<const u32x4*>(key.batch + static_cast<int64_t>(grp) * h.group_stride) +
            static_cast<uint64_t>(j) * w.total;
    }
    return w;
  };
  auto issue = [&](const Where& w, u32x4* v) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = w.u0 + j * blockDim.x;
      if (u < w.total) {
        const u32x4* src = kGather ? (u < w.split ? w.p0 : w.p1) + u : w.b + u;
        v[j] = load16<(NT & 1) != 0>(src);
      }
    }
  };
  auto put = [&](const Where& w, const u32x4* v) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = w.u0 + j * blockDim.x;
      if (u < w.total) {
        u32x4* dst = const_cast<u32x4*>(kGather ? w.b + u : (u < w.split ? w.p0 : w.p1) + u);
        store16<(NT & 2) != 0>(dst, v[j]);
      }
    }
  };
  uint32_t i = blockIdx.x;
  if (i >= ntiles) return;
  u32x4 cur[U], nxt[U];
  Where wc = locate(i);
  issue(wc, cur);
  for (;;) {
    const uint32_t n = i + stride;
    Where wn = wc;
    if (n < ntiles) {
      wn = locate(n);
      issue(wn, nxt);
    }
    put(wc, cur);
    if (n >= ntiles) break;
#pragma unroll
    for (int j = 0; j < U; ++j) cur[j] = nxt[j];
    wc = wn;
    i = n;
  }
}

// pool[rows[r]] -> batch[r] for every key of the replay in ONE launch, with
// the is_first / is_last annotation of replay.py:277-292 applied in flight.
template <int U, int NT, bool kLds = false>
__device__ __forceinline__ void gather_block(const MoveArgs& a, const MoveTables& tb, int block,
                                             const uint32_t* lds = nullptr) {
  const int k = find_key(tb, block);
  const KeyDesc key = tb.key[k];
  const int local = block - tb.first_block[k];
  const int unit = tb.unit[k];
  if (unit == 0) {
    move_wide<true, U, NT, kLds>(a, tb, key, k, local, lds);
    return;
  }
  const uint32_t klen = static_cast<uint32_t>(tb.key_len[k]);
  const int64_t upr = key.rowbytes / unit;
  const int64_t u = static_cast<int64_t>(local) * blockDim.x + threadIdx.x;
  if (u >= upr * tb.key_rows[k]) return;
  const int64_t r = u / upr;
  const int64_t off = (u - r * upr) * unit;
  const uint32_t seq = static_cast<uint32_t>(r) / klen, t = static_cast<uint32_t>(r) - seq * klen;
  const int64_t row = row_at<kLds>(a, tb, seq, t, lds);
  if (row < 0) return;   // not this rank's sequence (sharded pools): leave as is
  const uint8_t* src = key.pool + row * key.rowbytes + off;
  uint8_t* dst = key.batch + batch_offset(tb, key, klen, seq, t) + off;
  if (key.rowbytes == 1 && (k == tb.key_is_first || k == tb.key_is_last)) {
    // (annotated over the FULL sequence, then cut to the key's head: what
    // slicing the reference's annotated batch gives)
    uint8_t v = gload<uint8_t>(src);
    if (k == tb.key_is_first) {
      if (t == 0) v = 1;
    } else if (tb.is_first_pool && t + 1 < static_cast<uint32_t>(tb.seq_len)) {
      v |= gload<uint8_t>(tb.is_first_pool + row_at<kLds>(a, tb, seq, t + 1, lds));
    }
    gstore<uint8_t>(dst, v);
    return;
  }
  copy_bytes(src, dst, unit);
}

// One element of a masked key: value * keep in the key's own dtype — a real
// multiply, so -x -> -0.0 and NaN stays NaN exactly as numpy's
// `value * mask.astype(value.dtype)` (driver.py:84-87) — to the pool row and to
// the masked-action buffer.
template <typename T>
__device__ __forceinline__ void put_masked(const uint8_t* src, uint8_t* pool, uint8_t* out, bool keep) {
  const T v = gload<T>(src) * static_cast<T>(keep ? 1 : 0);
  if (pool) gstore<T>(pool, v);
  if (out) gstore<T>(out, v);
}

__device__ __forceinline__ void put_masked_bf16(const uint8_t* src, uint8_t* pool, uint8_t* out, bool keep) {
  // Widen to f32 (exact), multiply, narrow: the product is x, +-0 or NaN.
  const float x = __uint_as_float(static_cast<uint32_t>(gload<uint16_t>(src)) << 16);
  const uint16_t v = static_cast<uint16_t>(__float_as_uint(x * (keep ? 1.f : 0.f)) >> 16);
  if (pool) gstore<uint16_t>(pool, v);
  if (out) gstore<uint16_t>(out, v);
}

template <bool kLds = false>
__device__ __forceinline__ void scatter_masked(const MoveArgs& a, const MoveTables& tb, int k, const KeyDesc& key, int local,
                                               const uint32_t* lds = nullptr) {
  const int es = tb.unit[k];                       // element size of the key's dtype
  const int64_t epr = key.rowbytes / es;
  const int64_t e = static_cast<int64_t>(local) * blockDim.x + threadIdx.x;
  if (e >= epr * tb.n_rows) return;
  const int64_t r = e / epr;
  const int64_t off = (e - r * epr) * es;
  const uint32_t L = static_cast<uint32_t>(tb.seq_len), seq = static_cast<uint32_t>(r) / L;
  const int64_t row = row_at<kLds>(a, tb, seq, static_cast<uint32_t>(r) - seq * L, lds);
  const bool keep = gload<uint8_t>(tb.mask_flags + r) == 0;
  const uint8_t* src = key.batch + r * key.rowbytes + off;
  uint8_t* pool = row >= 0 ? key.pool + row * key.rowbytes + off : nullptr;
  uint8_t* out = tb.mask_out[k] ? tb.mask_out[k] + r * key.rowbytes + off : nullptr;
  switch (tb.mask_dtype[k]) {
    case kU8: case kBool: put_masked<uint8_t>(src, pool, out, keep); break;
    case kI8: put_masked<int8_t>(src, pool, out, keep); break;
    case kI16: put_masked<int16_t>(src, pool, out, keep); break;
    case kI32: put_masked<int32_t>(src, pool, out, keep); break;
    case kI64: put_masked<int64_t>(src, pool, out, keep); break;
    case kF16: put_masked<_Float16>(src, pool, out, keep); break;
    case kBF16: put_masked_bf16(src, pool, out, keep); break;
    case kF32: put_masked<float>(src, pool, out, keep); break;
    default: put_masked<double>(src, pool, out, keep); break;
  }
}

// batch[r] -> pool[rows[r]]; rows[r] < 0 are skipped (evicted update targets).
template <int U, int NT, bool kLds = false>
__device__ __forceinline__ void scatter_block(const MoveArgs& a, const MoveTables& tb, int block,
                                              const uint32_t* lds = nullptr) {
  const int k = find_key(tb, block);
  const KeyDesc key = tb.key[k];
  const int local = block - tb.first_block[k];
  const int unit = tb.unit[k];
  if ((tb.mask_bits >> k) & 1u) {
    scatter_masked<kLds>(a, tb, k, key, local, lds);
    return;
  }
  if (unit == 0) {
    move_wide<false, U, NT, kLds>(a, tb, key, k, local, lds);
    return;
  }
  const int64_t upr = key.rowbytes / unit;
  const int64_t u = static_cast<int64_t>(local) * blockDim.x + threadIdx.x;
  if (u >= upr * tb.n_rows) return;
  const int64_t r = u / upr;
  const int64_t off = (u - r * upr) * unit;
  const uint32_t L = static_cast<uint32_t>(tb.seq_len), seq = static_cast<uint32_t>(r) / L;
  const int64_t row = row_at<kLds>(a, tb, seq, static_cast<uint32_t>(r) - seq * L, lds);
  if (row < 0) return;
  if (k == tb.inline_key) {   // batch bytes of this key ride in the kernel arguments
    const uint32_t w = a.inline_words[tb.inline_key_word0 + r * (key.rowbytes >> 2) + (off >> 2)];
    gstore<uint32_t>(key.pool + row * key.rowbytes + off, w);
    return;
  }
  copy_bytes(key.batch + r * key.rowbytes + off, key.pool + row * key.rowbytes + off, unit);
}

// A flat launch has first_block[n_keys] virtual blocks and exactly that many
// workgroups.  (A grid capped at a few workgroups per CU that walks the virtual
// blocks with a stride measured equal at B=16 and 7-8 % slower at B=64/256: the
// dispatcher overlaps one workgroup's stores with the next one's loads better.)
constexpr int kFlatUnroll = 2;      // 16-byte units per lane: 2 beats 4/8 by 10-15 % (MI355X sweep)
constexpr int kFlatNT = 3;          // non-temporal loads and stores: ~3 % faster than plain on cold lines
constexpr int kFlatThreads = 256;   // 64..512 within noise, 1024 slower

// The kernel-argument segment as raw 16-byte units (MoveArgs is the only
// parameter of every mover, so it starts the segment).  Indexing the by-value
// parameter itself with a lane id would make the compiler copy the whole
// 3.7 KB struct into scratch, per lane (260 us instead of 11).
__device__ __forceinline__ const u32x4* kernarg_units() {
  return (const u32x4*)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr();
}

// Host-resident kernel arguments: every scalar read of the plan is a PCIe round
// trip, and a flat-mover wave makes three or four dependent ones (which key is
// mine -> its descriptor -> my rows) before its first payload load.  The staged
// variants bring the tables into LDS with one 16-byte load per lane of the first
// wave: one round trip, then LDS reads; only the inline row words still come from
// the argument block.  (Insert scatter of 64 rows: 13.5 -> ~9 us.)
__device__ __forceinline__ void stage_tables(const u32x4* bytes, MoveTables* dst) {
  constexpr uint32_t first = offsetof(MoveArgs, t) / 16;
  if (threadIdx.x < sizeof(MoveTables) / 16)
    reinterpret_cast<u32x4*>(dst)[threadIdx.x] = bytes[first + threadIdx.x];
  __syncthreads();
}
static_assert(offsetof(MoveArgs, t) % 16 == 0, "the tables are staged in 16-byte units");

// Each mover exists three times: arguments by value in the kernel-argument
// segment (default), the same with the tables staged through LDS (arguments in
// host memory, small launches), or read through a pointer to a copy in device
// memory (arguments in host memory, big launches: abi.cpp run_move).
template <bool kGather, bool kLds = false>
__device__ __forceinline__ void flat_block(const MoveArgs& a, const MoveTables& tb, const uint32_t* lds = nullptr) {
  if (kGather) gather_block<kFlatUnroll, kFlatNT, kLds>(a, tb, blockIdx.x, lds);
  else scatter_block<kFlatUnroll, kFlatNT, kLds>(a, tb, blockIdx.x, lds);
}
template <bool kGather>
__global__ __launch_bounds__(kFlatThreads) void flat_move_kernel(const MoveArgs a) { flat_block<kGather>(a, a.t); }
template <bool kGather>
__global__ __launch_bounds__(kFlatThreads) void flat_move_kernel_staged(const MoveArgs a) {
  __shared__ MoveTables tables;
  stage_tables(kernarg_units(), &tables);
  flat_block<kGather>(a, tables);
}
// The device copy of the argument block sits in a ring of fine-grained (uncached)
// memory the CPU wrote through the BAR: every read of it goes to memory.  The
// tables AND the first kFlatStagedSeqs spans therefore come into LDS with one
// 16-byte load per lane -- wave 0 the tables, wave 1 the spans, both in flight
// together: one latency per workgroup, where a wave's scalar reads of its spans
// were a second, dependent one in front of every payload load (an 85 MB
// write-back through this mover: 20.3 us before, 17.1 us now; the by-value mover
// with device-resident arguments takes 13 us).
static_assert(kFlatThreads >= 64 + 3 * kFlatStagedSeqs / 4, "wave 1 stages the spans");
static_assert(3 * kFlatStagedSeqs <= kInlineWords && sizeof(SpanHead) % 16 == 0, "the staged spans lie inside the block");
template <bool kGather>
__global__ __launch_bounds__(kFlatThreads) void flat_move_kernel_indirect(const MoveArgs* __restrict__ a) {
  __shared__ MoveTables tables;     // one load latency instead of a chain of scalar loads
  __shared__ __attribute__((aligned(16))) uint32_t spans[3 * kFlatStagedSeqs];
  const u32x4* bytes = reinterpret_cast<const u32x4*>(a);
  constexpr uint32_t first_table = offsetof(MoveArgs, t) / 16, first_span = sizeof(SpanHead) / 16;
  if (threadIdx.x < sizeof(MoveTables) / 16)
    reinterpret_cast<u32x4*>(&tables)[threadIdx.x] = bytes[first_table + threadIdx.x];
  else if (threadIdx.x >= 64 && threadIdx.x < 64 + 3 * kFlatStagedSeqs / 4)
    reinterpret_cast<u32x4*>(spans)[threadIdx.x - 64] = bytes[first_span + threadIdx.x - 64];
  __syncthreads();
  flat_block<kGather, true>(*a, tables, spans);
}

// Span-mode launch: the first `wide_workers` workgroups are the persistent wide
// movers, the rest are the virtual blocks of the narrow keys.
// The 160-byte head goes from the argument block to LDS with one 16-byte load
// per lane of the first ten lanes — a single memory latency however the
// compiler would have scheduled the individual field reads.
// `bytes`: the argument block as raw 16-byte units — the device copy for the
// indirect kernels, the kernel-argument segment itself for the by-value ones.
__device__ __forceinline__ void stage_head(const u32x4* bytes, StagedSpans* dst) {
  if (threadIdx.x < sizeof(StagedSpans) / 16)
    reinterpret_cast<u32x4*>(dst)[threadIdx.x] = bytes[threadIdx.x];
  __syncthreads();
}

// Shape of the persistent mover (MI355X, profiles/r04_gather_shapes.txt and
// r04_ab_span_shape.txt): 256 threads x 4 units per lane = tiles of 16 KB, four
// workgroups per CU, every worker slot filled -- B=16 10.1-10.3 us against
// 10.8-11.2 for 512 threads x 2 per CU; U=2 and 1024-thread shapes were slower.
constexpr int kSpanUnroll = 4;
constexpr int kSpanThreads = 256;
constexpr int kSpanPerCU = 4;
template <bool kGather, int NT>
__device__ __forceinline__ void span_move_body(const MoveArgs& a, const u32x4* bytes) {
  __shared__ StagedSpans staged;
  stage_head(bytes, &staged);
  if (static_cast<int>(blockIdx.x) < staged.head.wide_workers) {
    move_wide_spans<kGather, kSpanUnroll, NT>(a, staged);
    return;
  }
  const int block = static_cast<int>(blockIdx.x) - staged.head.wide_workers;
  if (kGather) gather_block<kFlatUnroll, NT>(a, a.t, block);
  else scatter_block<kFlatUnroll, NT>(a, a.t, block);
}

// NT: non-temporal hints, bit 0 loads, bit 1 stores.  3 (both) is the fastest
// gather by itself (B=16: 10.7 us against 11.2 / 10.9 / 12.2 for loads-only /
// stores-only / none); 1 (plain stores) leaves the batch in L2 / Infinity Cache
// for a reader that follows at once (EMB_GATHER_STORES=plain).
template <bool kGather, int NT>
__global__ __launch_bounds__(kSpanThreads) void span_move_kernel(const MoveArgs a) {
  // MoveArgs is the only parameter: it starts the kernel-argument segment.
  span_move_body<kGather, NT>(a, kernarg_units());
}
template <bool kGather, int NT>
__global__ __launch_bounds__(kSpanThreads) void span_move_kernel_indirect(const MoveArgs* __restrict__ a) {
  span_move_body<kGather, NT>(*a, reinterpret_cast<const u32x4*>(a));
}

// Host-resident kernel arguments (HIP_FORCE_DEV_KERNARG=0): one workgroup
// copies the mover's argument block from the kernel-argument segment into
// device memory, so that the mover's thousands of waves read it from L2
// instead of each crossing PCIe.
// One 16-byte load per lane: the whole block crosses PCIe in a single round
// trip (word-sized loads took four, ~10 us).
__global__ __launch_bounds__(256) void args_writer_kernel(const MoveArgs a, u32x4* __restrict__ dst) {
  const u32x4* src = reinterpret_cast<const u32x4*>(&a);
  for (uint32_t i = threadIdx.x; i < sizeof(MoveArgs) / 16; i += blockDim.x) dst[i] = src[i];
}
static_assert(sizeof(MoveArgs) % 16 == 0 && sizeof(MoveArgs) / 16 <= 256,
              "argument block is copied as one dwordx4 per lane");

// Which mover a launch gets (prepare_move).  The persistent span mover wins
// clearly while the launch is ramp-dominated and stays level with the flat
// mover's many short-lived workgroups far beyond that (MI355X, S0 rows, kernel us
// persistent / flat: B=8 6.8 / 8.7, B=16 10.7 / 13.4, B=32 22.5 / 23.5, B=64
// 41.7 / 42.1, B=128 81.6 / 79.4; Dreamer keys, 144 MB: 26.8 / 28.7), so gathers
// use it up to kSpanGatherMB of wide payload per launch.  Write-backs up to
// kSpanScatterMB: the flat scatter streams 84 MB of Dreamer latents at
// 6.4-6.8 TB/s (13 us), the persistent one takes 17.6 us for the same bytes.
// EMB_SPAN_MOVER=0 sends everything to the flat mover (the fallback).
constexpr int64_t kSpanGatherMB = 160, kSpanScatterMB = 40;
bool span_mover_enabled() {
  static const bool value = [] {
    const char* e = emb::knob("EMB_SPAN_MOVER");
    return !(e && e[0] == '0');
  }();
  return value;
}
// EMB_GATHER_STORES=plain: sample gathers store with plain instead of
// non-temporal stores.  What `nt` stores cost is paid by the NEXT reader of the
// batch, which finds nothing of it in L2 / Infinity Cache: measured with a kernel
// that reads 84 MB of the batch right behind the gather (rocprofv3 medians over
// five GPUs): 13.0-13.4 us behind plain stores, 16.6-18.3 us behind `nt` stores.
// The gather itself is faster with `nt` while the batch is small (60 MB: 12.5
// against 13.5 us) and slower when it is large (144 MB: 26.7 against 25.3 us;
// profiles/r05_ab_gather_stores.txt).  `plain` is the setting for a learner that
// reads the whole batch right after sampling.
int gather_nt() {      // non-temporal hints of a span gather: bit 0 loads, bit 1 stores
  static const int value = [] {
    const char* e = emb::knob("EMB_GATHER_STORES");
    return e && e[0] == 'p' ? 1 : 3;
  }();
  return value;
}
// Compute units of the current device (MI355X: 256), asked once: the persistent
// span mover sizes its grid by it.
int compute_units() {
  static const int value = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    (void)hipGetLastError();
    return n;
  }();
  return value;
}

int pick_unit(const KeyDesc& key) {
  const uint64_t mix = reinterpret_cast<uint64_t>(key.pool) |
                       reinterpret_cast<uint64_t>(key.batch) |
                       static_cast<uint64_t>(key.rowbytes);
  if (mix % 16 == 0) return key.rowbytes >= 2048 ? 0 : 16;
  if (mix % 8 == 0) return 8;
  if (mix % 4 == 0) return 4;
  if (mix % 2 == 0) return 2;
  return 1;
}

int dtype_size(int dtype) {
  switch (dtype) {
    case kU8: case kI8: case kBool: return 1;
    case kI16: case kF16: case kBF16: return 2;
    case kI32: case kF32: return 4;
    case kI64: case kF64: return 8;
    default: return 0;
  }
}

// Words of kernel-argument space the plan's tables need, or -1 if they cannot
// go inline.
int inline_words_needed(const MovePlan& plan) {
  int64_t words = 0;
  if (plan.spans_host) words = 3ll * plan.n_seq;
  else if (plan.rows_host) words = plan.n_rows;
  else return plan.inline_key >= 0 ? -1 : 0;
  if (plan.inline_key >= 0) words += static_cast<int64_t>(plan.n_rows) * (plan.key[plan.inline_key].rowbytes >> 2);
  return words <= kInlineWords ? static_cast<int>(words) : -1;
}

static_assert(sizeof(MoveArgs) <= kMoveArgsBytes, "MoveLaunch::args too small");

}  // namespace

hipError_t prepare_move(const MovePlan& plan, MoveLaunch* out, bool gather) {
  out->blocks = 0;
  const int inline_need = inline_words_needed(plan);
  const bool use_inline = inline_need > 0 && (plan.spans_host || plan.rows_host);
  if (plan.n_keys < 1 || plan.n_keys > kMaxKeys || plan.n_rows < 0 || (!plan.rows && !use_inline))
    return hipErrorInvalidValue;
  if (plan.n_rows == 0) return hipSuccess;
  // Span tables (sample, windowing, write-back) with at least one wide key go
  // to the persistent span mover; everything else to the flat mover.
  bool span_path = false;
  if (use_inline && plan.spans_host && span_mover_enabled() && plan.seq_len >= 1) {
    int64_t wide_bytes = 0;
    int wide_keys = 0;
    for (int k = 0; k < plan.n_keys; ++k)
      if (!((plan.mask_bits >> k) & 1u) && k != plan.inline_key && pick_unit(plan.key[k]) == 0) {
        const int64_t len = plan.key_len[k] > 0 && plan.key_len[k] < plan.seq_len ? plan.key_len[k] : plan.seq_len;
        wide_bytes += plan.key[k].rowbytes * (static_cast<int64_t>(plan.n_rows) / plan.seq_len) * len;
        ++wide_keys;
      }
    // With host-resident kernel arguments the big movers read their plan from a
    // ring in fine-grained (uncached) device memory: the span mover touches it
    // once per workgroup (the staged head), ~1000 workgroups; the flat mover's
    // ~10 000 short-lived workgroups each fetch their tables and spans from it
    // (staged through LDS since round 6: an 85 MB write-back 20.3 -> 17.1 us,
    // still behind the span mover's 15.2 us; profiles/r06_scatter_lab.txt) --
    // there the span mover takes every size.
    const int64_t limit_mb = gather ? kSpanGatherMB : kSpanScatterMB;
    span_path = wide_bytes > 0 && wide_keys <= kSpanKeys &&
                (plan.args_in_host_memory || wide_bytes <= limit_mb * 1000000);
  }
  const int unroll = span_path ? kSpanUnroll : kFlatUnroll;
  const int threads = span_path ? kSpanThreads : kFlatThreads;
  out->span = span_path;
  out->stage_tables = plan.args_in_host_memory;
  out->nt = gather && span_path ? gather_nt() : 3;
  MoveArgs& a = *reinterpret_cast<MoveArgs*>(out->args);
  SpanHead& h = a.head;
  MoveTables& t = a.t;
  std::memset(&h, 0, sizeof(h));
  for (int j = 0; j < kSpanKeys; ++j) h.tile0[j] = 0xFFFFFFFFu;
  t.group = plan.group > 0 ? plan.group : 0;
  t.group_stride = plan.group_stride;
  h.group = t.group;
  h.group_stride = t.group_stride;
  if (t.group && (plan.group_stride % 16 != 0 || plan.mask_bits || plan.inline_key >= 0))
    return hipErrorInvalidValue;       // gather-side layout only, 16-byte aligned groups
  t.n_keys = plan.n_keys;
  t.n_rows = plan.n_rows;
  t.seq_len = plan.seq_len < 1 ? 1 : plan.seq_len;
  // Context-only keys (gather): key k moves the first key_len[k] steps of every
  // sequence into a (n_seq, key_len[k], rowbytes) array.
  bool heads = false;
  for (int k = 0; k < plan.n_keys; ++k) {
    if (plan.key_len[k] < 0 || plan.key_len[k] > t.seq_len) return hipErrorInvalidValue;
    heads = heads || (plan.key_len[k] > 0 && plan.key_len[k] < t.seq_len);
  }
  if (heads && (!gather || plan.n_rows % t.seq_len != 0 || plan.mask_bits || plan.inline_key >= 0))
    return hipErrorInvalidValue;
  const int64_t n_seq = plan.n_rows / t.seq_len;
  t.key_is_first = plan.key_is_first;
  t.key_is_last = plan.key_is_last;
  t.is_first_pool = plan.is_first_pool;
  if (!t.is_first_pool && plan.key_is_first >= 0) t.is_first_pool = plan.key[plan.key_is_first].pool;
  t.rows = plan.rows;
  t.rows_mode = 0;
  t.inline_key = -1;
  t.inline_key_word0 = 0;
  if (use_inline) {
    int words;
    if (plan.spans_host) {
      t.rows_mode = 2;
      words = 3 * plan.n_seq;
      std::memcpy(a.inline_words, plan.spans_host, sizeof(uint32_t) * words);
    } else {
      t.rows_mode = 1;
      words = plan.n_rows;
      std::memcpy(a.inline_words, plan.rows_host, sizeof(uint32_t) * words);
    }
    if (plan.inline_key >= 0) {
      t.inline_key = plan.inline_key;
      t.inline_key_word0 = words;
      std::memcpy(a.inline_words + words, plan.inline_bytes,
                  static_cast<size_t>(plan.n_rows) * plan.key[plan.inline_key].rowbytes);
    }
  } else if (plan.inline_key >= 0) {
    return hipErrorInvalidValue;
  }
  t.mask_bits = plan.mask_bits;
  t.mask_flags = plan.mask_flags;
  if (plan.mask_bits && !plan.mask_flags) return hipErrorInvalidValue;
  int64_t blocks = 0;
  for (int k = 0; k < plan.n_keys; ++k) {
    t.key[k] = plan.key[k];
    t.key_len[k] = heads && plan.key_len[k] > 0 ? plan.key_len[k] : t.seq_len;
    t.key_rows[k] = heads ? static_cast<int32_t>(n_seq * t.key_len[k]) : plan.n_rows;
    t.mask_dtype[k] = plan.mask_dtype[k];
    t.mask_out[k] = plan.mask_out[k];
    const bool masked = (plan.mask_bits >> k) & 1u;
    if (masked) {
      const int es = dtype_size(plan.mask_dtype[k]);
      if (es == 0 || plan.key[k].rowbytes % es || k == t.inline_key) return hipErrorInvalidValue;
      t.unit[k] = es;
    } else
    t.unit[k] = (k == t.inline_key) ? 4 : pick_unit(plan.key[k]);
    t.first_block[k] = static_cast<int32_t>(blocks);
    if (t.unit[k] == 0 && span_path) {
      // tiles of threads * unroll units per sequence; no virtual blocks
      const int64_t per_seq = static_cast<int64_t>(t.key_len[k]) * (plan.key[k].rowbytes >> 4);
      const int64_t tps = (per_seq + threads * unroll - 1) / (threads * unroll);
      const int64_t first = h.ntiles;
      if (per_seq > UINT32_MAX / 2 || first + tps * plan.n_seq > UINT32_MAX / 2) return hipErrorInvalidValue;
      h.key[h.n_wide] = plan.key[k];
      h.tiles_per_seq[h.n_wide] = static_cast<uint32_t>(tps);
      h.units_per_seq[h.n_wide] = static_cast<uint32_t>(per_seq);
      h.tile0[h.n_wide] = static_cast<uint32_t>(first);
      h.ntiles = static_cast<uint32_t>(first + tps * plan.n_seq);
      ++h.n_wide;
    } else if (t.unit[k] == 0) {
      const int64_t units = static_cast<int64_t>(t.key_rows[k]) * (plan.key[k].rowbytes >> 4);
      if (units > UINT32_MAX / 2) return hipErrorInvalidValue;
      blocks += (units + threads * unroll - 1) / (threads * unroll);
    } else {
      const int64_t units = static_cast<int64_t>(t.key_rows[k]) * (plan.key[k].rowbytes / t.unit[k]);
      blocks += (units + threads - 1) / threads;
    }
    if (blocks > INT32_MAX) return hipErrorInvalidValue;
  }
  t.first_block[plan.n_keys] = static_cast<int32_t>(blocks);
  out->blocks = static_cast<uint32_t>(blocks);
  if (span_path) {
    // As many workers as the chip takes at once (trimming the count so that
    // every worker walks the same number of tiles was slower with this shape:
    // B=16 10.8 against 10.1 us).
    const int cus = plan.cu_limit > 0 ? std::min(plan.cu_limit, compute_units()) : compute_units();
    const int64_t workers = std::min<int64_t>(h.ntiles, int64_t(cus) * kSpanPerCU);
    h.wide_workers = static_cast<int32_t>(workers);
    h.seq_len = t.seq_len;
    if (blocks + h.wide_workers > INT32_MAX) return hipErrorInvalidValue;
    out->blocks = static_cast<uint32_t>(blocks + h.wide_workers);
  }
  out->threads = static_cast<uint32_t>(threads);
  return hipSuccess;
}

size_t move_args_bytes() { return sizeof(MoveArgs); }

namespace {
__global__ void marker_kernel() {}
}  // namespace

hipError_t launch_marker(hipStream_t stream, hipEvent_t stop) {
  hipExtLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, stream, nullptr, stop, 0);
  return hipGetLastError();
}

hipError_t launch_args_writer(const MoveLaunch& launch, void* device_dst, hipStream_t stream,
                              hipEvent_t stop) {
  const MoveArgs& a = *reinterpret_cast<const MoveArgs*>(launch.args);
  // (One writer: one on every XCD -- the block then is in every L2 when the mover
  // asks for it -- measured -0.15 us on the gather and +3.2 us on this kernel.)
  hipExtLaunchKernelGGL(args_writer_kernel, dim3(1), dim3(256), 0, stream, nullptr, stop, 0, a,
                        static_cast<u32x4*>(device_dst));
  return hipGetLastError();
}

hipError_t launch_move(const MoveLaunch& launch, bool gather, const void* device_args,
                       hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  if (launch.blocks == 0) return hipSuccess;
  const MoveArgs& a = *reinterpret_cast<const MoveArgs*>(launch.args);
  const MoveArgs* ap = static_cast<const MoveArgs*>(device_args);
  const dim3 grid(launch.blocks), block(launch.threads);
#define EMB_LAUNCH(BYVALUE_, INDIRECT_)                                                              \
  do {                                                                                               \
    if (ap) hipExtLaunchKernelGGL(INDIRECT_, grid, block, 0, stream, start, stop, 0, ap);            \
    else hipExtLaunchKernelGGL(BYVALUE_, grid, block, 0, stream, start, stop, 0, a);                 \
  } while (0)
  if (launch.span) {
    if (!gather) EMB_LAUNCH((span_move_kernel<false, 3>), (span_move_kernel_indirect<false, 3>));
    else if (launch.nt == 1) EMB_LAUNCH((span_move_kernel<true, 1>), (span_move_kernel_indirect<true, 1>));
    else EMB_LAUNCH((span_move_kernel<true, 3>), (span_move_kernel_indirect<true, 3>));
  } else if (launch.stage_tables && !ap) {
    if (gather) hipExtLaunchKernelGGL((flat_move_kernel_staged<true>), grid, block, 0, stream, start, stop, 0, a);
    else hipExtLaunchKernelGGL((flat_move_kernel_staged<false>), grid, block, 0, stream, start, stop, 0, a);
  } else {
    if (gather) EMB_LAUNCH((flat_move_kernel<true>), (flat_move_kernel_indirect<true>));
    else EMB_LAUNCH((flat_move_kernel<false>), (flat_move_kernel_indirect<false>));
  }
#undef EMB_LAUNCH
  return hipGetLastError();
}

// The kernel launch_move picks for this launch, spelled as the profiler prints
// it (without namespaces and parameter list).
const char* move_kernel_name(const MoveLaunch& launch, bool gather, bool indirect) {
  static thread_local char name[96];
  if (launch.span)
    std::snprintf(name, sizeof(name), "span_move_kernel%s<%s, %d>", indirect ? "_indirect" : "",
                  gather ? "true" : "false", gather ? launch.nt : 3);
  else
    std::snprintf(name, sizeof(name), "flat_move_kernel%s<%s>",
                  indirect ? "_indirect" : launch.stage_tables ? "_staged" : "", gather ? "true" : "false");
  return name;
}

namespace {

// ---------------------------------------------------------------- windowing --

__global__ __launch_bounds__(kThreads) void window_kernel(
    const uint8_t* src, uint8_t* dst, int64_t total, int64_t start, int64_t count,
    int64_t rowbytes, int unit, int64_t units_per_seq) {
  const int64_t b = blockIdx.y;
  const uint8_t* s = src + (b * total + start) * rowbytes;
  uint8_t* d = dst + b * count * rowbytes;
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
       u < units_per_seq; u += static_cast<int64_t>(gridDim.x) * kThreads)
    copy_bytes(s + u * unit, d + u * unit, unit);
}

// ---------------------------------------------------------------- obs stack --

template <typename Out>
__device__ __forceinline__ Out cvt(uint8_t v, float scale, float offset);
template <> __device__ __forceinline__ uint8_t cvt<uint8_t>(uint8_t v, float, float) { return v; }
template <> __device__ __forceinline__ float cvt<float>(uint8_t v, float s, float o) { return fmaf(static_cast<float>(v), s, o); }
template <> __device__ __forceinline__ __half cvt<__half>(uint8_t v, float s, float o) { return __float2half(fmaf(static_cast<float>(v), s, o)); }
template <> __device__ __forceinline__ __hip_bfloat16 cvt<__hip_bfloat16>(uint8_t v, float s, float o) { return __float2bfloat16(fmaf(static_cast<float>(v), s, o)); }

template <typename Out>
struct alignas(4 * sizeof(Out)) Quad { Out v[4]; };

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Streaming store of one Quad (4, 8 or 16 bytes): the policy batch is consumed
// by another kernel, keeping it dirty in this XCD's L2 only delays the next
// kernel boundary.
template <typename Out>
__device__ __forceinline__ void store_quad(Out* dst, const Quad<Out>& q) {
  if constexpr (sizeof(Quad<Out>) == 4) {
    uint32_t w;
    __builtin_memcpy(&w, &q, 4);
    __builtin_nontemporal_store(w, reinterpret_cast<uint32_t*>(dst));
  } else if constexpr (sizeof(Quad<Out>) == 8) {
    u32x2 w;
    __builtin_memcpy(&w, &q, 8);
    __builtin_nontemporal_store(w, reinterpret_cast<u32x2*>(dst));
  } else {
    u32x4 w;
    __builtin_memcpy(&w, &q, 16);
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(dst));
  }
}

// Each lane owns 4 consecutive pixels of one frame: it reads their 4*C bytes as
// C dwords (coalesced across lanes) and writes, per channel, one 4-element
// vector.  Output is (N, C, P) channels-first or (N, P, C) as stored.
template <typename Out, int C, bool kChannelsFirst>
__global__ __launch_bounds__(kThreads) void obs_stack_kernel(
    const uint8_t* src, const int32_t* env_ids, Out* dst, int64_t pixels,
    float scale, float offset) {
  const int64_t n = blockIdx.y;
  const int64_t e = env_ids ? env_ids[n] : n;
  const int64_t quads = pixels >> 2;
  const uint32_t* frame = reinterpret_cast<const uint32_t*>(src + e * pixels * C);
  Out* out = dst + n * pixels * C;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; q < quads;
       q += static_cast<int64_t>(gridDim.x) * kThreads) {
    uint32_t w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = frame[q * C + c];
    auto byte_at = [&w](int idx) {
      return static_cast<uint8_t>((w[idx >> 2] >> ((idx & 3) * 8)) & 0xFFu);
    };
    if (kChannelsFirst) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        Quad<Out> o;
#pragma unroll
        for (int p = 0; p < 4; ++p) o.v[p] = cvt<Out>(byte_at(p * C + c), scale, offset);
        store_quad(out + c * pixels + q * 4, o);
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) {
        Quad<Out> o;
#pragma unroll
        for (int p = 0; p < 4; ++p) o.v[p] = cvt<Out>(byte_at(j * 4 + p), scale, offset);
        store_quad(out + (q * C + j) * 4, o);
      }
    }
  }
}

// Any channel count / pixel tail: one element per lane.
template <typename Out>
__global__ __launch_bounds__(kThreads) void obs_stack_generic_kernel(
    const uint8_t* src, const int32_t* env_ids, Out* dst, int64_t pixels,
    int64_t channels, int layout, float scale, float offset) {
  const int64_t n = blockIdx.y;
  const int64_t e = env_ids ? env_ids[n] : n;
  const int64_t elems = pixels * channels;
  const uint8_t* frame = src + e * elems;
  Out* out = dst + n * elems;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < elems;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    int64_t s = i;
    if (layout == kLayoutChannelsFirst) {
      const int64_t c = i / pixels, p = i - c * pixels;
      s = p * channels + c;
    }
    out[i] = cvt<Out>(frame[s], scale, offset);
  }
}

template <typename Out>
hipError_t obs_stack_typed(const uint8_t* src, const int32_t* env_ids, void* dst,
                           int64_t n, int64_t pixels, int64_t channels, int layout,
                           float scale, float offset, hipStream_t stream) {
  Out* out = static_cast<Out*>(dst);
  const bool fast = pixels % 4 == 0 && channels >= 1 && channels <= 4 &&
                    reinterpret_cast<uint64_t>(src) % 4 == 0 &&
                    (pixels * channels) % 4 == 0 &&
                    reinterpret_cast<uint64_t>(dst) % (4 * sizeof(Out)) == 0;
  if (!fast) {
    const int64_t elems = pixels * channels;
    dim3 grid(static_cast<uint32_t>(std::min<int64_t>((elems + kThreads - 1) / kThreads, 64)),
              static_cast<uint32_t>(n));
    hipLaunchKernelGGL(obs_stack_generic_kernel<Out>, grid, dim3(kThreads), 0, stream,
                       src, env_ids, out, pixels, channels, layout, scale, offset);
    return hipGetLastError();
  }
  const int64_t quads = pixels / 4;
  dim3 grid(static_cast<uint32_t>(std::min<int64_t>((quads + kThreads - 1) / kThreads, 32)),
            static_cast<uint32_t>(n));
  const bool cf = layout == kLayoutChannelsFirst && channels > 1;
#define EMB_OBS(C)                                                                         \
  if (cf) hipLaunchKernelGGL((obs_stack_kernel<Out, C, true>), grid, dim3(kThreads), 0,   \
                             stream, src, env_ids, out, pixels, scale, offset);           \
  else hipLaunchKernelGGL((obs_stack_kernel<Out, C, false>), grid, dim3(kThreads), 0,     \
                          stream, src, env_ids, out, pixels, scale, offset);
  switch (channels) {
    case 1: EMB_OBS(1) break;
    case 2: EMB_OBS(2) break;
    case 3: EMB_OBS(3) break;
    default: EMB_OBS(4) break;
  }
#undef EMB_OBS
  return hipGetLastError();
}

// ------------------------------------------------- obs stack + early insert --
//
// The frames of a vectorised step are needed twice: by the policy (cast /
// transposed into its batch) and by the replay (copied into the pool rows the
// step will occupy).  Those rows are known before the policy runs — a worker's
// next row is its open chunk's cursor (replay_index.h peek) — so ONE launch
// reads every frame once and writes both: the policy batch exactly as
// obs_stack_kernel does, and the same 16 bytes per lane into the reserved pool
// row.  The other observation keys (reward, flags: a few bytes per env) and the
// step ids ride along in the workgroup that owns the frame's tail.  What is
// left for after the policy is the action (publish_one_kernel).
//
// Arguments: 64 bytes, all of them inside the kernel-argument preload (they
// arrive in SGPRs with the wave).  Everything per-env — the row table, the step
// ids, the narrow keys' descriptors — sits in a block in DEVICE memory that the
// host wrote through the BAR (abi.cpp ArgRing) or uploaded: with host-resident
// kernel arguments every wave's read of a by-value table would be a PCIe round
// trip of its own (1 800 waves: measured 11.5 us for this launch instead of 5).
// The row of a frame is read while the frame's loads are in flight (the
// policy-batch stores do not depend on it).
struct PreKey {
  const uint8_t* src;
  uint8_t* pool;
  int64_t rowbytes;
};
// The action of the PREVIOUS step, carried into this launch (see "carried
// publish" below): value * !flags[e] in `dtype` to row prev_rows[e] of `pool`.
struct PreCarry {
  const uint8_t* src;           // (n, rowbytes); null = nothing carried
  uint8_t* pool;
  const uint8_t* flags;         // the replay's is_last POOL (1-byte rows): the carried step's flag is at its own row
  int32_t rowbytes, dtype, elem, pad;
};
struct alignas(16) PreTable {
  uint8_t* stepid_pool;
  int32_t* rows_out;            // device int32[n]: the rows again, for the publish launch (may be null)
  PreKey narrow[kPreNarrow];
  PreCarry carry;
  uint32_t words[1];            // rows[n] | step ids, 5 words per row | the carried step's rows[n] (7 * n words)
};
struct PrewriteArgs {
  const uint8_t* frames;
  void* dst;
  uint8_t* frame_pool;
  int32_t pixels, frame_blocks;
  float scale, offset;
  int32_t n, n_narrow;
  const PreTable* table;        // device memory
};
static_assert(sizeof(PrewriteArgs) == 56, "obs_stack_insert_kernel's arguments (passed one by one) fit the 14 preloaded dwords");

// The narrow keys, the step id and the row for the publish launch of env n: one
// extra workgroup per env (the first n of the grid), so that this chain of
// dependent reads (table -> source bytes -> stores) runs beside the frame
// workgroups instead of behind one of them.  All loads are issued before the
// first store: three memory round trips, however many keys.
//
// Carried publish: when all that an insert has left after the policy is one
// small masked key (the action) and nobody needs the masked values back, the
// publish launch is not made at all -- the previous step's action rides in THIS
// launch (one element per lane of the env's narrow workgroup, the same typed
// multiply as publish_one_kernel), one dependent launch less per env step.
__device__ __forceinline__ void prewrite_carry(const PreTable& t, const uint32_t* tab, int32_t n_envs,
                                               int64_t n) {
  const PreCarry c = t.carry;                    // uniform: scalar loads
  if (!c.src) return;
  const int64_t prev = static_cast<int32_t>(gload<uint32_t>(tab + 6 * static_cast<int64_t>(n_envs) + n));
  const int64_t off = static_cast<int64_t>(threadIdx.x) * c.elem;
  if (prev < 0 || off >= c.rowbytes) return;
  // The carried step's is_last as the replay stored it (written by that step's
  // own early-insert launch, earlier on this stream) -- not the env's output
  // buffer, which an env with one output set has overwritten by now.
  const bool keep = gload<uint8_t>(c.flags + prev) == 0;
  const uint8_t* src = c.src + n * c.rowbytes + off;
  uint8_t* pool = c.pool + prev * c.rowbytes + off;
  switch (c.dtype) {
    case kU8: case kBool: put_masked<uint8_t>(src, pool, nullptr, keep); break;
    case kI8: put_masked<int8_t>(src, pool, nullptr, keep); break;
    case kI16: put_masked<int16_t>(src, pool, nullptr, keep); break;
    case kI32: put_masked<int32_t>(src, pool, nullptr, keep); break;
    case kI64: put_masked<int64_t>(src, pool, nullptr, keep); break;
    case kF16: put_masked<_Float16>(src, pool, nullptr, keep); break;
    case kBF16: put_masked_bf16(src, pool, nullptr, keep); break;
    case kF32: put_masked<float>(src, pool, nullptr, keep); break;
    default: put_masked<double>(src, pool, nullptr, keep); break;
  }
}

__device__ __forceinline__ void prewrite_narrow(const PrewriteArgs& a, int64_t n) {
  const PreTable& t = *a.table;
  const uint32_t* tab = t.words;
  prewrite_carry(t, tab, a.n, n);
  const int64_t row = static_cast<int32_t>(gload<uint32_t>(tab + n));
  if (row < 0) return;
  uint32_t sid = 0;
  if (threadIdx.x < kStepBytes / 4)
    sid = gload<uint32_t>(tab + a.n + n * (kStepBytes / 4) + threadIdx.x);
  PreKey key[kPreNarrow];
#pragma unroll
  for (int k = 0; k < kPreNarrow; ++k) key[k] = t.narrow[k];      // uniform: scalar loads
  uint8_t v[kPreNarrow];
#pragma unroll
  for (int k = 0; k < kPreNarrow; ++k) {
    v[k] = 0;
    if (k < a.n_narrow && static_cast<int64_t>(threadIdx.x) < key[k].rowbytes)
      v[k] = gload<uint8_t>(key[k].src + n * key[k].rowbytes + threadIdx.x);
  }
#pragma unroll
  for (int k = 0; k < kPreNarrow; ++k)
    if (k < a.n_narrow && static_cast<int64_t>(threadIdx.x) < key[k].rowbytes)
      gstore<uint8_t>(key[k].pool + row * key[k].rowbytes + threadIdx.x, v[k]);
  if (t.stepid_pool && threadIdx.x < kStepBytes / 4)
    gstore<uint32_t>(t.stepid_pool + row * kStepBytes + threadIdx.x * 4, sid);
  if (t.rows_out && threadIdx.x == 0) t.rows_out[n] = static_cast<int32_t>(row);
}
static_assert(kThreads >= 256, "a narrow key (<= 256 bytes per step) is one byte per lane");

// (Scalar parameters, not the struct: the kernel-argument preload only takes
// arguments passed as scalars / pointers -- a by-value struct is fetched with
// s_load by every wave, `.amdhsa_user_sgpr_kernarg_preload_length 0`.)
template <typename Out, int C, bool kChannelsFirst>
// (The preload covers 14 dwords = 56 bytes: exactly these.)
// ONE-dimensional grid: workgroups [0, n) are the narrow ones (their chain is the
// longest: first out), [n, n + n * frame_blocks) the frame blocks, env by env.
// The hardware hands consecutive workgroup ids to the 8 XCDs in turn: as a
// (frame_blocks + 1, n) grid with 7 + 1 blocks per env (84 x 84 x 4) every
// narrow workgroup landed on one XCD, which then took no frame block at all
// (tools/insert_lab.hip: 0.24 us of the launch).
__global__ __launch_bounds__(kThreads) void obs_stack_insert_kernel(
    const uint8_t* frames, const PreTable* table, uint8_t* frame_pool, void* dst,
    int32_t pixels_, int32_t frame_blocks, int32_t n_envs, int32_t n_narrow, float scale, float offset) {
  const PrewriteArgs a{frames, dst, frame_pool, pixels_, frame_blocks, scale, offset, n_envs, n_narrow, table};
  if (blockIdx.x < static_cast<uint32_t>(n_envs)) {
    prewrite_narrow(a, blockIdx.x);
    return;
  }
  const uint32_t id = blockIdx.x - static_cast<uint32_t>(n_envs);
  const int64_t n = id / static_cast<uint32_t>(frame_blocks);
  const uint32_t block = id - static_cast<uint32_t>(n) * static_cast<uint32_t>(frame_blocks);
  const int64_t pixels = a.pixels;
  const int64_t quads = pixels >> 2;
  const uint32_t* frame = reinterpret_cast<const uint32_t*>(a.frames + n * pixels * C);
  Out* out = static_cast<Out*>(a.dst) + n * pixels * C;
  const int64_t row = static_cast<int32_t>(gload<uint32_t>(a.table->words + n));
  uint32_t* pool = reinterpret_cast<uint32_t*>(a.frame_pool + row * pixels * C);
  const int64_t stride = static_cast<int64_t>(frame_blocks) * kThreads;
  for (int64_t q = static_cast<int64_t>(block) * kThreads + threadIdx.x; q < quads; q += stride) {
    uint32_t w[C];
    if constexpr (C == 4) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(frame) + q);
      w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) w[c] = frame[q * C + c];
    }
    auto byte_at = [&w](int idx) {
      return static_cast<uint8_t>((w[idx >> 2] >> ((idx & 3) * 8)) & 0xFFu);
    };
    if (kChannelsFirst) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        Quad<Out> o;
#pragma unroll
        for (int p = 0; p < 4; ++p) o.v[p] = cvt<Out>(byte_at(p * C + c), a.scale, a.offset);
        store_quad(out + c * pixels + q * 4, o);
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) {
        Quad<Out> o;
#pragma unroll
        for (int p = 0; p < 4; ++p) o.v[p] = cvt<Out>(byte_at(j * 4 + p), a.scale, a.offset);
        store_quad(out + (q * C + j) * 4, o);
      }
    }
    if (row >= 0) {
      if constexpr (C == 4) {
        __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4*>(pool) + q);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) __builtin_nontemporal_store(w[c], pool + q * C + c);
      }
    }
  }
}

// What is left of an insert after obs_stack_insert_kernel when the only other
// key is the action: rows[r] comes from the table that launch left in device
// memory, the value is written as src * !flags[r] (driver.py:72-74; a real
// multiply in the key's dtype) to its pool row and to `out`, the actions the
// next env step receives.  56 bytes of arguments, passed one by one: exactly
// the 14 dwords the kernel-argument preload covers.
struct PublishArgs {
  const uint8_t* src;
  uint8_t* pool;
  uint8_t* out;
  const int32_t* rows;
  const uint8_t* flags;      // null: plain copy
  int32_t n, rowbytes, dtype, elem;
};
static_assert(sizeof(PublishArgs) <= 64, "publish_one_kernel's arguments are preloaded");

__global__ __launch_bounds__(kThreads) void publish_one_kernel(
    const uint8_t* src_, uint8_t* pool_, uint8_t* out_, const int32_t* rows, const uint8_t* flags,
    int32_t n, int32_t rowbytes, int32_t dtype, int32_t elem) {
  const PublishArgs a{src_, pool_, out_, rows, flags, n, rowbytes, dtype, elem};
  const int64_t epr = a.rowbytes / a.elem;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= epr * a.n) return;
  const int64_t r = e / epr;
  const int64_t off = (e - r * epr) * a.elem;
  const int64_t row = a.rows[r];
  // dtype bit 8: `flags` is a pool of 1-byte rows, the flag of batch row r sits
  // at its pool row (a carried publish settled late, abi.cpp settle_carry).
  const bool by_row = (a.dtype & 0x100) != 0;
  const bool keep = !a.flags || (by_row ? row < 0 || gload<uint8_t>(a.flags + row) == 0
                                        : gload<uint8_t>(a.flags + r) == 0);
  const uint8_t* src = a.src + r * a.rowbytes + off;
  uint8_t* pool = row >= 0 ? a.pool + row * a.rowbytes + off : nullptr;
  uint8_t* out = a.out ? a.out + r * a.rowbytes + off : nullptr;
  switch (a.dtype & 0xFF) {
    case kU8: case kBool: put_masked<uint8_t>(src, pool, out, keep); break;
    case kI8: put_masked<int8_t>(src, pool, out, keep); break;
    case kI16: put_masked<int16_t>(src, pool, out, keep); break;
    case kI32: put_masked<int32_t>(src, pool, out, keep); break;
    case kI64: put_masked<int64_t>(src, pool, out, keep); break;
    case kF16: put_masked<_Float16>(src, pool, out, keep); break;
    case kBF16: put_masked_bf16(src, pool, out, keep); break;
    case kF32: put_masked<float>(src, pool, out, keep); break;
    default: put_masked<double>(src, pool, out, keep); break;
  }
}

template <typename Out>
hipError_t obs_stack_insert_typed(const PrewriteArgs& a, int64_t channels, int layout,
                                  hipStream_t stream, hipEvent_t stop) {
  // one workgroup per env for the narrow keys + frame_blocks per env for its frames
  dim3 grid(static_cast<uint32_t>(a.n) * static_cast<uint32_t>(a.frame_blocks + 1));
  const bool cf = layout == kLayoutChannelsFirst && channels > 1;
  // (hipExtLaunchKernelGGL only when a completion stamp is wanted: the plain
  // launch is the cheaper call.)
#define EMB_PRE_ARGS a.frames, a.table, a.frame_pool, a.dst, a.pixels, a.frame_blocks, a.n, a.n_narrow, a.scale, a.offset
#define EMB_PRE(C)                                                                                \
  if (cf && stop) hipExtLaunchKernelGGL((obs_stack_insert_kernel<Out, C, true>), grid,            \
                                        dim3(kThreads), 0, stream, nullptr, stop, 0, EMB_PRE_ARGS); \
  else if (cf) hipLaunchKernelGGL((obs_stack_insert_kernel<Out, C, true>), grid, dim3(kThreads),  \
                                  0, stream, EMB_PRE_ARGS);                                       \
  else if (stop) hipExtLaunchKernelGGL((obs_stack_insert_kernel<Out, C, false>), grid,            \
                                       dim3(kThreads), 0, stream, nullptr, stop, 0, EMB_PRE_ARGS); \
  else hipLaunchKernelGGL((obs_stack_insert_kernel<Out, C, false>), grid, dim3(kThreads), 0,      \
                          stream, EMB_PRE_ARGS);
  switch (channels) {
    case 1: EMB_PRE(1) break;
    case 2: EMB_PRE(2) break;
    case 3: EMB_PRE(3) break;
    default: EMB_PRE(4) break;
  }
#undef EMB_PRE
#undef EMB_PRE_ARGS
  return hipGetLastError();
}

// ------------------------------------------------------------- action mask --

// `flag` (optional): a word in pinned host memory that receives `seq` when every
// workgroup of the launch has stored its part -- with `out` in pinned memory too
// (the Driver bringing the next step's actions down to its env processes) the host
// sees the rows complete by reading one word, without an event.  `counter`: a
// zeroed device word of the caller's, left zeroed again.
__device__ __forceinline__ void notify_host(uint32_t* counter, uint32_t* flag, uint32_t seq) {
  if (!flag) return;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t seen = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (seen + 1 == gridDim.x) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void mask_rows_kernel(
    const T* act, T* out, int64_t n, int64_t row_elems, const uint8_t* is_last, uint32_t* counter,
    uint32_t* flag, uint32_t seq) {
  const int64_t total = n * row_elems;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t r = i / row_elems;
    // value * mask.astype(value.dtype): a real multiply, so -x -> -0.0 and
    // NaN stays NaN exactly as numpy does (driver.py:84-87).
    out[i] = act[i] * static_cast<T>(is_last[r] ? 0 : 1);
  }
  notify_host(counter, flag, seq);
}

// bf16 has no native multiply: widen to f32 (exact), multiply, narrow (the
// product is x, +-0 or NaN, all exactly representable).
template <>
__global__ __launch_bounds__(kThreads) void mask_rows_kernel<__hip_bfloat16>(
    const __hip_bfloat16* act, __hip_bfloat16* out, int64_t n, int64_t row_elems,
    const uint8_t* is_last, uint32_t* counter, uint32_t* flag, uint32_t seq) {
  const int64_t total = n * row_elems;
  const uint16_t* bits = reinterpret_cast<const uint16_t*>(act);
  uint16_t* obits = reinterpret_cast<uint16_t*>(out);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const float x = __uint_as_float(static_cast<uint32_t>(bits[i]) << 16);
    const float y = x * (is_last[i / row_elems] ? 0.f : 1.f);
    obits[i] = static_cast<uint16_t>(__float_as_uint(y) >> 16);
  }
  notify_host(counter, flag, seq);
}

// ------------------------------------------------------------ return scans --
//
// y_t = a_t + b_t * y_{t+1}.  A segment of W lanes owns one row; lane = time
// step.  Reverse inclusive Kogge-Stone over the affine maps
// (a1,b1) o (a2,b2) = (a1 + b1*a2, b1*b2) with __shfl_down inside the segment,
// rows longer than W are walked right-to-left in W-wide pieces with the
// running y carried in a register.  Episode boundaries need no flags: b_t = 0.

// Composite map of lanes [sl, W) of a segment: returns (A, B) with
// y_sl = A + B * y_{segment end + 1}.
template <int W>
__device__ __forceinline__ void affine_suffix(float& a, float& b, int sl) {
#pragma unroll
  for (int off = 1; off < W; off <<= 1) {
    const float ap = __shfl_down(a, off, W);
    const float bp = __shfl_down(b, off, W);
    if (sl + off < W) {
      a = fmaf(b, ap, a);
      b = b * bp;
    }
  }
}

// Four consecutive elements of a row at once.  Rows start wherever b*T puts
// them, so the vector types promise dword (floats) resp. byte (flags)
// alignment only; gfx950 serves such global loads in one instruction.
typedef float F4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint8_t B4 __attribute__((ext_vector_type(4), aligned(1)));
// (A branch-free form -- the short lane reads the four elements that END at
// its last one and shifts them down -- measured slower: 15.6 us against 13.4 at
// (65 536, 64), no gain at small sizes.)
__device__ __forceinline__ void load4(const float* p, int valid, float* out) {
  if (valid >= 4) {
    const F4 x = gload<F4>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = x[k];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = k < valid ? gload<float>(p + k) : 0.f;
  }
}
__device__ __forceinline__ void load4(const uint8_t* p, int valid, uint8_t* out) {
  if (valid >= 4) {
    const B4 x = gload<B4>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = x[k];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = k < valid ? gload<uint8_t>(p + k) : uint8_t{0};
  }
}
__device__ __forceinline__ void store4(float* p, int valid, const float* y) {
  if (valid >= 4) {
    F4 x;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = y[k];
    gstore<F4>(p, x);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < valid) gstore<float>(p + k, y[k]);
  }
}

// What differs between the scans: how (a_t, b_t) are formed from the inputs,
// the seed y_n, and what is stored.
// (Sizes: GaeOp<false> is exactly 64 bytes and LambdaOp 56, so that a scan's
// whole argument block arrives through the kernel-argument preload: with
// host-resident arguments anything beyond 64 bytes is a PCIe read in front of
// the first instruction of every wave.  B and T ride inside the op for that.)
struct NoGroups {};
struct Groups { int64_t group, gs_rew, gs_flag; };
template <bool kGrouped>
struct GaeOp : std::conditional_t<kGrouped, Groups, NoGroups> {   // ppo/agent.py:188-201
  const float* rew; const float* val; const uint8_t* last; const uint8_t* term;
  float* adv; float* tar;
  int32_t T, B; float live_scale, lam;
  // kGrouped: rew / last / term come straight out of a grouped packed batch
  // (distributed.py): row b then starts (b / group) * gs + (b % group) * T
  // elements into its key (gs_rew in floats, gs_flag in bytes).
  // `val` (the critic's output) and the results are always dense.
  // The op travels to the kernel as SCALAR arguments (unpack -> make): the
  // kernel-argument preload takes scalars and pointers, not by-value structs.
  template <typename F>
  void unpack(F&& f) const {
    if constexpr (kGrouped) f(rew, val, last, term, adv, tar, T, B, live_scale, lam, this->group, this->gs_rew, this->gs_flag);
    else f(rew, val, last, term, adv, tar, T, B, live_scale, lam);
  }
  template <typename... G>
  __host__ __device__ static GaeOp make(const float* rew, const float* val, const uint8_t* last,
                                        const uint8_t* term, float* adv, float* tar, int32_t T, int32_t B,
                                        float live_scale, float lam, G... groups) {
    GaeOp op;
    op.rew = rew; op.val = val; op.last = last; op.term = term; op.adv = adv; op.tar = tar;
    op.T = T; op.B = B; op.live_scale = live_scale; op.lam = lam;
    if constexpr (kGrouped) {
      const int64_t g[3] = {groups...};
      op.group = g[0]; op.gs_rew = g[1]; op.gs_flag = g[2];
    }
    return op;
  }
  __device__ float seed(int64_t) const { return 0.f; }
  __device__ void where(int64_t b, int64_t t, int64_t& ir, int64_t& il) const {
    ir = il = b * T + t;
    if constexpr (kGrouped) {
      const int64_t g = b / this->group, j = b - g * this->group;
      ir = g * this->gs_rew + j * T + t;
      il = g * this->gs_flag + j * T + t;
    }
  }
  __device__ void coef(int64_t b, int64_t t, float& a, float& bc, float& keep) const {
    const int64_t i = b * T + t;
    int64_t ir, il;
    where(b, t, ir, il);
    const bool tm = term[il + 1] != 0;
    const float live = tm ? 0.f : live_scale;
    const float cont = (tm || last[il + 1] != 0) ? 0.f : lam;
    keep = val[i];
    a = rew[ir + 1] + live * val[i + 1] - keep;
    bc = live * cont;
  }
  __device__ void store(int64_t b, int64_t t, float y, float keep) const {
    adv[b * (T - 1) + t] = y;
    tar[b * (T - 1) + t] = y + keep;
  }
  // Elements t0 .. t0+3 of row b (`valid` of them exist): same arithmetic as
  // coef/store, the row's index math done once.
  __device__ void coef4(int64_t b, int t0, int valid, float* a, float* bc, float* keep) const {
    const int64_t i = b * T + t0;
    int64_t ir, il;
    where(b, t0, ir, il);
    // val[t0 .. t0+valid]: one more than the elements, the last one's successor
    // (it exists: t0 + valid <= T - 1).
    float v[5], r[4];
    uint8_t tm[4], ls[4];
    float after;
    if (valid >= 4) {
      // the usual lane: all five loads issued back to back, one wait
      const F4 v4 = gload<F4>(val + i);
      after = gload<float>(val + i + 4);
      const F4 r4 = gload<F4>(rew + ir + 1);
      const B4 t4 = gload<B4>(term + il + 1);
      const B4 l4 = gload<B4>(last + il + 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = v4[k];
        r[k] = r4[k];
        tm[k] = t4[k];
        ls[k] = l4[k];
      }
    } else {
      load4(val + i, valid, v);
      after = gload<float>(val + i + valid);
      load4(rew + ir + 1, valid, r);
      load4(term + il + 1, valid, tm);
      load4(last + il + 1, valid, ls);
    }
    v[4] = after;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float next = k + 1 == valid ? after : v[k + 1];
      const bool t_ = tm[k] != 0;
      const float live = t_ ? 0.f : live_scale;
      const float cont = (t_ || ls[k] != 0) ? 0.f : lam;
      keep[k] = v[k];
      a[k] = r[k] + live * next - v[k];
      bc[k] = live * cont;
    }
  }
  __device__ void store4(int64_t b, int t0, int valid, const float* y, const float* keep) const {
    float z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = y[k] + keep[k];
    emb::store4(adv + b * (T - 1) + t0, valid, y);
    emb::store4(tar + b * (T - 1) + t0, valid, z);
  }
};
static_assert(sizeof(GaeOp<false>) == 64, "the dense GAE op is covered by the kernel-argument preload");

struct LambdaOp {   // dreamerv3/agent.py:482-490
  const uint8_t* last; const uint8_t* term; const float* rew; const float* boot;
  float* ret; int32_t T, B; float disc, lam;
  template <typename F>
  void unpack(F&& f) const { f(last, term, rew, boot, ret, T, B, disc, lam); }
  __host__ __device__ static LambdaOp make(const uint8_t* last, const uint8_t* term, const float* rew,
                                           const float* boot, float* ret, int32_t T, int32_t B,
                                           float disc, float lam) {
    return LambdaOp{last, term, rew, boot, ret, T, B, disc, lam};
  }
  __device__ float seed(int64_t b) const { return boot[b * T + T - 1]; }
  __device__ void coef(int64_t b, int64_t t, float& a, float& bc, float& keep) const {
    const int64_t i = b * T + t;
    const float live = (1.f - static_cast<float>(term[i + 1] != 0)) * disc;
    const float cont = (1.f - static_cast<float>(last[i + 1] != 0)) * lam;
    keep = 0.f;
    a = rew[i + 1] + (1.f - cont) * live * boot[i + 1];
    bc = live * cont;
  }
  __device__ void store(int64_t b, int64_t t, float y, float) const { ret[b * (T - 1) + t] = y; }
  __device__ void coef4(int64_t b, int t0, int valid, float* a, float* bc, float* keep) const {
    const int64_t i = b * T + t0 + 1;
    float r[4], bt[4];
    uint8_t tm[4], ls[4];
    if (valid >= 4) {
      // the usual lane: all four loads issued back to back, one wait (as GaeOp)
      const F4 r4 = gload<F4>(rew + i);
      const F4 b4 = gload<F4>(boot + i);
      const B4 t4 = gload<B4>(term + i);
      const B4 l4 = gload<B4>(last + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r[k] = r4[k];
        bt[k] = b4[k];
        tm[k] = t4[k];
        ls[k] = l4[k];
      }
    } else {
      load4(rew + i, valid, r);
      load4(boot + i, valid, bt);
      load4(term + i, valid, tm);
      load4(last + i, valid, ls);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float live = (1.f - static_cast<float>(tm[k] != 0)) * disc;
      const float cont = (1.f - static_cast<float>(ls[k] != 0)) * lam;
      keep[k] = 0.f;
      a[k] = r[k] + (1.f - cont) * live * bt[k];
      bc[k] = live * cont;
    }
  }
  __device__ void store4(int64_t b, int t0, int valid, const float* y, const float*) const {
    emb::store4(ret + b * (T - 1) + t0, valid, y);
  }
};
static_assert(sizeof(LambdaOp) <= 64, "covered by the kernel-argument preload");

// Short rows: a W-lane segment per row, rows longer than W walked right to
// left with the running value in a register.

template <int W, typename Op, typename... Args>
__global__ __launch_bounds__(kThreads) void scan_rows_kernel(Args... args) {
  const Op op = Op::make(args...);
  const int64_t B = op.B, n = op.T - 1;
  const int sl = threadIdx.x % W;
  const int64_t b = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / W;
  const bool row_ok = b < B;
  float carry = row_ok ? op.seed(b) : 0.f;
  for (int64_t base = ((n - 1) / W) * W; base >= 0; base -= W) {
    const int64_t t = base + sl;
    const bool ok = row_ok && t < n;
    float a = 0.f, bc = 1.f, keep = 0.f;     // (0, 1) = identity map
    if (ok) op.coef(b, t, a, bc, keep);
    affine_suffix<W>(a, bc, sl);
    const float y = fmaf(bc, carry, a);
    if (ok) op.store(b, t, y, keep);
    carry = __shfl(y, 0, W);
  }
}
//
// FOUR elements per lane: at large B the one-element-per-lane form above is
// bound by its instruction count, not by HBM ((65 536, 64): ~220 VALU
// instructions per wave and row, most of them 64-bit index arithmetic and
// shuffle addressing, 19.3 us = 50 % of peak): here the row's index math is done
// once per four elements, the loads are 16-byte / 4-byte vectors, the four
// elements of a lane are folded sequentially (3 fma pairs) and the Kogge-Stone
// runs over W = rowlen/4 lanes (4 rounds for T = 64 instead of 6).
template <int W, typename Op>
__device__ __forceinline__ void scan_rows4_body(const Op& op, uint32_t block) {
  const int64_t B = op.B;
  const int n = op.T - 1;
  const int sl = threadIdx.x % W;
  const int64_t b = (static_cast<int64_t>(block) * kThreads + threadIdx.x) / W;
  const int t0 = 4 * sl;
  const bool row_ok = b < B;
  const int valid = row_ok ? (n - t0 >= 4 ? 4 : (n - t0 > 0 ? n - t0 : 0)) : 0;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, bc[4] = {1.f, 1.f, 1.f, 1.f}, keep[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid > 0) op.coef4(b, t0, valid, a, bc, keep);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k >= valid) {            // (0, 1) = identity map right of the row's end
      a[k] = 0.f;
      bc[k] = 1.f;
    }
  // this lane's four elements as one map, then the maps of the lanes to the right
  float A = a[3], Bm = bc[3];
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    A = fmaf(bc[k], A, a[k]);
    Bm = bc[k] * Bm;
  }
  affine_suffix<W>(A, Bm, sl);
  const float seed = row_ok ? op.seed(b) : 0.f;
  const float first = fmaf(Bm, seed, A);               // y at t0
  float carry = __shfl_down(first, 1, W);              // y at t0 + 4 = the next lane's first
  if (sl == W - 1) carry = seed;
  float y[4];
  y[3] = fmaf(bc[3], carry, a[3]);
#pragma unroll
  for (int k = 2; k >= 0; --k) y[k] = fmaf(bc[k], y[k + 1], a[k]);
  if (valid > 0) op.store4(b, t0, valid, y, keep);
}
template <int W, typename Op, typename... Args>
__global__ __launch_bounds__(kThreads) void scan_rows4_kernel(Args... args) {
  scan_rows4_body<W>(Op::make(args...), blockIdx.x);
}

// Several lambda-return problems of one train step in ONE launch (DreamerV3
// computes the replay returns (B, T) and the imagined returns (B*K, H+1) in the
// same step, dreamerv3/agent.py:401-405,464-466): at these sizes each scan is
// pure launch latency, so two launches cost twice what one does.  Workgroups
// [first[i], first[i+1]) belong to problem i; every problem runs the
// four-steps-per-lane form with its own segment width.
constexpr int kScanMulti = 4;
struct LambdaMulti {
  LambdaOp op[kScanMulti];
  int32_t first[kScanMulti + 1];
  int32_t width[kScanMulti];
};
__global__ __launch_bounds__(kThreads) void lambda_multi_kernel(const LambdaMulti m) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < kScanMulti; ++k)
    if (blockIdx.x >= static_cast<uint32_t>(m.first[k])) i = k;
  const uint32_t block = blockIdx.x - static_cast<uint32_t>(m.first[i]);
  // (a copy selected with a uniform index: the by-value argument stays in SGPRs)
  LambdaOp op = m.op[0];
#pragma unroll
  for (int k = 1; k < kScanMulti; ++k)
    if (i == k) op = m.op[k];
  int width = m.width[0];
#pragma unroll
  for (int k = 1; k < kScanMulti; ++k)
    if (i == k) width = m.width[k];
  switch (width) {
    case 4: scan_rows4_body<4>(op, block); break;
    case 8: scan_rows4_body<8>(op, block); break;
    case 16: scan_rows4_body<16>(op, block); break;
    case 32: scan_rows4_body<32>(op, block); break;
    default: scan_rows4_body<64>(op, block); break;
  }
}

// Long rows: one workgroup of `waves` wavefronts per row.  Each wave reduces its
// 64 steps to one affine map, the per-wave maps are staged in LDS, every wave
// folds the maps to its right into its carry, and the workgroup walks the row
// right to left in pieces of 64 * waves steps with the carry handed on through
// LDS.
template <typename Op, typename... Args>
__global__ __launch_bounds__(1024) void scan_long_rows_kernel(Args... args) {
  const Op op = Op::make(args...);
  const int64_t n = op.T - 1;
  __shared__ float s_a[16], s_b[16], s_carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t span = 64 * waves;
  if (threadIdx.x == 0) s_carry = op.seed(b);
  for (int64_t base = ((n - 1) / span) * span; base >= 0; base -= span) {
    const int64_t t = base + threadIdx.x;
    const bool ok = t < n;
    float a = 0.f, bc = 1.f, keep = 0.f;
    if (ok) op.coef(b, t, a, bc, keep);
    affine_suffix<64>(a, bc, lane);
    if (lane == 0) {
      s_a[wave] = a;
      s_b[wave] = bc;
    }
    __syncthreads();
    float carry = s_carry;
    for (int w = waves - 1; w > wave; --w) carry = fmaf(s_b[w], carry, s_a[w]);
    const float y = fmaf(bc, carry, a);
    if (ok) op.store(b, t, y, keep);
    __syncthreads();
    if (threadIdx.x == 0) s_carry = y;
    // the next iteration's first __syncthreads orders this write before its reads
  }
}

template <typename Op>
hipError_t launch_scan(const Op& op, hipStream_t stream) {
  const int64_t B = op.B, n = op.T - 1;
  if (n > 256) {
    const int waves = static_cast<int>(std::min<int64_t>(16, (n + 63) / 64));
    op.unpack([&](auto... a) {
      hipLaunchKernelGGL((scan_long_rows_kernel<Op, decltype(a)...>), dim3(static_cast<uint32_t>(B)),
                         dim3(64 * waves), 0, stream, a...);
    });
    return hipGetLastError();
  }
  // Short rows in small batches (Dreamer's imagined returns, (1024, 16)) stay
  // with one element per lane: 64 workgroups instead of 16, 3.4 us against 3.7.
  if (n <= 16 && B <= 8192) {
    const int64_t rows_per_block = kThreads / 16;
    const dim3 grid(static_cast<uint32_t>((B + rows_per_block - 1) / rows_per_block));
    op.unpack([&](auto... a) {
      hipLaunchKernelGGL((scan_rows_kernel<16, Op, decltype(a)...>), grid, dim3(kThreads), 0, stream, a...);
    });
    return hipGetLastError();
  }
  const int W = n <= 16 ? 4 : n <= 32 ? 8 : n <= 64 ? 16 : n <= 128 ? 32 : 64;
  const int64_t rows_per_block = kThreads / W;
  const dim3 grid(static_cast<uint32_t>((B + rows_per_block - 1) / rows_per_block));
  op.unpack([&](auto... a) {
#define EMB_SCAN4(W_) \
  hipLaunchKernelGGL((scan_rows4_kernel<W_, Op, decltype(a)...>), grid, dim3(kThreads), 0, stream, a...)
    switch (W) {
      case 4: EMB_SCAN4(4); break;
      case 8: EMB_SCAN4(8); break;
      case 16: EMB_SCAN4(16); break;
      case 32: EMB_SCAN4(32); break;
      default: EMB_SCAN4(64); break;
    }
#undef EMB_SCAN4
  });
  return hipGetLastError();
}

// Time-major: lane = batch column (coalesced), the T-step recurrence runs
// sequentially in the reference's own order.
__global__ __launch_bounds__(kThreads) void director_score_kernel(
    const float* __restrict__ rew, const float* __restrict__ cont,
    const float* __restrict__ value, int64_t T, int64_t B, float discount, float lam,
    float* __restrict__ ret) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (b >= B) return;
  float v = value[(T - 1) * B + b];
  for (int64_t t = T - 2; t >= 0; --t) {
    const float d = cont[(t + 1) * B + b] * discount;
    const float interm = rew[t * B + b] + d * value[(t + 1) * B + b] * (1.f - lam);
    v = interm + d * lam * v;
    ret[t * B + b] = v;
  }
}

// Director manager steps (director/hierarchy.py:240-256), time-major: for every
// window j of k steps and column b:  w_i = prod_{i'<=i} cont[jk+i'],
// reward_out[j-1] = mean_i(shifted_reward[jk+i] * w_i)  (j >= 1; the reward is
// shifted by one step: shifted[0] = 0, shifted[t] = reward[t-1]),
// cont_out[j] = prod_i cont[jk+i].  Lane = column (coalesced), k is small.
__global__ __launch_bounds__(kThreads) void abstract_traj_kernel(
    const float* __restrict__ reward, const float* __restrict__ cont, int64_t T, int64_t B,
    int k, float* __restrict__ reward_out, float* __restrict__ cont_out) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (b >= B) return;
  float w = 1.f, acc = 0.f;
  for (int i = 0; i < k; ++i) {
    const int64_t t = j * k + i;
    w *= cont[t * B + b];
    const float r = (t == 0 || !reward) ? 0.f : reward[(t - 1) * B + b];
    acc += r * w;
  }
  if (cont_out) cont_out[j * B + b] = w;
  if (reward_out && j >= 1) reward_out[(j - 1) * B + b] = acc / static_cast<float>(k);
}

// ------------------------------------------------------------ synthetic env --

// Device-resident stand-in for N simulators: the episode logic
// of envs/dummy.py:38-48 with counter-hash frames, so gathers are verifiable.
//
// 56 bytes of arguments, passed as scalars so that the kernel-argument preload
// takes them (14 dwords; a by-value struct is not preloaded at all -- with
// host-resident arguments every wave would start with a PCIe read): the three
// flag outputs travel as 32-bit offsets from the reward pointer (the launcher
// falls back to the five-pointer form when they do not fit), the env count is
// the grid's y size, the generation bit rides in the sign bit of the episode
// length.  The per-env state is read from one half of `counters` and written to
// the other (`turn`), so that every workgroup of an env may read it while the
// last one writes: a frame is cut over gridDim.x workgroups instead of one.
struct SynthArgs {
  int32_t* counters;          // 2 x int32[2n]: {count, done} per env, two generations
  const uint8_t* reset;
  uint8_t* image;
  float* reward;
  int32_t off_first, off_last, off_terminal;   // bytes from `reward`
  int32_t frame_bytes, env0, episode_len, n_turn;   // n << 1 | turn
};
static_assert(sizeof(SynthArgs) <= 64, "synth_env_kernel's arguments (passed one by one, n from the grid) are preloaded");

__global__ __launch_bounds__(kThreads) void synth_env_kernel(
    int32_t* counters, const uint8_t* reset, uint8_t* image, float* reward, int32_t off_first,
    int32_t off_last, int32_t off_terminal, int32_t frame_bytes, int32_t env0, int32_t len_turn) {
  // 56 bytes = the 14 dwords the preload covers: the env count is the grid's y
  // size, the generation bit rides in the sign bit of the episode length.
  const SynthArgs a{counters, reset, image, reward, off_first, off_last, off_terminal,
                    frame_bytes, env0, len_turn & 0x7FFFFFFF,
                    static_cast<int32_t>(gridDim.y << 1 | (static_cast<uint32_t>(len_turn) >> 31))};
  const int64_t e = blockIdx.y;
  const int32_t n = a.n_turn >> 1, turn = a.n_turn & 1;
  const int32_t* __restrict__ in = a.counters + turn * 2 * n;
  int32_t count = in[2 * e];
  const bool was_done = in[2 * e + 1] != 0;
  const bool restart = (a.reset && a.reset[e]) || was_done;
  const int64_t length = a.episode_len + ((a.env0 + e) % 8) * 13;
  count = restart ? 0 : count + 1;
  const bool done = !restart && count >= length;
  const uint32_t salt = static_cast<uint32_t>((a.env0 + e) * 131 + static_cast<int64_t>(count) * 7);
  // byte i of the frame = (salt + i) & 0xFF, written 16 bytes per lane.
  u32x4* out = reinterpret_cast<u32x4*>(a.image + e * a.frame_bytes);
  const int64_t vecs = a.frame_bytes >> 4;
  auto word = [salt](int64_t byte0) {
    const uint32_t x = salt + static_cast<uint32_t>(byte0);
    return (x & 0xFF) | (((x + 1) & 0xFF) << 8) | (((x + 2) & 0xFF) << 16) | (((x + 3) & 0xFF) << 24);
  };
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < vecs;
       i += static_cast<int64_t>(gridDim.x) * kThreads)
    __builtin_nontemporal_store(
        u32x4{word(i * 16), word(i * 16 + 4), word(i * 16 + 8), word(i * 16 + 12)}, out + i);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int32_t* next = a.counters + (1 - turn) * 2 * n;
    next[2 * e] = count;
    next[2 * e + 1] = done ? 1 : 0;
    uint8_t* flags = reinterpret_cast<uint8_t*>(a.reward);
    a.reward[e] = restart ? 0.f : static_cast<float>(count % 7);
    flags[a.off_first + e] = restart ? 1 : 0;
    flags[a.off_last + e] = done ? 1 : 0;
    flags[a.off_terminal + e] = done ? 1 : 0;
  }
}

// The same step when the flag buffers are too far from `reward` for 32-bit
// offsets (80 bytes of arguments).
__global__ __launch_bounds__(kThreads) void synth_env_far_kernel(
    const SynthArgs a, uint8_t* is_first, uint8_t* is_last, uint8_t* is_terminal) {
  const int64_t e = blockIdx.y;
  const int32_t n = a.n_turn >> 1, turn = a.n_turn & 1;
  const int32_t* __restrict__ in = a.counters + turn * 2 * n;
  int32_t count = in[2 * e];
  const bool restart = (a.reset && a.reset[e]) || in[2 * e + 1] != 0;
  const int64_t length = a.episode_len + ((a.env0 + e) % 8) * 13;
  count = restart ? 0 : count + 1;
  const bool done = !restart && count >= length;
  const uint32_t salt = static_cast<uint32_t>((a.env0 + e) * 131 + static_cast<int64_t>(count) * 7);
  u32x4* out = reinterpret_cast<u32x4*>(a.image + e * a.frame_bytes);
  const int64_t vecs = a.frame_bytes >> 4;
  auto word = [salt](int64_t byte0) {
    const uint32_t x = salt + static_cast<uint32_t>(byte0);
    return (x & 0xFF) | (((x + 1) & 0xFF) << 8) | (((x + 2) & 0xFF) << 16) | (((x + 3) & 0xFF) << 24);
  };
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < vecs;
       i += static_cast<int64_t>(gridDim.x) * kThreads)
    __builtin_nontemporal_store(
        u32x4{word(i * 16), word(i * 16 + 4), word(i * 16 + 8), word(i * 16 + 12)}, out + i);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int32_t* next = a.counters + (1 - turn) * 2 * n;
    next[2 * e] = count;
    next[2 * e + 1] = done ? 1 : 0;
    a.reward[e] = restart ? 0.f : static_cast<float>(count % 7);
    is_first[e] = restart ? 1 : 0;
    is_last[e] = done ? 1 : 0;
    is_terminal[e] = done ? 1 : 0;
  }
}

}  // namespace

hipError_t launch_gather(const MovePlan& plan, hipStream_t stream, hipEvent_t start,
                         hipEvent_t stop) {
  MoveLaunch launch;
  const hipError_t e = prepare_move(plan, &launch, true);
  return e != hipSuccess ? e : launch_move(launch, true, nullptr, stream, start, stop);
}

bool plan_fits_inline(const MovePlan& plan) { return inline_words_needed(plan) > 0; }

hipError_t launch_scatter(const MovePlan& plan, hipStream_t stream) {
  MoveLaunch launch;
  const hipError_t e = prepare_move(plan, &launch, false);
  return e != hipSuccess ? e : launch_move(launch, false, nullptr, stream, nullptr, nullptr);
}

hipError_t launch_window(const uint8_t* src, uint8_t* dst, int64_t batch, int64_t total,
                         int64_t start, int64_t count, int64_t rowbytes,
                         hipStream_t stream) {
  if (batch <= 0 || count <= 0) return hipSuccess;
  const uint64_t mix = reinterpret_cast<uint64_t>(src) | reinterpret_cast<uint64_t>(dst) |
                       static_cast<uint64_t>(rowbytes);
  const int unit = mix % 16 == 0 ? 16 : mix % 8 == 0 ? 8 : mix % 4 == 0 ? 4 : mix % 2 == 0 ? 2 : 1;
  const int64_t units = count * rowbytes / unit;
  const int64_t bx = std::min<int64_t>((units + kThreads - 1) / kThreads, 1024);
  hipLaunchKernelGGL(window_kernel, dim3(static_cast<uint32_t>(bx), static_cast<uint32_t>(batch)),
                     dim3(kThreads), 0, stream, src, dst, total, start, count, rowbytes, unit, units);
  return hipGetLastError();
}

hipError_t launch_obs_stack(const uint8_t* src, const int32_t* env_ids, void* dst, int64_t n,
                            int64_t pixels, int64_t channels, int layout, int out_dtype,
                            float scale, float offset, hipStream_t stream) {
  if (n <= 0 || pixels <= 0 || channels <= 0) return hipSuccess;
  switch (out_dtype) {
    case kU8: return obs_stack_typed<uint8_t>(src, env_ids, dst, n, pixels, channels, layout, scale, offset, stream);
    case kF16: return obs_stack_typed<__half>(src, env_ids, dst, n, pixels, channels, layout, scale, offset, stream);
    case kBF16: return obs_stack_typed<__hip_bfloat16>(src, env_ids, dst, n, pixels, channels, layout, scale, offset, stream);
    case kF32: return obs_stack_typed<float>(src, env_ids, dst, n, pixels, channels, layout, scale, offset, stream);
    default: return hipErrorInvalidValue;
  }
}

namespace {
// Bytes of any alignment from `src` to `dst`: 16-byte units when both allow it, a
// byte tail.  `src` may be pinned host memory the GPU reads across PCIe (a piece
// of the Driver's shared observation slab): four units per lane in flight.
__global__ __launch_bounds__(kThreads) void copy_bytes_kernel(const uint8_t* __restrict__ src,
                                                             uint8_t* __restrict__ dst, int64_t bytes) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t vecs = aligned ? bytes >> 4 : 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  for (; i + 3 * stride < vecs; i += 4 * stride) {
    const u32x4 a = reinterpret_cast<const u32x4*>(src)[i], b = reinterpret_cast<const u32x4*>(src)[i + stride];
    const u32x4 c = reinterpret_cast<const u32x4*>(src)[i + 2 * stride], d = reinterpret_cast<const u32x4*>(src)[i + 3 * stride];
    reinterpret_cast<u32x4*>(dst)[i] = a;
    reinterpret_cast<u32x4*>(dst)[i + stride] = b;
    reinterpret_cast<u32x4*>(dst)[i + 2 * stride] = c;
    reinterpret_cast<u32x4*>(dst)[i + 3 * stride] = d;
  }
  for (; i < vecs; i += stride) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
  for (int64_t j = (vecs << 4) + static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < bytes; j += stride)
    dst[j] = src[j];
}
}  // namespace

hipError_t launch_copy_bytes(const void* src, void* dst, int64_t bytes, hipStream_t stream) {
  if (bytes <= 0) return hipSuccess;
  // one unit per lane up to 256 workgroups, then several per lane
  const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(256, ((bytes >> 4) + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), 0, stream,
                     static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), bytes);
  return hipGetLastError();
}

hipError_t launch_mask_rows(const void* act, void* out, int64_t n, int64_t row_elems, int dtype,
                            const uint8_t* is_last, hipStream_t stream, uint32_t* counter, uint32_t* flag,
                            uint32_t seq) {
  const int64_t total = n * row_elems;
  if (total <= 0) return flag ? hipErrorInvalidValue : hipSuccess;
  const dim3 grid(static_cast<uint32_t>(std::min<int64_t>((total + kThreads - 1) / kThreads, 2048)));
#define EMB_MASK(T) hipLaunchKernelGGL(mask_rows_kernel<T>, grid, dim3(kThreads), 0, stream, static_cast<const T*>(act), static_cast<T*>(out), n, row_elems, is_last, counter, flag, seq)
  switch (dtype) {
    case kU8: case kBool: EMB_MASK(uint8_t); break;
    case kI8: EMB_MASK(int8_t); break;
    case kI16: EMB_MASK(int16_t); break;
    case kI32: EMB_MASK(int32_t); break;
    case kI64: EMB_MASK(int64_t); break;
    case kF16: EMB_MASK(_Float16); break;
    case kBF16: EMB_MASK(__hip_bfloat16); break;
    case kF32: EMB_MASK(float); break;
    case kF64: EMB_MASK(double); break;
    default: return hipErrorInvalidValue;
  }
#undef EMB_MASK
  return hipGetLastError();
}

hipError_t launch_gae(const float* rew, const float* val, const uint8_t* last,
                      const uint8_t* term, int64_t B, int64_t T, float live_scale, float lam,
                      float* adv, float* tar, hipStream_t stream, int64_t group,
                      int64_t group_stride_bytes) {
  if (B <= 0 || T < 2) return hipSuccess;
  if (group < 0 || (group && group_stride_bytes % 4 != 0) || B > INT32_MAX || T > INT32_MAX)
    return hipErrorInvalidValue;
  if (group) {
    GaeOp<true> op;
    op.group = group;
    op.gs_rew = group_stride_bytes / 4;
    op.gs_flag = group_stride_bytes;
    op.rew = rew; op.val = val; op.last = last; op.term = term; op.adv = adv; op.tar = tar;
    op.T = static_cast<int32_t>(T); op.B = static_cast<int32_t>(B);
    op.live_scale = live_scale; op.lam = lam;
    return launch_scan(op, stream);
  }
  GaeOp<false> op;
  op.rew = rew; op.val = val; op.last = last; op.term = term; op.adv = adv; op.tar = tar;
  op.T = static_cast<int32_t>(T); op.B = static_cast<int32_t>(B);
  op.live_scale = live_scale; op.lam = lam;
  return launch_scan(op, stream);
}

hipError_t launch_lambda_return(const uint8_t* last, const uint8_t* term, const float* rew,
                                const float* boot, int64_t B, int64_t T, float disc, float lam,
                                float* ret, hipStream_t stream) {
  if (B <= 0 || T < 2) return hipSuccess;
  if (B > INT32_MAX || T > INT32_MAX) return hipErrorInvalidValue;
  return launch_scan(LambdaOp{last, term, rew, boot, ret, static_cast<int32_t>(T),
                              static_cast<int32_t>(B), disc, lam}, stream);
}

hipError_t launch_lambda_return_multi(int n_problems, const LambdaProblem* problems, hipStream_t stream) {
  if (n_problems < 1) return hipSuccess;
  bool together = n_problems <= kScanMulti;
  for (int i = 0; i < n_problems; ++i) {
    const LambdaProblem& q = problems[i];
    if (q.B > INT32_MAX || q.T > INT32_MAX) return hipErrorInvalidValue;
    together = together && q.T - 1 <= 256;        // long rows have a kernel of their own
  }
  if (!together || n_problems == 1) {
    for (int i = 0; i < n_problems; ++i) {
      const LambdaProblem& q = problems[i];
      const hipError_t err = launch_lambda_return(q.last, q.term, q.rew, q.boot, q.B, q.T, q.disc, q.lam,
                                                  q.ret, stream);
      if (err != hipSuccess) return err;
    }
    return hipSuccess;
  }
  LambdaMulti m;
  std::memset(&m, 0, sizeof(m));
  int64_t blocks = 0;
  int used = 0;
  for (int i = 0; i < n_problems; ++i) {
    const LambdaProblem& q = problems[i];
    if (q.B <= 0 || q.T < 2) continue;
    const int64_t n = q.T - 1;
    const int W = n <= 16 ? 4 : n <= 32 ? 8 : n <= 64 ? 16 : n <= 128 ? 32 : 64;
    m.op[used] = LambdaOp{q.last, q.term, q.rew, q.boot, q.ret, static_cast<int32_t>(q.T),
                          static_cast<int32_t>(q.B), q.disc, q.lam};
    m.width[used] = W;
    m.first[used] = static_cast<int32_t>(blocks);
    const int64_t rows_per_block = kThreads / W;
    blocks += (q.B + rows_per_block - 1) / rows_per_block;
    if (blocks > INT32_MAX) return hipErrorInvalidValue;
    ++used;
  }
  if (used == 0) return hipSuccess;
  for (int k = used; k <= kScanMulti; ++k) m.first[k] = static_cast<int32_t>(blocks);
  hipLaunchKernelGGL(lambda_multi_kernel, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), 0, stream, m);
  return hipGetLastError();
}

hipError_t launch_director_score(const float* rew, const float* cont, const float* value,
                                 int64_t T, int64_t B, float discount, float lam, float* ret,
                                 hipStream_t stream) {
  if (B <= 0 || T < 2) return hipSuccess;
  hipLaunchKernelGGL(director_score_kernel, dim3(static_cast<uint32_t>((B + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, stream, rew, cont, value, T, B, discount, lam, ret);
  return hipGetLastError();
}

hipError_t launch_abstract_traj(const float* reward, const float* cont, int64_t T, int64_t B,
                                int k, float* reward_out, float* cont_out, hipStream_t stream) {
  if (B <= 0 || T <= 0) return hipSuccess;
  if (k < 1 || T % k != 0) return hipErrorInvalidValue;
  const dim3 grid(static_cast<uint32_t>((B + kThreads - 1) / kThreads), static_cast<uint32_t>(T / k));
  hipLaunchKernelGGL(abstract_traj_kernel, grid, dim3(kThreads), 0, stream, reward, cont, T, B, k,
                     reward_out, cont_out);
  return hipGetLastError();
}

hipError_t launch_synth_env(uint8_t* image, float* reward, uint8_t* is_first, uint8_t* is_last,
                            uint8_t* is_terminal, int64_t n, int64_t frame_bytes, int64_t env0,
                            int64_t episode_len, const uint8_t* reset, int32_t* counters, int turn,
                            hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (frame_bytes % 16 != 0 || reinterpret_cast<uint64_t>(image) % 16 != 0 || n > (1 << 29) ||
      frame_bytes > INT32_MAX || env0 > INT32_MAX || episode_len > INT32_MAX)
    return hipErrorInvalidValue;
  SynthArgs a;
  a.counters = counters;
  a.reset = reset;
  a.image = image;
  a.reward = reward;
  a.frame_bytes = static_cast<int32_t>(frame_bytes);
  a.env0 = static_cast<int32_t>(env0);
  a.episode_len = static_cast<int32_t>(episode_len);
  a.n_turn = static_cast<int32_t>(n << 1 | (turn & 1));
  // A frame over a few workgroups: 64 envs x 4 = one workgroup per CU.
  const int64_t vecs = frame_bytes >> 4;
  constexpr int64_t per_env = 4;
  const uint32_t gx = static_cast<uint32_t>(std::max<int64_t>(1, std::min<int64_t>(per_env, vecs / kThreads)));
  const dim3 grid(gx, static_cast<uint32_t>(n));
  const int64_t base = reinterpret_cast<int64_t>(reward);
  const int64_t of = reinterpret_cast<int64_t>(is_first) - base, ol = reinterpret_cast<int64_t>(is_last) - base,
                ot = reinterpret_cast<int64_t>(is_terminal) - base;
  auto fits = [](int64_t x) { return x >= INT32_MIN && x <= INT32_MAX; };
  if (fits(of) && fits(ol) && fits(ot)) {
    a.off_first = static_cast<int32_t>(of);
    a.off_last = static_cast<int32_t>(ol);
    a.off_terminal = static_cast<int32_t>(ot);
    hipLaunchKernelGGL(synth_env_kernel, grid, dim3(kThreads), 0, stream, a.counters, a.reset, a.image,
                       a.reward, a.off_first, a.off_last, a.off_terminal, a.frame_bytes, a.env0,
                       static_cast<int32_t>(static_cast<uint32_t>(a.episode_len) |
                                            (static_cast<uint32_t>(turn & 1) << 31)));
  } else {
    a.off_first = a.off_last = a.off_terminal = 0;
    hipLaunchKernelGGL(synth_env_far_kernel, grid, dim3(kThreads), 0, stream, a, is_first, is_last,
                       is_terminal);
  }
  return hipGetLastError();
}

bool prewrite_supported(const PrewritePlan& p) {
  // (one-dimensional grid of n * (frame blocks + 1) workgroups, 32-bit pixel count)
  return p.n > 0 && p.n <= (1 << 24) && p.pixels > 0 && p.pixels <= INT32_MAX && p.pixels % 4 == 0 &&
         p.channels >= 1 && p.channels <= 4 &&
         reinterpret_cast<uint64_t>(p.frames) % 16 == 0 &&
         reinterpret_cast<uint64_t>(p.frame_pool) % 16 == 0 &&
         (p.pixels * p.channels) % 16 == 0 &&
         reinterpret_cast<uint64_t>(p.dst) % 16 == 0 && p.n_narrow >= 0 && p.n_narrow <= kPreNarrow &&
         (p.out_dtype == kU8 || p.out_dtype == kF16 || p.out_dtype == kBF16 || p.out_dtype == kF32);
}

size_t prewrite_table_bytes(int64_t n) {
  return offsetof(PreTable, words) + static_cast<size_t>(n) * 7 * sizeof(uint32_t);
}

void prewrite_fill_table(void* dst, const PrewritePlan& p, const int32_t* rows, const uint8_t* stepids) {
  // (dst may be write-combined device memory behind the BAR: written once, front to back.)
  PreTable head;
  head.stepid_pool = p.stepid_pool;
  head.rows_out = p.rows_out;
  for (int k = 0; k < kPreNarrow; ++k)
    head.narrow[k] = k < p.n_narrow ? PreKey{p.narrow[k].src, p.narrow[k].pool, p.narrow[k].rowbytes}
                                    : PreKey{nullptr, nullptr, 0};
  head.carry = PreCarry{nullptr, nullptr, nullptr, 0, 0, 1, 0};
  if (p.carry_src && p.carry_rows) {
    const int elem = dtype_size(p.carry_dtype);
    head.carry = PreCarry{p.carry_src, p.carry_pool, p.carry_flags, static_cast<int32_t>(p.carry_rowbytes),
                          p.carry_dtype, elem > 0 ? elem : 1, 0};
  }
  uint8_t* out = static_cast<uint8_t*>(dst);
  std::memcpy(out, &head, offsetof(PreTable, words));
  out += offsetof(PreTable, words);
  std::memcpy(out, rows, static_cast<size_t>(p.n) * sizeof(int32_t));
  out += static_cast<size_t>(p.n) * sizeof(int32_t);
  std::memcpy(out, stepids, static_cast<size_t>(p.n) * kStepBytes);
  out += static_cast<size_t>(p.n) * kStepBytes;
  if (head.carry.src) std::memcpy(out, p.carry_rows, static_cast<size_t>(p.n) * sizeof(int32_t));
}

bool carry_supported(int64_t rowbytes, int dtype) {
  const int elem = dtype_size(dtype);
  return elem > 0 && rowbytes > 0 && rowbytes % elem == 0 && rowbytes / elem <= kThreads &&
         rowbytes <= INT32_MAX;
}

hipError_t launch_obs_stack_insert(const PrewritePlan& p, hipStream_t stream, hipEvent_t stop) {
  if (!prewrite_supported(p) || !p.table_dev) return hipErrorInvalidValue;
  PrewriteArgs a;
  a.frames = p.frames;
  a.dst = p.dst;
  a.frame_pool = p.frame_pool;
  a.pixels = static_cast<int32_t>(p.pixels);
  a.frame_blocks = static_cast<int32_t>(std::min<int64_t>((p.pixels / 4 + kThreads - 1) / kThreads, 32));
  a.scale = p.scale;
  a.offset = p.offset;
  a.n = p.n;
  a.n_narrow = p.n_narrow;
  a.table = static_cast<const PreTable*>(p.table_dev);
  switch (p.out_dtype) {
    case kU8: return obs_stack_insert_typed<uint8_t>(a, p.channels, p.layout, stream, stop);
    case kF16: return obs_stack_insert_typed<__half>(a, p.channels, p.layout, stream, stop);
    case kBF16: return obs_stack_insert_typed<__hip_bfloat16>(a, p.channels, p.layout, stream, stop);
    default: return obs_stack_insert_typed<float>(a, p.channels, p.layout, stream, stop);
  }
}

hipError_t launch_publish_one(const void* src, void* pool, void* out, const int32_t* rows_dev,
                              const uint8_t* flags, int64_t n, int64_t rowbytes, int dtype,
                              hipStream_t stream, hipEvent_t stop, bool flags_by_row) {
  if (n <= 0 || rowbytes <= 0) return hipSuccess;
  PublishArgs a;
  a.src = static_cast<const uint8_t*>(src);
  a.pool = static_cast<uint8_t*>(pool);
  a.out = static_cast<uint8_t*>(out);
  a.rows = rows_dev;
  a.flags = flags;
  a.n = static_cast<int32_t>(n);
  a.rowbytes = static_cast<int32_t>(rowbytes);
  if (flags) {
    a.dtype = dtype | (flags_by_row ? 0x100 : 0);
    a.elem = dtype_size(dtype);
    if (a.elem == 0 || rowbytes % a.elem) return hipErrorInvalidValue;
  } else {
    a.dtype = kU8;       // plain copy: bytes
    a.elem = 1;
  }
  const int64_t elems = n * (rowbytes / a.elem);
  if (n > INT32_MAX || rowbytes > INT32_MAX) return hipErrorInvalidValue;
  const dim3 grid(static_cast<uint32_t>((elems + kThreads - 1) / kThreads));
  if (stop) hipExtLaunchKernelGGL(publish_one_kernel, grid, dim3(kThreads), 0, stream, nullptr, stop, 0,
                                  a.src, a.pool, a.out, a.rows, a.flags, a.n, a.rowbytes, a.dtype, a.elem);
  else hipLaunchKernelGGL(publish_one_kernel, grid, dim3(kThreads), 0, stream,
                          a.src, a.pool, a.out, a.rows, a.flags, a.n, a.rowbytes, a.dtype, a.elem);
  return hipGetLastError();
}

}  // namespace emb
