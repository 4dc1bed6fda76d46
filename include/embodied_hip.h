/* embodied_hip.h — C ABI of libembodied_hip.so (MI355X / gfx950).
 *
 * The reference (danijar/embodied) has no FFI: its boundary is a set of
 * duck-typed Python protocols (embodied/core/base.py:1-73) implemented in numpy.
 * This library is the native layer the MI355X build puts UNDER the same Python
 * classes (embodied_amd.Driver / Replay / selectors / streams / scans).  Every
 * entry point below names the reference code whose work it takes over.
 *
 * Conventions
 *   - every function returns int32 status: 0 ok, <0 error; the message of the
 *     last error on the calling thread is emb_last_error();
 *   - no exceptions and no C++/torch types cross the ABI: plain pointers, sizes;
 *   - device pointers are caller-owned (hipMalloc / torch storage); every launch
 *     goes to the caller's `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream);
 *   - handles are internally locked: add / sample / update may be called from
 *     different host threads (reference: tests/test_replay.py:306-357);
 *   - "host" arrays are ordinary CPU memory, "device" arrays are HBM;
 *   - step ids are 20 bytes: 16-byte big-endian chunk uid || 4-byte big-endian
 *     row-in-chunk (embodied/core/replay.py:90-91).
 */
#ifndef EMBODIED_HIP_H_
#define EMBODIED_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with hidden visibility: what this header declares is all
 * it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* 3 (round 4): + emb_configure, emb_scan_lambda_multi, emb_replay_carry_publish,
 * emb_replay_settle; emb_replay_profile_report which = 3.  Additions only: a
 * caller written against version 2 runs unchanged.                             */
/* 4 (round 5): + emb_replay_sample_heads, emb_direct_*.  Additions only.        */
/* 5 (round 6): + emb_comm_exchange_gather, emb_direct_allgather,
 * emb_direct_exchange_gather (additions); emb_configure refuses names that are
 * no knob; a time-out inside emb_direct_* is fatal for the communicator.        */
#define EMB_ABI_VERSION 5

#define EMB_OK 0
#define EMB_ERR_INVALID (-1)   /* bad argument / state                         */
#define EMB_ERR_HIP (-2)       /* a HIP runtime call failed                     */
#define EMB_ERR_EMPTY (-3)     /* sampling from an empty selector               */
#define EMB_ERR_POOL_FULL (-4) /* chunk pool exhausted: grow it and retry       */
#define EMB_ERR_NOT_FOUND (-5) /* unknown key                                   */
#define EMB_ERR_INTERNAL (-6)

#define EMB_STEPID_BYTES 20

/* dtype codes (emb_mask_actions, emb_obs_stack) */
#define EMB_U8 0
#define EMB_I8 1
#define EMB_I16 2
#define EMB_I32 3
#define EMB_I64 4
#define EMB_F16 5
#define EMB_BF16 6
#define EMB_F32 7
#define EMB_F64 8
#define EMB_BOOL 9

#define EMB_LAYOUT_SAME 0           /* (N, P, C) as stored by the envs          */
#define EMB_LAYOUT_CHANNELS_FIRST 1 /* (N, C, P)                                */

#define EMB_MODE_TRAIN 0
#define EMB_MODE_REPORT 1
#define EMB_MODE_EVAL 2

typedef struct emb_rng emb_rng_t;
typedef struct emb_tree emb_tree_t;
typedef struct emb_selector emb_selector_t;
typedef struct emb_replay emb_replay_t;

const char* emb_last_error(void);
int32_t emb_abi_version(void);

/* Tuning knobs (INTEGRATION.md lists them: EMB_SPAN_MOVER, EMB_DEFER_INDEX,
 * EMB_DEFER_MAX_GAP_US, EMB_ARGS_BAR, ...; csrc/knobs.h holds the list).  Each
 * is read once, by the first call that needs it, from emb_configure's value or
 * else from the environment variable of the same name.  emb_configure after
 * that first use is refused (EMB_ERR_INVALID): a setting is never half in
 * effect; a name that is no knob of the library is refused as well (ABI 5: a
 * retired or misspelt knob is an error, not a silent no-op).  value NULL
 * withdraws an earlier emb_configure.  The reference has no counterpart (its knobs are
 * Python config fields); this replaces "export EMB_...=" for host programs. */
int32_t emb_configure(const char* name, const char* value);
int32_t emb_device_count(int32_t* count);

/* A HIP stream whose kernels may use `n_cus` compute units only, beginning with
 * unit `first_cu` (ABI 5).  For a learner that samples / scans / writes back
 * beside the Driver's latency-bound step kernels (the reference dispatches its
 * train step asynchronously too, embodied/jax/agent.py:286-294): a sample gather
 * that fills every CU makes the step's small kernels queue behind it; confined
 * to a share of the chip it runs beside them.  The replay's movers size their
 * persistent grids by the stream's share.  The driver numbers compute units
 * round-robin over XCDs and shader engines, so a range is spread evenly.
 * emb_stream_destroy for streams made here only.                               */
int32_t emb_stream_create_on_cus(int32_t first_cu, int32_t n_cus, void** stream_out);
int32_t emb_stream_destroy(void* stream);

/* ---- numpy-compatible PRNG ------------------------------------------------
 * Replaces numpy.random.default_rng as used by selectors.py:34,42,240,305,212
 * and the per-batch seeds of embodied/jax/agent.py:405-408.  `words` is the
 * SeedSequence entropy: a Python int seed split into little-endian u32 words
 * (a list seed concatenates its elements' words).                            */
int32_t emb_rng_create(const uint32_t* words, int32_t n_words, emb_rng_t** out);
int32_t emb_rng_integers(emb_rng_t* rng, int64_t high, int64_t count, int64_t* out);
int32_t emb_rng_random(emb_rng_t* rng, int64_t count, double* out);
int32_t emb_rng_choice(emb_rng_t* rng, const double* p, int32_t k, int64_t count, int64_t* out);
int32_t emb_rng_destroy(emb_rng_t* rng);
/* ndarray.sum() of a contiguous float64 vector (pairwise), selectors.py:296. */
int32_t emb_np_sum(const double* values, int64_t n, double* out);

/* ---- SampleTree (selectors.py:231-354) ------------------------------------ */
int32_t emb_tree_create(int32_t branching, uint64_t seed, emb_tree_t** out);
int32_t emb_tree_insert(emb_tree_t* tree, int64_t key, double uprob);
int32_t emb_tree_remove(emb_tree_t* tree, int64_t key);
int32_t emb_tree_update(emb_tree_t* tree, int64_t key, double uprob);
int32_t emb_tree_sample(emb_tree_t* tree, int64_t* key);
int32_t emb_tree_len(emb_tree_t* tree, int64_t* n);
int32_t emb_tree_root_sum(emb_tree_t* tree, double* total);
/* leaf depths (root = 0) and node count incl. leaves: what the reference's
 * tests/test_sampletree.py:18-58 inspect.                                    */
int32_t emb_tree_shape(emb_tree_t* tree, int64_t cap, int64_t* depths, int64_t* n_leaves,
                       int64_t* n_nodes);
int32_t emb_tree_destroy(emb_tree_t* tree);

/* ---- selectors (selectors.py:7-228): __call__/__len__/__setitem__/__delitem__
 * /prioritize.  Item keys are int64 (Replay's itemid counter).              */
int32_t emb_selector_create_fifo(emb_selector_t** out);
int32_t emb_selector_create_uniform(uint64_t seed, emb_selector_t** out);
int32_t emb_selector_create_prioritized(double exponent, double initial, int32_t zero_on_sample,
                                        double maxfrac, int32_t branching, uint64_t seed,
                                        emb_selector_t** out);
/* members: name-sorted, zero fractions already dropped (selectors.py:205-211);
 * the mixture shares the members, which stay valid handles of their own.     */
int32_t emb_selector_create_mixture(emb_selector_t* const* members, const float* fractions,
                                    int32_t n, uint64_t seed, emb_selector_t** out);
/* Recency (embodied/core/selectors.py:60-125): age-biased draws from a b-ary
 * table of normalised block masses.  `table` = the levels of the reference's
 * `_build(uprobs)` (:107-125) one after the other, level l as bfactor^l rows of
 * bfactor probabilities (table_len = sum over levels); `entries` = len(uprobs).
 * One Generator.choice per level and draw (:98-105, with the one-token repair
 * DESIGN.md 6 describes), age scaled while fewer than `entries` items are held
 * (:77-79), a vanished age drawn again (:75-88 sleeps and retries).            */
int32_t emb_selector_create_recency(const double* table, int64_t table_len, int32_t depth,
                                    int32_t bfactor, int64_t entries, uint64_t seed,
                                    emb_selector_t** out);
/* a caller-implemented selector (any Python object with the protocol).       */
typedef struct {
  void* user;
  int64_t (*sample)(void* user);
  int64_t (*size)(void* user);
  void (*insert)(void* user, int64_t key, const uint8_t* stepids, int32_t n_steps);
  void (*remove)(void* user, int64_t key);
  void (*prioritize)(void* user, const uint8_t* stepids, const double* prios, int64_t n); /* may be NULL */
} emb_selector_callbacks_t;
int32_t emb_selector_create_callback(const emb_selector_callbacks_t* cb, emb_selector_t** out);
int32_t emb_selector_insert(emb_selector_t* sel, int64_t key, const uint8_t* stepids, int32_t n_steps);
int32_t emb_selector_remove(emb_selector_t* sel, int64_t key);
int32_t emb_selector_sample(emb_selector_t* sel, int64_t* key);
int32_t emb_selector_len(emb_selector_t* sel, int64_t* n);
int32_t emb_selector_prioritize(emb_selector_t* sel, const uint8_t* stepids, const double* prios, int64_t n);
int32_t emb_selector_destroy(emb_selector_t* sel);

/* ---- Replay (embodied/core/replay.py, chunk.py) ---------------------------
 * Payload lives in a caller-owned device chunk pool: for key k a buffer of
 * (n_slots * chunksize) rows of rowbytes[k] bytes.  The library keeps the
 * integer state (items, FIFO, chunk refcounts, online queue, selector) on the
 * host and moves rows with HIP kernels.                                      */
typedef struct {
  int64_t length;     /* Replay(length=...)                replay.py:16-21     */
  int64_t capacity;   /* in items; 0 = unbounded                              */
  int64_t chunksize;  /*                                    chunk.py:13        */
  int64_t n_slots;    /* chunk slots in the device pool                       */
  int32_t online;     /*                                    replay.py:39-42    */
  int32_t reserved;
  uint64_t uid_hi;    /* high 64 bits of chunk uids (replica id)              */
  /* Sharded pools (0/1 = off): worker w belongs to owner w / workers_per_owner,
   * whose chunks only use slots [o, o+1) * n_slots / owners.                 */
  int64_t owners;
  int64_t workers_per_owner;
} emb_replay_config_t;

/* selector may be NULL: Uniform(seed), as replay.py:26.                      */
int32_t emb_replay_create(const emb_replay_config_t* cfg, emb_selector_t* selector,
                          uint64_t seed, emb_replay_t** out);
int32_t emb_replay_destroy(emb_replay_t* rep);
/* Column schema, discovered from the first add (chunk.py:43-47).  Keys named
 * "stepid" (20 B), "is_first", "is_last" (1 B) get their reference meaning.  */
int32_t emb_replay_set_keys(emb_replay_t* rep, int32_t n_keys, const char* const* names,
                            const int64_t* rowbytes, void* const* pools);
/* After the caller re-allocated a larger pool (rows copied by the caller).   */
int32_t emb_replay_grow(emb_replay_t* rep, int64_t n_slots, void* const* pools);

/* Host-only index steps (no GPU needed; bit-exact with the reference):
 * add_index    = bookkeeping of Replay.add for n steps, one per workers[i], in
 *                order (replay.py:77-118); rows_out[i] = pool row to write,
 *                stepids_out = n x 20 bytes; new_chunks_out = chunks this call
 *                opened in RECYCLED slots (slots an evicted chunk had held).  A
 *                caller that batches the payload writes must, whenever it is
 *                non-zero, write out what it had batched BEFORE this call ahead
 *                of this call's rows: rows still waiting for the old chunk and
 *                rows of the new one would otherwise meet in one launch.
 * sample_index = `batch` sequence draws (replay.py:121-127,151-169,193-214):
 *                rows_out[batch*length] pool rows, online_out[batch] flags,
 *                workers_out[batch] the worker stream of each sequence.
 * resolve      = Replay.update's decode of stepid[i,0] -> `count` pool rows
 *                (replay.py:139-149,216-235); evicted -> rows -1, found 0.   */
int32_t emb_replay_add_index(emb_replay_t* rep, int64_t n, const int64_t* workers,
                             int32_t* rows_out, uint8_t* stepids_out, int32_t* new_chunks_out);
int32_t emb_replay_sample_index(emb_replay_t* rep, int64_t batch, int32_t mode,
                                int32_t* rows_out, uint8_t* online_out, int64_t* workers_out);
int32_t emb_replay_resolve(emb_replay_t* rep, int64_t n, const uint8_t* stepids, int64_t count,
                           int32_t* rows_out, uint8_t* found_out);
int32_t emb_replay_prioritize(emb_replay_t* rep, const uint8_t* stepids, const double* prios,
                              int64_t n);
int32_t emb_replay_len(emb_replay_t* rep, int64_t* items);
int32_t emb_replay_sampler_len(emb_replay_t* rep, int64_t* n);
/* Windows queued by online mode (replay.py:114-118) that the next train-mode
 * samples will serve before the selector is asked.                           */
int32_t emb_replay_online_pending(emb_replay_t* rep, int64_t* n);
int32_t emb_replay_free_slots(emb_replay_t* rep, int64_t* n);
/* out = {items, chunks, streams, inserts, samples, updates} (replay.py:58-74) */
int32_t emb_replay_stats(emb_replay_t* rep, int64_t out[6], int32_t reset);

/* Fused device steps: index + row table upload + ONE kernel launch.
 * add:    src[k] = device (n, rowbytes[k]) for every key except "stepid",
 *         which the library synthesises (entry ignored).   Replay.add
 * sample: dst[k] = device (batch, length, rowbytes[k]); is_first / is_last are
 *         annotated in flight (replay.py:277-292); online_out[batch] and
 *         first_stepids_out[batch*20] (host, optional) receive the online-queue
 *         flags and stepid[:,0] without a device read-back.   Replay.sample
 * update: stepids = host (B, 20) first step of each row; key_ids/src select the
 *         columns to overwrite, src[j] = device (B, T, rowbytes).  Replay.update */
int32_t emb_replay_add(emb_replay_t* rep, int64_t n, const int64_t* workers,
                       const void* const* src, void* stream);
/* Replay.add with the Driver's action mask fused in (driver.py:72-74,84-87 +
 * replay.py:77-118): the n_masked keys masked_keys[j] (replay key ids, dtype
 * codes masked_dtypes[j] = EMB_U8 .. EMB_BOOL) are stored as src * !is_last[row] in
 * their own dtype (a real multiply: -x -> -0.0, NaN stays NaN), and the same
 * masked values are written to masked_out[j] (device (n, rowbytes), may be
 * NULL): the actions the next env step receives.  is_last = device uint8/bool
 * (n).  One launch instead of mask + insert. */
int32_t emb_replay_add_masked(emb_replay_t* rep, int64_t n, const int64_t* workers,
                              const void* const* src, int32_t n_masked,
                              const int32_t* masked_keys, const int32_t* masked_dtypes,
                              void* const* masked_out, const void* is_last, void* stream);

/* Early insert: the vectorised step's observation keys go to their pool rows
 * BEFORE the policy runs, in the launch that builds the policy batch, so that
 * every frame is read once (driver.py:65 np.stack + jax/agent.py:230 layout /
 * dtype change + replay.py:77-118 / chunk.py:41-50 row copy in one pass).
 *
 * obs_stack_insert = emb_obs_stack(frames -> dst as `spec` says) and, when the
 * rows of the next add of exactly `workers` (one step each) are known — every
 * worker already has an open chunk and is listed once — the same launch also
 * writes: the frames to the pool rows of key `frame_key`, every other key k
 * with src[k] != NULL and rowbytes <= 256 (device (n, rowbytes[k]): reward,
 * flags ...; at most 8) and the step ids.  Nothing in the index changes: a
 * worker's next row is the cursor of its open chunk, and no item can reach a
 * row before it is published.  *token_out = a non-zero token if the early
 * insert happened, 0 if the call only did the obs stack (unknown worker, odd
 * frame shape, src[frame_key] != frames).
 *
 * publish = emb_replay_add_masked for the same step.  With the token of the
 * early insert, the same workers and stream, keys whose src[k] is the buffer
 * the early insert copied from are not copied again; what is left (the action)
 * goes out with a 56-byte-argument launch when it is a single small key.  With
 * token 0, or when anything does not match, publish is add_masked.            */
typedef struct emb_obs_spec {
  int64_t pixels, channels;     /* frames are (n, pixels, channels) uint8           */
  int32_t layout, out_dtype;    /* EMB_LAYOUT_*, EMB_U8 / EMB_F16 / EMB_BF16 / EMB_F32 */
  float scale, offset;          /* float outputs: value * scale + offset            */
} emb_obs_spec_t;
int32_t emb_replay_obs_stack_insert(emb_replay_t* rep, int64_t n, const int64_t* workers,
                                    int32_t frame_key, const void* frames,
                                    const emb_obs_spec_t* spec, void* dst, const void* const* src,
                                    void* stream, uint64_t* token_out);
int32_t emb_replay_publish(emb_replay_t* rep, int64_t n, const int64_t* workers,
                           const void* const* src, int32_t n_masked, const int32_t* masked_keys,
                           const int32_t* masked_dtypes, void* const* masked_out,
                           const void* is_last, uint64_t token, void* stream);

/* Carried publish (round 4).  With the early insert in use, what a publish has
 * left is usually one small masked key (the action): a launch of its own whose
 * only cost is its place in the chain of dependent launches of an env step.
 * emb_replay_carry_publish(rep, 1) lets emb_replay_publish skip that launch when
 * the caller passes a NULL masked_out for the key (nobody needs the masked
 * values back -- Driver: the env takes unmasked actions together with `reset`,
 * driver.py:72-75): the key is written, masked, by the NEXT
 * emb_replay_obs_stack_insert launch on the same stream, or by a launch of its
 * own before any other call reads or writes the pool (sample, update, add,
 * gather / scatter rows, chunk bookkeeping, grow all settle it first).
 * Contract while it is enabled: the key's source buffer of a publish stays
 * unchanged until the next call on this replay that launches (its early insert,
 * or anything that settles).  The is_last flags need not: a publish is carried
 * only when the step's early insert stored that very flag buffer as the
 * replay's `is_last` key, and the mask then reads the stored rows -- an env that
 * rewrites its flag outputs in place on the next step is fine.
 * emb_replay_settle(rep) settles explicitly (before the caller moves the pool,
 * reads it directly or frees the source).  Results are identical to an
 * immediate publish.                                                            */
int32_t emb_replay_carry_publish(emb_replay_t* rep, int32_t enable);
int32_t emb_replay_settle(emb_replay_t* rep);
int32_t emb_replay_sample(emb_replay_t* rep, int64_t batch, int32_t mode, void* const* dst,
                          uint8_t* online_out, uint8_t* first_stepids_out, void* stream);
/* The same with per-key head lengths: key k receives only the first
 * key_len[k] steps of every sampled sequence, dst[k] is (batch, key_len[k],
 * rowbytes[k]); 0 (or `length`) = the whole sequence, key_len NULL = all whole.
 * For keys whose consumer reads only the head of the window -- the reference's
 * DreamerV3 takes x[:, :K] (K = replay_context) of the sampled enc/ dyn/ dec/
 * latents (dreamerv3/agent.py:322-331) and its _assemble_batch already copies a
 * [start, stop) sub-range (embodied/core/replay.py:255-275).  Index draws,
 * PRNG stream and the is_first / is_last annotation (computed over the whole
 * sequence, then cut) are those of emb_replay_sample: every key equals the
 * full sample's [:, :key_len[k]].                                              */
int32_t emb_replay_sample_heads(emb_replay_t* rep, int64_t batch, int32_t mode, void* const* dst,
                                const int32_t* key_len, uint8_t* online_out,
                                uint8_t* first_stepids_out, void* stream);
/* The same with the batch side cut into groups of `group` sequences whose
 * starts are `group_stride` bytes (a multiple of 16) apart: dst[k] is key k's
 * place inside group 0, sequence s of key k lands at dst[k] + (s / group) *
 * group_stride + (s % group) * length * rowbytes[k].  This is the layout of a
 * packed batch cut into one block per destination rank, so that the DP-slice
 * exchange of SURVEY.md 8e (the reference assembles per-process slices into the
 * global batch, embodied/jax/internal.py:145-152) is ONE all-to-all.         */
int32_t emb_replay_sample_grouped(emb_replay_t* rep, int64_t batch, int32_t mode,
                                  void* const* dst, int32_t group, int64_t group_stride,
                                  uint8_t* online_out, uint8_t* first_stepids_out, void* stream);
int32_t emb_replay_update(emb_replay_t* rep, int64_t B, int64_t T, const uint8_t* stepids,
                          int32_t n_keys, const int32_t* key_ids, const void* const* src,
                          void* stream);
/* Move rows given an explicit host row table (multi-GPU owner-side gather,
 * fused sample+windowing, load from disk).  gather: a NULL dst[k] skips key k;
 * the flag annotation applies per `seq_len` rows.                             */
int32_t emb_replay_gather_rows(emb_replay_t* rep, const int32_t* rows, int64_t n_rows,
                               int64_t seq_len, void* const* dst, void* stream);
int32_t emb_replay_scatter_rows(emb_replay_t* rep, const int32_t* rows, int64_t n_rows,
                                int32_t n_keys, const int32_t* key_ids, const void* const* src,
                                void* stream);

/* Callers that add/update and sample from DIFFERENT streams (actor stream +
 * learner stream, run/parallel.py's split in one process) enable this: pool
 * writes and reads are then ordered across streams with events.             */
int32_t emb_replay_multistream(emb_replay_t* rep, int32_t enable);

/* HIP-event timing of the gather launches issued through this handle (on their
 * own stream): total milliseconds and launch count since the last reset.
 * enable = 1 stamps every gather, n > 1 every n-th one (a stamped launch costs
 * the host more than a plain one), 0 switches the stamps off.                */
int32_t emb_replay_profile(emb_replay_t* rep, int32_t enable);
int32_t emb_replay_profile_read(emb_replay_t* rep, int64_t* launches, double* total_ms,
                                int32_t reset);
/* The same counters per launch kind, with the name of the kernel the last
 * stamped launch of that kind ran (as a profiler prints it, without namespaces):
 * which = 0 the sample gathers, 1 the write-backs of emb_replay_update (stamped
 * at the same rate while profiling is on), 2 no kernel: `launches` = the
 * emb_replay_publish calls that handed their index bookkeeping to the library's
 * helper thread (total_ms: how many early inserts took predicted rows instead of
 * waiting for it); which = 3: carried publishes that rode in the next early
 * insert (total_ms: all carried publishes).  kernel_out may be NULL.            */
int32_t emb_replay_profile_report(emb_replay_t* rep, int32_t which, int64_t* launches,
                                  double* total_ms, int32_t reset, char* kernel_out,
                                  int32_t kernel_cap);

/* Checkpoint support (replay.py:294-388): chunk table in/out.
 * complete_all closes every worker's open chunk (replay.py:297-299); it needs
 * one free slot per open chunk (emb_replay_open_chunks) and fails with
 * EMB_ERR_POOL_FULL, leaving the index unchanged, when they are not there.
 * reserve_uids: chunk serials below `serial` are not issued (chunk files of an
 * earlier run in the same directory keep their ids: chunk.py:15-16 draws
 * random UUIDs, this build counts).                                          */
int32_t emb_replay_complete_all(emb_replay_t* rep);
int32_t emb_replay_open_chunks(emb_replay_t* rep, int64_t* n);
int32_t emb_replay_reserve_uids(emb_replay_t* rep, uint64_t serial);
int32_t emb_replay_chunks(emb_replay_t* rep, int64_t cap, uint64_t* uid, uint64_t* succ,
                          int64_t* fill, int64_t* slot, int64_t* time_ms, int64_t* n);
int32_t emb_replay_load_chunk(emb_replay_t* rep, uint64_t uid, uint64_t succ, int64_t fill,
                              int64_t time_ms, int64_t* slot);
int32_t emb_replay_load_items(emb_replay_t* rep, uint64_t uid, int64_t amount);

/* ---- Driver-side kernels (embodied/core/driver.py:55-87) ------------------ */
/* np.stack of per-env frames into the policy batch (driver.py:65) fused with
 * the layout/dtype change the agent applies next (jax/agent.py:230):
 * src = device (n_envs_total, pixels, channels) u8 slab; batch row j reads env
 * env_ids[j] (host int32[n], NULL = identity).                               */
int32_t emb_obs_stack(const void* src, const int32_t* env_ids, int64_t n, int64_t pixels,
                      int64_t channels, int32_t layout, int32_t out_dtype, float scale,
                      float offset, void* dst, void* stream);
/* emb_mask_actions that also tells the HOST when it is done (ABI 5): `flag`, a
 * 32-bit word in pinned, device-mapped host memory, receives `seq` once every row
 * has been stored (system-scope release behind the stores).  With `out` in pinned
 * memory as well this is how the Driver brings the next step's masked actions
 * (driver.py:72-75) down to its env processes: the host reads one word instead of
 * recording and waiting for an event.  `counter`: a zeroed 32-bit device word of
 * the caller's, zero again afterwards; one launch at a time per counter.        */
int32_t emb_mask_actions_notify(const void* act, void* out, int64_t n, int64_t row_elems,
                                int32_t dtype, const void* is_last, void* counter, void* flag,
                                uint32_t seq, void* stream);

/* dst[0 .. bytes) = src[0 .. bytes) by a KERNEL on `stream` (ABI 5).  `src` may be
 * pinned, device-mapped host memory: the Driver brings a piece of its shared
 * observation slab (embodied/core/driver.py:17-25,61-65: what env processes wrote)
 * to the device this way while other env processes are still stepping -- a copy
 * through the DMA engines costs ~10 us of set-up per call on this system, a kernel
 * that reads across PCIe does not.  Any alignment.                              */
int32_t emb_copy_bytes(const void* src, void* dst, int64_t bytes, void* stream);

/* acts zeroed where is_last (driver.py:72-74,84-87): out = act * ~is_last as a
 * real multiply in `dtype`; act, out (n, row_elems), out may alias act.      */
int32_t emb_mask_actions(const void* act, void* out, int64_t n, int64_t row_elems,
                         int32_t dtype, const void* is_last, void* stream);
/* Per-env policy carry rows by env id (embodied/jax/agent.py:173-181,
 * run/parallel.py:94-104): dst[j] = table[ids[j]] / table[ids[j]] = src[j].  */
int32_t emb_rows_gather(const void* table, int64_t rowbytes, const int32_t* ids, int64_t n,
                        void* dst, void* stream);
int32_t emb_rows_scatter(void* table, int64_t rowbytes, const int32_t* ids, int64_t n,
                         const void* src, void* stream);
/* Sequence windowing (streams.py:133-138): src (B, total, rowbytes) ->
 * dst (B, count, rowbytes) = src[:, start:start+count].                      */
int32_t emb_window(const void* src, void* dst, int64_t batch, int64_t total, int64_t start,
                   int64_t count, int64_t rowbytes, void* stream);
/* The same for every key of a batch in ONE launch: src[k] (B, total, rowbytes[k])
 * -> dst[k] (B, count, rowbytes[k]).                                          */
int32_t emb_window_keys(int32_t n_keys, const void* const* src, void* const* dst,
                        const int64_t* rowbytes, int64_t batch, int64_t total, int64_t start,
                        int64_t count, void* stream);

/* ---- return scans, float32 on device ------------------------------------- */
/* PPO GAE (ppo/agent.py:188-201): rew,val (B,T) f32; last,term (B,T) u8 ->
 * adv,tar (B,T-1).  live_scale = 1 - 1/hor.                                  */
int32_t emb_scan_gae(const void* rew, const void* val, const void* last, const void* term,
                     int64_t B, int64_t T, float live_scale, float lam, void* adv, void* tar,
                     void* stream);
/* The same with rew / last / term read straight out of a grouped packed batch
 * (emb_replay_sample_grouped, or the buffer a DP-slice all-to-all delivered):
 * row b of a key starts (b / group) * group_stride + (b % group) * T * itemsize
 * bytes into it; val, adv and tar are dense.  group = 0: identical to
 * emb_scan_gae.                                                               */
int32_t emb_scan_gae_grouped(const void* rew, const void* val, const void* last, const void* term,
                             int64_t B, int64_t T, float live_scale, float lam, void* adv,
                             void* tar, int64_t group, int64_t group_stride, void* stream);
/* DreamerV3 lambda-return (dreamerv3/agent.py:482-490) -> ret (B,T-1).       */
int32_t emb_scan_lambda(const void* last, const void* term, const void* rew, const void* boot,
                        int64_t B, int64_t T, float disc, float lam, void* ret, void* stream);
/* Several lambda-return problems of one train step in ONE launch: DreamerV3
 * computes the replay returns (B,T) and the imagined returns (B*K,H+1) in the
 * same step (dreamerv3/agent.py:401-405 imag_loss, :464-466 repl_loss, both
 * through lambda_return :482-490); at those sizes each scan is launch latency.
 * Same arithmetic as emb_scan_lambda per problem.  Up to 4 problems with rows of
 * at most 257 steps share a launch; anything else runs one launch each.        */
typedef struct emb_lambda_problem {
  const void* last; const void* term; const void* rew; const void* boot; void* ret;
  int64_t B, T;
  float disc, lam;
} emb_lambda_problem_t;
int32_t emb_scan_lambda_multi(int32_t n_problems, const emb_lambda_problem_t* problems, void* stream);
/* Director critic target, time-major (director/agent.py:430-445): rew (T-1,B),
 * cont,value (T,B) -> ret (T-1,B).  discount = 1 - 1/horizon.                */
int32_t emb_scan_director(const void* rew, const void* cont, const void* value, int64_t T,
                          int64_t B, float discount, float lam, void* ret, void* stream);

/* Director manager steps (director/hierarchy.py:240-256), time-major: windows
 * of k steps; reward (T-1,B) f32 [shifted by one step inside], cont (T,B) f32 ->
 * reward_out (T/k-1,B) = mean of cumprod(cont)-weighted rewards per window,
 * cont_out (T/k,B) = product of cont per window.  Either output may be NULL.  */
int32_t emb_abstract_traj(const void* reward, const void* cont, int64_t T, int64_t B, int32_t k,
                          void* reward_out, void* cont_out, void* stream);

/* ----------------------------------------------------------- collectives --
 * The two exchange steps of the sharded path on RCCL directly (xGMI inside one
 * node), for hosts that do not go through torch.distributed:
 *   trajectories  -> one all-gather of the packed (B, L, S) byte block per rank
 *                    (or all-to-all of its per-rank blocks: the DP slice)
 *   gradients     -> all-reduce (sum or mean) of one flat buffer
 *                    (embodied/jax/opt.py:52-54's pmean)
 * RCCL is opened with dlopen at the first call (the library itself does not
 * link against it).  Rank 0 makes the 128-byte id, the caller hands it to the
 * other ranks by whatever means it has (file, socket, torch.distributed
 * store), every rank calls emb_comm_init with the device it uses current.
 * Collectives are asynchronous on `stream`; buffers are caller-owned.         */
typedef struct emb_comm emb_comm_t;
#define EMB_COMM_ID_BYTES 128
int32_t emb_comm_unique_id(uint8_t* id_out /* [EMB_COMM_ID_BYTES] */);
int32_t emb_comm_init(const uint8_t* id, int32_t rank, int32_t world, emb_comm_t** out);
int32_t emb_comm_allgather_traj(emb_comm_t* comm, const void* send, void* recv,
                                int64_t bytes_per_rank, void* stream);
int32_t emb_comm_allreduce_grads(emb_comm_t* comm, void* buf, int64_t count, int32_t mean,
                                 void* stream);
/* The same with the element type named (EMB_F16 / EMB_BF16 / EMB_F32 / EMB_F64):
 * bf16 gradients halve the bytes on the links.                  */
int32_t emb_comm_allreduce_grads_as(emb_comm_t* comm, void* buf, int64_t count, int32_t dtype,
                                    int32_t mean, void* stream);
/* DP-slice exchange (SURVEY.md 8e "only its DP slice via all-to-all"): `send`
 * and `recv` hold `world` blocks of bytes_per_rank bytes; block r of `send`
 * goes to rank r, block r of `recv` arrives from rank r (block `rank` is a
 * local copy).  One fused group of RCCL point-to-point transfers.            */
int32_t emb_comm_alltoall_slices(emb_comm_t* comm, const void* send, void* recv,
                                 int64_t bytes_per_rank, void* stream);
/* Normaliser statistics over the data-parallel ranks (embodied/jax/utils.py:76-88;
 * used by the agents' return / advantage / value normalisers, ppo/agent.py:35-36,
 * dreamerv3/agent.py:70-72).
 * allgather_returns: every rank's `count` float32 values -> recv[world * count],
 *   rank order: the `perc` normaliser's jax.lax.all_gather in front of
 *   jnp.percentile (utils.py:83-88).
 * pmean_scalars: in-place mean over the ranks of `count` float32 values: the
 *   jax.lax.pmean of Normalize._mean (utils.py:76-81).                         */
int32_t emb_comm_allgather_returns(emb_comm_t* comm, const void* send, void* recv, int64_t count,
                                   void* stream);
int32_t emb_comm_pmean_scalars(emb_comm_t* comm, void* values, int64_t count, void* stream);

/* One train step's exchange, asynchronous on the communicator's OWN stream so
 * that it overlaps what the caller queues next (the reference hands train
 * outputs out one step late for the same reason, embodied/jax/agent.py:286-294):
 * the collectives start after everything queued on `after_stream` so far --
 * first the slices (bytes_per_rank > 0), then the gradients (count > 0).
 * emb_comm_wait makes `stream` wait for the exchange issued last; buffers stay
 * the caller's and must live until then.  Host cost: two event records and two
 * stream waits per train step, whatever the number of collectives.            */
int32_t emb_comm_exchange(emb_comm_t* comm, void* after_stream, const void* slices_send,
                          void* slices_recv, int64_t bytes_per_rank, void* grads, int64_t count,
                          int32_t dtype, int32_t mean);
/* The same with the trajectory ALL-GATHER in front of the gradients instead of the
 * DP-slice all-to-all (north_star's "RCCL all-gather of trajectories"; the
 * reference assembles every process's batch slice into the global batch,
 * embodied/jax/internal.py:145-152): `traj_send` = this rank's bytes_per_rank
 * bytes, `traj_recv` = world * bytes_per_rank bytes in rank order.  (ABI 5)    */
int32_t emb_comm_exchange_gather(emb_comm_t* comm, void* after_stream, const void* traj_send,
                                 void* traj_recv, int64_t bytes_per_rank, void* grads,
                                 int64_t count, int32_t dtype, int32_t mean);
int32_t emb_comm_wait(emb_comm_t* comm, void* stream);
int32_t emb_comm_destroy(emb_comm_t* comm);

/* ---- synthetic vector env (benchmark / test input) ---------- */
/* Episode logic of embodied/envs/dummy.py:38-48 for n device-resident envs;
 * counters = device int32[2][2n] state in two generations: the step reads
 * generation `turn` (0/1) and writes the other one (the caller alternates);
 * reset = device u8[n] or NULL.                                               */
int32_t emb_synth_env_step(void* image, void* reward, void* is_first, void* is_last,
                           void* is_terminal, int64_t n, int64_t frame_bytes, int64_t env0,
                           int64_t episode_len, const void* reset, void* counters,
                           int32_t turn, void* stream);

/* ---- direct xGMI schedule (round 5) ----------------------------------------
 * The two collectives of a train step -- the gradient all-reduce
 * (embodied/jax/opt.py:52-54) and the DP-slice all-to-all
 * (embodied/jax/internal.py:145-152) -- without RCCL: every rank writes its
 * peers' shares straight into their memory (hipIpc handles of one uncached,
 * fine-grained allocation per rank), all n-1 peers at once, one xGMI link each; flags in that
 * memory order the steps (csrc/direct_comm.hip states the schedule).  One node,
 * at most 8 ranks, one GPU per rank (or several ranks on one GPU: the tests).
 *
 * create: max_reduce_bytes = the largest gradient buffer, max_block_bytes = the
 *         largest all-to-all / all-gather block (bytes per rank); every wait
 *         inside a kernel gives up after timeout_ms.  A time-out is FATAL for
 *         the communicator (RCCL would keep waiting; a bounded wait must not
 *         turn into a silently wrong gradient): the kernel that gave up writes
 *         no result and raises no peer's flag, every later kernel returns at
 *         once and every later call of allreduce / alltoall / allgather /
 *         exchange / wait returns an error (the word is read from host-mapped
 *         memory, no synchronisation); emb_direct_status synchronises and
 *         reports it.  Choose timeout_ms like a collective watchdog (minutes),
 *         not like a latency bound.
 * handle / connect: each rank's 64-byte handle reaches every other rank by the
 *         caller's means (a process-group all-gather, a file); connect takes all
 *         `world` handles in rank order.
 * allreduce / alltoall / allgather: asynchronous on `stream`; same order of
 *         calls on every rank.  Results: the sum is taken in rank order in f32 by
 *         the rank that owns the shard and broadcast, so every rank holds the
 *         same bits.  The all-reduce buffer must be 16-byte aligned.  allgather
 *         (ABI 5): this rank's bytes_per_rank bytes reach every peer, all n-1
 *         links at once; recv = world * bytes_per_rank bytes in rank order.
 * exchange / exchange_gather / wait: emb_comm_exchange[_gather]'s contract on
 *         the transport's own stream.                                         */
#define EMB_DIRECT_HANDLE_BYTES 64
typedef struct emb_direct emb_direct_t;
int32_t emb_direct_create(int32_t rank, int32_t world, int64_t max_reduce_bytes,
                          int64_t max_block_bytes, int32_t timeout_ms, emb_direct_t** out);
int32_t emb_direct_handle(emb_direct_t* d, uint8_t* handle_out);
int32_t emb_direct_connect(emb_direct_t* d, const uint8_t* handles);
int32_t emb_direct_allreduce(emb_direct_t* d, void* buf, int64_t count, int32_t dtype,
                             int32_t mean, void* stream);
int32_t emb_direct_alltoall(emb_direct_t* d, const void* send, void* recv,
                            int64_t bytes_per_rank, void* stream);
int32_t emb_direct_exchange(emb_direct_t* d, void* after_stream, const void* slices_send,
                            void* slices_recv, int64_t bytes_per_rank, void* grads,
                            int64_t count, int32_t dtype, int32_t mean);
int32_t emb_direct_allgather(emb_direct_t* d, const void* send, void* recv,
                             int64_t bytes_per_rank, void* stream);
int32_t emb_direct_exchange_gather(emb_direct_t* d, void* after_stream, const void* traj_send,
                                   void* traj_recv, int64_t bytes_per_rank, void* grads,
                                   int64_t count, int32_t dtype, int32_t mean);
int32_t emb_direct_wait(emb_direct_t* d, void* stream);
/* The watchdog of the launches from now on (a short one while a set-up is being
 * checked, minutes for the job itself).  (ABI 5)                               */
int32_t emb_direct_set_timeout(emb_direct_t* d, int32_t timeout_ms);
int32_t emb_direct_status(emb_direct_t* d, int32_t* timed_out);
int32_t emb_direct_destroy(emb_direct_t* d);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* EMBODIED_HIP_H_ */
