/*
 * ultrare_hip.h -- C ABI of libultrare_hip.so, the MI355X (gfx950) engine behind
 * UltraRE's SISA hot path.
 *
 * The reference has no FFI: its boundary is the Python operator surface
 * (SURVEY.md 8b).  This header is the boundary *beneath* ultrare_amd's mirror of
 * that surface -- what a maintainer of the reference would bind with ctypes in
 * place of the torch calls cited on each entry point.
 *
 * Conventions
 *   - plain C: pointers, sizes, no torch / C++ types.
 *   - every `dev_*` / device pointer is caller-owned HIP device memory (the Python
 *     host passes torch tensors' data_ptr()); the library never frees it.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     work is enqueued on it; nothing here synchronises unless it says so.
 *   - return value: 0 = success, otherwise a hipError_t (or -1 for argument
 *     errors); ure_last_error() returns a thread-local message.
 *   - fp32 tables are row-major [rows][d]; d is a power of two, 4 <= d <= 256
 *     (the host pads other widths with zero columns, which stay zero).
 *   - citations are file:line under the reference repository.
 */
#ifndef ULTRARE_HIP_H
#define ULTRARE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URE_ABI_VERSION 9
#define URE_MAX_MODELS_PER_CALL 32
#define URE_SCORE_PARTIALS 2048       /* length of ure_score's sse buffer */

int ure_abi_version(void);
/* Hash of the sources (ultrare_amd/csrc, this header, compiler flags) the library was built from, as
 * computed by ultrare_amd/build.py; the host rebuilds a library whose hash is not the tree's, and
 * profiles/ records it so that a counter file is never quoted against other code. */
const char *ure_source_hash(void);
const char *ure_last_error(void);
/* Number of compute units, wavefront size and gcnArchName of device `dev`. */
int ure_device_info(int dev, int *n_cu, int *wave_size, char *arch, int arch_len);

/* ---------------------------------------------------------------------------
 * One SISA shard = one MF model trained in isolation (sisa.py:33-36).
 *
 * The shard's interactions are held twice, grouped by user and grouped by item
 * (built once by the host, stable in file order), so that whoever owns a
 * destination row can sum that row's gradient in registers and apply the
 * optimizer to it immediately -- no atomics, no gradient tables.
 * `ent_src` maps a slot back to its interaction's file-order index; once per epoch the
 * engine inverts the epoch's permutation into `file_tag` (read.py:133: batch s =
 * perm[s*B : (s+1)*B]) and gathers the slot-ordered `ent_tag` from it.
 * ------------------------------------------------------------------------- */
typedef struct ure_shard {
    /* Interactions.  Every destination row (user u -> row id u, item i -> row id
     * n_user + i) owns a segment of slots; the segments of all rows sit in ONE slot
     * array in schedule order (heaviest row first), each starting on a multiple of 8
     * slots and padded to a multiple of 8 with slots that never match a batch.      */
    const int32_t *ent_oid;  /* [n_slots] opposite id: item id in a user row, user id in an item row */
    const float   *ent_r;    /* [n_slots] rating / max_rating (read.py:66)                */
    uint16_t      *ent_tag;  /* [2][n_slots] batch number of the slot, double-buffered by epoch
                              * parity (written by the engine one epoch ahead; the caller
                              * initialises both halves to 0xFFFF)                         */
    const int32_t *ent_src;  /* [n_slots] file-order index of the slot's interaction; -1 in
                              * padding slots                                             */
    uint16_t      *file_tag; /* [N] scratch: batch number of interaction j = inverse of the
                              * permutation of the epoch being prepared                    */
    uint32_t      *inv_stage;/* [N] scratch of the inverse's radix partition               */
    int32_t       *inv_off;  /* [R][R+1] scratch, R = ceil(N / 2048) (unused when R > 1024) */
    /* Row schedule, heaviest first: {row id, first slot, end slot (padded), nnz}.
     * [0, n_active) rows with interactions in this shard | [n_active, rows) rows with none:
     * they only decay.                                                                */
    const int32_t *sched;    /* [n_user + n_item][4]                                      */
    /* Work units of the step kernel (ure_host_build_units, depends on d): one lane group
     * (d/4 lanes up to d = 32, d/8 beyond) walks one unit = a contiguous piece of one row's
     * segment, {row id or -1, first slot, end slot, leader | count << 16 | multi << 30}; a
     * workgroup takes 256 / lanes consecutive units, a row's units never straddle workgroups,
     * `leader` is the index inside the workgroup of the row's first unit, `count` its number of
     * units, `multi` is set on every unit of a workgroup that holds a row of several units.  */
    const int32_t *units;    /* [n_units][4]                                              */
    int32_t        n_units;  /* multiple of 256 / lanes                                   */
    int32_t        n_active;
    int64_t        n_slots;
    /* model state (utils.py:31-40, scratch.py:64-69) */
    float *U[2];            /* [n_user][d] ping-pong: step t reads [t&1], writes [(t+1)&1] */
    float *V[2];            /* [n_item][d]                                          */
    float *mU;              /* [n_user][d] SGD momentum buffer                      */
    float *mV;              /* [n_item][d]                                          */
    /* Rows [n_active, rows) of the schedule have no interaction in this shard: nobody gathers
     * them and they evolve linearly (weight decay + momentum only, scratch.py:65-68).  With
     * lazy_rows != 0 the step kernel skips them and ure_job_materialize() writes their closed
     * form a_T * w0, b_T * w0 when the tables are read; with 0 they are streamed every step
     * exactly like the reference does.                                              */
    const float *U0;        /* [n_user][d] initial tables (kept for the closed form)   */
    const float *V0;        /* [n_item][d]                                          */
    const float *lr_host;   /* [epochs] HOST copy of lr (read by ure_job_create only) */
    int32_t lazy_rows;
    /* Optional end-of-epoch snapshots (NULL = none): after the last step of epoch e the shard's
     * complete tables are copied to snapU[e] / snapV[e]; snap_a[e] = the closed-form weight scalar
     * of the lazy rows after epoch e (required with lazy_rows).  Lets a caller rebuild the
     * reference's per-epoch test logs (scratch.py:83-97) after training shards side by side. */
    float       *snapU;     /* [epochs][n_user][d]                                   */
    float       *snapV;     /* [epochs][n_item][d]                                   */
    const float *snap_a;    /* [epochs] device                                       */
    /* The same in compact form (instead of snapU / snapV; needs lazy_rows): only the n_active rows that have
     * interactions in the shard are stored, in schedule order -- snap[e][idx] = the row sched[idx] after epoch e.
     * Every other row of the tables is a_e * w0 by construction and is rebuilt where it is read
     * (ure_eval_series_compact).  At BASELINE.json configs[3] a shard has 60.8 k active rows of 222 k.   */
    float       *snap;      /* [epochs][n_active][d]                                 */
    const int32_t *row_slot;/* [n_user + n_item] row id -> its index in the schedule when it is one of the n_active rows, -1
                             * otherwise (ure_host_build_layouts writes it).  With it (and without touch_mode) the step kernel
                             * writes a row's compact snapshot itself in the last step of an epoch -- every active row is
                             * rewritten in every step -- and no snapshot launch is needed; NULL: a copy kernel per epoch end. */
    /* per-epoch inputs / outputs */
    const int32_t *perm;    /* [epochs][N] the epoch permutations (RandomSampler); may be NULL when file_tags is given */
    const float   *lr;      /* [epochs] learning rate of each epoch (StepLR)        */
    float         *sse;     /* [epochs][n_user] per-user sum over the epoch of (pred - r)^2;
                             * the epoch's training loss is the sum over users       */
    int32_t N, n_user, n_item, d;
    int32_t batch;          /* B (config.py:26)                                     */
    int32_t epochs;
    float   lam, mu;        /* weight decay, momentum (config.py:20,29)             */
    /* Touch mode (needs lazy_rows, at most 32000 steps per epoch, and the same lr / lam / mu schedule for
     * every shard of the job): a step visits only the rows it trains; rows are kept valid for their NEXT
     * own step and advanced over the steps in between by the optimizer's closed form (csrc/mf_touch.h;
     * epochs longer than 64 steps are worked off in windows of 64).
     * For jobs whose tables do not fit the caches (BASELINE.json configs[3], full MF at 25 M rows).  The
     * tables can then be read (ure_job_materialize, snapshots) at the shard's epoch boundaries only.   */
    int32_t touch_mode;     /* 0 off | 1 as above | 3 "indexed" (ABI 6; at most 1008 steps per epoch; csrc/mf_index.h): at every epoch start the
                             * epoch's slots are sorted by step, and a step launches over exactly the (row, step) runs it trains instead
                             * of testing every work unit's masks -- for epochs of hundreds of steps (full MF at 25 M rows: 750).  Tables
                             * readable at epoch boundaries, as mode 1; `units` is not read (n_units = 0), n_multi / n_split below.
                             * | 2 "masks one epoch ahead" (at most 63 steps per epoch, compact snapshots only): the
                             * batch tags are prepared two epochs ahead (ent_tag then holds THREE buffers, [3][n_slots]), a row's owner
                             * advances it across the epoch boundary at its last own step, and the dense pass at every epoch start of
                             * mode 1 is gone; the tables are readable (ure_job_materialize) only once training has finished, epoch
                             * ends through `snap`.  Same results as mode 1, bit for bit.                                        */
    /* touch mode: `units` covers exactly the first n_multi rows of the schedule (the longest); the rows
     * [n_multi, n_active) are worked off row by row from a per-step compaction of the rows that have
     * interactions in the step.  Epochs of at most 64 steps: n_multi = the rows longer than one scan pass
     * (8 * lanes slots).  Longer epochs: a row of up to 64 passes may stay on the row side -- the pass
     * masks let its lane group skip the passes without a slot of the step.                          */
    int32_t n_multi;
    /* Alternative to `perm` (ABI 5): [epochs][N] the optimizer step of the epoch in which every interaction trains,
     * file_tags[e][perm_e[b]] = b / batch -- what the device derives from a permutation with a two-phase radix partition
     * (csrc/tag_prep.h), computed by the host while it expands the permutation (ure_host_randperm_tags): HALF the bytes
     * of the permutations on their way to the device (the 180 MB of a 5-shard, 50-epoch ml-1m request were as long on
     * PCIe as its training on the GPU), and the partition phases fall away.  NULL: tags are derived from `perm`.       */
    const uint16_t *file_tags;
    /* touch_mode 3: the first n_multi rows of the schedule ("heavy": >= 16 slots per step on average) get a workgroup per step
     * each, the first n_split <= n_multi of them ("split": >= 384) one per 256 slots of the step -- their partial sums are
     * added in a fixed order by a second launch.  n_multi <= 256.  Any values are correct; they only place the work.     */
    int32_t n_split;
} ure_shard_t;

typedef struct ure_job ure_job_t;   /* a set of shards trained side by side */

/* Copies the n descriptors to the device (small hipMalloc owned by the job). */
int ure_job_create(const ure_shard_t *shards, int n_shards, ure_job_t **out);

/* ABI 7.  The start tables of a job's shards (utils.py:31-40: MF.init_weight) from where the host's draws were uploaded into the job's
 * padded tables, all shards in ONE launch: table i is src[i] [rows[i]][k] (dense) -> dst[i] [rows[i]][d] and, where dst2 and dst2[i]
 * are not NULL, dst2[i] (the copy ure_shard.U0 / V0 keep for the closed form).  Columns [k, d) are not written.  (csrc/job_io.hip) */
int ure_copy_rows_batch(int32_t n, const float *const *src, float *const *dst, float *const *dst2, const int64_t *rows, int32_t k, int32_t d,
                        void *stream);

/* ABI 7.  The per-epoch training loss of a job's shards (scratch.py:72-77 accumulates it batch by batch): ure_shard.sse holds the squared
 * errors per epoch and USER ([epochs][n_user] float32, each owner adds its batches' share); out[i][e] = their sum over the users of
 * shard i in double, always in the same order, all shards in one launch.  sse[i], out: device memory.  (csrc/job_io.hip)             */
int ure_epoch_sse_batch(int32_t n, const float *const *sse, const int64_t *n_user, int32_t epochs, double *out, void *stream);
int ure_job_destroy(ure_job_t *job);
/* Number of optimizer steps shard `s` needs in total = epochs * ceil(N/B), and
 * the maximum over the job's shards (the number of ticks to run). */
int64_t ure_job_shard_steps(const ure_job_t *job, int s);
int64_t ure_job_ticks(const ure_job_t *job);

/* Replaces baseTrain's loop (utils.py:58-91) + opt.step() (scratch.py:64-69) for
 * every shard of the job: tick t performs optimizer step t of each shard that
 * still has one (a shard's step t belongs to epoch t / ceil(N/B)); at the first
 * step of an epoch the shard's batch numbers are re-derived from perm[epoch].
 * Ticks [tick0, tick1) are enqueued on `stream`; tick0 must continue where the
 * previous call stopped (0 for a fresh model).  After the last tick the trained
 * tables are U[ticks_done & 1] / V[ticks_done & 1] of each shard, where
 * ticks_done = min(tick1, shard steps). */
int ure_job_train(ure_job_t *job, int64_t tick0, int64_t tick1, void *stream);
/* Brings the rows that lazy_rows skips up to date in the current tables (and momentum) of
 * every shard, for `ticks_done` = the number of ticks trained so far.  Call before reading
 * the tables; training may continue afterwards.  No-op for shards with lazy_rows == 0. */
int ure_job_materialize(ure_job_t *job, int64_t ticks_done, void *stream);
/* Touch mode accounting (synchronises): pairs[s] = the number of (row, step) pairs of shard s's current window of
 * (at most 64) steps in which a row is trained -- the rows the step kernel actually reads and rewrites, summed over
 * the window's steps (from the window's row masks) -- and window_steps[s] = the number of steps of that window;
 * pairs = -1 for a job that is not in touch mode.                                                                */
int ure_job_touch_rows(ure_job_t *job, int64_t *pairs, int64_t *window_steps);
/* Test aid, touch_mode 3 (synchronises): copies one array of shard `shard`'s slot index of its current epoch to HOST memory `out`
 * (capacity bytes; out == NULL: only *bytes is set).  which: 0 step_begin u32 [steps + 1] (first sorted slot of every step, then
 * their number) | 1 step_item u32 [steps + 1] | 2 items int32 [n][4] {row id | buffer << 31, first sorted slot, end, steps until
 * the row's next own step | class << 16} | 3 sorted slots u32 [n][4] {opposite id | buffer << 31, rating bits, row id,
 * step | class << 16} | 4 W u64 [words][rows] | 5 heavy_cnt u32 [steps] | 6 heavy_cum u32 [steps][257].           */
int ure_job_index_read(ure_job_t *job, int shard, int which, void *out, int64_t capacity, int64_t *bytes);
/* Measurement aid (bench.py's roofline leg): the same ticks, each kernel launch
 * bracketed by a pair of HIP events on `stream`; synchronises the stream and returns
 * the summed durations and launch counts of the step kernel and of the per-epoch
 * batch-tag kernel. */
int ure_job_train_profiled(ure_job_t *job, int64_t tick0, int64_t tick1, void *stream, double *step_ms,
                           int64_t *n_step, double *assign_ms, int64_t *n_assign);

/* read.py:133 / torch RandomSampler: out[t] = torch.randperm(n, generator seeded with
 * seeds[t]) for t < n_perms, as int32, computed on `n_threads` host threads (0 = all).
 * HOST memory; bit-identical to torch's CPU randperm for n < 2^32/20. */
int ure_host_randperm(const int64_t *seeds, int n_perms, int64_t n, int32_t *out, int n_threads);
/* The same permutations as batch tags (struct ure_shard: file_tags): tags[t][perm_t[b]] = b / batch, uint16 (needs
 * ceil(n / batch) <= 65535 steps per epoch).  HOST memory.                                                       */
int ure_host_randperm_tags(const int64_t *seeds, int n_perms, int64_t n, int32_t batch, uint16_t *tags, int n_threads);
/* ABI 7.  The same tags made ON THE DEVICE (csrc/perm_tags.hip): MT19937's outputs by its three parallel phases per block, the shuffle
 * with deterministic reservations (the sequential loop's permutation, whatever the timing), tags[f] = inverse[f] / batch.  One entry
 * per permutation -- the shards and epochs of a request in one launch --: its seed (as ure_host_randperm_tags takes it), where its n
 * tags go (device memory), n <= 2^20 rows, the batch size (ceil(n / batch) <= 65535).  perms: DEVICE memory; scratch: device memory of
 * ure_device_randperm_tags_scratch(largest n, groups) 32-bit words; `groups` workgroups of 1,024 lanes make one permutation each at
 * a time.  Word 2 * align64(n_max) * groups + g of the scratch is set to 0xdead if workgroup g gave up (it cannot; the caller clears
 * the words when it makes the scratch).                                                                                          */
typedef struct ure_perm {
    int64_t   seed;
    uint16_t *tags;
    int32_t   n;
    int32_t   batch;
} ure_perm_t;
int64_t ure_device_randperm_tags_scratch(int64_t n_max, int32_t groups);
int ure_device_randperm_tags(const ure_perm_t *perms, int32_t n_perms, int64_t n_max, uint32_t *scratch, int64_t scratch_words, int32_t groups,
                             void *stream);
/* ABI 9.  The same tags by MANY workgroups per permutation and for up to 2^28 rows (csrc/perm_chain.hip): the shuffle's result in closed
 * form -- the swaps grouped by the position they target, a row's value = the smallest swap that targets it or a short chase along
 * "largest member" links -- in six stream-ordered launches (MT19937 words; targets + counts per range of 1,024 targets; the swaps bucketed by
 * range; per (permutation, range) the target lists in LDS; resolve + divide).  No 2^20-row limit (read.py:127-133 for config.py:182-188's
 * full-MF run: 22.5 M rows at the 25 M shape) and a fraction of perm_tags.hip's latency for a request's first epochs.  perms: DEVICE
 * memory, as above; scratch: device memory of ure_device_shuffle_tags_scratch(largest n, n_perms) words (20 bytes per row and permutation
 * of the call), reusable by the next call on the same stream.  range_log2: 0 (the library chooses: 10 up to 2^24 rows, 11 up to 2^25, 12 up to 2^26, else 14), 10, 11, 12 or 14.
 * Word ure_device_shuffle_tags_flag(n_max, n_perms) of the scratch is set to 0xdead if the resolve pass met a link it cannot follow (it
 * cannot; such a row gets tag 0xFFFF, which matches no batch; the caller clears the word when it makes the scratch).                  */
int64_t ure_device_shuffle_tags_scratch(int64_t n_max, int32_t n_perms);
int64_t ure_device_shuffle_tags_flag(int64_t n_max, int32_t n_perms);
int ure_device_shuffle_tags(const ure_perm_t *perms, int32_t n_perms, int64_t n_max, uint32_t *scratch, int64_t scratch_words, int32_t range_log2,
                            void *stream);
/* Moves a torch CPU generator state (the bytes of torch.get_rng_state(): u64 seed, i32 left, i32 seeded,
 * u64 next, u64 state[624], ...) past `n_draws` 32-bit MT19937 outputs without producing them: the model
 * init fills the reference discards (utils.py:31-40: the nn.Embedding constructors' fills) and, in a
 * multi-rank run, the draws of the shards other ranks own.  In place.                       */
int ure_host_mt_advance(uint8_t *state, int64_t n_bytes, int64_t n_draws);
/* ABI 8.  MT19937 jump-ahead (csrc/mt_jump.cpp), what ure_host_mt_advance uses beyond 4,096 blocks: st [624] = the words of a
 * generator block (torch.get_rng_state()'s state[] narrowed to 32 bits) -> the block `blocks` regenerations later, in place, by
 * x^(624 blocks - 1) modulo the generator's characteristic polynomial applied to the raw word sequence (~0.1 ms for any distance;
 * the polynomial of a distance is memoised).  ure_host_mt_jump_support: the degrees of that polynomial's nonzero terms, ascending
 * (< 19,937 of them; capacity 0 only counts) -- what the device-side jump of ure_device_mf_init convolves with.
 * ure_host_mt_charpoly: the 135 exponents of the characteristic polynomial itself, ascending (tests recompute them). */
int ure_host_mt_jump_blocks(uint32_t *st, int64_t blocks);
int ure_host_mt_jump_support(int64_t blocks, uint16_t *support, int32_t capacity, int32_t *n_support);
int ure_host_mt_charpoly(uint16_t *exponents, int32_t capacity);
/* ABI 7.  n int64 values as `tensor.random_()` draws them (the per-epoch seeds of scratch.py:78-97: two 32-bit outputs each, the first
 * the high word, bit 63 cleared) from a COPY of a torch CPU generator state moved past skip_draws outputs.  HOST memory.          */
int ure_host_draw_int64(const uint8_t *state, int64_t n_bytes, int64_t skip_draws, int64_t n, int64_t *out);
/* ABI 7.  MF.init_weight's kept fills (utils.py:31-40) from a torch CPU generator state: the state is moved past `skip_draws` outputs
 * (the nn.Embedding constructors' discarded fills), then U0 [nu] and V0 [nv] (each 0 or >= 16 elements) are filled as
 * `tensor.normal_()` fills a contiguous float32 tensor on an AVX2-capable host: the uniforms in generator order, then Box-Muller
 * 16 at a time through the installed PyTorch's own avx_mathfun kernels (csrc/host_normal_avx2.cpp), the 16-blocks on n_threads
 * threads; the state ends where torch's would.  -4: those kernels are not in this build / not supported by this CPU.  The Python
 * side checks the function against torch once per process and keeps torch's fill when a bit differs.  HOST memory.           */
int ure_host_mf_init(uint8_t *state, int64_t n_bytes, int64_t skip_draws, float *U0, int64_t nu, float *V0, int64_t nv, int n_threads);
/* The same for all shards of a request in one call: shard s from states[s] (in / out), skip_draws[s], into U0[s] [nu], V0[s] [nv]; the
 * shards side by side on n_threads threads.                                                                                   */
int ure_host_mf_init_batch(int32_t n_shards, uint8_t *const *states, int64_t n_bytes, const int64_t *skip_draws, float *const *U0, int64_t nu,
                           float *const *V0, int64_t nv, int n_threads);
/* ABI 8.  The same fills made ON THE DEVICE (csrc/mf_init.hip): U0[s] [nu] and V0[s] [nv] are DEVICE memory, everything else as
 * ure_host_mf_init_batch -- states[s] (host, in / out) is moved past skip_draws[s] outputs and ends behind the two fills.  The host only
 * positions generators (ure_host_mt_jump_blocks); the device cuts every shard's draws into segments of 1,024 generator blocks, reaches
 * their start blocks by a doubling tree of jumps, regenerates and tempers the outputs in LDS and applies csrc/normal_math.h's Box-Muller
 * -- bit for bit `tensor.normal_()` on an AVX2 host (the Python side checks that once per process and keeps the host fill otherwise).
 * scratch: device memory of ure_device_mf_init_scratch(n_shards, nu, nv) 32-bit words, alive until the work on `stream` is done. */
int64_t ure_device_mf_init_scratch(int32_t n_shards, int64_t nu, int64_t nv);
int ure_device_mf_init(int32_t n_shards, uint8_t *const *states, int64_t n_bytes, const int64_t *skip_draws, float *const *U0, int64_t nu,
                       float *const *V0, int64_t nv, uint32_t *scratch, int64_t scratch_words, int n_threads, void *stream);
/* The Box-Muller half alone: data [16 n_blocks] uniforms -> normals in place (ATen's normal_fill_16_AVX2).  0, or -4 as above. */
int ure_host_normal_blocks(float *data, int64_t n_blocks, float mean, float std_);
/* ABI 8.  The same through the scalar restatement of that arithmetic which the device kernels run (csrc/normal_math.h), mean 0, std 1;
 * variant 0 is the arithmetic of record, 1-3 the other readings of its two ambiguous mul + add pairs (tests only). */
int ure_host_normal_blocks_scalar(float *data, int64_t n_blocks, int32_t variant);

/* ---------------------------------------------------------------------------
 * Host-side ingest (HOST memory throughout; linear time, `n_threads` = 0 means all cores)
 * ------------------------------------------------------------------------- */
/* read.py:37 pd.read_csv(dir, header=None, sep=','): rows `uid,iid,rating[,...]`.  The three
 * arrays are malloc'ed by the library; release each with ure_host_free(). */
int ure_host_read_csv(const char *path, int32_t **uid, int32_t **iid, double **rating, int64_t *n_rows, int n_threads);
void ure_host_free(void *p);
/* read.py:52-70: shard s receives, in file order, the rows whose user u has
 * shard_of_user[u] == s (-1: deleted user, dropped); rating / max_rating as float32 (out_rating)
 * and / or as the float64 quotient readRating returns (out_rating64); either may be NULL.
 * counts [n_shards] is always filled; with out_uid == NULL nothing else is written (size query),
 * otherwise the shards are written back to back (shard s starts at sum(counts[:s])). */
int ure_host_partition(const int32_t *uid, const int32_t *iid, const double *rating, int64_t n, const int32_t *shard_of_user,
                       int32_t n_user, int32_t n_shards, double max_rating, int64_t *counts, int32_t *out_uid,
                       int32_t *out_iid, float *out_rating, double *out_rating64);
/* The same partition written as readRating returns it (read.py:9-70): shard s is the float64 block [3][counts[s]] at
 * out + 3 * sum(counts[:s]) -- its uid row, its iid row, its rating / max_rating row.  out == NULL: counting pass only. */
int ure_host_partition64(const int32_t *uid, const int32_t *iid, const double *rating, int64_t n, const int32_t *shard_of_user,
                         int32_t n_user, int32_t n_shards, double max_rating, int64_t *counts, double *out);
/* Builds the slot layout of struct ure_shard from a shard's triples: ent_oid / ent_r / ent_src
 * (capacity 2 n + 8 (n_user + n_item) slots, *n_slots receives the used count), sched
 * [n_user + n_item][4], the number of active rows, and optionally u_pos / i_pos [n] (slot of each
 * interaction in its user's / item's segment). */
int ure_host_build_layout(const int32_t *uid, const int32_t *iid, const float *rating, int64_t n, int32_t n_user, int32_t n_item,
                          int32_t *ent_oid, float *ent_r, int32_t *ent_src, int32_t *sched,
                          int64_t *n_slots, int32_t *n_active, int32_t *u_pos, int32_t *i_pos);
/* The same for all shards of a call at once, side by side on `n_threads` host threads (0 = all), from the triples as
 * RatingData holds them (read.py:108-124: int64 ids, float64 ratings; a rating is cast to float32 once, read.py:124).
 * Shard s is written PACKED into region[s] -- int32 words, at least 6 n[s] + 29 (n_user + n_item) + 24 of them:
 *   ent_oid [k] | ent_r [k] | ent_src [k] | sched [n_user + n_item][4] | row_slot [n_user + n_item],   k = n_slots[s]
 * row_slot[r] = the row's index in the schedule when it has interactions in the shard (one of the n_active rows a compact
 * snapshot stores), -1 otherwise.  A region can be pinned host memory: it is what goes to the device, in one copy. */
int ure_host_build_layouts(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                           int32_t n_user, int32_t n_item, int32_t *const *region, int64_t *n_slots, int32_t *n_active, int n_threads);
/* The same, and behind every layout the work units of table width units_d (ure_host_build_units with unit_passes 1) of its schedule:
 *   region[s] = layout (3 k + 5 rows words) | pad to a multiple of 8 words | units [n_units[s]][4]
 * so that one copy takes a shard's layout AND its units to the device (as two native calls with Python between them the units
 * were 0.6-1.9 ms of a request's critical path).  region_words[s] = the words region[s] can take; n_units[s] = -1 when the
 * units did not fit behind the layout (ask ure_host_build_units then). */
int ure_host_build_layouts_units(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                                 int32_t n_user, int32_t n_item, int32_t *const *region, const int64_t *region_words, int64_t *n_slots,
                                 int32_t *n_active, int32_t units_d, int64_t *n_units, int n_threads);
/* ABI 9.  The same on a thread of the library's own: _start returns at once with a handle (0: bad arguments / no thread could be started),
 * _wait joins and returns what ure_host_build_layouts_units returned (its message in this thread's ure_last_error()).  dev_region != NULL:
 * every shard's region (layout + units, the words used) is copied to dev_region[s] -- DEVICE memory of `device` -- on `stream` as soon as
 * the shard is built, while the others are still being built; _wait returns when every copy has been queued.  Every argument of _start
 * must stay alive and unchanged until _wait has returned; every handle must be waited for exactly once. */
int64_t ure_host_build_layouts_units_start(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating,
                                           const int64_t *n, int32_t n_user, int32_t n_item, int32_t *const *region, const int64_t *region_words,
                                           int64_t *n_slots, int32_t *n_active, int32_t units_d, int64_t *n_units, int n_threads,
                                           int32_t *const *dev_region, int32_t device, void *stream);
int ure_host_build_layouts_units_wait(int64_t handle);
/* Cuts the active rows of a schedule into the work units of struct ure_shard for row width d and
 * packs them into workgroups.  unit_passes = scan passes of one lane group per unit: 1 everywhere, except touch
 * mode with epochs of several windows (more than 64 steps), where a unit of several passes skips the passes
 * without a slot of the step by their masks (8: full MF at the 25 M shape 106 -> 63 us per launch).
 * units == NULL: only *n_units is written (size query); otherwise `capacity` units may be written. */
int ure_host_build_units(const int32_t *sched, int32_t n_active, int32_t d, int32_t unit_passes, int32_t *units, int64_t capacity, int64_t *n_units);

/* ---------------------------------------------------------------------------
 * Evaluation (baseTest, utils.py:115-187)
 * ------------------------------------------------------------------------- */
/* utils.py:140-145: pred[j] = (sum_m U_m[uid[j]] . V_m[iid[j]]) / n_models_total.
 * Up to URE_MAX_MODELS_PER_CALL tables per call; for larger ensembles call
 * repeatedly with `first`/`last` marking the first and last chunk (pred holds the
 * running sum in between).  When `sse` is non-NULL the last chunk also writes partial sums of
 * (pred[j] - rating[j])^2 to sse[0 .. URE_SCORE_PARTIALS) (utils.py:148; double, device memory);
 * their total is the loss (summed in a fixed order by ure_eval_reduce, or by the caller). */
int ure_score(const float *const *U_tables, const float *const *V_tables, int n_models,
              int n_models_total, int first, int last,
              const int32_t *uid, const int32_t *iid, const float *rating, int64_t n, int d,
              float *pred, double *sse, void *stream);

/* utils.py:165-184 for users whose entries are contiguous: user t owns entries
 * [off[t], off[t+1]).  For each user: top-10 by prediction and by rating (ties:
 * higher position first = stable argsort reversed), hits[t] = #(rating[top_pred]
 * >= 4/5), ndcg[t] = the reference's positional NDCG@10 (utils.py:190-210).
 * `log2_tab` [10] (device, float64) = log2(2..10) followed by the ideal DCG computeDCG(ones(10)),
 * both computed by the host with numpy so that every division matches the reference bit for bit. */
/* Users [0, n_wide) get one wavefront each (segments of any length); users [n_wide, n_wide + n_half) MUST have at most 32
 * entries and share wavefronts two by two (ABI 6); the others MUST have at most 16 and share them four by four
 * (n_wide = n_users: every user a wavefront).  top_rating (optional, device
 * [n_users][10]): the top-10 positions of the RATINGS, which do not depend on the model -- ure_eval_rank_ratings
 * computes them once per test set; NULL = ranked inside the call.  With top_rating the call is two launches (rank, then
 * metrics by one thread per user); between them `hits` / `ndcg` hold the packed positions of the ranking.          */
int ure_eval_users(const int32_t *off, int32_t n_users, const float *pred, const float *rating,
                   const double *log2_tab, int32_t *hits, double *ndcg, const int32_t *top_rating, int32_t n_wide, int32_t n_half, void *stream);
int ure_eval_rank_ratings(const int32_t *off, int32_t n_users, const float *rating, int32_t *top_rating, void *stream);

/* utils.py:163,183-184: out3 (device, 3 doubles) = { sqrt(sum(sse[0..URE_SCORE_PARTIALS)) / n_rows), mean(ndcg), mean(hits / 10) }
 * from the outputs of ure_score / ure_eval_users, reduced on the device in a fixed order: lets a
 * caller queue many evaluations without synchronising and read all results at the end. */
int ure_eval_reduce(const int32_t *hits, const double *ndcg, int32_t n_users, const double *sse, int64_t n_rows, double *out3,
                    void *stream);

/* scratch.py:83-97 for all epochs of a shard at once: a SERIES of n_series evaluations on one test
 * set whose ensembles are the n_fixed fixed models (in list order; the models trained before the
 * shard) followed by (U_series + e * stride_u, V_series + e * stride_v), e.g. the end-of-epoch
 * snapshots of struct ure_shard.  Four launches in all; every member's result is identical to
 * ure_score + ure_eval_users + ure_eval_reduce on its own model list.  Scratch (device): base [n]
 * (unused when n_fixed == 0), pred [n_series][n], sse [n_series][URE_SCORE_PARTIALS], hits and ndcg
 * [n_series][n_users]; out [n_series][3] = (rmse, ndcg, hr) of each member. */
int ure_eval_series(const float *const *U_fixed, const float *const *V_fixed, int n_fixed, const float *U_series,
                    const float *V_series, int64_t stride_u, int64_t stride_v, int n_series, const int32_t *uid, const int32_t *iid,
                    const float *rating, int64_t n, int d, const int32_t *off, int32_t n_users, const double *log2_tab, float *base,
                    float *pred, double *sse, int32_t *hits, double *ndcg, double *out, const int32_t *top_rating, int32_t n_wide,
                    int32_t n_half, void *stream);

/* ABI 8.  In ure_eval_series / ure_eval_series_compact / ure_eval_series_own, U_fixed == V_fixed == NULL with n_fixed > 0 means: `base`
 * already holds the running sum over the n_fixed fixed models (the caller made it: one ure_score per distinct model with n_models = 1,
 * first = 1, last = 0, then ure_sum_vectors in the ensemble's order -- the additions ure_score makes over a list, so the series'
 * numbers are the same to the last bit).  A request of S shards scores every model once instead of once per later shard.
 * ure_sum_vectors: out[j] = ((0 + v_0[j]) + v_1[j]) + ... over n_vectors device vectors of n floats.                       */
int ure_sum_vectors(const float *const *vectors, int n_vectors, int64_t n, float *out, void *stream);

/* The same series on COMPACT end-of-epoch snapshots (struct ure_shard: snap): member e's own model is
 * row r -> row_slot[r] >= 0 ? snap[e * stride + row_slot[r] * d ..] : snap_a[e] * (U0 | V0)[r], with row ids
 * r = u for users and n_user_rows + i for items (row_slot [n_user_rows + n_item_rows] int32, device: the row's index
 * in the schedule when it is one of the n_active stored rows, -1 otherwise).  Results are identical to
 * ure_eval_series on full snapshots: the closed form is evaluated with the same fp32 product.              */
int ure_eval_series_compact(const float *const *U_fixed, const float *const *V_fixed, int n_fixed, const float *snap, int64_t stride,
                            const int32_t *row_slot, const float *U0, const float *V0, const float *snap_a, int32_t n_user_rows,
                            int n_series, const int32_t *uid, const int32_t *iid, const float *rating, int64_t n, int d,
                            const int32_t *off, int32_t n_users, const double *log2_tab, float *base, float *pred, double *sse,
                            int32_t *hits, double *ndcg, double *out, const int32_t *top_rating, int32_t n_wide, int32_t n_half, void *stream);

/* The same series in two halves, so that the first can run beside training.  ure_score_own_compact: own[e][j] = the score of
 * pair j under the shard's own model after epoch e alone, for n_series consecutive epochs whose compact snapshots exist (snap,
 * snap_a and own point at the first of them) -- queued by the host on a second stream as training proceeds.
 * ure_eval_series_own: the rest, once the fixed models are final -- pred = (sum of the fixed models' scores + own) / (n_fixed + 1)
 * with the additions in ure_eval_series' order, ranking, metrics.  Results are identical to ure_eval_series_compact.    */
int ure_score_own_compact(const float *snap, int64_t stride, const int32_t *row_slot, const float *U0, const float *V0, const float *snap_a,
                          int32_t n_user_rows, int n_series, const int32_t *uid, const int32_t *iid, int64_t n, int d, float *own, void *stream);
int ure_eval_series_own(const float *const *U_fixed, const float *const *V_fixed, int n_fixed, const float *own, int n_series, const int32_t *uid,
                        const int32_t *iid, const float *rating, int64_t n, int d, const int32_t *off, int32_t n_users, const double *log2_tab,
                        float *base, float *pred, double *sse, int32_t *hits, double *ndcg, double *out, const int32_t *top_rating, int32_t n_wide,
                        int32_t n_half, void *stream);

/* scratch.py:83-97 tests every epoch's ensemble on the shard's own test set and on the total test set, which config.py:144-148 builds
 * as the shards' test sets side by side: the shard's set is the total set's rows of the shard's users.  After a series on the TOTAL
 * set (ure_eval_series*: pred [n_series][pred_stride], hits / ndcg [n_series][n_users] still hold its per-pair predictions and per-user
 * metrics) this reduces the same three numbers over a subset of its users: sub_users [n_sub] = their indices in the total set's
 * user order, sub_pairs [n_pairs] = the indices of their pairs in the total set's pair order; out [n_series][3] = (sqrt(sum over
 * those pairs of (pred - rating)^2 / n_pairs), mean ndcg, mean hits / 10).  The caller establishes that the subset's rows ARE the
 * total set's rows of those users (ultrare_amd.engine.EvalSet.subset_of).                                                       */
int ure_eval_subset(const int32_t *sub_users, int32_t n_sub, const int32_t *sub_pairs, int32_t n_pairs, const float *pred, const float *rating,
                    const int32_t *hits, const double *ndcg, int64_t pred_stride, int32_t n_users, int n_series, double *out, void *stream);

/* sisa.py:55-56,110-111: dst[rows[t]][:] = src[rows[t]][:]. */
int ure_merge_rows(float *dst, const float *src, const int64_t *rows, int64_t n_rows, int d, void *stream);

/* ---------------------------------------------------------------------------
 * OT balanced grouping (utils.py:628-656)
 * ------------------------------------------------------------------------- */
/* utils.py:637: dist[c][i] = sum_j (X[i][j] - C[c][j])^2 in numpy's fp32 pairwise
 * order (bit-exact), any 1 <= d <= 256. */
int ure_ot_cost(const float *X, const float *C, int64_t n, int k, int d, float *dist, void *stream);
/* The same matrix as |x|^2 - 2 x.c + |c|^2 with the contraction on the matrix cores (v_mfma_f32_32x32x2_f32, fp32 in and
 * out).  It rounds differently from utils.py:637, so it is NOT the arithmetic of record: an optional fast path whose labels
 * the host cross-checks against ure_ot_cost's every round (method/utils.py::ot_cluster with URE_OT_MFMA=1).          */
int ure_ot_cost_mfma(const float *X, const float *C, int64_t n, int k, int d, float *dist, void *stream);
/* utils.py:648: C[c] = mean of the rows with label c, fp32 sequential in ascending
 * row id then one division by the count (bit-exact).  counts [k] receives sizes. */
int ure_ot_centroids(const float *X, const int32_t *label, int64_t n, int k, int d, float *C,
                     int32_t *counts, void *stream);
/* The same means from member lists (device): order[off[c] .. off[c+1]) = the points of cluster c in ascending id,
 * off [k + 1] int64 -- a stable counting sort of the labels.  Same fp32 order, ~n / k rows per thread instead of n. */
int ure_ot_centroids_members(const float *X, const int32_t *order, const int64_t *off, int64_t n, int k, int d, float *C,
                             int32_t *counts, void *stream);
/* utils.py:642-647: exact optimal transport between n points of mass 1/n and k
 * clusters of mass 1/k for cost dist [k][n] (HOST memory, fp32 widened to double
 * exactly), followed by label = argmax of each point's plan row (first maximum).
 * Exact integer min-cost-flow on the k-node cluster graph; runs on the host (the
 * LP is sequential and tiny next to training).  plan_nk (optional, [n][k]) receives
 * the plan in units of 1/(n*k); total_cost (optional) the objective <G, M>. */
int ure_ot_assign(const float *dist_host, int64_t n, int k, int32_t *label_host,
                  int32_t *plan_nk, double *total_cost);
/* The same exact LP from a warm start.  ure_ot_potentials: cluster potentials pi[k] (HOST, in: where to start -- zeros or
 * the previous round's; out: the iterate with the smallest imbalance) by `iters` sign-based steps of dual ascent on the
 * device cost matrix dist [k][n] (k <= 256; otherwise, or with iters == 0, pi = 0 and the solver starts cold);
 * *misplaced (optional) = points that imbalance leaves to move; synchronises `stream`.  ure_ot_assign_warm: every point starts on the cluster of its
 * cheapest reduced cost cost - pi, then successive shortest paths as in ure_ot_assign -- the result is the exact
 * optimum for any pi, only the number of augmentations (optional output) depends on it; pi == NULL or a poor start
 * falls back to ure_ot_assign.                                                                              */
int ure_ot_potentials(const float *dist, int64_t n, int k, int iters, double *pi_host, int64_t *misplaced, void *stream);
int ure_ot_assign_warm(const float *dist_host, int64_t n, int k, const double *pi, int32_t *label_host, int32_t *plan_nk,
                       double *total_cost, int64_t *augmentations);

/* ---------------------------------------------------------------------------
 * Comparison clusterers (utils.py:354-418: k-means / balanced k-means on the user embedding;
 * never called on the reference's CLI path, kept for the OT-vs-k-means comparison of its notebook)
 * ------------------------------------------------------------------------- */
/* utils.py:373-375: dist_nk[i][c] = -2 x_i.c_c + |x_i|^2 + |c_c|^2 in float32 with scipy's csr
 * evaluation order (labels are then identical to the reference's). */
int ure_kmeans_cost(const float *X, const float *C, int64_t n, int k, int d, float *dist_nk, void *stream);
/* utils.py:402-403: C[c] = mean of the rows with label c as scipy's sparse mean computes it
 * (row * float32(1/count), summed in ascending row order); counts [k] optional. */
int ure_kmeans_centroids(const float *X, const int32_t *label, int64_t n, int k, int d, float *C, int32_t *counts, void *stream);
/* utils.py:377-396 (host): capacity <= 0: label = argmin over groups; capacity > 0: balanced fill in
 * ascending order of distance, at most `capacity` users per group.  *inertia = np.sum(dist[arange(n),
 * label]) (numpy float32 pairwise sum), optional. */
int ure_host_kmeans_assign(const float *dist_nk, int64_t n, int32_t k, int64_t capacity, int32_t *label, double *inertia);

#ifdef __cplusplus
}
#endif
#endif /* ULTRARE_HIP_H */
