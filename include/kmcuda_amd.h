/*
 * kmcuda_amd.h -- step-level C ABI of the MI355X engine behind kmeans_cuda()/knn_cuda().
 *
 * The reference keeps these steps private (src/private.h:304-404: kmeans_cuda_setup,
 * kmeans_cuda_lloyd, kmeans_cuda_yy, knn_cuda_calc, cuda_transpose ...) and drives all GPUs
 * from one process with peer copies.  Row-sharded, one-process-per-GPU operation (RCCL
 * all-reduce of the per-iteration centroid deltas between kmamd_move_deltas and
 * kmamd_apply_delta) needs them exported.  Plain pointers and sizes only; every pointer
 * argument is a DEVICE pointer on the engine's GPU unless it says "host".
 * All calls return a KMCUDAResult code (kmcuda.h) and enqueue on the engine's stream.
 */
#ifndef KMCUDA_AMD_H
#define KMCUDA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmamd_engine kmamd_engine;

/* Creates the per-GPU workspace for n_rows local rows (reference: the allocations of
 * kmcuda.cc:423-470 + kmeans_cuda_setup kmeans.cu:750-772).  hip_stream: the stream every step is
 * enqueued on -- pass the framework's current stream to order with its work.  NULL: the engine
 * creates its own non-blocking stream (nothing else may touch its buffers without
 * kmamd_engine_sync).  (hipStream_t)-1: the caller works on the legacy default stream (handle 0,
 * torch's default): the engine creates an own BLOCKING stream, which the default stream orders
 * with implicitly in both directions. */
int kmamd_engine_create(kmamd_engine **out, int device, uint32_t n_rows, uint32_t features,
                        uint32_t clusters, int metric, int fp16x2, void *hip_stream);
void kmamd_engine_destroy(kmamd_engine *e);
void *kmamd_engine_stream(kmamd_engine *e);
int kmamd_engine_sync(kmamd_engine *e);

/* One assignment pass over the local rows (reference: kmeans_assign_lloyd, kmeans.cu:293-364,
 * incl. assignments_prev bookkeeping and the changed counter).  Bit-identical assignments. */
int kmamd_lloyd_assign(kmamd_engine *e, const float *samples, const float *centroids,
                       uint32_t *assignments, uint32_t *assignments_prev);
/* Same pass, reference arithmetic only (no matrix-core filter): the in-library cross-check. */
int kmamd_lloyd_assign_exact(kmamd_engine *e, const float *samples, const float *centroids,
                             uint32_t *assignments, uint32_t *assignments_prev);

/* Which matrix-core scheme the assignment filter runs: 0 = two-stage f16 MFMA (default): a coarse
 * pass with the high halves of the centred operands decides most rows, a contender pass in fp32 the
 * rest; 1 = f32 MFMA (cross-check).  Assignments are bit-identical in both (every stage carries a
 * rigorous bound and the exact kernels decide what is left); env KMCUDA_AMD_FILTER=f32 selects 1 at
 * engine creation. */
int kmamd_set_filter(kmamd_engine *e, int mode);

/* fp16x2 path: the engine's local rows as IEEE halves (n_rows x features halves, row-major, kept
 * alive by the caller).  When set, the f16 matrix-core filter of kmamd_lloyd_assign reads these rows
 * (half the HBM bytes) instead of the fp32 ones; `samples` must still point to the
 * same values widened to fp32 (the exact refine kernels read them).  Assignments are unchanged:
 * bit-identical to the fp32 path on the widened values.  NULL switches back. */
int kmamd_set_half_rows(kmamd_engine *e, const void *rows16);

/* Row cache of the two-stage filter's coarse pass.  on != 0: the caller promises that the rows given
 * to kmamd_lloyd_assign (samples, or the half rows) are the SAME, UNMODIFIED buffer on every call
 * from now on -- what the reference guarantees inside one kmeans_cuda() run, where the samples are
 * transposed once and iterated over (kmcuda.cc:505-508).  The next kmamd_lloyd_assign then stores
 * x - mu (mu = mean of that call's centroids, frozen from then on) as halves in the matrix-core
 * operand order (2 * features bytes per row + 8) and later passes stream that copy: half the HBM
 * bytes, whole 1-KB bursts, no conversion.  Assignments are unchanged (the bound is computed for the
 * mu in use).  Calling it again (on or off) drops the copy; off (the default) converts from the rows
 * on every pass.  If the copy cannot be allocated the engine silently stays uncached.
 * Env KMCUDA_AMD_ROW_CACHE=0 vetoes it (A/B runs). */
int kmamd_set_row_cache(kmamd_engine *e, int on);

/* counters: [0] reassigned rows since the last reset (d_changed_number, kmeans.cu:31),
 * [1] rows the filter handed to the full exact scan, [2] Yinyang passed rows (d_passed_number),
 * [3] rows the filter narrowed to two contenders (pair refine).  read = sync + copy to a host array; reset zeroes [0..3] (or one of them). */
int kmamd_counters_read(kmamd_engine *e, uint32_t *host_out4);
int kmamd_counters_reset(kmamd_engine *e, int which /* -1: all */);
/* Yinyang local filter with the second-best estimate (yinyang_hint.hip; KMCUDA_AMD_YY_HINT=0 turns it
 * off): running totals since the engine was created -- [0] rows it processed, [1] rows it handed to the
 * plain kernel (results are the reference's either way; the ratio is a performance figure), [2..5] the
 * hand-overs by first reason: no estimate, a skipped candidate whose bound does not hold, an evaluated
 * candidate whose bound exceeds the estimate, a final second minimum above the estimate. */
int kmamd_yy_hint_stats(kmamd_engine *e, uint32_t *host_out6);

/* Centroid update, split for the all-reduce (reference: kmeans_adjust, kmeans.cu:366-429):
 * move_deltas: delta[K*D] (fp64) = sum(moved-in rows) - sum(moved-out rows), dcount[K];
 * apply_delta: centroids = normalize(centroids*ccounts + delta), ccounts += dcount. */
int kmamd_move_deltas(kmamd_engine *e, const float *samples, const uint32_t *assignments_prev,
                      const uint32_t *assignments, double *delta, int32_t *dcount);
int kmamd_apply_delta(kmamd_engine *e, const double *delta, const int32_t *dcount,
                      float *centroids, uint32_t *ccounts);
/* The same two steps around ONE collective: buf is kmamd_reduce_len(e) = K*D + K + 4 doubles,
 *   [ delta (K*D) | dcount (K) | counters 0..3 ]
 * (counts and counters are exact in fp64), written by reduce_fill with no separate pack step, summed
 * over the row shards by the caller (one all-reduce per iteration, the exchange of SURVEY 8e), and
 * consumed by reduce_apply; buf[K*D + K] then holds the GLOBAL number of reassigned rows.  In steady
 * state reduce_fill enqueues without any host read (update.hip: launch_move_deltas). */
size_t kmamd_reduce_len(kmamd_engine *e);
int kmamd_reduce_fill(kmamd_engine *e, const float *samples, const uint32_t *assignments_prev,
                      const uint32_t *assignments, double *buf);
int kmamd_reduce_apply(kmamd_engine *e, const double *buf, float *centroids, uint32_t *ccounts);
/* reduce_apply with the reference's stop rule (check_changed, kmeans.cu:697-717: evaluated BEFORE the update)
 * decided ON THE DEVICE, so that the caller can enqueue the next pass without waiting for the count:
 *   stop_threshold  tolerance * N as a float; the update happens only if (float)buf[K*D + K] (the reduced number
 *                   of reassigned rows) exceeds it, and then counters[0] is zeroed for the next pass.  Otherwise
 *                   NOTHING is modified and the engine's stop flag is raised: from then on kmamd_lloyd_assign
 *                   returns without touching anything (a pass enqueued speculatively leaves the state as the
 *                   reference returns it) until kmamd_stop_clear.  < 0: no test.
 *   seq             the caller's pass number.  The kernel reports to pinned words of the engine (two slots,
 *                   seq & 1); kmamd_stop_report(seq) waits for THIS call only (an event behind it on the engine's
 *                   stream; nothing has to wait in front of the next pass) and returns [0..3] the reduced
 *                   counters, [4] 1 if it stopped, [5] seq.  Reading a pass after the one after next has been
 *                   enqueued is an error (its slot has been reused). */
int kmamd_reduce_apply_stop(kmamd_engine *e, const double *buf, float *centroids, uint32_t *ccounts,
                            float stop_threshold, uint32_t seq);
int kmamd_stop_report(kmamd_engine *e, uint32_t seq, uint32_t *host_out6);
/* kmamd_reduce_apply_stop with the NEXT kmamd_lloyd_assign's centroid preparation fused into the same launch
 * (L2 metric, two-stage filter with the row cache in its steady state; in any other state it IS
 * kmamd_reduce_apply_stop).  Contract: the caller does not modify `centroids` between this call and that
 * kmamd_lloyd_assign, which recognises the buffer by its address. */
int kmamd_reduce_apply_prepare(kmamd_engine *e, const double *buf, float *centroids, uint32_t *ccounts,
                               float stop_threshold, uint32_t seq);
int kmamd_stop_clear(kmamd_engine *e);
/* The caller has written `centroids` itself (new seeds, an imported set, a rounding pass) since the engine last
 * saw them: whatever kmamd_reduce_apply_prepare prepared for that buffer is void.  (The engine recognises the
 * buffer by its ADDRESS only; without this call the next kmamd_lloyd_assign would filter against the panels of
 * the old values.) */
int kmamd_centroids_written(kmamd_engine *e);
/* Bounds carried from pass to pass (the Yinyang phase of kmeans_cuda()'s default schedule; lloyd_carry.hip).  on != 0:
 * from the next kmamd_lloyd_assign on, a pass in the two-stage filter's steady state (row cache valid) leaves per
 * row an upper bound of the distance to its centroid and a lower bound of the distance to every other centroid --
 * read off the coarse stage's best two scores, no extra distance work --, the next preparation measures how far every
 * centroid has moved, and the next pass only looks at the rows whose bounds, moved by those drifts, no longer certify
 * the assignment (Hamerly's test with the rounding of the reference's arithmetic as margin; angular metric: one number
 * per row, the certified gap of the scores).  Assignments, previous
 * assignments and counters are exactly those of plain passes.  Contract: between two passes the centroids change only
 * through kmamd_reduce_apply* / kmamd_apply_delta or are announced with kmamd_centroids_written (which voids the
 * bounds), and `assignments` / `assignments_prev` are the buffers of the previous pass, untouched.
 * kmamd_carry_stats: rows_spared = row passes the bounds have decided since the engine was created (one stream
 * synchronisation), last_list = the length of the newest list the host has heard of (0xFFFFFFFF: none yet). */
int kmamd_set_carry(kmamd_engine *e, int on);
int kmamd_carry_stats(kmamd_engine *e, uint64_t *rows_spared, uint32_t *last_list);
/* A row that stage 2 settles between two contenders carries the pair, an upper bound of both distances and a lower
 * bound of every other centroid's (angular: the gap by which both scores exceed every other centroid's); while the
 * drifts leave the latter above the former (the gap positive) the row goes straight to the
 * two-contender kernel (the reference's arithmetic and tie rule: kmeans.cu:214-364 restricted to the two) instead of
 * through the filter.  rows_paired = such row passes since the engine was created (they are not in rows_spared).
 * KMCUDA_AMD_CARRY_PAIRS=0 in the environment: without (A/B). */
int kmamd_carry_pair_stats(kmamd_engine *e, uint64_t *rows_paired);
/* The default Lloyd filter's duo list (csrc/lloyd_duo.hip; the reference has no counterpart, its kmeans_assign_lloyd
 * -- kmeans.cu:293-364 -- scans every centroid for every row): rows of the LAST assignment pass that stage 1 could
 * not decide but whose two contenders it knew by index, so that stage 2 settled them without sweeping the centroids
 * again.  The engine uses the list where it pays (lists beyond one round of stage-2 blocks); KMCUDA_AMD_DUO=0 / 2 in
 * the environment: never / always. */
int kmamd_duo_rows(kmamd_engine *e, uint32_t *rows);
/* The host side of the carried bounds replayed without a device (tests): pass i + 1 would count list_len[i] rows if it
 * counts a list, a pass's report reaches the host `lag` >= 1 passes later; out[i] = 0 plain pass (the bounds are paused),
 * 1 whole pass that leaves bounds but has none to move, 2 whole pass that counts its would-be list, 3 listed pass.
 * changed (may be null): the passes' reassignment counts, which the host judges `lag` passes late too -- a run that
 * converges slowly (counts falling by less than a third per pass) answers two hopeless lists with one long pause. */
int kmamd_carry_policy_sim(uint32_t n_passes, uint32_t n_rows, float list_max, const uint32_t *list_len, uint32_t lag,
                           uint8_t *out, const uint32_t *changed);
/* Test / A-B hook for the update's host logic: 0 default, 1 radix path always, 2 always read the
 * counts before choosing (the pre-round-2 behaviour), 3 bucket path always without reading (exercises
 * the device-side fallback of oversized buckets).  Env KMCUDA_AMD_UPDATE=radix|sync|bucket sets it at
 * engine creation.  Sums are bit-identical on every path. */
int kmamd_set_update_mode(kmamd_engine *e, int mode);

/* Strict-parity centroid update: kmeans_adjust (kmeans.cu:366-429) operation for operation --
 * one serial fp32 Kahan chain per centroid over its move events in ascending row order with the
 * reference's single shared compensation term, then normalize.  Centroids and ccounts are
 * bit-identical to the reference's.  Needs ALL rows on this engine (the chain order is global),
 * so it is the single-GPU verification mode (kmeans_cuda: env KMCUDA_AMD_EXACT_UPDATE=1). */
int kmamd_adjust_exact(kmamd_engine *e, const float *samples, const uint32_t *assignments_prev,
                       const uint32_t *assignments, float *centroids, uint32_t *ccounts);

/* Yinyang steps (reference: kmeans.cu:431-672), all on the engine's n_rows local rows.
 *   groups   K host uint32: centroid -> group, >= G for a NaN centroid (kmamd_yy_configure uploads it
 *            and builds the group-sorted panel index)
 *   bounds   (G+1) x n_rows group-major: [0] upper bound, [1+g] lower bound to group g
 *   drifts   K*D old centroids followed by K per-centroid drifts; gdrifts: G per-group maxima
 * yy_init: kmeans_yy_init (:431-485).  yy_drifts: kmeans_yy_calc_drifts + _find_group_max_drifts
 * (:487-538).  yy_filters: kmeans_yy_global_filter then _local_filter (:540-672); counters[2]
 * (passed) must be reset by the caller, counters[0] accumulates reassignments.
 * yy_init and the local filter run the matrix-core filter in front of the exact arithmetic
 * (env KMCUDA_AMD_YY_EXACT=1 at configure time: plain exact kernels); either way bounds, drifts
 * and assignments are bit-identical to the reference arithmetic. */
int kmamd_yy_configure(kmamd_engine *e, uint32_t G, const uint32_t *groups_host);
int kmamd_yy_init(kmamd_engine *e, const float *samples, const float *centroids, const uint32_t *assignments,
                  float *bounds);
int kmamd_yy_drifts(kmamd_engine *e, const float *centroids, float *drifts, float *gdrifts);
int kmamd_yy_filters(kmamd_engine *e, const float *samples, const float *centroids, const float *drifts,
                     const float *gdrifts, uint32_t *assignments, uint32_t *assignments_prev, float *bounds,
                     uint32_t *passed);

/* AFK-MC2's random numbers (reference: kmeans_afkmc2_random_step, kmeans.cu:107-116: curand_init(seed, thread, step)):
 * out[t * n + i] = draw i of the XORWOW stream (seed, subsequence t, offset), t < threads, as the seeding kernels
 * generate them -- cuRAND's seed scrambling in front of the generator cuRAND and rocRAND share (csrc/seeding.hip).
 * A window for tests: the draws must equal the oracle's restatement (tests/test_gpu_afkmc2_rng.py). */
int kmamd_afkmc2_draws(kmamd_engine *e, uint64_t seed, uint64_t offset, uint32_t threads, uint32_t n, uint32_t *out);

/* out[c][r] = in[r][c], 4-byte elements (reference: cuda_transpose, transpose.cu:83-117). */
int kmamd_transpose(kmamd_engine *e, const float *in, uint32_t rows, uint32_t cols, float *out);

/* Which filter kmamd_lloyd_assign runs in front of the exact kernels for this engine's shape: 1 the register-resident
 * ones (features <= 256: lloyd.hip / lloyd_f16.hip), 2 the LDS-streamed one (features > 256: lloyd_wide.hip), 0 none
 * (the exact kernels alone: the streamed filter switched off or out of memory).  *padded_width: the feature count the
 * filter's operands are padded to (0 with no filter).  Both kinds carry bounds (kmamd_set_carry; kind 2 up to 4096
 * features and without pair certificates). */
int kmamd_filter_kind(kmamd_engine *e, uint32_t *padded_width);

/* Timing of the dominant kernel on the engine's own stream (HIP events): start/stop bracket
 * the next lloyd filter launches; elapsed returns the summed milliseconds and launch count. */
int kmamd_profile_reset(kmamd_engine *e);
int kmamd_profile_read(kmamd_engine *e, double *filter_ms, uint32_t *filter_launches,
                       double *exact_ms, double *update_ms);
/* stage 1 of the two-stage filter on its own (HIP events around that one launch, inside the filter span) */
int kmamd_profile_read_coarse(kmamd_engine *e, double *coarse_ms);
int kmamd_profile_enable(kmamd_engine *e, int on);

/* What the last kmeans_cuda() call of this process did: iterations of its Lloyd / Yinyang loops, the
 * wall-clock seconds spent in them (after upload + seeding, before the outputs are gathered), the
 * seconds before (upload, seeding), the number of row shards and of RCCL ranks (0: single GPU or the
 * one-device test hook).  Lets a driver time the drop-in entry point itself (bench.py --api). */
int kmamd_last_run_stats(uint32_t *iterations, double *loop_seconds, double *setup_seconds, uint32_t *shards,
                         uint32_t *rccl_ranks);

/* With KMCUDA_AMD_TIME_COLLECTIVE=1 in the environment of that call (several row shards): the summed duration of its
 * per-iteration all-reduces -- ncclAllReduce over the device mask, or the one-device stand-in under
 * KMCUDA_AMD_VIRTUAL_SHARDS -- as HIP events on the first shard's stream bracketed them (the wait for the slowest
 * shard's move sums is inside: that is what an iteration pays), and their number.  0 / 0 otherwise. */
int kmamd_last_run_collective(double *milliseconds, uint32_t *count);

/* host -> raw device pointer copy on `device` (what python.cc:330-345 does with cudaMemcpy for imported
 * centroids in device-pointer mode; lets a binding without a HIP runtime of its own fill caller-owned memory). */
int kmamd_copy_to_device(int device, void *dst, const void *host_src, size_t bytes);

/* Library identification: returns the gfx arch string this library was compiled for. */
const char *kmamd_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* KMCUDA_AMD_H */
