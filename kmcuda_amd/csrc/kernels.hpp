// kernels.hpp -- launch interfaces between the HIP kernels (*.hip) and the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmx {

struct LloydArgs {
  const float *samples;      // N x D row-major (this device's rows)
  uint32_t N, D, K;
  uint32_t K_pad;            // K rounded up to 32 (filter tiles)
  uint32_t DP;               // filter feature padding (0: no MFMA filter for this D)
  uint32_t Kt;               // K rounded up to 64 (row length of ct)
  const float *cfil;         // K_pad x DP sanitised, CENTRED centroid panel (c - mu)
  const float *bias;         // K_pad: -||c - mu||^2/2 (L2) / mu.(c - mu) (angular) / -inf (never chosen)
  const float *mu;           // DP: mean of the finite centroids (zero padded)
  const float *ct;           // D x Kt transposed centroids
  const float *csqr;         // K exact squared norms (reference's sum_squares)
  const uint32_t *stats;     // bits of: [0] max ||c - mu||^2, [1] max bias magnitude, [2] max ||c||^2,
                             //          [3] ||mu||^2, [4] max bias2 magnitude (original-row variant)
  float eps;                 // relative error coefficient of the filter bound
  float tie_slack;           // absolute slack (angular: acos plateau width)
  uint32_t *assignments, *assignments_prev;
  uint32_t *flagged;         // N: rows with three or more contenders (full exact scan)
  uint32_t *pairs;           // 3N: (row, i1, i2) rows with exactly two contenders
  uint32_t *counters;        // [0] changed, [1] flagged, [2] passed (yinyang), [3] pairs, [4] undecided by stage 1,
                             // [kStopFlag] the device-side stop flag
};
// counters[kStopFlag] != 0: the stop rule fired ON THE DEVICE (apply_delta_kernel with a threshold): the kernels
// that assign every row return at once, so iterations enqueued past the stop leave the state untouched
// Kernels with one wave per row (four rows per 256-thread block) stride over the rows from a bounded grid: a grid of
// N / 4 blocks x 256 threads passes 2^32 threads at N = 2^26 rows, which the runtime does not launch as asked (found by
// the reference's 167 772 160-row case, test.py:307-326: the k-means++ byte copy was only partly written)
constexpr uint32_t kWaveRowGridMax = 1u << 20;
inline uint32_t wave_row_grid(uint32_t n_rows) {
  const uint32_t g = (n_rows + 3u) / 4u;
  return g < kWaveRowGridMax ? (g ? g : 1u) : kWaveRowGridMax;
}
constexpr uint32_t kStopFlag = 8;
// counters[kCarryCursor]: length of the carried-bounds pass's row list (zeroed by the preparation kernel);
// counters[kCarrySkipped], [kCarrySkipped + 1]: one 64-bit total of the rows the bounds have spared since the engine was made
constexpr uint32_t kCarryCursor = 9, kCarrySkipped = 10;
// counters[kCarryPaired], [kCarryPaired + 1]: 64-bit total of the rows the carried PAIR certificates have sent straight to
// the two-contender kernel (lloyd_carry.hip)
constexpr uint32_t kCarryPaired = 12;
// counters[kDuoCount]: length of stage 1's list of rows whose (two) contenders it knows by index (lloyd_coarse.hpp:
// the four quarters' trackers); zeroed with counters[4] by the preparation kernels
constexpr uint32_t kDuoCount = 14;
// Bounds carried from one Lloyd pass to the next (lloyd_carry.hip): per row an upper bound of the distance to its
// centroid and a lower bound of the distance to every other finite centroid.
struct CarryArgs {
  float *ub = nullptr, *lb = nullptr;
  const uint32_t *row_list = nullptr;   // CARRY == 2: the rows of this pass (carry_skip_kernel's survivors) ...
  const uint32_t *n_list = nullptr;     // ... and their number, on the device
  uint32_t *host_report = nullptr;      // device address of 2 pinned words: [0] <- the list's length (N for a whole pass), [1] <- seq
  uint32_t seq = 0;
  // angular metric: ub[] holds the certified SCORE gap between the row's centroid and every other one instead (the
  // reference decides on products there; |x.(c_new - c_old)| <= ||x|| ||c_new - c_old|| moves it), lb[] is unused
  int angular = 0;
  // The rows stage 2 decides between two contenders (p1, p2) come out of it with l3[] > 0.  L2: an upper bound of BOTH
  // distances in ub[], lb[] void, l3[] a lower bound of the distance to every other finite centroid: while l3 stays
  // above ub under the drifts, the reference's nearest is one of the two and the pair kernel alone looks at the row.
  // Angular: l3[] = the gap by which both contenders' scores exceed every other centroid's, shrunk by the drifts.
  // l3[] == 0: no such statement (stage 1 writes that for every row it sees)
  float *l3 = nullptr;
  uint32_t *p1 = nullptr, *p2 = nullptr;
};
// the device-side stop rule of launch_apply_delta (reference: check_changed, kmeans.cu:697-717)
struct StopCtl {
  float threshold = -1.f;         // stop when (float)reassigned <= threshold (= tolerance * N in float); < 0: no test
  uint32_t *counters = nullptr;   // engine counters: [0] zeroed when the run goes on, [kStopFlag] raised when it stops
  uint32_t *host_tail = nullptr;  // device-visible pinned words: [0..3] reduced counters, [4] stopped, [5] seq
  uint32_t seq = 0;
};

uint32_t filter_dp_for(uint32_t D);
uint32_t lloyd_dp_for(uint32_t D);   // + 512 for 256 < D <= 512: two-stage f16 Lloyd filter only
hipError_t launch_centroid_prep(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad,
                                uint32_t DP, uint32_t Kt, float *csqr, float *bias, float *bias2, float *cfil,
                                float *ct, float *mu, bool freeze_mu, uint32_t *finite, uint32_t *stats, uint32_t *zero_a,
                                uint32_t *zero_b, uint32_t *zero_c, hipStream_t st);
hipError_t launch_lloyd_filter(const LloydArgs &a, hipStream_t st);
hipError_t launch_lloyd_pair(int metric, const LloydArgs &a, const float *centroids, uint32_t grid, hipStream_t st);
hipError_t launch_lloyd_exact(int metric, const LloydArgs &a, const uint32_t *rows, const uint32_t *nrows,
                              uint32_t grid, hipStream_t st);
// the pair rows and the full-scan rows of a pass in ONE launch, operands staged through LDS / prefetched three
// groups deep (lloyd.hip: lloyd_settle_kernel); supported: D % 4 == 0 and 16-byte aligned rows
bool lloyd_settle_supported(const LloydArgs &a, const float *centroids);
hipError_t launch_lloyd_settle(int metric, const LloydArgs &a, const float *centroids, hipStream_t st);

// lloyd_f16.hip -- the two-stage filter on the f16 matrix cores (centred operands), for fp32
// rows and for the fp16x2 path's half rows; decisions identical in kind, refine kernels shared
bool lloyd_filter_f16_supported(uint32_t D, uint32_t DP);
// panelhi = hi(c - mu) rows + clamped biases for the coarse pass, K_pad rounded up to 64 rows,
// (DP + 2) * 2 bytes per row; stats[5] = max ||c' - hi(c')||^2
hipError_t launch_centroid_panelhi(const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP,
                                   const uint32_t *finite, const float *mu, const float *bias, void *panelhi,
                                   uint32_t *stats, hipStream_t st);
// the steady-state preparation of the two-stage filter in one kernel (mean frozen); zeroes stats_next
// and the three list counters.  csqr / ct come from launch_centroid_rows on another stream
hipError_t launch_centroid_prep_frozen(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad,
                                       uint32_t DP, const float *mu, uint32_t *finite, float *bias, float *bias2,
                                       float *cfil, void *panelhi, uint32_t *stats, uint32_t *stats_next,
                                       uint32_t *zero_a, uint32_t *zero_b, uint32_t *zero_c,
                                       float *drift /* K: ||c - c of the previous pass||, or null; its maximum -> stats[6] */,
                                       uint32_t *zero_d /* or null */, hipStream_t st);
// L2 metric: the centroid update of launch_apply_delta (same formula, same StopCtl) fused in front of that
// preparation -- one launch between the all-reduce and stage 1
hipError_t launch_apply_prep_frozen(const double *delta, const double *dcount_d, float *centroids, uint32_t *ccounts,
                                    const StopCtl &stop, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP,
                                    const float *mu, uint32_t *finite, float *bias, float *bias2, float *cfil,
                                    void *panelhi, uint32_t *stats, uint32_t *stats_next, uint32_t *zero_a,
                                    uint32_t *zero_b, uint32_t *zero_c, float *drift, uint32_t *zero_d, hipStream_t st);
// lloyd_carry.hip -- passes that carry per-row distance bounds (CarryArgs): the coarse stage over every row
// (cy.row_list == nullptr: from the row cache) or over the listed rows, leaving fresh bounds; carry_skip moves the
// bounds by the drifts (launch_centroid_prep_frozen) and lists the rows they no longer decide into row_list /
// counters[kCarryCursor] (probe: only counts them)
hipError_t launch_lloyd_coarse_carry(const LloydArgs &a, const void *rows, bool half_rows, const void *xcache,
                                     const float *xmeta, const void *panelhi, uint32_t *undecided, float *und_thr,
                                     const CarryArgs &cy, uint32_t rows_hint, uint32_t *duo, hipStream_t st);
// (cy.l3 / p1 / p2 / finite / pairs: the pair certificates; pairs[3 counters[3]++] = (row, p1, p2))
hipError_t launch_carry_skip(uint32_t N, uint32_t K, const uint32_t *assignments, uint32_t *assignments_prev,
                             const CarryArgs &cy, const float *xmeta, const float *drift, const uint32_t *stats,
                             float tie_slack, uint32_t *row_list, const uint32_t *finite, uint32_t *pairs,
                             uint32_t *counters, bool probe, hipStream_t st, uint32_t wide_dg = 0);
// (wide_dg != 0: xmeta is the streamed filter's record table, lloyd_wide.hip, rows padded to wide_dg features)
// the reference's exact sum_squares (csqr) + the transposed panel (ct) alone: what the pair / exact kernels read
hipError_t launch_centroid_rows(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t Kt, float *csqr,
                                float *ct, hipStream_t st);
// stage 1 of the default filter: hi.hi products only; rows it cannot decide -> undecided[counters[4]++].
// xcache / xmeta: the engine's row cache (launch_row_cache) or nullptr (operands converted from rows)
// (duo != nullptr: undecided rows whose two contenders stage 1 knows by index go to duo[4 counters[kDuoCount]++]
//  = (row, contender, contender, best score of the others) instead, for launch_lloyd_duo)
hipError_t launch_lloyd_coarse(const LloydArgs &a, const void *rows, bool half_rows, const void *xcache,
                               const float *xmeta, const void *panelhi, uint32_t *undecided, float *und_thr,
                               uint32_t *duo, hipStream_t st);
// stage 2 without its sweep for the duo rows (lloyd_duo.hip): the two contenders scored in fp32, decided like
// launch_lloyd_refine's rows
// cy (a carried pass with pair certificates: cy->l3 != nullptr): the rows leave with stage 2's bounds and certificates
hipError_t launch_lloyd_duo(const LloydArgs &a, const uint32_t *duo, hipStream_t st, const CarryArgs *cy = nullptr);
// stage 2 of the default filter: the undecided rows' contenders (coarse score >= und_thr) scored in fp32
hipError_t launch_lloyd_refine(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                               const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                               uint32_t rows_hint /* expected list length, 0xFFFFFFFF = unknown */, hipStream_t st);
// the same in a carried pass (lloyd_carry.hip): the rows it settles leave with bounds and pair certificates (cy.l3)
hipError_t launch_lloyd_refine_carry(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                     const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                                     uint32_t rows_hint, const CarryArgs &cy, hipStream_t st);
// x' = x - mu as halves in the coarse kernel's operand order (N rounded up to 256 rows: DP*2 bytes per
// row) + (||x'||^2, x_0) per row (8 bytes); valid while mu is unchanged
hipError_t launch_row_cache(const void *rows, bool half_rows, uint32_t N, uint32_t D, uint32_t DP, const float *mu,
                            void *xcache, float *xmeta, hipStream_t st);

// lloyd_wide.hip -- the assignment filter for D beyond the register-resident kernels: both operands streamed through
// LDS (256 rows x 256 centroids per block, 64-feature chunks), best / second-best per row in registers, no score ever
// written; then the same sweep over the rows it listed for their contenders
hipError_t launch_row_halves(const void *rows, bool half_rows, uint32_t N, uint32_t D, uint32_t DG, const float *mu,
                             void *xg, float *meta /* 4 floats per row, N + 1 records: ||mu|| in the last */, hipStream_t st);
size_t wide_cont_words(uint32_t N);   // contender table: (1 + 16) words per row
// commits / lists (undecided, und_thr, counters[4]) every row, then writes the listed rows' contenders (und_cont).
// cy (carried bounds, as launch_lloyd_coarse_carry): every row (cy->row_list == nullptr) or the rows of cy->row_list,
// and the rows the pass looks at leave with fresh bounds (cy->ub / lb; no pair certificates on this path)
hipError_t launch_lloyd_wide(const LloydArgs &a, const void *xg, const float *meta, uint32_t DG /* % 64 == 0 */,
                             const void *panelhi, uint32_t *undecided, float *und_thr, uint32_t *und_cont, hipStream_t st,
                             const CarryArgs *cy = nullptr);
hipError_t launch_wide_contenders(int metric, const LloydArgs &a, const float *centroids, uint32_t DG,
                                  const uint32_t *und_rows, const uint32_t *und_cont, hipStream_t st);

// update.hip -- centroid update (reference: kmeans.cu:366-429 kmeans_adjust)
size_t sort_temp_bytes(size_t n, uint32_t max_key);
uint32_t move_bucket_stride(uint32_t K);
size_t move_bucket_words(uint32_t K);        // launch_move_deltas' bucket_work, zero-initialised by the owner
uint32_t move_bucket_cap(uint32_t N, uint32_t K);   // rows per (centroid, sign) bucket: bucket_rows is 2 K x cap words
hipError_t launch_inverse_assignments(const uint32_t *assignments, uint32_t N, uint32_t K, uint32_t *keys_tmp,
                                      uint32_t *vals_tmp, uint32_t *keys_sorted, uint32_t *inv,
                                      uint32_t *offsets, void *temp, size_t temp_bytes, hipStream_t st);
// host-side state of the update across calls (engine-owned)
struct MoveState {
  uint32_t *host = nullptr;            // 4 pinned words the kernels report to: [0] events, [1] largest list,
  uint32_t *host_dev = nullptr;        //   [2] undecided rows of the last assignment pass; host_dev: their device address
  uint32_t last_events = 0xFFFFFFFFu;  // newest event count the host knows
  int force = 0;                       // kmamd_set_update_mode
  uint32_t n_radix = 0, n_direct = 0;  // calls by path (KMCUDA_AMD_UPDATE_TRACE: kmeans_cuda prints them at its end)
};
hipError_t launch_move_deltas(const float *samples, uint32_t N, uint32_t D, uint32_t K, const uint32_t *prev,
                              const uint32_t *cur, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t *keys_sorted,
                              uint32_t *rows_sorted, uint32_t *offsets2, void *temp, size_t temp_bytes,
                              uint32_t *bucket_rows, uint32_t cap, double *delta, int32_t *dcount /* may be null */,
                              double *tail /* fused buffer's [dcount | counters], may be null */,
                              const uint32_t *counters, uint32_t *blockoff, uint32_t *bucket_work, MoveState *ms,
                              hipStream_t st);
hipError_t launch_adjust_exact(int metric, const float *samples, uint32_t N, uint32_t D, uint32_t K,
                               const uint32_t *prev, const uint32_t *cur, uint32_t *keys_tmp, uint32_t *vals_tmp,
                               uint32_t *keys_sorted, uint32_t *rows_sorted, uint32_t *offsets2, void *temp,
                               size_t temp_bytes, float *work, float *centroids, uint32_t *ccounts, hipStream_t st);
hipError_t launch_apply_delta(int metric, const double *delta, const int32_t *dcount /* or */,
                              const double *dcount_d /* the fused buffer's tail */, uint32_t K, uint32_t D,
                              float *centroids, uint32_t *ccounts, const StopCtl &stop, hipStream_t st);
// out[b][i] = sum over b of bufs[b][i] in ascending b, written to every buffer (the one-device stand-in for the
// all-reduce: KMCUDA_AMD_VIRTUAL_SHARDS)
hipError_t launch_sum_buffers(double *const *bufs_dev, uint32_t nbuf, size_t len, hipStream_t st);

// Code objects load on the first use of a kernel of their translation unit: ~24 ms per process in front of the first
// iteration's kernels (16 of them the update's radix sort).  kmeans_cuda() touches them from a helper thread while its
// own thread uploads and seeds (kmcuda_api.cpp: preload_code_objects).
hipError_t preload_update_code();
hipError_t preload_lloyd_f16_code();
hipError_t preload_lloyd_code();
hipError_t preload_lloyd_carry_code();
hipError_t preload_lloyd_duo_code();
hipError_t launch_gather_rows(const float *samples, const uint32_t *idx, uint32_t K, uint32_t D, float *out, hipStream_t st);
// seeding.hip (reference: kmeans.cu:42-67 kmeans_plus_plus, transpose.cu:6-14 copy_sample_t,
// kmeans.cu:674-691 kmeans_calc_average_distance)
hipError_t launch_kmpp_step(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroid,
                            uint32_t cc, float *dists, hipStream_t st);
// k-means++ with the chooser on the device (seeding.hip): step = distances + exact block sums + exponent
// range; totals_host = pinned { double sum_g, sum_d; uint32 emin, emax, bad, chosen } (32 bytes)
// fail (one device word, 0 while all is well): raised by launch_kmpp_choose's kernel at a step the device cannot
// decide; every kmpp kernel launched with it returns at once while it is set (the host enqueues steps ahead)
// out: the step's exponent cut (one device word, read; zero-initialised by the owner, written by launch_kmpp_choose)
// and where the non-zero distances below it are listed instead of summed (kmpp_outlier_bytes() + one count word,
// zero on entry, zeroed again by launch_kmpp_choose) -- seeding.hip: KmppOutlier
struct KmppOutlierBuf {
  uint32_t *ecut = nullptr;
  void *outl = nullptr;
  uint32_t *outl_count = nullptr;
};
size_t kmpp_outlier_bytes();
size_t kmpp_totals_bytes();
hipError_t launch_kmpp_step2(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroid,
                             uint32_t cc, float *dists, void *block_stats, double *bpre, void *totals,
                             const uint32_t *fail, const KmppOutlierBuf &out, hipStream_t st);
// the reference's chooser (kmcuda.cc:300-326) for step `step` with random number `choice`, on the device, over the
// concatenation of `nshards` row shards (1: the whole job on one GPU): every shard has run the step on its rows
// (launch_kmpp_step2 / _filtered with its own length and buffers); the kernel runs on shards[0]'s device, reads the
// other shards' totals / prefixes / distances in place (peer access), copies the chosen row from its owner into
// centroid slot `step` of EVERY shard, or raises every shard's *fail = step.  offset % 256 == 0 for every shard.
constexpr int kKmppMaxShards = 16;
struct KmppShardPtrs {
  const float *dists;
  const double *bpre;
  void *totals;
  const float *samples;
  float *centroids;
  uint32_t *fail;
  uint32_t offset, length;
  KmppOutlierBuf out;
};
hipError_t launch_kmpp_choose(const KmppShardPtrs *shards, uint32_t nshards, uint32_t N, double choice, uint32_t log2n,
                              uint32_t step, uint32_t D, hipStream_t st);
// filtered k-means++ steps (seeding.hip): a centred BYTE copy of the rows (per-row scale and measured residual), then
// per step the survivors of the k-NN candidate bound get the exact chain.  DP: D rounded up to 128; xs8: N x DP bytes;
// meta: N x 4 floats; stats: 4 words ([2..3]: exact chains run so far, 64 bits); list: N words.
hipError_t launch_kmpp_cache(const float *samples, uint32_t N, uint32_t D, uint32_t DP, double *part, float *mu,
                             void *xs8, float *meta, uint32_t *stats, hipStream_t st);
hipError_t launch_kmpp_step_filtered(int metric, const float *samples, uint32_t N, uint32_t D, uint32_t DP,
                                     const void *xs8, const float *meta, const float *mu, uint32_t *stats,
                                     uint32_t *list, const float *centroid, uint32_t cc, float *dists,
                                     void *block_stats, double *bpre, void *totals, const uint32_t *fail,
                                     const KmppOutlierBuf &out, hipStream_t st);
size_t kmpp_block_stat_bytes(uint32_t N);
size_t kmpp_blocks(uint32_t N);
size_t kmpp_prefix_doubles(uint32_t N);   // doubles of `bpre`
// AFK-MC2 seeding (seeding.hip; reference kmeans.cu:69-212)
hipError_t launch_afk_qdist(int metric, const float *samples, uint32_t N, uint32_t D, const float *c1, float *dists,
                            hipStream_t st);
hipError_t launch_afk_q(float *q, uint32_t N, float dsum, hipStream_t st);
// n draws of each of `threads` streams (seed, subsequence = thread, offset): out[thread * n + i] (tests)
hipError_t launch_afk_draws(unsigned long long seed, unsigned long long offset, uint32_t threads, uint32_t n, uint32_t *out,
                            hipStream_t st);
hipError_t launch_afk_random_step(uint32_t m, uint64_t seed, uint64_t seq, const float *q, uint32_t N,
                                  uint32_t *choices, float *rand_a, hipStream_t st);
hipError_t launch_afk_min_dist(int metric, uint32_t m, uint32_t k, const float *samples, uint32_t D,
                               const uint32_t *choices, const float *centroids, float *min_dists, hipStream_t st);
hipError_t launch_member_distances(int metric, const float *samples, uint32_t N, uint32_t D,
                                   const float *centroids, const uint32_t *assignments, uint32_t K,
                                   float *dists, hipStream_t st);

// yinyang.hip (reference: kmeans.cu:431-672)
hipError_t launch_yy_init(int metric, const float *xt, uint32_t len, uint32_t D, uint32_t G, const float *centroids,
                          const uint32_t *assignments, const uint32_t *cperm, const uint32_t *gstart, float *bounds,
                          hipStream_t st);
hipError_t launch_yy_group_max(uint32_t K, uint32_t D, uint32_t G, const uint32_t *groups, const float *drifts,
                               float *gdrifts, hipStream_t st);
hipError_t launch_yy_drifts(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t G,
                            const uint32_t *groups, float *drifts, float *gdrifts, hipStream_t st);

// yinyang_mfma.hip: the same steps with the matrix-core filter in front of the exact arithmetic
struct YyArgs {
  const float *samples;      // len x D row-major
  const float *centroids;    // K x D original values (exact chains)
  uint32_t len, D, DP, K, K_pad, G;
  const float *cfil, *bias, *mu;  // centred panel of centroid_prep (ascending c)
  const uint32_t *stats;
  float eps;
  const uint32_t *groups;    // K, padded with 0xFFFFFFFF to a multiple of 64
  const float *drifts, *gdrifts;
  uint32_t *assignments;
  float *bounds;
  const uint32_t *passed;
  const uint32_t *count_ptr; // length of `passed` (counters + 2, or counters + 5 for the fall-back list)
  uint32_t *counters;        // + [5] rows of the hinted kernel handed to the plain one
  uint32_t *stat_stripes;    // 64 x 16 words: the hinted kernel's statistics ([6] rows, [7] handed over, [8..11] why), summed by the reader
  // hinted local filter (yinyang_hint.hip)
  const void *panelhi;       // hi halves of the centred panel (centroid_panelhi_kernel), DP halves per row
  const void *xcache;        // the engine's row cache (lloyd_f16.hip: row_cache_kernel) for the SAME mean, or null:
  const float *xmeta;        //   hi halves of x - mu in operand order + (||x'||^2, ||x' - hi(x')||^2) per row
  float *hint;               // per passed row: S' >= upper bound (or +inf: no hint)
  uint32_t *flag_rows;       // rows the hinted kernel could not settle
  const uint32_t *gfirst, *gsecond;  // G: the two smallest member indices of every group (0xFFFFFFFF: none)
  // yy_init: group-sorted padded panel
  const float *pfil, *pbias;
  const uint32_t *pids, *pmeta, *cperm, *gstart;
  uint32_t nslots;
};
hipError_t launch_yy_local_mfma(int metric, const YyArgs &a, hipStream_t st);
// yinyang_hint.hip: coarse second-best estimate per passed row (f16 matrix cores), then the local filter
// with that estimate as its candidate threshold; rows it cannot settle go to flag_rows (counters[5])
bool yy_hint_supported(uint32_t DP);
hipError_t launch_yy_hint(int metric, const YyArgs &a, hipStream_t st);
hipError_t launch_yy_local_hint(int metric, const YyArgs &a, hipStream_t st);
// yinyang_init.hip: the same step with the exact chains fed from registers and LDS (original-value panel)
hipError_t launch_yy_init_lds(int metric, const YyArgs &a, hipStream_t st);
hipError_t launch_yy_orig_panel(int metric, const float *centroids, uint32_t D, uint32_t DP, const uint32_t *pids,
                                uint32_t nslots, float *pfil, float *pbias, hipStream_t st);
hipError_t launch_yy_global_filter(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                   const float *centroids, const float *drifts, const float *gdrifts,
                                   const uint32_t *assignments, uint32_t *assignments_prev, float *bounds,
                                   uint32_t *passed, uint32_t *counters, hipStream_t st);
hipError_t launch_yy_local_filter(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                  const float *centroids, const uint32_t *groups, const float *drifts,
                                  const float *gdrifts, uint32_t *assignments, float *bounds, const uint32_t *passed,
                                  uint32_t *counters, hipStream_t st);

// knn.hip (reference: knn.cu)
// queries (consecutive sorted positions of one cluster) per block: f32 filter 4 waves x 32, f16 filter
// knn16_waves(DP) x knn16_nset(DP) x 32
// Rows up to 256 features: 8 waves x 2 operand sets = 512 queries share every staged candidate tile, ONE block per CU
// (round 5: 4 waves x 2 sets, two blocks per CU, until then -- 2 bytes fetched per scored pair, 4.3 TB for config D's
// share, the kernel HBM co-bound; 8-wave blocks halve the fetches -- 2.27 TB by FETCH_SIZE -- and take 1.02 s against
// 1.10, profiles/r5e_knn_filter_8wave_blocks_ab.log).  Wider rows (DP = 512 / 768 / 1024): ONE operand set per wave (a
// lane's half of a 512-feature row in halves is 128 registers) and four waves (768 / 1024: a wave needs its SIMD to
// itself, 512 registers), one 32-candidate sub-tile per staged tile (32 / 48 / 64 KB).
#ifndef KNN16_WAVES_NARROW
#define KNN16_WAVES_NARROW 8    // waves per block, DP <= 256
#define KNN16_NSET 2            // 32-query operand sets per wave, DP <= 256
#endif
#ifndef KNN16_SUB
#define KNN16_SUB 2             // 32-candidate sub-tiles per staged tile (per barrier)
#endif
#ifndef KNN16_NBUF
#define KNN16_NBUF 2            // LDS ring of candidate tiles (knn_f16.hip)
#endif
#ifndef KNN16_PD
#define KNN16_PD 3              // candidate fragments in flight per wave
#endif
#define KNN16_PAD_ROWS 64   // rows of xs16 / entries of kbias the caller allocates (and the split zeroes) past N
constexpr int knn16_waves(int DP) { return DP > 256 ? 4 : KNN16_WAVES_NARROW; }
constexpr int knn16_blocks_per_cu(int DP) { return DP > 512 ? 1 : (DP > 256 ? 2 : (KNN16_WAVES_NARROW > 4 ? 1 : 2)); }
constexpr int knn16_nset(int DP) { return DP > 256 ? 1 : KNN16_NSET; }
constexpr int knn16_sub(int DP) { return DP > 256 ? 1 : KNN16_SUB; }
constexpr int knn16_nbuf(int DP) { return DP > 256 ? 2 : KNN16_NBUF; }   // (the wide instantiations' tiles: two buffers fill their LDS)
constexpr uint32_t KNN_QPB_F32 = 128;
constexpr uint32_t knn_qpb_f16(uint32_t DP) { return (uint32_t)knn16_waves((int)DP) * (uint32_t)knn16_nset((int)DP) * 32u; }
struct KnnArgs {
  const float *xs;          // N x DP cluster-sorted rows (zero padded to DP)
  const float *n2s;         // N plain squared norms of the sorted rows
  const uint32_t *inv;      // N: sorted position -> sample index (inverse assignments)
  const uint32_t *offsets;  // K+1: cluster c occupies positions [offsets[c], offsets[c+1])
  const float *mydist;      // N: exact distance of each sorted row to its own centroid
  const float *R, *C;       // K radii, K x K centroid distances
  const uint32_t *blocks;   // (cluster, first position) per block of KNN_QPB_* queries
  const uint32_t *stats;    // [0] max squared norm bits
  uint32_t N, D, DP, K, k;
  uint32_t p_base, p_end;   // this launch covers sorted positions [p_base, p_end)
  float eps;
  // f16 matrix-core filter (knn_f16.hip): centred hi/lo-split rows; n2s then holds the CENTRED squared
  // norms, mux[p] = mu.(x_p - mu), mu2 = ||mu||^2
  const void *xs16;         // (N + KNN16_PAD_ROWS) x DP halves
  const float *mux;
  const float *kbias;       // N + KNN16_PAD_ROWS: -0.5 * centred squared norm (L2) / mu.(x - mu) (angular)
  float mu2;
  // optional (L2, f16 filter): lb[c * lb_stride + (p - p_base)] <= d(x_p, centroid c) - R[c] in the reference's
  // arithmetic (knn_centroid_bounds_kernel); nullptr: the reference's prune test alone
  const float *lb = nullptr;
  size_t lb_stride = 0;
  // optional (f16 filter): slot i of the launch's block plan (sorted position p_base + i) handles the query at sorted
  // position qperm[i] of the same cluster (launch_knn_query_order); nullptr: i itself
  const uint32_t *qperm = nullptr;
  float *heaps;             // (p_end - p_base) x 2k
  uint32_t *out;            // (p_end - p_base) x k, sorted-position order
  unsigned long long *calced;  // [0] pairs the REFERENCE's prune rule visits (knn.cu:228); knn_f16.hip also: [1] pairs
                               // scored on the matrix cores (32 x 32 per wave, operand set and sub-tile), [2] of those:
                               // live, unpruned query x real candidate, [3] exact chains evaluated, [4] what [1] would
                               // be if an operand set without a visiting query were not scored
};
constexpr int KNN_STATS = 5;
hipError_t launch_knn_gather(const float *samples, uint32_t N, uint32_t D, uint32_t DP, const uint32_t *inv,
                             float *xs, float *n2s, uint32_t *stats, hipStream_t st);
hipError_t launch_knn_prep(int metric, const float *xs, uint32_t N, uint32_t D, uint32_t DP, const uint32_t *offsets,
                           uint32_t K, const float *centroids, float *mydist, float *rdist, float *R, float *C,
                           bool strict_h2, hipStream_t st);
hipError_t launch_knn_filter(int metric, const KnnArgs &a, uint32_t nblocks, hipStream_t st);
// lb[c * stride + (p - p_base)] for the sorted positions [p_base, p_end) and all K centroids (L2, D <= 1024)
hipError_t launch_knn_centroid_bounds(const float *xs, uint32_t D, uint32_t DP, uint32_t p_base, uint32_t p_end,
                                      const float *centroids, uint32_t K, const float *R, float *lb, size_t stride,
                                      hipStream_t st);
bool launch_knn_query_order(const float *lb, size_t stride, const uint32_t *offsets, uint32_t K, uint32_t p_base,
                            uint32_t p_end, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t *keys_sorted,
                            uint32_t *qperm, void *temp, size_t temp_bytes, hipStream_t st, int mode,
                            const float *mydist, const float *R);
// strict_h2 (both): the reference's half2 arithmetic on rows that hold half values (KMCUDA_AMD_FP16_STRICT)
hipError_t launch_knn_exact(int metric, const KnnArgs &a, bool strict_h2, hipStream_t st);
hipError_t launch_knn_split(int metric, const float *xs, uint32_t N, uint32_t D, uint32_t DP, const float *mu,
                            void *xs16, float *n2c, float *mux, float *kbias, uint32_t *stats, hipStream_t st);
hipError_t launch_knn_filter_f16(int metric, const KnnArgs &a, uint32_t nblocks, hipStream_t st);
hipError_t launch_knn_scatter(const uint32_t *sorted_out, const uint32_t *inv, uint32_t p_base, uint32_t p_end,
                              uint32_t k, uint32_t *neighbors, hipStream_t st);

// fp16x2 boundary conversions (seeding.hip)
hipError_t launch_half_to_float(const void *src, size_t n, float *dst, hipStream_t st);
hipError_t launch_float_to_half(const float *src, size_t n, void *dst, hipStream_t st);
hipError_t launch_quantize_half(float *v, size_t n, hipStream_t st);

// half2_strict.hip -- the reference's half2 arithmetic (fp_abstraction.h:100-182) kernel by kernel; plain
// thread-per-item kernels on fp32 words holding halves (KMCUDA_AMD_FP16_STRICT=1: verification mode)
hipError_t launch_h2_csqr(int metric, const float *centroids, uint32_t K, uint32_t D, float *sq2, hipStream_t st);
hipError_t launch_h2_assign(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroids, uint32_t K,
                            const float *sq2, uint32_t *assignments, uint32_t *assignments_prev, uint32_t *counters,
                            hipStream_t st);
hipError_t launch_h2_adjust(int metric, const float *samples, uint32_t N, uint32_t D, uint32_t K, const uint32_t *prev,
                            const uint32_t *cur, float *centroids, uint32_t *ccounts, hipStream_t st);
hipError_t launch_h2_to_row(int metric, const float *samples, uint32_t N, uint32_t D, const float *row, uint32_t cc,
                            int mode /* 0 k-means++ step, 1 squared distance */, float *dists, hipStream_t st);
hipError_t launch_h2_member(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroids,
                            const uint32_t *assignments, uint32_t K, float *dists, hipStream_t st);
hipError_t launch_h2_afk_min_dist(int metric, uint32_t m, uint32_t k, const float *samples, uint32_t D,
                                  const uint32_t *choices, const float *centroids, float *min_dists, hipStream_t st);
hipError_t launch_h2_yy_init(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                             const float *centroids, const uint32_t *assignments, const uint32_t *groups, float *bounds,
                             hipStream_t st);
hipError_t launch_h2_yy_drifts(int metric, const float *centroids, uint32_t K, uint32_t D, float *drifts, hipStream_t st);
hipError_t launch_h2_yy_global(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                               const float *centroids, const float *drifts, const float *gdrifts,
                               const uint32_t *assignments, uint32_t *assignments_prev, float *bounds, uint32_t *passed,
                               uint32_t *counters, hipStream_t st);
hipError_t launch_h2_yy_local(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                              const uint32_t *passed, const float *centroids, const uint32_t *groups, const float *drifts,
                              const float *gdrifts, uint32_t *assignments, float *bounds, uint32_t *counters,
                              hipStream_t st);

// transpose.hip (reference: transpose.cu:16-54)
hipError_t launch_transpose(const float *in, uint32_t rows, uint32_t cols, float *out, hipStream_t st);

}  // namespace kmx
