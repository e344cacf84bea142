// engine.hpp -- per-GPU workspace + step drivers (host side of the kernels).
// Reference counterpart: the allocations of src/kmcuda.cc:423-470 and the per-iteration launch
// code of src/kmeans.cu:934-1263, which keep everything replicated on every GPU; here one Engine
// owns one GPU's row shard.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "kernels.hpp"

namespace kmx {

enum Result {  // == KMCUDAResult (include/kmcuda.h)
  kSuccess = 0, kInvalidArguments, kNoSuchDevice, kMemoryAllocationFailure, kRuntimeError, kMemoryCopyError
};

#define KMX_HIP(call, ret)                                                                   \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      if (kmx::g_verbosity > 0)                                                              \
        printf("%s:%d -> %s (%s)\n", __FILE__, __LINE__, hipGetErrorString(e__), #call);     \
      return ret;                                                                            \
    }                                                                                        \
  } while (0)

extern int g_verbosity;

// Non-blocking streams of the library's own making are kept for the process: creating one and launching on it for
// the first time costs ~5 ms (kmeans_cuda() on 100 000 x 256 rows is 13 ms of work), so an engine takes an idle
// one from a per-device pool and hands it back, drained, when it goes.
hipStream_t pooled_stream_acquire(int device);
void pooled_stream_release(int device, hipStream_t s);

// The host's part in the carried-bounds passes (lloyd_carry.hip): pure bookkeeping over what the device REPORTS, one or
// two passes late, through a pinned pair of words (the length of a pass's row list and the pass's sequence number).
// Only speed depends on it -- the device-side list decides what a pass looks at -- but a wrong judgement here can switch
// the bounds off for a whole run (round 4: a stale report judged after a pause did).  kmamd_carry_policy_sim replays
// it on the CPU (tests/test_boundary_cpu.py).
struct CarryPolicy {
  float list_max = 0.5f;   // a listed pass when at most this share of the rows is on the list (KMCUDA_AMD_CARRY_MAX)
  // A counted list beyond that share makes the next pass a whole one too -- the skip kernel and the bounds' bookkeeping
  // for nothing -- so two such counts in a row pause the bounds: `backoff` plain passes (4, doubling up to 32), then
  // they are tried again.  (Round 4 only gave up beyond 90 %: config B's lists hover at 65-80 % from iteration 27 on and
  // twenty whole passes paid 0.5 ms each for bounds that spared nothing, profiles/r5ac_carry_trace_config_b.log.)
  uint32_t pause = 0, backoff = 4, hopeless = 0, seen_seq = 0;
  static constexpr uint32_t kNoList = 0xFFFFFFFFu;   // the report of a pass that had no list to count
  // When pausing is not enough.  Uniform rows (BASELINE config B) never get short lists: 98 % -> 92 % -> 81 % of the rows
  // over twenty iterations, while the reassignments fall 1.1x per iteration; the doubling pauses above still probed
  // thirteen times in its 46 iterations (round 5: 0.236 s against 0.231 for yinyang_t = 0).  So: a SECOND pause in a row
  // whose hopeless list is barely shorter than the one that started the first (> 90 % of it), in a run whose
  // reassignment count falls by less than a third per iteration (note_changed: the counts the host judges anyway), is
  // a long one (kSlowPause).  Both conditions are needed -- a run can converge slowly with short lists (a mixture's
  // tail: never paused), and right behind the hand-over point a mixture lists every row for a pass or two (large
  // drifts) before its lists collapse: its second episode is nothing like its first.  (Tried first: the long pause on
  // the first hopeless count of any slowly converging run -- it switched the bounds off for a whole angular mixture run
  // whose early reassignments fall slowly, tests/test_gpu_carry.py.)
  static constexpr uint32_t kSlowPause = 64;
  uint32_t changed_prev = 0, changed_last = 0, episode_list = 0;
  void note_changed(uint32_t changed) {
    changed_prev = changed_last;
    changed_last = changed;
  }
  bool converging_slowly() const {
    return changed_prev != 0 && changed_last != 0 && (float)changed_prev < 1.5f * (float)changed_last;
  }

  // A pass is about to run with carrying switched on: true = it runs plain (one pass of a pause is used up).
  bool paused() {
    if (pause == 0) return false;
    pause--;
    return true;
  }
  // A pass whose bounds are valid and have been moved once.  (last, last_seq): whatever report has landed; this_seq: the
  // sequence number this pass will report under.  Returns true for a LISTED pass (the list an earlier pass counted is
  // short enough), false for a whole pass that only counts its would-be list.  A report is judged once, by sequence
  // number, and never across a pause: what the passes before it reported describes drifts the pause was the answer to.
  bool decide(uint32_t last, uint32_t last_seq, uint32_t this_seq, uint32_t n_rows) {
    const bool listed = last != kNoList && (float)last <= list_max * (float)n_rows;
    if (last != kNoList && (int32_t)(last_seq - seen_seq) > 0) {
      seen_seq = last_seq;
      const float give_up = list_max > 0.f && list_max < 0.9f ? list_max : 0.9f;
      if ((float)last > give_up * (float)n_rows && list_max < 1.0f) {
        if (++hopeless >= 2) {
          const bool flat = episode_list != 0 && (float)last > 0.9f * (float)episode_list;
          pause = flat && converging_slowly() ? kSlowPause : backoff;
          episode_list = last;
          backoff = backoff < 32 ? 2 * backoff : 32;
          hopeless = 0;
          seen_seq = this_seq;
        }
      } else {
        hopeless = 0;
        backoff = 4;
        episode_list = 0;
      }
    }
    return listed;
  }
};

class Engine {
 public:
  Engine() = default;
  ~Engine();
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;

  int init(int device, uint32_t n_rows, uint32_t D, uint32_t K, int metric, int fp16x2, hipStream_t stream);

  // steps (all async on stream_)
  int lloyd_assign(const float *samples, const float *centroids, uint32_t *assignments,
                   uint32_t *assignments_prev, bool exact_only);
  // tail: the fused reduce buffer's [dcount | counters] behind delta (or null); dcount may be null then
  int move_deltas(const float *samples, const uint32_t *prev, const uint32_t *cur, double *delta, int32_t *dcount,
                  double *tail);
  // stop_threshold >= 0: the reference's stop rule decided on the device from the reduced counters behind
  // dcount_d (kernels.hpp: StopCtl); host_tail: 6 pinned words the kernel reports to ([0..3] counters, [4]
  // stopped, [5] seq)
  int apply_delta(const double *delta, const int32_t *dcount, const double *dcount_d, float *centroids,
                  uint32_t *ccounts, float stop_threshold = -1.f, bool report = false, uint32_t seq = 0);
  // the same update with the NEXT lloyd_assign's preparation fused behind it (L2 metric in the two-stage filter's
  // steady state; otherwise plain apply_delta): the caller must leave `centroids` alone until that call
  int apply_prepare(const double *delta, const double *dcount_d, float *centroids, uint32_t *ccounts,
                    float stop_threshold = -1.f, bool report = false, uint32_t seq = 0);
  bool steady_state(bool exact_only) const;
  int stop_ctl(float stop_threshold, bool report, uint32_t seq, StopCtl *ctl);
  const float *prepared_for_ = nullptr;   // the centroid buffer apply_prepare() has prepared the next pass for
  int stop_clear();   // lowers the device-side stop flag (start of a run)
  // the outcome of the apply_delta(..., report = true, seq) call: waits for THAT call only (an event behind it on
  // stream_), then [0..3] the reduced counters, [4] stopped, [5] seq
  int stop_report(uint32_t seq, uint32_t *host_out6);
  int adjust_exact(const float *samples, const uint32_t *prev, const uint32_t *cur, float *centroids,
                   uint32_t *ccounts);
  int prepare_centroids(const float *centroids);
  // Yinyang steps (reference: kmeans.cu:431-672); yy_configure uploads the centroid -> group map
  int yy_configure(uint32_t G, const uint32_t *groups_host);
  int yy_init(const float *samples, const float *centroids, const uint32_t *assignments, float *bounds);
  int yy_drifts(const float *centroids, float *drifts, float *gdrifts);
  int yy_filters(const float *samples, const float *centroids, const float *drifts, const float *gdrifts,
                 uint32_t *assignments, uint32_t *assignments_prev, float *bounds, uint32_t *passed);
  int counters_read(uint32_t *host4);
  int counters_reset(int which);
  int sync();

  // Device memory of the engine's lifetime.  Small buffers (most of them: a job has ~40, the nested job that
  // clusters the centroids into Yinyang groups has nothing else) are carved out of 4-MB slabs -- a hipMalloc costs
  // 0.1-0.2 ms whatever its size, which was 5 ms of set-up per job --, 256-byte aligned; large ones are their own.
  template <typename T>
  int alloc(T **p, size_t count) {
    void *q = alloc_bytes(count ? count * sizeof(T) : sizeof(T));
    if (!q) return kMemoryAllocationFailure;
    *p = static_cast<T *>(q);
    return kSuccess;
  }
  void *alloc_bytes(size_t bytes);
  // pinned words of the engine's lifetime, from ONE pinned block (device-visible, coherent)
  uint32_t *pinned_words(size_t n, uint32_t **dev_addr);

  int device_ = -1;
  hipStream_t stream_ = nullptr;
  bool own_stream_ = false, blocking_stream_ = false;
  hipStream_t side_stream_ = nullptr;   // the full-scan refine kernel beside the pair kernel
  bool own_side_stream_ = false;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  hipEvent_t ev_rows_ = nullptr;        // csqr / ct ready on the side stream (steady-state preparation)
  uint32_t N_ = 0, D_ = 0, K_ = 0, K_pad_ = 0, Kt_ = 0, DP_ = 0;
  int metric_ = 0, fp16x2_ = 0;
  // KMCUDA_AMD_FP16_STRICT (set by kmeans_cuda for an fp16x2 job): every step in the reference's half2
  // arithmetic (half2_strict.hip) instead of the fp32 arithmetic on the half values
  bool strict_h2_ = false;
  float *h2_sq_ = nullptr;   // K x 2 half2 squared norms
  float eps_ = 0, tie_slack_ = 0;

  // assignment workspace
  float *csqr_ = nullptr, *bias_ = nullptr, *bias2_ = nullptr, *cfil_ = nullptr, *ct_ = nullptr, *mu_ = nullptr;
  uint32_t *finite_ = nullptr;
  // fp16x2 path: the local rows as halves (caller-owned) + the hi/lo-split centred centroid panel
  const void *half_rows_ = nullptr;
  void *panelhi_ = nullptr;
  uint32_t *undecided_ = nullptr;
  float *und_thr_ = nullptr;      // per undecided row: coarse scores below it are ruled out
  uint32_t *duo_ = nullptr;       // 4 N: stage 1's rows with two contenders known by index (counters_[kDuoCount] of them)
  bool duo_on_ = true;            // KMCUDA_AMD_DUO=0: every undecided row takes stage 2's sweep (the A/B)
  bool duo_always_ = false;       // KMCUDA_AMD_DUO=2: the duo list whatever the lists' lengths (the tests); default: when it pays
  // 0: two-stage f16 matrix-core filter (hi.hi, then the contenders in fp32; default),
  // 1: f32 matrix-core filter (KMCUDA_AMD_FILTER=f32; cross-check)
  int filter_mode_ = 0;
  bool settle_ = true;   // KMCUDA_AMD_SETTLE=0: lloyd_pair + lloyd_exact instead of the one-launch lloyd_settle
  // D beyond the register-resident filters (lloyd_wide.hip): both operands streamed through LDS.  wide_dp_ = D rounded
  // up to 64 (0: not this path); KMCUDA_AMD_WIDE=0 leaves such shapes to the exact kernels (the cross-check)
  uint32_t wide_dp_ = 0;
  bool wide_ok_ = true;
  bool wide_failed_ = false;   // its buffers could not be allocated: the exact kernels serve the shape
  void *wide_rows16_ = nullptr;          // N x wide_dp_ halves: x - mu, row-major (this path's row cache)
  float *wide_meta_ = nullptr;       // 4 floats per row
  uint32_t *wide_cont_ = nullptr; // per listed row: the number of its contenders, then up to 16 of them
  int lloyd_assign_wide(const LloydArgs &a, const float *centroids, bool steady, bool rows_on_side, bool carry_was_valid);
  // row cache of the coarse filter stage (lloyd_f16.hip: row_cache_kernel); set_row_cache()
  bool row_cache_allowed_ = true;   // KMCUDA_AMD_ROW_CACHE=0 vetoes it
  bool row_cache_on_ = false, row_cache_valid_ = false, mu_frozen_ = false;
  void *xcache_ = nullptr;
  float *xmeta_ = nullptr;
  // Bounds carried from pass to pass (lloyd_carry.hip; kmamd_set_carry): per-row upper / lower distance bounds read
  // off the coarse stage's best two scores, moved by the centroids' drifts, sparing the rows they still decide.
  // (Angular metric: one number per row, the certified gap of the scores, in ub_; the centroids' bias changes behind
  // the drifts in drift_.)  Two-stage filter with a valid row cache only; any other state runs plain passes.
  bool carry_on_ = false;        // the caller wants it
  bool carry_valid_ = false;     // ub_ / lb_ describe the assignments and the centroids of the last pass
  uint32_t carry_preps_ = 0;     // centroid preparations since the last pass (exactly 1: drift_ is that update's)
  uint32_t carry_seq_ = 0;
  CarryPolicy carry_policy_;
  float *ub_ = nullptr, *lb_ = nullptr, *drift_ = nullptr;
  // the pair certificates (CarryArgs::l3 / p1 / p2).  KMCUDA_AMD_CARRY_PAIRS=0: without (A/B, tests)
  bool carry_pairs_ = true;
  float *l3_ = nullptr;
  uint32_t *p1_ = nullptr, *p2_ = nullptr;
  uint32_t *carry_list_ = nullptr;
  uint32_t *host_carry_ = nullptr, *host_carry_dev_ = nullptr;   // 2 pinned words: [0] the last list's length, [1] seq
  bool carry_usable() const;     // the state in which a pass can carry bounds
  int carry_stats(unsigned long long *rows_spared, uint32_t *last_list);
  int carry_pair_stats(unsigned long long *rows_paired);
  int duo_rows(uint32_t *rows);   // the last assignment pass's duo list (lloyd_duo.hip)
  // stats_: the ACTIVE half of a double-buffered 2 x 8 words (stats_base_): every preparation flips to
  // the other half, which the invariant keeps zero (memset, or zeroed by centroid_prep_frozen_kernel)
  uint32_t *stats_base_ = nullptr, *stats_ = nullptr, *flagged_ = nullptr, *pairs_ = nullptr, *counters_ = nullptr;
  // update workspace
  uint32_t *keys_tmp_ = nullptr, *vals_tmp_ = nullptr, *keys_sorted_ = nullptr, *rows_sorted_ = nullptr,
           *offsets2_ = nullptr, *move_blocks_ = nullptr, *bucket_work_ = nullptr, *bucket_rows_ = nullptr;
  uint32_t bucket_cap_ = 0;               // rows per (centroid, sign) bucket of the update's direct path
  uint32_t last_undecided_ = 0xFFFFFFFFu;     // an earlier pass's undecided rows (sizes stage 2's grid)
  MoveState ms_;                          // the update's host-side state (update.hip: launch_move_deltas)
  uint32_t *host_move_count_ = nullptr;   // 4 pinned words: [0] events [1] largest bucket [2] undecided rows
  void *sort_temp_ = nullptr;
  size_t sort_temp_bytes_ = 0;
  // Yinyang
  uint32_t G_ = 0, nslots_ = 0;
  bool yy_exact_ = false;  // KMCUDA_AMD_YY_EXACT=1: plain exact kernels (cross-check)
  bool yy_hint_ = true;    // KMCUDA_AMD_YY_HINT=0: local filter without the second-best estimate (yinyang_hint.hip)
  uint32_t *gfirst_ = nullptr, *gsecond_ = nullptr, *yy_flag_rows_ = nullptr;
  float *yy_hint_buf_ = nullptr;
  void *yy_panelhi_ = nullptr;
  int yy_hint_stats(uint32_t *host6);
  uint32_t *groups_ = nullptr, *cperm_ = nullptr, *gstart_ = nullptr, *pids_ = nullptr, *pmeta_ = nullptr;
  float *pfil_ = nullptr, *pbias_ = nullptr, *xt_ = nullptr;
  float *exact_work_ = nullptr;  // adjust_exact scratch when 64 centroid rows exceed LDS (lazy)
  uint32_t *host_counters_ = nullptr;  // pinned
  uint32_t *host_report_ = nullptr;    // pinned, 2 slots x 8 words: what apply_delta_kernel reports (StopCtl)
  uint32_t *host_report_dev_ = nullptr;
  hipEvent_t ev_report_[2] = {nullptr, nullptr};
  uint32_t *yy_stats_ = nullptr;       // 64 x 16 words: the hinted local filter's statistics, striped

  // profiling of the step kernels with HIP events on stream_
  bool profile_ = false;
  double filter_ms_ = 0, exact_ms_ = 0, update_ms_ = 0, coarse_ms_ = 0;
  uint32_t filter_launches_ = 0;
  struct Span { hipEvent_t a, b; int kind; };
  std::vector<Span> spans_;
  std::vector<size_t> open_spans_;
  void span_begin(int kind);
  void span_end();
  void profile_collect();
  void profile_reset();

 private:
  std::vector<void *> owned_;
  static constexpr size_t kSlabBytes = 4u << 20, kSlabMaxItem = 512u << 10;
  char *slab_ = nullptr;
  size_t slab_used_ = kSlabBytes;
  static constexpr size_t kPinnedWords = 64;
  uint32_t *pinned_ = nullptr, *pinned_dev_ = nullptr;
  size_t pinned_used_ = 0;
};

}  // namespace kmx
