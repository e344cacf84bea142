// engine.cpp -- Engine implementation + the step-level C ABI of include/kmcuda_amd.h.
#include "engine.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/kmcuda_amd.h"

namespace kmx {

int g_verbosity = 0;

namespace {
std::mutex g_stream_pool_mutex;
std::vector<std::vector<hipStream_t>> g_stream_pool;   // [device] -> idle streams
}  // namespace

hipStream_t pooled_stream_acquire(int device) {
  {
    std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
    if ((size_t)device < g_stream_pool.size() && !g_stream_pool[device].empty()) {
      hipStream_t s = g_stream_pool[device].back();
      g_stream_pool[device].pop_back();
      return s;
    }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return s;
}
void pooled_stream_release(int device, hipStream_t s) {
  if (!s) return;
  if (hipStreamSynchronize(s) != hipSuccess) {   // (a stream in an error state is not worth keeping)
    (void)hipGetLastError();
    (void)hipStreamDestroy(s);
    return;
  }
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  if (g_stream_pool.size() <= (size_t)device) g_stream_pool.resize(device + 1);
  if (g_stream_pool[device].size() < 32) g_stream_pool[device].push_back(s);
  else (void)hipStreamDestroy(s);
}

void *Engine::alloc_bytes(size_t bytes) {
  const size_t rounded = (bytes + 255u) & ~(size_t)255u;
  if (rounded <= kSlabMaxItem) {
    if (slab_used_ + rounded > kSlabBytes) {
      void *q = nullptr;
      if (hipMalloc(&q, kSlabBytes) != hipSuccess) return nullptr;
      owned_.push_back(q);
      slab_ = static_cast<char *>(q);
      slab_used_ = 0;
    }
    void *r = slab_ + slab_used_;
    slab_used_ += rounded;
    return r;
  }
  void *q = nullptr;
  if (hipMalloc(&q, bytes) != hipSuccess) return nullptr;
  owned_.push_back(q);
  return q;
}

uint32_t *Engine::pinned_words(size_t n, uint32_t **dev_addr) {
  if (!pinned_) {
    if (hipHostMalloc(reinterpret_cast<void **>(&pinned_), kPinnedWords * sizeof(uint32_t), hipHostMallocCoherent) != hipSuccess) {
      pinned_ = nullptr;
      return nullptr;
    }
    memset(pinned_, 0, kPinnedWords * sizeof(uint32_t));
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, pinned_, 0) != hipSuccess || dp == nullptr) {
      // (no half-initialised block: a later call must not hand out host words with a null device address)
      (void)hipHostFree(pinned_);
      pinned_ = nullptr;
      pinned_dev_ = nullptr;
      return nullptr;
    }
    pinned_dev_ = static_cast<uint32_t *>(dp);
  }
  n = (n + 3u) & ~(size_t)3u;   // 16-byte granules
  if (pinned_used_ + n > kPinnedWords) return nullptr;
  uint32_t *r = pinned_ + pinned_used_;
  if (dev_addr) *dev_addr = pinned_dev_ + pinned_used_;
  pinned_used_ += n;
  return r;
}

Engine::~Engine() {
  if (device_ >= 0) (void)hipSetDevice(device_);
  for (auto &s : spans_) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (void *p : owned_) (void)hipFree(p);
  if (pinned_) (void)hipHostFree(pinned_);
  if (side_stream_ && own_side_stream_) pooled_stream_release(device_, side_stream_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  for (hipEvent_t e : ev_report_)
    if (e) (void)hipEventDestroy(e);
  if (ev_rows_) (void)hipEventDestroy(ev_rows_);
  if (own_stream_ && stream_) {
    if (blocking_stream_) (void)hipStreamDestroy(stream_);
    else pooled_stream_release(device_, stream_);
  }
}

int Engine::init(int device, uint32_t n_rows, uint32_t D, uint32_t K, int metric, int fp16x2, hipStream_t stream) {
  if (getenv("KMCUDA_AMD_DEBUG")) g_verbosity = atoi(getenv("KMCUDA_AMD_DEBUG"));
  if (const char *f = getenv("KMCUDA_AMD_FILTER")) filter_mode_ = strcmp(f, "f32") == 0 ? 1 : 0;
  if (const char *c = getenv("KMCUDA_AMD_ROW_CACHE")) row_cache_allowed_ = atoi(c) != 0;
  // Pair certificates (lloyd_carry.hip): on under both metrics; KMCUDA_AMD_CARRY_PAIRS=0 is the A/B.  (Round 5 had the
  // angular ones off: a randomised whole call -- 300 000 x 16 half rows, K = 130 -- ended on other centroids with them
  // than with plain passes.  Cause, DESIGN_LOG 13.13: the reference's distance is p >= 1 ? 0 : acos(p), centroids whose
  // products with a row reach 1 tie and the lowest index wins; the pair kernel's exact arithmetic followed that, the
  // filters of the plain passes committed the largest product -- and which rows went which way depended on the host's
  // timing-dependent list reports, hence run-to-run differences.  Round 6: every filter leaves such rows to the exact
  // kernels (filter_common.hpp: clamp_limits), so every path gives the reference's answer.)
  carry_pairs_ = true;
  if (const char *c = getenv("KMCUDA_AMD_CARRY_PAIRS")) carry_pairs_ = atoi(c) != 0;
  if (const char *c = getenv("KMCUDA_AMD_SETTLE")) settle_ = atoi(c) != 0;
  if (const char *c = getenv("KMCUDA_AMD_DUO")) { duo_on_ = atoi(c) != 0; duo_always_ = atoi(c) == 2; }
  if (const char *c = getenv("KMCUDA_AMD_WIDE")) wide_ok_ = atoi(c) != 0;
  if (const char *c = getenv("KMCUDA_AMD_GEMM")) wide_ok_ = atoi(c) != 0;   // (the switch's name while stage 1 was a library GEMM)
  if (const char *u = getenv("KMCUDA_AMD_UPDATE"))
    ms_.force = strcmp(u, "radix") == 0 ? 1 : (strcmp(u, "sync") == 0 ? 2 : (strcmp(u, "bucket") == 0 ? 3 : 0));
  if (D == 0 || K < 1 || K >= 0x7FFFFFFFu) return kInvalidArguments;  // K == 1: Yinyang group clustering with one group
  if (fp16x2) return kInvalidArguments;  // fp16x2 kernels are not built yet ("next")
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return kNoSuchDevice;
  KMX_HIP(hipSetDevice(device), kNoSuchDevice);
  device_ = device;
  if (stream == (hipStream_t)(intptr_t)-1) {
    // the caller works on the legacy default (NULL) stream -- torch's default: an own BLOCKING stream
    // orders with it implicitly in both directions (a non-blocking one would race with that work)
    KMX_HIP(hipStreamCreateWithFlags(&stream_, hipStreamDefault), kRuntimeError);
    own_stream_ = true;
    blocking_stream_ = true;
  } else if (stream) {
    stream_ = stream;
  } else {
    stream_ = pooled_stream_acquire(device);
    if (!stream_) return kRuntimeError;
    own_stream_ = true;
  }
  N_ = n_rows; D_ = D; K_ = K; metric_ = metric; fp16x2_ = fp16x2;
  K_pad_ = (K + 31) / 32 * 32;
  Kt_ = (K + 63) / 64 * 64;
  DP_ = lloyd_dp_for(D);   // 0 beyond 512 features: no register-resident instantiation
  // 257..512 features: the register-resident filter exists (one operand set per wave, rows padded to 512) but the
  // LDS-streamed one of lloyd_wide.hip (rows padded to 64) is 14-33 % faster per plain pass (4M rows @ 1024: 512
  // features 7.52 -> 6.49 ms, 384 6.95 -> 4.96, 320 5.88 -> 3.96; at <= 256 features the register-resident one wins,
  // 2.20 against 3.17 ms: profiles/r5s_*, r5t_*) and carries bounds as well (whole calls on 2M-row mixtures at
  // tolerance 1e-4: 384 features 0.041 s against 0.048 s with the bounds on the register-resident filter, 512 features
  // 0.048 against 0.053: profiles/r5aj_*).  KMCUDA_AMD_WIDE_MIN_D=d moves the border: rows of at least d features are
  // streamed (513: the register-resident filter up to 512 features, the cross-check).
  long wide_min_d = 257;
  if (const char *c = getenv("KMCUDA_AMD_WIDE_MIN_D")) {
    const long d = atol(c);
    if (d > 0) wide_min_d = d;
  }
  if (wide_ok_ && (long)D >= wide_min_d) DP_ = 0;
  // Filter error bound coefficient (DESIGN.md "error bound"): gamma_D + (kappa + 3) u with
  // u = 2^-24, gamma_D <= 1.01 D u, kappa = 8 for the reference's Kahan chain; +2% margin.
  eps_ = (float)(1.02 * ((double)D + 12.0) * ldexp(1.0, -24));
  // Angular metric: acosf is many-to-one; products within ~2.4e-7 of each other may map to the
  // same angle, so near-ties go to the exact kernel which applies acosf like the reference.
  tie_slack_ = metric == 0 ? 0.f : 1e-6f;

  // no register-resident filter for this D: stage 1 streams both operands through LDS (lloyd_wide.hip), in 64-feature
  // chunks: operands padded to 64
  wide_dp_ = (DP_ == 0 && wide_ok_) ? (D + 63) / 64 * 64 : 0;
  const uint32_t dp = DP_ ? DP_ : (wide_dp_ ? wide_dp_ : 8);
  int rc;
  if ((rc = alloc(&csqr_, K))) return rc;
  if ((rc = alloc(&bias_, K_pad_))) return rc;
  if ((rc = alloc(&bias2_, K_pad_))) return rc;
  if ((rc = alloc(&cfil_, (size_t)K_pad_ * dp))) return rc;
  if ((rc = alloc(&ct_, (size_t)D * Kt_))) return rc;
  if ((rc = alloc(&stats_base_, 16))) return rc;
  stats_ = stats_base_;
  if ((rc = alloc(&mu_, dp))) return rc;
  if ((rc = alloc(&finite_, Kt_))) return rc;
  if ((rc = alloc(&flagged_, n_rows))) return rc;
  if ((rc = alloc(&pairs_, 3 * (size_t)n_rows))) return rc;
  if ((rc = alloc(&counters_, 16))) return rc;
  if ((rc = alloc(&keys_tmp_, 2 * (size_t)n_rows))) return rc;
  if ((rc = alloc(&vals_tmp_, 2 * (size_t)n_rows))) return rc;
  if ((rc = alloc(&keys_sorted_, 2 * (size_t)n_rows))) return rc;
  if ((rc = alloc(&rows_sorted_, 2 * (size_t)n_rows))) return rc;
  if ((rc = alloc(&offsets2_, 2 * (size_t)K + 2))) return rc;
  if ((rc = alloc(&move_blocks_, (size_t)n_rows / 1024 + 4))) return rc;
  if ((rc = alloc(&bucket_work_, move_bucket_words(K)))) return rc;
  KMX_HIP(hipMemsetAsync(bucket_work_, 0, move_bucket_words(K) * sizeof(uint32_t), stream_), kRuntimeError);
  bucket_cap_ = move_bucket_cap(n_rows, K);
  if ((rc = alloc(&bucket_rows_, 2 * (size_t)K * bucket_cap_))) return rc;
  KMX_HIP(hipMemsetAsync(stats_base_, 0, 16 * sizeof(uint32_t), stream_), kRuntimeError);
  host_move_count_ = pinned_words(5, &ms_.host_dev);
  if (!host_move_count_) return kMemoryAllocationFailure;
  host_move_count_[2] = 0xFFFFFFFFu;   // undecided rows: not known yet
  host_move_count_[4] = 0xFFFFFFFFu;   // duo rows: likewise
  ms_.host = host_move_count_;
  KMX_HIP(hipEventCreateWithFlags(&ev_rows_, hipEventDisableTiming), kRuntimeError);
  sort_temp_bytes_ = sort_temp_bytes(2 * (size_t)n_rows, 2 * K);
  {
    const size_t b2 = sort_temp_bytes(n_rows, K);
    if (b2 > sort_temp_bytes_) sort_temp_bytes_ = b2;
  }
  {
    char *t = nullptr;
    if ((rc = alloc(&t, sort_temp_bytes_ + 16))) return rc;
    sort_temp_ = t;
  }
  host_counters_ = pinned_words(8, nullptr);
  if (!host_counters_) return kMemoryAllocationFailure;
  KMX_HIP(hipMemsetAsync(counters_, 0, 16 * sizeof(uint32_t), stream_), kRuntimeError);
  return kSuccess;
}

void Engine::span_begin(int kind) {
  if (!profile_) return;
  Span s;
  (void)hipEventCreate(&s.a);
  (void)hipEventCreate(&s.b);
  s.kind = kind;
  (void)hipEventRecord(s.a, stream_);
  open_spans_.push_back(spans_.size());
  spans_.push_back(s);
}
void Engine::span_end() {  // closes the innermost open span (spans nest: the coarse kernel inside the filter)
  if (!profile_ || open_spans_.empty()) return;
  (void)hipEventRecord(spans_[open_spans_.back()].b, stream_);
  open_spans_.pop_back();
}
void Engine::profile_collect() {
  for (auto &s : spans_) {
    (void)hipEventSynchronize(s.b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s.a, s.b);
    if (s.kind == 0) { filter_ms_ += ms; filter_launches_++; }
    else if (s.kind == 1) exact_ms_ += ms;
    else if (s.kind == 3) coarse_ms_ += ms;
    else update_ms_ += ms;
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  spans_.clear();
  open_spans_.clear();
}
void Engine::profile_reset() {
  profile_collect();
  filter_ms_ = exact_ms_ = update_ms_ = coarse_ms_ = 0;
  filter_launches_ = 0;
}

int Engine::prepare_centroids(const float *centroids) {
  const uint32_t dp = DP_ ? DP_ : (wide_dp_ ? wide_dp_ : 8);
  prepared_for_ = nullptr;
  // this preparation's statistics go to the other half; both halves are zeroed here, which also leaves
  // the half after this one zero (the invariant centroid_prep_frozen_kernel relies on)
  stats_ = stats_ == stats_base_ ? stats_base_ + 8 : stats_base_;
  KMX_HIP(hipMemsetAsync(stats_base_, 0, 16 * sizeof(uint32_t), stream_), kRuntimeError);
  KMX_HIP(launch_centroid_prep(metric_, centroids, K_, D_, K_pad_, dp, Kt_, csqr_, bias_, bias2_, cfil_, ct_, mu_,
                               mu_frozen_, finite_, stats_, counters_ + 1, counters_ + 3, counters_ + 4, stream_),
          kRuntimeError);
  return kSuccess;
}

int Engine::yy_configure(uint32_t G, const uint32_t *groups_host) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  if (const char *v = getenv("KMCUDA_AMD_YY_EXACT")) yy_exact_ = atoi(v) != 0;
  if (const char *v = getenv("KMCUDA_AMD_YY_HINT")) yy_hint_ = atoi(v) != 0;
  G_ = G;
  // centroids in group order; group >= G (a NaN centroid keeps the 0xFFFFFFFF "assignment" of
  // its failed search, kmeans.cu:468-471) is left out
  std::vector<uint32_t> gstart(G + 1, 0), cperm;
  for (uint32_t c = 0; c < K_; c++)
    if (groups_host[c] < G) gstart[groups_host[c] + 1]++;
  for (uint32_t g = 0; g < G; g++) gstart[g + 1] += gstart[g];
  cperm.resize(gstart[G] ? gstart[G] : 1);
  {
    std::vector<uint32_t> fill(gstart.begin(), gstart.end() - 1);
    for (uint32_t c = 0; c < K_; c++)
      if (groups_host[c] < G) cperm[fill[groups_host[c]]++] = c;
  }
  // padded panel: every group gets whole 4-slot chunks (at least one, so that an empty group still
  // gets its FLT_MAX bound written), the panel whole 32-slot tiles.  A group of up to 8 chunks never
  // crosses a tile boundary (the tile is padded instead); a larger one starts a tile and carries a start
  // flag on the first chunk of every further tile: the kernels close a (part of a) group with contenders
  // from ONE tile, and the group's minimum is carried across its parts (yinyang_init.hip)
  std::vector<uint32_t> pids, pmeta;
  for (uint32_t g = 0; g < G; g++) {
    const uint32_t n = gstart[g + 1] - gstart[g];
    const uint32_t chunks = n ? (n + 3) / 4 : 1;
    const uint32_t used = (uint32_t)(pids.size() / 4) % 8;
    if (used && (chunks > 8 || used + chunks > 8)) {
      for (uint32_t ch = used; ch < 8; ch++) {
        pmeta.push_back((g ? g - 1 : 0) << 1);
        for (uint32_t q = 0; q < 4; q++) pids.push_back(0xFFFFFFFFu);
      }
    }
    for (uint32_t ch = 0; ch < chunks; ch++) {
      const bool tile_start = (pids.size() / 4) % 8 == 0;
      pmeta.push_back((g << 1) | ((ch == 0 || tile_start) ? 1u : 0u));
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t i = ch * 4 + q;
        pids.push_back(i < n ? cperm[gstart[g] + i] : 0xFFFFFFFFu);
      }
    }
  }
  while (pids.size() % 32) {
    if (pids.size() % 4 == 0) pmeta.push_back(((G ? G - 1 : 0) << 1));
    pids.push_back(0xFFFFFFFFu);
  }
  while (pmeta.size() < pids.size() / 4) pmeta.push_back(((G ? G - 1 : 0) << 1));
  nslots_ = (uint32_t)pids.size();
  // the two smallest member indices of every group: where the reference's ascending scan first meets it
  std::vector<uint32_t> gfirst(G ? G : 1, 0xFFFFFFFFu), gsecond(G ? G : 1, 0xFFFFFFFFu);
  for (uint32_t g = 0; g < G; g++) {
    if (gstart[g + 1] - gstart[g] >= 1) gfirst[g] = cperm[gstart[g]];       // cperm is ascending inside a group
    if (gstart[g + 1] - gstart[g] >= 2) gsecond[g] = cperm[gstart[g] + 1];
  }
  int rc;
  if ((rc = alloc(&gfirst_, gfirst.size()))) return rc;
  if ((rc = alloc(&gsecond_, gsecond.size()))) return rc;
  KMX_HIP(hipMemcpy(gfirst_, gfirst.data(), gfirst.size() * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  KMX_HIP(hipMemcpy(gsecond_, gsecond.data(), gsecond.size() * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  const uint32_t k_pad64 = (K_ + 63u) / 64u * 64u;   // whole 64-centroid super-tiles (yy_local_hint_kernel's DMA)
  if ((rc = alloc(&groups_, k_pad64))) return rc;
  if (!yy_stats_) {
    if ((rc = alloc(&yy_stats_, 64 * 16))) return rc;
    KMX_HIP(hipMemsetAsync(yy_stats_, 0, 64 * 16 * sizeof(uint32_t), stream_), kRuntimeError);
  }
  if ((rc = alloc(&cperm_, cperm.size()))) return rc;
  if ((rc = alloc(&gstart_, G + 1))) return rc;
  if ((rc = alloc(&pids_, pids.size()))) return rc;
  if ((rc = alloc(&pmeta_, pmeta.size()))) return rc;
  const uint32_t dp = DP_ ? DP_ : 8;
  if ((rc = alloc(&pfil_, (size_t)nslots_ * dp))) return rc;
  if ((rc = alloc(&pbias_, nslots_))) return rc;
  {
    std::vector<uint32_t> gp(k_pad64, 0xFFFFFFFFu);
    for (uint32_t c = 0; c < K_; c++) gp[c] = groups_host[c];
    KMX_HIP(hipMemcpy(groups_, gp.data(), k_pad64 * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  }
  KMX_HIP(hipMemcpy(cperm_, cperm.data(), cperm.size() * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  KMX_HIP(hipMemcpy(gstart_, gstart.data(), (G + 1) * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  KMX_HIP(hipMemcpy(pids_, pids.data(), pids.size() * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  KMX_HIP(hipMemcpy(pmeta_, pmeta.data(), pmeta.size() * sizeof(uint32_t), hipMemcpyHostToDevice), kMemoryCopyError);
  return kSuccess;
}

static void fill_yy_args(Engine &e, YyArgs &a, const float *samples, const float *centroids) {
  a.samples = samples; a.centroids = centroids;
  a.len = e.N_; a.D = e.D_; a.DP = e.DP_; a.K = e.K_; a.K_pad = e.K_pad_; a.G = e.G_;
  a.cfil = e.cfil_; a.bias = e.bias_; a.mu = e.mu_; a.stats = e.stats_; a.eps = e.eps_;
  a.groups = e.groups_; a.drifts = nullptr; a.gdrifts = nullptr; a.assignments = nullptr; a.bounds = nullptr;
  a.passed = nullptr; a.counters = e.counters_; a.count_ptr = e.counters_ + 2;
  a.panelhi = nullptr; a.hint = nullptr; a.flag_rows = nullptr; a.gfirst = e.gfirst_; a.gsecond = e.gsecond_;
  a.xcache = nullptr; a.xmeta = nullptr; a.stat_stripes = e.yy_stats_;
  a.pfil = e.pfil_; a.pbias = e.pbias_; a.pids = e.pids_; a.pmeta = e.pmeta_; a.cperm = e.cperm_;
  a.gstart = e.gstart_; a.nslots = e.nslots_;
}

int Engine::yy_init(const float *samples, const float *centroids, const uint32_t *assignments, float *bounds) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  if (N_ == 0) return kSuccess;
  if (strict_h2_) {
    KMX_HIP(launch_h2_yy_init(metric_, samples, N_, D_, K_, G_, centroids, assignments, groups_, bounds, stream_),
            kRuntimeError);
    return kSuccess;
  }
  if (DP_ && DP_ <= 256 && !yy_exact_) {
    int rc = prepare_centroids(centroids);
    if (rc) return rc;
    YyArgs a;
    fill_yy_args(*this, a, samples, centroids);
    a.assignments = const_cast<uint32_t *>(assignments);
    a.bounds = bounds;
    KMX_HIP(launch_yy_orig_panel(metric_, centroids, D_, DP_, pids_, nslots_, pfil_, pbias_, stream_), kRuntimeError);
    KMX_HIP(launch_yy_init_lds(metric_, a, stream_), kRuntimeError);
    return kSuccess;
  }
  if (!xt_) {
    int rc = alloc(&xt_, (size_t)N_ * D_);
    if (rc) return rc;
  }
  KMX_HIP(launch_transpose(samples, N_, D_, xt_, stream_), kRuntimeError);
  KMX_HIP(launch_yy_init(metric_, xt_, N_, D_, G_, centroids, assignments, cperm_, gstart_, bounds, stream_),
          kRuntimeError);
  return kSuccess;
}

int Engine::yy_drifts(const float *centroids, float *drifts, float *gdrifts) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  if (strict_h2_) {   // the per-centroid drifts in half2 arithmetic; the group maxima are plain fp32 compares
    KMX_HIP(launch_h2_yy_drifts(metric_, centroids, K_, D_, drifts, stream_), kRuntimeError);
    KMX_HIP(launch_yy_group_max(K_, D_, G_, groups_, drifts, gdrifts, stream_), kRuntimeError);
    return kSuccess;
  }
  KMX_HIP(launch_yy_drifts(metric_, centroids, K_, D_, G_, groups_, drifts, gdrifts, stream_), kRuntimeError);
  return kSuccess;
}

int Engine::yy_filters(const float *samples, const float *centroids, const float *drifts, const float *gdrifts,
                       uint32_t *assignments, uint32_t *assignments_prev, float *bounds, uint32_t *passed) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  carry_valid_ = false;   // (the reference's filters move assignments behind the carried bounds' back)
  if (N_ == 0) return kSuccess;
  if (strict_h2_) {
    KMX_HIP(launch_h2_yy_global(metric_, samples, N_, D_, K_, G_, centroids, drifts, gdrifts, assignments,
                                assignments_prev, bounds, passed, counters_, stream_),
            kRuntimeError);
    KMX_HIP(launch_h2_yy_local(metric_, samples, N_, D_, K_, G_, passed, centroids, groups_, drifts, gdrifts,
                               assignments, bounds, counters_, stream_),
            kRuntimeError);
    return kSuccess;
  }
  KMX_HIP(launch_yy_global_filter(metric_, samples, N_, D_, K_, G_, centroids, drifts, gdrifts, assignments,
                                  assignments_prev, bounds, passed, counters_, stream_),
          kRuntimeError);
  if (DP_ && DP_ <= 256 && !yy_exact_) {
    int rc = prepare_centroids(centroids);
    if (rc) return rc;
    YyArgs a;
    fill_yy_args(*this, a, samples, centroids);
    a.drifts = drifts; a.gdrifts = gdrifts; a.assignments = assignments; a.bounds = bounds; a.passed = passed;
    if (yy_hint_ && yy_hint_supported(DP_)) {
      // second-best estimate per passed row, the local filter against it, the plain kernel for the rest
      if (!yy_hint_buf_) {
        uint16_t *phi = nullptr;
        if ((rc = alloc(&phi, (size_t)((K_pad_ + 63u) / 64u * 64u) * (DP_ + 2)))) return rc;
        if ((rc = alloc(&yy_flag_rows_, N_))) return rc;
        if ((rc = alloc(&yy_hint_buf_, N_))) return rc;
        yy_panelhi_ = phi;
      }
      KMX_HIP(launch_centroid_panelhi(centroids, K_, D_, K_pad_, DP_, finite_, mu_, bias_, yy_panelhi_, stats_,
                                      stream_),
              kRuntimeError);
      KMX_HIP(hipMemsetAsync(counters_ + 5, 0, sizeof(uint32_t), stream_), kRuntimeError);
      a.panelhi = yy_panelhi_; a.hint = yy_hint_buf_; a.flag_rows = yy_flag_rows_;
      // the coarse Lloyd stage's row cache holds exactly the sweeps' B operands when it was built for this mean
      if (row_cache_on_ && row_cache_valid_ && mu_frozen_ && xcache_ && metric_ == 0 && D_ == DP_) {
        a.xcache = xcache_;
        a.xmeta = xmeta_;
      }
      KMX_HIP(launch_yy_hint(metric_, a, stream_), kRuntimeError);
      KMX_HIP(launch_yy_local_hint(metric_, a, stream_), kRuntimeError);
      a.passed = yy_flag_rows_;
      a.count_ptr = counters_ + 5;
    }
    KMX_HIP(launch_yy_local_mfma(metric_, a, stream_), kRuntimeError);
    return kSuccess;
  }
  KMX_HIP(launch_yy_local_filter(metric_, samples, N_, D_, K_, G_, centroids, groups_, drifts, gdrifts, assignments,
                                 bounds, passed, counters_, stream_),
          kRuntimeError);
  return kSuccess;
}

int Engine::lloyd_assign(const float *samples, const float *centroids, uint32_t *assignments,
                         uint32_t *assignments_prev, bool exact_only) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  // carried bounds describe the assignments of the LAST pass: whatever this pass turns out to be, they hold afterwards
  // only if it was a carried pass itself (set at its end) -- an exact pass, the f32 filter, a rebuilt panel void them
  const bool carry_was_valid = carry_valid_;
  carry_valid_ = false;
  if (strict_h2_) {
    if (!h2_sq_) {
      int rc = alloc(&h2_sq_, 2 * (size_t)K_);
      if (rc) return rc;
    }
    span_begin(1);
    KMX_HIP(launch_h2_csqr(metric_, centroids, K_, D_, h2_sq_, stream_), kRuntimeError);
    KMX_HIP(launch_h2_assign(metric_, samples, N_, D_, centroids, K_, h2_sq_, assignments, assignments_prev, counters_,
                             stream_),
            kRuntimeError);
    span_end();
    return kSuccess;
  }
  // row cache (two-stage f16 filter only): x - mu as halves in operand order, built on the first pass
  // after set_row_cache(1); mu is frozen from then on, so the copy stays valid for every later pass
  const bool two_stage = !exact_only && DP_ != 0 && filter_mode_ == 0 && lloyd_filter_f16_supported(D_, DP_);
  const bool want_cache = row_cache_on_ && two_stage && N_ != 0;
  const bool build_cache = want_cache && !row_cache_valid_;
  if (build_cache) mu_frozen_ = false;  // take the mean of THESE centroids
  if (!side_stream_) {
    // (non-blocking even beside a blocking main stream: fork / join events order it completely)
    side_stream_ = pooled_stream_acquire(device_);
    if (!side_stream_) return kRuntimeError;
    own_side_stream_ = true;
  }
  if (!ev_fork_) {   // (a borrowed side stream -- the nested group-clustering job's -- comes without events)
    KMX_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming), kRuntimeError);
    KMX_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming), kRuntimeError);
  }
  if (two_stage && !panelhi_) {
    uint16_t *phi = nullptr;
    int rc = alloc(&phi, (size_t)((K_pad_ + 63u) / 64u * 64u) * (DP_ + 2));  // whole 64-row super-tiles + their biases
    if (rc) return rc;
    if (!undecided_ && ((rc = alloc(&undecided_, N_)) || (rc = alloc(&und_thr_, N_)))) return rc;
    if (duo_on_ && !duo_ && alloc(&duo_, 4 * (size_t)N_) != kSuccess) {   // (an optimisation: without it stage 2 sweeps for every listed row)
      (void)hipGetLastError();
      duo_ = nullptr;
      duo_on_ = false;
    }
    panelhi_ = phi;
  }
  // Steady state of the two-stage filter (mean frozen, cache valid): ONE preparation kernel in front of
  // stage 1; the reference's serial sum_squares chain and the transposed panel, which only the pair /
  // exact kernels read, are computed beside stage 1 on the side stream.
  const bool steady = steady_state(exact_only);
  // (the streamed filter's steady state: the same ONE preparation kernel, operands padded to wide_dp_)
  const bool wide_sel = !exact_only && DP_ == 0 && wide_dp_ != 0 && !wide_failed_;
  const bool wide_steady = wide_sel && mu_frozen_ && row_cache_on_ && row_cache_valid_ && N_ != 0 && panelhi_ != nullptr &&
                           wide_rows16_ != nullptr && !strict_h2_;
  const bool prepared = steady && prepared_for_ == centroids;   // apply_prepare() has done this pass's preparation
  prepared_for_ = nullptr;
  bool rows_on_side = false;
  if (steady || wide_steady) {
    KMX_HIP(hipEventRecord(ev_fork_, stream_), kRuntimeError);   // the centroids are final here
    KMX_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0), kRuntimeError);
    KMX_HIP(launch_centroid_rows(metric_, centroids, K_, D_, Kt_, csqr_, ct_, side_stream_), kRuntimeError);
    KMX_HIP(hipEventRecord(ev_rows_, side_stream_), kRuntimeError);
    rows_on_side = true;
    if (!prepared) {
      uint32_t *next = stats_;
      stats_ = stats_ == stats_base_ ? stats_base_ + 8 : stats_base_;
      KMX_HIP(launch_centroid_prep_frozen(metric_, centroids, K_, D_, K_pad_, DP_ ? DP_ : wide_dp_, mu_, finite_, bias_,
                                          bias2_, cfil_, panelhi_, stats_, next, counters_ + 1, counters_ + 3, counters_ + 4,
                                          carry_on_ ? drift_ : nullptr, carry_on_ && drift_ ? counters_ + kCarryCursor : nullptr,
                                          stream_),
              kRuntimeError);
      carry_preps_++;
    }
  } else {
    int rc = prepare_centroids(centroids);
    if (rc) return rc;
  }
  LloydArgs a;
  a.samples = samples; a.N = N_; a.D = D_; a.K = K_; a.K_pad = K_pad_; a.DP = DP_; a.Kt = Kt_;
  a.cfil = cfil_; a.bias = bias_; a.mu = mu_; a.ct = ct_; a.csqr = csqr_; a.stats = stats_;
  a.eps = eps_; a.tie_slack = tie_slack_;
  a.assignments = assignments; a.assignments_prev = assignments_prev;
  a.flagged = flagged_; a.pairs = pairs_; a.counters = counters_;
  if (N_ == 0) return kSuccess;
  if (wide_sel) {
    const int rc = lloyd_assign_wide(a, centroids, wide_steady, rows_on_side, carry_was_valid);
    if (rc != kNoSuchDevice + 100) return rc;   // (that code: no memory for its buffers -- the exact kernel below serves the shape)
  }
  if (exact_only || DP_ == 0 || (DP_ > 256 && filter_mode_ != 0)) {
    span_begin(1);
    const uint32_t grid = N_ < 8192u ? N_ : 8192u;
    KMX_HIP(launch_lloyd_exact(metric_, a, nullptr, nullptr, grid, stream_), kRuntimeError);
    span_end();
    return kSuccess;
  }
  // counters_[1] / [3] / [4] (the filter's list lengths) were zeroed by the preparation
  span_begin(0);
  if (two_stage) {
    const void *rows = half_rows_ ? half_rows_ : (const void *)samples;
    const bool half = half_rows_ != nullptr;
    bool duo_listed = false;   // stage 1 wrote a duo list (plain passes)
    if (!steady)
      KMX_HIP(launch_centroid_panelhi(centroids, K_, D_, K_pad_, DP_, finite_, mu_, bias_, panelhi_, stats_, stream_),
              kRuntimeError);
    if (build_cache) {
      const size_t npad = ((size_t)N_ + 255) / 256 * 256;
      if (!xcache_) {
        uint16_t *xc = nullptr;
        float *xm = nullptr;
        // no memory for the copy: not an error, the operands are converted from the rows every pass
        if (alloc(&xc, npad * DP_) == kSuccess && alloc(&xm, npad * 3 + 2) == kSuccess) {
          xcache_ = xc;
          xmeta_ = xm;
        } else {
          (void)hipGetLastError();
          row_cache_on_ = false;
        }
      }
      if (xcache_) {
        KMX_HIP(launch_row_cache(rows, half, N_, D_, DP_, mu_, xcache_, xmeta_, stream_), kRuntimeError);
        row_cache_valid_ = true;
        mu_frozen_ = true;
      }
    }
    const bool cached = row_cache_on_ && row_cache_valid_;
    // Carried bounds (lloyd_carry.hip): in the steady state the pass can leave per-row distance bounds behind and the
    // next one only looks at the rows they do not decide.  The drift of this pass's centroids against the last pass's
    // is the preparation kernel's (exactly one preparation since: anything else voids the bounds).
    bool carry = carry_on_ && steady && cached;
    if (carry && carry_policy_.paused()) {   // the bounds decided next to nothing lately: plain passes for a while
      carry = false;
      if (getenv("KMCUDA_AMD_CARRY_TRACE")) fprintf(stderr, "[carry] paused (%u more)\n", carry_policy_.pause);
    }
    if (carry && !ub_) {
      // (no memory: not an error, plain passes)
      const bool pairs = carry_pairs_;   // (the pair certificates)
      if (alloc(&ub_, N_) != kSuccess || alloc(&lb_, N_) != kSuccess ||
          alloc(&drift_, 2 * (size_t)K_) != kSuccess ||   // (+ K bias changes: the angular metric)
          (pairs && (alloc(&l3_, N_) != kSuccess || alloc(&p1_, N_) != kSuccess || alloc(&p2_, N_) != kSuccess)) ||
          alloc(&carry_list_, N_) != kSuccess || !(host_carry_ = pinned_words(2, &host_carry_dev_))) {
        (void)hipGetLastError();
        carry_on_ = false;
        ub_ = nullptr;
        l3_ = nullptr;
      } else {
        host_carry_[0] = 0xFFFFFFFFu;
        host_carry_[1] = 0;
        if (const char *v = getenv("KMCUDA_AMD_CARRY_MAX")) carry_policy_.list_max = (float)atof(v);
      }
    }
    span_begin(3);  // the dominant kernel on its own, inside the filter span
    // The duo list pays when it takes whole ROUNDS of blocks off stage 2's sweep: a list that fits one round of its
    // 128-row blocks (two per CU) is one sweep long either way, and the second kernel only adds its launch.  By an
    // earlier pass's list lengths (whatever the update's report has delivered), the row count before any report.
    duo_listed = duo_on_ && duo_;
    if (duo_listed && !duo_always_) {
      const uint32_t und_prev = host_move_count_[2], duo_prev = host_move_count_[4];
      const uint64_t listed_rows = (und_prev != 0xFFFFFFFFu && duo_prev != 0xFFFFFFFFu) ? (uint64_t)und_prev + duo_prev
                                                                                          : (uint64_t)N_ / 16u;
      duo_listed = listed_rows > 512u * 128u;
    }
    CarryArgs cy_refine;   // (stage 2 leaves bounds only in a carried pass)
    if (carry && carry_on_) {
      CarryArgs cy;
      cy.ub = ub_; cy.lb = lb_; cy.host_report = host_carry_dev_; cy.seq = ++carry_seq_;
      cy.angular = metric_ != 0;
      cy.l3 = l3_; cy.p1 = p1_; cy.p2 = p2_;
      cy_refine = cy;
      const bool moved = carry_was_valid && carry_preps_ == 1;   // drift_ / stats_[6] belong to the bounds
      uint32_t hint = 0xFFFFFFFFu;
      bool listed = false;
      if (moved) {
        // what the host knows of an EARLIER pass's list (a pinned word the coarse kernel writes; only speed depends
        // on it): a short list -> the listed pass; else every row from the row cache, the list only counted
        const uint32_t last = host_carry_[0], last_seq = host_carry_[1];
        listed = carry_policy_.decide(last, last_seq, carry_seq_, N_);
        if (listed) hint = last;
        KMX_HIP(launch_carry_skip(N_, K_, assignments, assignments_prev, cy, xmeta_, drift_, stats_, tie_slack_,
                                  carry_list_, finite_, pairs_, counters_, !listed, stream_),
                kRuntimeError);
        cy.n_list = counters_ + kCarryCursor;
      }
      if (listed) cy.row_list = carry_list_;
      static const bool trace = getenv("KMCUDA_AMD_CARRY_TRACE") != nullptr;
      if (trace) {   // (debugging aid: synchronises)
        uint32_t w[8] = {0};
        (void)hipStreamSynchronize(stream_);
        (void)hipMemcpy(w, stats_, sizeof(w), hipMemcpyDeviceToHost);
        float f[8];
        memcpy(f, w, sizeof(f));
        fprintf(stderr, "[carry] stats: max ||c'||^2 %g, max |bias| %g, max ||c||^2 %g, max residual^2 %g, max drift %g, max bias change %g; "
                "stage 2 took %u rows of an earlier pass\n", f[0], f[1], f[2], f[5], f[6], f[7], host_move_count_[2]);
      }
      if (trace)
        fprintf(stderr, "[carry] pass %u: bounds %s, %u preparation(s) since, last reported list %u (pass %u), %s\n",
                carry_seq_, carry_was_valid ? "valid" : "void", carry_preps_, host_carry_[0], host_carry_[1],
                listed ? "listed pass" : (moved ? "whole pass, list counted" : "whole pass"));
      KMX_HIP(launch_lloyd_coarse_carry(a, rows, half, xcache_, xmeta_, panelhi_, undecided_, und_thr_, cy, hint,
                                        duo_listed ? duo_ : nullptr, stream_),
              kRuntimeError);
      carry_valid_ = true;
    } else {
      KMX_HIP(launch_lloyd_coarse(a, rows, half, cached ? xcache_ : nullptr, xmeta_, panelhi_, undecided_, und_thr_,
                                  duo_listed ? duo_ : nullptr, stream_),
              kRuntimeError);
    }
    // the rows stage 1 listed with their two contenders (lloyd_duo.hip): stage 2's decision without its sweep, beside
    // stage 2's sweep over the rest (a round of 67-KB blocks that leaves most of the chip's waves free)
    if (duo_listed) {
      KMX_HIP(hipEventRecord(ev_fork_, stream_), kRuntimeError);
      KMX_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0), kRuntimeError);
      KMX_HIP(launch_lloyd_duo(a, duo_, side_stream_, cy_refine.l3 ? &cy_refine : nullptr), kRuntimeError);
      KMX_HIP(hipEventRecord(ev_join_, side_stream_), kRuntimeError);
    }
    carry_preps_ = 0;
    span_end();
    // stage 2's grid follows an EARLIER pass's list length (whatever the update's async copy has
    // delivered to the pinned word; the kernel strides over the device-side count, so only speed
    // depends on it)
    last_undecided_ = host_move_count_[2];
    if (cy_refine.l3) {
      KMX_HIP(launch_lloyd_refine_carry(a, rows, half, panelhi_, undecided_, und_thr_, counters_ + 4, last_undecided_,
                                        cy_refine, stream_),
              kRuntimeError);
    } else {
      KMX_HIP(launch_lloyd_refine(a, rows, half, panelhi_, undecided_, und_thr_, counters_ + 4, last_undecided_,
                                  stream_),
              kRuntimeError);
    }
    if (duo_listed) KMX_HIP(hipStreamWaitEvent(stream_, ev_join_, 0), kRuntimeError);
  } else {
    KMX_HIP(launch_lloyd_filter(a, stream_), kRuntimeError);
  }
  span_end();
  span_begin(1);
  if (settle_ && lloyd_settle_supported(a, centroids)) {
    // both lists in one launch (KMCUDA_AMD_SETTLE=0: the two older kernels below, the cross-check)
    if (rows_on_side) KMX_HIP(hipStreamWaitEvent(stream_, ev_rows_, 0), kRuntimeError);   // csqr / ct
    KMX_HIP(launch_lloyd_settle(metric_, a, centroids, stream_), kRuntimeError);
    span_end();
    return kSuccess;
  }
  // the two refine kernels work on disjoint row lists and are latency bound (one 256-step exact
  // chain per contender): the full-scan kernel runs on a side stream beside the pair kernel
  KMX_HIP(hipEventRecord(ev_fork_, stream_), kRuntimeError);
  KMX_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0), kRuntimeError);
  const uint32_t grid = N_ < 4096u ? N_ : 4096u;  // grid-strides over the device-side flagged count
  KMX_HIP(launch_lloyd_exact(metric_, a, flagged_, counters_ + 1, grid, side_stream_), kRuntimeError);
  KMX_HIP(hipEventRecord(ev_join_, side_stream_), kRuntimeError);
  if (rows_on_side) KMX_HIP(hipStreamWaitEvent(stream_, ev_rows_, 0), kRuntimeError);   // csqr for the pair kernel
  KMX_HIP(launch_lloyd_pair(metric_, a, centroids, (N_ + 127) / 128 < 2048u ? (N_ + 127) / 128 : 2048u, stream_),
          kRuntimeError);
  KMX_HIP(hipStreamWaitEvent(stream_, ev_join_, 0), kRuntimeError);
  span_end();
  return kSuccess;
}

// D beyond the register-resident filters: lloyd_wide.hip.  prepare_centroids() has run (csqr, ct, mean -- frozen
// while a row copy is alive --, centred fp32 panel, biases, statistics, list counters zeroed).
int Engine::lloyd_assign_wide(const LloydArgs &a0, const float *centroids, bool steady, bool rows_on_side,
                              bool carry_was_valid) {
  constexpr int kNoFilter = kNoSuchDevice + 100;
  LloydArgs a = a0;
  const uint32_t DG = wide_dp_;
  const uint32_t k_pad64 = (K_pad_ + 63u) / 64u * 64u;
  // This path's own memory (half copy of the rows, the listed rows' contender table) is an optimisation: a job whose
  // rows fit but whose copies do not runs on the exact kernels, as it did before this path existed
  auto no_memory = [&]() {
    (void)hipGetLastError();
    if (g_verbosity > 0) printf("rows wider than 256 features: no memory for the streamed filter's buffers -- exact kernels\n");
    wide_failed_ = true;   // (wide_dp_ stays: the preparation's buffers are sized by it)
    return kNoFilter;
  };
  if (!panelhi_) {
    uint16_t *phi = nullptr;
    if (alloc(&phi, (size_t)k_pad64 * (DG + 2))) return no_memory();
    panelhi_ = phi;
  }
  if (!wide_cont_) {
    if ((!undecided_ && (alloc(&undecided_, N_) || alloc(&und_thr_, N_))) || alloc(&wide_cont_, wide_cont_words(N_))) {
      wide_cont_ = nullptr;
      return no_memory();
    }
  }
  if (!wide_rows16_) {
    uint16_t *xg = nullptr;
    if (alloc(&xg, (size_t)N_ * DG) || alloc(&wide_meta_, ((size_t)N_ + 1) * 4)) return no_memory();
    wide_rows16_ = xg;
  }
  span_begin(0);
  // hi halves of the centred centroids (+ their residual maximum, stats[5]); the steady state's one preparation
  // kernel has written them
  if (!steady)
    KMX_HIP(launch_centroid_panelhi(centroids, K_, D_, K_pad_, DG, finite_, mu_, bias_, panelhi_, stats_, stream_),
            kRuntimeError);
  // the rows as centred halves: kept while the caller has promised fixed rows (the mean is then frozen, any mean
  // being valid), otherwise rebuilt for this pass's mean
  if (!(row_cache_on_ && row_cache_valid_)) {
    const void *rows = half_rows_ ? half_rows_ : (const void *)a.samples;
    KMX_HIP(launch_row_halves(rows, half_rows_ != nullptr, N_, D_, DG, mu_, wide_rows16_, wide_meta_, stream_), kRuntimeError);
    if (row_cache_on_) {
      row_cache_valid_ = true;
      mu_frozen_ = true;
    }
  }
  // Carried bounds, as in the register-resident filter's steady state (lloyd_assign): the pass leaves per-row bounds,
  // the next one only looks at the rows they do not decide.  No pair certificates here: a row the later stages settle
  // is listed again.  (The drift bound's rounding allowance covers fp32 sums of up to ~6000 terms.)
  bool carry = carry_on_ && steady && DG <= 4096u;
  if (carry && carry_policy_.paused()) {
    carry = false;
    if (getenv("KMCUDA_AMD_CARRY_TRACE")) fprintf(stderr, "[carry] paused (%u more)\n", carry_policy_.pause);
  }
  if (carry && !ub_) {
    if (alloc(&ub_, N_) != kSuccess || alloc(&lb_, N_) != kSuccess || alloc(&drift_, 2 * (size_t)K_) != kSuccess ||
        alloc(&carry_list_, N_) != kSuccess || !(host_carry_ = pinned_words(2, &host_carry_dev_))) {
      (void)hipGetLastError();   // (no memory: not an error, plain passes)
      carry_on_ = false;
      ub_ = nullptr;
    } else {
      host_carry_[0] = 0xFFFFFFFFu;
      host_carry_[1] = 0;
      if (const char *v = getenv("KMCUDA_AMD_CARRY_MAX")) carry_policy_.list_max = (float)atof(v);
    }
  }
  span_begin(3);   // the dominant kernels on their own, inside the filter span
  if (carry && carry_on_) {
    CarryArgs cy;
    cy.ub = ub_; cy.lb = lb_; cy.host_report = host_carry_dev_; cy.seq = ++carry_seq_;
    cy.angular = metric_ != 0;
    const bool moved = carry_was_valid && carry_preps_ == 1;   // drift_ / stats_[6] belong to the bounds
    bool listed = false;
    if (moved) {
      const uint32_t last = host_carry_[0], last_seq = host_carry_[1];
      listed = carry_policy_.decide(last, last_seq, carry_seq_, N_);
      KMX_HIP(launch_carry_skip(N_, K_, a.assignments, a.assignments_prev, cy, wide_meta_, drift_, stats_, tie_slack_,
                                carry_list_, finite_, pairs_, counters_, !listed, stream_, DG),
              kRuntimeError);
      cy.n_list = counters_ + kCarryCursor;
    }
    if (listed) cy.row_list = carry_list_;
    if (getenv("KMCUDA_AMD_CARRY_TRACE"))
      fprintf(stderr, "[carry] streamed pass %u: bounds %s, %u preparation(s) since, last reported list %u (pass %u), %s\n",
              carry_seq_, carry_was_valid ? "valid" : "void", carry_preps_, host_carry_[0], host_carry_[1],
              listed ? "listed pass" : (moved ? "whole pass, list counted" : "whole pass"));
    KMX_HIP(launch_lloyd_wide(a, wide_rows16_, wide_meta_, DG, panelhi_, undecided_, und_thr_, wide_cont_, stream_, &cy),
            kRuntimeError);
    carry_valid_ = true;
  } else {
    KMX_HIP(launch_lloyd_wide(a, wide_rows16_, wide_meta_, DG, panelhi_, undecided_, und_thr_, wide_cont_, stream_), kRuntimeError);
  }
  carry_preps_ = 0;
  span_end();
  if (rows_on_side) KMX_HIP(hipStreamWaitEvent(stream_, ev_rows_, 0), kRuntimeError);   // csqr / ct: the exact chains
  KMX_HIP(launch_wide_contenders(metric_, a, centroids, DG, undecided_, wide_cont_, stream_), kRuntimeError);
  span_end();
  span_begin(1);
  if (settle_ && lloyd_settle_supported(a, centroids)) {
    KMX_HIP(launch_lloyd_settle(metric_, a, centroids, stream_), kRuntimeError);
  } else {
    KMX_HIP(launch_lloyd_pair(metric_, a, centroids, (N_ + 127) / 128 < 2048u ? (N_ + 127) / 128 : 2048u, stream_),
            kRuntimeError);
    const uint32_t grid = N_ < 4096u ? N_ : 4096u;
    KMX_HIP(launch_lloyd_exact(metric_, a, flagged_, counters_ + 1, grid, stream_), kRuntimeError);
  }
  span_end();
  return kSuccess;
}

int Engine::move_deltas(const float *samples, const uint32_t *prev, const uint32_t *cur, double *delta,
                        int32_t *dcount, double *tail) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  span_begin(2);
  // (the kernel also reports the length of this pass's undecided list to the pinned words, for a LATER pass's
  // stage-2 grid: nobody waits for it)
  KMX_HIP(launch_move_deltas(samples, N_, D_, K_, prev, cur, keys_tmp_, vals_tmp_, keys_sorted_, rows_sorted_,
                             offsets2_, sort_temp_, sort_temp_bytes_, bucket_rows_, bucket_cap_, delta, dcount, tail,
                             counters_, move_blocks_, bucket_work_, &ms_, stream_),
          kRuntimeError);
  span_end();
  return kSuccess;
}

int Engine::apply_delta(const double *delta, const int32_t *dcount, const double *dcount_d, float *centroids,
                        uint32_t *ccounts, float stop_threshold, bool report, uint32_t seq) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  prepared_for_ = nullptr;   // whatever preparation there was belongs to the old centroids
  StopCtl ctl;
  if (stop_threshold >= 0.f || report) {
    if (!dcount_d) return kInvalidArguments;   // the stop rule reads the fused buffer's reduced counters
    int rc = stop_ctl(stop_threshold, report, seq, &ctl);
    if (rc) return rc;
  }
  span_begin(2);
  KMX_HIP(launch_apply_delta(metric_, delta, dcount, dcount_d, K_, D_, centroids, ccounts, ctl, stream_), kRuntimeError);
  span_end();
  // (on the engine's own stream: an event on the caller's legacy default stream would order every blocking
  // stream of the device behind it)
  if (report) KMX_HIP(hipEventRecord(ev_report_[seq & 1u], stream_), kRuntimeError);
  return kSuccess;
}

int Engine::stop_ctl(float stop_threshold, bool report, uint32_t seq, StopCtl *ctl) {
  ctl->threshold = stop_threshold;
  ctl->counters = counters_;
  ctl->seq = seq;
  if (report) {
    if (!host_report_) {
      host_report_ = pinned_words(16, &host_report_dev_);
      if (!host_report_) return kMemoryAllocationFailure;
      memset(host_report_, 0xFF, 16 * sizeof(uint32_t));
      for (hipEvent_t &e : ev_report_) KMX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), kRuntimeError);
    }
    ctl->host_tail = host_report_dev_ + 8 * (seq & 1u);
  }
  return kSuccess;
}

int Engine::stop_report(uint32_t seq, uint32_t *host_out6) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  if (!host_report_) return kInvalidArguments;
  KMX_HIP(hipEventSynchronize(ev_report_[seq & 1u]), kRuntimeError);
  const volatile uint32_t *t = host_report_ + 8 * (seq & 1u);
  for (int i = 0; i < 6; i++) host_out6[i] = t[i];
  return host_out6[5] == seq ? kSuccess : kRuntimeError;   // another call has reused the slot
}

// the steady state of the two-stage filter: mean frozen, row cache valid, panel buffers allocated
bool Engine::steady_state(bool exact_only) const {
  const bool two_stage = !exact_only && DP_ != 0 && filter_mode_ == 0 && lloyd_filter_f16_supported(D_, DP_) && !strict_h2_;
  return two_stage && mu_frozen_ && row_cache_on_ && row_cache_valid_ && N_ != 0 && panelhi_ != nullptr;
}

// apply_delta + the NEXT lloyd_assign's preparation in one launch (L2, steady state; anything else: apply_delta).
// The caller promises not to touch `centroids` before that lloyd_assign (which recognises the buffer by address).
int Engine::apply_prepare(const double *delta, const double *dcount_d, float *centroids, uint32_t *ccounts,
                          float stop_threshold, bool report, uint32_t seq) {
  if (!(metric_ == 0 && steady_state(false) && dcount_d)) {
    prepared_for_ = nullptr;
    return apply_delta(delta, nullptr, dcount_d, centroids, ccounts, stop_threshold, report, seq);
  }
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  StopCtl ctl;   // (without a threshold or a report the update is unconditional: a raised stop flag is not looked at)
  if (stop_threshold >= 0.f || report) {
    int rc = stop_ctl(stop_threshold, report, seq, &ctl);
    if (rc) return rc;
  }
  uint32_t *next = stats_;
  stats_ = stats_ == stats_base_ ? stats_base_ + 8 : stats_base_;
  span_begin(2);
  KMX_HIP(launch_apply_prep_frozen(delta, dcount_d, centroids, ccounts, ctl, K_, D_, K_pad_, DP_, mu_, finite_, bias_,
                                   bias2_, cfil_, panelhi_, stats_, next, counters_ + 1, counters_ + 3, counters_ + 4,
                                   carry_on_ ? drift_ : nullptr, carry_on_ && drift_ ? counters_ + kCarryCursor : nullptr,
                                   stream_),
          kRuntimeError);
  carry_preps_++;
  span_end();
  if (report) KMX_HIP(hipEventRecord(ev_report_[seq & 1u], stream_), kRuntimeError);
  prepared_for_ = centroids;
  return kSuccess;
}

int Engine::carry_stats(unsigned long long *rows_spared, uint32_t *last_list) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  unsigned long long v = 0;
  KMX_HIP(hipMemcpyAsync(&v, counters_ + kCarrySkipped, sizeof(v), hipMemcpyDeviceToHost, stream_), kMemoryCopyError);
  KMX_HIP(hipStreamSynchronize(stream_), kRuntimeError);
  if (rows_spared) *rows_spared = v;
  if (last_list) *last_list = host_carry_ ? host_carry_[0] : 0xFFFFFFFFu;
  return kSuccess;
}

int Engine::carry_pair_stats(unsigned long long *rows_paired) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  unsigned long long v = 0;
  KMX_HIP(hipMemcpyAsync(&v, counters_ + kCarryPaired, sizeof(v), hipMemcpyDeviceToHost, stream_), kMemoryCopyError);
  KMX_HIP(hipStreamSynchronize(stream_), kRuntimeError);
  if (rows_paired) *rows_paired = v;
  return kSuccess;
}

int Engine::duo_rows(uint32_t *rows) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  uint32_t v = 0;
  KMX_HIP(hipMemcpyAsync(&v, counters_ + kDuoCount, sizeof(v), hipMemcpyDeviceToHost, stream_), kMemoryCopyError);
  KMX_HIP(hipStreamSynchronize(stream_), kRuntimeError);
  if (rows) *rows = v;
  return kSuccess;
}

int Engine::stop_clear() {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  carry_valid_ = false;   // (a pass enqueued behind a raised flag wrote no bounds: start over)
  KMX_HIP(hipMemsetAsync(counters_ + kStopFlag, 0, sizeof(uint32_t), stream_), kRuntimeError);
  return kSuccess;
}

int Engine::adjust_exact(const float *samples, const uint32_t *prev, const uint32_t *cur, float *centroids,
                         uint32_t *ccounts) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  if (strict_h2_) {
    KMX_HIP(launch_h2_adjust(metric_, samples, N_, D_, K_, prev, cur, centroids, ccounts, stream_), kRuntimeError);
    return kSuccess;
  }
  if (2ull * N_ >= 0xFFFFFFFFull) return kInvalidArguments;
  if ((size_t)D_ * 64 * sizeof(float) > 128 * 1024 && !exact_work_) {
    int rc = alloc(&exact_work_, (size_t)((K_ + 63) / 64) * 64 * D_);
    if (rc) return rc;
  }
  span_begin(2);
  KMX_HIP(launch_adjust_exact(metric_, samples, N_, D_, K_, prev, cur, keys_tmp_, vals_tmp_, keys_sorted_,
                              rows_sorted_, offsets2_, sort_temp_, sort_temp_bytes_, exact_work_, centroids, ccounts,
                              stream_),
          kRuntimeError);
  span_end();
  return kSuccess;
}

int Engine::counters_read(uint32_t *host4) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  KMX_HIP(hipMemcpyAsync(host_counters_, counters_, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_),
          kMemoryCopyError);
  KMX_HIP(hipStreamSynchronize(stream_), kRuntimeError);
  memcpy(host4, host_counters_, 4 * sizeof(uint32_t));
  return kSuccess;
}

int Engine::yy_hint_stats(uint32_t *host6) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  for (int i = 0; i < 6; i++) host6[i] = 0;
  if (!yy_stats_) return kSuccess;
  std::vector<uint32_t> st(64 * 16);
  KMX_HIP(hipMemcpyAsync(st.data(), yy_stats_, st.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_),
          kMemoryCopyError);
  KMX_HIP(hipStreamSynchronize(stream_), kRuntimeError);
  for (int b = 0; b < 64; b++)
    for (int i = 0; i < 6; i++) host6[i] += st[b * 16 + 6 + i];   // striped by block number (yinyang_hint.hip)
  return kSuccess;
}

int Engine::counters_reset(int which) {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  if (which < 0) KMX_HIP(hipMemsetAsync(counters_, 0, 4 * sizeof(uint32_t), stream_), kRuntimeError);
  else KMX_HIP(hipMemsetAsync(counters_ + which, 0, sizeof(uint32_t), stream_), kRuntimeError);
  return kSuccess;
}

int Engine::sync() {
  KMX_HIP(hipSetDevice(device_), kNoSuchDevice);
  KMX_HIP(hipStreamSynchronize(stream_), kRuntimeError);
  return kSuccess;
}

}  // namespace kmx

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
struct kmamd_engine {
  kmx::Engine e;
};

extern "C" {

int kmamd_engine_create(kmamd_engine **out, int device, uint32_t n_rows, uint32_t features, uint32_t clusters,
                        int metric, int fp16x2, void *hip_stream) {
  if (!out) return kmx::kInvalidArguments;
  *out = nullptr;
  kmamd_engine *h = new kmamd_engine();
  const int rc = h->e.init(device, n_rows, features, clusters, metric, fp16x2, (hipStream_t)hip_stream);
  if (rc != kmx::kSuccess) {
    delete h;
    return rc;
  }
  *out = h;
  return kmx::kSuccess;
}

void kmamd_engine_destroy(kmamd_engine *e) { delete e; }
void *kmamd_engine_stream(kmamd_engine *e) { return (void *)e->e.stream_; }
int kmamd_engine_sync(kmamd_engine *e) { return e->e.sync(); }

int kmamd_lloyd_assign(kmamd_engine *e, const float *samples, const float *centroids, uint32_t *assignments,
                       uint32_t *assignments_prev) {
  return e->e.lloyd_assign(samples, centroids, assignments, assignments_prev, false);
}
int kmamd_lloyd_assign_exact(kmamd_engine *e, const float *samples, const float *centroids, uint32_t *assignments,
                             uint32_t *assignments_prev) {
  return e->e.lloyd_assign(samples, centroids, assignments, assignments_prev, true);
}
int kmamd_counters_read(kmamd_engine *e, uint32_t *host_out4) { return e->e.counters_read(host_out4); }
int kmamd_counters_reset(kmamd_engine *e, int which) { return e->e.counters_reset(which); }
int kmamd_yy_hint_stats(kmamd_engine *e, uint32_t *host_out6) { return e->e.yy_hint_stats(host_out6); }
int kmamd_move_deltas(kmamd_engine *e, const float *samples, const uint32_t *assignments_prev,
                      const uint32_t *assignments, double *delta, int32_t *dcount) {
  return e->e.move_deltas(samples, assignments_prev, assignments, delta, dcount, nullptr);
}
int kmamd_apply_delta(kmamd_engine *e, const double *delta, const int32_t *dcount, float *centroids,
                      uint32_t *ccounts) {
  return e->e.apply_delta(delta, dcount, nullptr, centroids, ccounts);
}
size_t kmamd_reduce_len(kmamd_engine *e) { return (size_t)e->e.K_ * e->e.D_ + e->e.K_ + 4; }
int kmamd_reduce_fill(kmamd_engine *e, const float *samples, const uint32_t *assignments_prev,
                      const uint32_t *assignments, double *buf) {
  return e->e.move_deltas(samples, assignments_prev, assignments, buf, nullptr, buf + (size_t)e->e.K_ * e->e.D_);
}
int kmamd_reduce_apply(kmamd_engine *e, const double *buf, float *centroids, uint32_t *ccounts) {
  return e->e.apply_delta(buf, nullptr, buf + (size_t)e->e.K_ * e->e.D_, centroids, ccounts);
}
int kmamd_reduce_apply_stop(kmamd_engine *e, const double *buf, float *centroids, uint32_t *ccounts,
                            float stop_threshold, uint32_t seq) {
  return e->e.apply_delta(buf, nullptr, buf + (size_t)e->e.K_ * e->e.D_, centroids, ccounts, stop_threshold, true, seq);
}
int kmamd_reduce_apply_prepare(kmamd_engine *e, const double *buf, float *centroids, uint32_t *ccounts,
                               float stop_threshold, uint32_t seq) {
  return e->e.apply_prepare(buf, buf + (size_t)e->e.K_ * e->e.D_, centroids, ccounts, stop_threshold, true, seq);
}
int kmamd_stop_report(kmamd_engine *e, uint32_t seq, uint32_t *host_out6) { return e->e.stop_report(seq, host_out6); }
int kmamd_stop_clear(kmamd_engine *e) { return e->e.stop_clear(); }
int kmamd_centroids_written(kmamd_engine *e) {
  e->e.prepared_for_ = nullptr;
  e->e.carry_valid_ = false;
  return kmx::kSuccess;
}
int kmamd_set_carry(kmamd_engine *e, int on) {
  e->e.carry_on_ = on != 0;
  e->e.carry_valid_ = false;
  return kmx::kSuccess;
}
int kmamd_carry_stats(kmamd_engine *e, uint64_t *rows_spared, uint32_t *last_list) {
  unsigned long long v = 0;
  const int rc = e->e.carry_stats(&v, last_list);
  if (rows_spared) *rows_spared = v;
  return rc;
}
// The carried-bounds host policy replayed without a device (CarryPolicy, engine.hpp): pass i + 1 would count
// list_len[i] rows if it counts a list; a pass's report reaches the host `lag` passes later.  out[i]: 0 a plain pass
// (paused), 1 a whole pass that leaves bounds (none were valid: nothing to count), 2 a whole pass that counts its
// would-be list, 3 a listed pass.
int kmamd_carry_policy_sim(uint32_t n_passes, uint32_t n_rows, float list_max, const uint32_t *list_len, uint32_t lag,
                           uint8_t *out, const uint32_t *changed) {
  if (!list_len || !out || lag == 0) return kmx::kInvalidArguments;
  kmx::CarryPolicy policy;
  policy.list_max = list_max;
  std::vector<uint32_t> reported(n_passes + 1, kmx::CarryPolicy::kNoList);   // by sequence number (1-based)
  std::vector<bool> has_report(n_passes + 1, false);
  bool valid = false;
  for (uint32_t i = 0; i < n_passes; i++) {
    const uint32_t seq = i + 1;
    // (the host has judged the passes at least `lag` back: their reassignment counts)
    if (changed && i >= lag) policy.note_changed(changed[i - lag]);
    if (policy.paused()) {
      out[i] = 0;
      valid = false;
      continue;
    }
    if (!valid) {
      out[i] = 1;
      reported[seq] = kmx::CarryPolicy::kNoList;
    } else {
      // the newest report that has landed: of a pass at least `lag` passes back
      uint32_t last = kmx::CarryPolicy::kNoList, last_seq = 0;
      for (uint32_t q = seq > lag ? seq - lag : 0; q >= 1; q--)
        if (has_report[q]) { last = reported[q]; last_seq = q; break; }
      out[i] = policy.decide(last, last_seq, seq, n_rows) ? 3 : 2;
      reported[seq] = list_len[i];
    }
    has_report[seq] = true;
    valid = true;
  }
  return kmx::kSuccess;
}
int kmamd_carry_pair_stats(kmamd_engine *e, uint64_t *rows_paired) {
  unsigned long long v = 0;
  const int rc = e->e.carry_pair_stats(&v);
  if (rows_paired) *rows_paired = v;
  return rc;
}
int kmamd_duo_rows(kmamd_engine *e, uint32_t *rows) { return e->e.duo_rows(rows); }
int kmamd_set_update_mode(kmamd_engine *e, int mode) {
  if (mode < 0 || mode > 3) return kmx::kInvalidArguments;
  e->e.ms_.force = mode;
  return kmx::kSuccess;
}
int kmamd_set_filter(kmamd_engine *e, int mode) {
  if (mode < 0 || mode > 1) return kmx::kInvalidArguments;
  e->e.filter_mode_ = mode;
  e->e.prepared_for_ = nullptr;
  return kmx::kSuccess;
}
int kmamd_set_half_rows(kmamd_engine *e, const void *rows16) {
  e->e.half_rows_ = rows16;
  e->e.prepared_for_ = nullptr;
  e->e.row_cache_valid_ = false;  // a cache built from other rows is stale
  return kmx::kSuccess;
}
int kmamd_set_row_cache(kmamd_engine *e, int on) {
  e->e.row_cache_on_ = on != 0 && e->e.row_cache_allowed_;
  e->e.row_cache_valid_ = false;  // (re)built by the next kmamd_lloyd_assign
  e->e.prepared_for_ = nullptr;
  if (!e->e.row_cache_on_) e->e.mu_frozen_ = false;
  return kmx::kSuccess;
}
int kmamd_adjust_exact(kmamd_engine *e, const float *samples, const uint32_t *assignments_prev,
                       const uint32_t *assignments, float *centroids, uint32_t *ccounts) {
  return e->e.adjust_exact(samples, assignments_prev, assignments, centroids, ccounts);
}
int kmamd_afkmc2_draws(kmamd_engine *e, uint64_t seed, uint64_t offset, uint32_t threads, uint32_t n, uint32_t *out) {
  if (hipSetDevice(e->e.device_) != hipSuccess) return kmx::kNoSuchDevice;
  return kmx::launch_afk_draws(seed, offset, threads, n, out, e->e.stream_) == hipSuccess ? kmx::kSuccess : kmx::kRuntimeError;
}
int kmamd_transpose(kmamd_engine *e, const float *in, uint32_t rows, uint32_t cols, float *out) {
  if (hipSetDevice(e->e.device_) != hipSuccess) return kmx::kNoSuchDevice;
  return kmx::launch_transpose(in, rows, cols, out, e->e.stream_) == hipSuccess ? kmx::kSuccess : kmx::kRuntimeError;
}
int kmamd_yy_configure(kmamd_engine *e, uint32_t G, const uint32_t *groups_host) {
  return e->e.yy_configure(G, groups_host);
}
int kmamd_yy_init(kmamd_engine *e, const float *samples, const float *centroids, const uint32_t *assignments,
                  float *bounds) {
  return e->e.yy_init(samples, centroids, assignments, bounds);
}
int kmamd_yy_drifts(kmamd_engine *e, const float *centroids, float *drifts, float *gdrifts) {
  return e->e.yy_drifts(centroids, drifts, gdrifts);
}
int kmamd_yy_filters(kmamd_engine *e, const float *samples, const float *centroids, const float *drifts,
                     const float *gdrifts, uint32_t *assignments, uint32_t *assignments_prev, float *bounds,
                     uint32_t *passed) {
  return e->e.yy_filters(samples, centroids, drifts, gdrifts, assignments, assignments_prev, bounds, passed);
}
int kmamd_filter_kind(kmamd_engine *e, uint32_t *padded_width) {
  const kmx::Engine &g = e->e;
  const bool streamed = g.DP_ == 0 && g.wide_dp_ != 0 && !g.wide_failed_;
  if (padded_width) *padded_width = g.DP_ ? g.DP_ : (streamed ? g.wide_dp_ : 0);
  return g.DP_ ? 1 : (streamed ? 2 : 0);
}
int kmamd_profile_enable(kmamd_engine *e, int on) {
  e->e.profile_collect();
  e->e.profile_ = on != 0;
  return kmx::kSuccess;
}
int kmamd_profile_reset(kmamd_engine *e) {
  e->e.profile_reset();
  return kmx::kSuccess;
}
int kmamd_profile_read(kmamd_engine *e, double *filter_ms, uint32_t *filter_launches, double *exact_ms,
                       double *update_ms) {
  e->e.profile_collect();
  if (filter_ms) *filter_ms = e->e.filter_ms_;
  if (filter_launches) *filter_launches = e->e.filter_launches_;
  if (exact_ms) *exact_ms = e->e.exact_ms_;
  if (update_ms) *update_ms = e->e.update_ms_;
  return kmx::kSuccess;
}
int kmamd_profile_read_coarse(kmamd_engine *e, double *coarse_ms) {
  e->e.profile_collect();
  if (coarse_ms) *coarse_ms = e->e.coarse_ms_;
  return kmx::kSuccess;
}
int kmamd_copy_to_device(int device, void *dst, const void *host_src, size_t bytes) {
  if (hipSetDevice(device) != hipSuccess) return kmx::kNoSuchDevice;
  return hipMemcpy(dst, host_src, bytes, hipMemcpyHostToDevice) == hipSuccess ? kmx::kSuccess : kmx::kMemoryCopyError;
}
const char *kmamd_build_arch(void) { return "gfx950"; }

}  // extern "C"
