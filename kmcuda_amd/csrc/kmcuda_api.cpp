// kmcuda_api.cpp -- the drop-in C ABI: kmeans_cuda() and knn_cuda() of include/kmcuda.h.
//
// Host orchestration mirroring the reference's contract (src/kmcuda.cc:402-531 kmeans_cuda,
// src/kmeans.cu:934-1026 kmeans_cuda_lloyd, :1028-1263 kmeans_cuda_yy): same argument
// validation and return codes, same srand()/rand()-driven seeding, same stop rule evaluated
// BEFORE the centroid update, same "iteration %d: %u reassignments" progress lines.
//
// What is different by design: samples stay row-major and are ROW-SHARDED over the GPUs of
// the device mask (the reference replicates everything on every GPU and exchanges slices by
// N*(N-1) peer copies); the only per-iteration exchange is one all-reduce of the centroid
// deltas (RCCL over xGMI, loaded lazily so single-GPU use has no RCCL dependency).
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/kmcuda.h"
#include "../../include/kmcuda_amd.h"
#include "engine.hpp"

using namespace kmx;

#define INFO(...) do { if (verbosity > 0) { printf(__VA_ARGS__); } } while (false)
#define DEBUG(...) do { if (verbosity > 1) { printf(__VA_ARGS__); } } while (false)
#define RETERR(call) do { int rc__ = (call); if (rc__ != 0) return static_cast<KMCUDAResult>(rc__); } while (false)

namespace {

constexpr double kYinyangGroupTolerance = 0.02;      // kmeans.cu:27
constexpr double kYinyangDraftReassignments = 0.11;  // kmeans.cu:28
constexpr double kYinyangRefreshEpsilon = 1e-4;      // kmeans.cu:29

// ---------------------------------------------------------------------------------------
// RCCL, loaded on first multi-GPU use
// ---------------------------------------------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  bool load() {
    if (handle) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
    AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
  }
};
constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // rccl.h ncclDataType_t / ncclRedOp_t

// ---------------------------------------------------------------------------------------
// device list from the mask (reference: setup_devices, kmcuda.cc:63-137)
// ---------------------------------------------------------------------------------------
std::vector<int> setup_devices(uint32_t device, int verbosity) {
  std::vector<int> devs;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return devs;
  if (device == 0) device = ndev >= 32 ? 0xFFFFFFFFu : ((1u << ndev) - 1);
  for (int dev = 0; device; dev++, device >>= 1) {
    if (!(device & 1)) continue;
    if (dev >= ndev || hipSetDevice(dev) != hipSuccess) {
      INFO("failed to hipSetDevice(%d)\n", dev);
      continue;
    }
    devs.push_back(dev);
  }
  return devs;
}

// contiguous row blocks, one per shard (the reference's distribute(), private.h:240-273,
// aligns to 512 B; rows here are whole samples so plain balanced blocks do)
// Shards of at least 1024 rows start on multiples of 256 rows: the k-means++ chooser's exact block sums (256 rows) and
// the reference's butterfly sums (aligned groups of 32 rows, kmeans.cu:63-66) then never straddle two shards, and the
// per-shard sums concatenate into the one-shard sums (seeding.hip: kmpp_choose_kernel).
std::vector<std::pair<uint32_t, uint32_t>> row_plan(uint32_t N, size_t nshards) {
  std::vector<std::pair<uint32_t, uint32_t>> plan;
  const bool align = nshards > 1 && (uint64_t)N / nshards >= 1024u;
  auto start = [&](size_t i) -> uint32_t {
    if (i >= nshards) return N;
    const uint32_t o = (uint32_t)(((uint64_t)N * i) / nshards);
    return align ? (o & ~255u) : o;
  };
  for (size_t i = 0; i < nshards; i++) plan.emplace_back(start(i), start(i + 1) - start(i));
  return plan;
}

// sum of `n` floats the way the reference forms it on device (kmeans.cu:63-66, :687-690):
// float butterfly over each aligned group of 32, then a double accumulation
static float butterfly_group(const float *v, uint32_t base, uint32_t n) {
  float lane[32];
  for (int l = 0; l < 32; l++) lane[l] = (base + l < n) ? v[base + l] : 0.f;
  for (int off = 16; off > 0; off /= 2)
    for (int l = 0; l < 32; l++) lane[l] = lane[l] + ((l + off < 32) ? lane[l + off] : lane[l]);
  return lane[0];
}

double butterfly_sum(const float *v, uint32_t n) {
  const uint32_t groups = (n + 31) / 32;
  if (groups < 4096) {
    double sum = 0.0;
    for (uint32_t g = 0; g < groups; g++) sum += (double)butterfly_group(v, g * 32, n);
    return sum;
  }
  // the group sums are independent (threads), the double accumulation keeps its sequential order
  std::vector<float> gs(groups);
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 4 : (nt > 16 ? 16 : nt);
  std::vector<std::thread> pool;
  const uint32_t per = (groups + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    const uint32_t g0 = t * per, g1 = std::min(groups, g0 + per);
    if (g0 >= g1) break;
    pool.emplace_back([&gs, v, n, g0, g1]() {
      for (uint32_t g = g0; g < g1; g++) gs[g] = butterfly_group(v, g * 32, n);
    });
  }
  for (auto &th : pool) th.join();
  double sum = 0.0;
  for (uint32_t g = 0; g < groups; g++) sum += (double)gs[g];
  return sum;
}

// ---------------------------------------------------------------------------------------
// One host thread per shard for the lifetime of the job.  A multi-GPU iteration is a dozen launches per GPU
// and lasts well under a millisecond on a 1M-row shard: enqueueing eight GPUs from one thread, or creating and
// joining eight threads per iteration, costs about as much as the GPUs take to run it.
// ---------------------------------------------------------------------------------------
class ShardWorkers {
 public:
  ~ShardWorkers() {
    {
      std::lock_guard<std::mutex> l(m_);
      quit_ = true;
      gen_++;
    }
    cv_go_.notify_all();
    for (auto &t : threads_) t.join();
  }
  size_t size() const { return threads_.size(); }
  void start(size_t n) {
    rcs_.assign(n, 0);
    for (size_t i = 0; i < n; i++) threads_.emplace_back([this, i]() { loop(i); });
  }
  // fn(i) on worker i, for every i; returns when all are done, with the first non-zero code
  int run(const std::function<int(size_t)> &fn) {
    {
      std::lock_guard<std::mutex> l(m_);
      fn_ = &fn;
      pending_ = threads_.size();
      gen_++;
    }
    cv_go_.notify_all();
    std::unique_lock<std::mutex> l(m_);
    cv_done_.wait(l, [this]() { return pending_ == 0; });
    for (int rc : rcs_)
      if (rc) return rc;
    return 0;
  }

 private:
  void loop(size_t i) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(size_t)> *fn;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_go_.wait(l, [&]() { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
        fn = fn_;
      }
      const int rc = (*fn)(i);
      {
        std::lock_guard<std::mutex> l(m_);
        rcs_[i] = rc;
        if (--pending_ == 0) cv_done_.notify_one();
      }
    }
  }
  std::mutex m_;
  std::condition_variable cv_go_, cv_done_;
  std::vector<std::thread> threads_;
  std::vector<int> rcs_;
  const std::function<int(size_t)> *fn_ = nullptr;
  uint64_t gen_ = 0;
  size_t pending_ = 0;
  bool quit_ = false;
};

// ---------------------------------------------------------------------------------------
// one GPU's share of the job
// ---------------------------------------------------------------------------------------
struct Shard {
  int dev = 0;
  uint32_t offset = 0, length = 0;
  std::unique_ptr<Engine> eng;
  const float *samples = nullptr;  // length x D on this device
  float *centroids = nullptr;      // K x D (replicated)
  uint32_t *assignments = nullptr, *prev = nullptr, *ccounts = nullptr;
  double *reduce = nullptr;        // the iteration's ONE exchange: [delta K*D | dcount K | counters 4] (kmcuda_amd.h)
  float *dists = nullptr;          // length floats (k-means++ / average distance)
  // Yinyang state (allocated when the Yinyang phase starts; reference: kmcuda.cc:448-470)
  float *bounds = nullptr;         // (G+1) x length, group-major (kmeans.cu:431-485 layout)
  float *drifts = nullptr;         // K*D old centroids + K per-centroid drifts
  float *gdrifts = nullptr;        // G per-group max drifts
  uint32_t *passed = nullptr;      // length
  hipEvent_t ev_filled = nullptr;  // several shards on one device: this shard's reduce buffer is written
  std::vector<void *> owned;
  ~Shard() {
    (void)hipSetDevice(dev);
    if (ev_filled) (void)hipEventDestroy(ev_filled);
    for (void *p : owned) (void)hipFree(p);
  }
  template <typename T>
  int alloc(T **p, size_t count) {   // (through the engine once it exists: small buffers share its slabs)
    if (eng) return eng->alloc(p, count);
    void *q = nullptr;
    if (hipMalloc(&q, count ? count * sizeof(T) : sizeof(T)) != hipSuccess) return kmcudaMemoryAllocationFailure;
    owned.push_back(q);
    *p = static_cast<T *>(q);
    return 0;
  }
};

// glibc's rand() (random_r, TYPE_3: 31 words, r[k] = r[k - 31] + r[k - 3], output r[k] >> 1) restated, for the one place
// that draws N numbers (init = "random").  attach(seed): called right after srand(seed); checks the restatement against
// the library's own first draws (false: not this generator -- the caller uses rand()), then moves the library onto a
// scratch buffer (initstate) while detach() overwrites the library's OWN state table with the restated generator's state
// and moves it back (setstate): the library's rand() goes on exactly where the same number of rand() calls would have
// left it, in its own memory.
class GlibcRand {
 public:
  bool attach(uint32_t seed) {
    seed_ = seed;
    reseed();
    uint32_t mine[4];
    for (uint32_t &v : mine) v = next();
    bool same = true;
    for (uint32_t v : mine) same = same && (uint32_t)rand() == v;
    srand(seed);   // back to the start for either path
    if (!same) return false;
    reseed();
    return true;
  }
  uint32_t next() {
    s_[f_] += s_[r_];
    const uint32_t out = s_[f_] >> 1;
    f_ = f_ + 1 == 31 ? 0 : f_ + 1;
    r_ = r_ + 1 == 31 ? 0 : r_ + 1;
    return out;
  }
  void detach() {
    // The generator goes back to the C library's OWN state table (or whatever buffer the caller had installed), with
    // the restated generator's words in it: initstate() parks the library on a scratch buffer and returns the table
    // it was using, the table is overwritten, setstate() moves the library back onto it.  Nothing of this library
    // stays referenced by rand() afterwards (round 4 left the library on a static buffer of ours: a dlclose away
    // from unmapped memory).  attach() has verified that the table is the 31-word TYPE_3 one.
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    int32_t scratch[32];
    char *table = initstate(1u, reinterpret_cast<char *>(scratch), sizeof(scratch));
    if (!table) return;
    int32_t *st = reinterpret_cast<int32_t *>(table);
    for (int i = 0; i < 31; i++) st[1 + i] = (int32_t)s_[i];
    st[0] = (int32_t)(r_ * 5 + 3);   // glibc: rear * MAX_TYPES + TYPE_3 (setstate() reads type and position from it)
    (void)setstate(table);           // (stores the scratch buffer's position word first: into `scratch`, ours)
  }

 private:
  void reseed() {   // srandom_r: 16807 LCG fills the table, 310 outputs are discarded
    int32_t word = seed_ ? (int32_t)seed_ : 1;
    s_[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
      const long hi = word / 127773, lo = word % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t)w;
      s_[i] = (uint32_t)word;
    }
    f_ = 3;
    r_ = 0;
    for (int i = 0; i < 310; i++) (void)next();
  }
  uint32_t s_[31], seed_ = 0;
  int f_ = 3, r_ = 0;
};

// what the last kmeans_cuda() call did (kmamd_last_run_stats): iterations of the Lloyd / Yinyang loops and
// the wall time spent in them (everything after seeding), for drivers that time the drop-in entry point
struct RunStats {
  uint32_t iterations = 0; double loop_seconds = 0, setup_seconds = 0; uint32_t shards = 0, rccl = 0;
  // kmamd_last_run_collective: the per-iteration all-reduce as HIP events on the first shard's stream saw it
  double collective_ms = 0; uint32_t collectives = 0;
};
RunStats g_last_run;        // published once per kmeans_cuda() call, under g_last_run_mutex (the Python module
std::mutex g_last_run_mutex;  // releases the GIL around the call: two threads may be inside the library)

class Job {
 public:
  uint32_t N = 0, D = 0, K = 0;
  int metric = 0, verbosity = 0;
  bool fp16 = false;  // fp16x2 boundary: half buffers outside, fp32 arithmetic on the half values inside
  // KMCUDA_AMD_FP16_STRICT=1 (fp16x2 jobs, one GPU): every step in the reference's half2 ARITHMETIC
  // (half2_strict.hip) -- the verification mode the oracle's half2 restatement is compared with
  bool strict_h2 = false;
  RunStats stats;   // this job's own counts (a nested group-clustering job has its own)
  std::vector<std::unique_ptr<Shard>> shards;
  ShardWorkers workers;   // (after `shards`: joined before the shards go)
  Rccl rccl;
  std::vector<void *> comms;
  // several shards on ONE device (KMCUDA_AMD_VIRTUAL_SHARDS): the all-reduce's stand-in is a sum kernel on the
  // first shard's stream, ordered with the others by events -- the dependency structure of the real collective
  double **reduce_ptrs_dev = nullptr;
  hipEvent_t ev_summed = nullptr;
  bool speculate = true;   // KMCUDA_AMD_SPECULATE=0: every iteration waits for its stop test before its update
  // KMCUDA_AMD_TIME_COLLECTIVE=1 (bench.py --api): a pair of HIP events around every iteration's all-reduce (its
  // stand-in under KMCUDA_AMD_VIRTUAL_SHARDS) on the first shard's stream, read when the loops are over -- so that a
  // multi-GPU run says by itself how long the collective takes (kmamd_last_run_collective)
  bool time_collective = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> coll_events;
  size_t coll_used = 0;
  int collective_mark(bool begin) {
    if (!time_collective) return 0;
    // (the caller's current device is its own business: the events belong to the first shard's device, and the device
    //  that was current comes back -- the grouped all-reduce sets its own per shard, the stand-in has set the first's)
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(shards[0]->dev);
    int rc = 0;
    if (begin) {
      if (coll_used == coll_events.size()) {
        if (coll_events.size() >= 4096) {   // (enough of a sample)
          time_collective = false;
        } else {
          hipEvent_t a = nullptr, b = nullptr;
          if (hipEventCreate(&a) != hipSuccess) {
            rc = kmcudaRuntimeError;
          } else if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);   // (a pair or nothing: ~Job destroys pairs)
            rc = kmcudaRuntimeError;
          } else {
            coll_events.emplace_back(a, b);
          }
        }
      }
      if (rc == 0 && time_collective &&
          hipEventRecord(coll_events[coll_used].first, shards[0]->eng->stream_) != hipSuccess)
        rc = kmcudaRuntimeError;
    } else if (coll_used < coll_events.size()) {
      if (hipEventRecord(coll_events[coll_used++].second, shards[0]->eng->stream_) != hipSuccess) rc = kmcudaRuntimeError;
    }
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    return rc;
  }
  void collective_collect() {   // after sync_all()
    for (size_t i = 0; i < coll_used; i++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, coll_events[i].first, coll_events[i].second) == hipSuccess) {
        stats.collective_ms += ms;
        stats.collectives++;
      }
    }
    coll_used = 0;
  }

  ~Job() {
    for (void *c : comms)
      if (c) rccl.CommDestroy(c);
    if (!shards.empty()) (void)hipSetDevice(shards[0]->dev);
    if (reduce_ptrs_dev) (void)hipFree(reduce_ptrs_dev);
    if (ev_summed) (void)hipEventDestroy(ev_summed);
    for (auto &e : coll_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }

  // on_stream: the (single) shard's engine works on this existing stream instead of creating one -- the nested job
  // that clusters the centroids into groups rides on its parent's: a stream's first launches cost milliseconds
  int setup(const std::vector<int> &devs, int nvirtual, uint32_t N_, uint32_t D_, uint32_t K_, int metric_,
            int verbosity_, const float *samples, int32_t device_ptrs, hipStream_t on_stream = nullptr) {
    N = N_; D = D_; K = K_; metric = metric_; verbosity = verbosity_;
    std::vector<int> shard_devs = devs;
    if (nvirtual > 1 && devs.size() == 1) shard_devs.assign(nvirtual, devs[0]);  // test hook
    auto plan = row_plan(N, shard_devs.size());
    for (size_t i = 0; i < shard_devs.size(); i++) {
      auto sh = std::make_unique<Shard>();
      sh->dev = shard_devs[i];
      sh->offset = plan[i].first;
      sh->length = plan[i].second;
      if (hipSetDevice(sh->dev) != hipSuccess) return kmcudaNoSuchDevice;
      sh->eng = std::make_unique<Engine>();
      int rc = sh->eng->init(sh->dev, sh->length, D, K, metric, 0, shard_devs.size() == 1 ? on_stream : nullptr);
      if (rc) return rc;
      sh->eng->strict_h2_ = strict_h2;
      const float *src = samples + (size_t)sh->offset * D;
      if (fp16) {
        // `samples` holds halves: stage the raw halves on the device, widen into the fp32 working copy
        const uint16_t *hsrc = reinterpret_cast<const uint16_t *>(samples) + (size_t)sh->offset * D;
        const size_t n = (size_t)sh->length * D;
        float *buf = nullptr;
        if ((rc = sh->alloc(&buf, n))) return rc;
        const uint16_t *dev_half = hsrc;
        uint16_t *tmp = nullptr;
        if (!(device_ptrs >= 0 && device_ptrs == sh->dev)) {
          if (hipMalloc((void **)&tmp, (n ? n : 1) * sizeof(uint16_t)) != hipSuccess) return kmcudaMemoryAllocationFailure;
          hipError_t e = device_ptrs < 0
                             ? hipMemcpyAsync(tmp, hsrc, n * sizeof(uint16_t), hipMemcpyHostToDevice, sh->eng->stream_)
                             : hipMemcpyPeerAsync(tmp, sh->dev, hsrc, device_ptrs, n * sizeof(uint16_t), sh->eng->stream_);
          if (e != hipSuccess) { (void)hipFree(tmp); return kmcudaMemoryCopyError; }
          dev_half = tmp;
        }
        hipError_t e = launch_half_to_float(dev_half, n, buf, sh->eng->stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(sh->eng->stream_);
        if (tmp) sh->owned.push_back(tmp);  // kept: the f16 matrix-core filter reads the rows as halves
        if (e != hipSuccess) return kmcudaRuntimeError;
        sh->samples = buf;
        sh->eng->half_rows_ = dev_half;
      } else if (device_ptrs >= 0 && device_ptrs == sh->dev) {
        sh->samples = src;  // already resident, used in place and never modified
      } else {
        float *buf = nullptr;
        if ((rc = sh->alloc(&buf, (size_t)sh->length * D))) return rc;
        hipError_t e;
        if (device_ptrs < 0)
          e = hipMemcpyAsync(buf, src, (size_t)sh->length * D * sizeof(float), hipMemcpyHostToDevice, sh->eng->stream_);
        else
          e = hipMemcpyPeerAsync(buf, sh->dev, src, device_ptrs, (size_t)sh->length * D * sizeof(float),
                                 sh->eng->stream_);
        if (e != hipSuccess) return kmcudaMemoryCopyError;
        sh->samples = buf;
      }
      if ((rc = sh->alloc(&sh->centroids, (size_t)K * D))) return rc;
      if ((rc = sh->alloc(&sh->assignments, sh->length))) return rc;
      if ((rc = sh->alloc(&sh->prev, sh->length))) return rc;
      if ((rc = sh->alloc(&sh->ccounts, K))) return rc;
      if ((rc = sh->alloc(&sh->reduce, (size_t)K * D + K + 4))) return rc;
      if ((rc = sh->alloc(&sh->dists, sh->length))) return rc;
      shards.push_back(std::move(sh));
    }
    bool distinct = shards.size() > 1;
    for (size_t i = 0; i < shards.size() && distinct; i++)
      for (size_t j = 0; j < i; j++)
        if (shards[i]->dev == shards[j]->dev) distinct = false;
    // KMCUDA_AMD_FORCE_RCCL=1 (test hook): a ONE-shard job also creates its (one-rank) communicator and sends its
    // reduce buffer through ncclAllReduce every iteration -- the library loading, the entry points' signatures, the
    // enum values and the stream the collective is enqueued on get exercised on a box with a single GPU
    const bool force_rccl = shards.size() == 1 && getenv("KMCUDA_AMD_FORCE_RCCL") && atoi(getenv("KMCUDA_AMD_FORCE_RCCL")) != 0;
    if (distinct || force_rccl) {
      if (!rccl.load()) {
        INFO("failed to load librccl.so: multi-GPU operation is unavailable\n");
        return kmcudaRuntimeError;
      }
      std::vector<int> ids;
      for (auto &s : shards) ids.push_back(s->dev);
      comms.resize(ids.size(), nullptr);
      if (rccl.CommInitAll(comms.data(), (int)ids.size(), ids.data()) != 0) return kmcudaRuntimeError;
    } else if (shards.size() > 1) {
      std::vector<double *> ptrs;
      for (auto &s : shards) {
        ptrs.push_back(s->reduce);
        (void)hipSetDevice(s->dev);
        if (hipEventCreateWithFlags(&s->ev_filled, hipEventDisableTiming) != hipSuccess) return kmcudaRuntimeError;
      }
      (void)hipSetDevice(shards[0]->dev);
      if (hipMalloc((void **)&reduce_ptrs_dev, ptrs.size() * sizeof(double *)) != hipSuccess)
        return kmcudaMemoryAllocationFailure;
      if (hipMemcpy(reduce_ptrs_dev, ptrs.data(), ptrs.size() * sizeof(double *), hipMemcpyHostToDevice) != hipSuccess)
        return kmcudaMemoryCopyError;
      if (hipEventCreateWithFlags(&ev_summed, hipEventDisableTiming) != hipSuccess) return kmcudaRuntimeError;
    }
    if (const char *v = getenv("KMCUDA_AMD_SPECULATE")) speculate = atoi(v) != 0;
    if (const char *v = getenv("KMCUDA_AMD_TIME_COLLECTIVE")) time_collective = atoi(v) != 0 && (shards.size() > 1 || !comms.empty());
    return sync_all();  // uploads complete: later cross-stream reads of the samples are safe
  }

  int sync_all() {
    for (auto &s : shards) RETERR(s->eng->sync());
    return 0;
  }

  // ---- replicated-state helpers ----
  int broadcast_centroids_from_host(const float *host) {
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      if (hipMemcpyAsync(s->centroids, host, (size_t)K * D * sizeof(float), hipMemcpyHostToDevice, s->eng->stream_) !=
          hipSuccess)
        return kmcudaMemoryCopyError;
    }
    return sync_all();
  }

  // copies sample row `index` (global numbering) into centroid slot `slot` on every shard
  int copy_sample_to_centroid(uint32_t index, uint32_t slot) {
    Shard *owner = nullptr;
    for (auto &s : shards)
      if (index >= s->offset && index < s->offset + s->length) owner = s.get();
    if (!owner) return kmcudaRuntimeError;
    const float *src = owner->samples + (size_t)(index - owner->offset) * D;
    for (auto &s : shards) {
      hipError_t e;
      float *dst = s->centroids + (size_t)slot * D;
      // the copy is enqueued on the DESTINATION shard's stream (ordered after anything pending there),
      // with that stream's device current
      (void)hipSetDevice(s->dev);
      if (s->dev == owner->dev)
        e = hipMemcpyAsync(dst, src, D * sizeof(float), hipMemcpyDeviceToDevice, s->eng->stream_);
      else
        e = hipMemcpyPeerAsync(dst, s->dev, src, owner->dev, D * sizeof(float), s->eng->stream_);
      if (e != hipSuccess) return kmcudaMemoryCopyError;
    }
    return 0;
  }

  int read_sample_value(uint32_t index, uint32_t feature, float *out) {
    for (auto &s : shards)
      if (index >= s->offset && index < s->offset + s->length) {
        (void)hipSetDevice(s->dev);
        RETERR(s->eng->sync());
        if (hipMemcpy(out, s->samples + (size_t)(index - s->offset) * D + feature, sizeof(float),
                      hipMemcpyDeviceToHost) != hipSuccess)
          return kmcudaMemoryCopyError;
        return 0;
      }
    return kmcudaRuntimeError;
  }

  int read_sample_row(uint32_t index, float *out) {
    for (auto &s : shards)
      if (index >= s->offset && index < s->offset + s->length) {
        (void)hipSetDevice(s->dev);
        RETERR(s->eng->sync());
        if (hipMemcpy(out, s->samples + (size_t)(index - s->offset) * D, D * sizeof(float), hipMemcpyDeviceToHost) !=
            hipSuccess)
          return kmcudaMemoryCopyError;
        return 0;
      }
    return kmcudaRuntimeError;
  }

  // ---- seeding (reference: kmeans_init_centroids, kmcuda.cc:189-400) ----
  int init_centroids(KMCUDAInitMethod method, uint32_t seed, const float *host_centroids, int32_t device_ptrs,
                     uint32_t afk_m = 0) {
    if (metric == kmcudaDistanceMetricCosine && !fp16) {  // kmcuda.cc:195-220: three unit-norm probes, fp32 only
      std::vector<float> probe(D);
      for (uint32_t s : {0u, N / 2, N - 1}) {
        RETERR(read_sample_row(s, probe.data()));
        double norm = 0;
        for (uint32_t i = 0; i < D; i++) norm += probe[i] * probe[i];
        const float high = 1.00001, low = 0.99999;
        if (norm > high || norm < low) {
          INFO("error: angular distance: samples[%u] has L2 norm = %f which is outside [%f, %f]\n", s, norm, low, high);
          return kmcudaInvalidArguments;
        }
      }
    }
    srand(seed);  // kmcuda.cc:222
    switch (method) {
      case kmcudaInitMethodImport: {
        if (fp16) {  // K x D halves -> fp32 replicas
          const size_t n = (size_t)K * D;
          for (auto &s : shards) {
            (void)hipSetDevice(s->dev);
            uint16_t *tmp = nullptr;
            if (hipMalloc((void **)&tmp, n * sizeof(uint16_t)) != hipSuccess) return kmcudaMemoryAllocationFailure;
            hipError_t e = device_ptrs < 0
                               ? hipMemcpy(tmp, host_centroids, n * sizeof(uint16_t), hipMemcpyHostToDevice)
                               : hipMemcpyPeer(tmp, s->dev, host_centroids, device_ptrs, n * sizeof(uint16_t));
            if (e == hipSuccess) e = launch_half_to_float(tmp, n, s->centroids, s->eng->stream_);
            if (e == hipSuccess) e = hipStreamSynchronize(s->eng->stream_);
            (void)hipFree(tmp);
            if (e != hipSuccess) return kmcudaMemoryCopyError;
          }
          return 0;
        }
        if (device_ptrs < 0) return broadcast_centroids_from_host(host_centroids);
        for (auto &s : shards) {
          (void)hipSetDevice(s->dev);
          hipError_t e = hipMemcpyPeerAsync(s->centroids, s->dev, host_centroids, device_ptrs,
                                            (size_t)K * D * sizeof(float), s->eng->stream_);
          if (e != hipSuccess) return kmcudaMemoryCopyError;
        }
        return sync_all();
      }
      case kmcudaInitMethodRandom: {
        INFO("randomly picking initial centroids...\n");
        // The reference shuffles ALL N indices with libstdc++'s std::random_shuffle over rand() and takes the first K
        // (kmcuda.cc:245-253): for i = 1 .. N-1: swap(a[i], a[rand() % (i + 1)]).  Position i is untouched before its
        // own step and a value that leaves the first K never returns, so the first K entries follow from a K-entry
        // array: steps i < K swap inside it, a step i >= K whose draw j falls below K puts i at j.  Same draws, same
        // rows, no N-entry array; with glibc's generator restated (GlibcRand: a rand() call is 10-20 ns behind its
        // lock -- 45 ms of a 4M-row call, 90 ms at 8M rows) the N - 1 draws take a few ms, and rand()'s own state is
        // set to what N - 1 calls would have left.
        std::vector<uint32_t> chosen(K);
        for (uint32_t c = 0; c < K; c++) chosen[c] = c;
        GlibcRand fast;
        if (fast.attach(seed)) {
          for (uint32_t i = 1; i < N && i < K; i++) std::swap(chosen[i], chosen[fast.next() % (i + 1)]);
          for (uint32_t i = K; i < N; i++) {
            const uint32_t j = fast.next() % (i + 1);
            if (j < K) chosen[j] = i;
          }
          fast.detach();
        } else {   // another C library: its rand(), call by call
          for (uint32_t i = 1; i < N && i < K; i++) std::swap(chosen[i], chosen[rand() % (i + 1)]);
          for (uint32_t i = K; i < N; i++) {
            const uint32_t j = rand() % (i + 1);
            if (j < K) chosen[j] = i;
          }
        }
        DEBUG("shuffle complete, copying to device(s)...\n");
        if (shards.size() == 1) {   // one gather launch (the index list rides in the dists buffer: N >= K floats)
          Shard &s = *shards[0];
          (void)hipSetDevice(s.dev);
          uint32_t *idx = reinterpret_cast<uint32_t *>(s.dists);
          if (hipMemcpyAsync(idx, chosen.data(), K * sizeof(uint32_t), hipMemcpyHostToDevice, s.eng->stream_) != hipSuccess)
            return kmcudaMemoryCopyError;
          if (launch_gather_rows(s.samples, idx, K, D, s.centroids, s.eng->stream_) != hipSuccess) return kmcudaRuntimeError;
          return sync_all();   // (chosen[] must outlive the copy)
        }
        for (uint32_t c = 0; c < K; c++) RETERR(copy_sample_to_centroid(chosen[c], c));
        return sync_all();
      }
      case kmcudaInitMethodPlusPlus: {
        float smoke = NAN;
        uint32_t first_index = 0;
        while (smoke != smoke) {  // kmcuda.cc:265-270
          first_index = rand() % N;
          RETERR(read_sample_value(first_index, 0, &smoke));
        }
        RETERR(copy_sample_to_centroid(first_index, 0));
        INFO("performing kmeans++...\n");
        // Two ways to the reference's choice (kmcuda.cc:286-326).  HOST: all N distances come back
        // (pinned, 4 N bytes per step), butterfly sum and sequential double prefix sums as the
        // reference.  DEVICE (one shard; seeding.hip): sums of floats in double are exact -- order free
        // -- while the distances' exponent range is narrow enough, so the step kernel's exact block
        // sums give the same choice from 32 bytes per step; a step whose range is too wide (or that
        // holds a NaN / inf distance) takes the host way.
        float *host_dists = nullptr;
        struct HostFree { float **p; ~HostFree() { if (*p) (void)hipHostFree(*p); } } host_dists_guard{&host_dists};
        // Several row shards (round 5): every shard runs the step on its rows and leaves its own exact block sums; ONE
        // chooser kernel on the first shard's device reads them where they lie (peer access) as the concatenation
        // they are and copies the seed's row into every replica (seeding.hip: kmpp_choose_kernel) -- no N-sized
        // transfer, no host in the loop.  Needs whole 256-row blocks on every shard but the last (row_plan()) and
        // peer access from the first shard's device to the others'; otherwise the host chooser below.
        bool device_chooser = !strict_h2 && getenv("KMCUDA_AMD_KMPP_HOST") == nullptr && shards.size() <= (size_t)kKmppMaxShards;
        for (auto &s : shards) {
          device_chooser = device_chooser && kmpp_blocks(s->length) <= (1u << 20) &&   // (two-level prefix: 1024 x 1024 blocks of 256 rows)
                           (s->offset % 256u == 0u) && s->length != 0;
          if (device_chooser && s->dev != shards[0]->dev) {
            int can = 0;
            (void)hipSetDevice(shards[0]->dev);
            if (hipDeviceCanAccessPeer(&can, shards[0]->dev, s->dev) != hipSuccess || !can) {
              device_chooser = false;
            } else {
              const hipError_t pe = hipDeviceEnablePeerAccess(s->dev, 0);
              if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) device_chooser = false;
            }
            (void)hipGetLastError();
          }
        }
        struct Totals { double sum_g, sum_d; uint32_t emin, emax, bad, chosen; };
        Totals *totals_host = nullptr;
        struct TotalsFree { Totals **p; ~TotalsFree() { if (*p) (void)hipHostFree(*p); } } totals_guard{&totals_host};
        // Per shard: block statistics, local prefixes, totals, the fail flag, an event behind its step -- and for the
        // filtered steps (seeding.hip) a centred byte copy of its rows (xs8; n2c: 4 floats per row): a step's first kernel
        // drops every row that provably is no closer to the new seed than to an earlier one, the exact chains run for
        // the rest.  Needs DP + 20 bytes per row beside the rows; without that memory (or KMCUDA_AMD_KMPP_FILTER=0) every
        // step is the plain one.  Same dists[] after every step either way, hence the same seeds.
        struct KppShard {
          int dev = 0;
          void *block_stats = nullptr, *totals = nullptr;
          double *bpre = nullptr;
          uint32_t *fail = nullptr;
          KmppOutlierBuf out;   // the step's exponent cut, the distances below it (seeding.hip: KmppOutlier)
          hipEvent_t ev_step = nullptr;
          void *xs8 = nullptr;
          float *n2c = nullptr, *mu = nullptr;
          uint32_t *stats = nullptr, *list = nullptr;
          double *part = nullptr;
          void release() {
            (void)hipSetDevice(dev);
            if (ev_step) (void)hipEventDestroy(ev_step);
            for (void *q : {xs8, (void *)n2c, (void *)mu, (void *)stats, (void *)list, (void *)part})
              if (q) (void)hipFree(q);
            ev_step = nullptr; xs8 = nullptr; n2c = nullptr; mu = nullptr; stats = nullptr; list = nullptr; part = nullptr;
          }
        };
        std::vector<KppShard> kpp(shards.size());
        struct KppFree { std::vector<KppShard> *v; ~KppFree() { for (auto &k : *v) k.release(); } } kpp_guard{&kpp};
        hipEvent_t ev_chosen = nullptr;
        struct EvFree { hipEvent_t *e; int dev; ~EvFree() { if (*e) { (void)hipSetDevice(dev); (void)hipEventDestroy(*e); } } } ev_guard{&ev_chosen, shards[0]->dev};
        if (device_chooser) {
          for (size_t q = 0; q < shards.size(); q++) {
            Shard &s = *shards[q];
            KppShard &k = kpp[q];
            k.dev = s.dev;
            (void)hipSetDevice(s.dev);
            unsigned char *bs = nullptr, *td = nullptr;
            int rc;
            if ((rc = s.alloc(&bs, kmpp_block_stat_bytes(s.length)))) return rc;
            if ((rc = s.alloc(&td, kmpp_totals_bytes()))) return rc;
            if ((rc = s.alloc(&k.bpre, kmpp_prefix_doubles(s.length)))) return rc;
            if ((rc = s.alloc(&k.fail, 4))) return rc;   // [0] the flag, [1] the exponent cut, [2] the listed values' number
            unsigned char *ob = nullptr;
            if ((rc = s.alloc(&ob, kmpp_outlier_bytes()))) return rc;
            k.block_stats = bs;
            k.totals = td;
            k.out.ecut = k.fail + 1;
            k.out.outl_count = k.fail + 2;
            k.out.outl = ob;
            if (hipMemsetAsync(k.fail, 0, 4 * sizeof(uint32_t), s.eng->stream_) != hipSuccess) return kmcudaRuntimeError;
            if (hipMemsetAsync(td, 0, kmpp_totals_bytes(), s.eng->stream_) != hipSuccess) return kmcudaRuntimeError;
            if (shards.size() > 1 && hipEventCreateWithFlags(&k.ev_step, hipEventDisableTiming) != hipSuccess) return kmcudaRuntimeError;
          }
          (void)hipSetDevice(shards[0]->dev);
          if (shards.size() > 1 && hipEventCreateWithFlags(&ev_chosen, hipEventDisableTiming) != hipSuccess) return kmcudaRuntimeError;
          if (hipHostMalloc(reinterpret_cast<void **>(&totals_host), sizeof(Totals), hipHostMallocDefault) != hipSuccess)
            return kmcudaMemoryAllocationFailure;
        }
        const uint32_t kpp_dp = ((uint32_t)D + 127u) / 128u * 128u;
        // (small jobs: the plain step is a few launches of nothing; KMCUDA_AMD_KMPP_FILTER=2 filters them too: tests)
        bool kpp_filter = device_chooser && K >= 8 && N >= 65536u && kpp_dp <= 8192u;
        if (const char *v = getenv("KMCUDA_AMD_KMPP_FILTER")) {
          const int f = atoi(v);
          kpp_filter = f >= 2 ? (device_chooser && K >= 3 && kpp_dp <= 8192u) : (kpp_filter && f != 0);
        }
        if (kpp_filter) {
          bool ok = true;
          for (size_t q = 0; q < shards.size() && ok; q++) {
            Shard &s = *shards[q];
            KppShard &k = kpp[q];
            (void)hipSetDevice(s.dev);
            ok = hipMalloc(&k.xs8, (size_t)s.length * kpp_dp) == hipSuccess &&   // (bytes)
                 hipMalloc(reinterpret_cast<void **>(&k.n2c), (size_t)s.length * 4 * sizeof(float)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&k.mu), (size_t)kpp_dp * sizeof(float)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&k.stats), 4 * sizeof(uint32_t)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&k.list), (size_t)s.length * sizeof(uint32_t)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&k.part), (size_t)64 * D * sizeof(double)) == hipSuccess;
          }
          if (!ok) {
            (void)hipGetLastError();
            kpp_filter = false;
            for (auto &k : kpp) {   // (the events stay: release() is the end of the seeding)
              (void)hipSetDevice(k.dev);
              for (void *q : {k.xs8, (void *)k.n2c, (void *)k.mu, (void *)k.stats, (void *)k.list, (void *)k.part})
                if (q) (void)hipFree(q);
              k.xs8 = nullptr; k.n2c = nullptr; k.mu = nullptr; k.stats = nullptr; k.list = nullptr; k.part = nullptr;
            }
            DEBUG("k-means++: no memory for the byte copy of the rows, plain steps\n");
          } else {
            for (size_t q = 0; q < shards.size(); q++) {
              Shard &s = *shards[q];
              KppShard &k = kpp[q];
              (void)hipSetDevice(s.dev);
              if (launch_kmpp_cache(s.samples, s.length, D, kpp_dp, k.part, k.mu, k.xs8, k.n2c, k.stats, s.eng->stream_) != hipSuccess)
                return kmcudaRuntimeError;
            }
          }
        }
        uint32_t log2n = 0;
        while ((1ull << log2n) < (uint64_t)N) log2n++;
        uint32_t host_steps = 0;
        auto progress = [&](uint32_t i) {
          if (verbosity > 1 || (verbosity > 0 && (K < 100 || i % (K / 100) == 0))) {
            printf("\rstep %d", i);
            fflush(stdout);
          }
        };
        // the reference's chooser on the host (kmcuda.cc:286-326) for step i with random number `choice`: all N
        // distances come back, butterfly sum, sequential double prefix sums; copies the seed
        auto host_choose = [&](uint32_t i, double choice) -> int {
          host_steps++;
          if (!host_dists &&
              hipHostMalloc(reinterpret_cast<void **>(&host_dists), (size_t)N * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return kmcudaMemoryAllocationFailure;
          for (auto &s : shards) {
            (void)hipSetDevice(s->dev);
            if (hipMemcpyAsync(host_dists + s->offset, s->dists, (size_t)s->length * sizeof(float),
                               hipMemcpyDeviceToHost, s->eng->stream_) != hipSuccess)
              return kmcudaMemoryCopyError;
          }
          RETERR(sync_all());
          const double dist_sum = butterfly_sum(host_dists, N);
          const uint32_t choice_approx = choice * N;
          const double choice_sum = choice * dist_sum;
          uint32_t j = 0;
          if (choice_approx < 100) {
            double dist_sum2 = 0;
            for (j = 0; j < N && dist_sum2 < choice_sum; j++) dist_sum2 += host_dists[j];
          } else {
            double dist_sum2 = 0;
            for (uint32_t t = 0; t < choice_approx; t++) dist_sum2 += host_dists[t];
            if (dist_sum2 < choice_sum) {
              for (j = choice_approx; j < N && dist_sum2 < choice_sum; j++) dist_sum2 += host_dists[j];
            } else {
              for (j = choice_approx; j > 1 && dist_sum2 >= choice_sum; j--) dist_sum2 -= host_dists[j];
              j++;
            }
          }
          if (j == 0 || j > N) {
            INFO("\ninternal bug in kmeans_init_centroids: j = %u\n", j);
            return kmcudaRuntimeError;
          }
          return copy_sample_to_centroid(j - 1, i);
        };
        if (device_chooser) {
          // The host only draws the random numbers (one rand() per step, in order, as the reference) and enqueues:
          // distances, statistics, prefix sums, the chooser and the copy of the chosen row all run on the device
          // (seeding.hip).  A step the device cannot decide raises a flag that turns the rest of the enqueued
          // kernels into no-ops; the host looks after every batch of steps, chooses that seed the reference's way from the
          // distances as that step left them, lowers the flag and goes on behind it.
          // Several shards: one host thread enqueues every shard's step, the chooser on the first shard's stream behind
          // events of the others' steps, and the others' next steps behind an event of the chooser (the events must be
          // RECORDED before they are waited for: one thread, in order).
          Shard &lead = *shards[0];
          const bool many = shards.size() > 1;
          std::vector<KmppShardPtrs> views(shards.size());
          for (size_t q = 0; q < shards.size(); q++) {
            Shard &s = *shards[q];
            views[q].dists = s.dists; views[q].bpre = kpp[q].bpre; views[q].totals = kpp[q].totals;
            views[q].samples = s.samples; views[q].centroids = s.centroids; views[q].fail = kpp[q].fail;
            views[q].offset = s.offset; views[q].length = s.length;
            views[q].out = kpp[q].out;
          }
          if (many) RETERR(sync_all());   // (the first seed's peer copies and the flags' memsets have landed everywhere)
          std::vector<double> choices(K, 0.0);
          // (steps enqueued behind an undecided one are wasted launches: the batch starts small, doubles while the
          //  device decides, and falls back to one step after a hand-over -- a data set whose distances span too many
          //  binades hands over every step)
          constexpr uint32_t kBatchMax = 256;
          uint32_t batch = 8;
          uint32_t drawn = 1;   // choices[1 .. drawn) are drawn
          uint32_t i = 1;
          while (i < K) {
            const uint32_t end = i + batch < K ? i + batch : K;
            for (uint32_t t = i; t < end; t++) {
              progress(t);
              if (t >= drawn) {
                choices[t] = ((rand() + .0) / RAND_MAX);   // kmcuda.cc:300
                drawn = t + 1;
              }
              for (size_t q = 0; q < shards.size(); q++) {
                Shard &s = *shards[q];
                KppShard &k = kpp[q];
                (void)hipSetDevice(s.dev);
                hipStream_t st = s.eng->stream_;
                const float *newest = s.centroids + (size_t)(t - 1) * D;
                const hipError_t se =
                    (kpp_filter && t >= 2)
                        ? launch_kmpp_step_filtered(metric, s.samples, s.length, D, kpp_dp, k.xs8, k.n2c, k.mu, k.stats, k.list,
                                                    newest, t, s.dists, k.block_stats, k.bpre, k.totals, k.fail, k.out, st)
                        : launch_kmpp_step2(metric, s.samples, s.length, D, newest, t, s.dists, k.block_stats, k.bpre, k.totals,
                                            k.fail, k.out, st);
                if (se != hipSuccess) return kmcudaRuntimeError;
                if (many && q != 0 && hipEventRecord(k.ev_step, st) != hipSuccess) return kmcudaRuntimeError;
              }
              (void)hipSetDevice(lead.dev);
              for (size_t q = 1; q < shards.size(); q++)
                if (hipStreamWaitEvent(lead.eng->stream_, kpp[q].ev_step, 0) != hipSuccess) return kmcudaRuntimeError;
              if (launch_kmpp_choose(views.data(), (uint32_t)views.size(), N, choices[t], log2n, t, D, lead.eng->stream_) != hipSuccess)
                return kmcudaRuntimeError;
              if (many) {
                if (hipEventRecord(ev_chosen, lead.eng->stream_) != hipSuccess) return kmcudaRuntimeError;
                for (size_t q = 1; q < shards.size(); q++) {
                  (void)hipSetDevice(shards[q]->dev);
                  if (hipStreamWaitEvent(shards[q]->eng->stream_, ev_chosen, 0) != hipSuccess) return kmcudaRuntimeError;
                }
              }
            }
            uint32_t failed = 0;
            (void)hipSetDevice(lead.dev);
            if (hipMemcpyAsync(totals_host, kpp[0].fail, sizeof(uint32_t), hipMemcpyDeviceToHost, lead.eng->stream_) != hipSuccess ||
                hipStreamSynchronize(lead.eng->stream_) != hipSuccess)
              return kmcudaMemoryCopyError;
            memcpy(&failed, totals_host, sizeof(uint32_t));
            if (failed == 0) {
              i = end;
              batch = batch * 2 < kBatchMax ? batch * 2 : kBatchMax;
              continue;
            }
            batch = 1;
            if (failed < i || failed >= end) return kmcudaRuntimeError;
            RETERR(host_choose(failed, choices[failed]));   // (waits for every shard: sync_all)
            for (size_t q = 0; q < shards.size(); q++) {
              (void)hipSetDevice(shards[q]->dev);
              if (hipMemsetAsync(kpp[q].fail, 0, sizeof(uint32_t), shards[q]->eng->stream_) != hipSuccess) return kmcudaRuntimeError;
            }
            if (many) RETERR(sync_all());   // (the seed's peer copies and the lowered flags, before the chooser reads across shards)
            i = failed + 1;
          }
        } else {
          for (uint32_t i = 1; i < K; i++) {
            progress(i);
            for (auto &s : shards) {
              (void)hipSetDevice(s->dev);
              const hipError_t ke = strict_h2
                  ? launch_h2_to_row(metric, s->samples, s->length, D, s->centroids + (size_t)(i - 1) * D, i, 0, s->dists,
                                     s->eng->stream_)
                  : launch_kmpp_step(metric, s->samples, s->length, D, s->centroids + (size_t)(i - 1) * D, i, s->dists,
                                     s->eng->stream_);
              if (ke != hipSuccess) return kmcudaRuntimeError;
            }
            RETERR(host_choose(i, ((rand() + .0) / RAND_MAX)));
          }
        }
        if (device_chooser) DEBUG("k-means++: %u of %u steps took the host chooser\n", host_steps, K - 1);
        RETERR(sync_all());
        if (kpp_filter) {
          unsigned long long chains = 0;
          bool read = true;
          for (auto &k : kpp) {
            unsigned long long v = 0;
            (void)hipSetDevice(k.dev);
            read = read && hipMemcpy(&v, k.stats + 2, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess;
            chains += v;
          }
          if (read)
            DEBUG("k-means++: the filtered steps ran %llu exact chains for %llu (row, step) pairs\n", chains,
                  (unsigned long long)N * (K - 2));
        }
        break;
      }
      case kmcudaInitMethodAFKMC2: {   // kmcuda.cc:337-396
        uint32_t m = afk_m;
        if (m == 0) {
          m = 200;
        } else if (m > N / 2) {
          INFO("afkmc2: m > %u is not supported (got %u)\n", N / 2, m);
          return kmcudaInvalidArguments;
        }
        float smoke = NAN;
        uint32_t first_index = 0;
        while (smoke != smoke) {
          first_index = rand() % N;
          RETERR(read_sample_value(first_index, 0, &smoke));
        }
        INFO("afkmc2: calculating q (c0 = %u)... ", first_index / D);
        RETERR(copy_sample_to_centroid(first_index, 0));
        // the candidate kernels index rows globally: with several shards the rows are gathered on the first
        Shard &s0 = *shards[0];
        (void)hipSetDevice(s0.dev);
        hipStream_t st = s0.eng->stream_;
        const float *all = s0.samples;
        if (shards.size() > 1) {
          float *gathered = nullptr;
          int rc = s0.alloc(&gathered, (size_t)N * D);
          if (rc) return rc;
          RETERR(sync_all());
          for (auto &s : shards) {
            const hipError_t e = s->dev == s0.dev
                ? hipMemcpy(gathered + (size_t)s->offset * D, s->samples, (size_t)s->length * D * sizeof(float), hipMemcpyDeviceToDevice)
                : hipMemcpyPeer(gathered + (size_t)s->offset * D, s0.dev, s->samples, s->dev, (size_t)s->length * D * sizeof(float));
            if (e != hipSuccess) return kmcudaMemoryCopyError;
          }
          all = gathered;
        }
        float *qdev = nullptr, *rand_dev = nullptr, *mind_dev = nullptr;
        uint32_t *choice_dev = nullptr;
        {
          int rc;
          if ((rc = s0.alloc(&qdev, N))) return rc;
          if ((rc = s0.alloc(&rand_dev, m))) return rc;
          if ((rc = s0.alloc(&mind_dev, m))) return rc;
          if ((rc = s0.alloc(&choice_dev, m))) return rc;
        }
        if (hipMemsetAsync(choice_dev, 0, m * sizeof(uint32_t), st) != hipSuccess) return kmcudaRuntimeError;
        std::vector<float> q(N), rand_a(m), p_cand(m);
        std::vector<uint32_t> cand_ind(m);
        // q(x) = 1 / 2N + d(x, c1)^2 / (2 sum d^2), with c1 = sample (first_index / D): the reference
        // seeds q from THAT sample while centroid 0 is sample first_index (kmcuda.cc:356-362)
        if ((strict_h2 ? launch_h2_to_row(metric, all, N, D, all + (size_t)(first_index / D) * D, 0, 1, qdev, st)
                       : launch_afk_qdist(metric, all, N, D, all + (size_t)(first_index / D) * D, qdev, st)) != hipSuccess)
          return kmcudaRuntimeError;
        if (hipMemcpyAsync(q.data(), qdev, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
          return kmcudaMemoryCopyError;
        const double dsum = butterfly_sum(q.data(), N);   // kmeans.cu:92-95 (warp sums, double accumulation)
        if (launch_afk_q(qdev, N, (float)dsum, st) != hipSuccess) return kmcudaRuntimeError;
        if (hipMemcpyAsync(q.data(), qdev, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
          return kmcudaMemoryCopyError;
        INFO("done\n");
        for (uint32_t k = 1; k < K; k++) {
          if (verbosity > 1 || (verbosity > 0 && (K < 100 || k % (K / 100) == 0))) {
            printf("\rstep %d", k);
            fflush(stdout);
          }
          (void)hipSetDevice(s0.dev);
          if (launch_afk_random_step(m, seed, k, qdev, N, choice_dev, rand_dev, st) != hipSuccess ||
              (strict_h2 ? launch_h2_afk_min_dist(metric, m, k, all, D, choice_dev, s0.centroids, mind_dev, st)
                         : launch_afk_min_dist(metric, m, k, all, D, choice_dev, s0.centroids, mind_dev, st)) != hipSuccess)
            return kmcudaRuntimeError;
          if (hipMemcpyAsync(cand_ind.data(), choice_dev, m * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(rand_a.data(), rand_dev, m * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipMemcpyAsync(p_cand.data(), mind_dev, m * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
              hipStreamSynchronize(st) != hipSuccess)
            return kmcudaMemoryCopyError;
          float curr_prob = 0;   // the Metropolis-Hastings chain, kmcuda.cc:381-388
          uint32_t curr_ind = 0;
          for (uint32_t j = 0; j < m; j++) {
            const float cand_prob = p_cand[j] / q[cand_ind[j]];
            if (curr_prob == 0 || cand_prob / curr_prob > rand_a[j]) {
              curr_ind = j;
              curr_prob = cand_prob;
            }
          }
          RETERR(copy_sample_to_centroid(cand_ind[curr_ind], k));
        }
        RETERR(sync_all());
        break;
      }
    }
    INFO("\rdone            \n");
    return 0;
  }

  // runs fn(shard) for every shard -- on one host thread per shard when there are several, so that a
  // step that has to wait for its GPU (the first iterations' update reads an event count) does not hold
  // up the enqueueing for the other GPUs.  Returns the first non-zero code.
  int for_shards(const std::function<int(Shard &)> &fn) {
    if (shards.size() == 1) {
      (void)hipSetDevice(shards[0]->dev);
      return fn(*shards[0]);
    }
    if (workers.size() == 0) workers.start(shards.size());   // persistent: one thread per shard, for the job's lifetime
    const std::function<int(size_t)> each = [&](size_t i) {
      (void)hipSetDevice(shards[i]->dev);
      return fn(*shards[i]);
    };
    return workers.run(each);
  }

  // ---- the per-iteration collective: ONE all-reduce of [delta (fp64 K*D) | dcount K | counters 4] ----
  int allreduce_fused() {
    if (shards.size() == 1 && comms.empty()) return 0;
    const size_t len = (size_t)K * D + K + 4;
    if (!comms.empty()) {
      RETERR(collective_mark(true));
      if (rccl.GroupStart() != 0) return kmcudaRuntimeError;
      for (size_t i = 0; i < shards.size(); i++) {
        auto &s = shards[i];
        (void)hipSetDevice(s->dev);
        if (rccl.AllReduce(s->reduce, s->reduce, len, kNcclFloat64, kNcclSum, comms[i], s->eng->stream_) != 0)
          return kmcudaRuntimeError;
      }
      if (rccl.GroupEnd() != 0) return kmcudaRuntimeError;
      return collective_mark(false);
    }
    // several shards on ONE device (test hook KMCUDA_AMD_VIRTUAL_SHARDS): a fixed-order sum kernel on the first
    // shard's stream once every shard's buffer is written; the other streams go on when it is done.  No host wait:
    // the host loop is exercised exactly as with the real collective
    Shard &f = *shards[0];
    (void)hipSetDevice(f.dev);
    for (auto &s : shards) {
      if (hipEventRecord(s->ev_filled, s->eng->stream_) != hipSuccess) return kmcudaRuntimeError;
      if (s.get() != &f && hipStreamWaitEvent(f.eng->stream_, s->ev_filled, 0) != hipSuccess) return kmcudaRuntimeError;
    }
    RETERR(collective_mark(true));   // (behind the waits: the stand-in's own time, not the slowest shard's lag)
    if (launch_sum_buffers(reduce_ptrs_dev, (uint32_t)shards.size(), len, f.eng->stream_) != hipSuccess)
      return kmcudaRuntimeError;
    RETERR(collective_mark(false));
    if (hipEventRecord(ev_summed, f.eng->stream_) != hipSuccess) return kmcudaRuntimeError;
    for (auto &s : shards)
      if (s.get() != &f && hipStreamWaitEvent(s->eng->stream_, ev_summed, 0) != hipSuccess) return kmcudaRuntimeError;
    return 0;
  }

  // reference: check_changed, kmeans.cu:697-717.  Returns 1 to stop, 0 to go on, <0 error.
  // Reads the shards' device counters (one stream synchronisation per shard).
  int check_changed(int iter, float tolerance, bool print, uint32_t *passed_total = nullptr) {
    uint32_t overall_changed = 0, overall_passed = 0, pair_rows = 0, scan_rows = 0;
    for (auto &s : shards) {
      uint32_t c[4];
      if (s->eng->counters_read(c) != 0) return -kmcudaMemoryCopyError;
      overall_changed += c[0];
      overall_passed += c[2];
      scan_rows += c[1];
      pair_rows += c[3];
    }
    return judge_changed(iter, tolerance, print, overall_changed, overall_passed, pair_rows, scan_rows, passed_total);
  }

  // the same test from the REDUCED buffer's tail (the counters rode along in the iteration's all-reduce):
  // one 32-byte read on the first shard instead of a synchronisation per GPU
  int check_changed_reduced(int iter, float tolerance, bool print, uint32_t *passed_total = nullptr) {
    Shard &f = *shards[0];
    (void)hipSetDevice(f.dev);
    double tail[4];
    if (hipMemcpyAsync(tail, f.reduce + (size_t)K * D + K, sizeof(tail), hipMemcpyDeviceToHost, f.eng->stream_) != hipSuccess ||
        hipStreamSynchronize(f.eng->stream_) != hipSuccess)
      return -kmcudaMemoryCopyError;
    return judge_changed(iter, tolerance, print, (uint32_t)tail[0], (uint32_t)tail[2], (uint32_t)tail[3], (uint32_t)tail[1],
                         passed_total);
  }

  int judge_changed(int iter, float tolerance, bool print, uint32_t overall_changed, uint32_t overall_passed,
                    uint32_t pair_rows, uint32_t scan_rows, uint32_t *passed_total) {
    if (print && passed_total == nullptr)
      DEBUG("filter: %u rows settled by two exact chains, %u by a full exact scan\n", pair_rows, scan_rows);
    if (print && passed_total != nullptr)
      DEBUG("local filter: %u exact distances queued, %u wave flushes\n", pair_rows, scan_rows);
    if (passed_total) *passed_total = overall_passed;
    if (print) INFO("iteration %d: %u reassignments\n", iter, overall_changed);
    stamp(iter);
    for (auto &s : shards) s->eng->carry_policy_.note_changed(overall_changed);   // (how fast the run converges: engine.hpp)
    if (overall_changed <= tolerance * N) return 1;  // counters are NOT zeroed on stop (kmeans.cu:707-709)
    for (auto &s : shards)
      if (s->eng->counters_reset(0) != 0) return -kmcudaRuntimeError;
    return 0;
  }

  // KMCUDA_AMD_TIMING: when the host learned an iteration's outcome (ms since the job was made; a measurement aid)
  const bool timing_ = getenv("KMCUDA_AMD_TIMING") != nullptr;
  const std::chrono::steady_clock::time_point born_ = std::chrono::steady_clock::now();
  void stamp(int iter) const {
    if (timing_)
      fprintf(stderr, "[timing]   iteration %d judged at %.3f ms\n", iter,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - born_).count());
  }
  bool exact_update = false;  // KMCUDA_AMD_EXACT_UPDATE=1: the reference's serial Kahan chain (single shard)

  // Default schedule: iteration number from which on a judged "goes on" starts the carried bounds (0: no such
  // start pending; kmeans_cuda sets it at the hand-over point)
  int carry_after = 0;
  void maybe_start_carry(int judged_iter) {
    if (carry_after == 0 || judged_iter < carry_after) return;
    carry_after = 0;
    INFO("carrying per-sample distance bounds from pass to pass\n");
    for (auto &s : shards) {
      s->eng->carry_on_ = true;
      s->eng->carry_valid_ = false;
    }
  }

  // the update in three stream-ordered phases (reference: kmeans_adjust launch + peer exchange,
  // kmeans.cu:1002-1024): fill every shard's reduce buffer, ONE all-reduce, apply on every shard
  int fill_deltas() {
    const uint32_t kd = K * D;
    return for_shards([kd](Shard &s) {
      return s.eng->move_deltas(s.samples, s.prev, s.assignments, s.reduce, nullptr, s.reduce + kd);
    });
  }
  // stop_threshold >= 0: every shard's apply kernel evaluates the stop rule itself from the reduced counters (the
  // same words on every shard) and leaves everything untouched when it fires; report: the first shard's kernel
  // reports the outcome to its engine's pinned words (Engine::stop_report(seq) reads them)
  int apply_deltas(float stop_threshold = -1.f, bool report = false, uint32_t seq = 0) {
    const uint32_t kd = K * D;
    for (size_t i = 0; i < shards.size(); i++) {
      auto &s = shards[i];
      // (the next pass's centroid preparation rides in the same launch where it can: Engine::apply_prepare;
      // fp16x2 rounds the centroids to halves first, so there it cannot)
      if (fp16)
        RETERR(s->eng->apply_delta(s->reduce, nullptr, s->reduce + kd, s->centroids, s->ccounts, stop_threshold,
                                   report && i == 0, seq));
      else
        RETERR(s->eng->apply_prepare(s->reduce, s->reduce + kd, s->centroids, s->ccounts, stop_threshold,
                                     report && i == 0, seq));
    }
    return quantize_centroids();
  }

  int adjust() {
    if (exact_update) {
      Shard &s = *shards[0];
      RETERR(s.eng->adjust_exact(s.samples, s.prev, s.assignments, s.centroids, s.ccounts));
      return quantize_centroids();
    }
    RETERR(fill_deltas());
    RETERR(allreduce_fused());
    return apply_deltas();
  }

  // fp16x2: the reference keeps centroids in half2, i.e. every update is rounded to half
  int quantize_centroids() {
    if (!fp16) return 0;
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      if (launch_quantize_half(s->centroids, (size_t)K * D, s->eng->stream_) != hipSuccess) return kmcudaRuntimeError;
    }
    return 0;
  }

  // reference: prepare_mem, kmeans.cu:719-746
  int prepare_mem(bool resume) {
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      RETERR(s->eng->counters_reset(0));
      if (!resume) {
        hipStream_t st = s->eng->stream_;
        if (hipMemsetAsync(s->ccounts, 0, K * sizeof(uint32_t), st) != hipSuccess) return kmcudaRuntimeError;
        if (hipMemsetAsync(s->assignments, 0xff, (size_t)s->length * sizeof(uint32_t), st) != hipSuccess)
          return kmcudaRuntimeError;
        if (hipMemsetAsync(s->prev, 0xff, (size_t)s->length * sizeof(uint32_t), st) != hipSuccess)
          return kmcudaRuntimeError;
      }
    }
    return 0;
  }

  // reference: kmeans_cuda_lloyd, kmeans.cu:934-1026.  Per iteration: assignment on every shard, the
  // shards' move sums, ONE all-reduce that also carries the reassignment counters, the stop test on the
  // reduced counters (before the update, as kmeans.cu:991-1000), the update.
  //
  // The stop test is decided ON THE DEVICE by the update kernel (update.hip: apply_delta_kernel + StopCtl), so
  // the host enqueues iteration i + 1 before it has seen iteration i's count: the GPUs never wait for the 32-byte
  // round trip (on a 1M-row shard it is worth several percent of an iteration).  When the rule fires the update
  // kernel leaves everything untouched and raises a device flag that turns the already-enqueued next pass into
  // no-ops, so the state is exactly what the reference returns (assignments of the stop iteration, centroids one
  // update behind); the host reads the outcome one iteration late from pinned words and prints the same lines.
  //   leave:  asked after every iteration that goes on; true = "return BEFORE the next update" (the caller wants
  //           to start Yinyang from the reference's hand-over state).  It takes effect one iteration later: that
  //           iteration is judged by the host before its update, as in the reference.
  //   *left:  the loop returned for `leave`, not for the stop rule
  using LeaveFn = std::function<bool(int iter, uint32_t changed)>;
  //   after:  0 = a fresh run.  > 0: carry on after iteration `after`, whose stop test said "go on" and whose
  //           (reduced) move sums still sit in the reduce buffers: its update first, then iteration after + 1
  int lloyd(float tolerance, int *iterations, const LeaveFn *leave = nullptr, bool *left = nullptr, int after = 0) {
    if (after == 0) RETERR(prepare_mem(false));
    if (left) *left = false;
    // the samples do not change inside one kmeans_cuda() call: let the coarse filter stage keep its
    // centred half copy of the rows across the iterations (kmamd_set_row_cache, include/kmcuda_amd.h)
    for (auto &s : shards) {
      if (after == 0) {
        s->eng->row_cache_on_ = s->eng->row_cache_allowed_;
        s->eng->row_cache_valid_ = false;
      }
      RETERR(s->eng->stop_clear());
    }
    if (exact_update) {   // strict-parity mode: the reference's sequence, step by step
      if (after) RETERR(adjust());
      for (int iter = after + 1;; iter++) {
        for (auto &s : shards)
          RETERR(s->eng->lloyd_assign(s->samples, s->centroids, s->assignments, s->prev, false));
        const int status = check_changed(iter, tolerance, true);
        if (status < 0) return -status;
        stats.iterations++;
        if (status == 1) {
          if (iterations) *iterations = iter;
          return 0;
        }
        maybe_start_carry(iter);
        RETERR(adjust());
      }
    }
    const float threshold = tolerance * N;   // the float product of kmeans.cu:707
    bool leave_next = false;
    int unjudged = 0;   // a speculative iteration whose outcome the host has not looked at yet
    if (after) {
      for (auto &s : shards) RETERR(s->eng->counters_reset(0));   // (the stop that ended the last run left it, kmeans.cu:707-709)
      RETERR(apply_deltas());
    }
    for (int iter = after + 1;; iter++) {
      const bool spec = speculate && !leave_next;
      RETERR(for_shards([](Shard &s) {
        const int rc = s.eng->lloyd_assign(s.samples, s.centroids, s.assignments, s.prev, false);
        if (rc) return rc;
        const uint32_t kd = s.eng->K_ * s.eng->D_;
        return s.eng->move_deltas(s.samples, s.prev, s.assignments, s.reduce, nullptr, s.reduce + kd);
      }));
      RETERR(allreduce_fused());
      if (spec) RETERR(apply_deltas(threshold, true, (uint32_t)iter));
      if (unjudged) {
        uint32_t changed = 0;
        const int status = judge_reported(unjudged, &changed);
        if (status < 0) return -status;
        if (status == 1) {   // the pass enqueued above found the flag raised: nothing was touched
          if (iterations) *iterations = unjudged;
          return 0;
        }
        if (leave && (*leave)(unjudged, changed)) leave_next = true;
        maybe_start_carry(unjudged);   // (the run has gone on past iteration `unjudged`)
        unjudged = 0;
      }
      if (spec) {
        unjudged = iter;
        continue;
      }
      const int status = check_changed_reduced(iter, tolerance, true);   // the host waits: one 32-byte read
      if (status < 0) return -status;
      stats.iterations++;
      if (status == 1 || leave_next) {
        if (left) *left = status != 1;
        if (iterations) *iterations = iter;
        return 0;
      }
      maybe_start_carry(iter);
      RETERR(apply_deltas(threshold));   // (the same decision once more on the device; zeroes the counter)
    }
  }

  // the outcome of speculative iteration `iter`, as the first shard's update kernel reported it.  1: stopped, 0: went on
  int judge_reported(int iter, uint32_t *changed) {
    uint32_t t[6];
    if (shards[0]->eng->stop_report((uint32_t)iter, t) != 0) {
      INFO("internal error: no report of iteration %d\n", iter);
      return -kmcudaRuntimeError;
    }
    DEBUG("filter: %u rows settled by two exact chains, %u by a full exact scan\n", t[3], t[1]);
    INFO("iteration %d: %u reassignments\n", iter, t[0]);
    stamp(iter);
    if (changed) *changed = t[0];
    for (auto &s : shards) s->eng->carry_policy_.note_changed(t[0]);
    stats.iterations++;
    return t[4] ? 1 : 0;
  }

  // ---- Yinyang (reference: kmeans_cuda_yy, kmeans.cu:1028-1263) ----
  // Clusters the K centroids into G groups: k-means++ (srand(0), kmeans.cu:1081-1084 passes seed 0)
  // followed by Lloyd at YINYANG_GROUP_TOLERANCE on the centroids themselves (kmeans.cu:1062-1100).
  // Runs as a nested single-GPU job whose "samples" are shard 0's centroid replica.
  int cluster_groups(uint32_t G, std::vector<uint32_t> *groups) {
    Shard &first = *shards[0];
    RETERR(first.eng->sync());
    const bool timing = getenv("KMCUDA_AMD_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "[timing] group clustering, %s: %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
      t0 = t1;
    };
    Job gjob;
    gjob.strict_h2 = strict_h2;   // (before setup: the engines take the flag there)
    std::vector<int> one{first.dev};
    RETERR(gjob.setup(one, 0, K, D, G, metric, verbosity, first.centroids, first.dev, first.eng->stream_));  // fp32 replica in place
    gjob.shards[0]->eng->side_stream_ = first.eng->side_stream_;   // (borrowed with the main one; may be null)
    lap("setup");
    gjob.exact_update = exact_update;
    gjob.fp16 = fp16;  // centroids_yy is half2 in the reference too (kmeans.cu:1084-1091)
    RETERR(gjob.init_centroids(kmcudaInitMethodPlusPlus, 0, nullptr, first.dev));
    if (timing) RETERR(gjob.sync_all());
    lap("k-means++");
    RETERR(gjob.lloyd((float)kYinyangGroupTolerance, nullptr));
    RETERR(gjob.sync_all());
    lap("Lloyd");
    groups->resize(K);
    (void)hipSetDevice(first.dev);
    if (hipMemcpy(groups->data(), gjob.shards[0]->assignments, K * sizeof(uint32_t), hipMemcpyDeviceToHost) !=
        hipSuccess)
      return kmcudaMemoryCopyError;
    return 0;
  }

  // rand() as cluster_groups() leaves it: srand(0) (kmeans.cu:1081-1084 passes seed 0 to the nested k-means++), the
  // draws for its first seed -- a centroid whose first feature is not NaN, kmcuda.cc:265-270 -- and one draw per further
  // seed (kmcuda.cc:300)
  int replay_group_seeding_draws(uint32_t G) {
    Shard &first = *shards[0];
    srand(0);
    float smoke = NAN;
    while (smoke != smoke) {
      const uint32_t idx = rand() % K;
      (void)hipSetDevice(first.dev);
      RETERR(first.eng->sync());
      if (hipMemcpy(&smoke, first.centroids + (size_t)idx * D, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return kmcudaMemoryCopyError;
    }
    for (uint32_t t = 1; t < G; t++) (void)rand();
    return 0;
  }

  int yinyang(float tolerance, uint32_t G, int iter, const std::vector<uint32_t> &groups) {
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      RETERR(s->eng->stop_clear());   // (the Lloyd phase's stop raised it; this phase decides on the host)
      RETERR(s->eng->yy_configure(G, groups.data()));
      int rc;
      if ((rc = s->alloc(&s->bounds, (size_t)s->length * (G + 1)))) return rc;
      if ((rc = s->alloc(&s->drifts, (size_t)K * D + K))) return rc;
      if ((rc = s->alloc(&s->gdrifts, G))) return rc;
      if ((rc = s->alloc(&s->passed, s->length))) return rc;
    }
    RETERR(sync_all());  // host vectors above go out of use
    RETERR(prepare_mem(true));
    bool refresh = true;
    // Per iteration (kmeans.cu:1112-1260): stop test on the filters' counters, bounds refresh when nearly every row
    // passed, the update, the drifts, the two filters.  With several shards the counters (reassigned, passed) ride
    // in the update's ONE all-reduce -- the shards' move sums are formed and reduced BEFORE the test, which then is
    // one 32-byte read on the first shard instead of a stream synchronisation per GPU (round 3: check_changed) --
    // and every shard's kernels are enqueued by its own worker thread (for_shards), as in lloyd().  The strict
    // update (one serial chain over all rows, single shard) has no reduce buffer and keeps the plain sequence.
    for (;; iter++) {
      if (!exact_update) {
        RETERR(fill_deltas());
        RETERR(allreduce_fused());
      }
      if (!refresh) {
        uint32_t passed_total = 0;
        const int status = exact_update ? check_changed(iter, tolerance, true, &passed_total)
                                        : check_changed_reduced(iter, tolerance, true, &passed_total);
        if (status < 0) return -status;
        if (status == 1) {
          if (verbosity > 1) {
            for (auto &s : shards) {
              uint32_t hs[6] = {0, 0, 0, 0, 0, 0};
              (void)hipSetDevice(s->dev);
              if (s->eng->yy_hint_stats(hs) == 0 && hs[0])
                printf("local filter with the second-best estimate: %u rows, %u of them handed to the plain kernel "
                       "(no estimate %u, bound not holding %u, candidate bound %u, second minimum %u)\n",
                       hs[0], hs[1], hs[2], hs[3], hs[4], hs[5]);
            }
          }
          return 0;
        }
        DEBUG("passed number: %u\n", passed_total);
        if (1.f - (passed_total + 0.f) / N < kYinyangRefreshEpsilon) refresh = true;  // kmeans.cu:1136-1138
      }
      if (refresh) {
        INFO("refreshing Yinyang bounds...\n");
        RETERR(for_shards([](Shard &s) { return s.eng->yy_init(s.samples, s.centroids, s.assignments, s.bounds); }));
        refresh = false;
      }
      const size_t kd_bytes = (size_t)K * D * sizeof(float);
      RETERR(for_shards([kd_bytes](Shard &s) {   // kmeans.cu:1159: keep the old centroids for the drifts
        return hipMemcpyAsync(s.drifts, s.centroids, kd_bytes, hipMemcpyDeviceToDevice, s.eng->stream_) == hipSuccess
                   ? 0 : (int)kmcudaMemoryCopyError;
      }));
      if (exact_update) RETERR(adjust());
      else RETERR(apply_deltas());
      RETERR(for_shards([](Shard &s) {
        int rc = s.eng->yy_drifts(s.centroids, s.drifts, s.gdrifts);
        if (rc) return rc;
        if ((rc = s.eng->counters_reset(2))) return rc;  // d_passed_number = 0, kmeans.cu:1225-1229
        if ((rc = s.eng->counters_reset(1))) return rc;  // statistics of the local filter (flushes, exact distances)
        if ((rc = s.eng->counters_reset(3))) return rc;
        return s.eng->yy_filters(s.samples, s.centroids, s.drifts, s.gdrifts, s.assignments, s.prev, s.bounds, s.passed);
      }));
      stats.iterations++;
    }
  }

  int gather_outputs(float *centroids, uint32_t *assignments, int32_t device_ptrs) {
    RETERR(sync_all());
    Shard &first = *shards[0];
    (void)hipSetDevice(first.dev);
    const void *cen_src = first.centroids;
    size_t cen_bytes = (size_t)K * D * sizeof(float);
    if (fp16) {  // K x D halves out
      uint16_t *hbuf = nullptr;
      int rc = first.alloc(&hbuf, (size_t)K * D);
      if (rc) return rc;
      if (launch_float_to_half(first.centroids, (size_t)K * D, hbuf, first.eng->stream_) != hipSuccess ||
          hipStreamSynchronize(first.eng->stream_) != hipSuccess)
        return kmcudaRuntimeError;
      cen_src = hbuf;
      cen_bytes = (size_t)K * D * sizeof(uint16_t);
    }
    if (device_ptrs < 0) {
      if (hipMemcpy(centroids, cen_src, cen_bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return kmcudaMemoryCopyError;
      for (auto &s : shards) {
        (void)hipSetDevice(s->dev);
        if (hipMemcpy(assignments + s->offset, s->assignments, (size_t)s->length * sizeof(uint32_t),
                      hipMemcpyDeviceToHost) != hipSuccess)
          return kmcudaMemoryCopyError;
      }
    } else {
      if (hipMemcpyPeer(centroids, device_ptrs, cen_src, first.dev, cen_bytes) != hipSuccess)
        return kmcudaMemoryCopyError;
      for (auto &s : shards)
        if (hipMemcpyPeer(assignments + s->offset, device_ptrs, s->assignments, s->dev,
                          (size_t)s->length * sizeof(uint32_t)) != hipSuccess)
          return kmcudaMemoryCopyError;
    }
    return 0;
  }

  // reference: kmeans_cuda_calc_average_distance, kmeans.cu:1265-1300
  int average_distance(float *out) {
    INFO("calculating the average distance...\n");
    std::vector<float> host(N);
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      if ((strict_h2 ? launch_h2_member(metric, s->samples, s->length, D, s->centroids, s->assignments, K, s->dists,
                                        s->eng->stream_)
                     : launch_member_distances(metric, s->samples, s->length, D, s->centroids, s->assignments, K, s->dists,
                                               s->eng->stream_)) != hipSuccess)
        return kmcudaRuntimeError;
      if (hipMemcpyAsync(host.data() + s->offset, s->dists, (size_t)s->length * sizeof(float), hipMemcpyDeviceToHost,
                         s->eng->stream_) != hipSuccess)
        return kmcudaMemoryCopyError;
    }
    RETERR(sync_all());
    *out = (float)(butterfly_sum(host.data(), N) / N);
    return 0;
  }
};


// ---------------------------------------------------------------------------------------
// k-NN host side (reference: knn_cuda, kmcuda.cc:572-730 + knn_cuda_calc, knn.cu:381-532).
// Every GPU of the mask holds the whole corpus (as in the reference, kmcuda.cc:593-598) in
// CLUSTER-SORTED order and searches a contiguous slice of the sorted positions; there is no
// data-path collective.  The small replicated pieces (radii, K x K centroid distances) are
// recomputed on every GPU instead of exchanged.
// ---------------------------------------------------------------------------------------
struct KnnShard {
  int dev = 0;
  hipStream_t stream = nullptr;
  const float *samples = nullptr, *centroids = nullptr;
  const uint32_t *assignments = nullptr;
  float *xs = nullptr, *n2s = nullptr, *mydist = nullptr, *rdist = nullptr, *R = nullptr, *C = nullptr, *heaps = nullptr;
  float *mu = nullptr, *mux = nullptr, *kbias = nullptr;
  uint16_t *xs16 = nullptr;
  uint32_t *inv = nullptr, *offsets = nullptr, *keys_tmp = nullptr, *vals_tmp = nullptr, *keys_sorted = nullptr,
           *stats = nullptr, *blocks = nullptr, *out = nullptr;
  unsigned long long *calced = nullptr;
  void *sort_temp = nullptr;
  uint32_t first_block = 0, nblocks = 0, p_base = 0, p_end = 0;
  std::vector<void *> owned;
  ~KnnShard() {
    (void)hipSetDevice(dev);
    for (void *p : owned) (void)hipFree(p);
    pooled_stream_release(dev, stream);
  }
  template <typename T>
  int alloc(T **p, size_t count) {
    void *q = nullptr;
    if (hipMalloc(&q, count ? count * sizeof(T) : sizeof(T)) != hipSuccess) return kmcudaMemoryAllocationFailure;
    owned.push_back(q);
    *p = static_cast<T *>(q);
    return 0;
  }
  // brings `count` elements of a caller buffer onto this device (or uses it in place)
  template <typename T>
  int stage_in(const T *src, size_t count, int32_t device_ptrs, const T **dst) {
    if (device_ptrs >= 0 && device_ptrs == dev) {
      *dst = src;
      return 0;
    }
    T *buf = nullptr;
    int rc = alloc(&buf, count);
    if (rc) return rc;
    hipError_t e = device_ptrs < 0
                       ? hipMemcpyAsync(buf, src, count * sizeof(T), hipMemcpyHostToDevice, stream)
                       : hipMemcpyPeerAsync(buf, dev, src, device_ptrs, count * sizeof(T), stream);
    if (e != hipSuccess) return kmcudaMemoryCopyError;
    *dst = buf;
    return 0;
  }
};

// brings `count` halves of a caller buffer onto the shard's device and widens them to fp32
static int stage_in_half(KnnShard &sh, const void *src, size_t count, int32_t device_ptrs, const float **dst) {
  float *buf = nullptr;
  int rc = sh.alloc(&buf, count);
  if (rc) return rc;
  const uint16_t *dev_half = reinterpret_cast<const uint16_t *>(src);
  uint16_t *tmp = nullptr;
  if (!(device_ptrs >= 0 && device_ptrs == sh.dev)) {
    if ((rc = sh.alloc(&tmp, count))) return rc;
    hipError_t e = device_ptrs < 0 ? hipMemcpyAsync(tmp, src, count * sizeof(uint16_t), hipMemcpyHostToDevice, sh.stream)
                                   : hipMemcpyPeerAsync(tmp, sh.dev, src, device_ptrs, count * sizeof(uint16_t), sh.stream);
    if (e != hipSuccess) return kmcudaMemoryCopyError;
    dev_half = tmp;
  }
  if (launch_half_to_float(dev_half, count, buf, sh.stream) != hipSuccess) return kmcudaRuntimeError;
  *dst = buf;
  return 0;
}

class KnnJob {
 public:
  std::vector<std::unique_ptr<KnnShard>> shards;

  int run(const std::vector<int> &devs, int nvirtual, uint32_t k, int metric, uint32_t N, uint32_t D, uint32_t K,
          int32_t device_ptrs, int verbosity, bool fp16, const float *samples, const float *centroids,
          const uint32_t *assignments, uint32_t *neighbors) {
    std::vector<int> shard_devs = devs;
    if (nvirtual > 1 && devs.size() == 1) shard_devs.assign(nvirtual, devs[0]);  // test hook
    // measurement hook KMCUDA_AMD_KNN_SHARD="i/n": plan the queries for n GPUs but run ONLY share i, on the
    // first GPU of the mask (what one rank of an n-GPU search does: whole corpus resident, 1/n of the
    // queries); the other rows of `neighbors` are left untouched
    size_t plan_shards = shard_devs.size(), plan_first = 0;
    if (const char *only = getenv("KMCUDA_AMD_KNN_SHARD")) {
      unsigned i = 0, n = 0;
      if (sscanf(only, "%u/%u", &i, &n) == 2 && n >= 1 && i < n) {
        plan_shards = n;
        plan_first = i;
        shard_devs.assign(1, devs[0]);
      }
    }
    const char *force_exact = getenv("KMCUDA_AMD_KNN_EXACT");
    // KMCUDA_AMD_FP16_STRICT (fp16x2 only): radii, centroid distances and every candidate distance in the reference's
    // half2 arithmetic (knn.hip, half2_ops.hpp) -- the verification mode of half2_strict.hip for this entry point;
    // no matrix-core filter (its bound is stated against the fp32 arithmetic)
    const char *strict_env = getenv("KMCUDA_AMD_FP16_STRICT");
    const bool strict_h2 = fp16 && strict_env && atoi(strict_env) != 0;
    if (strict_h2) INFO("k-NN: the reference's half2 arithmetic (KMCUDA_AMD_FP16_STRICT)\n");
    const char *fenv = getenv("KMCUDA_AMD_FILTER");
    const bool want_f32 = fenv && strcmp(fenv, "f32") == 0;
    uint32_t dp_filter = ((force_exact && atoi(force_exact)) || strict_h2) ? 0 : filter_dp_for(D);
    // 256 < D <= 1024: the f16 filter's one-operand-set instantiations (knn_f16.hip: 512 with two blocks per CU;
    // 768 / 1024 with one -- the queries' operands alone are 192 / 256 registers; the f32 filter stops at 256)
    if (!dp_filter && !(force_exact && atoi(force_exact)) && !strict_h2 && !want_f32 && D > 256 && D <= 1024)
      dp_filter = D <= 512 ? 512u : (D <= 768 ? 768u : 1024u);
    const uint32_t DP = dp_filter ? dp_filter : D;
    if (!dp_filter) INFO("k-NN: every candidate is evaluated with the exact arithmetic (no matrix-core filter)\n");
    // which matrix-core instruction filters the candidates: f16 on centred hi/lo-split rows (default,
    // needs DP >= 16) or f32 (KMCUDA_AMD_FILTER=f32)
    const bool use_f16 = dp_filter >= 16 && !want_f32;
    // mu = mean of the finite centroid rows (any vector works: distances are translation invariant)
    std::vector<float> mu_host(DP, 0.f);
    float mu2 = 0.f;
    if (use_f16) {
      std::vector<float> cen((size_t)K * D);
      if (fp16) {
        std::vector<uint16_t> raw((size_t)K * D);
        if (device_ptrs < 0) memcpy(raw.data(), centroids, raw.size() * sizeof(uint16_t));
        else if (hipMemcpy(raw.data(), centroids, raw.size() * sizeof(uint16_t), hipMemcpyDeviceToHost) != hipSuccess)
          return kmcudaMemoryCopyError;
        for (size_t i = 0; i < raw.size(); i++) {  // half -> float on the host
          const uint32_t hbits = raw[i], sign = (hbits & 0x8000u) << 16, ex = (hbits >> 10) & 0x1Fu, man = hbits & 0x3FFu;
          float v;
          if (ex == 0) v = ldexpf((float)man, -24);
          else if (ex == 31) v = man ? NAN : INFINITY;
          else v = ldexpf((float)(man | 0x400u), (int)ex - 25);
          cen[i] = sign ? -v : v;
        }
      } else if (device_ptrs < 0) {
        memcpy(cen.data(), centroids, cen.size() * sizeof(float));
      } else if (hipMemcpy(cen.data(), centroids, cen.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        return kmcudaMemoryCopyError;
      }
      std::vector<double> acc(D, 0.0);
      uint32_t nfin = 0;
      for (uint32_t c = 0; c < K; c++) {
        bool fin = true;
        for (uint32_t f = 0; f < D && fin; f++) fin = std::isfinite(cen[(size_t)c * D + f]);
        if (!fin) continue;
        for (uint32_t f = 0; f < D; f++) acc[f] += cen[(size_t)c * D + f];
        nfin++;
      }
      for (uint32_t f = 0; f < D; f++) {
        mu_host[f] = nfin ? (float)(acc[f] / nfin) : 0.f;
        mu2 += mu_host[f] * mu_host[f];
      }
      mu2 *= 1.0001f;
    }
    const size_t sort_bytes = sort_temp_bytes(N, K);
    for (int dev : shard_devs) {
      auto sh = std::make_unique<KnnShard>();
      sh->dev = dev;
      if (hipSetDevice(dev) != hipSuccess) return kmcudaNoSuchDevice;
      if (!(sh->stream = pooled_stream_acquire(dev))) return kmcudaRuntimeError;
      if (fp16) {  // half buffers -> fp32 working copies (fp32 arithmetic on the half values)
        RETERR(stage_in_half(*sh, samples, (size_t)N * D, device_ptrs, &sh->samples));
        RETERR(stage_in_half(*sh, centroids, (size_t)K * D, device_ptrs, &sh->centroids));
      } else {
        RETERR(sh->stage_in(samples, (size_t)N * D, device_ptrs, &sh->samples));
        RETERR(sh->stage_in(centroids, (size_t)K * D, device_ptrs, &sh->centroids));
      }
      RETERR(sh->stage_in(assignments, (size_t)N, device_ptrs, &sh->assignments));
      int rc;
      if ((rc = sh->alloc(&sh->xs, (size_t)N * DP))) return rc;
      if ((rc = sh->alloc(&sh->n2s, N))) return rc;
      if ((rc = sh->alloc(&sh->mydist, N))) return rc;
      if ((rc = sh->alloc(&sh->rdist, N))) return rc;
      if ((rc = sh->alloc(&sh->R, K))) return rc;
      if ((rc = sh->alloc(&sh->C, (size_t)K * K))) return rc;
      if ((rc = sh->alloc(&sh->inv, N))) return rc;
      if ((rc = sh->alloc(&sh->offsets, (size_t)K + 2))) return rc;
      if ((rc = sh->alloc(&sh->keys_tmp, N))) return rc;
      if ((rc = sh->alloc(&sh->vals_tmp, N))) return rc;
      if ((rc = sh->alloc(&sh->keys_sorted, N))) return rc;
      if ((rc = sh->alloc(&sh->stats, 4))) return rc;
      if ((rc = sh->alloc(&sh->calced, KNN_STATS))) return rc;
      if (use_f16) {
        if ((rc = sh->alloc(&sh->xs16, ((size_t)N + KNN16_PAD_ROWS) * DP))) return rc;
        if ((rc = sh->alloc(&sh->kbias, (size_t)N + KNN16_PAD_ROWS))) return rc;
        if ((rc = sh->alloc(&sh->mu, DP))) return rc;
        if ((rc = sh->alloc(&sh->mux, N))) return rc;
        if (hipMemcpyAsync(sh->mu, mu_host.data(), DP * sizeof(float), hipMemcpyHostToDevice, sh->stream) != hipSuccess)
          return kmcudaMemoryCopyError;
      }
      char *t = nullptr;
      if ((rc = sh->alloc(&t, sort_bytes + 16))) return rc;
      sh->sort_temp = t;
      shards.push_back(std::move(sh));
    }
    // ---- per GPU: inverse assignments, sorted copy, radii, centroid distances ----
    INFO("initializing the inverse assignments...\n");
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      if (hipMemsetAsync(s->calced, 0, KNN_STATS * sizeof(unsigned long long), s->stream) != hipSuccess) return kmcudaRuntimeError;
      if (launch_inverse_assignments(s->assignments, N, K, s->keys_tmp, s->vals_tmp, s->keys_sorted, s->inv,
                                     s->offsets, s->sort_temp, sort_bytes, s->stream) != hipSuccess)
        return kmcudaRuntimeError;
      if (launch_knn_gather(s->samples, N, D, DP, s->inv, s->xs, s->n2s, s->stats, s->stream) != hipSuccess)
        return kmcudaRuntimeError;
    }
    INFO("calculating the cluster radiuses...\n");
    INFO("calculating the centroid distance matrix...\n");
    for (auto &s : shards) {
      (void)hipSetDevice(s->dev);
      if (launch_knn_prep(metric, s->xs, N, D, DP, s->offsets, K, s->centroids, s->mydist, s->rdist, s->R, s->C,
                          strict_h2, s->stream) != hipSuccess)
        return kmcudaRuntimeError;
    }
    // ---- the block list: KNN_QPB_* consecutive sorted positions of one cluster per block ----
    std::vector<uint32_t> offsets(K + 1);
    {
      KnnShard &f = *shards[0];
      (void)hipSetDevice(f.dev);
      if (hipStreamSynchronize(f.stream) != hipSuccess) return kmcudaRuntimeError;
      if (hipMemcpy(offsets.data(), f.offsets, (K + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return kmcudaMemoryCopyError;
    }
    std::vector<uint32_t> blocks;  // (cluster, first position) pairs
    const uint32_t qpb = use_f16 ? knn_qpb_f16(DP) : KNN_QPB_F32;
    for (uint32_t c = 0; c < K; c++)
      for (uint32_t p = offsets[c]; p < offsets[c + 1]; p += qpb) {
        blocks.push_back(c);
        blocks.push_back(p);
      }
    const uint32_t total_blocks = (uint32_t)(blocks.size() / 2);
    const uint32_t assigned = offsets[K];  // positions >= assigned belong to no cluster (NaN samples)
    for (size_t si = 0; si < shards.size(); si++) {
      KnnShard &s = *shards[si];
      const size_t i = plan_first + si;
      s.first_block = (uint32_t)((uint64_t)total_blocks * i / plan_shards);
      const uint32_t next = (uint32_t)((uint64_t)total_blocks * (i + 1) / plan_shards);
      s.nblocks = next - s.first_block;
      s.p_base = s.nblocks ? blocks[2 * (size_t)s.first_block + 1] : assigned;
      s.p_end = next < total_blocks ? blocks[2 * (size_t)next + 1] : assigned;
      if (!s.nblocks) s.p_end = s.p_base;
      if (i + 1 == plan_shards && !dp_filter) s.p_end = N;  // the exact kernel also fills the unassigned rows
    }
    if (use_f16) {  // after the radii / member distances, which read the plain norms' buffer no more
      for (auto &s : shards) {
        (void)hipSetDevice(s->dev);
        if (launch_knn_split(metric, s->xs, N, D, DP, s->mu, s->xs16, s->n2s, s->mux, s->kbias, s->stats, s->stream) != hipSuccess)
          return kmcudaRuntimeError;
      }
    }
    INFO("searching for the nearest neighbors...\n");
    for (auto &sp : shards) {
      KnnShard &s = *sp;
      (void)hipSetDevice(s.dev);
      const uint32_t len = s.p_end - s.p_base;
      int rc;
      if ((rc = s.alloc(&s.heaps, (size_t)len * 2 * k))) return rc;
      if ((rc = s.alloc(&s.out, (size_t)len * k))) return rc;
      // KMCUDA_AMD_KNN_XCD=1 (an experiment of round 5, off by default): the blocks of one query cluster -- which visit
      // the same candidate clusters in the same order, 512 bytes per candidate and block: 4.3 TB of fetches for config
      // D's share -- dispatched to ONE XCD (workgroup indices congruent mod 8), cluster after cluster, so that an XCD's
      // L2 holds two or three such streams instead of sixteen and a stream's followers hit the tiles its leader has
      // just fetched.  Measured: FETCH_SIZE 4.17 TB against 4.27, knn_cuda 1.13-1.15 s against 1.11-1.13
      // (profiles/r5d_knn_dispatch_order_ab.log): the blocks of a stream drift further apart than an L2 holds (a
      // cluster's slab alone is 4 MB).  Slots a shorter list leaves empty carry the marker 0xFFFFFFFF (the kernel
      // returns at once).  Only the order of independent blocks changes: the lists are the same.
      std::vector<uint32_t> plan(blocks.begin() + 2 * (size_t)s.first_block,
                                 blocks.begin() + 2 * (size_t)(s.first_block + s.nblocks));
      uint32_t launch_blocks = s.nblocks;
      {
        const char *xe = getenv("KMCUDA_AMD_KNN_XCD");
        if (use_f16 && s.nblocks >= 64 && xe && atoi(xe) != 0) {
          constexpr uint32_t kXcds = 8;
          std::vector<std::vector<uint32_t>> lists(kXcds);   // block numbers (into plan) per XCD
          uint32_t b = 0;
          while (b < s.nblocks) {
            uint32_t e = b;
            while (e < s.nblocks && plan[2 * (size_t)e] == plan[2 * (size_t)b]) e++;   // one cluster's blocks
            uint32_t best = 0;
            for (uint32_t x = 1; x < kXcds; x++)
              if (lists[x].size() < lists[best].size()) best = x;
            for (uint32_t q = b; q < e; q++) lists[best].push_back(q);
            b = e;
          }
          size_t longest = 0;
          for (auto &l : lists) longest = l.size() > longest ? l.size() : longest;
          std::vector<uint32_t> ordered(2 * kXcds * longest, 0xFFFFFFFFu);
          for (uint32_t x = 0; x < kXcds; x++)
            for (size_t i = 0; i < lists[x].size(); i++) {
              ordered[2 * (kXcds * i + x)] = plan[2 * (size_t)lists[x][i]];
              ordered[2 * (kXcds * i + x) + 1] = plan[2 * (size_t)lists[x][i] + 1];
            }
          plan.swap(ordered);
          launch_blocks = (uint32_t)(kXcds * longest);
        }
      }
      if ((rc = s.alloc(&s.blocks, plan.size() ? plan.size() : 2))) return rc;
      if (!plan.empty() &&
          hipMemcpy(s.blocks, plan.data(), plan.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return kmcudaMemoryCopyError;
      KnnArgs a;
      a.xs = s.xs; a.n2s = s.n2s; a.inv = s.inv; a.offsets = s.offsets; a.mydist = s.mydist; a.R = s.R; a.C = s.C;
      a.blocks = s.blocks; a.stats = s.stats; a.N = N; a.D = D; a.DP = DP; a.K = K; a.k = k;
      a.p_base = s.p_base; a.p_end = s.p_end;
      a.eps = (float)(1.02 * ((double)D + 12.0) * ldexp(1.0, -24));  // as the Lloyd filter
      a.heaps = s.heaps; a.out = s.out; a.calced = s.calced;
      a.xs16 = s.xs16; a.mux = s.mux; a.kbias = s.kbias; a.mu2 = mu2;
      // The tighter cluster test of the f16 search (knn_f16.hip: the query's own distance to every centroid instead of
      // the triangle bound for it).  4 K bytes per query; without that memory, or with KMCUDA_AMD_KNN_TIGHT=0, the
      // reference's prune test decides alone.  Same neighbour lists either way.
      if (use_f16 && metric == 0 && D <= 1024 && len != 0) {
        const char *tight = getenv("KMCUDA_AMD_KNN_TIGHT");
        if (!(tight && atoi(tight) == 0)) {
          float *lb = nullptr;
          if (s.alloc(&lb, (size_t)K * len) == 0) {
            if (launch_knn_centroid_bounds(s.xs, D, DP, s.p_base, s.p_end, s.centroids, K, s.R, lb, len, s.stream) !=
                hipSuccess)
              return kmcudaRuntimeError;
            a.lb = lb;
            a.lb_stride = len;
            // queries that want the same clusters into the same waves (update.hip: launch_knn_query_order): by the
            // other cluster that can come closest, then by their distance to their own centroid (mode 3);
            // KMCUDA_AMD_KNN_ORDER=0: sorted-position order, 1: the closest other cluster alone (rounds 4-5), 2: the
            // distance alone (A/B: 1.911 / 1.882 / 1.860 / 1.839e12 pairs scored for config D's share, profiles/r6aj_*)
            const char *ord = getenv("KMCUDA_AMD_KNN_ORDER");
            uint32_t *qperm = nullptr;
            const int ord_mode = ord ? atoi(ord) : 3;
            if (ord_mode != 0 && s.alloc(&qperm, len) == 0) {
              if (launch_knn_query_order(lb, len, s.offsets, K, s.p_base, s.p_end, s.keys_tmp, s.vals_tmp, s.keys_sorted,
                                         qperm, s.sort_temp, sort_bytes, s.stream, ord_mode, s.mydist, s.R))
                a.qperm = qperm;
              else
                (void)hipGetLastError();
            }
          } else {
            (void)hipGetLastError();
            DEBUG("k-NN: no memory for the per-query centroid bounds, the reference's prune test alone\n");
          }
        }
      }
      const hipError_t e = !dp_filter ? launch_knn_exact(metric, a, strict_h2, s.stream)
                           : use_f16 ? launch_knn_filter_f16(metric, a, launch_blocks, s.stream)
                                     : launch_knn_filter(metric, a, s.nblocks, s.stream);
      if (e != hipSuccess) return kmcudaRuntimeError;
    }
    // ---- outputs: rows back in sample order ----
    if (device_ptrs >= 0 && dp_filter && assigned < N) {
      // rows without a cluster (NaN samples) get no neighbours: 0xFFFFFFFF, as the host branch writes
      // (the scatter below only covers the assigned positions)
      (void)hipSetDevice(device_ptrs);
      if (hipMemset(neighbors, 0xFF, (size_t)N * k * sizeof(uint32_t)) != hipSuccess) return kmcudaRuntimeError;
    }
    unsigned long long dists_calced = 0;
    std::vector<uint32_t> host_inv, host_out;
    if (device_ptrs < 0) {
      host_inv.resize(N);
      KnnShard &f = *shards[0];
      (void)hipSetDevice(f.dev);
      if (hipMemcpy(host_inv.data(), f.inv, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return kmcudaMemoryCopyError;
    }
    for (auto &sp : shards) {
      KnnShard &s = *sp;
      (void)hipSetDevice(s.dev);
      if (hipStreamSynchronize(s.stream) != hipSuccess) {
        INFO("k-NN kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
        return kmcudaRuntimeError;
      }
      unsigned long long cs[KNN_STATS] = {0, 0, 0, 0, 0};
      if (hipMemcpy(cs, s.calced, sizeof(cs), hipMemcpyDeviceToHost) != hipSuccess) return kmcudaMemoryCopyError;
      const unsigned long long c = cs[0];
      DEBUG("#%d dists_calced: %llu\n", s.dev, c);
      // what the f16 search actually did (knn_f16.hip; measurement: KMCUDA_AMD_KNN_STATS=1 prints it at any verbosity)
      if (cs[1] && (verbosity > 1 || getenv("KMCUDA_AMD_KNN_STATS")))
        printf("#%d k-NN filter: %llu pairs by the reference's prune rule, %llu scored on the matrix cores "
               "(%llu of them live query x real candidate; %llu if an operand set nobody visits with were skipped), "
               "%llu exact chains\n", s.dev, cs[0], cs[1], cs[2], cs[4], cs[3]);
      dists_calced += c;
      const uint32_t len = s.p_end - s.p_base;
      if (!len) continue;
      if (device_ptrs < 0) {
        host_out.resize((size_t)len * k);
        if (hipMemcpy(host_out.data(), s.out, (size_t)len * k * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
          return kmcudaMemoryCopyError;
        for (uint32_t i = 0; i < len; i++)
          memcpy(neighbors + (size_t)host_inv[s.p_base + i] * k, host_out.data() + (size_t)i * k, k * sizeof(uint32_t));
      } else {
        // scatter on the caller's device
        const uint32_t *src_out = s.out, *src_inv = s.inv;
        uint32_t *tmp_out = nullptr, *tmp_inv = nullptr;
        struct TmpFree { uint32_t **a, **b; ~TmpFree() { if (*a) (void)hipFree(*a); if (*b) (void)hipFree(*b); } }
            tmp_guard{&tmp_out, &tmp_inv};   // also on the early returns below
        if (s.dev != device_ptrs) {
          (void)hipSetDevice(device_ptrs);
          if (hipMalloc((void **)&tmp_out, (size_t)len * k * sizeof(uint32_t)) != hipSuccess ||
              hipMalloc((void **)&tmp_inv, (size_t)N * sizeof(uint32_t)) != hipSuccess)
            return kmcudaMemoryAllocationFailure;
          if (hipMemcpyPeer(tmp_out, device_ptrs, s.out, s.dev, (size_t)len * k * sizeof(uint32_t)) != hipSuccess ||
              hipMemcpyPeer(tmp_inv, device_ptrs, s.inv, s.dev, (size_t)N * sizeof(uint32_t)) != hipSuccess)
            return kmcudaMemoryCopyError;
          src_out = tmp_out;
          src_inv = tmp_inv;
        }
        (void)hipSetDevice(device_ptrs);
        hipError_t e = launch_knn_scatter(src_out, src_inv, s.p_base, s.p_end, k, neighbors, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return kmcudaRuntimeError;
      }
    }
    if (dp_filter && assigned < N) {  // rows without a cluster: no neighbours (the reference reads out of bounds)
      if (device_ptrs < 0) {
        for (uint32_t p = assigned; p < N; p++)
          for (uint32_t i = 0; i < k; i++) neighbors[(size_t)host_inv[p] * k + i] = UINT32_MAX;
      }
    }
    INFO("calculated %f of all the distances\n", (dists_calced + .0) / ((double)N * N));  // knn.cu:529-530
    return 0;
  }
};

// The Yinyang phase of the default schedule.  The reference hands over from Lloyd to its bounds at 11 %
// reassignments whatever the data; its bounds refresh (kmeans_yy_init: >= G exact chains per row) costs 38 assignment
// passes on this hardware and its filters more than a pass per row that passes them (profiles/r2n_*, r3y_*): measured on
// both benchmark data sets it never pays.  The default keeps the reference's sequence up to and including the hand-over
// point (group clustering, its progress lines, its srand(0)) and then goes on with assignment passes that CARRY
// per-row bounds from pass to pass at no refresh cost (lloyd_carry.hip).  Every pass either way is the reference's
// arithmetic.  KMCUDA_AMD_YY=reference: the reference's own schedule and kernels.  KMCUDA_AMD_YY_SWITCH=<fraction>
// (tests): hand over to the reference's bounds once the reassigned fraction is at most that.
struct SwitchRule {
  double N, force = -1.0;
  explicit SwitchRule(uint32_t n) : N(n) {
    if (const char *v = getenv("KMCUDA_AMD_YY_SWITCH")) force = atof(v);
  }
  bool now(uint32_t changed) const { return force >= 0 && (double)changed <= force * N; }
};

// Once per process and device: the iteration kernels' code objects are loaded by a helper thread while the calling
// thread uploads the rows and seeds (they would otherwise load one by one in front of the first iteration: 24 ms of a
// 45-ms call on 4M rows).  Fire and forget: a launch that needs a code object the helper is still loading waits for it
// inside the runtime; the caller joins the helper before it returns (never a thread inside the runtime at process
// exit).  KMCUDA_AMD_PRELOAD=0: off (A/B).
struct PreloadThreads {
  std::vector<std::thread> threads;
  ~PreloadThreads() {
    for (auto &t : threads)
      if (t.joinable()) t.join();
  }
};
void preload_code_objects(int dev, PreloadThreads *keep) {
  static std::mutex m;
  static std::vector<bool> done;
  {
    std::lock_guard<std::mutex> lock(m);
    if (done.size() <= (size_t)dev) done.resize(dev + 1, false);
    if (done[dev]) return;
    done[dev] = true;
  }
  if (const char *v = getenv("KMCUDA_AMD_PRELOAD"))
    if (atoi(v) == 0) return;
  keep->threads.emplace_back([dev] {
    if (hipSetDevice(dev) != hipSuccess) return;
    (void)preload_update_code();
    (void)preload_lloyd_f16_code();
    (void)preload_lloyd_code();
    (void)preload_lloyd_carry_code();
    (void)preload_lloyd_duo_code();
    (void)hipGetLastError();
  });
}

int virtual_shards() {
  const char *v = getenv("KMCUDA_AMD_VIRTUAL_SHARDS");
  return v ? atoi(v) : 0;
}

}  // namespace

extern "C" {

KMCUDAResult kmeans_cuda(KMCUDAInitMethod init, const void *init_params, float tolerance, float yinyang_t,
                         KMCUDADistanceMetric metric, uint32_t samples_size, uint16_t features_size,
                         uint32_t clusters_size, uint32_t seed, uint32_t device, int32_t device_ptrs, int32_t fp16x2,
                         int32_t verbosity, const float *samples, float *centroids, uint32_t *assignments,
                         float *average_distance) {
  kmx::g_verbosity = verbosity;
  DEBUG("arguments: %d %p %.3f %.2f %d %u %u %u %u %u %d %d %p %p %p %p\n", init, init_params, tolerance, yinyang_t,
        metric, samples_size, (unsigned)features_size, clusters_size, seed, device, fp16x2, verbosity,
        (const void *)samples, (void *)centroids, (void *)assignments, (void *)average_distance);
  // reference: check_kmeans_args, kmcuda.cc:19-61 (same order)
  if (clusters_size < 2 || clusters_size == UINT32_MAX) return kmcudaInvalidArguments;
  if (features_size == 0) return kmcudaInvalidArguments;
  if (samples_size < clusters_size) return kmcudaInvalidArguments;
  int ndev = 0;
  (void)hipGetDeviceCount(&ndev);
  if (ndev < 32 && device > (1u << ndev)) return kmcudaNoSuchDevice;
  if (samples == nullptr || centroids == nullptr || assignments == nullptr) return kmcudaInvalidArguments;
  if (tolerance < 0 || tolerance > 1) return kmcudaInvalidArguments;
  if (yinyang_t < 0 || yinyang_t > 0.5) return kmcudaInvalidArguments;
  INFO("reassignments threshold: %u\n", uint32_t(tolerance * samples_size));
  const uint32_t yy_groups_size = yinyang_t * clusters_size;  // float product, truncated (kmcuda.cc:417)
  DEBUG("yinyang groups: %u\n", yy_groups_size);
  auto devs = setup_devices(device, verbosity);
  if (devs.empty()) return kmcudaNoSuchDevice;

  // device-pointer inputs: whatever the caller still has in flight on that GPU (any stream) must have
  // landed before our own non-blocking streams read it -- the reference got this from the legacy
  // default stream's implicit synchronisation
  if (device_ptrs >= 0 && hipSetDevice(device_ptrs) == hipSuccess) (void)hipDeviceSynchronize();
  PreloadThreads preload;   // (joined when this call returns, whichever way)
  for (int d : devs) preload_code_objects(d, &preload);
  const auto t_begin = std::chrono::steady_clock::now();
  // KMCUDA_AMD_TIMING=1: wall-clock laps of the call's phases on stderr (a measurement aid; each lap waits for the GPUs)
  const bool timing = getenv("KMCUDA_AMD_TIMING") != nullptr;
  auto t_lap = t_begin;
  Job job;
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)job.sync_all();
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[timing] %s: %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_lap).count());
    t_lap = t1;
  };
  job.fp16 = fp16x2 != 0;
  if (const char *v = getenv("KMCUDA_AMD_FP16_STRICT")) job.strict_h2 = job.fp16 && atoi(v) != 0;
  // fp16x2: features_size counts half2 pairs (kmcuda.h:107-108); internally one feature per half
  const uint32_t feats = fp16x2 ? 2u * features_size : features_size;
  RETERR(job.setup(devs, virtual_shards(), samples_size, feats, clusters_size, metric, verbosity, samples,
                   device_ptrs));
  if (const char *v = getenv("KMCUDA_AMD_EXACT_UPDATE")) job.exact_update = atoi(v) != 0;
  if (job.strict_h2) job.exact_update = true;   // the reference's serial update, in half2 arithmetic
  lap("set-up (engines, uploads)");
  if (job.exact_update && job.shards.size() > 1) {
    INFO("KMCUDA_AMD_EXACT_UPDATE needs all rows on one GPU (the reference's update order is global)\n");
    return kmcudaInvalidArguments;
  }
  const uint32_t afk_m = (init == kmcudaInitMethodAFKMC2 && init_params) ? *reinterpret_cast<const uint32_t *>(init_params) : 0;
  RETERR(job.init_centroids(init, seed, centroids, device_ptrs, afk_m));
  RETERR(job.sync_all());
  lap("seeding");
  const auto t_loop = std::chrono::steady_clock::now();
  job.stats.setup_seconds = std::chrono::duration<double>(t_loop - t_begin).count();
  job.stats.shards = (uint32_t)job.shards.size();
  job.stats.rccl = job.comms.empty() ? 0u : (uint32_t)job.comms.size();

  if (yy_groups_size == 0 || kYinyangDraftReassignments <= tolerance) {  // kmeans.cu:1037-1050
    if (yy_groups_size == 0) INFO("too few clusters for this yinyang_t => Lloyd\n");
    else INFO("tolerance is too high (>= %.2f) => Lloyd\n", kYinyangDraftReassignments);
    RETERR(job.lloyd(tolerance, nullptr));
  } else {
    // KMCUDA_AMD_YY=reference: the reference's fixed schedule (Lloyd down to 11 % reassignments, then its bounds).
    // Default: the same hand-over point, then assignment passes that carry per-sample bounds (SwitchRule above).
    // The strict parity modes keep the reference's schedule.
    const char *yym = getenv("KMCUDA_AMD_YY");
    const bool wide = job.shards[0]->eng->DP_ == 0 && job.shards[0]->eng->wide_dp_ != 0;   // D > 256: lloyd_wide.hip
    // (KMCUDA_AMD_YY=carry: the default schedule also under the strict update -- every pass is the reference's Lloyd
    //  arithmetic, so the whole call then equals the reference's kmeans_cuda_lloyd bit for bit: the parity test of the
    //  carried bounds against the oracle end to end, tests/test_gpu_carry.py)
    const bool carry_asked = yym && strcmp(yym, "carry") == 0;
    const bool adaptive = !(yym && strcmp(yym, "reference") == 0) && (!job.exact_update || (carry_asked && !job.strict_h2)) &&
                          ((job.shards[0]->eng->DP_ != 0 && job.shards[0]->eng->filter_mode_ == 0) || wide);
    INFO("running Lloyd until reassignments drop below %u\n", (uint32_t)(kYinyangDraftReassignments * samples_size));
    const SwitchRule rule(samples_size);
    int iter = 0;
    RETERR(job.lloyd((float)kYinyangDraftReassignments, &iter));
    lap("Lloyd down to 11 % reassignments");
    const int st = job.check_changed(iter, tolerance, false);  // kmeans.cu:1058
    if (st < 0) return static_cast<KMCUDAResult>(-st);
    if (st == 0) {
      // the reference's hand-over point: the K centroids are clustered into groups here (its progress lines are
      // part of what a caller sees, its srand(0) of what a caller's rand() sees) -- whichever schedule follows
      std::vector<uint32_t> groups;
      // The groups are only ever USED by the reference's bounds (KMCUDA_AMD_YY=reference / the forced switch).  What a
      // caller can observe of their clustering besides -- its progress lines, and the C library's rand() state after
      // its srand(0) and draws -- is kept either way: a silent call (verbosity 0) that will not use the groups only
      // replays the draws (13 ms -> 0.05 ms at K = 1024, G = 102).
      if (!adaptive || rule.force >= 0 || verbosity > 0) {
        RETERR(job.cluster_groups(yy_groups_size, &groups));
      } else {
        RETERR(job.replay_group_seeding_draws(yy_groups_size));
      }
      lap("group clustering, whole");
      bool bounds = !adaptive;
      if (adaptive) {
        const char *cv = getenv("KMCUDA_AMD_CARRY");
        // (rows wider than 256 features: the streamed filter carries the bounds itself, up to 4096 features)
        const bool carry = !(cv && atoi(cv) == 0) && (!wide || job.shards[0]->eng->wide_dp_ <= 4096u);
        // The first carried pass only LEAVES bounds (and allocates them: 24 bytes per row); rows are spared from the
        // second one on.  A run that stops within a pass or two of the hand-over point never earns that back (round 4:
        // the 4M-row mixture at tolerance 0.01, one iteration after the hand-over, +12-16 %), and how long a run will
        // last cannot be read off its first counts (they fall by a factor of 4-8 per iteration, the tail by 1.1: an
        // extrapolation tried in round 5 switched the bounds off for the 37-iteration run,
        // profiles/r5e_configs_carry_extrapolation_rule_misfires.log).  So the bounds start once the host has SEEN the
        // run go on past an iteration after the hand-over (Job::lloyd, carry_after).  KMCUDA_AMD_CARRY=1: at once
        // (tests: every pass after the hand-over is then a carried one).
        const bool at_once = carry && cv && atoi(cv) != 0;
        INFO(at_once ? "Lloyd goes on, carrying per-sample distance bounds from pass to pass\n" : "Lloyd goes on\n");
        for (auto &s : job.shards) {
          s->eng->carry_on_ = at_once;
          s->eng->carry_valid_ = false;
        }
        job.carry_after = (carry && !at_once) ? iter + 1 : 0;
        const Job::LeaveFn leave = [&rule](int, uint32_t changed) { return rule.now(changed); };
        RETERR(job.lloyd(tolerance, &iter, rule.force >= 0 ? &leave : nullptr, &bounds, iter));
        if (carry && verbosity > 1) {
          unsigned long long spared = 0, paired = 0;
          for (auto &s : job.shards) {
            unsigned long long v = 0;
            if (s->eng->carry_stats(&v, nullptr) == 0) spared += v;
            if (s->eng->carry_pair_stats(&v) == 0) paired += v;
          }
          printf("carried bounds: %llu sample passes decided without looking at the sample\n", spared);
          if (paired) printf("carried pairs: %llu sample passes settled between two carried contenders\n", paired);
        }
        for (auto &s : job.shards) s->eng->carry_on_ = false;
        job.carry_after = 0;
        lap("Lloyd after the hand-over point");
      }
      if (bounds) RETERR(job.yinyang(tolerance, yy_groups_size, iter, groups));
      if (bounds) lap("Yinyang");
    }
  }
  RETERR(job.sync_all());
  job.stats.loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
  job.collective_collect();
  if (getenv("KMCUDA_AMD_UPDATE_TRACE"))   // (a measurement aid: which way the updates went; nothing is printed inside the loop)
    for (auto &s : job.shards)
      fprintf(stderr, "[update] shard on device %d: %u radix, %u direct; loop %.4f s\n", s->dev, s->eng->ms_.n_radix, s->eng->ms_.n_direct,
              job.stats.loop_seconds);
  {
    std::lock_guard<std::mutex> lock(g_last_run_mutex);
    g_last_run = job.stats;
  }
  lap("loop end");
  if (average_distance) RETERR(job.average_distance(average_distance));
  RETERR(job.gather_outputs(centroids, assignments, device_ptrs));
  lap("outputs");
  DEBUG("return kmcudaSuccess\n");
  return kmcudaSuccess;
}

KMCUDAResult knn_cuda(uint16_t k, KMCUDADistanceMetric metric, uint32_t samples_size, uint16_t features_size,
                      uint32_t clusters_size, uint32_t device, int32_t device_ptrs, int32_t fp16x2, int32_t verbosity,
                      const float *samples, const float *centroids, const uint32_t *assignments, uint32_t *neighbors) {
  kmx::g_verbosity = verbosity;
  DEBUG("arguments: %u %d %u %u %u %u %d %d %d %p %p %p %p\n", (unsigned)k, metric, samples_size,
        (unsigned)features_size, clusters_size, device, device_ptrs, fp16x2, verbosity, (const void *)samples,
        (const void *)centroids, (const void *)assignments, (void *)neighbors);
  // reference: check_knn_args, kmcuda.cc:537-570 (its result is ignored there, :583-584; we report it)
  if (k == 0) return kmcudaInvalidArguments;
  if (clusters_size < 2 || clusters_size == UINT32_MAX) return kmcudaInvalidArguments;
  if (features_size == 0) return kmcudaInvalidArguments;
  if (samples_size < clusters_size) return kmcudaInvalidArguments;
  int ndev = 0;
  (void)hipGetDeviceCount(&ndev);
  if (ndev < 32 && device > (1u << ndev)) return kmcudaNoSuchDevice;
  if (!samples || !centroids || !assignments || !neighbors) return kmcudaInvalidArguments;
  auto devs = setup_devices(device, verbosity);
  if (devs.empty()) return kmcudaNoSuchDevice;
  if (device_ptrs >= 0 && hipSetDevice(device_ptrs) == hipSuccess) (void)hipDeviceSynchronize();  // as kmeans_cuda
  KnnJob job;
  const uint32_t feats = fp16x2 ? 2u * features_size : features_size;  // kmcuda.h:107-108
  RETERR(job.run(devs, virtual_shards(), k, metric, samples_size, feats, clusters_size, device_ptrs, verbosity,
                 fp16x2 != 0, samples, centroids, assignments, neighbors));
  DEBUG("return kmcudaSuccess\n");
  return kmcudaSuccess;
}

int kmamd_last_run_stats(uint32_t *iterations, double *loop_seconds, double *setup_seconds, uint32_t *shards,
                         uint32_t *rccl_ranks) {
  std::lock_guard<std::mutex> lock(g_last_run_mutex);
  if (iterations) *iterations = g_last_run.iterations;
  if (loop_seconds) *loop_seconds = g_last_run.loop_seconds;
  if (setup_seconds) *setup_seconds = g_last_run.setup_seconds;
  if (shards) *shards = g_last_run.shards;
  if (rccl_ranks) *rccl_ranks = g_last_run.rccl;
  return kmcudaSuccess;
}

int kmamd_last_run_collective(double *milliseconds, uint32_t *count) {
  std::lock_guard<std::mutex> lock(g_last_run_mutex);
  if (milliseconds) *milliseconds = g_last_run.collective_ms;
  if (count) *count = g_last_run.collectives;
  return kmcudaSuccess;
}

}  // extern "C"
