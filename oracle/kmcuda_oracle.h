/*
 * kmcuda_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the arithmetic of src-d/kmcuda's hot path, used as the
 * parity checker for the HIP kernels.  Nothing under kmcuda_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity pinning: the reference is CUDA-only (needs nvcc + an NVIDIA GPU), so it cannot be
 * built into oracle/_ref here.  The oracle is pinned instead against the reference's own
 * known-answer tests (src/test.py): iteration-count pins 7 / 4 / 15+3 / 8 / 5 / 9 and
 * scikit-learn agreement thresholds, see tests/test_oracle_pins.py.
 *
 * Every function cites the reference file:line (relative to /root/reference/src) it follows.
 * Samples are row-major N x D here (the public API layout, kmcuda.h:107); the reference
 * transposes them on device, which changes addressing only, never the arithmetic.
 */
#ifndef KMCUDA_ORACLE_H
#define KMCUDA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { KMO_L2 = 0, KMO_COS = 1 };                 /* kmcuda.h:75-81 */
enum { KMO_INIT_RANDOM = 0, KMO_INIT_PLUSPLUS = 1, KMO_INIT_AFKMC2 = 2, KMO_INIT_IMPORT = 3 }; /* kmcuda.h:57-72 */

/* fp_abstraction.h:88-90  _fma(acc,v1,v2) = __fmaf_rd(v1,v2,acc): fused a*b+c rounded toward -inf */
float kmo_fma_rd(float a, float b, float c);
float kmo_fma_rd_portable(float a, float b, float c);
int   kmo_have_avx512(void);
int   kmo_num_threads(void);

/* metric_abstraction.h:21-36 (L2) / :149-158 (cos -> 1) */
void kmo_sum_squares(int metric, uint32_t K, uint32_t D, const float *centroids, float *csqr);
/* kmeans.cu:330-341 Kahan dot with round-down FMA */
float kmo_kahan_dot(const float *a, const float *b, uint32_t D);
/* metric_abstraction.h:59-101 (L2) / :179-218 (cos): distance / distance_t / distance_tt */
float kmo_distance(int metric, const float *a, const float *b, uint32_t D);

/* kmeans.cu:293-364 kmeans_assign_lloyd (== _smallc :214-291 arithmetic) */
void kmo_lloyd_assign(int metric, uint32_t N, uint32_t D, uint32_t K, const float *samples,
                      const float *centroids, uint32_t *assignments, uint32_t *assignments_prev,
                      uint32_t *changed);
/* kmeans.cu:366-429 kmeans_adjust + metric_abstraction.h:138-144,255-272 normalize */
void kmo_adjust(int metric, uint32_t N, uint32_t D, uint32_t K, const float *samples,
                const uint32_t *assignments_prev, const uint32_t *assignments,
                float *centroids, uint32_t *ccounts);

/* kmeans.cu:431-485 */
void kmo_yy_init(int metric, uint32_t N, uint32_t D, uint32_t K, uint32_t G, const float *samples,
                 const float *centroids, const uint32_t *assignments, const uint32_t *groups,
                 float *bounds);
/* kmeans.cu:487-499 ; drifts has K*D + K floats, old centroids in [0,K*D) */
void kmo_yy_calc_drifts(int metric, uint32_t D, uint32_t K, const float *centroids, float *drifts);
/* kmeans.cu:501-538 */
void kmo_yy_group_max_drifts(uint32_t D, uint32_t K, uint32_t G, const uint32_t *groups, float *drifts);
/* kmeans.cu:540-582 ; returns number passed */
uint32_t kmo_yy_global_filter(int metric, uint32_t N, uint32_t D, uint32_t K, uint32_t G,
                              const float *samples, const float *centroids, const uint32_t *groups,
                              const float *drifts, const uint32_t *assignments,
                              uint32_t *assignments_prev, float *bounds, uint32_t *passed);
/* kmeans.cu:584-672 ; returns number changed */
uint32_t kmo_yy_local_filter(int metric, uint32_t N, uint32_t D, uint32_t K, uint32_t G,
                             const float *samples, const uint32_t *passed, uint32_t npassed,
                             const float *centroids, const uint32_t *groups, const float *drifts,
                             uint32_t *assignments, float *bounds);

/* kmcuda.cc:245-261 (random: srand + libstdc++ random_shuffle) and :262-336 (k-means++),
 * kmeans.cu:42-67,774-828 (kmeans_plus_plus kernel + host sum).  Calls srand(seed). */
int kmo_init_centroids(int method, int metric, uint32_t N, uint32_t D, uint32_t K, uint32_t seed,
                       const float *samples, float *centroids);

/* kmeans.cu:674-691,1265-1300 */
float kmo_average_distance(int metric, uint32_t N, uint32_t D, const float *samples,
                           const float *centroids, const uint32_t *assignments);

/* Whole kmeans_cuda() flow (kmcuda.cc:402-531 -> kmeans.cu:934-1263).
 * iter_log receives the "iteration %d: %u reassignments" counts in print order
 * (kmeans.cu:706), *n_iter_log how many (capacity iter_log_cap).  init==IMPORT reads centroids.
 * Returns 0 on success, 1 on invalid arguments (kmcuda.cc:19-61). */
int kmo_kmeans(int init, float tolerance, float yinyang_t, int metric, uint32_t N, uint32_t D,
               uint32_t K, uint32_t seed, const float *samples, float *centroids,
               uint32_t *assignments, float *average_distance,
               uint32_t *iter_log, uint32_t iter_log_cap, uint32_t *n_iter_log);

/* fp16x2 storage mode of THIS repository (fp32 arithmetic on half values, centroids rounded to half
 * after every update); affects kmo_kmeans only.  kmo_quantize_half: float -> half (RN) -> float. */
void kmo_set_fp16_storage(int on);
/* fp16x2 arithmetic of the oracle: 0 = fp32, 1 = the storage mode above, 2 = the REFERENCE's half2
 * arithmetic (fp_abstraction.h:100-182: packed binary16 add / sub / mul / fma rounded to nearest, two
 * interleaved Kahan accumulators folded by _fin = hi + lo, __int2half_rd constants, half compares);
 * inputs must hold half-representable values and D must be even.  Affects every kmo_* entry point that
 * computes distances, assignments or updates.  kmo_h_rn / kmo_h_from_int_rd expose the two roundings. */
void kmo_set_fp16_mode(int mode);
float kmo_h_rn(double v);
float kmo_h_from_int_rd(long long v);
/* m of init = KMO_INIT_AFKMC2 (0 => 200), kmcuda.h:89-92 */
void kmo_set_afkmc2_m(uint32_t m);
/* AFK-MC2's seed scrambling: 0 = rocRAND's constants (default: meets all four of the reference's pins), 1 = cuRAND's as
 * quoted in kmcuda_oracle.c (three of four) */
void kmo_set_afkmc2_seeding(int curand);
/* raw XORWOW draws of the stream (seed, subsequence, offset) under either seed scrambling */
void kmo_xorwow_draws(int curand_seeding, unsigned long long seed, unsigned long long subsequence,
                      unsigned long long offset, uint32_t n, uint32_t *out);
float kmo_quantize_half(float x);

/* knn.cu:19-58 / :61-131 / :133-243 and kmcuda.cc:648-691 (inverse assignments) */
void kmo_knn_inverse(uint32_t N, uint32_t K, const uint32_t *assignments, uint32_t *inv, uint32_t *offsets);
void kmo_knn_radiuses(int metric, uint32_t N, uint32_t D, uint32_t K, const float *samples,
                      const float *centroids, const uint32_t *inv, const uint32_t *offsets, float *radiuses);
void kmo_knn_cluster_distances(int metric, uint32_t D, uint32_t K, const float *centroids, float *dists);
int kmo_knn(uint32_t k, int metric, uint32_t N, uint32_t D, uint32_t K, const float *samples,
            const float *centroids, const uint32_t *assignments, uint32_t *neighbors,
            uint64_t *dists_calced);

#ifdef __cplusplus
}
#endif
#endif
