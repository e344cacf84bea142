/*
 * kmcuda.h -- drop-in C ABI of the MI355X-native K-means / K-nn hot path.
 *
 * Binary- and source-compatible with the public header of src-d/kmcuda
 * (reference: src/kmcuda.h:41-81 enums, :118-123 kmeans_cuda, :150-155 knn_cuda, :168-194
 * string maps), so that code written against libKMCUDA links against this library unchanged.
 * Everything behind the two entry points is new: HIP/CDNA4 kernels (gfx950), row-major
 * samples, matrix-core assignment with exact-arithmetic refinement.
 *
 * Not thread safe (same contract as the reference, kmcuda.h:25-26): both entry points call
 * srand()/rand() and own process-wide device state for the duration of the call.
 */
#ifndef KMCUDA_KMCUDA_H
#define KMCUDA_KMCUDA_H

#include <stdint.h>

/* Return codes (reference kmcuda.h:41-54). */
typedef enum {
  kmcudaSuccess = 0,              /* all good */
  kmcudaInvalidArguments,         /* argument validation failed (kmcuda.cc:19-61, :537-570) */
  kmcudaNoSuchDevice,             /* device mask names a GPU that does not exist */
  kmcudaMemoryAllocationFailure,  /* hipMalloc failed */
  kmcudaRuntimeError,             /* a kernel launch / runtime call failed */
  kmcudaMemoryCopyError           /* hipMemcpy failed */
} KMCUDAResult;

/* Centroid seeding (reference kmcuda.h:57-72). */
typedef enum {
  kmcudaInitMethodRandom = 0,  /* first K entries of a random_shuffle over rand() */
  kmcudaInitMethodPlusPlus,    /* k-means++ (distance-proportional, as the reference does it) */
  kmcudaInitMethodAFKMC2,      /* AFK-MC2 (init_params: uint32_t* m, 0 => 200); draws restated from the published XORWOW */
  kmcudaInitMethodImport       /* caller supplies the centroids in `centroids` */
} KMCUDAInitMethod;

/* Distance metric (reference kmcuda.h:75-81). */
typedef enum {
  kmcudaDistanceMetricL2,     /* Euclidean */
  kmcudaDistanceMetricCosine  /* angular; samples must have unit L2 norm */
} KMCUDADistanceMetric;

#ifdef __cplusplus
extern "C" {
#endif

/*
 * K-means (Lloyd, optionally Yinyang-accelerated) on the GPUs named by `device`.
 * Replaces reference src/kmcuda.h:118-123 / src/kmcuda.cc:402-531.
 *
 *   init, init_params  seeding method; init_params is a uint32_t* (m) for AFK-MC2, else ignored
 *   tolerance          stop when the reassigned fraction drops to <= tolerance   [0, 1]
 *   yinyang_t          relative number of Yinyang groups, 0 disables Yinyang     [0, 0.5]
 *   metric             L2 or angular
 *   samples_size       N;  features_size D (half2 count when fp16x2);  clusters_size K
 *   seed               passed to srand()
 *   device             bit mask of GPUs (bit n = GPU n), 0 = all
 *   device_ptrs        < 0: samples/centroids/assignments are host pointers; otherwise the GPU
 *                      index they live on (never modified in place by this implementation)
 *   fp16x2             samples are N x (2*D) halves
 *   verbosity          0 silent, 1 progress ("iteration %d: %u reassignments"), >=2 debug
 *   samples            [N x D] row major, in
 *   centroids          [K x D] row major, out (in as well when init == Import)
 *   assignments        [N], out
 *   average_distance   optional out: mean distance sample -> its centroid
 */
KMCUDAResult kmeans_cuda(
    KMCUDAInitMethod init, const void *init_params, float tolerance, float yinyang_t,
    KMCUDADistanceMetric metric, uint32_t samples_size, uint16_t features_size,
    uint32_t clusters_size, uint32_t seed, uint32_t device, int32_t device_ptrs,
    int32_t fp16x2, int32_t verbosity, const float *samples, float *centroids,
    uint32_t *assignments, float *average_distance);

/*
 * K nearest neighbours of every sample, pruned with precomputed K-means clusters.
 * Replaces reference src/kmcuda.h:150-155 / src/kmcuda.cc:572-730.
 *   neighbors          [N x k] row major, out, ascending distance
 * Unlike the reference (kmcuda.cc:583-584 discards the validation result) invalid arguments
 * are reported.
 */
KMCUDAResult knn_cuda(
    uint16_t k, KMCUDADistanceMetric metric, uint32_t samples_size,
    uint16_t features_size, uint32_t clusters_size, uint32_t device,
    int32_t device_ptrs, int32_t fp16x2, int32_t verbosity,
    const float *samples, const float *centroids, const uint32_t *assignments,
    uint32_t *neighbors);

#ifdef __cplusplus
}  /* extern "C" */

#include <string>
#include <unordered_map>

/* String maps for bindings (reference kmcuda.h:168-194). */
namespace {
namespace kmcuda {
const std::unordered_map<std::string, KMCUDAInitMethod> init_methods{
    {"kmeans++", kmcudaInitMethodPlusPlus}, {"k-means++", kmcudaInitMethodPlusPlus},
    {"afkmc2", kmcudaInitMethodAFKMC2},     {"afk-mc2", kmcudaInitMethodAFKMC2},
    {"random", kmcudaInitMethodRandom}};
const std::unordered_map<std::string, KMCUDADistanceMetric> metrics{
    {"euclidean", kmcudaDistanceMetricL2}, {"L2", kmcudaDistanceMetricL2},
    {"l2", kmcudaDistanceMetricL2},        {"cos", kmcudaDistanceMetricCosine},
    {"cosine", kmcudaDistanceMetricCosine}, {"angular", kmcudaDistanceMetricCosine}};
const std::unordered_map<int, const char *> statuses{
    {kmcudaSuccess, "Success"},
    {kmcudaInvalidArguments, "InvalidArguments"},
    {kmcudaNoSuchDevice, "NoSuchDevice"},
    {kmcudaMemoryAllocationFailure, "MemoryAllocationFailure"},
    {kmcudaRuntimeError, "RuntimeError"},
    {kmcudaMemoryCopyError, "MemoryCopyError"}};
}  // namespace kmcuda
}  // namespace
#endif /* __cplusplus */

#endif /* KMCUDA_KMCUDA_H */
