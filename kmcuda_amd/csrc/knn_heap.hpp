// knn_heap.hpp -- the k-NN heap step, ONE copy for every search kernel (knn.hip, knn_f16.hip).
#pragma once
#include <stdint.h>

namespace kmx {

// push_sample (knn.cu:133-175): replace-root + sift-down on a max-heap of interleaved (distance, index)
// pairs.  Which child moves up on equal distances decides the order of tied neighbours in the output, so the
// compares are the reference's, one for one.
__device__ __forceinline__ void knn_push_sample(uint32_t k, float dist, uint32_t index, float *heap) {
  uint32_t pos = 0;
  uint32_t *heapi = reinterpret_cast<uint32_t *>(heap);
  while (true) {
    float left = 0.f, right = 0.f;
    bool left_le, right_le;
    if ((2 * pos + 1) < k) { left = heap[4 * pos + 2]; left_le = dist >= left; } else left_le = true;
    if ((2 * pos + 2) < k) { right = heap[4 * pos + 4]; right_le = dist >= right; } else right_le = true;
    if (left_le && right_le) {
      heap[2 * pos] = dist;
      heapi[2 * pos + 1] = index;
      break;
    }
    bool go_right;
    if (!left_le && !right_le) go_right = left <= right;
    else go_right = left_le;
    if (go_right) {
      heap[2 * pos] = right;
      heapi[2 * pos + 1] = heapi[4 * pos + 5];
      pos = 2 * pos + 2;
    } else {
      heap[2 * pos] = left;
      heapi[2 * pos + 1] = heapi[4 * pos + 3];
      pos = 2 * pos + 1;
    }
  }
}

}  // namespace kmx
