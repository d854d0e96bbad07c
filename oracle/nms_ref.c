/* TEST INFRASTRUCTURE -- plain C restatement of the reference's greedy 3-D NMS
 * (utils.py:122-157, compute_iou utils.py:50-70): float32 arithmetic in numpy's operation order,
 * built with -ffp-contract=off so no FMA is formed.  Used by tests to cross-check the numpy oracle and
 * the HIP kernel; never linked into the product.
 *
 * order[] must hold the indices sorted by score descending (ties: higher index first -- numpy's
 * argsort()[::-1] on the golden vectors); the sort itself is done by the caller. */
#include <stdint.h>
#include <stdlib.h>

static float fmax32(float a, float b) { return a > b ? a : b; }
static float fmin32(float a, float b) { return a < b ? a : b; }

int32_t cfun_ref_nms(const float* boxes, const int32_t* order, int32_t n, float threshold, int32_t max_num,
                     int32_t* keep) {
  unsigned char* dead = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int32_t cnt = 0;
  for (int32_t a = 0; a < n; ++a) {
    const int32_t i = order[a];
    if (dead[i]) continue;
    keep[cnt++] = i;
    if (cnt >= max_num) break;
    const float* bi = boxes + 6 * (size_t)i;
    const float vi = (bi[3] - bi[0]) * (bi[4] - bi[1]) * (bi[5] - bi[2]);
    for (int32_t b = a + 1; b < n; ++b) {
      const int32_t j = order[b];
      if (dead[j]) continue;
      const float* bj = boxes + 6 * (size_t)j;
      const float vj = (bj[3] - bj[0]) * (bj[4] - bj[1]) * (bj[5] - bj[2]);
      const float z1 = fmax32(bi[0], bj[0]), z2 = fmin32(bi[3], bj[3]);
      const float y1 = fmax32(bi[1], bj[1]), y2 = fmin32(bi[4], bj[4]);
      const float x1 = fmax32(bi[2], bj[2]), x2 = fmin32(bi[5], bj[5]);
      const float inter = fmax32(x2 - x1, 0.f) * fmax32(y2 - y1, 0.f) * fmax32(z2 - z1, 0.f);
      const float uni = vi + vj - inter;
      const float iou = inter / (uni + 1e-6f);
      if (iou > threshold) dead[j] = 1;
    }
  }
  free(dead);
  return cnt;
}
