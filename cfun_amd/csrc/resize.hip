// Input pipeline: the volume resize in front of the network, on the device.
//   * heart: utils.resize_image(mode = 'self') (utils.py:389-393) = skimage.transform.resize(image, (max, max, min, 1),
//     order = 1, mode = 'constant', preserve_range, no anti-aliasing, clip) -- for a 3-D volume skimage >= 0.19 evaluates
//     this as scipy.ndimage.zoom(order = 1, mode = 'grid-constant', cval = 0, grid_mode = True): output index o samples
//     the input at c = (o + 0.5) * n_in / n_out - 0.5 per axis, linear weights, samples outside [0, n_in) are 0;
//   * LiTS: mold_inputs (LiTS_2017/model.py:1741-1761) = centre the volume in a zero PAD_IMAGE_SHAPE frame, then
//     resize(order = 0): index floor((o + 0.5) * n_pad / n_out).
// One kernel does both: the source is addressed through element strides (the loader's [H, W, D] array is read in place
// and written in the network's [D, H, W] order) and may sit at an offset inside a larger, virtual zero frame -- the padded
// copy the reference builds on the host is never materialised.  HBM-bound: one read of the source, one write of the result.
#include "common.h"

namespace {

struct ResizeArgs {
  int64_t s[3];     // element strides of the source axes (in output-axis order)
  int n[3];         // source extent per axis
  int P[3];         // virtual zero-padded frame extent per axis (>= n)
  int off[3];       // position of the source inside the frame
  int m[3];         // output extent
};

__device__ __forceinline__ float fetch(const float* __restrict__ in, const ResizeArgs& a, int i0, int i1, int i2) {
  i0 -= a.off[0]; i1 -= a.off[1]; i2 -= a.off[2];
  if ((unsigned)i0 >= (unsigned)a.n[0] || (unsigned)i1 >= (unsigned)a.n[1] || (unsigned)i2 >= (unsigned)a.n[2]) return 0.f;
  return in[i0 * a.s[0] + i1 * a.s[1] + i2 * a.s[2]];
}

template <int ORDER>
__global__ void __launch_bounds__(256)
k_resize3d(const float* __restrict__ in, float* __restrict__ out, ResizeArgs a, const float* __restrict__ clip_minmax) {
  const int64_t total = (int64_t)a.m[0] * a.m[1] * a.m[2];
  float lo = 0.f, hi = 0.f;
  if (clip_minmax) { lo = clip_minmax[0]; hi = clip_minmax[1]; }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int c2 = (int)(t % a.m[2]); t /= a.m[2];
    const int c1 = (int)(t % a.m[1]);
    const int c0 = (int)(t / a.m[1]);
    const int o[3] = {c0, c1, c2};
    float v;
    if (ORDER == 0) {
      int idx[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        int k = (int)floor(((double)o[d] + 0.5) * ((double)a.P[d] / (double)a.m[d]));
        idx[d] = k < 0 ? 0 : (k > a.P[d] - 1 ? a.P[d] - 1 : k);
      }
      v = fetch(in, a, idx[0], idx[1], idx[2]);
    } else {
      int i0[3];
      float w1[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double c = ((double)o[d] + 0.5) * ((double)a.P[d] / (double)a.m[d]) - 0.5;
        const double f = floor(c);
        i0[d] = (int)f;
        w1[d] = (float)(c - f);
      }
      v = 0.f;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const float w = (dz ? w1[0] : 1.f - w1[0]) * (dy ? w1[1] : 1.f - w1[1]) * (dx ? w1[2] : 1.f - w1[2]);
            // frame samples outside [0, P) are the constant 0 of mode = 'grid-constant'; inside the frame but outside the
            // source they are the zero padding -- the same value, so one bounds test serves both
            const int z = i0[0] + dz, y = i0[1] + dy, x = i0[2] + dx;
            if (w != 0.f && z >= 0 && z < a.P[0] && y >= 0 && y < a.P[1] && x >= 0 && x < a.P[2]) v += w * fetch(in, a, z, y, x);
          }
      if (clip_minmax) v = fminf(fmaxf(v, lo), hi);       // skimage's clip = True: to the range of the input
    }
    out[i] = v;
  }
}

}  // namespace

extern "C" int cfun_resize3d(const float* in, const int64_t* strides, const int32_t* dims, const int32_t* frame,
                             const int32_t* offset, float* out, const int32_t* out_dims, int32_t order,
                             const float* clip_minmax, cfun_stream_t stream) {
  if (!in || !out || !strides || !dims || !out_dims || (order != 0 && order != 1)) return CFUN_EINVAL;
  ResizeArgs a;
  for (int d = 0; d < 3; ++d) {
    a.s[d] = strides[d];
    a.n[d] = dims[d];
    a.P[d] = frame ? frame[d] : dims[d];
    a.off[d] = offset ? offset[d] : 0;
    a.m[d] = out_dims[d];
    if (a.n[d] <= 0 || a.m[d] <= 0 || a.P[d] < a.n[d] || a.off[d] < 0 || a.off[d] + a.n[d] > a.P[d]) return CFUN_EINVAL;
  }
  const int64_t total = (int64_t)a.m[0] * a.m[1] * a.m[2];
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (order == 0) hipLaunchKernelGGL(k_resize3d<0>, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), in, out, a, clip_minmax);
  else hipLaunchKernelGGL(k_resize3d<1>, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), in, out, a, clip_minmax);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}
