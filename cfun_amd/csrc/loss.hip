// Mask-head losses on NDHWC logits: softmax (model.py:794,799), cross-entropy (model.py:909-935) and the
// 3-D Sobel "edge agreement" loss (model.py:938-981).  All HBM-bound single passes; per-block fp64
// partial sums + a one-block finalize keep the scalar losses deterministic.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;
constexpr int kMaxC = 32;

inline unsigned vox_grid(int64_t nvox) {
  int64_t b = (nvox + kBlock - 1) / kBlock;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

__device__ __forceinline__ double block_sum(double v) {
  __shared__ double red[kBlock / 64];
  v = cfun_wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kBlock / 64; ++w) s += red[w];
  return s;  // valid in thread 0
}

__global__ void k_finalize_sum(const double* __restrict__ partial, int blocks, double mul, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 64) s += partial[i];
  s = cfun_wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * mul);
}

// ------------------------------------------------------------------ softmax / CE
template <int CT>  // CT > 0: compile-time channel count (registers); CT == 0: runtime C <= kMaxC
__global__ void __launch_bounds__(kBlock)
k_softmax_fwd(const float* __restrict__ logits, float* __restrict__ probs, int64_t nvox, int Crt) {
  const int C = CT > 0 ? CT : Crt;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * kBlock) {
    const float* l = logits + v * C;
    float x[CT > 0 ? CT : kMaxC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) { x[c] = l[c]; m = fmaxf(m, x[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) { x[c] = expf(x[c] - m); s += x[c]; }
    float* p = probs + v * C;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) p[c] = x[c] / s;
  }
}

// dl = p * (g - sum_c g*p)
template <int CT>
__global__ void __launch_bounds__(kBlock)
k_softmax_bwd(const float* __restrict__ probs, const float* __restrict__ dprobs, float* __restrict__ dlogits,
              int64_t nvox, int Crt) {
  const int C = CT > 0 ? CT : Crt;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * kBlock) {
    float p[CT > 0 ? CT : kMaxC], g[CT > 0 ? CT : kMaxC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) { p[c] = probs[v * C + c]; g[c] = dprobs[v * C + c]; dot += p[c] * g[c]; }
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) dlogits[v * C + c] = p[c] * (g[c] - dot);
  }
}

template <int CT>
__global__ void __launch_bounds__(kBlock)
k_ce_fwd(const float* __restrict__ logits, const uint8_t* __restrict__ labels, double* __restrict__ partial,
         int64_t nvox, int Crt) {
  const int C = CT > 0 ? CT : Crt;
  double acc = 0.0;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * kBlock) {
    const float* l = logits + v * C;
    float x[CT > 0 ? CT : kMaxC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) { x[c] = l[c]; m = fmaxf(m, x[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) s += expf(x[c] - m);
    const int lab = labels[v];
    float xl = 0.f;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c == lab) xl = x[c];
    acc += (double)((m + logf(s)) - xl);
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <int CT>
__global__ void __launch_bounds__(kBlock)
k_ce_bwd(const float* __restrict__ logits, const uint8_t* __restrict__ labels, const float* __restrict__ gscale,
         float* __restrict__ dlogits, int64_t nvox, int Crt) {
  const int C = CT > 0 ? CT : Crt;
  const float gs = gscale[0] / (float)nvox;
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * kBlock) {
    const float* l = logits + v * C;
    float x[CT > 0 ? CT : kMaxC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) { x[c] = l[c]; m = fmaxf(m, x[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) { x[c] = expf(x[c] - m); s += x[c]; }
    const int lab = labels[v];
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) dlogits[v * C + c] = gs * (x[c] * inv - (c == lab ? 1.f : 0.f));
  }
}

// ------------------------------------------------------------------ Sobel edge loss
// kernel_x[dz][dy][dx] = A[dz]*B[dy]*A[dx] (derivative along y), kernel_y = A[dy]*B[dz]*A[dx] (derivative
// along z), A = (1,2,1), B = (1,0,-1)  (model.py:947-951; F.conv3d is a cross-correlation, valid padding).
__device__ __forceinline__ void sobel_w(int dz, int dy, int dx, float* w0, float* w1) {
  const float A[3] = {1.f, 2.f, 1.f}, B[3] = {1.f, 0.f, -1.f};
  *w0 = A[dz] * B[dy] * A[dx];
  *w1 = A[dy] * B[dz] * A[dx];
}

template <int CT>  // classes incl. background; channels 1..CT-1 contribute
__device__ __forceinline__ void sobel_at(const float* __restrict__ probs, const uint8_t* __restrict__ labels,
                                         int64_t nbase, int z, int y, int x, int H, int W, float (&p0)[CT],
                                         float (&p1)[CT], float (&t0)[CT], float (&t1)[CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) { p0[c] = 0.f; p1[c] = 0.f; t0[c] = 0.f; t1[c] = 0.f; }
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        float w0, w1;
        sobel_w(dz, dy, dx, &w0, &w1);
        if (w0 == 0.f && w1 == 0.f) continue;
        const int64_t vi = nbase + ((int64_t)(z + dz) * H + (y + dy)) * W + (x + dx);
        const float* pp = probs + vi * CT;
        const int lab = labels[vi];
#pragma unroll
        for (int c = 1; c < CT; ++c) {
          const float pv = pp[c];
          const float tv = (c == lab) ? 1.f : 0.f;
          p0[c] += w0 * pv; p1[c] += w1 * pv;
          t0[c] += w0 * tv; t1[c] += w1 * tv;
        }
      }
}

template <int CT>
__global__ void __launch_bounds__(kBlock)
k_edge_fwd(const float* __restrict__ probs, const uint8_t* __restrict__ labels, double* __restrict__ partial, int n,
           int D, int H, int W) {
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int64_t per = (int64_t)Do * Ho * Wo, total = per * n;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho); t /= Ho;
    const int z = (int)(t % Do);
    const int64_t r = t / Do;
    float p0[CT], p1[CT], t0[CT], t1[CT];
    sobel_at<CT>(probs, labels, r * D * H * W, z, y, x, H, W, p0, p1, t0, t1);
#pragma unroll
    for (int c = 1; c < CT; ++c) {
      const float pm = sqrtf(p0[c] * p0[c] + p1[c] * p1[c] + p0[c] * p0[c]);   // channel 0 twice (model.py:969-972)
      const float tm = sqrtf(t0[c] * t0[c] + t1[c] * t1[c] + t0[c] * t0[c]);
      const float d = pm - tm;
      acc += (double)(d * d);
    }
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// dc[o][c][0..1] = dL/d(c0), dL/d(c1) at every valid output voxel o
template <int CT>
__global__ void __launch_bounds__(kBlock)
k_edge_bwd_dc(const float* __restrict__ probs, const uint8_t* __restrict__ labels, const float* __restrict__ gscale,
              float* __restrict__ dc, int n, int D, int H, int W) {
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int64_t per = (int64_t)Do * Ho * Wo, total = per * n;
  const float gs = gscale[0] * 2.f / ((float)per * (float)n);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho); t /= Ho;
    const int z = (int)(t % Do);
    const int64_t r = t / Do;
    float p0[CT], p1[CT], t0[CT], t1[CT];
    sobel_at<CT>(probs, labels, r * D * H * W, z, y, x, H, W, p0, p1, t0, t1);
    float* o = dc + i * (2 * (CT - 1));
#pragma unroll
    for (int c = 1; c < CT; ++c) {
      const float pm = sqrtf(p0[c] * p0[c] + p1[c] * p1[c] + p0[c] * p0[c]);
      const float tm = sqrtf(t0[c] * t0[c] + t1[c] * t1[c] + t0[c] * t0[c]);
      const float k = gs * (pm - tm) / pm;        // 0/0 -> NaN exactly like torch's sqrt backward (App. A-13)
      o[(c - 1) * 2] = k * 2.f * p0[c];
      o[(c - 1) * 2 + 1] = k * p1[c];
    }
  }
}

template <int CT>
__global__ void __launch_bounds__(kBlock)
k_edge_bwd_gather(const float* __restrict__ dc, float* __restrict__ dprobs, int n, int D, int H, int W) {
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int64_t total = (int64_t)n * D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;
    const int z = (int)(t % D);
    const int64_t r = t / D;
    float g[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) g[c] = 0.f;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          float w0, w1;
          sobel_w(dz, dy, dx, &w0, &w1);
          if (w0 == 0.f && w1 == 0.f) continue;
          const int oz = z - dz, oy = y - dy, ox = x - dx;
          if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
          const float* d = dc + (((r * Do + oz) * Ho + oy) * Wo + ox) * (2 * (CT - 1));
#pragma unroll
          for (int c = 1; c < CT; ++c) g[c] += w0 * d[(c - 1) * 2] + w1 * d[(c - 1) * 2 + 1];
        }
#pragma unroll
    for (int c = 0; c < CT; ++c) dprobs[i * CT + c] = g[c];
  }
}

#define DISPATCH_C(C, CALL)            \
  if ((C) == 8) { CALL(8) }            \
  else if ((C) == 3) { CALL(3) }       \
  else if ((C) == 2) { CALL(2) }       \
  else { CALL(0) }

}  // namespace

extern "C" {

int cfun_softmax_fwd(const float* logits, float* probs, int64_t nvox, int32_t C, cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || C > kMaxC) return CFUN_EINVAL;
#define CALL(CT) hipLaunchKernelGGL(k_softmax_fwd<CT>, dim3(vox_grid(nvox)), dim3(kBlock), 0, cfun_st(stream), logits, probs, nvox, C);
  DISPATCH_C(C, CALL)
#undef CALL
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, int64_t nvox, int32_t C,
                     cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || C > kMaxC) return CFUN_EINVAL;
#define CALL(CT) hipLaunchKernelGGL(k_softmax_bwd<CT>, dim3(vox_grid(nvox)), dim3(kBlock), 0, cfun_st(stream), probs, dprobs, dlogits, nvox, C);
  DISPATCH_C(C, CALL)
#undef CALL
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

size_t cfun_loss_workspace_bytes(int64_t nvox) { (void)nvox; return kMaxBlocks * sizeof(double); }

int cfun_softmax_ce_fwd(const float* logits, const uint8_t* labels, float* loss, int64_t nvox, int32_t C, void* ws,
                        size_t ws_bytes, cfun_stream_t stream) {
  if (C <= 0 || C > kMaxC) return CFUN_EINVAL;
  if (nvox <= 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), cfun_st(stream));
  if (ws_bytes < kMaxBlocks * sizeof(double)) return CFUN_EWORKSPACE;
  const unsigned blocks = vox_grid(nvox);
#define CALL(CT) hipLaunchKernelGGL(k_ce_fwd<CT>, dim3(blocks), dim3(kBlock), 0, cfun_st(stream), logits, labels, (double*)ws, nvox, C);
  DISPATCH_C(C, CALL)
#undef CALL
  hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks, 1.0 / (double)nvox, loss);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_softmax_ce_bwd(const float* logits, const uint8_t* labels, const float* gscale, float* dlogits,
                        int64_t nvox, int32_t C, cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || C > kMaxC) return CFUN_EINVAL;
#define CALL(CT) hipLaunchKernelGGL(k_ce_bwd<CT>, dim3(vox_grid(nvox)), dim3(kBlock), 0, cfun_st(stream), logits, labels, gscale, dlogits, nvox, C);
  DISPATCH_C(C, CALL)
#undef CALL
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_edge_loss_fwd(const float* probs, const uint8_t* labels, float* loss, int32_t n, int32_t D, int32_t H,
                       int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (C != 8 && C != 3) return CFUN_EINVAL;
  if (n <= 0 || D < 3 || H < 3 || W < 3) return (int)hipMemsetAsync(loss, 0, sizeof(float), cfun_st(stream));
  if (ws_bytes < kMaxBlocks * sizeof(double)) return CFUN_EWORKSPACE;
  const int64_t per = (int64_t)(D - 2) * (H - 2) * (W - 2);
  const unsigned blocks = vox_grid(per * n);
  if (C == 8) hipLaunchKernelGGL(k_edge_fwd<8>, dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (double*)ws, n, D, H, W);
  else hipLaunchKernelGGL(k_edge_fwd<3>, dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (double*)ws, n, D, H, W);
  hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks,
                     1.0 / ((double)per * (double)n), loss);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

size_t cfun_edge_loss_bwd_workspace_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C) {
  if (n <= 0 || D < 3 || H < 3 || W < 3 || C < 2) return 256;
  return cfun_align_up((size_t)n * (D - 2) * (H - 2) * (W - 2) * 2 * (C - 1) * sizeof(float), 256);
}

int cfun_edge_loss_bwd(const float* probs, const uint8_t* labels, const float* gscale, float* dprobs, int32_t n,
                       int32_t D, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (C != 8 && C != 3) return CFUN_EINVAL;
  const int64_t total = (int64_t)n * D * H * W;
  if (total <= 0) return CFUN_OK;
  if (D < 3 || H < 3 || W < 3) return (int)hipMemsetAsync(dprobs, 0, total * C * sizeof(float), cfun_st(stream));
  if (ws_bytes < cfun_edge_loss_bwd_workspace_bytes(n, D, H, W, C)) return CFUN_EWORKSPACE;
  const int64_t per = (int64_t)(D - 2) * (H - 2) * (W - 2);
  if (C == 8) {
    hipLaunchKernelGGL(k_edge_bwd_dc<8>, dim3(vox_grid(per * n)), dim3(kBlock), 0, cfun_st(stream), probs, labels, gscale, (float*)ws, n, D, H, W);
    hipLaunchKernelGGL(k_edge_bwd_gather<8>, dim3(vox_grid(total)), dim3(kBlock), 0, cfun_st(stream), (const float*)ws, dprobs, n, D, H, W);
  } else {
    hipLaunchKernelGGL(k_edge_bwd_dc<3>, dim3(vox_grid(per * n)), dim3(kBlock), 0, cfun_st(stream), probs, labels, gscale, (float*)ws, n, D, H, W);
    hipLaunchKernelGGL(k_edge_bwd_gather<3>, dim3(vox_grid(total)), dim3(kBlock), 0, cfun_st(stream), (const float*)ws, dprobs, n, D, H, W);
  }
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // extern "C"
