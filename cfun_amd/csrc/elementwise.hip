// HBM-bound passes of the hot path: activation derivatives, InstanceNorm3d(+LeakyReLU) forward/backward,
// per-channel reductions, nearest-x2 upsample backward, MaxPool3d(2,2), halo pack/unpack.
// All NDHWC fp32, float4 (16 B/lane) accesses along the channel axis, grid-stride loops.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kBlock = 256;

inline unsigned ew_grid(int64_t work_items) {
  int64_t b = (work_items + kBlock - 1) / kBlock;
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ------------------------------------------------------------------ simple elementwise
// launch of the per-sample apply kernels: (blocks, N) with >= 4 voxels per thread
inline unsigned apply_blocks(int64_t V, int lanes) {
  int64_t b = (V + (int64_t)lanes * 4 - 1) / ((int64_t)lanes * 4);
  if (b < 1) b = 1;
  if (b > 16384) b = 16384;
  return (unsigned)b;
}

__global__ void __launch_bounds__(kBlock) k_lrelu_fwd(const float4* x, float4* y, int64_t n4, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    float4 v = x[i];
    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
    y[i] = v;
  }
}
__global__ void k_lrelu_fwd_tail(const float* x, float* y, int64_t from, int64_t n, float slope) {
  int64_t i = from + threadIdx.x;
  if (i < n) y[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}
__global__ void __launch_bounds__(kBlock)
k_lrelu_bwd(const float4* x, const float4* dy, float4* dx, int64_t n4, float slope, const float4* add) {
  // add: a second gradient of x (x also feeds a residual / another branch), summed here instead of by a pass of its own
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = x[i];
    float4 g = dy[i];
    g.x = v.x > 0.f ? g.x : g.x * slope; g.y = v.y > 0.f ? g.y : g.y * slope;
    g.z = v.z > 0.f ? g.z : g.z * slope; g.w = v.w > 0.f ? g.w : g.w * slope;
    if (add) { const float4 a = add[i]; g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w; }
    dx[i] = g;
  }
}
__global__ void k_lrelu_bwd_tail(const float* x, const float* dy, float* dx, int64_t from, int64_t n, float slope) {
  int64_t i = from + threadIdx.x;
  if (i < n) dx[i] = x[i] > 0.f ? dy[i] : dy[i] * slope;
}
// channel-strided variants (C % 4 == 0): rows of C floats inside a wider NDHWC buffer (zero-copy concat, section 3.5);
// cg = C/4 float4 groups per voxel, xg / yg = row strides in float4s
__global__ void __launch_bounds__(kBlock)
k_lrelu_fwd_s(const float4* x, float4* y, int64_t nvox, int cg, int64_t xg, int64_t yg, float slope) {
  // thread = (voxel lane, group): consecutive threads walk a voxel's row, then the next voxel -- coalesced on both
  // sides, and no per-element 64-bit division (a flat index / cg costs 5x the kernel time)
  const int lanes = kBlock / cg, c = threadIdx.x % cg, vl = threadIdx.x / cg;
  if (vl >= lanes) return;
  for (int64_t v = (int64_t)blockIdx.x * lanes + vl; v < nvox; v += (int64_t)gridDim.x * lanes) {
    float4 a = x[v * xg + c];
    a.x = a.x > 0.f ? a.x : a.x * slope; a.y = a.y > 0.f ? a.y : a.y * slope;
    a.z = a.z > 0.f ? a.z : a.z * slope; a.w = a.w > 0.f ? a.w : a.w * slope;
    y[v * yg + c] = a;
  }
}
__global__ void __launch_bounds__(kBlock)
k_lrelu_bwd_s(const float4* x, const float4* dy, float4* dx, int64_t nvox, int cg, int64_t dyg, float slope, const float4* add) {
  const int lanes = kBlock / cg, c = threadIdx.x % cg, vl = threadIdx.x / cg;
  if (vl >= lanes) return;
  for (int64_t v = (int64_t)blockIdx.x * lanes + vl; v < nvox; v += (int64_t)gridDim.x * lanes) {
    const float4 a = x[v * cg + c];
    float4 g = dy[v * dyg + c];
    g.x = a.x > 0.f ? g.x : g.x * slope; g.y = a.y > 0.f ? g.y : g.y * slope;
    g.z = a.z > 0.f ? g.z : g.z * slope; g.w = a.w > 0.f ? g.w : g.w * slope;
    if (add) { const float4 b = add[v * cg + c]; g.x += b.x; g.y += b.y; g.z += b.z; g.w += b.w; }
    dx[v * cg + c] = g;
  }
}
__global__ void __launch_bounds__(kBlock) k_add(const float4* a, const float4* b, float4* o, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 u = a[i], v = b[i];
    o[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}
__global__ void k_add_tail(const float* a, const float* b, float* o, int64_t from, int64_t n) {
  int64_t i = from + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

// g = dy * act'(y) * scale ; one thread per element (C may be any size)
__global__ void __launch_bounds__(kBlock)
k_act_bwd(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ scale,
          float* __restrict__ g, int64_t total, int C, int64_t vox_per_n, int act, float slope, int scale_mode) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    float d = dy[i];
    if (act != CFUN_ACT_NONE) {
      const float yy = y[i];
      if (!(yy > 0.f)) d = (act == CFUN_ACT_RELU) ? 0.f : d * slope;
    }
    if (scale_mode) {
      const int64_t v = i / C;
      const int c = (int)(i - v * C);
      d *= (scale_mode == 1) ? scale[c] : scale[(v / vox_per_n) * C + c];
    }
    g[i] = d;
  }
}

// VEC = 4: float4 accesses (C % 4 == 0, 16-byte aligned); VEC = 1: scalar path for odd channel counts
// (e.g. the 3-class LiTS heads) -- same kernels, one channel per "group".
template <int VEC> struct Vec;
template <> struct Vec<4> {
  static __device__ __forceinline__ void load(const float* p, int64_t e, float (&a)[4]) {
    const float4 v = reinterpret_cast<const float4*>(p)[e];
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, int64_t e, const float (&a)[4]) {
    reinterpret_cast<float4*>(p)[e] = make_float4(a[0], a[1], a[2], a[3]);
  }
};
template <> struct Vec<1> {
  static __device__ __forceinline__ void load(const float* p, int64_t e, float (&a)[1]) { a[0] = p[e]; }
  static __device__ __forceinline__ void store(float* p, int64_t e, const float (&a)[1]) { p[e] = a[0]; }
};

// ------------------------------------------------------------------ per-(n,c) reductions over voxels
// Thread = (voxel lane, 4-channel group).  Accumulates NQ quantities in fp64, reduces the voxel lanes through
// LDS and writes partial[n][block][q][c] (deterministic two-stage reduction; no atomics).
template <int VEC>
struct StatSumSq {  // sum x, sum x^2
  static constexpr int NQ = 2;
  const float* x;
  __device__ void operator()(int64_t e, int, int, double (&acc)[2][VEC], int64_t, int) const {
    float a[VEC];
    Vec<VEC>::load(x, e, a);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[0][j] += a[j]; acc[1][j] += (double)a[j] * a[j]; }
  }
};
template <int VEC>
struct StatSum {  // sum g
  static constexpr int NQ = 1;
  const float* x;
  __device__ void operator()(int64_t e, int, int, double (&acc)[1][VEC], int64_t, int) const {
    float a[VEC];
    Vec<VEC>::load(x, e, a);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[0][j] += a[j];
  }
};
template <int VEC>
struct StatNormBwd {  // sum gn, sum gn*xhat with gn = dy*lrelu'(xhat)
  static constexpr int NQ = 2;
  const float* x;
  const float* dy;
  const float* stats;  // [N,C,2]
  int C;
  float slope;
  int64_t dyg;         // dy row stride in VEC groups (C / VEC when dense)
  __device__ void operator()(int64_t e, int n, int c0, double (&acc)[2][VEC], int64_t vox, int cg) const {
    float a[VEC], b[VEC];
    Vec<VEC>::load(x, e, a);
    Vec<VEC>::load(dy, vox * dyg + cg, b);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float mean = stats[((int64_t)n * C + c0 + j) * 2], rstd = stats[((int64_t)n * C + c0 + j) * 2 + 1];
      const float xh = (a[j] - mean) * rstd;
      const float gn = xh > 0.f ? b[j] : b[j] * slope;
      acc[0][j] += gn;
      acc[1][j] += (double)gn * xh;
    }
  }
};

template <class F, int VEC>
__global__ void __launch_bounds__(kBlock)
k_channel_reduce(F f, double* __restrict__ partial, int64_t V, int C, int lanes) {
  constexpr int NQ = F::NQ;
  __shared__ double sm[kBlock * NQ * VEC];
  const int CG = C / VEC;
  const int tid = threadIdx.x;
  const int cg = tid % CG, vl = tid / CG;
  const int n = blockIdx.y;
  double acc[NQ][VEC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[q][j] = 0.0;
  if (vl < lanes) {
    for (int64_t v = (int64_t)blockIdx.x * lanes + vl; v < V; v += (int64_t)gridDim.x * lanes)
      f(((int64_t)n * V + v) * CG + cg, n, cg * VEC, acc, (int64_t)n * V + v, cg);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[(tid * NQ + q) * VEC + j] = acc[q][j];
  __syncthreads();
  if (vl == 0) {
    for (int l = 1; l < lanes; ++l) {
      const int t2 = l * CG + cg;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[q][j] += sm[(t2 * NQ + q) * VEC + j];
    }
    double* out = partial + (((int64_t)n * gridDim.x + blockIdx.x) * NQ) * C;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < VEC; ++j) out[(int64_t)q * C + cg * VEC + j] = acc[q][j];
  }
}

// mode 0: stats (mean, rstd) from (sum, sumsq); mode 1: plain sums (float) ; mode 2: means of the two sums.
// One wave per (n, c): lanes stride over the per-block partials, fp64 wave reduction (deterministic order).
__global__ void __launch_bounds__(kBlock)
k_channel_finalize(const double* __restrict__ partial, float* __restrict__ out, int NC, int C, int blocks, int NQ,
                   int64_t V, float eps, int mode) {
  const int i = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= NC) return;
  const int n = i / C, c = i - n * C;
  double s[2] = {0.0, 0.0};
  for (int b = lane; b < blocks; b += 64)
    for (int q = 0; q < NQ; ++q) s[q] += partial[(((int64_t)n * blocks + b) * NQ + q) * C + c];
  s[0] = cfun_wave_sum_d(s[0]);
  s[1] = cfun_wave_sum_d(s[1]);
  if (lane != 0) return;
  if (mode == 0) {
    const double mean = s[0] / (double)V;
    double var = s[1] / (double)V - mean * mean;
    if (var < 0.0) var = 0.0;
    out[(int64_t)i * 2] = (float)mean;
    out[(int64_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  } else if (mode == 1) {
    out[i] = (float)s[0];
  } else {
    out[(int64_t)i * 2] = (float)(s[0] / (double)V);
    out[(int64_t)i * 2 + 1] = (float)(s[1] / (double)V);
  }
}

// block i = (n, c): rows part[n][q][c][0..slots) for q = 0, 1; fixed summation order (thread-strided, then a tree)
__global__ void __launch_bounds__(kBlock)
k_stats_finalize_rows(const double* __restrict__ part, float* __restrict__ stats, int C, int slots, int64_t V, float eps) {
  __shared__ double sm[2][kBlock / 64];
  const int i = blockIdx.x, n = i / C, c = i - n * C, tid = threadIdx.x;
  const double* r0 = part + (((int64_t)n * 2 + 0) * C + c) * slots;
  const double* r1 = part + (((int64_t)n * 2 + 1) * C + c) * slots;
  double s0 = 0.0, s1 = 0.0;
  for (int b = tid; b < slots; b += kBlock) { s0 += r0[b]; s1 += r1[b]; }
  s0 = cfun_wave_sum_d(s0);
  s1 = cfun_wave_sum_d(s1);
  if ((tid & 63) == 0) { sm[0][tid >> 6] = s0; sm[1][tid >> 6] = s1; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) { a += sm[0][w]; b += sm[1][w]; }
    const double mean = a / (double)V;
    double var = b / (double)V - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(int64_t)i * 2] = (float)mean;
    stats[(int64_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// out[c] = sum_v g[v][c] in ONE launch for the short tensors of the detector (bias gradients of the FPN / RPN / classifier
// convs: <= 4 096 rows): a block owns 16 channels (64 contiguous bytes per voxel row) x 64 voxel lanes, fp64 sums, fixed
// order -- instead of k_channel_reduce + k_channel_finalize (two launches of ~5 us for ~2 us of work each)
__global__ void __launch_bounds__(kBlock)
k_channel_sum_direct(const float* __restrict__ g, float* __restrict__ out, int64_t V, int C) {
  __shared__ double sm[kBlock * 4];
  const int tid = threadIdx.x, q = tid & 3, vl = tid >> 2;      // 4 channel quads x 64 voxel lanes
  const int c0 = blockIdx.x * 16 + q * 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (c0 < C) {
    for (int64_t v = vl; v < V; v += kBlock / 4) {
      const float4 a = *reinterpret_cast<const float4*>(g + v * C + c0);
      acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[tid * 4 + j] = acc[j];
  __syncthreads();
  if (tid < 16 && blockIdx.x * 16 + tid < C) {      // thread = channel: the 64 lanes' sums in lane order
    const int qq = tid >> 2, j = tid & 3;
    double s = 0.0;
    for (int l = 0; l < kBlock / 4; ++l) s += sm[(l * 4 + qq) * 4 + j];
    out[blockIdx.x * 16 + tid] = (float)s;
  }
}

// most rows for the one-launch sum; CFUN_SUM_DIRECT_LOG2 overrides.  Its C / 16 workgroups walk all rows with 64 lanes
// each: measured (tools/bench_channel_sum.py, profiles/round4_channel_sum.txt) it matches reduce + finalize up to 4 096
// rows and loses beyond (16 384 rows: 34 vs 14 us, 65 536: 134 - 330 vs 14 - 19 us), whatever the channel count.
inline int64_t sum_direct_max() {
  static int64_t lim = -1;
  if (lim < 0) {
    const char* e = getenv("CFUN_SUM_DIRECT_LOG2");
    lim = (int64_t)1 << (e ? atoi(e) : 12);
  }
  return lim;
}

struct ReducePlan {
  int lanes, blocks;
};
inline int vec_of(int C) { return (C & 3) ? 1 : 4; }
inline ReducePlan reduce_plan(int N, int64_t V, int C) {
  ReducePlan r;
  const int CG = C / vec_of(C);
  r.lanes = kBlock / CG;
  int64_t want = (1024 + N - 1) / N;                       // ~4 blocks per CU in total
  int64_t maxb = (V + (int64_t)r.lanes * 16 - 1) / ((int64_t)r.lanes * 16);  // >= 16 voxels per thread
  if (maxb < 1) maxb = 1;
  r.blocks = (int)(want < maxb ? want : maxb);
  return r;
}
inline size_t reduce_ws(int N, int64_t V, int C, int NQ) {
  const ReducePlan r = reduce_plan(N, V, C);
  return cfun_align_up((size_t)N * r.blocks * NQ * C * sizeof(double), 256);
}

// thread = (voxel lane, channel group) of sample blockIdx.y: the group and its statistics are fixed per thread, the
// voxel index advances by a constant -- no per-element 64-bit division; y may be channel-strided (yg groups per row)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_instnorm_lrelu_fwd(const float* __restrict__ x, const float* __restrict__ stats, float* __restrict__ y, int64_t V, int C,
                     float slope, int64_t yg, int lanes) {
  const int CG = C / VEC;
  const int cg = threadIdx.x % CG, vl = threadIdx.x / CG;
  const int n = blockIdx.y;
  if (vl >= lanes) return;
  float mean[VEC], rstd[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mean[j] = stats[((int64_t)n * C + cg * VEC + j) * 2];
    rstd[j] = stats[((int64_t)n * C + cg * VEC + j) * 2 + 1];
  }
  for (int64_t v = (int64_t)blockIdx.x * lanes + vl; v < V; v += (int64_t)gridDim.x * lanes) {
    const int64_t vox = (int64_t)n * V + v;
    float a[VEC];
    Vec<VEC>::load(x, vox * CG + cg, a);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (a[j] - mean[j]) * rstd[j];
      a[j] = xh > 0.f ? xh : xh * slope;
    }
    Vec<VEC>::store(y, vox * yg + cg, a);
  }
}

// dx = rstd * (gn - mean(gn) - xhat * mean(gn*xhat))
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_instnorm_lrelu_bwd(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ means,
                     const float* __restrict__ dy, float* __restrict__ dx, int64_t V, int C, float slope, int64_t dyg,
                     int lanes, const float* __restrict__ add) {
  const int CG = C / VEC;
  const int cg = threadIdx.x % CG, vl = threadIdx.x / CG;
  const int n = blockIdx.y;
  if (vl >= lanes) return;
  float mean[VEC], rstd[VEC], m0[VEC], m1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int64_t k = ((int64_t)n * C + cg * VEC + j) * 2;
    mean[j] = stats[k]; rstd[j] = stats[k + 1]; m0[j] = means[k]; m1[j] = means[k + 1];
  }
  for (int64_t v = (int64_t)blockIdx.x * lanes + vl; v < V; v += (int64_t)gridDim.x * lanes) {
    const int64_t vox = (int64_t)n * V + v;
    float a[VEC], b[VEC], r[VEC];
    Vec<VEC>::load(x, vox * CG + cg, a);
    Vec<VEC>::load(dy, vox * dyg + cg, b);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (a[j] - mean[j]) * rstd[j];
      const float gn = xh > 0.f ? b[j] : b[j] * slope;
      r[j] = rstd[j] * (gn - m0[j] - xh * m1[j]);
    }
    if (add) {      // a second gradient of x (x also feeds a residual): summed here instead of by a pass of its own
      float e[VEC];
      Vec<VEC>::load(add, vox * CG + cg, e);
#pragma unroll
      for (int j = 0; j < VEC; ++j) r[j] += e[j];
    }
    Vec<VEC>::store(dx, vox * CG + cg, r);
  }
}

// ------------------------------------------------------------------ upsample backward, maxpool, halo
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_upsample2_bwd(const float* __restrict__ hi, float* __restrict__ lo, int64_t total, int D, int H, int W, int CG) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int cg = (int)(t % CG); t /= CG;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;
    const int z = (int)(t % D);
    const int64_t n = t / D;
    float sacc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sacc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * z + (k >> 2), yy = 2 * y + ((k >> 1) & 1), xx = 2 * x + (k & 1);
      float v[VEC];
      Vec<VEC>::load(hi, ((((int64_t)n * 2 * D + zz) * 2 * H + yy) * 2 * W + xx) * CG + cg, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) sacc[j] += v[j];
    }
    Vec<VEC>::store(lo, i, sacc);
  }
}

__global__ void __launch_bounds__(kBlock)
k_maxpool2_fwd(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int64_t total, int Do,
               int Ho, int Wo, int C) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); t /= Ho;
    const int zo = (int)(t % Do);
    const int64_t n = t / Do;
    float best = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // scan order (d,h,w), first maximum wins (torch max_pool3d)
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const float v = x[((((int64_t)n * 2 * Do + zz) * 2 * Ho + yy) * 2 * Wo + xx) * C + c];
      if (v > best || v != v) { best = v; bi = k; }
    }
    y[i] = best;
    idx[i] = (uint8_t)bi;
  }
}

__global__ void __launch_bounds__(kBlock)
k_maxpool2_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx, int64_t total,
               int Do, int Ho, int Wo, int C) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); t /= Ho;
    const int zo = (int)(t % Do);
    const int64_t n = t / Do;
    const int bi = idx[i];
    const float g = dy[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      dx[((((int64_t)n * 2 * Do + zz) * 2 * Ho + yy) * 2 * Wo + xx) * C + c] = (k == bi) ? g : 0.f;
    }
  }
}

// copy `planes` z-planes starting at z0 between a [N,D,H,W,C] tensor and a dense [N,planes,H,W,C] buffer
__global__ void __launch_bounds__(kBlock)
k_halo_copy(const float* __restrict__ src, float* __restrict__ dst, int64_t per_n, int64_t plane_elems, int D, int z0,
            int64_t total, int to_buf) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t n = i / per_n, r = i - n * per_n;
    const int64_t t = (n * D + z0) * plane_elems + r;
    if (to_buf) dst[i] = src[t]; else dst[t] = src[i];
  }
}

// ---------------------------------------------------------------------------------------- weight layouts
// All three conversions are batched 2-D transposes with padding:
//   dst[b*d_b + r*d_r + c*d_c] = r < R ? src[b*s_b + r*s_r + c*s_c] : 0      for r < Rpad, c < C
// with s_c == 1 and d_r == 1, so a 32x32 tile is read contiguously along c, turned in LDS and written contiguously
// along r (a thread-per-element gather ran at a tenth of the bandwidth: 1.4 ms per step for 48 MB of weights).
struct TransposeJob {
  float* dst;
  int R, Rpad, C;
  int64_t s_b, s_r, d_b, d_c;
  int tiles_r, tiles_c;
  unsigned blocks;
};

__device__ __forceinline__ void transpose_pad_tile(const float* __restrict__ src, float* __restrict__ dst, int R, int Rpad,
                                                   int C, int64_t s_b, int64_t s_r, int64_t d_b, int64_t d_c, int tiles_r,
                                                   int tiles_c, int t, float (*tile)[33]) {
  const int tc = t % tiles_c; t /= tiles_c;
  const int tr = t % tiles_r;
  const int b = t / tiles_r;
  const int r0 = tr * 32, c0 = tc * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(int64_t)b * s_b + (int64_t)r * s_r + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < Rpad && c < C) dst[(int64_t)b * d_b + (int64_t)c * d_c + r] = tile[tx][ty + 8 * k];
  }
}

// two transposes of the same source in one launch (forward + data-gradient weight layouts): blocks [0, a.blocks) run
// job a, the rest job b
__global__ void __launch_bounds__(kBlock)
k_transpose_pad2(const float* __restrict__ src, TransposeJob a, TransposeJob b) {
  __shared__ float tile[32][33];
  const bool first = blockIdx.x < a.blocks;
  const TransposeJob& j = first ? a : b;
  transpose_pad_tile(src, j.dst, j.R, j.Rpad, j.C, j.s_b, j.s_r, j.d_b, j.d_c, j.tiles_r, j.tiles_c,
                     (int)(first ? blockIdx.x : blockIdx.x - a.blocks), tile);
}

__global__ void __launch_bounds__(kBlock)
k_transpose_pad(const float* __restrict__ src, float* __restrict__ dst, int R, int Rpad, int C, int64_t s_b, int64_t s_r,
                int64_t d_b, int64_t d_c, int tiles_r, int tiles_c) {
  __shared__ float tile[32][33];
  int t = blockIdx.x;
  const int tc = t % tiles_c; t /= tiles_c;
  const int tr = t % tiles_r;
  const int b = t / tiles_r;
  const int r0 = tr * 32, c0 = tc * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(int64_t)b * s_b + (int64_t)r * s_r + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < Rpad && c < C) dst[(int64_t)b * d_b + (int64_t)c * d_c + r] = tile[tx][ty + 8 * k];
  }
}

inline int transpose_pad(const float* src, float* dst, int B, int R, int Rpad, int C, int64_t s_b, int64_t s_r, int64_t d_b,
                         int64_t d_c, hipStream_t st) {
  const int tiles_r = (Rpad + 31) / 32, tiles_c = (C + 31) / 32;
  const int64_t blocks = (int64_t)B * tiles_r * tiles_c;
  if (blocks <= 0) return CFUN_OK;
  if (blocks > 0x7fffffffLL) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_transpose_pad, dim3((unsigned)blocks), dim3(kBlock), 0, st, src, dst, R, Rpad, C, s_b, s_r, d_b, d_c,
                     tiles_r, tiles_c);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// ---------------------------------------------------------------------------------------- optimizer tail
// (model.py:1538-1545, 1641-1645: clip_grad_norm_(5.0) + SGD with momentum and weight decay)
constexpr int kNormBlocks = 512;   // partial sums of squares per call

__global__ void __launch_bounds__(kBlock)
k_sumsq_partials(const float* __restrict__ g, int64_t n, double* __restrict__ partials) {
  __shared__ double red[kBlock / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double v = (double)g[i];
    acc += v * v;
  }
  acc = cfun_wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) s += red[w];
    partials[blockIdx.x] = s;
  }
}

__global__ void k_norm_finalize(const double* __restrict__ partials, int count, float* __restrict__ norm) {
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 64) s += partials[i];
  s = cfun_wave_sum_d(s);
  if (threadIdx.x == 0) norm[0] = (float)sqrt(s);
}

// g' = g * min(1, max_norm / (norm + 1e-6)); d = g' + wd * p; m = first ? d : momentum * m + d; p -= lr * m
__global__ void __launch_bounds__(kBlock)
k_sgd_momentum_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, int64_t n, float lr,
                    float momentum, float wd, float max_norm, const float* __restrict__ norm, int first) {
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = __fdiv_rn(max_norm, __fadd_rn(norm[0], 1e-6f));
    coef = coef > 1.f ? 1.f : coef;
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    float d = __fmul_rn(g[i], coef);
    if (wd != 0.f) d = __fadd_rn(d, __fmul_rn(wd, p[i]));
    const float b = first ? d : __fadd_rn(__fmul_rn(momentum, m[i]), d);
    m[i] = b;
    p[i] = __fsub_rn(p[i], __fmul_rn(lr, b));
  }
}

}  // namespace

// ====================================================================== C ABI
// (mean, rstd) per (n, channel) from per-slot fp64 sums of x and x*x (conv epilogues, conv3d.hip).  slot_minor = 0:
// part[n][slot][2][C] (k_channel_finalize's layout: few slots); slot_minor = 1: part[n][2][C][slots] -- the tile kernels'
// thousands of slots, one 256-thread block per (n, channel) reads its two rows contiguously
int cfun_stats_finalize(const double* part, float* stats, int N, int slots, int C, int64_t V, float eps, int slot_minor,
                        hipStream_t st) {
  const int NC = N * C;
  if (NC <= 0) return CFUN_OK;
  if (slot_minor)
    hipLaunchKernelGGL(k_stats_finalize_rows, dim3((unsigned)NC), dim3(kBlock), 0, st, part, stats, C, slots, V, eps);
  else
    hipLaunchKernelGGL(k_channel_finalize, dim3((unsigned)((NC * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, part, stats, NC,
                       C, slots, 2, V, eps, 0);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

extern "C" {

int cfun_lrelu_fwd(const float* x, float* y, int64_t n, float slope, cfun_stream_t stream) {
  if (n <= 0) return CFUN_OK;
  const int64_t n4 = (cfun_aligned16(x) && cfun_aligned16(y)) ? n / 4 : 0;
  if (n4) hipLaunchKernelGGL(k_lrelu_fwd, dim3(ew_grid(n4)), dim3(kBlock), 0, cfun_st(stream), (const float4*)x, (float4*)y, n4, slope);
  for (int64_t from = n4 * 4; from < n; from += 1024)
    hipLaunchKernelGGL(k_lrelu_fwd_tail, dim3(1), dim3(1024), 0, cfun_st(stream), x, y, from, n, slope);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_lrelu_bwd(const float* x, const float* dy, float* dx, int64_t n, float slope, cfun_stream_t stream) {
  if (n <= 0) return CFUN_OK;
  const int64_t n4 = (cfun_aligned16(x) && cfun_aligned16(dy) && cfun_aligned16(dx)) ? n / 4 : 0;
  if (n4) hipLaunchKernelGGL(k_lrelu_bwd, dim3(ew_grid(n4)), dim3(kBlock), 0, cfun_st(stream), (const float4*)x, (const float4*)dy, (float4*)dx, n4, slope, (const float4*)nullptr);
  for (int64_t from = n4 * 4; from < n; from += 1024)
    hipLaunchKernelGGL(k_lrelu_bwd_tail, dim3(1), dim3(1024), 0, cfun_st(stream), x, dy, dx, from, n, slope);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_lrelu_fwd_strided(const float* x, float* y, int64_t nvox, int32_t C, int64_t x_stride, int64_t y_stride,
                           float slope, cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || (C & 3) || (x_stride & 3) || (y_stride & 3) || x_stride < C || y_stride < C) return CFUN_EINVAL;
  if (!cfun_aligned16(x) || !cfun_aligned16(y)) return CFUN_EALIGN;
  if (C / 4 > kBlock) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_lrelu_fwd_s, dim3(apply_blocks(nvox, kBlock / (C / 4))), dim3(kBlock), 0, cfun_st(stream),
                     (const float4*)x, (float4*)y, nvox, C / 4, x_stride / 4, y_stride / 4, slope);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_lrelu_bwd_add(const float* x, const float* dy, const float* add, float* dx, int64_t nvox, int32_t C,
                       int64_t dy_stride, float slope, cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || (C & 3) || (dy_stride & 3) || dy_stride < C) return CFUN_EINVAL;
  if (!cfun_aligned16(x) || !cfun_aligned16(dy) || !cfun_aligned16(dx) || (add && !cfun_aligned16(add))) return CFUN_EALIGN;
  if (C / 4 > kBlock) return CFUN_EINVAL;
  if (dy_stride == C)
    hipLaunchKernelGGL(k_lrelu_bwd, dim3(ew_grid(nvox * (C / 4))), dim3(kBlock), 0, cfun_st(stream), (const float4*)x,
                       (const float4*)dy, (float4*)dx, nvox * (C / 4), slope, (const float4*)add);
  else
    hipLaunchKernelGGL(k_lrelu_bwd_s, dim3(apply_blocks(nvox, kBlock / (C / 4))), dim3(kBlock), 0, cfun_st(stream),
                       (const float4*)x, (const float4*)dy, (float4*)dx, nvox, C / 4, dy_stride / 4, slope, (const float4*)add);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_lrelu_bwd_strided(const float* x, const float* dy, float* dx, int64_t nvox, int32_t C, int64_t dy_stride,
                           float slope, cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || (C & 3) || (dy_stride & 3) || dy_stride < C) return CFUN_EINVAL;
  if (!cfun_aligned16(x) || !cfun_aligned16(dy) || !cfun_aligned16(dx)) return CFUN_EALIGN;
  if (C / 4 > kBlock) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_lrelu_bwd_s, dim3(apply_blocks(nvox, kBlock / (C / 4))), dim3(kBlock), 0, cfun_st(stream),
                     (const float4*)x, (const float4*)dy, (float4*)dx, nvox, C / 4, dy_stride / 4, slope, (const float4*)nullptr);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_add(const float* a, const float* b, float* out, int64_t n, cfun_stream_t stream) {
  if (n <= 0) return CFUN_OK;
  const int64_t n4 = (cfun_aligned16(a) && cfun_aligned16(b) && cfun_aligned16(out)) ? n / 4 : 0;
  if (n4) hipLaunchKernelGGL(k_add, dim3(ew_grid(n4)), dim3(kBlock), 0, cfun_st(stream), (const float4*)a, (const float4*)b, (float4*)out, n4);
  for (int64_t from = n4 * 4; from < n; from += 1024)
    hipLaunchKernelGGL(k_add_tail, dim3(1), dim3(1024), 0, cfun_st(stream), a, b, out, from, n);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_act_bwd(const float* y, const float* dy, const float* scale, float* g, int64_t nvox, int32_t C,
                 int64_t vox_per_n, int32_t act, float slope, int32_t scale_mode, cfun_stream_t stream) {
  const int64_t total = nvox * C;
  if (total <= 0) return CFUN_OK;
  if (scale_mode && !scale) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_act_bwd, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), y, dy, scale, g, total, C,
                     vox_per_n > 0 ? vox_per_n : 1, act, slope, scale_mode);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

size_t cfun_channel_sum_workspace_bytes(int64_t nvox, int32_t C) {
  if (C <= 0 || C / vec_of(C) > kBlock) return 0;
  return reduce_ws(1, nvox, C, 1);
}

int cfun_channel_sum(const float* g, float* out, int64_t nvox, int32_t C, void* ws, size_t ws_bytes,
                     cfun_stream_t stream) {
  if (C <= 0 || C / vec_of(C) > kBlock) return CFUN_EINVAL;
  if (nvox <= 0) return (int)hipMemsetAsync(out, 0, C * sizeof(float), cfun_st(stream));
  if (vec_of(C) == 4 && !cfun_aligned16(g)) return CFUN_EALIGN;
  if (ws_bytes < reduce_ws(1, nvox, C, 1)) return CFUN_EWORKSPACE;
  if (vec_of(C) == 4 && nvox <= sum_direct_max()) {      // few rows: one launch (see k_channel_sum_direct)
    hipLaunchKernelGGL(k_channel_sum_direct, dim3((unsigned)((C + 15) / 16)), dim3(kBlock), 0, cfun_st(stream), g, out, nvox, C);
    CFUN_LAUNCH_CHECK();
    return CFUN_OK;
  }
  const ReducePlan r = reduce_plan(1, nvox, C);
  if (vec_of(C) == 4) {
    auto kern = k_channel_reduce<StatSum<4>, 4>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, 1), dim3(kBlock), 0, cfun_st(stream), StatSum<4>{g}, (double*)ws, nvox, C, r.lanes);
  } else {
    auto kern = k_channel_reduce<StatSum<1>, 1>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, 1), dim3(kBlock), 0, cfun_st(stream), StatSum<1>{g}, (double*)ws, nvox, C, r.lanes);
  }
  hipLaunchKernelGGL(k_channel_finalize, dim3((C * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, cfun_st(stream),
                     (const double*)ws, out, C, C, r.blocks, 1, nvox, 0.f, 1);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

size_t cfun_instnorm_workspace_bytes(int32_t N, int64_t V, int32_t C) {
  if (C <= 0 || C / vec_of(C) > kBlock || N <= 0 || V <= 0) return 0;
  // forward: partial sums; backward: partial sums + the [N,C,2] means
  return reduce_ws(N, V, C, 2) + cfun_align_up((size_t)N * C * 2 * sizeof(float), 256);
}

int cfun_instnorm_stats(const float* x, float* stats, int32_t N, int64_t V, int32_t C, float eps, void* ws,
                        size_t ws_bytes, cfun_stream_t stream) {
  if (N <= 0 || V <= 0) return CFUN_OK;
  if (C <= 0 || C / vec_of(C) > kBlock) return CFUN_EINVAL;
  if (vec_of(C) == 4 && !cfun_aligned16(x)) return CFUN_EALIGN;
  if (ws_bytes < reduce_ws(N, V, C, 2)) return CFUN_EWORKSPACE;
  const ReducePlan r = reduce_plan(N, V, C);
  if (vec_of(C) == 4) {
    auto kern = k_channel_reduce<StatSumSq<4>, 4>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, N), dim3(kBlock), 0, cfun_st(stream), StatSumSq<4>{x}, (double*)ws, V, C, r.lanes);
  } else {
    auto kern = k_channel_reduce<StatSumSq<1>, 1>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, N), dim3(kBlock), 0, cfun_st(stream), StatSumSq<1>{x}, (double*)ws, V, C, r.lanes);
  }
  hipLaunchKernelGGL(k_channel_finalize, dim3((N * C * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, cfun_st(stream),
                     (const double*)ws, stats, N * C, C, r.blocks, 2, V, eps, 0);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_instnorm_lrelu_fwd_strided(const float* x, const float* stats, float* y, int32_t N, int64_t V, int32_t C,
                                    int64_t y_stride, float slope, cfun_stream_t stream) {
  if (N <= 0 || V <= 0) return CFUN_OK;
  if (C <= 0 || y_stride < C || C / vec_of(C) > kBlock) return CFUN_EINVAL;
  const int vec = vec_of(C);
  if (y_stride != C && (vec != 4 || (y_stride & 3))) return CFUN_EINVAL;
  if (vec == 4 && (!cfun_aligned16(x) || !cfun_aligned16(y))) return CFUN_EALIGN;
  const int lanes = kBlock / (C / vec);
  const dim3 grid(apply_blocks(V, lanes), (unsigned)N);
  if (vec == 4) hipLaunchKernelGGL(k_instnorm_lrelu_fwd<4>, grid, dim3(kBlock), 0, cfun_st(stream), x, stats, y, V, C, slope, y_stride / 4, lanes);
  else hipLaunchKernelGGL(k_instnorm_lrelu_fwd<1>, grid, dim3(kBlock), 0, cfun_st(stream), x, stats, y, V, C, slope, (int64_t)C, lanes);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_instnorm_lrelu_fwd(const float* x, const float* stats, float* y, int32_t N, int64_t V, int32_t C,
                            float slope, cfun_stream_t stream) {
  return cfun_instnorm_lrelu_fwd_strided(x, stats, y, N, V, C, C, slope, stream);
}

int cfun_instnorm_lrelu_bwd(const float* x, const float* stats, const float* dy, float* dx, int32_t N, int64_t V,
                            int32_t C, float slope, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  return cfun_instnorm_lrelu_bwd_strided(x, stats, dy, dx, N, V, C, C, slope, ws, ws_bytes, stream);
}

int cfun_instnorm_lrelu_bwd_strided(const float* x, const float* stats, const float* dy, float* dx, int32_t N, int64_t V,
                                    int32_t C, int64_t dy_stride, float slope, void* ws, size_t ws_bytes,
                                    cfun_stream_t stream) {
  return cfun_instnorm_lrelu_bwd_add(x, stats, dy, nullptr, dx, N, V, C, dy_stride, slope, ws, ws_bytes, stream);
}

int cfun_instnorm_lrelu_bwd_add(const float* x, const float* stats, const float* dy, const float* add, float* dx, int32_t N,
                                int64_t V, int32_t C, int64_t dy_stride, float slope, void* ws, size_t ws_bytes,
                                cfun_stream_t stream) {
  if (N <= 0 || V <= 0) return CFUN_OK;
  if (add && vec_of(C) == 4 && !cfun_aligned16(add)) return CFUN_EALIGN;
  if (C <= 0 || C / vec_of(C) > kBlock || dy_stride < C) return CFUN_EINVAL;
  const int vec = vec_of(C);
  if (dy_stride != C && (vec != 4 || (dy_stride & 3))) return CFUN_EINVAL;
  const int64_t dyg = dy_stride / vec;
  if (vec == 4 && (!cfun_aligned16(x) || !cfun_aligned16(dy) || !cfun_aligned16(dx))) return CFUN_EALIGN;
  if (ws_bytes < cfun_instnorm_workspace_bytes(N, V, C)) return CFUN_EWORKSPACE;
  const ReducePlan r = reduce_plan(N, V, C);
  double* partial = (double*)ws;
  float* means = (float*)((char*)ws + reduce_ws(N, V, C, 2));
  if (vec == 4) {
    auto kern = k_channel_reduce<StatNormBwd<4>, 4>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, N), dim3(kBlock), 0, cfun_st(stream), StatNormBwd<4>{x, dy, stats, C, slope, dyg}, partial, V, C, r.lanes);
  } else {
    auto kern = k_channel_reduce<StatNormBwd<1>, 1>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, N), dim3(kBlock), 0, cfun_st(stream), StatNormBwd<1>{x, dy, stats, C, slope, dyg}, partial, V, C, r.lanes);
  }
  hipLaunchKernelGGL(k_channel_finalize, dim3((N * C * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, cfun_st(stream),
                     (const double*)partial, means, N * C, C, r.blocks, 2, V, 0.f, 2);
  const dim3 grid(apply_blocks(V, r.lanes), (unsigned)N);
  if (vec == 4) hipLaunchKernelGGL(k_instnorm_lrelu_bwd<4>, grid, dim3(kBlock), 0, cfun_st(stream), x, stats, (const float*)means, dy, dx, V, C, slope, dyg, r.lanes, add);
  else hipLaunchKernelGGL(k_instnorm_lrelu_bwd<1>, grid, dim3(kBlock), 0, cfun_st(stream), x, stats, (const float*)means, dy, dx, V, C, slope, dyg, r.lanes, add);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// The two halves of cfun_instnorm_lrelu_bwd, for a volume that is depth-sharded over several GPUs (cfun_amd.dist): the
// per-(n,c) means of gn and gn*xhat over the LOCAL voxels are written to `means` [N,C,2]; the caller combines them across
// ranks (weighted by the local voxel counts, one small all-reduce) and passes the global means to the apply half.
int cfun_instnorm_bwd_means(const float* x, const float* stats, const float* dy, float* means, int32_t N, int64_t V,
                            int32_t C, int64_t dy_stride, float slope, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (N <= 0 || V <= 0) return CFUN_OK;
  if (C <= 0 || C / vec_of(C) > kBlock || dy_stride < C) return CFUN_EINVAL;
  const int vec = vec_of(C);
  if (dy_stride != C && (vec != 4 || (dy_stride & 3))) return CFUN_EINVAL;
  const int64_t dyg = dy_stride / vec;
  if (vec == 4 && (!cfun_aligned16(x) || !cfun_aligned16(dy))) return CFUN_EALIGN;
  if (ws_bytes < reduce_ws(N, V, C, 2)) return CFUN_EWORKSPACE;
  const ReducePlan r = reduce_plan(N, V, C);
  double* partial = (double*)ws;
  if (vec == 4) {
    auto kern = k_channel_reduce<StatNormBwd<4>, 4>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, N), dim3(kBlock), 0, cfun_st(stream), StatNormBwd<4>{x, dy, stats, C, slope, dyg}, partial, V, C, r.lanes);
  } else {
    auto kern = k_channel_reduce<StatNormBwd<1>, 1>;
    hipLaunchKernelGGL(kern, dim3(r.blocks, N), dim3(kBlock), 0, cfun_st(stream), StatNormBwd<1>{x, dy, stats, C, slope, dyg}, partial, V, C, r.lanes);
  }
  hipLaunchKernelGGL(k_channel_finalize, dim3((N * C * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, cfun_st(stream),
                     (const double*)partial, means, N * C, C, r.blocks, 2, V, 0.f, 2);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_instnorm_lrelu_bwd_apply(const float* x, const float* stats, const float* means, const float* dy, float* dx,
                                  int32_t N, int64_t V, int32_t C, int64_t dy_stride, float slope, cfun_stream_t stream) {
  if (N <= 0 || V <= 0) return CFUN_OK;
  if (C <= 0 || C / vec_of(C) > kBlock || dy_stride < C) return CFUN_EINVAL;
  const int vec = vec_of(C);
  if (dy_stride != C && (vec != 4 || (dy_stride & 3))) return CFUN_EINVAL;
  const int64_t dyg = dy_stride / vec;
  if (vec == 4 && (!cfun_aligned16(x) || !cfun_aligned16(dy) || !cfun_aligned16(dx))) return CFUN_EALIGN;
  const ReducePlan r = reduce_plan(N, V, C);
  const dim3 grid(apply_blocks(V, r.lanes), (unsigned)N);
  if (vec == 4) hipLaunchKernelGGL(k_instnorm_lrelu_bwd<4>, grid, dim3(kBlock), 0, cfun_st(stream), x, stats, means, dy, dx, V, C, slope, dyg, r.lanes, (const float*)nullptr);
  else hipLaunchKernelGGL(k_instnorm_lrelu_bwd<1>, grid, dim3(kBlock), 0, cfun_st(stream), x, stats, means, dy, dx, V, C, slope, dyg, r.lanes, (const float*)nullptr);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_upsample2_bwd(const float* hi, float* lo, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C,
                       cfun_stream_t stream) {
  if (C <= 0) return CFUN_EINVAL;
  const int vec = vec_of(C);
  if (vec == 4 && (!cfun_aligned16(hi) || !cfun_aligned16(lo))) return CFUN_EALIGN;
  const int64_t total = (int64_t)N * D * H * W * (C / vec);
  if (total <= 0) return CFUN_OK;
  if (vec == 4) hipLaunchKernelGGL(k_upsample2_bwd<4>, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), hi, lo, total, D, H, W, C / vec);
  else hipLaunchKernelGGL(k_upsample2_bwd<1>, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), hi, lo, total, D, H, W, C / vec);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int32_t N, int32_t Do, int32_t Ho, int32_t Wo,
                      int32_t C, cfun_stream_t stream) {
  const int64_t total = (int64_t)N * Do * Ho * Wo * C;
  if (total <= 0) return CFUN_OK;
  hipLaunchKernelGGL(k_maxpool2_fwd, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), x, y, idx, total, Do, Ho, Wo, C);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_maxpool2_bwd(const float* dy, const uint8_t* idx, float* dx, int32_t N, int32_t Do, int32_t Ho, int32_t Wo,
                      int32_t C, cfun_stream_t stream) {
  const int64_t total = (int64_t)N * Do * Ho * Wo * C;
  if (total <= 0) return CFUN_OK;
  hipLaunchKernelGGL(k_maxpool2_bwd, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), dy, idx, dx, total, Do, Ho, Wo, C);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int32_t cfun_sumsq_partials_count(void) { return kNormBlocks; }

int cfun_sumsq_partials(const float* g, int64_t n, double* partials, cfun_stream_t stream) {
  if (n < 0) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_sumsq_partials, dim3(kNormBlocks), dim3(kBlock), 0, cfun_st(stream), g, n, partials);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_norm_finalize(const double* partials, int32_t count, float* norm, cfun_stream_t stream) {
  if (count < 0) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_norm_finalize, dim3(1), dim3(64), 0, cfun_st(stream), partials, count, norm);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_sgd_momentum_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float weight_decay,
                           float max_norm, const float* norm, int32_t first_step, cfun_stream_t stream) {
  if (n <= 0) return CFUN_OK;
  if (max_norm > 0.f && norm == nullptr) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_sgd_momentum_step, dim3(ew_grid(n)), dim3(kBlock), 0, cfun_st(stream), p, g, m, n, lr, momentum,
                     weight_decay, max_norm, norm, first_step);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_weight_pack(const float* w, float* wp, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream) {
  if (Co <= 0 || Ci <= 0 || T <= 0) return CFUN_EINVAL;
  const int CoP = (Co + 15) / 16 * 16;
  // batch = ci, r = co (padded to CoP), c = tap:  w[co][ci][t] -> wp[t][ci][co]
  return transpose_pad(w, wp, Ci, Co, CoP, T, /*s_b*/ T, /*s_r*/ (int64_t)Ci * T, /*d_b*/ CoP, /*d_c*/ (int64_t)Ci * CoP,
                       cfun_st(stream));
}

int cfun_weight_pack_both(const float* w, float* wp, float* wpT, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream) {
  if (Co <= 0 || Ci <= 0 || T <= 0) return CFUN_EINVAL;
  const int CoP = (Co + 15) / 16 * 16, CiP = (Ci + 15) / 16 * 16;
  TransposeJob a, b;
  // a: batch = ci, r = co (padded to CoP), c = tap:  w[co][ci][t] -> wp[t][ci][co]   (= cfun_weight_pack)
  a.dst = wp; a.R = Co; a.Rpad = CoP; a.C = T; a.s_b = T; a.s_r = (int64_t)Ci * T; a.d_b = CoP; a.d_c = (int64_t)Ci * CoP;
  a.tiles_r = (CoP + 31) / 32; a.tiles_c = (T + 31) / 32;
  // b: batch = co, r = ci (padded to CiP), c = tap:  w[co][ci][t] -> wpT[t][co][ci]  (= pack + pack_transpose)
  b.dst = wpT; b.R = Ci; b.Rpad = CiP; b.C = T; b.s_b = (int64_t)Ci * T; b.s_r = T; b.d_b = CiP; b.d_c = (int64_t)Co * CiP;
  b.tiles_r = (CiP + 31) / 32; b.tiles_c = (T + 31) / 32;
  const int64_t na = (int64_t)Ci * a.tiles_r * a.tiles_c, nb = (int64_t)Co * b.tiles_r * b.tiles_c;
  if (na + nb > 0x7fffffffLL) return CFUN_EINVAL;
  a.blocks = (unsigned)na; b.blocks = (unsigned)nb;
  hipLaunchKernelGGL(k_transpose_pad2, dim3((unsigned)(na + nb)), dim3(kBlock), 0, cfun_st(stream), w, a, b);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_weight_pack_transpose(const float* wp, float* wpT, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream) {
  if (Co <= 0 || Ci <= 0 || T <= 0) return CFUN_EINVAL;
  const int CoP = (Co + 15) / 16 * 16, CiP = (Ci + 15) / 16 * 16;
  // batch = tap, r = ci (padded to CiP), c = co:  wp[t][ci][co] -> wpT[t][co][ci]
  return transpose_pad(wp, wpT, T, Ci, CiP, Co, /*s_b*/ (int64_t)Ci * CoP, /*s_r*/ CoP, /*d_b*/ (int64_t)Co * CiP,
                       /*d_c*/ CiP, cfun_st(stream));
}

int cfun_weight_unpack(const float* dwp, float* dw, int32_t Co, int32_t Ci, int32_t T, cfun_stream_t stream) {
  if (Co <= 0 || Ci <= 0 || T <= 0) return CFUN_EINVAL;
  const int CoP = (Co + 15) / 16 * 16;
  // batch = ci, r = tap, c = co:  dwp[t][ci][co] -> dw[co][ci][t]
  return transpose_pad(dwp, dw, Ci, T, T, Co, /*s_b*/ CoP, /*s_r*/ (int64_t)Ci * CoP, /*d_b*/ T, /*d_c*/ (int64_t)Ci * T,
                       cfun_st(stream));
}

int cfun_halo_pack(const float* x, float* buf, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, int32_t z0,
                   int32_t planes, cfun_stream_t stream) {
  if (z0 < 0 || planes < 0 || z0 + planes > D) return CFUN_EINVAL;
  const int64_t plane = (int64_t)H * W * C, per_n = plane * planes, total = per_n * N;
  if (total <= 0) return CFUN_OK;
  hipLaunchKernelGGL(k_halo_copy, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), x, buf, per_n, plane, D, z0, total, 1);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_halo_unpack(const float* buf, float* x, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, int32_t z0,
                     int32_t planes, cfun_stream_t stream) {
  if (z0 < 0 || planes < 0 || z0 + planes > D) return CFUN_EINVAL;
  const int64_t plane = (int64_t)H * W * C, per_n = plane * planes, total = per_n * N;
  if (total <= 0) return CFUN_OK;
  hipLaunchKernelGGL(k_halo_copy, dim3(ew_grid(total)), dim3(kBlock), 0, cfun_st(stream), buf, x, per_n, plane, D, z0, total, 0);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // extern "C"
