// Classifier head as weight-streaming GEMMs (model.py:750-784): conv1 is Conv3d(C -> fc, kernel = pool size) on a
// pool-sized input, i.e. y[r][o] = sum_k x[r][k] * W[o][k] with K = C * pd * ph * pw (221 184 at the heart config), a
// handful of RoIs (R <= 12 in training, <= 64 at inference) and a 113 MB weight that is read exactly once per pass --
// HBM-bound, arithmetic intensity R/2 flop per byte.  The weight is streamed in the checkpoint's own OIDHW layout (rows
// of K contiguous floats per output channel): no packed copy, which would cost a second pass over the largest tensor of
// the model every step.
//
//   forward   k_fc_fwd:        K is split into chunks of 256 over the grid (864 workgroups at the heart config); a
//                              workgroup stages its [R x 256] slice of x once in LDS and its four waves walk all O/16
//                              output-channel tiles: 16-byte weight loads straight from HBM (16 per lane in flight) feed
//                              v_mfma_f32_16x16x4_f32 (exact fp32; the matrix pipe is ~40 % busy at the full HBM rate,
//                              a VALU kernel would be issue-bound for R = 64).  Per-chunk partial sums, then
//             k_fc_finish:     fixed-order sum over the chunks + scale / shift (bias, folded BatchNorm) + ReLU.
//   dW        k_fc_bwd_weight: dW[o][k] = sum_r g[r][o] * x[r][k]; one 16-byte store per (o, 4 k): a pure streaming write
//                              of the weight gradient (113 MB), x slice in LDS, g wave-uniform.
//   dx        k_fc_bwd_data:   dx[r][k] = sum_o g[r][o] * W[o][k]; every lane owns 4 consecutive k and streams the O
//                              weight rows (coalesced 1 KB per wave and row), R accumulators in registers.
// All sums have a fixed order (no atomics): results are bit-reproducible.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kKC = 256;        // K chunk of the forward / dW kernels
constexpr int kXS = kKC + 4;    // LDS row stride of the x slice (floats): 16-byte aligned rows, 4-bank skew per row
constexpr int kRT = 16;         // RoIs per register tile

// ------------------------------------------------------------------------------------------------ forward
template <int NR>   // r-tiles of 16 RoIs (R <= 16 * NR)
__global__ void __launch_bounds__(256)
k_fc_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ partial, int R, int K, int O) {
  CFUN_DYN_LDS(float, Xl);   // [16 * NR][kXS]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = blockIdx.x * kKC;
  const int kc = K - k0 < kKC ? K - k0 : kKC;   // multiple of 4
  // stage x[:, k0:k0+kc] (rows >= R and columns >= kc are zero)
  for (int it = tid; it < 16 * NR * (kKC / 4); it += 256) {
    const int r = it / (kKC / 4), c4 = (it % (kKC / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c4 < kc) v = *reinterpret_cast<const float4*>(x + (int64_t)r * K + k0 + c4);
    *reinterpret_cast<float4*>(Xl + r * kXS + c4) = v;
  }
  __syncthreads();
  const int lo = lane & 15, kq = lane >> 4;
  const int notile = (O + 15) / 16;
  for (int ot = wv; ot < notile; ot += 4) {
    const int o = ot * 16 + lo;
    const float* wrow = w + (int64_t)(o < O ? o : O - 1) * K + k0;   // (rows >= O: clamped reads, results never stored)
    f32x4 acc[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) acc[nr] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the weight is the only HBM stream: all 16 loads of the chunk are issued before the first MFMA
    float4 a[kKC / 16];
#pragma unroll
    for (int i = 0; i < kKC / 16; ++i) {
      const int off = i * 16 + 4 * kq;
      const bool ok = off < kc;
      a[i] = *reinterpret_cast<const float4*>(wrow + (ok ? off : 0));
      if (!ok) a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < kKC / 16; ++i) {
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        // lane (row lo, k-group kq) holds elements k = i*16 + 4*kq + j of its row for the j-th of four MFMAs: any
        // bijection between MFMA k-slots and the chunk's k indices is a valid summation order, as long as A and B agree
        const float4 b = *reinterpret_cast<const float4*>(Xl + (nr * 16 + lo) * kXS + i * 16 + 4 * kq);
        acc[nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[nr], 0, 0, 0);
        acc[nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[nr], 0, 0, 0);
        acc[nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[nr], 0, 0, 0);
        acc[nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[nr], 0, 0, 0);
      }
    }
    // D[o][r]: lane -> r = lo (+16 nr), o = ot*16 + 4*kq + reg  ->  partial[chunk][r][o..o+3]
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int r = nr * 16 + lo, ob = ot * 16 + 4 * kq;
      if (r >= R || ob >= O) continue;
      float* dst = partial + ((int64_t)blockIdx.x * R + r) * O + ob;
      if ((O & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[nr][0], acc[nr][1], acc[nr][2], acc[nr][3]);
      } else {     // the 2-way class head: O = 2
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ob + q < O) dst[q] = acc[nr][q];
      }
    }
  }
}

// y[r][o] = act(scale[o] * sum_chunk partial[chunk][r][o] + shift[o]).  The partials are a [nchunks x R*O] matrix summed
// along its rows: a block of 1024 threads owns 16 consecutive columns, thread (g, c) adds rows g, g + 64, ... (64-byte
// coalesced segments, ~14 independent loads per thread at the heart shapes), then the 64 row-group sums of a column are
// added in a fixed order -- deterministic, and short enough (a few microseconds) not to matter beside the weight stream.
__global__ void __launch_bounds__(1024)
k_fc_finish(const float* __restrict__ partial, int nchunks, const float* __restrict__ scale,
            const float* __restrict__ shift, float* __restrict__ y, int R, int O, int act) {
  __shared__ float red[64][17];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t ncol = (int64_t)R * O, col = (int64_t)blockIdx.x * 16 + c;
  float s = 0.f;
  if (col < ncol)
    for (int row = g; row < nchunks; row += 64) s += partial[(int64_t)row * ncol + col];
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && col < ncol) {
    float v = 0.f;
    for (int i = 0; i < 64; ++i) v += red[i][c];
    const int o = (int)(col % O);
    if (scale) v *= scale[o];
    if (shift) v += shift[o];
    y[col] = cfun_apply_act(v, act, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// grid = K chunks; a workgroup writes dW[:, k0..k0+255]: the x slice is staged once, wave w walks the output rows
// 4w .. 4w+3 (+16, +32, ...), lane -> 4 consecutive k, so every store is a full 1 KB row segment per wave
__global__ void __launch_bounds__(256)
k_fc_bwd_weight(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dw, int R, int K, int O) {
  CFUN_DYN_LDS(float, smem);
  const int OP = (O + 15) / 16 * 16;
  float* Xl = smem;                 // [R][kXS]
  float* Gl = smem + R * kXS;       // [R][OP]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = blockIdx.x * kKC;
  const int kc = K - k0 < kKC ? K - k0 : kKC;
  for (int it = tid; it < R * (kKC / 4); it += 256) {
    const int r = it / (kKC / 4), c4 = (it % (kKC / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < kc) v = *reinterpret_cast<const float4*>(x + (int64_t)r * K + k0 + c4);
    *reinterpret_cast<float4*>(Xl + r * kXS + c4) = v;
  }
  for (int it = tid; it < R * OP; it += 256) {
    const int r = it / OP, o = it % OP;
    Gl[it] = o < O ? g[(int64_t)r * O + o] : 0.f;
  }
  __syncthreads();
  const int c4 = lane * 4;
  for (int ob = 4 * wv; ob < O; ob += 16) {
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < R; ++r) {
      const float4 xv = *reinterpret_cast<const float4*>(Xl + r * kXS + c4);
      const float4 gv = *reinterpret_cast<const float4*>(Gl + r * OP + ob);     // wave-uniform (LDS broadcast)
      acc[0].x += gv.x * xv.x; acc[0].y += gv.x * xv.y; acc[0].z += gv.x * xv.z; acc[0].w += gv.x * xv.w;
      acc[1].x += gv.y * xv.x; acc[1].y += gv.y * xv.y; acc[1].z += gv.y * xv.z; acc[1].w += gv.y * xv.w;
      acc[2].x += gv.z * xv.x; acc[2].y += gv.z * xv.y; acc[2].z += gv.z * xv.z; acc[2].w += gv.z * xv.w;
      acc[3].x += gv.w * xv.x; acc[3].y += gv.w * xv.y; acc[3].z += gv.w * xv.z; acc[3].w += gv.w * xv.w;
    }
    if (c4 < kc) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ob + q < O) *reinterpret_cast<float4*>(dw + (int64_t)(ob + q) * K + k0 + c4) = acc[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------ data gradient
// grid = (K / 512, r-tiles of 16); lane owns 4 consecutive k; g tile [16][O] in LDS, read wave-uniformly.  128-thread
// blocks (432 of them at the heart shapes: every CU has work) with 16 weight rows in flight per lane
__global__ void __launch_bounds__(128)
k_fc_bwd_data(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, int R, int K, int O) {
  CFUN_DYN_LDS(float, Gl);   // [O][kRT]: g transposed so that one o's 16 RoIs are 4 float4
  const int tid = threadIdx.x;
  const int r0 = blockIdx.y * kRT;
  for (int it = tid; it < O * kRT; it += 128) {
    const int o = it / kRT, rr = it % kRT;
    Gl[it] = (r0 + rr < R) ? g[(int64_t)(r0 + rr) * O + o] : 0.f;
  }
  __syncthreads();
  const int64_t k = ((int64_t)blockIdx.x * 128 + tid) * 4;
  if (k >= K) return;
  float4 acc[kRT];
#pragma unroll
  for (int r = 0; r < kRT; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* wp = w + k;
  constexpr int U = 16;   // weight rows in flight per lane
  int o = 0;
  for (; o + U <= O; o += U) {
    float4 wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = *reinterpret_cast<const float4*>(wp + (int64_t)(o + u) * K);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int r4 = 0; r4 < kRT / 4; ++r4) {
        const float4 gv = *reinterpret_cast<const float4*>(Gl + (o + u) * kRT + 4 * r4);
        const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4& a = acc[4 * r4 + j];
          a.x += gs[j] * wv[u].x; a.y += gs[j] * wv[u].y; a.z += gs[j] * wv[u].z; a.w += gs[j] * wv[u].w;
        }
      }
    }
  }
  for (; o < O; ++o) {
    const float4 w1 = *reinterpret_cast<const float4*>(wp + (int64_t)o * K);
#pragma unroll
    for (int r = 0; r < kRT; ++r) {
      const float gs = Gl[o * kRT + r];
      acc[r].x += gs * w1.x; acc[r].y += gs * w1.y; acc[r].z += gs * w1.z; acc[r].w += gs * w1.w;
    }
  }
#pragma unroll
  for (int r = 0; r < kRT; ++r)
    if (r0 + r < R) *reinterpret_cast<float4*>(dx + (int64_t)(r0 + r) * K + k) = acc[r];
}

inline int fc_chunks(int K) { return (K + kKC - 1) / kKC; }

}  // namespace

// dynamic LDS a workgroup may use on this device (160 KB on MI355X); queried once
static size_t fc_lds_limit() {
  static size_t lim = 0;
  if (!lim) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
      v = 64 * 1024;
    lim = (size_t)v;
  }
  return lim;
}

extern "C" size_t cfun_fc_workspace_bytes(int32_t R, int32_t K, int32_t O) {
  if (R <= 0 || K <= 0 || O <= 0) return 0;
  return (size_t)fc_chunks(K) * R * O * sizeof(float);
}

extern "C" int cfun_fc_fwd(const float* x, const float* w, const float* scale, const float* shift, float* y, int32_t R,
                           int32_t K, int32_t O, int32_t act, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (R < 0 || K <= 0 || O <= 0 || (K & 3) || R > 64) return CFUN_EINVAL;
  if (R == 0) return CFUN_OK;
  if (!x || !w || !y || !cfun_aligned16(x) || !cfun_aligned16(w) || !cfun_aligned16(ws)) return CFUN_EINVAL;
  if (ws_bytes < cfun_fc_workspace_bytes(R, K, O)) return CFUN_EINVAL;
  if (act != CFUN_ACT_NONE && act != CFUN_ACT_RELU) return CFUN_EINVAL;
  hipStream_t st = cfun_st(stream);
  const int nch = fc_chunks(K), nr = (R + 15) / 16;
  const size_t lds = (size_t)16 * nr * kXS * sizeof(float);
  if (lds > fc_lds_limit()) return CFUN_EINVAL;
  auto kern = nr == 1 ? k_fc_fwd<1> : nr == 2 ? k_fc_fwd<2> : nr == 3 ? k_fc_fwd<3> : k_fc_fwd<4>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(nch), dim3(256), lds, st, x, w, (float*)ws, R, K, O);
  CFUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fc_finish, dim3((unsigned)(((int64_t)R * O + 15) / 16)), dim3(1024), 0, st, (const float*)ws, nch, scale,
                     shift, y, R, O, act);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// k_fc_bwd_weight keeps its x slice and all of g in LDS: R * (kXS + O rounded to 16) floats.  The largest R that fits
// (>= 1 for every O the kernels accept); callers with more rows split them and add the partial gradients.
extern "C" int32_t cfun_fc_bwd_weight_max_rows(int32_t O) {
  if (O <= 0) return 0;
  const size_t per_row = (size_t)(kXS + (O + 15) / 16 * 16) * sizeof(float);
  const size_t r = fc_lds_limit() / per_row;
  return (int32_t)(r > 64 ? 64 : r);
}

extern "C" int cfun_fc_bwd_weight(const float* x, const float* g, float* dw, int32_t R, int32_t K, int32_t O,
                                  cfun_stream_t stream) {
  if (R < 0 || K <= 0 || O <= 0 || (K & 3) || R > 64) return CFUN_EINVAL;
  if (R > cfun_fc_bwd_weight_max_rows(O)) return CFUN_EINVAL;      // (not a raw hipError from hipFuncSetAttribute)
  if (!dw || !cfun_aligned16(dw)) return CFUN_EINVAL;
  hipStream_t st = cfun_st(stream);
  if (R == 0) return (int)hipMemsetAsync(dw, 0, (size_t)O * K * sizeof(float), st);
  if (!x || !g || !cfun_aligned16(x)) return CFUN_EINVAL;
  const size_t lds = (size_t)(R * kXS + R * ((O + 15) / 16 * 16)) * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fc_bwd_weight),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_fc_bwd_weight, dim3(fc_chunks(K)), dim3(256), lds, st, x, g, dw, R, K, O);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

extern "C" int cfun_fc_bwd_data(const float* g, const float* w, float* dx, int32_t R, int32_t K, int32_t O,
                                cfun_stream_t stream) {
  if (R < 0 || K <= 0 || O <= 0 || (K & 3) || R > 64) return CFUN_EINVAL;
  if (R == 0) return CFUN_OK;
  if (!g || !w || !dx || !cfun_aligned16(w) || !cfun_aligned16(dx)) return CFUN_EINVAL;
  const size_t lds = (size_t)O * kRT * sizeof(float);
  if (lds > fc_lds_limit()) return CFUN_EINVAL;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fc_bwd_data),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_fc_bwd_data, dim3((K / 4 + 127) / 128, (R + kRT - 1) / kRT), dim3(128), lds, cfun_st(stream), g, w,
                     dx, R, K, O);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}
