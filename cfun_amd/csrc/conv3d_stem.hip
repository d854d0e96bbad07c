// C_in = 1 stem convolutions (conv3d_c1_1 of the U-Net, mask_branch.py:23; the P3D stem backbone.py:123-126 and its
// LiTS variant): the HBM-bound 3-D convs of the path (AI 13-49 flop/B) -- the output is 16-24x
// larger than the input, so the kernel is a coalesced streaming WRITE fed from an LDS-staged input tile:
//   * a 256-thread workgroup owns 2(z) x 4(y) x 32(x) output voxels; the input halo tile (C = 1) is loaded once,
//     coalesced along x, zero padding = the bounds check;
//   * every thread keeps all C_out accumulators of ONE voxel in registers; taps are LDS reads at compile-time
//     offsets (conflict-free along x), weights are wave-uniform scalar loads (SGPR operands of v_fmac);
//   * fused epilogue: folded-BN / dropout scale, shift, ReLU / LeakyReLU; each lane stores its voxel's C_out
//     contiguous floats as float4s, so a wave writes one contiguous 64*C_out*4-byte span.
// The generic direct kernel computed 8 channels per thread in 3 passes over the input with 32-byte scattered
// stores (0.35 ms per stem = 0.85 TB/s); this one is write-bound.
#include "conv3d_mfma.h"
#include <stdlib.h>

#include <utility>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));


template <int KD, int KH, int KW, int S, int CO>
struct StemTile {
  static constexpr int TZ = 2, TY = 4, TX = 32;
  static constexpr int IZ = (TZ - 1) * S + KD, IY = (TY - 1) * S + KH, IX = (TX - 1) * S + KW;
  static constexpr int IVOX = IZ * IY * IX;
};

template <int KD, int KH, int KW, int S, int CO>
__global__ void __launch_bounds__(256)
k_conv_stem(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
            const float* __restrict__ shift, float* __restrict__ y, CfunConv3dParams p, int ntz, int nty, int ntx) {
  using T = StemTile<KD, KH, KW, S, CO>;
  __shared__ float tile[T::IVOX];
  unsigned b = blockIdx.x;
  const int tx = b % ntx; b /= ntx;
  const int ty = b % nty; b /= nty;
  const int tz = b % ntz;
  const int n = b / ntz;
  const int z0 = tz * T::TZ, y0 = ty * T::TY, x0 = tx * T::TX;
  const int iz0 = z0 * S - p.pd, iy0 = y0 * S - p.ph, ix0 = x0 * S - p.pw;
  const float* xn = x + (int64_t)n * p.Di * p.Hi * p.Wi;
  // all of the thread's halo loads are issued before the first LDS write (a rolled loop serialises one HBM round trip
  // per iteration: 4 x ~1.5 us per workgroup was half of the kernel's time)
  constexpr int NL = (T::IVOX + 255) / 256;
  float stage[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int i = threadIdx.x + k * 256;
    const int lx = i % T::IX, ly = (i / T::IX) % T::IY, lz = i / (T::IX * T::IY);
    const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
    stage[k] = 0.f;
    if (i < T::IVOX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi)
      stage[k] = xn[((int64_t)gz * p.Hi + gy) * p.Wi + gx];
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < T::IVOX) tile[i] = stage[k];
  }
  __syncthreads();
  const int lx = threadIdx.x % T::TX, ly = (threadIdx.x / T::TX) % T::TY, lz = threadIdx.x / (T::TX * T::TY);
  const float* t0 = tile + ((lz * S) * T::IY + ly * S) * T::IX + lx * S;
  // accumulators as float2 pairs: v_pk_fma_f32 (2 FMAs per lane per issue -- the only way to the 157 TFLOP/s fp32 VALU
  // rate; plain v_fmac tops out at half of it, which made this kernel VALU- rather than write-bound)
  f32x2 acc[CO / 2];
#pragma unroll
  for (int j = 0; j < CO / 2; ++j) acc[j] = f32x2{0.f, 0.f};
  // one (dz, dy) tap row per iteration, NOT unrolled: fully unrolled, hipcc hoists all KD*KH*KW*CO scalar weight loads
  // to the top and spills them through v_writelane / v_readlane (12 extra instructions per FMA)
#pragma unroll 1
  for (int r = 0; r < KD * KH; ++r) {
    const int dz = r / KH, dy = r - dz * KH;
    const float* trow = t0 + (dz * T::IY + dy) * T::IX;
    const float* wrow = wp + (int64_t)r * KW * p.CoP;               // wave-uniform: scalar loads
#pragma unroll
    for (int dx = 0; dx < KW; ++dx) {
      const float xs = trow[dx];
      const f32x2 xv = {xs, xs};
      const f32x2* w = reinterpret_cast<const f32x2*>(wrow + dx * p.CoP);   // CoP % 4 == 0: 8-byte aligned pairs
#pragma unroll
      for (int j = 0; j < CO / 2; ++j) acc[j] = __builtin_elementwise_fma(xv, w[j], acc[j]);
    }
  }
  // epilogue in registers, then through LDS so that the stores are fully coalesced: an output row of the tile is
  // TX*CO contiguous floats; lanes write consecutive float4s of it instead of CO floats at a CO*4-byte lane stride
  __shared__ float outt[T::TZ * T::TY * T::TX * CO];
#pragma unroll
  for (int q = 0; q < CO / 4; ++q) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * q + e;
      float v = acc[j >> 1][j & 1];
      if (p.scale_mode == 1) v *= scale[j];
      else if (p.scale_mode == 2) v *= scale[n * CO + j];
      if (p.has_shift) v += shift[j];
      o[e] = cfun_apply_act(v, p.act, p.slope);
    }
    *reinterpret_cast<float4*>(outt + threadIdx.x * CO + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);   // conflict-free
  }
  __syncthreads();
  // wave w stores the (z, y) rows 2w and 2w+1 of the tile: ROW4 consecutive float4s each, 64 per instruction --
  // row-uniform index arithmetic (the flat "i / ROW4, i % ROW4" loop cost as many VALU cycles as a third of the FMAs)
  constexpr int ROW4 = T::TX * CO / 4;                      // float4s per (z, y) output row of the tile
  constexpr int RPW = T::TZ * T::TY / 4;                    // rows per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr;
    const int oz = z0 + row / T::TY, oy = y0 + row % T::TY;
    if (oz >= p.Do || oy >= p.Ho) continue;                 // wave-uniform
    float* yrow = y + ((((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + x0) * CO;
    const float* orow = outt + row * T::TX * CO;
    const int nq = (p.Wo - x0 < T::TX ? p.Wo - x0 : T::TX) * (CO / 4);     // float4s of the row inside the volume
#pragma unroll
    for (int q = lane; q < ROW4; q += 64)
      if (q < nq) *reinterpret_cast<float4*>(yrow + q * 4) = *reinterpret_cast<const float4*>(orow + q * 4);
  }
}

// The 3 x 20 weights of one (dz, dy) tap row of the 1 -> 20 conv: three 20-float groups, 32 floats apart in the packed
// [tap][1][CoP = 32] weights, in 60 SGPRs at once.  Written by hand because hipcc reuses ONE 20-register window for the three
// dx groups: load -> s_waitcnt -> 20 FMAs -> load -> ... exposes the scalar-cache latency three times per row (counters,
// profiles/round4_pmc_stem.txt: 1 007 VALU instructions per wave keep the SIMDs 59 % busy, the waves wait 41 % of their
// time); with distinct destination registers the six loads go out together and one wait covers the row.
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x16s __attribute__((ext_vector_type(16)));
struct StemRow60 {
  f32x16s a[3];
  f32x4s b[3];
};
__device__ __forceinline__ void stem_load_row60(StemRow60& w, const float* wrow) {      // wrow: wave-uniform, CoP = 32
#ifdef CFUN_HIP_EMULATION
  for (int dx = 0; dx < 3; ++dx) {
    for (int i = 0; i < 16; ++i) w.a[dx][i] = wrow[dx * 32 + i];
    for (int i = 0; i < 4; ++i) w.b[dx][i] = wrow[dx * 32 + 16 + i];
  }
#else
  asm volatile(
      "s_load_dwordx16 %0, %6, 0x0\n\t"
      "s_load_dwordx4 %1, %6, 0x40\n\t"
      "s_load_dwordx16 %2, %6, 0x80\n\t"
      "s_load_dwordx4 %3, %6, 0xc0\n\t"
      "s_load_dwordx16 %4, %6, 0x100\n\t"
      "s_load_dwordx4 %5, %6, 0x140\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(w.a[0]), "=&s"(w.b[0]), "=&s"(w.a[1]), "=&s"(w.b[1]), "=&s"(w.a[2]), "=&s"(w.b[2])
      : "s"(wrow)
      : "memory");
#endif
}
template <int DX, int F>      // floats F, F + 1 of group DX (F even)
__device__ __forceinline__ f32x2 stem_row_pair(const StemRow60& w) {
  if constexpr (F < 16) return f32x2{w.a[DX][F], w.a[DX][F + 1]};
  else return f32x2{w.b[DX][F - 16], w.b[DX][F - 15]};
}
template <int... I, class Fn>
__device__ __forceinline__ void stem_static_for(std::integer_sequence<int, I...>, Fn&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

// ---- 3x3x3 stride-1 variant with ZPT output voxels (a z-column) per thread.  Each (dz, dy) tap row costs a wave one
// scalar-memory round trip for its 3 x CO weights; with one voxel per thread the 9 dependent round trips are the
// kernel's floor (measured: 0.050 ms of 0.100 with the FMAs and the stores removed).  A thread that owns ZPT voxels
// spends the same round trips on ZPT times the arithmetic and output bytes, and re-uses the halo planes between its
// voxels: workgroup tile = 2*ZPT (z) x 4 (y) x 32 (x).
template <int CO, int ZPT>
__global__ void __launch_bounds__(256)
k_conv_stem333z(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
                const float* __restrict__ shift, float* __restrict__ y, CfunConv3dParams p, int ntz, int nty, int ntx) {
  constexpr int TZ = 2 * ZPT, TY = 4, TX = 32, IZ = TZ + 2, IY = TY + 2, IX = TX + 2, IVOX = IZ * IY * IX;
  __shared__ float tile[IVOX];
  __shared__ float outt[2 * TY * TX * CO];
  unsigned b = blockIdx.x;
  const int tx = b % ntx; b /= ntx;
  const int ty = b % nty; b /= nty;
  const int tz = b % ntz;
  const int n = b / ntz;
  const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
  const int iz0 = z0 - p.pd, iy0 = y0 - p.ph, ix0 = x0 - p.pw;
  const float* xn = x + (int64_t)n * p.Di * p.Hi * p.Wi;
  constexpr int NL = (IVOX + 255) / 256;
  float stage[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int i = threadIdx.x + k * 256;
    const int lx = i % IX, ly = (i / IX) % IY, lz = i / (IX * IY);
    const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
    stage[k] = 0.f;
    if (i < IVOX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi)
      stage[k] = xn[((int64_t)gz * p.Hi + gy) * p.Wi + gx];
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < IVOX) tile[i] = stage[k];
  }
  __syncthreads();
  const int lx = threadIdx.x % TX, ly = (threadIdx.x / TX) % TY, lzg = threadIdx.x / (TX * TY);   // lzg: 0 | 1
  const float* t0 = tile + ((lzg * ZPT) * IY + ly) * IX + lx;
  f32x2 acc[ZPT][CO / 2];
#pragma unroll
  for (int zi = 0; zi < ZPT; ++zi)
#pragma unroll
    for (int j = 0; j < CO / 2; ++j) acc[zi][j] = f32x2{0.f, 0.f};
#pragma unroll 1
  for (int r = 0; r < 9; ++r) {          // (dz, dy) tap rows: rolled, see k_conv_stem
    const int dz = r / 3, dy = r - dz * 3;
    const float* trow = t0 + (dz * IY + dy) * IX;
    const float* wrow = wp + (int64_t)r * 3 * p.CoP;               // wave-uniform: scalar loads
#ifndef CFUN_STEM_NO_ROW60      /* build-time A/B only */
    if constexpr (CO == 20) {
      float xs[3][ZPT];                                            // the row's LDS reads are in flight under the weight loads
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int zi = 0; zi < ZPT; ++zi) xs[dx][zi] = trow[zi * IY * IX + dx];
      StemRow60 w;
      stem_load_row60(w, wrow);
      stem_static_for(std::make_integer_sequence<int, 30>{}, [&](auto jj) {
        constexpr int J = decltype(jj)::value, dx = J / 10, j = J % 10;
        const f32x2 wv = stem_row_pair<dx, 2 * j>(w);
#pragma unroll
        for (int zi = 0; zi < ZPT; ++zi) {
          const f32x2 xv = {xs[dx][zi], xs[dx][zi]};
          acc[zi][j] = __builtin_elementwise_fma(xv, wv, acc[zi][j]);
        }
      });
    } else
#endif
    {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const f32x2* w = reinterpret_cast<const f32x2*>(wrow + dx * p.CoP);
#pragma unroll
        for (int zi = 0; zi < ZPT; ++zi) {
          const float xs = trow[zi * IY * IX + dx];
          const f32x2 xv = {xs, xs};
#pragma unroll
          for (int j = 0; j < CO / 2; ++j) acc[zi][j] = __builtin_elementwise_fma(xv, w[j], acc[zi][j]);
        }
      }
    }
  }
  constexpr int ROW4 = TX * CO / 4;      // float4s per (z, y) output row of the tile
  constexpr int RPW = 2 * TY / 4;        // rows per wave and pass
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int zi = 0; zi < ZPT; ++zi) {     // pass zi: the two z-planes lzg*ZPT + zi through the 2 x TY x TX x CO LDS buffer
    if (zi) __syncthreads();
#pragma unroll
    for (int q = 0; q < CO / 4; ++q) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = 4 * q + e;
        float v = acc[zi][j >> 1][j & 1];
        if (p.scale_mode == 1) v *= scale[j];
        else if (p.scale_mode == 2) v *= scale[n * CO + j];
        if (p.has_shift) v += shift[j];
        o[e] = cfun_apply_act(v, p.act, p.slope);
      }
      *reinterpret_cast<float4*>(outt + threadIdx.x * CO + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wave * RPW + rr;                       // = lzg * TY + ly of the producing threads
      const int oz = z0 + (row / TY) * ZPT + zi, oy = y0 + row % TY;
      if (oz >= p.Do || oy >= p.Ho) continue;                // wave-uniform
      float* yrow = y + ((((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + x0) * CO;
      const float* orow = outt + row * TX * CO;
      const int nq = (p.Wo - x0 < TX ? p.Wo - x0 : TX) * (CO / 4);
#pragma unroll
      for (int q = lane; q < ROW4; q += 64)
        if (q < nq) *reinterpret_cast<float4*>(yrow + q * 4) = *reinterpret_cast<const float4*>(orow + q * 4);
    }
  }
}

// measured on MI355X, conv3d_c1_1 at 4 x 96^3 (tools/bench_layers.py, profiles/round2_ab_layers.log): one voxel per
// thread 0.093 ms = 3.2 TB/s, ZPT = 2 0.081 ms = 3.7 TB/s, ZPT = 4 0.085 ms (3 456 workgroups: the tail shows)
inline int stem_zpt() {      // tuning knob: CFUN_STEM_ZPT = 1 (one voxel per thread) | 2 (default) | 4
  static int v = -1;
  if (v < 0) { const char* e = getenv("CFUN_STEM_ZPT"); v = e ? atoi(e) : 2; if (v != 1 && v != 4) v = 2; }
  return v;
}

template <int CO, int ZPT>
int launch_stem333z(const float* x, const float* wp, const float* scale, const float* shift, float* y,
                    const CfunConv3dParams& p, hipStream_t st) {
  const int ntz = (p.Do + 2 * ZPT - 1) / (2 * ZPT), nty = (p.Ho + 3) / 4, ntx = (p.Wo + 31) / 32;
  const int64_t blocks = (int64_t)p.N * ntz * nty * ntx;
  if (blocks <= 0) return CFUN_OK;
  if (blocks > 0x7fffffffLL) return CFUN_EINVAL;
  hipLaunchKernelGGL((k_conv_stem333z<CO, ZPT>), dim3((unsigned)blocks), dim3(256), 0, st, x, wp, scale, shift, y, p, ntz,
                     nty, ntx);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

template <int KD, int KH, int KW, int S, int CO>
int launch_stem(const float* x, const float* wp, const float* scale, const float* shift, float* y,
                const CfunConv3dParams& p, hipStream_t st) {
  using T = StemTile<KD, KH, KW, S, CO>;
  const int ntz = (p.Do + T::TZ - 1) / T::TZ, nty = (p.Ho + T::TY - 1) / T::TY, ntx = (p.Wo + T::TX - 1) / T::TX;
  const int64_t blocks = (int64_t)p.N * ntz * nty * ntx;
  if (blocks <= 0) return CFUN_OK;
  if (blocks > 0x7fffffffLL) return CFUN_EINVAL;
  hipLaunchKernelGGL((k_conv_stem<KD, KH, KW, S, CO>), dim3((unsigned)blocks), dim3(256), 0, st, x, wp, scale, shift, y, p,
                     ntz, nty, ntx);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // namespace

// 1 when the stem kernel covers this forward call (conv3d.hip asks before falling back to the generic direct kernel)
int cfun_conv_stem_supported(const CfunConv3dParams* p) {
  if (p->Ci != 1 || p->up2 || p->d2s || p->res_mode) return 0;
  const int k = p->kd * 100 + p->kh * 10 + p->kw;
  return (k == 333 && p->stride == 1 && p->Co == 20) || (k == 377 && p->stride == 2 && p->Co == 16) ||
         (k == 577 && p->stride == 2 && p->Co == 24);
}

int cfun_conv_stem_fwd(const float* x, const float* wp, const float* scale, const float* shift, float* y,
                       const CfunConv3dParams* p, hipStream_t st) {
  const int k = p->kd * 100 + p->kh * 10 + p->kw;
  // (the z-column kernels read their weight rows with hand-written loads that assume CoP = 32: stem_load_row60)
  if (k == 333 && stem_zpt() == 4 && p->CoP == 32) return launch_stem333z<20, 4>(x, wp, scale, shift, y, *p, st);
  if (k == 333 && stem_zpt() == 2 && p->CoP == 32) return launch_stem333z<20, 2>(x, wp, scale, shift, y, *p, st);
  if (k == 333) return launch_stem<3, 3, 3, 1, 20>(x, wp, scale, shift, y, *p, st);
  if (k == 377) return launch_stem<3, 7, 7, 2, 16>(x, wp, scale, shift, y, *p, st);
  return launch_stem<5, 7, 7, 2, 24>(x, wp, scale, shift, y, *p, st);
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1x1 convs with 8 output channels on large volumes (conv3d_l4 40->8 @ 4x96^3, ds3 80->8, the fused RPN heads
// 256->8): pure streaming -- read C_in floats per voxel, write 8.  One thread per voxel keeps its 8 accumulators,
// reads its row as float4s and takes the weights as wave-uniform scalar loads; the MFMA tile kernel spends a
// barrier per 4-channel chunk on 4 KB of input here and runs at 1.5 TB/s.
namespace {

// IN: the input prologue (CfunConvFusion.in_stats / in_act) -- x is read as in_act((x - mean) * rstd); the (mean, rstd)
// table [N][Ci][2] of the whole launch sits in LDS (a thread's sample is not wave-uniform at sample boundaries)
template <int CO, bool IN>
__global__ void __launch_bounds__(256)
k_conv_pointwise(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
                 const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
                 CfunConv3dParams p, int64_t total, const float* __restrict__ in_stats, int in_act, float in_slope) {
  CFUN_DYN_LDS(float4, stab);       // IN: [N * Ci / 4][2] float4 rows (mean0, rstd0, mean1, rstd1 | mean2, ...)
  if constexpr (IN) {
    const int rows = p.N * (p.Ci >> 2) * 2;
    for (int i = threadIdx.x; i < rows; i += 256)
      stab[i] = in_stats ? reinterpret_cast<const float4*>(in_stats)[i] : make_float4(0.f, 1.f, 0.f, 1.f);
    __syncthreads();
  }
  const int64_t per_n_in = (int64_t)p.Do * p.Ho * p.Wo;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const float4* xr = reinterpret_cast<const float4*>(x + v * p.Ci);
    const float4* srow = stab + (IN ? (v / per_n_in) * (p.Ci >> 2) * 2 : 0);
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = 0.f;
#pragma unroll 2
    for (int c = 0; c < p.Ci; c += 4) {
      float4 xv = xr[c >> 2];
      if constexpr (IN) xv = cfun_mfma::norm_act_in4(xv, srow[(c >> 2) * 2], srow[(c >> 2) * 2 + 1], in_act, in_slope);
      const float* w = wp + (int64_t)c * p.CoP;          // wave-uniform: scalar loads
#pragma unroll
      for (int j = 0; j < CO; ++j) {
        acc[j] = fmaf(xv.x, w[j], acc[j]);
        acc[j] = fmaf(xv.y, w[p.CoP + j], acc[j]);
        acc[j] = fmaf(xv.z, w[2 * p.CoP + j], acc[j]);
        acc[j] = fmaf(xv.w, w[3 * p.CoP + j], acc[j]);
      }
    }
    const int64_t per_n = (int64_t)p.Do * p.Ho * p.Wo;
    const int n = (int)(v / per_n);
    int64_t ridx = v;
    if (p.res_mode && p.res_up2) {
      int64_t t = v - n * per_n;
      const int xo = (int)(t % p.Wo); t /= p.Wo;
      const int yo = (int)(t % p.Ho);
      const int zo = (int)(t / p.Ho);
      ridx = (((int64_t)n * (p.Do >> 1) + (zo >> 1)) * (p.Ho >> 1) + (yo >> 1)) * (p.Wo >> 1) + (xo >> 1);
    }
#pragma unroll
    for (int j = 0; j < CO; j += 4) {
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = acc[j + k];
        if (p.scale_mode == 1) a *= scale[j + k];
        else if (p.scale_mode == 2) a *= scale[n * CO + j + k];
        if (p.has_shift) a += shift[j + k];
        if (p.res_mode) a += res[ridx * CO + j + k];
        r[k] = cfun_apply_act(a, p.act, p.slope);
      }
      *reinterpret_cast<float4*>(y + v * CO + j) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
}

// The same conv with the rows staged through LDS (round 4).  One thread per voxel reading ITS row as float4s makes every
// load instruction of a wave touch 64 different cache lines, each of which the following chunks of the row touch again:
// with 8+ waves per CU the rows in flight (10 KB per wave at C_in = 40) overflow the 32 KB L1 and the kernel ran at
// 3.1 TB/s.  Here a workgroup's 256 voxels x CH channels are ONE contiguous span (CH = C_in) or 128 / 160-byte row
// segments (C_in = k * CH) that the waves read with consecutive lanes on consecutive float4s; rows land in LDS with a
// stride of CH + 4 floats ((CH + 4) / 4 odd: the 8 lanes of one ds_read_b128 phase hit 8 different 4-bank groups) and
// every thread then reads its own row from there.  The next segment's loads are in flight under the FMAs.
template <int CO, int CH, bool IN>
__global__ void __launch_bounds__(256)
k_conv_pointwise_t(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
                   const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
                   CfunConv3dParams p, int64_t total, const float* __restrict__ in_stats, int in_act, float in_slope) {
  constexpr int Q = CH / 4, S = CH + 4;                 // float4s per row segment, LDS row stride
  static_assert((S / 4) % 2 == 1, "row stride must be an odd number of float4s");
  CFUN_DYN_LDS(float4, smem4);
  float* Xl = reinterpret_cast<float*>(smem4);          // [256][S]
  float4* stab = smem4 + 256 * S / 4;                   // IN: [N * Ci / 4][2] (mean, rstd) rows, as in k_conv_pointwise
  const int tid = threadIdx.x;
  if constexpr (IN) {
    const int rows = p.N * (p.Ci >> 2) * 2;
    for (int i = tid; i < rows; i += 256)
      stab[i] = in_stats ? reinterpret_cast<const float4*>(in_stats)[i] : make_float4(0.f, 1.f, 0.f, 1.f);
  }
  const int64_t per_n = (int64_t)p.Do * p.Ho * p.Wo;
  const int npass = p.Ci / CH;
  const int64_t ngroups = (total + 255) >> 8;
  const int64_t nseg = ngroups * npass;                 // segment = (group of 256 voxels, channel pass)
  float4 xin[Q];
  auto prefetch = [&](int64_t seg) __attribute__((always_inline)) {
    const int64_t v0 = (seg / npass) << 8;
    const int c0 = (int)(seg % npass) * CH;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int it = tid + i * 256;
      const int64_t v = v0 + it / Q;
      const int64_t off = v * p.Ci + c0 + (it % Q) * 4;
      xin[i] = *reinterpret_cast<const float4*>(x + (v < total ? off : 0));
    }
  };
  int64_t seg = (int64_t)blockIdx.x * npass;
  const int64_t seg_step = (int64_t)gridDim.x * npass;
  if (seg < nseg) prefetch(seg);
  for (; seg < nseg; seg += seg_step) {                 // this block's groups; the passes of a group run back to back
    const int64_t v = ((seg / npass) << 8) + tid;
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = 0.f;
    for (int ps = 0; ps < npass; ++ps) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int it = tid + i * 256;
        *reinterpret_cast<float4*>(Xl + (it / Q) * S + (it % Q) * 4) = make_float4(xin[i].x, xin[i].y, xin[i].z, xin[i].w);
      }
      __syncthreads();
      const int64_t nxt = ps + 1 < npass ? seg + ps + 1 : seg + seg_step;
      if (nxt < nseg) prefetch(nxt);
      const int c0 = ps * CH;
      const float4* srow = stab + (IN ? ((v < total ? v : 0) / per_n) * (p.Ci >> 2) * 2 + (c0 >> 2) * 2 : 0);
      const float* xr = Xl + tid * S;
#pragma unroll
      for (int c = 0; c < CH; c += 4) {
        float4 xv = *reinterpret_cast<const float4*>(xr + c);
        if constexpr (IN) xv = cfun_mfma::norm_act_in4(xv, srow[(c >> 2) * 2], srow[(c >> 2) * 2 + 1], in_act, in_slope);
        const float* w = wp + (int64_t)(c0 + c) * p.CoP;          // wave-uniform: scalar loads
#pragma unroll
        for (int j = 0; j < CO; ++j) {
          acc[j] = fmaf(xv.x, w[j], acc[j]);
          acc[j] = fmaf(xv.y, w[p.CoP + j], acc[j]);
          acc[j] = fmaf(xv.z, w[2 * p.CoP + j], acc[j]);
          acc[j] = fmaf(xv.w, w[3 * p.CoP + j], acc[j]);
        }
      }
    }
    if (v >= total) continue;
    const int n = (int)(v / per_n);
    int64_t ridx = v;
    if (p.res_mode && p.res_up2) {
      int64_t t = v - n * per_n;
      const int xo = (int)(t % p.Wo); t /= p.Wo;
      const int yo = (int)(t % p.Ho);
      const int zo = (int)(t / p.Ho);
      ridx = (((int64_t)n * (p.Do >> 1) + (zo >> 1)) * (p.Ho >> 1) + (yo >> 1)) * (p.Wo >> 1) + (xo >> 1);
    }
#pragma unroll
    for (int j = 0; j < CO; j += 4) {
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = acc[j + k];
        if (p.scale_mode == 1) a *= scale[j + k];
        else if (p.scale_mode == 2) a *= scale[n * CO + j + k];
        if (p.has_shift) a += shift[j + k];
        if (p.res_mode) a += res[ridx * CO + j + k];
        r[k] = cfun_apply_act(a, p.act, p.slope);
      }
      *reinterpret_cast<float4*>(y + v * CO + j) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
}

template <int CH>
void launch_pointwise_t(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                        const CfunConv3dParams* p, int64_t total, const float* in_stats, int in_act, float in_slope,
                        hipStream_t st) {
  const bool in = in_stats || in_act != CFUN_ACT_NONE;
  const size_t lds = (size_t)256 * (CH + 4) * sizeof(float) + (in ? (size_t)p->N * p->Ci * 2 * sizeof(float) : 0);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 12) blocks = 256 * 12;
  if (in)
    hipLaunchKernelGGL((k_conv_pointwise_t<8, CH, true>), dim3((unsigned)blocks), dim3(256), lds, st, x, wp, scale, shift, res, y,
                       *p, total, in_stats, in_act, in_slope);
  else
    hipLaunchKernelGGL((k_conv_pointwise_t<8, CH, false>), dim3((unsigned)blocks), dim3(256), lds, st, x, wp, scale, shift, res, y,
                       *p, total, nullptr, 0, 0.f);
}

// CFUN_POINTWISE_LDS = 0: the one-thread-per-row kernel for every shape (A/B).  `in`: the launch carries the input prologue,
// whose (mean, rstd) table shares the workgroup's LDS with the staged rows -- the staged kernel only while both fit the 64 KB
// a launch gets without raising hipFuncAttributeMaxDynamicSharedMemorySize (N * Ci <= 2 560 at CH = 40); beyond that the
// per-row kernel, which holds the table alone (the launch failed there with an invalid-value error).
int pointwise_staged_ch(const CfunConv3dParams* p, bool in) {
  static int knob = -2;
  if (knob == -2) {
    const char* e = getenv("CFUN_POINTWISE_LDS");
    knob = e ? atoi(e) : -1;
  }
  if (knob == 0) return 0;
  const int ch = p->Ci % 40 == 0 ? 40 : p->Ci % 32 == 0 ? 32 : 0;
  if (ch == 0) return 0;
  const size_t lds = (size_t)256 * (ch + 4) * sizeof(float) + (in ? (size_t)p->N * p->Ci * 2 * sizeof(float) : 0);
  return lds <= 64 * 1024 ? ch : 0;
}

}  // namespace

int cfun_conv_pointwise_supported(const CfunConv3dParams* p) {
  if (p->kd != 1 || p->kh != 1 || p->kw != 1 || p->stride != 1 || p->up2 || p->d2s) return 0;
  if (p->Co != 8 || (p->Ci & 3)) return 0;
  return (int64_t)p->N * p->Do * p->Ho * p->Wo >= 32768;      // small volumes: the split-K MFMA path
}

// the (mean, rstd) table of the input prologue must fit LDS: N * Ci * 2 floats
int cfun_conv_pointwise_in_supported(const CfunConv3dParams* p) {
  return cfun_conv_pointwise_supported(p) && (size_t)p->N * p->Ci * 2 * sizeof(float) <= 48 * 1024;
}

int cfun_conv_pointwise_fwd(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                            float* y, const CfunConv3dParams* p, const float* in_stats, int in_act, float in_slope,
                            hipStream_t st) {
  const int64_t total = (int64_t)p->N * p->Do * p->Ho * p->Wo;
  if (total <= 0) return CFUN_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if ((in_stats || in_act != CFUN_ACT_NONE) && !cfun_conv_pointwise_in_supported(p)) return CFUN_EINVAL;
  if (const int ch = pointwise_staged_ch(p, in_stats || in_act != CFUN_ACT_NONE)) {
    if (ch == 40) launch_pointwise_t<40>(x, wp, scale, shift, res, y, p, total, in_stats, in_act, in_slope, st);
    else launch_pointwise_t<32>(x, wp, scale, shift, res, y, p, total, in_stats, in_act, in_slope, st);
    CFUN_LAUNCH_CHECK();
    return CFUN_OK;
  }
  if (in_stats || in_act != CFUN_ACT_NONE) {
    const size_t lds = (size_t)p->N * p->Ci * 2 * sizeof(float);
    hipLaunchKernelGGL((k_conv_pointwise<8, true>), dim3((unsigned)blocks), dim3(256), lds, st, x, wp, scale, shift, res, y, *p,
                       total, in_stats, in_act, in_slope);
  } else {
    hipLaunchKernelGGL((k_conv_pointwise<8, false>), dim3((unsigned)blocks), dim3(256), 0, st, x, wp, scale, shift, res, y, *p,
                       total, nullptr, 0, 0.f);
  }
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}
