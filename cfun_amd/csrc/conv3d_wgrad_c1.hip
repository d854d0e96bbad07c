// Weight gradient of the single-input-channel stems (mask_branch.py:23 conv3d_c1_1, backbone.py:124 C1.0) on the
// fp32 matrix cores.  With C_in = 1 the generic wgrad tile (16 input channels as the M axis) would be 94 % padding,
// so here the M axis is the TAP index instead:
//     dW[tap][co] = sum_v X[v*S + tap - pad] * G[v][co]          (a [taps x voxels] x [voxels x co] GEMM)
//   A[i = tap][k = voxel]  is gathered from the single-channel LDS halo tile with a per-lane tap offset,
//   B[k = voxel][j = co]   is the staged gradient tile, D[tap][co] accumulates in registers over all tiles of
//   the workgroup; the 4 waves split the voxel groups and each writes its own partial (summed by
//   cfun_reduce_partials, deterministic).  HBM-bound: one pass over G and X.
#include "conv3d_mfma.h"

int cfun_wgrad_finish(const float*, CfunWgradDst, const CfunConv3dParams*, int, hipStream_t);

namespace {

using cfun_mfma::cdiv;
using cfun_mfma::pad_row16;

template <int KD, int KH, int KW, int S>
struct C1Tile {
  static constexpr int TD = (S == 1) ? 2 : 1, TH = 4, TW = 16;
  static constexpr int TVOX = TD * TH * TW;
  static constexpr int TAPS = KD * KH * KW;
  static constexpr int MSUB = cdiv(TAPS, 16);
  static constexpr int IZ = (TD - 1) * S + KD, IY = (TH - 1) * S + KH, IX = (TW - 1) * S + KW;
  static constexpr int IVOX = IZ * IY * IX;
  static constexpr int X_LOADS = cdiv(IVOX, 256);
};

template <int KD, int KH, int KW, int S, int NSUB>
__global__ void __launch_bounds__(256)
k_wgrad_c1_mfma(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial,
                CfunConv3dParams p, int ntz, int nty, int ntx, int tiles_per_chunk, int ntiles) {
  using T = C1Tile<KD, KH, KW, S>;
  constexpr int NT = 16 * NSUB, GS = pad_row16(NT);
  constexpr int G_ITEMS = T::TVOX * (NT / 4);
  constexpr int G_LOADS = cdiv(G_ITEMS, 256);
  CFUN_DYN_LDS(float, smem);
  float* Xl = smem;                                 // [IVOX]
  float* Gl = smem + ((T::IVOX + 3) & ~3);          // [TVOX][GS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int chunk = blockIdx.x;
  const int sh = p.up2 ? 1 : 0;
  const int Dv = p.Di << sh, Hv = p.Hi << sh, Wv = p.Wi << sh;

  // per-lane tap offsets (lane&15 selects the tap inside each 16-tap M-subtile)
  int toff[T::MSUB];
  float tmask[T::MSUB];
#pragma unroll
  for (int m = 0; m < T::MSUB; ++m) {
    const int tap = m * 16 + (lane & 15);
    const bool ok = tap < T::TAPS;
    const int tt = ok ? tap : 0;
    const int dz = tt / (KH * KW), dy = (tt / KW) % KH, dx = tt % KW;
    toff[m] = (dz * T::IY + dy) * T::IX + dx;
    tmask[m] = ok ? 1.f : 0.f;
  }
  f32x4 acc[T::MSUB][NSUB];
#pragma unroll
  for (int m = 0; m < T::MSUB; ++m)
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) acc[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};

  float xin[T::X_LOADS];
  float4 gin[G_LOADS];
  auto prefetch = [&](int tile) {
    int t = tile;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz;
    const int n = t / ntz;
    const int z0 = tz * T::TD, y0 = ty * T::TH, x0 = tx * T::TW;
#pragma unroll
    for (int i = 0; i < T::X_LOADS; ++i) {
      const int idx = tid + i * 256;
      xin[i] = 0.f;
      if (idx < T::IVOX) {
        const int ix = idx % T::IX, iy = (idx / T::IX) % T::IY, iz = idx / (T::IX * T::IY);
        const int vz = z0 * S - p.pd + iz, vy = y0 * S - p.ph + iy, vx = x0 * S - p.pw + ix;
        if (vz >= 0 && vz < Dv && vy >= 0 && vy < Hv && vx >= 0 && vx < Wv)
          xin[i] = x[(((int64_t)n * p.Di + (vz >> sh)) * p.Hi + (vy >> sh)) * p.Wi + (vx >> sh)];
      }
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      gin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < G_ITEMS) {
        const int vox = it / (NT / 4), col = (it % (NT / 4)) * 4;
        const int lx = vox % T::TW, ly = (vox / T::TW) % T::TH, lz = vox / (T::TW * T::TH);
        const int oz = z0 + lz, oy = y0 + ly, ox = x0 + lx;
        if (col < p.Co && oz < p.Do && oy < p.Ho && ox < p.Wo)
          gin[i] = *reinterpret_cast<const float4*>(g + ((((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Co + col);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < T::X_LOADS; ++i) {
      const int idx = tid + i * 256;
      if (idx < T::IVOX) Xl[idx] = xin[i];
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < G_ITEMS) *reinterpret_cast<float4*>(Gl + (it / (NT / 4)) * GS + (it % (NT / 4)) * 4) = gin[i];
    }
  };

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = (t_begin + tiles_per_chunk < ntiles) ? t_begin + tiles_per_chunk : ntiles;
  const float* Gw = Gl + (lane >> 4) * GS + (lane & 15);
  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
    for (int grp = wv; grp < T::TVOX / 4; grp += 4) {   // 4 consecutive x per group; waves split the groups
      const int xq = grp & 3, ly = (grp >> 2) & 3, lz = grp >> 4;
      float b[NSUB];
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) b[nn] = Gw[(grp * 4) * GS + nn * 16];
      const float* Xg = Xl + ((lz * S) * T::IY + ly * S) * T::IX + (xq * 4 + (lane >> 4)) * S;
#pragma unroll
      for (int m = 0; m < T::MSUB; ++m) {
        const float a = Xg[toff[m]] * tmask[m];
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn)
          acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nn], acc[m][nn], 0, 0, 0);
      }
    }
  }
  // D[i = tap][j = co]: lane -> co = nn*16 + (lane&15), tap = m*16 + (lane>>4)*4 + r.  The 4 waves hold partial
  // sums over disjoint voxel groups: combine them through LDS (wave order 0..3, deterministic), one partial per
  // workgroup: partial[chunk][tap][0][CoP].
  __syncthreads();                       // all waves are done with Xl / Gl
  float* red = smem;                     // [4 waves][MSUB*16 taps][NT]   (<= 4*256*32 floats = 128 KB for 5x7x7)
  constexpr int MROWS = T::MSUB * 16;
#pragma unroll
  for (int m = 0; m < T::MSUB; ++m)
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[((wv * MROWS) + m * 16 + (lane >> 4) * 4 + r) * NT + nn * 16 + (lane & 15)] = acc[m][nn][r];
  __syncthreads();
  float* out = partial + (int64_t)chunk * T::TAPS * p.CoP;
  for (int e = tid; e < T::TAPS * NT; e += 256) {
    const int tap = e / NT, co = e - tap * NT;
    if (co < p.CoP)
      out[(int64_t)tap * p.CoP + co] = (red[(0 * MROWS + tap) * NT + co] + red[(1 * MROWS + tap) * NT + co]) +
                                       (red[(2 * MROWS + tap) * NT + co] + red[(3 * MROWS + tap) * NT + co]);
  }
}

struct C1Plan {
  int ntz, nty, ntx, ntiles, nchunks, tiles_per_chunk;
};

template <int KD, int KH, int KW, int S>
C1Plan c1_plan(const CfunConv3dParams& p) {
  using T = C1Tile<KD, KH, KW, S>;
  C1Plan w;
  w.ntz = cdiv(p.Do, T::TD); w.nty = cdiv(p.Ho, T::TH); w.ntx = cdiv(p.Wo, T::TW);
  w.ntiles = p.N * w.ntz * w.nty * w.ntx;
  int want = w.ntiles < 512 ? (w.ntiles > 0 ? w.ntiles : 1) : 512;   // HBM-bound: ~2 workgroups per CU
  w.tiles_per_chunk = cdiv(w.ntiles > 0 ? w.ntiles : 1, want);
  w.nchunks = cdiv(w.ntiles > 0 ? w.ntiles : 1, w.tiles_per_chunk);
  return w;
}

template <int KD, int KH, int KW, int S, int NSUB>
int launch_c1(const float* x, const float* g, CfunWgradDst dst, const CfunConv3dParams& p, void* ws, size_t ws_bytes,
              hipStream_t st) {
  using T = C1Tile<KD, KH, KW, S>;
  constexpr int NT = 16 * NSUB, GS = pad_row16(NT);
  const C1Plan w = c1_plan<KD, KH, KW, S>(p);
  const int64_t nout = (int64_t)T::TAPS * p.CoP;
  if ((size_t)w.nchunks * nout * sizeof(float) > ws_bytes) return CFUN_EWORKSPACE;
  size_t lds = (size_t)(((T::IVOX + 3) & ~3) + T::TVOX * GS) * sizeof(float);
  const size_t lds_red = (size_t)4 * T::MSUB * 16 * NT * sizeof(float);
  if (lds_red > lds) lds = lds_red;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_c1_mfma<KD, KH, KW, S, NSUB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  auto kern = k_wgrad_c1_mfma<KD, KH, KW, S, NSUB>;
  hipLaunchKernelGGL(kern, dim3((unsigned)w.nchunks), dim3(256), lds, st, x, g, (float*)ws, p, w.ntz, w.nty, w.ntx,
                     w.tiles_per_chunk, w.ntiles);
  CFUN_LAUNCH_CHECK();
  return cfun_wgrad_finish((const float*)ws, dst, &p, w.nchunks, st);
}

template <int KD, int KH, int KW, int S>
int dispatch_c1(const float* x, const float* g, CfunWgradDst dst, const CfunConv3dParams& p, void* ws, size_t ws_bytes,
                hipStream_t st) {
  if (p.CoP <= 16) return launch_c1<KD, KH, KW, S, 1>(x, g, dst, p, ws, ws_bytes, st);
  return launch_c1<KD, KH, KW, S, 2>(x, g, dst, p, ws, ws_bytes, st);
}

}  // namespace

// 0 = shape not handled here
int cfun_wgrad_c1_supported(const CfunConv3dParams* p) {
  if (p->Ci != 1 || (p->Co & 3) || p->CoP > 32 || p->up2) return 0;
  if (p->kd == 3 && p->kh == 3 && p->kw == 3 && p->stride == 1) return 1;
  if (p->kd == 3 && p->kh == 7 && p->kw == 7 && p->stride == 2) return 2;
  if (p->kd == 5 && p->kh == 7 && p->kw == 7 && p->stride == 2) return 3;
  return 0;
}

size_t cfun_wgrad_c1_ws(const CfunConv3dParams* p) {
  C1Plan w;
  int taps;
  switch (cfun_wgrad_c1_supported(p)) {
    case 1: w = c1_plan<3, 3, 3, 1>(*p); taps = 27; break;
    case 2: w = c1_plan<3, 7, 7, 2>(*p); taps = 147; break;
    case 3: w = c1_plan<5, 7, 7, 2>(*p); taps = 245; break;
    default: return 0;
  }
  return (size_t)w.nchunks * taps * p->CoP * sizeof(float);
}

int cfun_wgrad_c1(const float* x, const float* g, CfunWgradDst dst, const CfunConv3dParams* p, void* ws, size_t ws_bytes,
                  hipStream_t st) {
  switch (cfun_wgrad_c1_supported(p)) {
    case 1: return dispatch_c1<3, 3, 3, 1>(x, g, dst, *p, ws, ws_bytes, st);
    case 2: return dispatch_c1<3, 7, 7, 2>(x, g, dst, *p, ws, ws_bytes, st);
    case 3: return dispatch_c1<5, 7, 7, 2>(x, g, dst, *p, ws, ws_bytes, st);
    default: return CFUN_EINVAL;
  }
}
