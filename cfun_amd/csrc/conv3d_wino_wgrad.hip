// Weight gradient of the 3x3x3 stride-1 convs with the x axis (k_wgrad_wino) or x and y (k_wgrad_wino2, opt-in) in the
// Winograd F(2,3) domain; forward / data gradient and the shape rules: conv3d_wino.hip.
#include <stdlib.h>

#include "conv3d_mfma.h"

int cfun_wino_shape_ok(const CfunConv3dParams* p);      // conv3d_wino.hip

// dW of the same convs with the x axis in the F(2,3) domain.  For every x-pair of outputs the four products
//   dU_p[ci][co] += V_p[ci] * dM_p[co],   dM = (g_e, g_e + g_o, g_e - g_o, -g_o),   V as in the forward kernel
// replace the 2 x 3 products of the direct sum; at the end dW = G^T dU:
//   dw0 = dU0 + (dU1 + dU2)/2,  dw1 = (dU1 - dU2)/2,  dw2 = dU3 + (dU1 + dU2)/2.
//   block  = 256 threads, one 16-channel ci subtile x (16*NSUB) co, a range of 2(z) x 4(y) x 16(x) voxel tiles
//   wave w = Winograd point w: all 9 (dz,dy) offsets -> accumulators [9][NSUB]; MFMA k-step = 4 consecutive x-pairs
//   LDS    = the raw halo tile [24 rows][x parity][9][16 ci] and gradient tile [8 rows][x parity][8][co]: V and dM are
//            formed from two LDS reads each when the fragment is read (V_p = X[o1] + s*X[o2], dM_p = a*g_e + b*g_o with
//            wave-uniform offsets / signs), so the tiles stay as small as the direct kernel's
//   end    = per (dz,dy) the four waves' sums meet in LDS, G^T is applied and the 3 taps are written in the partial
//            layout of k_wgrad_mfma ([chunk][tap][Ci][CoP]); cfun_wgrad_finish reduces the chunks as before.
namespace {

using cfun_mfma::cdiv;

constexpr int WG_IY = 6, WG_XH = 9, WG_XROWS = 4 * WG_IY;      // halo rows (z,y); 9 columns per x parity
constexpr int WG_XVOX = WG_XROWS * 2 * WG_XH;                  // 432 staged voxels x 16 channels
constexpr int WG_GVOX = 2 * 4 * 16;                            // 128 gradient voxels
constexpr int WG_XROW = 2 * WG_XH * 16;                        // floats per halo row

template <int NSUB, bool D2S>
__global__ void __launch_bounds__(256)
k_wgrad_wino(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial, CfunConv3dParams p,
             int ntz, int nty, int ntx, int ncisub, int ncot, int tiles_per_chunk, int ntiles) {
  constexpr int NT = 16 * NSUB, GS = cfun_mfma::pad_row16(NT);
  constexpr int X_ITEMS = WG_XVOX * 4, X_LOADS = cdiv(X_ITEMS, 256);
  constexpr int G_ITEMS = WG_GVOX * (NT / 4), G_LOADS = cdiv(G_ITEMS, 256);
  CFUN_DYN_LDS(float4, smem4);
  float* Xl = reinterpret_cast<float*>(smem4);      // [row][parity][9][16]
  float* Gl = Xl + WG_XVOX * 16;                    // [(lrow*2 + parity)*8 + j][GS]

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned lid = cfun_mfma::xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot; lid /= ncot;
  const int cis = lid % ncisub;
  const int chunk = lid / ncisub;
  const int ci0 = cis * 16, cobase = cot * NT;

  f32x4 acc[9][NSUB];
#pragma unroll
  for (int r = 0; r < 9; ++r)
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) acc[r][nn] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: unconditional loads (out-of-range items read offset 0), zero padding applied at commit time
  float4 xin[X_LOADS], gin[G_LOADS];
  unsigned xvalid = 0, gvalid = 0;
  auto prefetch = [&](int tile) {
    int t = tile;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz;
    const int n = t / ntz;
    const int z0 = tz * 2, y0 = ty * 4, x0 = tx * 16;
    xvalid = 0; gvalid = 0;
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int it = tid + i * 256;
      const int idx = it >> 2, c = ci0 + (it & 3) * 4;
      const int xh = idx % WG_XH, par = (idx / WG_XH) & 1, row = idx / (2 * WG_XH);
      const int vz = z0 - p.pd + row / WG_IY, vy = y0 - 1 + row % WG_IY, vx = x0 - 1 + 2 * xh + par;
      const bool ok = (it < X_ITEMS) & (c < p.Ci) & (vz >= 0) & (vz < p.Di) & (vy >= 0) & (vy < p.Hi) & (vx >= 0) & (vx < p.Wi);
      const unsigned off = ((((unsigned)n * p.Di + vz) * p.Hi + vy) * p.Wi + vx) * p.Ci + c;
      xin[i] = *reinterpret_cast<const float4*>(x + (ok ? off : 0u));
      xvalid |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      const int slot = it / (NT / 4), col = (it % (NT / 4)) * 4;
      const int j = slot & 7, par = (slot >> 3) & 1, lrow = slot >> 4;
      const int oz = z0 + (lrow >> 2), oy = y0 + (lrow & 3), ox = x0 + 2 * j + par;
      bool ok = (it < G_ITEMS) & (cobase + col < p.Co) & (oz < p.Do) & (oy < p.Ho) & (ox < p.Wo);
      unsigned off;
      if (D2S) {   // g is the hi-res gradient of y [N,2Do,2Ho,2Wo,Cq]: gather parity q, channel o
        const int CqP = p.Co >> 3, Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;
        const int co = cobase + col, q = co / CqP, o = co - q * CqP;
        ok = ok & (o < Cq);
        off = ((((unsigned)n * 2 * p.Do + 2 * oz + (q >> 2)) * 2 * p.Ho + 2 * oy + ((q >> 1) & 1)) * 2 * p.Wo + 2 * ox +
               (q & 1)) * Cq + o;
      } else {
        off = ((((unsigned)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Co + cobase + col;
      }
      gin[i] = *reinterpret_cast<const float4*>(g + (ok ? off : 0u));
      gvalid |= (ok ? 1u : 0u) << i;
    }
  };
  auto commit = [&]() {
    auto keep = [](unsigned bit, const float4& v) {
      const float m = bit ? 1.f : 0.f;
      return make_float4(bit ? v.x : m, bit ? v.y : m, bit ? v.z : m, bit ? v.w : m);
    };
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < X_ITEMS) *reinterpret_cast<float4*>(Xl + it * 4) = keep((xvalid >> i) & 1u, xin[i]);
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < G_ITEMS)
        *reinterpret_cast<float4*>(Gl + (it / (NT / 4)) * GS + (it % (NT / 4)) * 4) = keep((gvalid >> i) & 1u, gin[i]);
    }
  };

  // point of this wave: V_p = X[o1] + s2 * X[o2] over (E[j], O[j], E[j+1], O[j+1]) = float offsets (0, 144, 16, 160);
  // dM_p = ce * g_e + cg * g_o
  const int o1 = wv == 0 ? 0 : wv == 2 ? 16 : WG_XH * 16;
  const int o2 = wv == 2 ? WG_XH * 16 : wv == 3 ? WG_XH * 16 + 16 : 16;
  const float s2 = wv == 1 ? 1.f : -1.f;
  const float ce = wv == 3 ? 0.f : 1.f, cg = wv == 0 ? 0.f : wv == 1 ? 1.f : -1.f;
  // fragments: A row i = ci (lane & 15), k = x-pair lane >> 4 of the quad;  B col = co (lane & 15), same k
  const float* Xa1 = Xl + o1 + (lane >> 4) * 16 + (lane & 15);
  const float* Xa2 = Xl + o2 + (lane >> 4) * 16 + (lane & 15);
  const float* Gw = Gl + (lane >> 4) * GS + (lane & 15);

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = (t_begin + tiles_per_chunk < ntiles) ? t_begin + tiles_per_chunk : ntiles;
  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
#pragma unroll 1
    for (int grp = 0; grp < 16; ++grp) {       // (output row lrow = (lz, ly), quad q of 4 x-pairs)
      const int lrow = grp >> 1, q = grp & 1;
      const int hrow = (lrow >> 2) * WG_IY + (lrow & 3);
      float b[NSUB], a[9];
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) {
        const float* gp = Gw + ((lrow * 2) * 8 + q * 4) * GS + nn * 16;
        b[nn] = ce * gp[0] + cg * gp[8 * GS];
      }
      const float* x1 = Xa1 + hrow * WG_XROW + q * 64;
      const float* x2 = Xa2 + hrow * WG_XROW + q * 64;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int o = ((r / 3) * WG_IY + (r % 3)) * WG_XROW;
        a[r] = x1[o] + s2 * x2[o];
      }
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn)
          acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[nn], acc[r][nn], 0, 0, 0);
    }
  }

  // ---- dW = G^T dU per (dz,dy): D[i = ci][j = co], lane -> co = lane & 15, rows (lane >> 4)*4 + r
  float* S = Xl;                                   // [4 points][16 ci][NT]
  float* out = partial + (int64_t)chunk * 27 * p.Ci * p.CoP;
#pragma unroll
  for (int r9 = 0; r9 < 9; ++r9) {
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        S[(wv * 16 + (lane >> 4) * 4 + r) * NT + nn * 16 + (lane & 15)] = acc[r9][nn][r];
    __syncthreads();
    for (int e = tid; e < 16 * NT; e += 256) {
      const int ci = ci0 + e / NT, co = cobase + e % NT;
      if (ci >= p.Ci || co >= p.CoP) continue;
      const float u0 = S[e], u1 = S[16 * NT + e], u2 = S[2 * 16 * NT + e], u3 = S[3 * 16 * NT + e];
      const float h = 0.5f * (u1 + u2);
      float* o = out + ((int64_t)(r9 * 3) * p.Ci + ci) * p.CoP + co;
      o[0] = u0 + h;
      o[(int64_t)p.Ci * p.CoP] = 0.5f * (u1 - u2);
      o[2 * (int64_t)p.Ci * p.CoP] = u3 + h;
    }
  }
}

// ---- the same weight gradient with y in the Winograd domain as well: per 2 x 2 block of outputs 16 products
//   dU[py][px][ci][co] += V[py][px][ci] * dM[py][px][co]     (V = B^T d B of the 4 x 4 input patch, dM = A (g) A^T)
// replace the 4 x 9 of the direct sum (4/9 of the MFMAs); dW = G^T dU G at the end.  512 threads: wave = (py, half of
// the px), 3 (dz) x 2 (px) x NSUB accumulators, MFMA k-step = 4 consecutive blocks of one row pair.  The tiles stay raw in
// LDS as in k_wgrad_wino: per dz a wave reads 2 rows x 3 columns of x and forms its two V values, per channel subtile
// 2 x 2 gradient values for its two dM values -- all with wave-uniform offsets and coefficients.
template <int NSUB, bool D2S>
__global__ void __launch_bounds__(512)
k_wgrad_wino2(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial, CfunConv3dParams p,
              int ntz, int nty, int ntx, int ncisub, int ncot, int tiles_per_chunk, int ntiles) {
  constexpr int NT = 16 * NSUB, GS = cfun_mfma::pad_row16(NT);
  constexpr int X_ITEMS = WG_XVOX * 4, X_LOADS = cdiv(X_ITEMS, 512);
  constexpr int G_ITEMS = WG_GVOX * (NT / 4), G_LOADS = cdiv(G_ITEMS, 512);
  CFUN_DYN_LDS(float4, smem4);
  float* Xl = reinterpret_cast<float*>(smem4);      // [row][parity][9][16]
  float* Gl = Xl + WG_XVOX * 16;                    // [(lrow*2 + parity)*8 + j][GS]

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned lid = cfun_mfma::xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot; lid /= ncot;
  const int cis = lid % ncisub;
  const int chunk = lid / ncisub;
  const int ci0 = cis * 16, cobase = cot * NT;

  f32x4 acc[3][2][NSUB];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) acc[dz][h][nn] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 xin[X_LOADS], gin[G_LOADS];
  unsigned xvalid = 0, gvalid = 0;
  auto prefetch = [&](int tile) {
    int t = tile;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz;
    const int n = t / ntz;
    const int z0 = tz * 2, y0 = ty * 4, x0 = tx * 16;
    xvalid = 0; gvalid = 0;
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int it = tid + i * 512;
      const int idx = it >> 2, c = ci0 + (it & 3) * 4;
      const int xh = idx % WG_XH, par = (idx / WG_XH) & 1, row = idx / (2 * WG_XH);
      const int vz = z0 - p.pd + row / WG_IY, vy = y0 - 1 + row % WG_IY, vx = x0 - 1 + 2 * xh + par;
      const bool ok = (it < X_ITEMS) & (c < p.Ci) & (vz >= 0) & (vz < p.Di) & (vy >= 0) & (vy < p.Hi) & (vx >= 0) & (vx < p.Wi);
      const unsigned off = ((((unsigned)n * p.Di + vz) * p.Hi + vy) * p.Wi + vx) * p.Ci + c;
      xin[i] = *reinterpret_cast<const float4*>(x + (ok ? off : 0u));
      xvalid |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 512;
      const int slot = it / (NT / 4), col = (it % (NT / 4)) * 4;
      const int j = slot & 7, par = (slot >> 3) & 1, lrow = slot >> 4;
      const int oz = z0 + (lrow >> 2), oy = y0 + (lrow & 3), ox = x0 + 2 * j + par;
      bool ok = (it < G_ITEMS) & (cobase + col < p.Co) & (oz < p.Do) & (oy < p.Ho) & (ox < p.Wo);
      unsigned off;
      if (D2S) {
        const int CqP = p.Co >> 3, Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;
        const int co = cobase + col, q = co / CqP, o = co - q * CqP;
        ok = ok & (o < Cq);
        off = ((((unsigned)n * 2 * p.Do + 2 * oz + (q >> 2)) * 2 * p.Ho + 2 * oy + ((q >> 1) & 1)) * 2 * p.Wo + 2 * ox +
               (q & 1)) * Cq + o;
      } else {
        off = ((((unsigned)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Co + cobase + col;
      }
      gin[i] = *reinterpret_cast<const float4*>(g + (ok ? off : 0u));
      gvalid |= (ok ? 1u : 0u) << i;
    }
  };
  auto commit = [&]() {
    auto keep = [](unsigned bit, const float4& v) {
      const float m = bit ? 1.f : 0.f;
      return make_float4(bit ? v.x : m, bit ? v.y : m, bit ? v.z : m, bit ? v.w : m);
    };
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int it = tid + i * 512;
      if (it < X_ITEMS) *reinterpret_cast<float4*>(Xl + it * 4) = keep((xvalid >> i) & 1u, xin[i]);
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 512;
      if (it < G_ITEMS)
        *reinterpret_cast<float4*>(Gl + (it / (NT / 4)) * GS + (it % (NT / 4)) * 4) = keep((gvalid >> i) & 1u, gin[i]);
    }
  };

  // this wave: y point py, x points 2h and 2h+1
  const int py = wv >> 1, h = wv & 1;
  // B^T d along y: rows (rA, rB) of the 4-row patch, V = X[rA] + sy * X[rB]
  const int rA = py == 0 ? 0 : py == 2 ? 2 : 1, rB = py == 0 ? 2 : py == 1 ? 2 : py == 2 ? 1 : 3;
  const float sy = py == 1 ? 1.f : -1.f;
  // along x the wave needs 3 of the patch's 4 columns (E[j], O[j], E[j+1], O[j+1] = float offsets 0, 144, 16, 160):
  //   h = 0: c = (0,1,2): lo = c0 - c2, hi = c1 + c2;   h = 1: c = (1,2,3): lo = c(2) - c(1) = r1 - r0, hi = c(1) - c(3) = r0 - r2
  const int cofs0 = h ? 144 : 0, cofs1 = h ? 16 : 144, cofs2 = h ? 160 : 16;
  const float lo0 = h ? -1.f : 1.f, lo1 = h ? 1.f : 0.f, lo2 = h ? 0.f : -1.f;
  const float hi0 = h ? 1.f : 0.f, hi1 = h ? 0.f : 1.f, hi2 = h ? -1.f : 1.f;
  // dM = A g A^T: point index 0..3 -> coefficients on (even, odd): (1,0), (1,1), (1,-1), (0,-1)
  const float ye = py == 3 ? 0.f : 1.f, yo = py == 0 ? 0.f : py == 1 ? 1.f : -1.f;
  const float xle = 1.f, xlo = h ? -1.f : 0.f;                    // px = 2h:   h=0 -> (1,0), h=1 -> (1,-1)
  const float xhe = h ? 0.f : 1.f, xho = h ? -1.f : 1.f;          // px = 2h+1: h=0 -> (1,1), h=1 -> (0,-1)
  const float* Xw = Xl + (lane >> 4) * 16 + (lane & 15);
  const float* Gw = Gl + (lane >> 4) * GS + (lane & 15);

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = (t_begin + tiles_per_chunk < ntiles) ? t_begin + tiles_per_chunk : ntiles;
  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
#pragma unroll 1
    for (int ks = 0; ks < 8; ++ks) {            // (lz, row pair yp, quad q of 4 x-pairs)
      const int lz = ks >> 2, yp = (ks >> 1) & 1, q = ks & 1;
      float b[2][NSUB];
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) {
        const float* gp = Gw + (((lz * 4 + 2 * yp) * 2) * 8 + q * 4) * GS + nn * 16;      // row 2yp: even cols; + 8*GS odd cols
        const float te = ye * gp[0] + yo * gp[2 * 8 * GS];                                 // row 2yp+1 = + 2*8*GS
        const float to = ye * gp[8 * GS] + yo * gp[3 * 8 * GS];
        b[0][nn] = xle * te + xlo * to;
        b[1][nn] = xhe * te + xho * to;
      }
      const float* xq = Xw + ((lz * WG_IY + 2 * yp) * WG_XROW) + q * 64;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const float* xa = xq + (dz * WG_IY + rA) * WG_XROW;
        const float* xb = xq + (dz * WG_IY + rB) * WG_XROW;
        const float a0 = xa[cofs0], a1 = xa[cofs1], a2 = xa[cofs2];
        const float b0 = xb[cofs0], b1 = xb[cofs1], b2 = xb[cofs2];
        const float vlo = (lo0 * a0 + (lo1 * a1 + lo2 * a2)) + sy * (lo0 * b0 + (lo1 * b1 + lo2 * b2));
        const float vhi = (hi0 * a0 + (hi1 * a1 + hi2 * a2)) + sy * (hi0 * b0 + (hi1 * b1 + hi2 * b2));
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn) {
          acc[dz][0][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(vlo, b[0][nn], acc[dz][0][nn], 0, 0, 0);
          acc[dz][1][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(vhi, b[1][nn], acc[dz][1][nn], 0, 0, 0);
        }
      }
    }
  }

  // ---- dW = G^T dU G per dz: the 16 points meet in LDS; D[i = ci][j = co], lane -> co = lane & 15, rows (lane >> 4)*4 + r
  float* S = Xl;                                   // [16 points][16 ci][NT]
  float* out = partial + (int64_t)chunk * 27 * p.Ci * p.CoP;
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          S[((py * 4 + 2 * h + hh) * 16 + (lane >> 4) * 4 + r) * NT + nn * 16 + (lane & 15)] = acc[dz][hh][nn][r];
    __syncthreads();
    for (int e = tid; e < 16 * NT; e += 512) {
      const int ci = ci0 + e / NT, co = cobase + e % NT;
      if (ci >= p.Ci || co >= p.CoP) continue;
      float t[3][4];                               // G^T along y: rows (1,.5,.5,0), (0,.5,-.5,0), (0,.5,.5,1)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float u0 = S[(0 * 4 + px) * 16 * NT + e], u1 = S[(1 * 4 + px) * 16 * NT + e], u2 = S[(2 * 4 + px) * 16 * NT + e],
                    u3 = S[(3 * 4 + px) * 16 * NT + e];
        const float hsum = 0.5f * (u1 + u2);
        t[0][px] = u0 + hsum; t[1][px] = 0.5f * (u1 - u2); t[2][px] = u3 + hsum;
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float hsum = 0.5f * (t[ky][1] + t[ky][2]);
        float* o = out + ((int64_t)((dz * 3 + ky) * 3) * p.Ci + ci) * p.CoP + co;
        o[0] = t[ky][0] + hsum;
        o[(int64_t)p.Ci * p.CoP] = 0.5f * (t[ky][1] - t[ky][2]);
        o[2 * (int64_t)p.Ci * p.CoP] = t[ky][3] + hsum;
      }
    }
  }
}

struct WgPlanW {
  int nsub, ntz, nty, ntx, ntiles, ncisub, ncot, nchunks, tiles_per_chunk;
};

WgPlanW make_wg_plan(const CfunConv3dParams& p) {
  WgPlanW w;
  w.nsub = p.CoP <= 16 ? 1 : ((p.CoP + 31) / 32 * 32 < (p.CoP + 47) / 48 * 48 ? 2 : 3);   // fewest padded columns, widest on ties
  w.ntz = cdiv(p.Do, 2); w.nty = cdiv(p.Ho, 4); w.ntx = cdiv(p.Wo, 16);
  w.ntiles = p.N * w.ntz * w.nty * w.ntx;
  w.ncisub = cdiv(p.Ci, 16);
  w.ncot = cdiv(p.CoP, 16 * w.nsub);
  // workgroups per launch: ~2 per CU (as cfun_mfma::wgrad_plan) while a (ci, co) tile pair gets many voxel chunks; with
  // many tile pairs (wide layers on small volumes: 160 -> 160 @ 24^3, 320 -> 320 @ 12^3) a chunk count of 3 - 10 quantises
  // badly, and 3 - 4 per CU measured 7 - 15 % faster (tools/bench_layers.py, round 3).  CFUN_WINO_WGRAD_WGS overrides.
  static int knob = -1;
  if (knob < 0) {
    const char* e = getenv("CFUN_WINO_WGRAD_WGS");
    knob = e && atoi(e) > 0 ? atoi(e) : 0;
  }
  const int pairs = w.ncisub * w.ncot;
  const int total = knob ? knob : pairs >= 100 ? 1024 : pairs >= 48 ? 768 : 512;
  int want = total / pairs;
  if (want > w.ntiles) want = w.ntiles;
  if (want < 1) want = 1;
  w.tiles_per_chunk = cdiv(w.ntiles, want);
  if (w.tiles_per_chunk < 1) w.tiles_per_chunk = 1;
  w.nchunks = cdiv(w.ntiles, w.tiles_per_chunk);
  if (w.nchunks < 1) w.nchunks = 1;
  return w;
}

// y in the Winograd domain for the weight gradient?  CFUN_WINO_WGRAD_2D: 0 = never, 1 = every supported shape
int wino_wgrad_2d(const CfunConv3dParams& p) {
  static int knob = -2;
  if (knob == -2) {
    const char* e = getenv("CFUN_WINO_WGRAD_2D");
    knob = e ? atoi(e) : -1;
  }
  if (p.algo == CFUN_ALGO_WINO) return 0;       // tests: the 1-D kernel
  if (p.algo == CFUN_ALGO_WINO2 || knob == 1) return 1;
  return 0;
}

template <int NSUB>
int launch_wg2(const float* x, const float* g, float* partial, const CfunConv3dParams& p, const WgPlanW& w, hipStream_t st) {
  constexpr int GS = cfun_mfma::pad_row16(16 * NSUB);
  size_t lds = (size_t)(WG_XVOX * 16 + WG_GVOX * GS) * sizeof(float);
  const size_t scratch = (size_t)16 * 16 * 16 * NSUB * sizeof(float);
  if (lds < scratch) lds = scratch;
  auto kern = p.d2s ? k_wgrad_wino2<NSUB, true> : k_wgrad_wino2<NSUB, false>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(w.nchunks * w.ncisub * w.ncot)), dim3(512), lds, st, x, g, partial, p, w.ntz,
                     w.nty, w.ntx, w.ncisub, w.ncot, w.tiles_per_chunk, w.ntiles);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

template <int NSUB>
int launch_wg(const float* x, const float* g, float* partial, const CfunConv3dParams& p, const WgPlanW& w, hipStream_t st) {
  constexpr int GS = cfun_mfma::pad_row16(16 * NSUB);
  size_t lds = (size_t)(WG_XVOX * 16 + WG_GVOX * GS) * sizeof(float);
  const size_t scratch = (size_t)4 * 16 * 16 * NSUB * sizeof(float);
  if (lds < scratch) lds = scratch;
  auto kern = p.d2s ? k_wgrad_wino<NSUB, true> : k_wgrad_wino<NSUB, false>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(w.nchunks * w.ncisub * w.ncot)), dim3(256), lds, st, x, g, partial, p, w.ntz,
                     w.nty, w.ntx, w.ncisub, w.ncot, w.tiles_per_chunk, w.ntiles);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // namespace

int cfun_wino_wgrad_supported(const CfunConv3dParams* p) {
  static int knob = -2;       // CFUN_WINO_WGRAD: 0 = never, 1 = every supported shape
  if (knob == -2) {
    const char* e = getenv("CFUN_WINO_WGRAD");
    knob = e ? atoi(e) : -1;
  }
  if (knob == 0 || !cfun_wino_shape_ok(p)) return 0;
  const int64_t lim = (int64_t)1 << 31;      // 32-bit element offsets
  if ((int64_t)p->N * p->Di * p->Hi * p->Wi * p->Ci >= lim || (int64_t)p->N * p->Do * p->Ho * p->Wo * p->Co >= lim) return 0;
  if (knob == 1 || p->algo == CFUN_ALGO_WINO || p->algo == CFUN_ALGO_WINO2) return 1;
  if (p->d2s) return 0;       // the folded 5x5x5 conv (C_in = 8): the direct kernel's packed tap pairs win (1.03 vs 1.29 ms)
  // C_in <= 8 (packed tap groups: 7 / 14 fragment rows) and 17..20 (fused plain + packed rows) stay on the direct kernels:
  // measured with tools/bench_layers.py (8->20 0.172 vs 0.191 ms, 20->20 1.17 vs 1.20; 12->20 0.251 -> 0.189, 40->40 3.41 -> 2.65)
  return !(p->Ci <= 8 || (p->Ci > 16 && p->Ci <= 20));
}

size_t cfun_wino_wgrad_workspace_bytes(const CfunConv3dParams* p) {
  const WgPlanW w = make_wg_plan(*p);
  return (size_t)w.nchunks * 27 * p->Ci * p->CoP * sizeof(float);
}

// partial sums [nchunks][27][Ci][CoP] into ws; *nparts = nchunks (reduced by cfun_wgrad_finish)
int cfun_wino_wgrad(const float* x, const float* g, float* ws, const CfunConv3dParams* p, int* nparts, hipStream_t st) {
  const WgPlanW w = make_wg_plan(*p);
  *nparts = w.nchunks;
  if (wino_wgrad_2d(*p)) {
    switch (w.nsub) {
      case 1: return launch_wg2<1>(x, g, ws, *p, w, st);
      case 2: return launch_wg2<2>(x, g, ws, *p, w, st);
      default: return launch_wg2<3>(x, g, ws, *p, w, st);
    }
  }
  switch (w.nsub) {
    case 1: return launch_wg<1>(x, g, ws, *p, w, st);
    case 2: return launch_wg<2>(x, g, ws, *p, w, st);
    default: return launch_wg<3>(x, g, ws, *p, w, st);
  }
}
