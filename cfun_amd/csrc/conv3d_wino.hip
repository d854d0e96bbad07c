// 3x3x3 stride-1 convolution with the x axis in the Winograd F(2,3) domain, on the fp32 matrix cores.
//
// Along x every pair of outputs is computed from 4 transformed inputs and 4 transformed taps instead of 2 x 3
// products (Lavin & Gray's minimal filtering, 1-D): the implicit GEMM runs 9 (dz,dy) x 4 points instead of 27 taps
// per two output columns, i.e. 2/3 of the MFMAs of k_conv_mfma for the same result.  z and y stay direct.
//   input transform  (while staging a chunk of 4 channels into LDS):  v = (d0-d2, d1+d2, d2-d1, d1-d3)
//   weight transform (k_wino_weights, once per launch, from the packed [tap][Ci][CoP] weights):
//                    u = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//   output transform (registers, before the fused epilogue):          y_even = m0+m1+m2,  y_odd = m1-m2-m3
// All coefficients are 0, +-1, 1/2: the result differs from the direct sum only by fp32 rounding order (measured in
// tests/kernel_cases.py against the fp64 oracle next to the direct kernel).
//
//   block  = 256 threads (4 waves) -> 4(z) x 4(y) x 16(x) output voxels x (16*NSUB) output channels (same tile,
//            raster and XCD remap as k_conv_mfma)
//   wave w = z-plane w; MFMA columns = 2 rows x 8 x-pairs, 2 row groups; accumulators [2 row groups][4 points][NSUB]
//   LDS    = V [4 ch][6 z][6 y][8 pairs] x float4 (the 4 points) + U [9 (dz,dy)][4 ch][16*NSUB] x float4: a fragment
//            read is one ds_read_b128 per (dz,dy) for all 4 points (8 consecutive lanes = 128 contiguous bytes)
#include <stdlib.h>

#include <utility>

#include "conv3d_mfma.h"

namespace {

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

using cfun_mfma::cdiv;

constexpr int TD = 4, TH = 4, TW = 16, IZ = 6, IY = 6, NPAIR = 8;
constexpr int VPLANE4 = IZ * IY * NPAIR;      // float4 (= the 4 points of one x-pair) per channel of the halo tile

// u[r9][ci][co][point] from wp[tap = r9*3 + dx][ci][co]  (flip: the data gradient reads tap 26 - t)
__global__ void __launch_bounds__(256)
k_wino_weights(const float* __restrict__ wp, float4* __restrict__ u, int Ci, int CoP, int flip) {
  const int64_t per = (int64_t)Ci * CoP, total = 9 * per;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int r9 = (int)(i / per);
  const int64_t e = i - r9 * per;
  float g[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int tap = r9 * 3 + dx;
    g[dx] = wp[(int64_t)(flip ? 26 - tap : tap) * per + e];
  }
  u[i] = make_float4(g[0], 0.5f * ((g[0] + g[2]) + g[1]), 0.5f * ((g[0] + g[2]) - g[1]), g[2]);
}

// TWOD: u2[(dz*4 + py)][ci][co][px] = (G g G^T)[py][px] of the 3x3 (ky,kx) slice dz
__global__ void __launch_bounds__(256)
k_wino2_weights(const float* __restrict__ wp, float4* __restrict__ u, int Ci, int CoP, int flip) {
  const int64_t per = (int64_t)Ci * CoP, total = 3 * per;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int dz = (int)(i / per);
  const int64_t e = i - dz * per;
  float t[4][3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    float g[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int tap = (dz * 3 + ky) * 3 + kx;
      g[ky] = wp[(int64_t)(flip ? 26 - tap : tap) * per + e];
    }
    t[0][kx] = g[0]; t[1][kx] = 0.5f * ((g[0] + g[2]) + g[1]); t[2][kx] = 0.5f * ((g[0] + g[2]) - g[1]); t[3][kx] = g[2];
  }
#pragma unroll
  for (int py = 0; py < 4; ++py)
    u[(int64_t)(dz * 4 + py) * per + e] = make_float4(t[py][0], 0.5f * ((t[py][0] + t[py][2]) + t[py][1]),
                                                     0.5f * ((t[py][0] + t[py][2]) - t[py][1]), t[py][2]);
}

// S2D (data gradient of a depth-to-space conv): the logical input [N,Di,Hi,Wi,8*cq] is gathered from the hi-res gradient
// [N,2Di,2Hi,2Wi,cq]; chunk c = 4 channels o4 of parity q = c / (cq/4), read at hi-res voxel 2*(z,y,x) + q.
// TWOD: y is in the Winograd domain as well (F(2x2,3x3) per z tap): 3 (dz) x 16 points instead of 27 taps per 2x2 outputs =
// 4/9 of the MFMAs.  Staging is unchanged (rows stay x-transformed in LDS); the y transform of the input is applied when
// the fragment is read (two rows, one add per point), MFMA columns are the 2 x 8 tiles of the wave's 4 x 16 plane and the
// accumulators are [4 py][4 px][NSUB].
// SB (TWOD only): one LDS buffer and two barriers per chunk like the 1-D loop, and registers capped for two waves per
// SIMD at NSUB = 2 (128 + 128): the second resident workgroup hides the staging instead of the second buffer.
template <int NSUB, bool S2D, bool TWOD, bool STATS = false, bool SB = false>
__global__ void __launch_bounds__(256, (TWOD && SB) ? (NSUB == 1 ? 3 : 2) : 1)
k_conv_wino(const float* __restrict__ x, const float4* __restrict__ u, const float* __restrict__ scale,
            const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y, CfunConv3dParams p,
            int ntz, int nty, int ntx, int ncot, float* __restrict__ partial, int chunks_per_split, int s2d_cq,
            cfun_mfma::ConvMode md) {
  constexpr int NT = 16 * NSUB;
  constexpr int UROWS = TWOD ? 12 : 9;        // (dz,py) or (dz,dy) groups of 4 channel rows
  constexpr int W_ITEMS = UROWS * 4 * NT;     // float4 (= 4 x-points of one output channel) items per chunk
  constexpr int W_LOADS = cdiv(W_ITEMS, 256);
  CFUN_DYN_LDS(float4, smem);
  float4* Vl = smem;                          // [4 ch][36 rows][8 pairs] x 4 points
  float4* Ul = smem + 4 * VPLANE4;            // [9 (dz,dy)][4 ch][NT] x 4 points

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned lid = cfun_mfma::xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot; lid /= ncot;
  const unsigned per_n = (unsigned)(ntz * nty * ntx);
  const int n = lid / per_n;
  int tz, ty, tx;
  cfun_mfma::tile_raster(lid - (unsigned)n * per_n, ntz, nty, ntx, tz, ty, tx);
  const int z0 = tz * TD, y0 = ty * TH, x0 = tx * TW;
  const int cobase = cot * NT;

  // ---- staging descriptors.  X item = (halo row r = (z, y), column pair jj of 9): the two voxels x = 2jj, 2jj+1 for TWO
  // consecutive channel chunks (each 16-byte piece; both come out of the same 64-byte sector, so issued back to back
  // the second merges with the first's miss instead of fetching the sector from L2 again one chunk later -- the tile's
  // 104 KB footprint does not survive in the L1 between chunks).  A wave owns 7 whole rows per pass (lane = 9*row + jj,
  // lane 63 idle), so the two voxels an x-pair needs from the next column pair are one lane away (__shfl).  Loads are
  // unconditional (out-of-volume voxels read element 0) and the zero padding is applied in commit(): selects at load time
  // made hipcc wrap every load in a branch.
  constexpr int X_PASSES = 2;                   // 4 waves x 7 rows x 2 passes = 56 >= 36 halo rows
  const int xrow0 = wv * 7 + lane / 9, xjj = lane % 9;
  const float* in_ptr[X_PASSES][2];
  unsigned in_ok = 0;
#pragma unroll
  for (int i = 0; i < X_PASSES; ++i) {
    const int r = i * 28 + xrow0;
#pragma unroll
    for (int k = 0; k < 2; ++k) in_ptr[i][k] = x;
    if (lane < 63 && r < IZ * IY) {
      const int vz = z0 - p.pd + r / IY, vy = y0 - p.ph + r % IY, vx = x0 - p.pw + 2 * xjj;
      if (vz >= 0 && vz < p.Di && vy >= 0 && vy < p.Hi) {
        const int64_t row = (((int64_t)n * p.Di + vz) * p.Hi + vy) * p.Wi;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (vx + k >= 0 && vx + k < p.Wi) {
            in_ok |= 1u << (i * 2 + k);
            in_ptr[i][k] = S2D ? x + ((((int64_t)n * 2 * p.Di + 2 * vz) * 2 * p.Hi + 2 * vy) * 2 * p.Wi + 2 * (vx + k)) * s2d_cq
                               : x + (row + vx + k) * p.Ci;
          }
      }
    }
  }
  int w_off[W_LOADS];      // float4 index into u for chunk 0, or -1
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    const int it = tid + i * 256, row = it / NT, co = it % NT;
    // items past the tile's rows / the padded width re-read a valid element: their LDS slot is not written / their
    // output channels are never stored, so no select is needed (selects made hipcc wrap every load in a branch)
    w_off[i] = (it < W_ITEMS && cobase + co < p.CoP) ? (((row >> 2) * p.Ci + (row & 3)) * p.CoP + cobase + co) : 0;
  }
  const int w_step = 4 * p.CoP;
  float4 xp[X_PASSES][2][2], win[W_LOADS];      // [pass][voxel][chunk of the pair]
  const int cpq = S2D ? (s2d_cq >> 2) : 1;       // chunks per parity
  auto chunk_xoff = [&](int c) -> int64_t {
    if (!S2D) return (int64_t)c * 4;
    const int q = c / cpq, o4 = c - q * cpq;
    return ((int64_t)((q >> 2) * 2 * p.Hi + ((q >> 1) & 1)) * 2 * p.Wi + (q & 1)) * s2d_cq + o4 * 4;
  };
  auto prefetch_x = [&](int c, int c1) {          // chunks c and c1 (= c + 1, or c again at the end) of this thread's voxels
    const int64_t xo = chunk_xoff(c), xo1 = chunk_xoff(c1);
#pragma unroll
    for (int i = 0; i < X_PASSES; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(in_ptr[i][k] + xo);
        const float4 w = *reinterpret_cast<const float4*>(in_ptr[i][k] + xo1);
        xp[i][k][0] = make_float4(v.x, v.y, v.z, v.w);
        xp[i][k][1] = make_float4(w.x, w.y, w.z, w.w);
      }
  };
  auto prefetch_w = [&](int c) {
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {     // (columns >= CoP: never stored)
      const float4 v = u[w_off[i] + c * w_step];
      win[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  constexpr int LBUF = 4 * VPLANE4 + UROWS * 4 * NT;      // float4 per LDS buffer (TWOD runs two of them)
  auto commit_x = [&](float4* Vl, int half) {      // half: which chunk of the prefetched pair
#pragma unroll
    for (int i = 0; i < X_PASSES; ++i) {
      float d[4][4];                                // [voxel 0..3 of the x-pair][channel]
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool ok = (in_ok >> (i * 2 + k)) & 1u;
        const float4 lo = xp[i][k][0], hi = xp[i][k][1];
        d[k][0] = ok ? (half ? hi.x : lo.x) : 0.f; d[k][1] = ok ? (half ? hi.y : lo.y) : 0.f;
        d[k][2] = ok ? (half ? hi.z : lo.z) : 0.f; d[k][3] = ok ? (half ? hi.w : lo.w) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) d[2 + k][cc] = __shfl(d[k][cc], (lane + 1) & 63, 64);     // voxels 2jj+2, 2jj+3
      const int r = i * 28 + xrow0;
      if (lane < 63 && r < IZ * IY && xjj < NPAIR) {
        float4* v = Vl + r * NPAIR + xjj;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          v[cc * VPLANE4] = make_float4(d[0][cc] - d[2][cc], d[1][cc] + d[2][cc], d[2][cc] - d[1][cc], d[1][cc] - d[3][cc]);
      }
    }
  };
  auto commit_w = [&](float4* Ul) {
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < W_ITEMS) Ul[it] = win[i];
    }
  };
  auto commit = [&](int buf, int half) {
    commit_x(smem + buf * LBUF, half);
    commit_w(smem + buf * LBUF + 4 * VPLANE4);
  };

  constexpr int NMG = TWOD ? 4 : 2;           // row groups (1-D) / y points (2-D)
  f32x4 acc[NMG][4][NSUB];
#pragma unroll
  for (int mg = 0; mg < NMG; ++mg)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) acc[mg][pt][nn] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment bases (float4 = the 4 points): B column (lane & 15) = (row (lane>>3)&1 of the row group, pair lane&7),
  // k = channel lane>>4;  A row (lane & 15) = output channel
  const float4* Vw0 = Vl + (lane >> 4) * VPLANE4 + (wv * IY + (TWOD ? 2 : 1) * ((lane >> 3) & 1)) * NPAIR + (lane & 7);
  const float4* Uw0 = Ul + (lane >> 4) * NT + (lane & 15);

  const int nchunks = p.Ci >> 2;
  const int c_begin = blockIdx.y * chunks_per_split;
  const int c_end = (c_begin + chunks_per_split < nchunks) ? c_begin + chunks_per_split : nchunks;
  if constexpr (TWOD) {
    auto twod_phase = [&](int voff, int uoff) {      // float4 offsets of the V / U buffers to read
      const float4* Vw = Vw0 + voff;
      const float4* Uw = Uw0 + uoff;
      // 12 steps (dz,py): the 4 x-transformed rows of the next dz and the next step's weight fragment are in flight
      // under the current step's 4*NSUB MFMAs
      float4 rows[2][4], a2[2][NSUB];
      auto load_rows = [&](int dz, float4 (&r)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = Vw[(dz * IY + k) * NPAIR];
      };
      auto load_a = [&](int st, float4 (&a)[NSUB]) {
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn) a[nn] = Uw[st * 4 * NT + nn * 16];
      };
      load_rows(0, rows[0]);
      load_a(0, a2[0]);
#ifndef CFUN_HIP_EMULATION
      __builtin_amdgcn_sched_group_barrier(0x100, 4 + NSUB, 0);
#endif
      auto step = [&](auto s_) {
        constexpr int st = decltype(s_)::value, dz = st / 4, py = st % 4;
        const float4(&r)[4] = rows[dz & 1];
        // B^T d along y: (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
        constexpr int r1 = py == 0 ? 0 : py == 2 ? 2 : 1, r2 = py == 0 ? 2 : py == 1 ? 2 : py == 2 ? 1 : 3;
        constexpr float sg = py == 1 ? 1.f : -1.f;
        const float bv[4] = {r[r1].x + sg * r[r2].x, r[r1].y + sg * r[r2].y, r[r1].z + sg * r[r2].z, r[r1].w + sg * r[r2].w};
        float av[NSUB][4];
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn) {
          const float4 t = a2[st & 1][nn];
          av[nn][0] = t.x; av[nn][1] = t.y; av[nn][2] = t.z; av[nn][3] = t.w;
        }
        if (st + 1 < 12) load_a(st + 1, a2[(st + 1) & 1]);
        if (py == 3 && dz < 2) load_rows(dz + 1, rows[(dz + 1) & 1]);
#ifdef CFUN_HIP_EMULATION
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
          for (int nn = 0; nn < NSUB; ++nn)
            acc[py][px][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nn][px], bv[px], acc[py][px][nn], 0, 0, 0);
#else
        // The 16 * NSUB accumulators must live in AGPRs (they do not fit beside the staging registers in the 256
        // architectural VGPRs); left to itself hipcc splits their live ranges across both files and copies them around
        // every chunk (848 v_accvgpr moves in the loop).  The "+a" constraint pins them; the phases are fenced by hand.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");          // VALU-written B operand -> MFMA read
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
          for (int nn = 0; nn < NSUB; ++nn)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[py][px][nn]) : "v"(av[nn][px]), "v"(bv[px]));
        __builtin_amdgcn_sched_barrier(0);
#endif
      };
      static_for(std::make_integer_sequence<int, 12>{}, step);
    };
    if constexpr (SB) {
      // a chunk's MFMA phase is short here (48 * NSUB MFMAs): BOTH chunks of the register pair are transformed into LDS at
      // once (two V buffers, one U buffer), so the next pair's loads are in flight under two MFMA phases instead of one.
      // LDS = [V0][V1][U]; the second chunk's U is committed between the phases (two more barriers, 6 stores)
      float4* const V1 = smem + 4 * VPLANE4;
      float4* const Ub = smem + 8 * VPLANE4;
      if (c_begin < c_end) { prefetch_x(c_begin, c_begin + 1 < c_end ? c_begin + 1 : c_begin); prefetch_w(c_begin); }
      for (int c = c_begin; c < c_end; c += 2) {
        __syncthreads();
        commit_x(smem, 0);
        if (c + 1 < c_end) commit_x(V1, 1);
        commit_w(Ub);
        __syncthreads();
        if (c + 2 < c_end) prefetch_x(c + 2, c + 3 < c_end ? c + 3 : c + 2);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          if (c + h >= c_end) break;
          if (h) {
            __syncthreads();
            commit_w(Ub);
            __syncthreads();
          }
          if (c + h + 1 < c_end) prefetch_w(c + h + 1);
          twod_phase(h * 4 * VPLANE4, 4 * VPLANE4);
        }
      }
    } else {
      // without SB, TWOD runs one wave per SIMD (192 accumulator registers at NSUB = 3), so nothing else hides the staging:
      // LDS is double buffered (2 x 55 KB) -- the next chunk is committed to the other buffer after this chunk's MFMAs
      // were issued, one barrier per chunk
      if (c_begin < c_end) {
        prefetch_x(c_begin, c_begin + 1 < c_end ? c_begin + 1 : c_begin);
        prefetch_w(c_begin);
        commit(0, 0);
        __syncthreads();
      }
      for (int c = c_begin; c < c_end; ++c) {
        if (c + 1 < c_end) prefetch_w(c + 1);
        twod_phase(((c - c_begin) & 1) * LBUF, ((c - c_begin) & 1) * LBUF);
        if (c + 1 < c_end) {
          const int half = (c + 1 - c_begin) & 1;
          commit(((c - c_begin) & 1) ^ 1, half);
          if (half && c + 2 < c_end) prefetch_x(c + 2, c + 3 < c_end ? c + 3 : c + 2);     // both halves consumed: next pair
        }
        __syncthreads();
      }
    }
  } else {
    auto mfma_phase = [&]() {
      const float4* Vw = Vw0;
      const float4* Uw = Uw0;
    float4 a[NSUB], b[2];
      auto frag = [&](int r9) {
        const int dz = r9 / 3, dy = r9 - dz * 3;
  #pragma unroll
        for (int nn = 0; nn < NSUB; ++nn) a[nn] = Uw[r9 * 4 * NT + nn * 16];
  #pragma unroll
        for (int mg = 0; mg < 2; ++mg) b[mg] = Vw[(dz * IY + mg * 2 + dy) * NPAIR];
      };
      frag(0);
  #ifndef CFUN_HIP_EMULATION
      __builtin_amdgcn_sched_group_barrier(0x100, NSUB + 2, 0);
  #endif
  #pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) {
        float av[NSUB][4], bv[2][4];
  #pragma unroll
        for (int nn = 0; nn < NSUB; ++nn) { av[nn][0] = a[nn].x; av[nn][1] = a[nn].y; av[nn][2] = a[nn].z; av[nn][3] = a[nn].w; }
  #pragma unroll
        for (int mg = 0; mg < 2; ++mg) { bv[mg][0] = b[mg].x; bv[mg][1] = b[mg].y; bv[mg][2] = b[mg].z; bv[mg][3] = b[mg].w; }
        if (r9 + 1 < 9) frag(r9 + 1);      // the next (dz,dy)'s fragments are in flight under this one's MFMAs
  #pragma unroll
        for (int pt = 0; pt < 4; ++pt)
  #pragma unroll
          for (int mg = 0; mg < 2; ++mg)
  #pragma unroll
            for (int nn = 0; nn < NSUB; ++nn)
              acc[mg][pt][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nn][pt], bv[mg][pt], acc[mg][pt][nn], 0, 0, 0);
  #ifndef CFUN_HIP_EMULATION
        // keep that order in the schedule: the LDS reads first, then the MFMA block that hides their latency
        if (r9 + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, NSUB + 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NSUB, 0);
  #endif
      }
    };
    if (c_begin < c_end) { prefetch_x(c_begin, c_begin + 1 < c_end ? c_begin + 1 : c_begin); prefetch_w(c_begin); }
    for (int c = c_begin; c < c_end; ++c) {
      __syncthreads();
      const int half = (c - c_begin) & 1;
      commit(0, half);
      __syncthreads();
      if (c + 1 < c_end) prefetch_w(c + 1);
      if (half && c + 1 < c_end) prefetch_x(c + 1, c + 2 < c_end ? c + 2 : c + 1);        // both halves consumed: next pair
      mfma_phase();
    }
  }

  // ---- output transform + epilogue: lane owns voxels (z0+wv, y0+2mg+r, x0+2j+{0,1}), channels nn*16+(lane>>4)*4..+3
  const int oz = z0 + wv, oxe = x0 + 2 * (lane & 7);
  constexpr bool stats_on = STATS;      // the per-tile sums of y, y*y per channel (md.out_part): its own instantiation
  if (!stats_on && oz >= p.Do) return;
  float* red = reinterpret_cast<float*>(smem);
  if (stats_on) __syncthreads();      // every wave has left the main loop: the LDS tiles are dead, smem becomes `red`
  auto emit = [&](int oy, int ox, int co, const f32x4& a4, float (&sa)[4], float (&sb)[4]) {
    if (oz >= p.Do || oy >= p.Ho || ox >= p.Wo || co >= p.Co) return;
    const int64_t v = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    float4 r = make_float4(a4[0], a4[1], a4[2], a4[3]);
    if (gridDim.y > 1) {
      *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.y * p.N * p.Do * p.Ho * p.Wo + v) * p.Co + co) = r;
      return;
    }
    if (p.scale_mode) {
      const float4 s4 = *reinterpret_cast<const float4*>(scale + (p.scale_mode == 2 ? n * p.Co : 0) + co);
      r.x *= s4.x; r.y *= s4.y; r.z *= s4.z; r.w *= s4.w;
    }
    if (p.has_shift) {
      const float4 t = *reinterpret_cast<const float4*>(shift + co);
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    // depth-to-space (parity-folded up-convs): channel co = parity q * CqP + oc goes to hi-res voxel 2*(oz,oy,ox) + q of a
    // tensor with Cq channels; the residual is read at the low-res voxel
    const int CqP = p.Co >> 3, Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;
    const int q = p.d2s ? co / CqP : 0, oc = p.d2s ? co - q * CqP : co;
    if (p.d2s && oc >= Cq) return;
    if (p.res_mode) {
      const float4 t = *reinterpret_cast<const float4*>(res + v * (p.d2s ? Cq : p.Co) + oc);
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    r.x = cfun_apply_act(r.x, p.act, p.slope); r.y = cfun_apply_act(r.y, p.act, p.slope);
    r.z = cfun_apply_act(r.z, p.act, p.slope); r.w = cfun_apply_act(r.w, p.act, p.slope);
    if (stats_on) {
      sa[0] += r.x; sa[1] += r.y; sa[2] += r.z; sa[3] += r.w;
      sb[0] += r.x * r.x; sb[1] += r.y * r.y; sb[2] += r.z * r.z; sb[3] += r.w * r.w;
    }
    if (p.d2s) {
      const int64_t hv = (((int64_t)n * 2 * p.Do + 2 * oz + (q >> 2)) * 2 * p.Ho + 2 * oy + ((q >> 1) & 1)) * 2 * p.Wo +
                         2 * ox + (q & 1);
      *reinterpret_cast<float4*>(y + hv * Cq + oc) = r;
    } else {
      *reinterpret_cast<float4*>(y + v * p.Co + co) = r;
    }
  };
  constexpr int SR = cfun_mfma::STAT_ROUND, ES = cfun_mfma::stat_es(NT < SR ? NT : SR);
  const int tile = (int)(lid - (unsigned)n * per_n);
  auto park = [&](int nn, const float (&sa)[4], const float (&sb)[4]) {      // statistics rounds: windows of SR channels
    const int base = (nn * 16 / SR) * SR, end = (nn + 1) * 16;
    cfun_mfma::quad_park_16(sa, sb, red, ES, wv, lane, nn * 16 - base);
    if (end % SR == 0 || end == NT) cfun_mfma::stat_round_flush(red, ES, tid, base, end - base, cobase, n, tile, p, md);
  };
  if constexpr (TWOD) {     // Y = A^T M A: lane owns the 2 x 2 outputs of tile (ty = (lane>>3)&1, j = lane&7)
    const int oy = y0 + 2 * ((lane >> 3) & 1);
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) {
      float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
      f32x4 e[4], o[4];
#pragma unroll
      for (int py = 0; py < 4; ++py) {
        e[py] = (acc[py][0][nn] + acc[py][1][nn]) + acc[py][2][nn];
        o[py] = (acc[py][1][nn] - acc[py][2][nn]) - acc[py][3][nn];
      }
      const int co = cobase + nn * 16 + (lane >> 4) * 4;
      emit(oy, oxe, co, (e[0] + e[1]) + e[2], sa, sb);
      emit(oy, oxe + 1, co, (o[0] + o[1]) + o[2], sa, sb);
      emit(oy + 1, oxe, co, (e[1] - e[2]) - e[3], sa, sb);
      emit(oy + 1, oxe + 1, co, (o[1] - o[2]) - o[3], sa, sb);
      if constexpr (stats_on) park(nn, sa, sb);
    }
  } else {
    // voxel by voxel, all channel subtiles of a voxel back to back: the 1-D kernel runs the chip-filling launches (96^3),
    // where an L2 line that has received only one subtile's 64 bytes does not survive until the next subtile's stores a
    // few hundred instructions later -- WRITE_SIZE was 1.43 x the output bytes with the subtile loop outermost
    // (profiles/round4_pmc_write_probe.txt; 1.02 - 1.04 x on the 48^3 launches either way)
    float sa[NSUB][4], sb[NSUB][4];
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn)
#pragma unroll
      for (int j = 0; j < 4; ++j) { sa[nn][j] = 0.f; sb[nn][j] = 0.f; }
#pragma unroll
    for (int mg = 0; mg < 2; ++mg) {
      const int oy = y0 + mg * 2 + ((lane >> 3) & 1);
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn)
        emit(oy, oxe, cobase + nn * 16 + (lane >> 4) * 4, (acc[mg][0][nn] + acc[mg][1][nn]) + acc[mg][2][nn], sa[nn], sb[nn]);
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn)
        emit(oy, oxe + 1, cobase + nn * 16 + (lane >> 4) * 4, (acc[mg][1][nn] - acc[mg][2][nn]) - acc[mg][3][nn], sa[nn], sb[nn]);
    }
    if constexpr (stats_on) {
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) park(nn, sa[nn], sb[nn]);
    }
  }
}

// 16-channel subtiles per block: fewest padded channels, widest on ties, at most 3 (two waves per SIMD: 134 VGPR + 96
// accumulators).  80 channels = 5 tiles of 16 measured as fast as one 80-wide tile at one wave per SIMD on 48^3 and 10 %
// faster on 24^3 (tools/bench_layers.py, profiles/round2_ab_layers.log)
int wino_nsub(int co, int twod) {
  static int mx = 0;      // tuning knob: CFUN_WINO_MAX_NSUB
  if (!mx) {
    const char* e = getenv("CFUN_WINO_MAX_NSUB");
    mx = e ? atoi(e) : -1;
    if (mx == 0 || mx > 5) mx = -1;
  }
  const int cap = (mx < 0 || twod) ? 3 : mx;
  int best = 1, best_pad = 1 << 30;
  for (int n = 1; n <= cap; ++n) {
    const int nt = 16 * n, pad = (co + nt - 1) / nt * nt;
    if (pad <= best_pad) { best_pad = pad; best = n; }
  }
  return best;
}

// y in the Winograd domain as well?  CFUN_WINO_2D: 0 = never, 1 = every supported shape, unset = measured shapes:
// round 3 (tools/bench_layers.py, profiles/round3_layers_wino_1d_vs_2d.log): at one wave per SIMD the 2-D kernel loses
// 0.6 % on the chip-filling 40 -> 40 @ 4 x 96^3 launch but wins 2 - 11 % forward and data gradient on everything at
// 48^3 and below (80 -> 80 @ 48^3: 1.02 -> 0.92 ms, @ 24^3: 0.192 -> 0.170 ms; the folded 5^3 conv's data gradient 1.31
// -> 1.22 ms), where fewer MFMAs per tile matter more than the second resident wave -- so AUTO takes it up to 2^19
// output voxels per launch, and for single-tile outputs of any size.
static int env_knob(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : -1;
}

int wino_2d(const CfunConv3dParams& p) {
  static int knob = -2;
  if (knob == -2) {
    const char* e = getenv("CFUN_WINO_2D");
    knob = e ? atoi(e) : -1;
  }
  if (p.algo == CFUN_ALGO_WINO) return 0;       // tests: the 1-D kernel
  if (p.algo == CFUN_ALGO_WINO2 || knob == 1) return 1;
  if (knob == 0) return 0;
  // (C_out <= 16 -- the folded 5^3 conv's data gradient -- runs one co tile: 64 accumulators, three waves per SIMD either way)
  return p.Co <= 16 || (int64_t)p.N * p.Do * p.Ho * p.Wo <= ((int64_t)1 << 19);
}

struct Plan {
  int nsub, twod, sb, ntz, nty, ntx, ncot, ksplit, cps;
  int64_t nblk;
  size_t u_bytes, part_bytes;
};

Plan make_plan(const CfunConv3dParams& p, size_t ws_for_partials) {
  static const int sb_knob = env_knob("CFUN_WINO_SB");      // 0: the double-buffered one-wave-per-SIMD loop for every NSUB
  Plan w;
  w.twod = wino_2d(p);
  w.nsub = wino_nsub(p.Co, w.twod);
  // 2-D tiles of 16 / 32 channels run two (three) waves per SIMD on the single-U-buffer loop (k_conv_wino's SB).
  // (Measured and dropped, round 3: C_out = 40 as a 32-wide launch at two waves per SIMD plus a 16-wide one for the last
  // 8 channels instead of 48-wide tiles at one -- 5 % faster than the 1-D kernel on 4 x 96^3 (2.14 vs 2.27 ms), but the
  // second launch stages the whole input again for 8 channels: 3.2 GB through the L2's fabric side per call instead of
  // 1.5 GB; both in ONE launch, the 16-wide workgroup next to the 32-wide one of the same voxels, measured 2.5 ms.)
  w.sb = w.twod && w.nsub <= 2 && sb_knob != 0;
  const int nt = 16 * w.nsub;
  w.ntz = cdiv(p.Do, TD); w.nty = cdiv(p.Ho, TH); w.ntx = cdiv(p.Wo, TW); w.ncot = cdiv(p.Co, nt);
  w.nblk = (int64_t)p.N * w.ntz * w.nty * w.ntx * w.ncot;
  w.u_bytes = cfun_align_up((size_t)(w.twod ? 48 : 36) * p.Ci * p.CoP * sizeof(float), 256);
  w.ksplit = cfun_mfma::splitk_factor(w.nblk, p.Ci >> 2, p, ws_for_partials);
  w.cps = cdiv(p.Ci >> 2, w.ksplit);
  w.part_bytes = w.ksplit > 1 ? (size_t)w.ksplit * p.N * p.Do * p.Ho * p.Wo * p.Co * sizeof(float) : 0;
  return w;
}

template <int NSUB>
int launch(const float* x, const float4* u, const float* scale, const float* shift, const float* res, float* y,
           const CfunConv3dParams& p, const Plan& w, float* partial, int s2d_cq, const cfun_mfma::ConvMode& md, hipStream_t st) {
  const bool sb = w.twod && w.sb && NSUB <= 2;
  const size_t lds = sb ? (size_t)(8 * VPLANE4 + 48 * 16 * NSUB) * sizeof(float4)      // [V0][V1][U]
                        : (size_t)(w.twod ? 2 : 1) * (4 * VPLANE4 + (w.twod ? 48 : 36) * 16 * NSUB) * sizeof(float4);
  auto kern = s2d_cq ? k_conv_wino<NSUB, true, false> : k_conv_wino<NSUB, false, false>;
  if constexpr (NSUB <= 3) {     // 16 accumulator sets per wave: 64 * NSUB registers
    if (w.twod) kern = s2d_cq ? k_conv_wino<NSUB, true, true> : k_conv_wino<NSUB, false, true>;
  } else if (w.twod) {
    return CFUN_EINVAL;
  }
  if constexpr (NSUB <= 2) {
    if (sb) kern = s2d_cq ? k_conv_wino<NSUB, true, true, false, true> : k_conv_wino<NSUB, false, true, false, true>;
  }
  if (md.out_part) {             // epilogue statistics (forward only: never with the s2d gather)
    if (s2d_cq) return CFUN_EINVAL;
    kern = k_conv_wino<NSUB, false, false, true>;
    if constexpr (NSUB <= 3) {
      if (w.twod) kern = k_conv_wino<NSUB, false, true, true>;
    }
    if constexpr (NSUB <= 2) {
      if (sb) kern = k_conv_wino<NSUB, false, true, true, true>;
    }
  }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)w.nblk, (unsigned)w.ksplit), dim3(256), lds, st, x, u, scale, shift, res, y, p,
                     w.ntz, w.nty, w.ntx, w.ncot, partial, w.cps, s2d_cq, md);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // namespace

// ---- entry points used by conv3d.hip

// CFUN_WINO: 0 = never, 1 = every supported shape, unset = shapes where it measured faster (tools/bench_layers.py)
static int wino_knob() {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("CFUN_WINO");
    v = e ? atoi(e) : -1;
  }
  return v;
}

// can the Winograd kernels run this conv at all (forward; the weight gradient -- conv3d_wino_wgrad.hip -- adds its own limits)
int cfun_wino_shape_ok(const CfunConv3dParams* p) {
  if (wino_knob() == 0 || p->algo == CFUN_ALGO_DIRECT || p->algo == CFUN_ALGO_MFMA) return 0;
  // depth padding 0 / 1 / 2: depth-sharded slabs arrive with their halo planes (pd = 0; their data gradient has pd = 2)
  if (p->kd != 3 || p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pd < 0 || p->pd > 2 || p->ph != 1 || p->pw != 1) return 0;
  if (p->up2 || p->tap_skip || (p->res_up2 && !p->d2s) || (p->Ci & 3) || (p->Co & 3)) return 0;
  if (p->d2s) {     // a lane's float4 stays inside one parity group; split-K partials have no depth-to-space finish
    const int cqp = p->Co >> 3, cq = p->d2s_cq > 0 ? p->d2s_cq : cqp;
    if ((cqp & 3) || (cq & 3)) return 0;
  }
  if (p->Do != p->Di + 2 * p->pd - 2 || p->Ho != p->Hi || p->Wo != p->Wi) return 0;
  return (int64_t)p->N * p->Do * p->Ho * p->Wo > 0;
}

int cfun_wino_supported(const CfunConv3dParams* p) {
  if (!cfun_wino_shape_ok(p)) return 0;
  if (wino_knob() == 1 || p->algo == CFUN_ALGO_WINO || p->algo == CFUN_ALGO_WINO2) return 1;
  // (the folded 5x5x5 'finetune' conv -- d2s, C_in = 8 -- has two channel chunks and is bound by its stores: no gain measured)
  return p->Co >= 32 && p->Ci >= 16 && !p->d2s;
}

// data gradient of a depth-to-space conv p (no tap skipping, no per-parity channel padding) as the Winograd conv q over
// the gathered hi-res gradient
int cfun_wino_s2d_dgrad_supported(const CfunConv3dParams* p, const CfunConv3dParams* q) {
  if (wino_knob() == 0 || p->algo == CFUN_ALGO_DIRECT || p->algo == CFUN_ALGO_MFMA) return 0;
  if (!p->d2s || p->tap_skip || p->up2) return 0;
  const int cqp = p->Co >> 3, cq = p->d2s_cq > 0 ? p->d2s_cq : cqp;
  if (cq != cqp || (cq & 3)) return 0;
  if (q->kd != 3 || q->kh != 3 || q->kw != 3 || q->stride != 1 || q->pd < 0 || q->pd > 2 || q->ph != 1 || q->pw != 1) return 0;
  if (q->Do != q->Di + 2 * q->pd - 2 || q->Ho != q->Hi || q->Wo != q->Wi || (q->Ci & 3) || (q->Co & 3)) return 0;
  return (int64_t)q->N * q->Do * q->Ho * q->Wo > 0;
}

extern "C" int cfun_conv3d_wino_plan(const CfunConv3dParams* p, int32_t out[4]) {
  if (!p || !out || !cfun_wino_supported(p)) return CFUN_EINVAL;
  const Plan w = make_plan(*p, (size_t)-1);
  out[0] = w.twod; out[1] = w.nsub; out[2] = w.sb;
  out[3] = w.ncot * 16 * w.nsub;
  return CFUN_OK;
}

size_t cfun_wino_workspace_bytes(const CfunConv3dParams* p) {
  const Plan w = make_plan(*p, (size_t)-1);
  return w.u_bytes + cfun_align_up(w.part_bytes, 256);
}

// wp: packed weights [27][Ci][CoP] of the conv that is run (the data gradient passes the transposed pack and flip = 1)
// s2d_cq > 0: x is the hi-res gradient of a depth-to-space conv with s2d_cq channels, p->Ci = 8 * s2d_cq (see k_conv_wino)
// statistics slots per sample a launch with ws_bytes of workspace fills (fz->out_part)
int cfun_wino_stat_slots(const CfunConv3dParams* p, size_t ws_bytes) {
  Plan w = make_plan(*p, 0);
  if (ws_bytes < w.u_bytes) return 0;
  w = make_plan(*p, ws_bytes - w.u_bytes);
  return w.ksplit > 1 ? -cfun_splitk_stat_slots(p) : w.ntz * w.nty * w.ntx * (p->d2s ? 8 : 1);
}

// y in the Winograd domain as well for conv p (which transform cfun_wino_fwd applies / expects prepared)
int cfun_wino_is_2d(const CfunConv3dParams* p) { return wino_2d(*p); }

// prepared: wp already is the transformed U this launch reads (cfun_weight_prepare: CFUN_WOP_WINO1/2 or their _T forms)
int cfun_wino_fwd(const float* x, const float* wp, int flip, int s2d_cq, const float* scale, const float* shift,
                  const float* res, float* y, const CfunConv3dParams* p, void* ws, size_t ws_bytes,
                  const cfun_mfma::ConvMode* fz, int prepared, hipStream_t st) {
  Plan w = make_plan(*p, 0);
  if (!ws || ws_bytes < w.u_bytes) return CFUN_EWORKSPACE;
  w = make_plan(*p, ws_bytes - w.u_bytes);
  if (w.nblk > 0x7fffffffLL) return CFUN_EINVAL;
  const float4* u = prepared ? (const float4*)wp : (const float4*)ws;
  float* partial = (float*)((char*)ws + w.u_bytes);
  const int64_t nu = (int64_t)9 * p->Ci * p->CoP;
  if (prepared) {
    if (!cfun_aligned16(wp)) return CFUN_EALIGN;
  } else {
    if (w.twod)
      hipLaunchKernelGGL(k_wino2_weights, dim3((unsigned)((nu / 3 + 255) / 256)), dim3(256), 0, st, wp, (float4*)ws, p->Ci, p->CoP, flip);
    else
      hipLaunchKernelGGL(k_wino_weights, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, wp, (float4*)ws, p->Ci, p->CoP, flip);
    CFUN_LAUNCH_CHECK();
  }
  cfun_mfma::ConvMode md = {0, 0, 0, 0, 0, nullptr, 0, 0.f, nullptr, 0};
  if (fz) { md.in_stats = fz->in_stats; md.in_act = fz->in_act; md.in_slope = fz->in_slope; md.out_part = fz->out_part; }
  md.out_slots = w.ntz * w.nty * w.ntx * (p->d2s ? 8 : 1);
  double* finish_part = md.out_part;
  if (w.ksplit > 1) md.out_part = nullptr;     // statistics by the split-K finish instead
  int rc;
  switch (w.nsub) {
    case 1: rc = launch<1>(x, u, scale, shift, res, y, *p, w, partial, s2d_cq, md, st); break;
    case 2: rc = launch<2>(x, u, scale, shift, res, y, *p, w, partial, s2d_cq, md, st); break;
    case 3: rc = launch<3>(x, u, scale, shift, res, y, *p, w, partial, s2d_cq, md, st); break;
    case 4: rc = launch<4>(x, u, scale, shift, res, y, *p, w, partial, s2d_cq, md, st); break;
    default: rc = launch<5>(x, u, scale, shift, res, y, *p, w, partial, s2d_cq, md, st); break;
  }
  if (rc) return rc;
  if (w.ksplit > 1) return cfun_splitk_finish(partial, w.ksplit, scale, shift, res, y, p, finish_part, st);
  return CFUN_OK;
}
