// LDS-tiled, im2col-free implicit-GEMM 3-D convolution on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, same rate as the fp32 vector peak but one VGPR per operand).
//
//   forward / data-gradient kernel (k_conv_mfma):
//     block  = 256 threads (4 waves) -> 4(z) x 4(y) x 16(x) output voxels x (16*NSUB) output channels
//     wave w = z-plane w of the tile; 4 m-subtiles (rows y) of 16 voxels along x
//     K loop = input channels in chunks of 4 (one MFMA k-step) x all taps, operands from LDS:
//       X tile   [4 ch][IZ*IY*IX voxels incl. halo]   (channel planes padded so that a wave's 4x16
//                                                       fragment read is bank-conflict free)
//       W chunk  [tap][4 ch][16*NSUB (+16) co]
//     A = W^T (rows = co), B = X (cols = voxels)  =>  D[co][voxel]: each lane owns 4 consecutive output
//     channels of one voxel -> one float4 NDHWC store, fused epilogue (scale, shift, residual, activation).
//     The next chunk's global loads are issued before the MFMA block of the current one (register prefetch).
//
//   weight-gradient kernel (k_wgrad_mfma):
//     block  = 2(z) x 4(y) x 16(x) voxels per step, one 16-channel ci subtile x (16*NSUB) co, a range of
//              spatial tiles; taps are split over the 4 waves (or the voxel groups, when taps < 4)
//     A = X^T (rows = ci, k = 4 consecutive voxels), B = G (k = voxels, cols = co)  =>  D[ci][co] per tap,
//     accumulated in registers over all tiles of the block, written as a partial; a second kernel sums the
//     partials (deterministic, no atomics).
#pragma once
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// conv3d.hip: sums split-K partials in order and applies the fused epilogue; with stat_part also the per-(sample, block)
// sums of y and y*y per channel (cfun_splitk_stat_slots(p) slots per sample, k_channel_finalize's layout)
int cfun_splitk_finish(const float* partial, int ksplit, const float* scale, const float* shift, const float* res,
                       float* y, const CfunConv3dParams* p, double* stat_part, hipStream_t st);
int cfun_splitk_stat_slots(const CfunConv3dParams* p);

namespace cfun_mfma {

constexpr int pad_plane(int v, int rs) { return rs == 1 ? v + ((16 - (v % 32)) + 32) % 32 : (v | 1); }
constexpr int pad_row16(int v) { return (v % 32 == 16) ? v : v + 16; }  // v is a multiple of 16
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
// LDS row stride (floats) of a weight / gradient row of nt channels: = 16 (mod 32) so that the four k-rows of an
// MFMA fragment read fall on disjoint banks
constexpr int row_stride(int nt) { return nt <= 16 ? 16 : nt <= 48 ? 48 : nt <= 80 ? 80 : pad_row16((nt + 15) / 16 * 16); }

// bijective XCD-aware remap: consecutive logical ids stay on one XCD (blocks are dealt round-robin to the 8 XCDs)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, local = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

// Tile raster: consecutive logical ids sweep super-tiles of 4(z) x 4(y) x all(x) output tiles, so that the ~100
// workgroups an XCD runs at a time share their halos through its 4 MB L2 in z as well as in x / y (a plain x,y,z
// raster revisits a z-neighbour one whole tile-slab later, after the L2 has turned over).  Bijective for any counts.
__device__ __forceinline__ void tile_raster(unsigned t, int ntz, int nty, int ntx, int& tz, int& ty, int& tx) {
  constexpr int SZ = 4, SY = 4;
  const unsigned band = (unsigned)(SZ * nty * ntx);
  const int zb = (int)(t / band);
  unsigned r = t - (unsigned)zb * band;
  const int bz = ntz - zb * SZ < SZ ? ntz - zb * SZ : SZ;
  const unsigned sup = (unsigned)(bz * SY * ntx);
  const int yb = (int)(r / sup);
  r -= (unsigned)yb * sup;
  const int by = nty - yb * SY < SY ? nty - yb * SY : SY;
  const int tzi = (int)(r / (unsigned)(by * ntx));
  r -= (unsigned)(tzi * by * ntx);
  tz = zb * SZ + tzi;
  ty = yb * SY + (int)(r / (unsigned)ntx);
  tx = (int)(r % (unsigned)ntx);
}

// Internal launch modes derived by the C entry points (conv3d.hip) from CfunConv3dParams.
struct ConvMode {
  int flip;      // mirrored taps: the data gradient of a stride-1 conv
  int in_s2d;    // the logical input [N,Di,Hi,Wi,8*in_cq] is gathered from a hi-res tensor [N,2Di,2Hi,2Wi,in_cq]
                 // (the output gradient of a depth-to-space conv); weight rows of parity q start at q*in_cqp
  int in_cq, in_cqp;
  int tap_skip;  // 0 none | 1 by the output-channel subtile's parity (every 16-column subtile lies inside one parity group)
                 // | 2 by the input chunk's parity (flipped taps)
  // ---- InstanceNorm / LeakyReLU folded into the conv (CfunConvFusion, include/cfun_hip.h)
  const float* in_stats;  // [N][Ci][2] {mean, rstd} or null: the staged input is in_act((x - mean) * rstd)
  int in_act;             // CFUN_ACT_* applied to the (normalised) input at commit time; zero padding stays zero
  float in_slope;
  double* out_part;       // null, or per-(sample, slot) sums of y and y*y per channel, slot-minor: [N][2][Cy][slots] (the
  int out_slots;          // finalize reads a channel's slots contiguously); slots = tiles per sample (x 8 parities for d2s)
};

// The input prologue (CfunConvFusion.in_stats / in_act): what the conv reads in place of a staged value v of a channel
// with statistics (mean, rstd) -- the arithmetic of k_instnorm_lrelu_fwd / k_lrelu_fwd, so that the fused and the
// materialised paths agree bit for bit.
// (act is NONE or LRELU with 0 <= slope <= 1: max(xh, xh * slope) == (xh > 0 ? xh : xh * slope), two instructions
// instead of the compare / select chains of the generic cfun_apply_act; slope = 1 encodes "no activation")
__device__ __forceinline__ float norm_act_in(float v, float mean, float rstd, int, float slope) {
  const float xh = (v - mean) * rstd;
  return fmaxf(xh, xh * slope);
}
__device__ __forceinline__ float4 norm_act_in4(const float4& v, const float4& s01, const float4& s23, int act, float slope) {
  // s01 = (mean0, rstd0, mean1, rstd1), s23 = (mean2, rstd2, mean3, rstd3): a row of stats[n][c..c+3][2]
  return make_float4(norm_act_in(v.x, s01.x, s01.y, act, slope), norm_act_in(v.y, s01.z, s01.w, act, slope),
                     norm_act_in(v.z, s23.x, s23.y, act, slope), norm_act_in(v.w, s23.z, s23.w, act, slope));
}

// Per-tile, per-channel sums of the tile's final outputs (InstanceNorm statistics from the producer's epilogue), as
// accurate as the stand-alone fp64 pass: a lane sums ITS (at most 4) voxels per channel in fp32, parks the pair (sum,
// sum of squares) in LDS -- entry e = wave*16 + (lane & 15), `es` floats per entry -- and one thread per (channel, quantity)
// adds the 64 entries in fp64 in a fixed order and writes one slot of the fp64 partial buffer.  `red` reuses the main
// loop's tiles (the caller has passed a __syncthreads() after the loop); channels are taken in rounds of at most
// STAT_ROUND so that 64 * (2 * STAT_ROUND + 8) floats fit every kernel's LDS.
constexpr int STAT_ROUND = 32;
constexpr int stat_es(int chans) { return 2 * chans + 8; }            // (+8: the 16 x-lanes of a quad land on different banks)
constexpr int stat_lds_floats(int nt) { return 64 * stat_es(nt < STAT_ROUND ? nt : STAT_ROUND); }
// the 16 lanes with equal lane>>4 share the quad (16-wide MFMA subtiles): local channel cl + (lane>>4)*4 + j
__device__ __forceinline__ void quad_park_16(const float (&sa)[4], const float (&sb)[4], float* red, int es, int wv, int lane, int cl) {
  float* r = red + (wv * 16 + (lane & 15)) * es + (cl + (lane >> 4) * 4) * 2;
  *reinterpret_cast<float4*>(r) = make_float4(sa[0], sb[0], sa[1], sb[1]);
  *reinterpret_cast<float4*>(r + 4) = make_float4(sa[2], sb[2], sa[3], sb[3]);
}
// all 64 lanes share the quad (remainder quads: lane = voxel (row lane>>4, x = lane&15)): the 4 rows are added first
__device__ __forceinline__ void quad_park_wave(const float (&sa)[4], const float (&sb)[4], float* red, int es, int wv, int lane, int cl) {
  float u[4], v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u[j] = sa[j]; v[j] = sb[j];
    u[j] += __shfl_xor(u[j], 16, 64); v[j] += __shfl_xor(v[j], 16, 64);
    u[j] += __shfl_xor(u[j], 32, 64); v[j] += __shfl_xor(v[j], 32, 64);
  }
  if ((lane >> 4) == 0) {
    float* r = red + (wv * 16 + (lane & 15)) * es + cl * 2;
    *reinterpret_cast<float4*>(r) = make_float4(u[0], v[0], u[1], v[1]);
    *reinterpret_cast<float4*>(r + 4) = make_float4(u[2], v[2], u[3], v[3]);
  }
}
// one round: tile-local channels [c0, c0 + nch) are parked; sum them and write the slot.  All 256 threads call it.
__device__ __forceinline__ void stat_round_flush(const float* red, int es, int tid, int c0, int nch, int cobase, int n, int tile,
                                                 const CfunConv3dParams& p, const ConvMode& md) {
  __syncthreads();
  if (tid < 2 * nch) {
    const int cl = tid >> 1, qn = tid & 1, co = cobase + c0 + cl;
    if (co < p.Co) {
      double sum = 0.0;
#pragma unroll 8
      for (int e = 0; e < 64; ++e) sum += (double)red[e * es + cl * 2 + qn];
      int slot = tile, ch = co, cy = p.Co;
      if (p.d2s) {
        const int CqP = p.Co >> 3;
        cy = p.d2s_cq > 0 ? p.d2s_cq : CqP;
        const int q = co / CqP;
        ch = co - q * CqP;
        slot = tile * 8 + q;
      }
      if (ch < cy) md.out_part[(((int64_t)n * 2 + qn) * cy + ch) * md.out_slots + slot] = sum;
    }
  }
  __syncthreads();
}

// taps of a parity-folded "nearest x2 -> 3x3x3" kernel that are non-zero for output parity q = (pz,py,px):
// per axis the 2 low-resolution taps {p, p+1} of {0,1,2}
__device__ __forceinline__ unsigned parity_tapmask(int q, bool flip) {
  unsigned m = 0;
  const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
  for (int dz = pz; dz <= pz + 1; ++dz)
    for (int dy = py; dy <= py + 1; ++dy)
      for (int dx = px; dx <= px + 1; ++dx) {
        const int tap = (dz * 3 + dy) * 3 + dx;
        m |= 1u << (flip ? 26 - tap : tap);
      }
  return m;
}

template <int KD, int KH, int KW, int S>
struct FwdTile {
  static constexpr int TD = 4, TH = 4, TW = 16;
  static constexpr int TAPS = KD * KH * KW;
  static constexpr bool COMPACT = TAPS == 1;     // 1x1x1: stage exactly the voxels that are read
  static constexpr int RS = COMPACT ? 1 : S;     // x stride of a fragment read in LDS
  static constexpr int IZ = COMPACT ? TD : (TD - 1) * S + KD;
  static constexpr int IY = COMPACT ? TH : (TH - 1) * S + KH;
  static constexpr int IX = COMPACT ? TW : (TW - 1) * S + KW;
  static constexpr int IVOX = IZ * IY * IX;
  static constexpr int PLANEP = pad_plane(IVOX, RS);
  static constexpr int IN_LOADS = cdiv(IVOX, 256);
};

// SPECIAL = false: plain conv (md.in_s2d == 0, md.tap_skip == 0) -- the hot instantiation carries none of the
// parity-fold bookkeeping.  SPECIAL = true (3x3x3 stride 1 only): s2d gather of the input and/or tap skipping.
// Output-channel tile = 16*NSUB + 4*REM channels.  The REM (0..2) trailing 4-channel groups run on
// v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products: lane 4b+j supplies voxel j of block b, lane 4b+i the
// weight of channel i; result lane = voxel, 4 registers = 4 channels -- measured with tools/probe_mfma.py), so
// Co = 20 / 40 / 8 tiles carry no channel padding (a 16-wide MFMA subtile would be 75 % / 50 % / 50 % idle).
// STATS: the epilogue also produces the per-tile sums of y and y*y per channel (md.out_part) -- an instantiation of its
// own, so that the plain kernels keep their register budget (the sums cost the 64-channel tile a resident wave)
// MODE 0: plain; 1: s2d gather of the input without tap skipping (data gradient of the folded 5^3 conv); 2: the parity-folded
// up-conv's forward (md.tap_skip == 1): live taps slot-major, (subtile, slot) MFMA loop; 3: its data gradient (s2d gather,
// md.tap_skip == 2: the 8 live taps of the input chunk's parity, mirrored), slot-major as well.
template <int KD, int KH, int KW, int S, int NSUB, int MODE, int REM, bool STATS = false>
__global__ void __launch_bounds__(256)
k_conv_mfma(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
            const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
            CfunConv3dParams p, ConvMode md, int ntz, int nty, int ntx, int ncot, float* __restrict__ partial,
            int chunks_per_split) {
  constexpr bool SPECIAL = MODE != 0, UPF = MODE == 2, UPD = MODE == 3;
  static_assert(!UPD || (KD == 3 && KH == 3 && KW == 3 && S == 1 && NSUB > 0 && REM == 0), "up-conv data-gradient tiles");
  static_assert(!UPF || (KD == 3 && KH == 3 && KW == 3 && S == 1 && NSUB > 0 && REM == 0), "up-conv forward tiles");
  using T = FwdTile<KD, KH, KW, S>;
  constexpr int TAPS = T::TAPS, NT = 16 * NSUB + 4 * REM, NTP = row_stride(NT), NS1 = NSUB > 0 ? NSUB : 1;
  constexpr int W_ITEMS = TAPS * NT;  // float4 items per weight chunk: TAPS*4 rows x NT/4
  constexpr int W_LOADS = cdiv(W_ITEMS, 256);
  CFUN_DYN_LDS(float, smem);
  float* Xl = smem;                      // [4][PLANEP]
  float* Wl = smem + 4 * T::PLANEP;      // [TAPS*4][NTP]

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot; lid /= ncot;
  const unsigned per_n = (unsigned)(ntz * nty * ntx);
  const int n = lid / per_n;
  int tz, ty, tx;
  tile_raster(lid - (unsigned)n * per_n, ntz, nty, ntx, tz, ty, tx);
  const int z0 = tz * T::TD, y0 = ty * T::TH, x0 = tx * T::TW;
  const int cobase = cot * NT;
  const int sh = p.up2 ? 1 : 0;
  const int Dv = p.Di << sh, Hv = p.Hi << sh, Wv = p.Wi << sh;

  // ---- per-thread staging descriptors (independent of the channel chunk)
  int64_t in_off[T::IN_LOADS];  // element offset of the voxel's channel 0, or -1 when padded / unused
#pragma unroll
  for (int i = 0; i < T::IN_LOADS; ++i) {
    const int idx = tid + i * 256;
    in_off[i] = -1;
    if (idx < T::IVOX) {
      const int ix = idx % T::IX, iy = (idx / T::IX) % T::IY, iz = idx / (T::IX * T::IY);
      int vz, vy, vx;
      if (T::COMPACT) { vz = (z0 + iz) * S - p.pd; vy = (y0 + iy) * S - p.ph; vx = (x0 + ix) * S - p.pw; }
      else { vz = z0 * S - p.pd + iz; vy = y0 * S - p.ph + iy; vx = x0 * S - p.pw + ix; }
      if (vz >= 0 && vz < Dv && vy >= 0 && vy < Hv && vx >= 0 && vx < Wv) {
        if (SPECIAL && md.in_s2d)   // parity-0 voxel of the hi-res tensor; the chunk's parity offset is added in prefetch()
          in_off[i] = ((((int64_t)n * 2 * p.Di + 2 * vz) * 2 * p.Hi + 2 * vy) * 2 * p.Wi + 2 * vx) * md.in_cq;
        else
          in_off[i] = ((((int64_t)n * p.Di + (vz >> sh)) * p.Hi + (vy >> sh)) * p.Wi + (vx >> sh)) * p.Ci;
      }
    }
  }
  // chunk c -> (offset into x added to in_off, first weight row, parity of the chunk)
  const int cpq = (SPECIAL && md.in_s2d) ? (md.in_cq >> 2) : 1;   // chunks per parity
  auto chunk_xoff = [&](int c) -> int64_t {
    if (!SPECIAL || !md.in_s2d) return (int64_t)c * 4;
    const int q = c / cpq, o4 = c - q * cpq;
    return ((int64_t)((q >> 2) * 2 * p.Hi + ((q >> 1) & 1)) * 2 * p.Wi + (q & 1)) * md.in_cq + o4 * 4;
  };
  auto chunk_wrow = [&](int c) -> int {
    if (!SPECIAL || !md.in_s2d) return c * 4;
    const int q = c / cpq, o4 = c - q * cpq;
    return q * md.in_cqp + o4 * 4;
  };
  float4 xin[T::IN_LOADS], win[W_LOADS];
  // tap skipping: only the 8 live taps {p, p+1}^3 of the column's (tap_skip 1) / the chunk's (tap_skip 2) parity are staged
  // -- slot j of the weight stage holds live tap j (8 * NT float4 items per chunk instead of 27 * NT).  tap_skip 1 keeps
  // them slot-major in LDS ([8 slots][4 ch][NTP]: the MFMA loop walks (subtile, slot)), tap_skip 2 at their tap's row.
  const int w_items = (UPF || (SPECIAL && TAPS == 27 && md.tap_skip)) ? 8 * NT : W_ITEMS;
  const int CqPw = p.Co >> 3;      // d2s: padded channels per parity (column -> parity for tap_skip 1)
  int q_staged = 0;
  auto live_tap_of = [](int q, int j) {      // unflipped weight tap of live slot j = (a,b,c) for parity q = (pz,py,px)
    return (((q >> 2) + (j >> 2)) * 3 + (((q >> 1) & 1) + ((j >> 1) & 1))) * 3 + ((q & 1) + (j & 1));
  };
  // input prologue: the chunk's 4 channels carry (mean, rstd) of sample n -- wave-uniform, fetched with the chunk
  const bool in_fused = md.in_stats != nullptr || md.in_act != CFUN_ACT_NONE;
  float4 ns01 = make_float4(0.f, 1.f, 0.f, 1.f), ns23 = make_float4(0.f, 1.f, 0.f, 1.f);
  auto prefetch = [&](int c) {
    const int64_t xo = chunk_xoff(c);
    const int wrow = chunk_wrow(c);
    if (md.in_stats) {
      const float4* sp = reinterpret_cast<const float4*>(md.in_stats + ((int64_t)n * p.Ci + c * 4) * 2);
      ns01 = sp[0]; ns23 = sp[1];
    }
#pragma unroll
    for (int i = 0; i < T::IN_LOADS; ++i)
      xin[i] = in_off[i] >= 0 ? *reinterpret_cast<const float4*>(x + in_off[i] + xo) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (SPECIAL && TAPS == 27 && md.tap_skip == 2) q_staged = c / cpq;
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const int it = tid + i * 256;
      win[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < w_items) {
        const int row = it / (NT / 4), col = (it % (NT / 4)) * 4;
        int tapw = row >> 2;
        const int cc = row & 3;
        if (UPF) tapw = live_tap_of((cobase + col) / CqPw, tapw);   // the column's parity
        else if (SPECIAL && TAPS == 27 && md.tap_skip) tapw = live_tap_of(q_staged, tapw);      // weight tap of live slot j
        else if (md.flip) tapw = TAPS - 1 - tapw;
        if (cobase + col < p.CoP)
          win[i] = *reinterpret_cast<const float4*>(wp + ((int64_t)tapw * p.Ci + wrow + cc) * p.CoP + cobase + col);
      }
    }
  };
  auto commit = [&]() {
    if (in_fused) {       // (padded voxels were fetched as zeros and must stay zeros: the conv pads the NORMALISED tensor)
#pragma unroll
      for (int i = 0; i < T::IN_LOADS; ++i)
        if (in_off[i] >= 0) xin[i] = norm_act_in4(xin[i], ns01, ns23, md.in_act, md.in_slope);
    }
#pragma unroll
    for (int i = 0; i < T::IN_LOADS; ++i) {
      const int idx = tid + i * 256;
      if (idx < T::IVOX) {
        Xl[idx] = xin[i].x; Xl[T::PLANEP + idx] = xin[i].y;
        Xl[2 * T::PLANEP + idx] = xin[i].z; Xl[3 * T::PLANEP + idx] = xin[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < w_items) {
        int row = it / (NT / 4);
        const int col = (it % (NT / 4)) * 4;
        if (!UPD && SPECIAL && TAPS == 27 && md.tap_skip == 2) {      // live slot -> the row of its tap in the MFMA loop's mirrored order
          const int t = live_tap_of(q_staged, row >> 2);
          row = (md.flip ? TAPS - 1 - t : t) * 4 + (row & 3);
        }
        *reinterpret_cast<float4*>(Wl + row * NTP + col) = win[i];
      }
    }
  };

  f32x4 acc[4][NS1], accr[REM > 0 ? REM : 1];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < NS1; ++nn) acc[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < (REM > 0 ? REM : 1); ++q) accr[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* Xw = Xl + (lane >> 4) * T::PLANEP + (wv * T::RS * T::IY) * T::IX + (lane & 15) * T::RS;
  const float* Ww = Wl + (lane >> 4) * NTP + (lane & 15);
  // remainder quads: lane = voxel (row lane>>4, x = lane&15) of the wave's 4x16 plane; weights of channel lane&3
  const float* Xr = Xl + ((wv * T::RS) * T::IY + (lane >> 4) * T::RS) * T::IX + (lane & 15) * T::RS;
  const float* Wr = Wl + 16 * NSUB + (lane & 3);

  // p.Ci is the number of weight rows per tap; with in_s2d only the valid channels of each parity are visited
  const int nchunks = (SPECIAL && md.in_s2d) ? 8 * cpq : (p.Ci >> 2);
  const int CqP = p.Co >> 3;   // d2s: padded channels per parity
  unsigned tapmask = 0xffffffffu;
  // split-K (small volumes: too few tiles to fill 256 CUs): blockIdx.y owns a range of channel chunks and
  // stores raw accumulators to partial[blockIdx.y]; cfun_splitk_finish sums them in order and runs the epilogue
  const int c_begin = blockIdx.y * chunks_per_split;
  const int c_end = (c_begin + chunks_per_split < nchunks) ? c_begin + chunks_per_split : nchunks;
  if (c_begin < c_end) prefetch(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();           // every wave is done reading the previous chunk
    commit();
    __syncthreads();
    if (c + 1 < c_end) prefetch(c + 1);
    if (SPECIAL && TAPS == 27 && md.tap_skip == 2) tapmask = parity_tapmask(c / cpq, true);
    if constexpr (UPF) {
      // parity-folded up-conv: subtile nn belongs to parity q = (pz,py,px) and reads the 2x2x2 taps {q, q+1} -- a loop over
      // (subtile, live slot) with wave-uniform LDS offsets instead of 27 unrolled taps each behind a mask test (round 4:
      // -7 ... -17 % on the four up-conv forwards, profiles/round4_upconv_forward.log)
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) {
        const int q = (cobase + nn * 16) / CqP;      // wave-uniform
        const float* xq = Xw + (((q >> 2) * T::IY + ((q >> 1) & 1)) * T::IX + (q & 1));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = Ww[j * 4 * NTP + nn * 16];
          const float* xt = xq + (((j >> 2) * T::IY + ((j >> 1) & 1)) * T::IX + (j & 1));
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xt[m * T::RS * T::IX], acc[m][nn], 0, 0, 0);
        }
#ifndef CFUN_HIP_EMULATION
        __builtin_amdgcn_sched_barrier(0);      // one subtile's 40 LDS reads at a time: hoisted across subtiles they cost 200+ registers
#endif
      }
    } else if constexpr (UPD) {
      // data gradient of the parity-folded up-conv: the chunk's 4 channels belong to output parity q of the forward conv and
      // meet only its 8 live taps, mirrored (tap 26 - t): the same (slot) loop with wave-uniform LDS offsets
      const int q = c / cpq;
      const float* xq = Xw + (((2 - (q >> 2)) * T::IY + (2 - ((q >> 1) & 1))) * T::IX + (2 - (q & 1)));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a[NS1];
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn) a[nn] = Ww[j * 4 * NTP + nn * 16];
        const float* xt = xq - (((j >> 2) * T::IY + ((j >> 1) & 1)) * T::IX + (j & 1));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float b = xt[m * T::RS * T::IX];
#pragma unroll
          for (int nn = 0; nn < NSUB; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nn], b, acc[m][nn], 0, 0, 0);
        }
#ifndef CFUN_HIP_EMULATION
        if (j & 1) __builtin_amdgcn_sched_barrier(0);      // two slots' reads in flight: hoisting all 8 costs a resident wave
#endif
      }
    } else {
#pragma unroll
    for (int dz = 0; dz < KD; ++dz)
#pragma unroll
      for (int dy = 0; dy < KH; ++dy)
#pragma unroll
        for (int dx = 0; dx < KW; ++dx) {
          const int tap = (dz * KH + dy) * KW + dx;
          if (SPECIAL && TAPS == 27 && !((tapmask >> (tap & 31)) & 1u)) continue;   // wave-uniform: folded-zero taps
          if constexpr (NSUB > 0) {
            float a[NS1];
#pragma unroll
            for (int nn = 0; nn < NSUB; ++nn) a[nn] = Ww[tap * 4 * NTP + nn * 16];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const float b = Xw[(dz * T::IY + (m * T::RS + dy)) * T::IX + dx];
#pragma unroll
              for (int nn = 0; nn < NSUB; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nn], b, acc[m][nn], 0, 0, 0);
            }
          }
          if constexpr (REM > 0) {
            float xb[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) xb[cc] = Xr[cc * T::PLANEP + (dz * T::IY + dy) * T::IX + dx];
#pragma unroll
            for (int q = 0; q < REM; ++q)
#pragma unroll
              for (int cc = 0; cc < 4; ++cc)
                accr[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(Wr[(tap * 4 + cc) * NTP + 4 * q], xb[cc], accr[q], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue.  16-wide subtiles: lane owns voxel (z0+wv, y0+m, x0+(lane&15)), channels nn*16 + (lane>>4)*4..+3;
  // remainder quads: lane owns voxel (z0+wv, y0+(lane>>4), x0+(lane&15)), channels 16*NSUB + 4q..+3
  const int oz = z0 + wv, ox = x0 + (lane & 15);
  constexpr bool stats_on = STATS;
  const bool vox_ok = oz < p.Do && ox < p.Wo;
  if (!stats_on && !vox_ok) return;
  if (stats_on) __syncthreads();      // every wave has left the main loop: its LDS tiles are dead, `smem` becomes `red`
  auto emit = [&](int oy, int co, const f32x4& a4, float (&sa)[4], float (&sb)[4]) {
    if (!vox_ok || oy >= p.Ho || co >= p.Co) return;
    const int64_t v = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    float4 r = make_float4(a4[0], a4[1], a4[2], a4[3]);
    if (gridDim.y > 1) {       // split-K partial: raw sums, plain layout
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
    const int Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;          // valid channels per parity (= channels of y)
    const int q = p.d2s ? co / CqP : 0, oc = p.d2s ? co - q * CqP : co;
    if (p.d2s && oc >= Cq) return;                          // per-parity channel padding
    if (p.res_mode) {
      int64_t rv = v;
      if (p.res_up2 && !p.d2s)
        rv = (((int64_t)n * (p.Do >> 1) + (oz >> 1)) * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
      const float4 t = *reinterpret_cast<const float4*>(res + rv * (p.d2s ? Cq : p.Co) + oc);
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
  // statistics rounds: the tile's channels (16-wide subtiles, then the remainder quads) in windows of STAT_ROUND
  constexpr int ES = stat_es(NT < STAT_ROUND ? NT : STAT_ROUND);
  const int tile = (int)(lid - (unsigned)n * per_n);
  if constexpr (NSUB > 0) {
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) {
      float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) emit(y0 + m, cobase + nn * 16 + (lane >> 4) * 4, acc[m][nn], sa, sb);
      if constexpr (stats_on) {
        const int base = (nn * 16 / STAT_ROUND) * STAT_ROUND;
        quad_park_16(sa, sb, smem, ES, wv, lane, nn * 16 - base);
        const int end = (nn + 1) * 16;                    // window complete, or the tile's last channel parked
        if (end % STAT_ROUND == 0 || end == NT) stat_round_flush(smem, ES, tid, base, end - base, cobase, n, tile, p, md);
      }
    }
  }
  if constexpr (REM > 0) {
    constexpr int RBASE = (16 * NSUB / STAT_ROUND) * STAT_ROUND;      // the window the remainder quads fall into
    static_assert(16 * NSUB + 4 * REM - RBASE <= STAT_ROUND, "remainder quads must fit the last window");
#pragma unroll
    for (int q = 0; q < REM; ++q) {
      float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
      emit(y0 + (lane >> 4), cobase + 16 * NSUB + 4 * q, accr[q], sa, sb);
      if constexpr (stats_on) quad_park_wave(sa, sb, smem, ES, wv, lane, 16 * NSUB + 4 * q - RBASE);
    }
    if constexpr (stats_on) stat_round_flush(smem, ES, tid, RBASE, NT - RBASE, cobase, n, tile, p, md);
  }
}

// how many ways to split the channel chunks so that a small volume still fills the chip (0 workspace => 1)
inline int splitk_factor(int64_t nblk, int nchunks, const CfunConv3dParams& p, size_t ws_bytes) {
  // fewer than ~3 workgroups per CU leaves the SIMDs with a single wave each: split until ~1024 workgroups
  if (p.d2s || nblk >= 768 || nchunks < 8) return 1;
  int k = nblk <= 128 ? (int)((512 + nblk - 1) / nblk) : (int)((1024 + nblk - 1) / nblk);
  if (k > nchunks / 4) k = nchunks / 4;
  if (k > 16) k = 16;
  const size_t per = (size_t)p.N * p.Do * p.Ho * p.Wo * p.Co * sizeof(float);
  while (k > 1 && (size_t)k * per > ws_bytes) --k;
  return k < 1 ? 1 : k;
}
inline size_t splitk_workspace(int64_t nblk, int nchunks, const CfunConv3dParams& p) {
  const int k = splitk_factor(nblk, nchunks, p, (size_t)-1);
  return k > 1 ? (size_t)k * p.N * p.Do * p.Ho * p.Wo * p.Co * sizeof(float) : 0;
}

template <int KD, int KH, int KW, int S, int NSUB, int REM = 0>
int launch_conv_mfma(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                     float* y, const CfunConv3dParams& p, const ConvMode& md, void* ws, size_t ws_bytes, hipStream_t st) {
  using T = FwdTile<KD, KH, KW, S>;
  constexpr int NT = 16 * NSUB + 4 * REM, NTP = row_stride(NT);
  if (REM > 0 && (p.Co % NT) != 0) return CFUN_EINVAL;
  const int ntz = cdiv(p.Do, T::TD), nty = cdiv(p.Ho, T::TH), ntx = cdiv(p.Wo, T::TW), ncot = cdiv(p.Co, NT);
  const int64_t nblk = (int64_t)p.N * ntz * nty * ntx * ncot;
  if (nblk == 0) return CFUN_OK;
  if (nblk > 0x7fffffffLL) return CFUN_EINVAL;
  constexpr bool kHasSpecial = (KD == 3 && KH == 3 && KW == 3 && S == 1);
  const bool special = md.in_s2d || md.tap_skip;
  if (special && !kHasSpecial) return CFUN_EINVAL;
  // (tap_skip 1 keeps only the 8 live taps of each column in LDS)
  const bool upd = kHasSpecial && NSUB > 0 && REM == 0 && md.in_s2d && md.tap_skip == 2;      // MODE 3
  const size_t lds = (size_t)(4 * T::PLANEP + ((md.tap_skip == 1 || upd) ? 8 : T::TAPS) * 4 * NTP) * sizeof(float);
  const int nchunks = md.in_s2d ? 8 * (md.in_cq >> 2) : (p.Ci >> 2);
  const int ksplit = splitk_factor(nblk, nchunks, p, ws_bytes);
  const bool stats = md.out_part != nullptr && ksplit == 1;      // (split-K: the finish pass takes the statistics)
  auto kern = stats ? k_conv_mfma<KD, KH, KW, S, NSUB, 0, REM, true> : k_conv_mfma<KD, KH, KW, S, NSUB, 0, REM, false>;
  if constexpr (kHasSpecial) {
    if (special && md.tap_skip != 1)
      kern = stats ? k_conv_mfma<KD, KH, KW, S, NSUB, 1, REM, true> : k_conv_mfma<KD, KH, KW, S, NSUB, 1, REM, false>;
    if constexpr (NSUB > 0 && REM == 0) {
      if (md.tap_skip == 1) kern = stats ? k_conv_mfma<KD, KH, KW, S, NSUB, 2, 0, true> : k_conv_mfma<KD, KH, KW, S, NSUB, 2, 0, false>;
      if (upd) kern = k_conv_mfma<KD, KH, KW, S, NSUB, 3, 0, false>;      // (a data gradient: no statistics epilogue)
    } else {
      if (md.tap_skip == 1) return CFUN_EINVAL;
    }
  }
  size_t lds_k = lds;
  if (stats && lds_k < stat_lds_floats(NT) * sizeof(float)) lds_k = stat_lds_floats(NT) * sizeof(float);
  if (lds_k > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
    if (e != hipSuccess) return (int)e;
  }
  const int cps = cdiv(nchunks, ksplit);
  ConvMode mk = md;                // epilogue statistics: by this kernel's tiles, or by the split-K finish
  mk.out_slots = ntz * nty * ntx * (p.d2s ? 8 : 1);
  if (ksplit > 1) mk.out_part = nullptr;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)ksplit), dim3(256), lds_k, st, x, wp, scale, shift, res, y, p, mk,
                     ntz, nty, ntx, ncot, (float*)ws, cps);
  CFUN_LAUNCH_CHECK();
  if (ksplit > 1) return cfun_splitk_finish((const float*)ws, ksplit, scale, shift, res, y, &p, md.out_part, st);
  return CFUN_OK;
}

// tile code -> output channels per workgroup: NSUB + 8*REM (16*NSUB + 4*REM channels)
inline int tile_channels(int code) { return 16 * (code & 7) + 4 * (code >> 3); }

// statistics slots per sample that launch_conv_mfma fills for (p, tile code nsub) given ws_bytes of split-K workspace:
// > 0 by the conv's tiles (slot-minor layout), < 0 by the split-K finish (-(blocks), k_channel_finalize's layout)
inline int fwd_stat_slots(int nsub, const CfunConv3dParams& p, const ConvMode& md, size_t ws_bytes) {
  const int nt = tile_channels(nsub);
  const int tiles = cdiv(p.Do, 4) * cdiv(p.Ho, 4) * cdiv(p.Wo, 16);
  const int64_t nblk = (int64_t)p.N * tiles * cdiv(p.Co, nt);
  const int nchunks = md.in_s2d ? 8 * (md.in_cq >> 2) : (p.Ci >> 2);
  return splitk_factor(nblk, nchunks, p, ws_bytes) > 1 ? -cfun_splitk_stat_slots(&p) : tiles * (p.d2s ? 8 : 1);
}

template <int KD, int KH, int KW, int S>
size_t fwd_workspace(int nsub, const CfunConv3dParams& p, const ConvMode& md) {
  using T = FwdTile<KD, KH, KW, S>;
  const int nt = tile_channels(nsub);
  const int64_t nblk = (int64_t)p.N * cdiv(p.Do, T::TD) * cdiv(p.Ho, T::TH) * cdiv(p.Wo, T::TW) * cdiv(p.Co, nt);
  const int nchunks = md.in_s2d ? 8 * (md.in_cq >> 2) : (p.Ci >> 2);
  return splitk_workspace(nblk, nchunks, p);
}

// shapes that instantiate the remainder-quad tiles (the level-1 U-Net convs and the 8-channel heads)
template <int KD, int KH, int KW, int S>
constexpr bool has_rem_tiles() {
  return (KD == 3 && KH == 3 && KW == 3) || (KD == 1 && KH == 1 && KW == 1 && S == 1);
}

template <int KD, int KH, int KW, int S>
constexpr int max_nsub() { return KD * KH * KW > 27 ? 1 : 5; }   // 5x5x5: LDS / accumulator budget allows 16 channels

template <int KD, int KH, int KW, int S>
int dispatch_nsub(int nsub, const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                  float* y, const CfunConv3dParams& p, const ConvMode& flip, void* ws, size_t wsb, hipStream_t st) {
  // nsub >= 8 encodes a tile with remainder quads: nsub = NSUB + 8*REM  (tiles 20 = (1,1), 40 = (2,2), 8 = (0,2))
  if constexpr (has_rem_tiles<KD, KH, KW, S>()) {
    if (nsub == 1 + 8 * 1) return launch_conv_mfma<KD, KH, KW, S, 1, 1>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
    if (nsub == 2 + 8 * 2) return launch_conv_mfma<KD, KH, KW, S, 2, 2>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
    if (nsub == 0 + 8 * 2) return launch_conv_mfma<KD, KH, KW, S, 0, 2>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
  }
  if (nsub >= 8 || nsub < 1) return CFUN_EINVAL;
  if constexpr (max_nsub<KD, KH, KW, S>() == 1) {
    if (nsub != 1) return CFUN_EINVAL;
    return launch_conv_mfma<KD, KH, KW, S, 1>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
  } else {
    switch (nsub) {
      case 1: return launch_conv_mfma<KD, KH, KW, S, 1>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
      case 2: return launch_conv_mfma<KD, KH, KW, S, 2>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
      case 3: return launch_conv_mfma<KD, KH, KW, S, 3>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
      case 4: return launch_conv_mfma<KD, KH, KW, S, 4>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
      default: return launch_conv_mfma<KD, KH, KW, S, 5>(x, wp, scale, shift, res, y, p, flip, ws, wsb, st);
    }
  }
}

// ============================================================================================ weight gradient
struct WgIn {          // input prologue of the weight-gradient kernels (kernel argument; see WgPlan)
  const float* stats;
  int act;
  float slope;
};

template <int KD, int KH, int KW, int S>
struct WgTile {
  static constexpr int TD = (S == 1) ? 2 : 1, TH = 4, TW = 16;   // stride 2: smaller tile, the halo tile is 4x larger
  static constexpr int TVOX = TD * TH * TW;  // 128 (64 for stride 2)
  static constexpr int TAPS = KD * KH * KW;
  static constexpr bool COMPACT = TAPS == 1;
  static constexpr int RS = COMPACT ? 1 : S;
  static constexpr int IZ = COMPACT ? TD : (TD - 1) * S + KD;
  static constexpr int IY = COMPACT ? TH : (TH - 1) * S + KH;
  static constexpr int IX = COMPACT ? TW : (TW - 1) * S + KW;
  static constexpr int IVOX = IZ * IY * IX;
  static constexpr int XS = 16;                         // floats per staged voxel (one ci subtile)
  static constexpr int X_LOADS = cdiv(IVOX * 4, 256);   // float4 items
  static constexpr int TSPLIT = TAPS >= 4 ? 4 : 1;      // taps over waves, else voxel groups over waves
  static constexpr int KSPLIT = 4 / TSPLIT;
  static constexpr int TPW = cdiv(TAPS, TSPLIT);        // taps per wave
};

// TSKIP = true (3x3x3 stride 1, parity-folded up2 weights): each co tile lies inside one parity group whose 8 live
// taps {p,p+1}^3 are dealt 2 per wave; the 19 folded-zero taps are written as zeros.  The hot loop is branch-free.
//
// PACK > 1 (27-tap shapes whose C_in <= R = 16/PACK, i.e. <= 8): the 16 rows of the A fragment carry PACK taps x R
// channels instead of 1 tap x 16 (mostly zero) channels -- row i reads tap group*PACK + i/R, channel i%R -- so the
// block runs ceil(27/PACK) instead of 27 tap rows of MFMA work.
// IN: the input prologue (WgIn) -- x is staged as in_act((x - mean) * rstd).  The (mean, rstd) rows of the committed
// tile's sample sit in a small LDS table behind the tiles (refilled when the sample changes, by the prefetch that runs
// between two commits), so the plain instantiation keeps its registers and the prologue costs two LDS reads per item.
template <int KD, int KH, int KW, int S, int NSUB, bool TSKIP, int PACK, bool IN>
__device__ __forceinline__ void
wgrad_body(float* smem, const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial,
           const CfunConv3dParams& p, int ntz, int nty, int ntx, int cot, int cis, int chunk, int tiles_per_chunk,
           int ntiles, const WgIn& fin) {
  using T = WgTile<KD, KH, KW, S>;
  constexpr int TAPS = T::TAPS, NT = 16 * NSUB, GS = pad_row16(NT);
  constexpr int NLIVE = TSKIP ? 8 : TAPS;                         // taps this block accumulates
  constexpr int R = 16 / PACK;                                    // channels per packed tap
  // taps over the waves (TSPLIT) or voxel groups over the waves (KSPLIT partial sums per chunk).  TSKIP: all 8 live taps in
  // EVERY wave and the voxel groups dealt over the waves -- with 2 taps per wave a group's 2 + NSUB fragment reads fed only
  // 2 * NSUB MFMAs (counters, round 4: 4 VALU + 4 SALU + 0.8 LDS instructions per MFMA, the pipe 0.49 busy); 8 + NSUB reads
  // now feed 8 * NSUB
  // (tiles of <= 32 columns only: at 48 / 80 columns the 8 * NSUB accumulators cost the second resident wave and the small
  // volumes pay for 4 x the partials -- 80 -> 40 @ 24^3 0.45 -> 0.57 ms, measured; 40 -> 20 @ 48^3 1.14 -> 1.02)
  constexpr int TSPLIT = (TSKIP && NSUB <= 2) ? 1 : T::TSPLIT, KSPLIT = 4 / TSPLIT;
  constexpr int TPW = cdiv(cdiv(NLIVE, PACK), TSPLIT);            // tap groups per wave
  constexpr int G_ITEMS = T::TVOX * (NT / 4);
  constexpr int G_LOADS = cdiv(G_ITEMS, 256);
  float* Xl = smem;                       // [IVOX][16]
  float* Gl = smem + T::IVOX * T::XS;     // [TVOX][GS]
  float4* Sl = reinterpret_cast<float4*>(Gl + T::TVOX * GS);      // IN: [4 channel quads of the subtile][2] stats rows

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const int ci0 = cis * 16;
  // co tile: plain = cot*NT; TSKIP = tile t of parity group q (groups are CqP columns wide, tiles never straddle one)
  int cobase = cot * NT, colimit = NT, qpar = 0;
  if (TSKIP) {
    const int CqP = p.Co >> 3, tpp = cdiv(CqP, NT);
    qpar = cot / tpp;
    const int t = cot - qpar * tpp;
    cobase = qpar * CqP + t * NT;
    colimit = CqP - t * NT < NT ? CqP - t * NT : NT;
  }
  const int sh = p.up2 ? 1 : 0;
  const int Dv = p.Di << sh, Hv = p.Hi << sh, Wv = p.Wi << sh;
  const int tslot = wv % TSPLIT, kslot = wv / TSPLIT;

  f32x4 acc[TPW][NSUB];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) acc[t][nn] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Staging of tile t+1 is issued before the MFMA phase of tile t and must not stall it: every load is
  // UNCONDITIONAL (out-of-range items read offset 0) and the zero padding is applied from the validity bits at
  // commit time -- exec-masked loads made hipcc wrap each one in nested branches and wait (vmcnt(0)) mid-way.
  float4 xin[T::X_LOADS], gin[G_LOADS];
  unsigned xvalid = 0, gvalid = 0;
  int n_table = -1;      // IN: the sample whose statistics the LDS table holds
  auto prefetch = [&](int tile) {
    int t = tile;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz;
    const int n = t / ntz;
    const int z0 = tz * T::TD, y0 = ty * T::TH, x0 = tx * T::TW;
    xvalid = 0; gvalid = 0;
    if constexpr (IN) {
      if (n != n_table) {      // (wave-uniform; the previous commit is done with the table: we are past its barrier)
        n_table = n;
        if (tid < 8) {
          const int cs = ci0 + (tid >> 1) * 4;
          Sl[tid] = (fin.stats && cs < p.Ci) ? reinterpret_cast<const float4*>(fin.stats + ((int64_t)n * p.Ci + cs) * 2)[tid & 1]
                                             : make_float4(0.f, 1.f, 0.f, 1.f);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < T::X_LOADS; ++i) {
      const int it = tid + i * 256;
      const int idx = it >> 2, c = ci0 + (it & 3) * 4;
      const int ix = idx % T::IX, iy = (idx / T::IX) % T::IY, iz = idx / (T::IX * T::IY);
      int vz, vy, vx;
      if (T::COMPACT) { vz = (z0 + iz) * S - p.pd; vy = (y0 + iy) * S - p.ph; vx = (x0 + ix) * S - p.pw; }
      else { vz = z0 * S - p.pd + iz; vy = y0 * S - p.ph + iy; vx = x0 * S - p.pw + ix; }
      const bool ok = (it < T::IVOX * 4) & (c < p.Ci) & (vz >= 0) & (vz < Dv) & (vy >= 0) & (vy < Hv) & (vx >= 0) &
                      (vx < Wv);
      // 32-bit element offsets (the launcher rejects tensors of 2^31 elements or more): half the address registers
      const unsigned off = ((((unsigned)n * p.Di + (vz >> sh)) * p.Hi + (vy >> sh)) * p.Wi + (vx >> sh)) * p.Ci + c;
      xin[i] = *reinterpret_cast<const float4*>(x + (ok ? off : 0u));
      xvalid |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      const int vox = it / (NT / 4), col = (it % (NT / 4)) * 4;
      const int lx = vox % T::TW, ly = (vox / T::TW) % T::TH, lz = vox / (T::TW * T::TH);
      const int oz = z0 + lz, oy = y0 + ly, ox = x0 + lx;
      bool ok = (it < G_ITEMS) & (col < colimit) & (cobase + col < p.Co) & (oz < p.Do) & (oy < p.Ho) & (ox < p.Wo);
      unsigned off;
      if (p.d2s) {   // g is the hi-res gradient of y [N,2Do,2Ho,2Wo,Cq]: gather parity q, channel o
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
    // (component-wise select: `cond ? xin[i] : zero` on the struct takes its address and sends the array to scratch)
    auto keep = [](unsigned bit, const float4& v) {
      const float m = bit ? 1.f : 0.f;
      return make_float4(bit ? v.x : m, bit ? v.y : m, bit ? v.z : m, bit ? v.w : m);
    };
#pragma unroll
    for (int i = 0; i < T::X_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < T::IVOX * 4)
        *reinterpret_cast<float4*>(Xl + (it >> 2) * T::XS + (it & 3) * 4) =
            keep((xvalid >> i) & 1u, IN ? norm_act_in4(xin[i], Sl[(it & 3) * 2], Sl[(it & 3) * 2 + 1], fin.act, fin.slope) : xin[i]);
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < G_ITEMS)
        *reinterpret_cast<float4*>(Gl + (it / (NT / 4)) * GS + (it % (NT / 4)) * 4) = keep((gvalid >> i) & 1u, gin[i]);
    }
  };

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = (t_begin + tiles_per_chunk < ntiles) ? t_begin + tiles_per_chunk : ntiles;
  // A fragment: lane -> row i = lane&15 (ci, or PACK taps x R ci), voxel k = lane>>4 ; B: co = lane&15, voxel k
  const float* Xw = Xl + (lane >> 4) * T::RS * T::XS + (PACK == 1 ? (lane & 15) : 0);
  const float* Gw = Gl + (lane >> 4) * GS + (lane & 15);

  // live tap j -> (tap index, LDS offset of its halo shift); j is clamped, so every offset is valid and the hot loop
  // has no branches -- surplus slots (27 taps over 4 waves) recompute the last tap and are not stored
  auto live_tap = [&](int j, int& tap, int& off) {
    int dz, dy, dx;
    if (TSKIP) {   // 8 live taps of parity (pz,py,px): offsets {p, p+1} per axis
      dz = (qpar >> 2) + (j >> 2); dy = ((qpar >> 1) & 1) + ((j >> 1) & 1); dx = (qpar & 1) + (j & 1);
      tap = (dz * 3 + dy) * 3 + dx;
    } else {
      tap = j;
      dz = j / (KH * KW); dy = (j / KW) % KH; dx = j % KW;
    }
    off = ((dz * T::IY + dy) * T::IX + dx) * T::XS;
  };
  // PACK == 1: toff is wave-uniform; PACK > 1: per lane (row i -> tap group*PACK + i/R, channel i%R)
  int toff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    int j = (t * TSPLIT + tslot) * PACK + (PACK == 1 ? 0 : (lane & 15) / R), tap;
    if (j >= NLIVE) j = NLIVE - 1;
    live_tap(j, tap, toff[t]);
    if (PACK > 1) toff[t] += (lane & 15) % R;
  }

  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
    // voxel groups (lz, ly, xq), 4 consecutive x per group = one MFMA k-step; branch-free body, unrolled so that
    // the fragment reads of the following groups overlap the MFMAs (hand-placed sched_barrier pipelining measured
    // slower: it pins hipcc's own interleave, tools/bench_layers.py)
#pragma unroll 4
    for (int it = 0; it < T::TVOX / 4 / KSPLIT; ++it) {   // constant trip count: MFMA loops only unroll evenly
      const int grp = it * KSPLIT + kslot;
      const int xq = grp & 3, ly = (grp >> 2) & 3, lz = grp >> 4;
      float b[NSUB], a[TPW];
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) b[nn] = Gw[(grp * 4) * GS + nn * 16];
      const float* Xg = Xw + ((lz * T::RS * T::IY + ly * T::RS) * T::IX + xq * 4 * T::RS) * T::XS;
#pragma unroll
      for (int t = 0; t < TPW; ++t) a[t] = Xg[toff[t]];
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn)
          acc[t][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[nn], acc[t][nn], 0, 0, 0);
    }
  }

  // partial[(chunk*KSPLIT + kslot)][tap][ci][CoP]; D[i][j=co]: lane -> co = lane&15, row i = (lane>>4)*4 + r
  float* out = partial + (int64_t)(chunk * KSPLIT + kslot) * TAPS * p.Ci * p.CoP;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (lane >> 4) * 4 + r;
      const int j = (t * TSPLIT + tslot) * PACK + i / R, ci = ci0 + i % R;
      if (j >= NLIVE || ci >= p.Ci) continue;
      int tap, off;
      live_tap(j, tap, off);
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) {
        const int co = cobase + nn * 16 + (lane & 15);
        if (co < p.CoP && nn * 16 + (lane & 15) < colimit) out[((int64_t)tap * p.Ci + ci) * p.CoP + co] = acc[t][nn][r];
      }
    }
  }
  if (TSKIP) {   // the folded-zero taps of this (ci subtile, co tile) region -- in every wave's own partial (`out` is per kslot)
    const unsigned live = parity_tapmask(qpar, false);
    for (int tap = 0; tap < TAPS; ++tap) {
      if ((live >> tap) & 1u) continue;
      for (int e = (KSPLIT == 1 ? tid : lane); e < 16 * NT; e += (KSPLIT == 1 ? 256 : 64)) {
        const int ci = ci0 + e / NT, col = e % NT, co = cobase + col;
        if (ci < p.Ci && col < colimit && co < p.CoP) out[((int64_t)tap * p.Ci + ci) * p.CoP + co] = 0.f;
      }
    }
  }
}

// PACK > 1 is used when the whole C_in fits one packed subtile (C_in <= 8: the folded 5x5x5 'finetune' conv).
// Packing the remainder subtile of C_in = 20 / 40 beside plain subtiles was measured and dropped: the dispatcher
// deals workgroups round-robin over the CUs (tools/probe_dispatch.py), the light packed workgroups finish early
// and the kernel time stays that of the plain ones; rebalancing by tile count lost to the per-tile staging cost.
template <int KD, int KH, int KW, int S, int NSUB, bool TSKIP, int PACK, bool IN = false>
__global__ void __launch_bounds__(256)
k_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial,
             CfunConv3dParams p, int ntz, int nty, int ntx, int ncisub, int ncot, int tiles_per_chunk, int ntiles, WgIn fin) {
  CFUN_DYN_LDS(float, smem);
  unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot; lid /= ncot;
  const int cis = lid % ncisub;
  const int chunk = lid / ncisub;
  wgrad_body<KD, KH, KW, S, NSUB, TSKIP, PACK, IN>(smem, x, g, partial, p, ntz, nty, ntx, cot, cis, chunk,
                                                   tiles_per_chunk, ntiles, fin);
}

// ---- C_in = 16*NPL + R with R = 16/PACK <= 8 remainder channels and 27 taps (the level-1 U-Net convs: C_in = 20 ->
// NPL 1, PACK 4; C_in = 40 -> NPL 2, PACK 2): ONE workgroup accumulates all input channels of its tiles -- NPL plain
// 16-channel subtiles of 27 tap rows each, and the remainder channels as ceil(27/PACK) packed rows of PACK taps x R
// channels (row i reads tap PACK*q + i/R, channel 16*NPL + i%R) -- instead of NPL+1 workgroups of 27 rows each, the
// last one mostly zero padding.  34 instead of 54 (68 instead of 81) MFMA rows per tile, G staged once, and all
// workgroups are equal (packing only the remainder in its own workgroup loses to the round-robin dispatcher).
template <int KD, int KH, int KW, int S>
constexpr bool wgrad_fused_shape() { return KD * KH * KW == 27; }

// 0: not applicable; else NPL | PACK << 4
inline int wgrad_fused_mode(const CfunConv3dParams& p, int nsub) {
  if (p.kd * p.kh * p.kw != 27 || (p.d2s && p.tap_skip) || nsub > 3) return 0;
  if (p.Ci > 16 && p.Ci <= 20) return 1 | (4 << 4);
  if (p.Ci > 32 && p.Ci <= 40 && p.stride == 1) return 2 | (2 << 4);
  return 0;
}

template <int KD, int KH, int KW, int S, int NSUB, int NPL, int PACK, bool IN = false>
__global__ void __launch_bounds__(256)
k_wgrad_fused(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial, CfunConv3dParams p,
              int ntz, int nty, int ntx, int ncot, int tiles_per_chunk, int ntiles, WgIn fin) {
  using T = WgTile<KD, KH, KW, S>;
  constexpr int TAPS = T::TAPS, NT = 16 * NSUB, GS = pad_row16(NT);
  constexpr int R = 16 / PACK, CH = 16 * NPL + R, C4 = CH / 4;
  constexpr int XS = CH == 20 ? 24 : CH;        // floats per staged voxel; fragment reads at most 2-way conflicted
  constexpr int TPW = cdiv(TAPS, 4);            // 7 plain tap rows per wave and subtile
  constexpr int NQ = cdiv(TAPS, PACK);          // packed rows
  constexpr int TPP = cdiv(NQ, 4);              // per wave
  constexpr int X_ITEMS = T::IVOX * C4, X_LOADS = cdiv(X_ITEMS, 256);
  constexpr int G_ITEMS = T::TVOX * (NT / 4), G_LOADS = cdiv(G_ITEMS, 256);
  CFUN_DYN_LDS(float, smem);
  float* Xl = smem;                      // [IVOX][XS]
  float* Gl = smem + T::IVOX * XS;       // [TVOX][GS]
  float4* Sl = reinterpret_cast<float4*>(Gl + T::TVOX * GS);      // IN: [C4 channel quads][2] stats rows (see wgrad_body)

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot;
  const int chunk = lid / ncot;
  const int cobase = cot * NT;
  const int sh = p.up2 ? 1 : 0;
  const int Dv = p.Di << sh, Hv = p.Hi << sh, Wv = p.Wi << sh;

  f32x4 acc[NPL][TPW][NSUB], accp[TPP][NSUB];
#pragma unroll
  for (int nn = 0; nn < NSUB; ++nn) {
#pragma unroll
    for (int s = 0; s < NPL; ++s)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[s][t][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < TPP; ++u) accp[u][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  float4 xin[X_LOADS], gin[G_LOADS];
  unsigned xvalid = 0, gvalid = 0;
  int n_table = -1;      // IN: the sample whose statistics the LDS table holds
  auto prefetch = [&](int tile) {   // unconditional loads + validity bits, as in wgrad_body
    int t = tile;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz;
    const int n = t / ntz;
    if constexpr (IN) {
      if (n != n_table) {
        n_table = n;
        if (tid < 2 * C4) {
          const int cs = (tid >> 1) * 4;
          Sl[tid] = (fin.stats && cs < p.Ci) ? reinterpret_cast<const float4*>(fin.stats + ((int64_t)n * p.Ci + cs) * 2)[tid & 1]
                                             : make_float4(0.f, 1.f, 0.f, 1.f);
        }
      }
    }
    const int z0 = tz * T::TD, y0 = ty * T::TH, x0 = tx * T::TW;
    xvalid = 0; gvalid = 0;
#pragma unroll
    for (int i = 0; i < X_LOADS; ++i) {
      const int it = tid + i * 256;
      const int idx = it / C4, c = (it - idx * C4) * 4;
      const int ix = idx % T::IX, iy = (idx / T::IX) % T::IY, iz = idx / (T::IX * T::IY);
      const int vz = z0 * S - p.pd + iz, vy = y0 * S - p.ph + iy, vx = x0 * S - p.pw + ix;
      const bool ok = (it < X_ITEMS) & (c < p.Ci) & (vz >= 0) & (vz < Dv) & (vy >= 0) & (vy < Hv) & (vx >= 0) & (vx < Wv);
      const unsigned off = ((((unsigned)n * p.Di + (vz >> sh)) * p.Hi + (vy >> sh)) * p.Wi + (vx >> sh)) * p.Ci + c;
      xin[i] = *reinterpret_cast<const float4*>(x + (ok ? off : 0u));
      xvalid |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      const int vox = it / (NT / 4), col = (it % (NT / 4)) * 4;
      const int lx = vox % T::TW, ly = (vox / T::TW) % T::TH, lz = vox / (T::TW * T::TH);
      const int oz = z0 + lz, oy = y0 + ly, ox = x0 + lx;
      bool ok = (it < G_ITEMS) & (cobase + col < p.Co) & (oz < p.Do) & (oy < p.Ho) & (ox < p.Wo);
      unsigned off;
      if (p.d2s) {   // hi-res gradient [N,2Do,2Ho,2Wo,Cq]: gather parity q, channel o
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
      if (it < X_ITEMS) {
        const int idx = it / C4, c = (it - idx * C4) * 4;
        *reinterpret_cast<float4*>(Xl + idx * XS + c) =
            keep((xvalid >> i) & 1u, IN ? norm_act_in4(xin[i], Sl[(c >> 2) * 2], Sl[(c >> 2) * 2 + 1], fin.act, fin.slope) : xin[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int it = tid + i * 256;
      if (it < G_ITEMS)
        *reinterpret_cast<float4*>(Gl + (it / (NT / 4)) * GS + (it % (NT / 4)) * 4) = keep((gvalid >> i) & 1u, gin[i]);
    }
  };

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = (t_begin + tiles_per_chunk < ntiles) ? t_begin + tiles_per_chunk : ntiles;
  const float* Xv = Xl + (lane >> 4) * T::RS * XS;   // this lane's voxel k = lane>>4 of a 4-voxel group
  const float* Gw = Gl + (lane >> 4) * GS + (lane & 15);
  auto tap_off = [&](int tap) {
    const int dz = tap / (KH * KW), dy = (tap / KW) % KH, dx = tap % KW;
    return ((dz * T::IY + dy) * T::IX + dx) * XS;
  };
  int toff[TPW], toffp[TPP];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {       // plain rows: tap t*4 + wave, channel lane&15 (surplus slots recompute tap 26)
    const int tap = t * 4 + wv;
    toff[t] = tap_off(tap < TAPS ? tap : TAPS - 1) + (lane & 15);
  }
#pragma unroll
  for (int u = 0; u < TPP; ++u) {       // packed rows: q = u*4 + wave; row i -> tap PACK*q + i/R, channel 16*NPL + i%R
    const int tap = (u * 4 + wv) * PACK + (lane & 15) / R;
    toffp[u] = tap_off(tap < TAPS ? tap : TAPS - 1) + 16 * NPL + (lane & 15) % R;
  }

  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
#pragma unroll 2
    for (int grp = 0; grp < T::TVOX / 4; ++grp) {
      const int xq = grp & 3, ly = (grp >> 2) & 3, lz = grp >> 4;
      float b[NSUB], a[NPL][TPW], ap[TPP];
#pragma unroll
      for (int nn = 0; nn < NSUB; ++nn) b[nn] = Gw[(grp * 4) * GS + nn * 16];
      const float* Xg = Xv + ((lz * T::RS * T::IY + ly * T::RS) * T::IX + xq * 4 * T::RS) * XS;
#pragma unroll
      for (int s = 0; s < NPL; ++s)
#pragma unroll
        for (int t = 0; t < TPW; ++t) a[s][t] = Xg[toff[t] + 16 * s];
#pragma unroll
      for (int u = 0; u < TPP; ++u) ap[u] = Xg[toffp[u]];
#pragma unroll
      for (int s = 0; s < NPL; ++s)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int nn = 0; nn < NSUB; ++nn)
            acc[s][t][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][t], b[nn], acc[s][t][nn], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TPP; ++u)
#pragma unroll
        for (int nn = 0; nn < NSUB; ++nn)
          accp[u][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[u], b[nn], accp[u][nn], 0, 0, 0);
    }
  }

  // partial[chunk][tap][ci][CoP]; D[i][j=co]: lane -> co = lane&15, row i = (lane>>4)*4 + r
  float* out = partial + (int64_t)chunk * TAPS * p.Ci * p.CoP;
#pragma unroll
  for (int nn = 0; nn < NSUB; ++nn) {
    const int co = cobase + nn * 16 + (lane & 15);
    if (co >= p.CoP) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (lane >> 4) * 4 + r;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tap = t * 4 + wv;
        if (tap < TAPS) {
#pragma unroll
          for (int s = 0; s < NPL; ++s) out[((int64_t)tap * p.Ci + 16 * s + i) * p.CoP + co] = acc[s][t][nn][r];
        }
      }
#pragma unroll
      for (int u = 0; u < TPP; ++u) {
        const int tap = (u * 4 + wv) * PACK + i / R, ci = 16 * NPL + i % R;
        if (tap < TAPS && ci < p.Ci) out[((int64_t)tap * p.Ci + ci) * p.CoP + co] = accp[u][nn][r];
      }
    }
  }
}

struct WgPlan {
  int ntz, nty, ntx, ntiles, ncisub, ncot, nchunks, tiles_per_chunk, nsub, kslots;
  // input prologue (CfunConvFusion): x is read as in_act((x - mean) * rstd); set by the caller after wgrad_plan
  const float* in_stats;
  int in_act;
  float in_slope;
};


template <int KD, int KH, int KW, int S>
WgPlan wgrad_plan(const CfunConv3dParams& p, int nsub) {
  using T = WgTile<KD, KH, KW, S>;
  WgPlan w;
  w.nsub = nsub;
  w.ntz = cdiv(p.Do, T::TD); w.nty = cdiv(p.Ho, T::TH); w.ntx = cdiv(p.Wo, T::TW);
  w.ntiles = p.N * w.ntz * w.nty * w.ntx;
  w.ncisub = (wgrad_fused_shape<KD, KH, KW, S>() && wgrad_fused_mode(p, nsub)) ? 1 : cdiv(p.Ci, 16);
  w.ncot = (p.d2s && p.tap_skip) ? 8 * cdiv(p.Co >> 3, 16 * nsub) : cdiv(p.CoP, 16 * nsub);
  int want = 512 / (w.ncisub * w.ncot);   // ~2 workgroups per CU in total; fewer partials to reduce
  if (want < 1) want = 1;
  if (want > w.ntiles) want = w.ntiles;
  if (want < 1) want = 1;
  w.tiles_per_chunk = cdiv(w.ntiles, want);
  if (w.tiles_per_chunk < 1) w.tiles_per_chunk = 1;
  w.nchunks = cdiv(w.ntiles, w.tiles_per_chunk);
  if (w.nchunks < 1) w.nchunks = 1;
  w.kslots = (p.d2s && p.tap_skip && T::TAPS == 27 && nsub <= 2) ? 4 : T::KSPLIT;      // (TSKIP: voxel groups over the waves, wgrad_body)
  w.in_stats = nullptr; w.in_act = 0; w.in_slope = 0.f;
  return w;
}

template <int KD, int KH, int KW, int S, int NSUB>
int launch_wgrad_mfma(const float* x, const float* g, float* partial, const CfunConv3dParams& p, const WgPlan& w,
                      hipStream_t st) {
  using T = WgTile<KD, KH, KW, S>;
  constexpr int NT = 16 * NSUB, GS = pad_row16(NT);
  const size_t lds = (size_t)(T::IVOX * T::XS + T::TVOX * GS) * sizeof(float) + 8 * sizeof(float4);      // (+ stats table)
  // the kernel addresses x and g with 32-bit element offsets
  const int64_t lim = (int64_t)1 << 31;
  if ((int64_t)p.N * p.Di * p.Hi * p.Wi * p.Ci >= lim || (int64_t)p.N * p.Do * p.Ho * p.Wo * p.Co >= lim) return CFUN_EINVAL;
  if constexpr (wgrad_fused_shape<KD, KH, KW, S>() && NSUB <= 3) {
    const int mode = wgrad_fused_mode(p, NSUB);
    if (mode) {
      const int xs = (mode & 15) == 1 ? 24 : 40;
      const size_t ldsf = (size_t)(T::IVOX * xs + T::TVOX * GS) * sizeof(float) + 2 * (xs / 4) * sizeof(float4);   // (+ stats table)
      const bool in = w.in_stats != nullptr || w.in_act != CFUN_ACT_NONE;
      void (*kf)(const float*, const float*, float*, CfunConv3dParams, int, int, int, int, int, int, WgIn) =
          in ? k_wgrad_fused<KD, KH, KW, S, NSUB, 1, 4, true> : k_wgrad_fused<KD, KH, KW, S, NSUB, 1, 4, false>;
      if constexpr (S == 1) {
        if ((mode & 15) == 2) kf = in ? k_wgrad_fused<KD, KH, KW, S, NSUB, 2, 2, true> : k_wgrad_fused<KD, KH, KW, S, NSUB, 2, 2, false>;
      }
      if (ldsf > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
        if (e != hipSuccess) return (int)e;
      }
      hipLaunchKernelGGL(kf, dim3((unsigned)(w.nchunks * w.ncot)), dim3(256), ldsf, st, x, g, partial, p, w.ntz, w.nty,
                         w.ntx, w.ncot, w.tiles_per_chunk, w.ntiles, WgIn{w.in_stats, w.in_act, w.in_slope});
      CFUN_LAUNCH_CHECK();
      return CFUN_OK;
    }
  }
  const bool in = w.in_stats != nullptr || w.in_act != CFUN_ACT_NONE;
  auto kern = k_wgrad_mfma<KD, KH, KW, S, NSUB, false, 1, false>;
  auto pick = [&](auto in_c) {
    constexpr bool I = decltype(in_c)::value;
    kern = k_wgrad_mfma<KD, KH, KW, S, NSUB, false, 1, I>;
    if constexpr (KD * KH * KW == 27) {   // C_in <= 8: PACK taps per A fragment
      const int pack = w.ncisub > 1 ? 1 : p.Ci <= 4 ? 4 : p.Ci <= 8 ? 2 : 1;
      const bool ts = S == 1 && p.d2s && p.tap_skip;
      if constexpr (S == 1) {
        if (ts) kern = pack == 4 ? k_wgrad_mfma<KD, KH, KW, S, NSUB, true, 4, I>
                     : pack == 2 ? k_wgrad_mfma<KD, KH, KW, S, NSUB, true, 2, I>
                                 : k_wgrad_mfma<KD, KH, KW, S, NSUB, true, 1, I>;
      }
      if (!ts) kern = pack == 4 ? k_wgrad_mfma<KD, KH, KW, S, NSUB, false, 4, I>
                    : pack == 2 ? k_wgrad_mfma<KD, KH, KW, S, NSUB, false, 2, I>
                                : k_wgrad_mfma<KD, KH, KW, S, NSUB, false, 1, I>;
    }
  };
  if (in) pick(std::true_type{}); else pick(std::false_type{});
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t nblk = (int64_t)w.nchunks * w.ncisub * w.ncot;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, x, g, partial, p, w.ntz, w.nty, w.ntx, w.ncisub,
                     w.ncot, w.tiles_per_chunk, w.ntiles, WgIn{w.in_stats, w.in_act, w.in_slope});
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

template <int KD, int KH, int KW, int S>
int dispatch_wgrad(const float* x, const float* g, float* partial, const CfunConv3dParams& p, const WgPlan& w,
                   hipStream_t st) {
  if constexpr (max_nsub<KD, KH, KW, S>() == 1) {
    if (w.nsub != 1) return CFUN_EINVAL;
    return launch_wgrad_mfma<KD, KH, KW, S, 1>(x, g, partial, p, w, st);
  } else {
    switch (w.nsub) {
      case 1: return launch_wgrad_mfma<KD, KH, KW, S, 1>(x, g, partial, p, w, st);
      case 2: return launch_wgrad_mfma<KD, KH, KW, S, 2>(x, g, partial, p, w, st);
      case 3: return launch_wgrad_mfma<KD, KH, KW, S, 3>(x, g, partial, p, w, st);
      case 4: return launch_wgrad_mfma<KD, KH, KW, S, 4>(x, g, partial, p, w, st);
      default: return launch_wgrad_mfma<KD, KH, KW, S, 5>(x, g, partial, p, w, st);
    }
  }
}

}  // namespace cfun_mfma

// per-shape translation units (conv3d_mfma_*.hip) export these C++ entry points
#define CFUN_MFMA_DECL(NAME)                                                                                        \
  int cfun_mfma_fwd_##NAME(int nsub, const float* x, const float* wp, const float* scale, const float* shift,       \
                           const float* res, float* y, const CfunConv3dParams& p, const cfun_mfma::ConvMode& flip,  \
                           void* ws, size_t wsb, hipStream_t st);                                                   \
  size_t cfun_mfma_fwd_ws_##NAME(int nsub, const CfunConv3dParams& p, const cfun_mfma::ConvMode& md);               \
  void cfun_mfma_wgrad_plan_##NAME(const CfunConv3dParams& p, int nsub, cfun_mfma::WgPlan* w);                      \
  int cfun_mfma_wgrad_##NAME(const float* x, const float* g, float* partial, const CfunConv3dParams& p,             \
                             const cfun_mfma::WgPlan& w, hipStream_t st);

#define CFUN_MFMA_DEFINE(NAME, KD, KH, KW, S)                                                                       \
  int cfun_mfma_fwd_##NAME(int nsub, const float* x, const float* wp, const float* scale, const float* shift,       \
                           const float* res, float* y, const CfunConv3dParams& p, const cfun_mfma::ConvMode& flip,  \
                           void* ws, size_t wsb, hipStream_t st) {                                                  \
    return cfun_mfma::dispatch_nsub<KD, KH, KW, S>(nsub, x, wp, scale, shift, res, y, p, flip, ws, wsb, st);        \
  }                                                                                                                 \
  size_t cfun_mfma_fwd_ws_##NAME(int nsub, const CfunConv3dParams& p, const cfun_mfma::ConvMode& md) {              \
    return cfun_mfma::fwd_workspace<KD, KH, KW, S>(nsub, p, md);                                                    \
  }                                                                                                                 \
  void cfun_mfma_wgrad_plan_##NAME(const CfunConv3dParams& p, int nsub, cfun_mfma::WgPlan* w) {                     \
    *w = cfun_mfma::wgrad_plan<KD, KH, KW, S>(p, nsub);                                                             \
  }                                                                                                                 \
  int cfun_mfma_wgrad_##NAME(const float* x, const float* g, float* partial, const CfunConv3dParams& p,             \
                             const cfun_mfma::WgPlan& w, hipStream_t st) {                                          \
    return cfun_mfma::dispatch_wgrad<KD, KH, KW, S>(x, g, partial, p, w, st);                                       \
  }

CFUN_MFMA_DECL(k333s1)
CFUN_MFMA_DECL(k333s2)
CFUN_MFMA_DECL(k111s1)
CFUN_MFMA_DECL(k111s2)
CFUN_MFMA_DECL(k133s1)
CFUN_MFMA_DECL(k311s1)
CFUN_MFMA_DECL(k555s1)
CFUN_MFMA_DECL(k222s1)
