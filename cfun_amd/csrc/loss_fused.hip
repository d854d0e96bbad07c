// The 'finetune' mask losses as ONE forward and ONE backward pass over the hi-res tensors (round 6):
//
//   forward   logits, labels -> probs = softmax(logits) (model.py:799), cross entropy (model.py:909-935) and the 3-D Sobel
//             edge loss (model.py:938-981, channel 0 twice in the magnitude: 969-972)
//   backward  probs, labels, (g_ce, g_edge) -> dlogits of both losses through the softmax
//
// Until round 5 the forward was three passes (k_softmax_fwd, k_ce_fwd, k_edge_march2 -- the last one also wrote a 56-byte
// coefficient record per voxel, 1.5 GB at 4 x 192^3) and the backward gathered that record field: 5.2 + 3.4 GB of traffic for
// 1.8 + 1.8 GB of tensors, and the march re-loaded every probability 6 times from global memory (12 neighbour loads per two
// outputs), which made it VALU-issue bound at 0.15 of the HBM peak.  Here a workgroup owns a 32 (y) x 16 (x) output tile of
// one z segment and MARCHES along z: each plane's 34 x 18 input voxels are loaded ONCE (two 16-byte loads per voxel),
// soft-maxed, written out and deposited in LDS (class-major planes, row stride padded to 24 floats: conflict-free for the
// 4-row x 16-column wave footprint); the in-plane Sobel sums of a thread's output PAIR (y, y+1) come from 12 LDS reads per
// class, the z direction from a register ring of three planes, exactly the arithmetic of k_edge_march2.  In training the
// forward also applies the z part of the TRANSPOSED stencil to the coefficients while they are in registers and writes that
// field U (same size as the old coefficient record); the backward is then a 2-D stencil per plane through an LDS tile plus the
// softmax / cross-entropy gradient -- no coupling along z, no register ring, HBM-bound.
// (Measured and dropped on MI355X, round 6: a backward that RECOMPUTES the coefficients from the probabilities -- no field at
// all, 3.6 GB instead of 6.8 GB per step -- needs three register rings: 300 VGPRs = one wave per SIMD, 2.4 ms; capped at 256
// VGPRs it spills 67 dwords per lane, 3.2 ms; the gather it was to replace took 1.25 ms.  profiles/round6_losses_*.log)
#include "common.h"

namespace {

constexpr int kFB = 256;                     // threads per workgroup: 16 row pairs x 16 columns
constexpr int kTY = 32, kTX = 16;            // output tile
constexpr int kIY = kTY + 2, kIX = kTX + 2;  // its input footprint
constexpr int kIXP = 24;                     // padded LDS row stride (floats): rows 2*ty of a wave's 4 ty land 16 banks apart
constexpr int kPlane = kIY * kIXP;
constexpr int kMaxBlocksF = 2048;

// softmax arithmetic of the fused forward: v_exp_f32 / v_log_f32 (1 ulp on the base-2 function; the base change costs <= 2e-6
// relative at |x| = 30) instead of the ~12-instruction expf / logf expansions, and ONE division per voxel -- the pass is
// VALU-bound, the eight exps and eight divisions per voxel were 40 % of its instructions
__device__ __forceinline__ float cfun_softmax_exp(float v) {
#ifdef CFUN_HIP_EMULATION
  return expf(v);
#else
  return __expf(v);
#endif
}
__device__ __forceinline__ float cfun_softmax_log(float v) {
#ifdef CFUN_HIP_EMULATION
  return logf(v);
#else
  return __logf(v);
#endif
}

__device__ __forceinline__ double block_sum_f(double v, double* red, int nwaves) {
  v = cfun_wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nwaves; ++w) s += red[w];
  __syncthreads();
  return s;
}

// both kernels sit at ~250 VGPRs: two waves per SIMD (hipcc left to itself takes 257 / 280 and one)
#ifdef CFUN_HIP_EMULATION
#define CFUN_OCC2
#define CFUN_OCC_BWD(VPT)
#define CFUN_OCC_FWD(VPT)
#else
#define CFUN_OCC2 __attribute__((amdgpu_waves_per_eu(2)))
#define CFUN_OCC_BWD(VPT) __attribute__((amdgpu_waves_per_eu((VPT) == 2 ? 2 : 4)))
#define CFUN_OCC_FWD(VPT) __attribute__((amdgpu_waves_per_eu((VPT) == 2 ? 2 : 4)))
#endif

template <int CT>
struct PlaneF {      // in-plane Sobel sums of one (y, x) column at one z: classes 1..CT-1 at index c-1; targets packed
  float dy[CT - 1], sm[CT - 1];
  uint32_t tdy[2], tsm[2];
};

// the in-plane sums of the thread's VPT outputs (rows r0 .. r0 + VPT - 1 of the tile, column c0) from the LDS plane: the x sums
// of rows r0 .. r0 + VPT + 1 serve all of them
template <int CT, int VPT>
__device__ __forceinline__ void plane_rows(const float* __restrict__ lp, const uint8_t* __restrict__ ll, int r0, int c0,
                                           PlaneF<CT> (&P)[VPT]) {
  uint32_t t[VPT + 2][2];
#pragma unroll
  for (int j = 0; j < VPT + 2; ++j) {
    const int base = (r0 + j) * kIXP + c0;
    const unsigned l0 = ll[base], l1 = ll[base + 1], l2 = ll[base + 2];
    // packed 8-bit fields (classes 0-3 | 4-7), weights 1, 2, 1 along x; labels >= CT contribute nothing (k_edge_march2)
    const uint32_t v0 = l0 < (unsigned)CT ? 1u << ((l0 & 3u) * 8u) : 0u;
    const uint32_t v1 = l1 < (unsigned)CT ? 2u << ((l1 & 3u) * 8u) : 0u;
    const uint32_t v2 = l2 < (unsigned)CT ? 1u << ((l2 & 3u) * 8u) : 0u;
    t[j][0] = ((l0 & 4u) ? 0u : v0) + ((l1 & 4u) ? 0u : v1) + ((l2 & 4u) ? 0u : v2);
    t[j][1] = ((l0 & 4u) ? v0 : 0u) + ((l1 & 4u) ? v1 : 0u) + ((l2 & 4u) ? v2 : 0u);
  }
#pragma unroll
  for (int o = 0; o < VPT; ++o)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      P[o].tdy[h] = t[o][h] + 0x04040404u - t[o + 2][h];          // dy + 4 per field
      P[o].tsm[h] = t[o][h] + 2u * t[o + 1][h] + t[o + 2][h];
    }
#pragma unroll
  for (int c = 0; c < CT - 1; ++c) {
    float a[VPT + 2];
#pragma unroll
    for (int j = 0; j < VPT + 2; ++j) {
      const float* q = lp + c * kPlane + (r0 + j) * kIXP + c0;
      a[j] = (q[0] + 2.f * q[1]) + q[2];
    }
#pragma unroll
    for (int o = 0; o < VPT; ++o) {
      P[o].dy[c] = a[o] - a[o + 2];
      P[o].sm[c] = (a[o] + a[o + 2]) + 2.f * a[o + 1];
    }
  }
}

// one output voxel from the ring (planes zo, zo+1, zo+2): accumulates (|grad p| - |grad t|)^2 into `acc` and / or returns the
// coefficients dL/dc0, dL/dc1 per class (scaled by gs) in ov[2c], ov[2c+1] -- the formulas of k_edge_march2
template <int CT, bool LOSS, bool COEF>
__device__ __forceinline__ void edge_point(const PlaneF<CT>& A0, const PlaneF<CT>& A1, const PlaneF<CT>& A2, float gs, bool live,
                                           float& acc, float (&ov)[2 * (CT - 1)]) {
  uint32_t t0b[2], t1b[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    t0b[h] = A0.tdy[h] + 2u * A1.tdy[h] + A2.tdy[h];            // t0 + 16 per 8-bit field
    t1b[h] = A0.tsm[h] + 0x10101010u - A2.tsm[h];               // t1 + 16
  }
#pragma unroll
  for (int c = 0; c < CT - 1; ++c) {
    const float p0 = A0.dy[c] + 2.f * A1.dy[c] + A2.dy[c], p1 = A0.sm[c] - A2.sm[c];
    const int cls = c + 1;
    const float t0 = (float)((int)((t0b[cls >> 2] >> ((cls & 3) * 8)) & 0xffu) - 16);
    const float t1 = (float)((int)((t1b[cls >> 2] >> ((cls & 3) * 8)) & 0xffu) - 16);
    const float sp = p0 * p0 + p1 * p1 + p0 * p0;                  // channel 0 twice (model.py:969-972)
    const float ip = cfun_fast_rsq(sp);                            // inf at 0
    const float pm = sp > 0.f ? sp * ip : 0.f;
    const float tm = cfun_fast_sqrt(t0 * t0 + t1 * t1 + t0 * t0);
    if (LOSS) {
      const float d = pm - tm;
      acc += live ? d * d : 0.f;
    }
    if (COEF) {
      const float k = gs * (pm - tm) * ip;     // 0 * inf -> NaN exactly where torch's sqrt backward gives 0/0 (App. A-13)
      ov[2 * c] = live ? k * 2.f * p0 : 0.f;
      ov[2 * c + 1] = live ? k * p1 : 0.f;
    }
  }
}

template <int CT>
__device__ __forceinline__ void load_vox(const float* __restrict__ src, int64_t v, float (&x)[CT]) {
  if (CT % 4 == 0) {
#pragma unroll
    for (int q = 0; q < CT / 4; ++q) {
      const float4 f = reinterpret_cast<const float4*>(src + v * CT)[q];
      x[4 * q] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < CT; ++c) x[c] = src[v * CT + c];
  }
}

template <int CT>
__device__ __forceinline__ void store_vox(float* __restrict__ dst, int64_t v, const float (&x)[CT]) {
  if (CT % 4 == 0) {
#pragma unroll
    for (int q = 0; q < CT / 4; ++q)
      reinterpret_cast<float4*>(dst + v * CT)[q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
  } else {
#pragma unroll
    for (int c = 0; c < CT; ++c) dst[v * CT + c] = x[c];
  }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// partial[b] = block b's sum of -log softmax[label] over the voxels it owns; partial[kMaxBlocksF + b] = its sum of the edge terms.
// WRITE_U (training): the backward's operand -- per OUTPUT column (y, x) in [0, Ho) x [0, Wo) and INPUT plane z in [0, D) the z
// part of the transposed stencil applied to the unit-gradient coefficients dc = (dL/dc0, dL/dc1):
//     U0[z] = dc0[z] + 2 dc0[z-1] + dc0[z-2],   U1[z] = dc1[z] - dc1[z-2]        (dc = 0 outside [0, Do))
// as 2 (CT-1) floats per (z, y, x): (U0, U1) of class 1, of class 2, ...  The backward is then a 2-D stencil per plane.
// A segment that writes U starts its march two planes early (dc[z0-2], dc[z0-1] belong to the previous segment's loss but
// to this segment's U).
template <int CT, bool WRITE_U, int VPT>
__global__ void __launch_bounds__(kFB * 2 / VPT) CFUN_OCC_FWD(VPT)
k_mask_fused_fwd(const float* __restrict__ logits, const uint8_t* __restrict__ labels, float* __restrict__ probs,
                 double* __restrict__ partial, float* __restrict__ U, int n, int D, int H, int W, int ZS) {
  static_assert(CT >= 2 && CT <= 8, "packed target sums: 8 classes");
  CFUN_DYN_LDS(float, lds);
  float* lp = lds;                                                     // [2][CT-1][kPlane]
  uint8_t* ll = reinterpret_cast<uint8_t*>(lds + 2 * (CT - 1) * kPlane);   // [2][kPlane]
  constexpr int NT = kFB * 2 / VPT;                                    // VPT outputs (tile rows) per thread: 2 = 256 threads, 1 = 512
  __shared__ double red[NT / 64];
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int nseg = (D + ZS - 1) / ZS, tY = (Ho + kTY - 1) / kTY, tX = (Wo + kTX - 1) / kTX;
  const int64_t tiles = (int64_t)n * nseg * tY * tX;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const float gs = 2.f / ((float)Do * (float)Ho * (float)Wo * (float)n);
  double ce_acc = 0.0, ed_acc = 0.0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    int64_t t = tile;
    const int bx = (int)(t % tX); t /= tX;
    const int by = (int)(t % tY); t /= tY;
    const int seg = (int)(t % nseg);
    const int64_t r = t / nseg;
    const int x0 = bx * kTX, y0 = by * kTY, z0 = seg * ZS;
    const bool lastx = bx == tX - 1, lasty = by == tY - 1;
    const int z1 = z0 + ZS < D ? z0 + ZS : D;                          // owned planes [z0, z1): probs, CE, U; outputs zq < Do
    const int zl0 = WRITE_U ? z0 - 2 : z0, zl1 = z1 + 2;               // step zl loads plane zl and emits output plane zl - 2
    const int64_t nbase = r * D * H * W;
    const int oy = y0 + VPT * ty, ox = x0 + tx;                        // the thread's outputs (oy .. oy + VPT - 1) x ox
    PlaneF<CT> P[3][VPT];
    float d1[VPT][WRITE_U ? 2 * (CT - 1) : 1], d2[VPT][WRITE_U ? 2 * (CT - 1) : 1];     // dc[zq-1], dc[zq-2]
#pragma unroll
    for (int o = 0; o < VPT; ++o) {
#pragma unroll
      for (int c = 0; c < (WRITE_U ? 2 * (CT - 1) : 1); ++c) { d1[o][c] = 0.f; d2[o][c] = 0.f; }
#pragma unroll
      for (int k = 0; k < 2; ++k) {                                    // (the ring's first two steps read these: masked by `live`)
#pragma unroll
        for (int c = 0; c < CT - 1; ++c) { P[k][o].dy[c] = 0.f; P[k][o].sm[c] = 0.f; }
        P[k][o].tdy[0] = P[k][o].tdy[1] = P[k][o].tsm[0] = P[k][o].tsm[1] = 0u;
      }
    }
    float accf = 0.f;
    for (int zl = zl0; zl < zl1; ++zl) {
      float* bp = lp + (zl & 1) * (CT - 1) * kPlane;
      uint8_t* bl = ll + (zl & 1) * kPlane;
      const bool inz = zl >= 0 && zl < D;
      const bool ownz = zl >= z0 && zl < z1;
      // ---- phase 1: the plane's 34 x 18 voxels, each loaded once: softmax, probs out, cross entropy, fg probs + label -> LDS
      // (three voxel slots per thread, branch-free: clamped addresses, so that all loads of the plane are in flight together)
      if (inz) {
        constexpr int NS = (kIY * kIX + NT - 1) / NT;
        float xs[NS][CT];
        unsigned labs[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const int i = threadIdx.x + q * NT;
          const int ly = i / kIX, lx = i - ly * kIX;
          const bool in = i < kIY * kIX && y0 + ly < H && x0 + lx < W;
          const int64_t v = in ? nbase + ((int64_t)zl * H + (y0 + ly)) * W + (x0 + lx) : nbase;
          load_vox<CT>(logits, v, xs[q]);
          labs[q] = in ? labels[v] : 255u;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const int i = threadIdx.x + q * NT;
          const int ly = i / kIX, lx = i - ly * kIX;
          const bool in = i < kIY * kIX && y0 + ly < H && x0 + lx < W;
          float (&x)[CT] = xs[q];
          const unsigned lab = labs[q];
          float m = -INFINITY;
#pragma unroll
          for (int c = 0; c < CT; ++c) m = fmaxf(m, x[c]);
          float xl = 0.f;
#pragma unroll
          for (int c = 0; c < CT; ++c)
            if (c == (int)lab) xl = x[c];
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < CT; ++c) { x[c] = cfun_softmax_exp(x[c] - m); s += x[c]; }
          const float inv = 1.f / s;
#pragma unroll
          for (int c = 0; c < CT; ++c) x[c] = in ? x[c] * inv : 0.f;
          if (in && ownz && (ly < kTY || lasty) && (lx < kTX || lastx)) {     // every voxel has exactly one owner
            store_vox<CT>(probs, nbase + ((int64_t)zl * H + (y0 + ly)) * W + (x0 + lx), x);
            ce_acc += (double)((m + cfun_softmax_log(s)) - xl);
          }
          if (i < kIY * kIX) {
#pragma unroll
            for (int c = 1; c < CT; ++c) bp[(c - 1) * kPlane + ly * kIXP + lx] = x[c];
            bl[ly * kIXP + lx] = (uint8_t)lab;
          }
        }
      }
      __syncthreads();
      // ---- phase 2: in-plane sums of the output pair -> ring; output plane zq = zl - 2 (valid once three planes are in)
      if (inz) plane_rows<CT, VPT>(bp, bl, VPT * ty, tx, P[2]);
      const int zq = zl - 2;
      const bool have = zq >= 0 && zq < Do && zl - zl0 >= 2;           // the ring holds planes zq, zq+1, zq+2
      const bool mine = zq >= z0 && zq < z1;                           // this segment's loss terms / U planes
#pragma unroll
      for (int o = 0; o < VPT; ++o) {
        const bool live = have && oy + o < Ho && ox < Wo;
        float d0[2 * (CT - 1)];
        float term = 0.f;
        edge_point<CT, true, WRITE_U>(P[0][o], P[1][o], P[2][o], gs, live, term, d0);
        accf += mine ? term : 0.f;
        if (WRITE_U) {
          if (mine && oy + o < Ho && ox < Wo) {
            float2* up = reinterpret_cast<float2*>(U + ((((r * D + zq) * Ho + (oy + o)) * Wo + ox)) * (2 * (CT - 1)));
#pragma unroll
            for (int c = 0; c < CT - 1; ++c)
              up[c] = make_float2((d0[2 * c] + 2.f * d1[o][2 * c]) + d2[o][2 * c], d0[2 * c + 1] - d2[o][2 * c + 1]);
          }
#pragma unroll
          for (int c = 0; c < 2 * (CT - 1); ++c) { d2[o][c] = d1[o][c]; d1[o][c] = d0[c]; }
        }
      }
#pragma unroll
      for (int o = 0; o < VPT; ++o) { P[0][o] = P[1][o]; P[1][o] = P[2][o]; }
    }
    ed_acc += (double)accf;
    __syncthreads();        // the next tile's first plane reuses an LDS buffer this tile's last phase 2 may still read
  }
  const double s0 = block_sum_f(ce_acc, red, NT / 64);
  const double s1 = block_sum_f(ed_acc, red, NT / 64);
  if (threadIdx.x == 0) { partial[blockIdx.x] = s0; partial[kMaxBlocksF + blockIdx.x] = s1; }
}

__global__ void k_finalize_sum2(const double* __restrict__ partial, int blocks, double mul0, double mul1, float* __restrict__ out) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 64) { a += partial[i]; b += partial[kMaxBlocksF + i]; }
  a = cfun_wave_sum_d(a);
  b = cfun_wave_sum_d(b);
  if (threadIdx.x == 0) { out[0] = (float)(a * mul0); out[1] = (float)(b * mul1); }
}

// --------------------------------------------------------------------------------------------------------------- backward
// One input plane per step, no coupling along z left:  g(y,x) = sum_{j,i} B[j]A[i] U0(y-j,x-i) + A[j]A[i] U1(y-j,x-i), then
//   dlogits = softmax_bwd(probs, g_edge * g) + g_ce / nvox * (probs - onehot(label)).
// A workgroup owns 32 x 16 input voxels of kBZ consecutive planes; the 34 x 18 U columns a plane's voxels read ((y-2 .. y) x
// (x-2 .. x), zero outside [0, Ho) x [0, Wo)) are staged in LDS once per plane (class-major, the forward's padded rows), the
// NEXT plane's granules are already in flight in registers while the current one is finished; a thread finishes the voxel
// pair (y, y+1).
constexpr int kUSP = kPlane + 25;            // backward: LDS plane stride (841 = 9 mod 32: the staging writes of one lane group
                                             // -- consecutive class pairs of the same column -- land 9 banks apart)
constexpr int kBZ = 24;                      // backward: planes per workgroup (one (y, x) tile, consecutive z)

// VPT voxels of the tile column pair (y, y+1) per thread: 2 = 256 threads (a pair shares its four U rows: 84 LDS reads per
// voxel, ~250 VGPRs, two waves per SIMD), 1 = 512 threads (126 reads per voxel, half the registers, more waves in flight)
template <int CT, int VPT>
__global__ void __launch_bounds__(kFB * 2 / VPT) CFUN_OCC_BWD(VPT)
k_mask_fused_bwd(const float* __restrict__ U, const float* __restrict__ probs, const uint8_t* __restrict__ labels,
                 const float* __restrict__ g2, float* __restrict__ dlogits, int n, int D, int H, int W) {
  CFUN_DYN_LDS(float, lds);                                            // [2 (CT-1)][kUSP]: U0 planes 0 .. CT-2, U1 planes CT-1 ..
  constexpr int NT = kFB * 2 / VPT;
  constexpr int NP = CT - 1;                                           // class pairs (U0, U1)
  constexpr int RG = kIX * NP;                                         // float2 granules per tile row (126 at 8 classes)
  constexpr int RPP = NT / RG;                                         // tile rows staged per pass
  constexpr int NLD = (kIY + RPP - 1) / RPP;                           // passes = granules per thread
  static_assert(RPP >= 1, "a tile row per pass");
  const int Ho = H - 2, Wo = W - 2;
  const int tY = (H + kTY - 1) / kTY, tX = (W + kTX - 1) / kTX, nzc = (D + kBZ - 1) / kBZ;
  const int64_t tiles = (int64_t)n * nzc * tY * tX;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const float ge = g2[1];
  const float gce = g2[0] / (float)((int64_t)n * D * H * W);
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    int64_t t = tile;
    const int bx = (int)(t % tX); t /= tX;
    const int by = (int)(t % tY); t /= tY;
    const int zc = (int)(t % nzc);
    const int64_t r = t / nzc;
    const int x0 = bx * kTX, y0 = by * kTY;                            // first owned voxel; U tile origin = (y0 - 2, x0 - 2)
    const int zb = zc * kBZ, ze = zb + kBZ < D ? zb + kBZ : D;
    // Staging map: a tile row is RG float2 granules, contiguous in memory.  Thread (h, e) = (tid / RG, tid % RG) takes
    // granule e of rows h, h + RPP, h + 2 RPP, ...: memory and LDS offsets are affine in the pass index (one VGPR each, the
    // row stride goes into scalar offsets) -- a flat tid + q * 256 map costs a 64-bit address per granule (355 VGPRs).
    const int h = threadIdx.x / RG, e = threadIdx.x - h * RG;
    const bool lane_on = threadIdx.x < RPP * RG;
    const int lx = e / NP, k = e - lx * NP;
    const int ux = x0 - 2 + lx;
    const bool col_ok = lane_on && ux >= 0 && ux < Wo;
    const int lds0 = lane_on ? k * kUSP + h * kIXP + lx : NP * kUSP - 1;      // (idle lanes: a dead slot in the last plane's padding)
    const int lrow = lane_on ? RPP * kIXP : 0;
    const int64_t grow = (int64_t)Wo * NP;                             // granules per memory row
    const int64_t g0 = ((int64_t)(y0 - 2 + h) * Wo + (x0 - 2)) * NP + e;
    float2 pre[NLD];
#define CFUN_STAGE_LOAD(up)                                                         \
  _Pragma("unroll") for (int q = 0; q < NLD; ++q) {                                 \
    const int uy = y0 - 2 + h + RPP * q;                                            \
    const bool ok = col_ok && h + RPP * q < kIY && uy >= 0 && uy < Ho;              \
    const float2 v = (up)[ok ? g0 + (int64_t)(RPP * q) * grow : 0];   /* branch-free: clamped address, selected value */ \
    pre[q] = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);                           \
  }
    {
      const float2* up = reinterpret_cast<const float2*>(U) + (r * D + zb) * (int64_t)Ho * Wo * NP;
      CFUN_STAGE_LOAD(up)
    }
    for (int z = zb; z < ze; ++z) {
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const int lo = (h + RPP * q < kIY) ? lds0 + q * lrow : NP * kUSP - 1;      // (rows beyond the tile: the dead slot)
        lds[lo] = pre[q].x; lds[NP * kUSP + lo] = pre[q].y;
      }
      __syncthreads();
      if (z + 1 < ze) {        // the next plane's granules travel while this plane is finished
        const float2* up = reinterpret_cast<const float2*>(U) + (r * D + z + 1) * (int64_t)Ho * Wo * NP;
        CFUN_STAGE_LOAD(up)
      }
      const int ry = 2 + VPT * ty, rx = 2 + tx;                        // tile-local U position of the thread's first voxel
      const int gy = y0 + VPT * ty, gx = x0 + tx;
      const int64_t v0 = ((r * D + z) * (int64_t)H + gy) * W + gx;
      float pr[VPT][CT];
      int lab[VPT];
#pragma unroll
      for (int o = 0; o < VPT; ++o) {
        const bool in = gy + o < H && gx < W;
        const int64_t vv = in ? v0 + (int64_t)o * W : 0;               // (clamped: branch-free)
        load_vox<CT>(probs, vv, pr[o]);
        lab[o] = labels[vv];
      }
      float g[VPT][CT - 1];
#pragma unroll
      for (int c = 0; c < CT - 1; ++c) {
        float s0[VPT + 2], s1[VPT + 2];
#pragma unroll
        for (int j = 0; j < VPT + 2; ++j) {                            // x sums (A = 1,2,1 over x, x-1, x-2) of U rows ry-2 .. ry+VPT-1
          const float* u0 = lds + c * kUSP + (ry - 2 + j) * kIXP + rx;
          const float* u1 = lds + (NP + c) * kUSP + (ry - 2 + j) * kIXP + rx;
          s0[j] = (u0[0] + 2.f * u0[-1]) + u0[-2];
          s1[j] = (u1[0] + 2.f * u1[-1]) + u1[-2];
        }
        // voxel row y = ry + o receives from U rows y (B = 1, A = 1), y-1 (B = 0, A = 2), y-2 (B = -1, A = 1)
#pragma unroll
        for (int o = 0; o < VPT; ++o) g[o][c] = ge * ((s0[2 + o] - s0[o]) + ((s1[2 + o] + s1[o]) + 2.f * s1[1 + o]));
      }
#pragma unroll
      for (int o = 0; o < VPT; ++o) {
        if (gy + o < H && gx < W) {
          float dot = 0.f;
#pragma unroll
          for (int c = 0; c < CT - 1; ++c) dot += pr[o][c + 1] * g[o][c];
          float out[CT];
          out[0] = pr[o][0] * (0.f - dot) + gce * (pr[o][0] - (lab[o] == 0 ? 1.f : 0.f));
#pragma unroll
          for (int c = 1; c < CT; ++c) out[c] = pr[o][c] * (g[o][c - 1] - dot) + gce * (pr[o][c] - (c == lab[o] ? 1.f : 0.f));
          store_vox<CT>(dlogits, v0 + (int64_t)o * W, out);
        }
      }
      __syncthreads();        // the next plane restages the LDS tile
    }
#undef CFUN_STAGE_LOAD
  }
}

inline unsigned fused_grid(int64_t tiles, int cap) {
  if (tiles > cap) tiles = cap;
  return (unsigned)(tiles < 1 ? 1 : tiles);
}

// z planes per segment: enough segments to fill the chip a few times over, long enough to amortise the ring's 2 (4 with U)
// extra planes
inline int pick_zs(int planes, int64_t tiles_per_plane_seg) {
  static const int knob = [] { const char* e = getenv("CFUN_FUSED_ZS"); return e ? atoi(e) : 0; }();      // tuning knob
  if (knob > 0) return knob < planes ? knob : planes;
  int zs = 32;
  while (zs > 8 && tiles_per_plane_seg * ((planes + zs - 1) / zs) < 1536) zs /= 2;
  return zs < planes ? zs : (planes > 0 ? planes : 1);
}

template <int CT>
size_t fwd_lds() { return (size_t)2 * (CT - 1) * kPlane * sizeof(float) + 2 * kPlane; }
template <int CT>
size_t bwd_lds() { return (size_t)2 * (CT - 1) * kUSP * sizeof(float); }

}  // namespace

extern "C" {

size_t cfun_mask_fused_workspace_bytes(void) { return 2 * kMaxBlocksF * sizeof(double); }

int cfun_mask_fused_supported(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C) {
  return (C == 8 || C == 3) && n > 0 && D >= 3 && H >= 3 && W >= 3;
}

size_t cfun_mask_fused_u_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C) {
  if (!cfun_mask_fused_supported(n, D, H, W, C)) return 256;
  return cfun_align_up((size_t)n * D * (H - 2) * (W - 2) * 2 * (C - 1) * sizeof(float), 256);
}

int cfun_mask_fused_fwd(const float* logits, const uint8_t* labels, float* probs, float* losses, float* u, int32_t n,
                        int32_t D, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (!cfun_mask_fused_supported(n, D, H, W, C) || !logits || !labels || !probs || !losses) return CFUN_EINVAL;
  if (ws_bytes < cfun_mask_fused_workspace_bytes()) return CFUN_EWORKSPACE;
  if (C % 4 == 0 && (!cfun_aligned16(logits) || !cfun_aligned16(probs))) return CFUN_EINVAL;
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int64_t per_seg = (int64_t)n * ((Ho + kTY - 1) / kTY) * ((Wo + kTX - 1) / kTX);
  const int ZS = pick_zs(D, per_seg);
  const int64_t tiles = per_seg * ((D + ZS - 1) / ZS);
  const unsigned blocks = fused_grid(tiles, kMaxBlocksF);
  // (VPT = 1 -- 512 threads, one output per thread, 128 VGPRs -- measured 1.33 - 1.43 ms against 1.09 for the pair form on the same
  // box: the pair shares its row sums and the pass is VALU-bound; profiles/round6_losses_vpt.log)
#define LAUNCH(CT, WU) hipLaunchKernelGGL((k_mask_fused_fwd<CT, WU, 2>), dim3(blocks), dim3(kFB), fwd_lds<CT>(), cfun_st(stream), \
                                          logits, labels, probs, (double*)ws, u, n, D, H, W, ZS)
  if (C == 8) { if (u) LAUNCH(8, true); else LAUNCH(8, false); }
  else { if (u) LAUNCH(3, true); else LAUNCH(3, false); }
#undef LAUNCH
  const double nvox = (double)n * D * H * W, per = (double)Do * Ho * Wo;
  hipLaunchKernelGGL(k_finalize_sum2, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks, 1.0 / nvox,
                     1.0 / (per * (double)n), losses);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_mask_fused_bwd(const float* u, const float* probs, const uint8_t* labels, const float* g2, float* dlogits, int32_t n,
                        int32_t D, int32_t H, int32_t W, int32_t C, cfun_stream_t stream) {
  if (!cfun_mask_fused_supported(n, D, H, W, C) || !u || !probs || !labels || !g2 || !dlogits) return CFUN_EINVAL;
  if (C % 4 == 0 && (!cfun_aligned16(probs) || !cfun_aligned16(dlogits))) return CFUN_EINVAL;
  const int64_t tiles = (int64_t)n * ((D + kBZ - 1) / kBZ) * ((H + kTY - 1) / kTY) * ((W + kTX - 1) / kTX);
  const unsigned blocks = fused_grid(tiles, 1 << 20);
  // one voxel per thread (512 threads, 115 VGPRs) is the default: HBM-bound, more waves in flight win 2 - 4 % over the pair form
  // (CFUN_FUSED_BWD_VPT=2), profiles/round6_losses_vpt.log
  static const int vpt = [] { const char* e = getenv("CFUN_FUSED_BWD_VPT"); return (e && e[0] == '2') ? 2 : 1; }();
  if (C == 8 && vpt == 1) hipLaunchKernelGGL((k_mask_fused_bwd<8, 1>), dim3(blocks), dim3(2 * kFB), bwd_lds<8>(), cfun_st(stream), u, probs, labels, g2, dlogits, n, D, H, W);
  else if (C == 8) hipLaunchKernelGGL((k_mask_fused_bwd<8, 2>), dim3(blocks), dim3(kFB), bwd_lds<8>(), cfun_st(stream), u, probs, labels, g2, dlogits, n, D, H, W);
  else hipLaunchKernelGGL((k_mask_fused_bwd<3, 2>), dim3(blocks), dim3(kFB), bwd_lds<3>(), cfun_st(stream), u, probs, labels, g2, dlogits, n, D, H, W);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // extern "C"
