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
  __syncthreads();   // red[] is reused by a back-to-back call (k_ce_w_fwd sums numerator and denominator)
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
// Both are separable, so every kernel MARCHES along z: a thread owns one (y,x) column segment, turns each input
// plane's 3x3 (y,x) neighbourhood into the two in-plane sums  dy = B(y)A(x)*f  and  sm = A(y)A(x)*f  (9 loads
// instead of 27 per voxel) and combines a ring of three planes:  c0 = A(z)*dy,  c1 = B(z)*sm.
constexpr int kZSeg = 16;   // output planes per thread
// floats per voxel record of the coefficient field dc: (dc0, dc1) per foreground class.  (Padding the 56-byte records of 8
// classes to 64 bytes -- every access 16-byte aligned -- was measured: the gather reads 14 % more and got 21 % slower.)
constexpr int dc_stride(int ct) { return 2 * (ct - 1); }

// The TARGET side of the stencil is integer arithmetic on one-hot labels: per input plane and class, R_j = sum_i A[i] *
// [label(y+j, x+i) == c] (<= 4) for the three rows j; dy = R_0 - R_2 and sm = R_0 + 2 R_1 + R_2.  All classes are carried
// in PACKED 8-bit fields (two 32-bit words: classes 0-3 and 4-7): one shift-and-add per neighbour instead of a
// compare / two selects / two adds per neighbour AND class (which was 60 % of the kernel's instructions -- the march is
// bound by VALU issue, not by HBM), biased so that the signed combinations stay non-negative per field:
//   dyb = dy + 4  in [0, 8],   sm in [0, 16],   t0 + 16 = dyb_a + 2 dyb_b + dyb_c in [0, 32],   t1 + 16 = sm_a + 16 - sm_c.
struct TargetPlane {
  uint32_t dyb[2], sm[2];
};

template <int CT>
struct PlaneSums {   // classes 1..CT-1 are stored at index c-1
  float dy[CT - 1], sm[CT - 1];
};

// MODE 0: accumulate sum (|grad p| - |grad t|)^2 ; MODE 1: write dc[o][c-1][0..1] = dL/d(c0), dL/d(c1) scaled by gscale;
// MODE 2: both in one pass -- the loss, and dc for an upstream gradient of 1 (the training forward saves it, so the
// backward needs no second march over the probabilities)
// ---- two y-outputs per thread (round 3).  The march is bound by the NUMBER of load instructions per voxel-plane (the class
// split that raised occupancy instead made it slower), so a thread now owns the column PAIR (y, y+1):
// the x-direction sums A(x)*f of the FOUR rows y .. y+3 (12 neighbour loads) serve both outputs -- out 0 combines rows
// 0,1,2, out 1 rows 1,2,3 -- instead of 2 x 9 loads, and the in-plane arithmetic halves with them (one FMA per neighbour
// and class for the row sum, then dy = r0 - r2, sm = r0 + 2 r1 + r2 per output).
template <int CT>
struct RowSums {
  float a[4][CT - 1];     // A(x)-weighted sums of the probabilities of rows y .. y+3, classes 1 .. CT-1
  uint32_t t[4][2];       // the same for the one-hot targets, packed 8-bit fields (classes 0-3 | 4-7)
};

template <int CT>
__device__ __forceinline__ void row_sums(const float* __restrict__ probs, const uint8_t* __restrict__ labels, int64_t nbase,
                                         int z, int y, int x, int H, int W, RowSums<CT>& S) {
  const float A[3] = {1.f, 2.f, 1.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int yy = y + j < H ? y + j : H - 1;      // (row 3 past the volume: only when output 1 is not stored)
#pragma unroll
    for (int c = 0; c < CT - 1; ++c) S.a[j][c] = 0.f;
    S.t[j][0] = 0u; S.t[j][1] = 0u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int64_t vi = nbase + ((int64_t)z * H + yy) * W + (x + i);
      const float* pp = probs + vi * CT;
      const unsigned lab = labels[vi];
      const uint32_t v = lab < (unsigned)CT ? (i == 1 ? 2u : 1u) << ((lab & 3u) * 8u) : 0u;
      S.t[j][0] += (lab & 4u) ? 0u : v;
      S.t[j][1] += (lab & 4u) ? v : 0u;
#pragma unroll
      for (int c = 1; c < CT; ++c) S.a[j][c - 1] += A[i] * pp[c];
    }
  }
}

template <int CT>
__device__ __forceinline__ void pair_planes(const RowSums<CT>& S, PlaneSums<CT> (&P)[2], TargetPlane (&T)[2]) {
#pragma unroll
  for (int o = 0; o < 2; ++o) {
#pragma unroll
    for (int c = 0; c < CT - 1; ++c) {
      P[o].dy[c] = S.a[o][c] - S.a[o + 2][c];
      P[o].sm[c] = (S.a[o][c] + S.a[o + 2][c]) + 2.f * S.a[o + 1][c];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      T[o].dyb[h] = S.t[o][h] + 0x04040404u - S.t[o + 2][h];
      T[o].sm[h] = S.t[o][h] + 2u * S.t[o + 1][h] + S.t[o + 2][h];
    }
  }
}

template <int CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_edge_march2(const float* __restrict__ probs, const uint8_t* __restrict__ labels, const float* __restrict__ gscale,
              double* __restrict__ partial, float* __restrict__ dc, int n, int D, int H, int W) {
  static_assert(CT <= 8, "packed target sums: 8 classes");
  const int Do = D - 2, Ho = H - 2, Wo = W - 2, Hp = (Ho + 1) / 2;
  const int nseg = (Do + kZSeg - 1) / kZSeg;
  // A block owns a 2-D patch of columns -- 16 x-columns x 16 y-pairs (32 rows) of one (sample, z segment) -- so that the
  // rows its threads share stay in ITS L1 / its XCD's L2: with 256 consecutive (x, y) columns per block (round 2) a block
  // covered 1.35 rows and read 3.35, and its row neighbours ran on other XCDs (blocks are dealt round-robin), i.e. every
  // plane was fetched ~2.5 times (FETCH_SIZE: 2.6 GB for 0.93 GB of probabilities); the patch reads 34 x 18 for 32 x 16.
  const int tX = (Wo + 15) / 16, tY = (Hp + 15) / 16;
  const int64_t tiles = (int64_t)n * nseg * tY * tX;
  const float gs = MODE == 0 ? 0.f : (MODE == 1 ? gscale[0] : 1.f) * 2.f / ((float)Do * (float)Ho * (float)Wo * (float)n);
  double acc = 0.0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    int64_t t = tile;
    const int x = (int)(t % tX) * 16 + (threadIdx.x & 15); t /= tX;
    const int y = 2 * ((int)(t % tY) * 16 + (threadIdx.x >> 4)); t /= tY;
    const int seg = (int)(t % nseg);
    const int64_t r = t / nseg;
    if (x >= Wo || y >= Ho) continue;
    const int64_t nbase = r * D * H * W;
    const int z0 = seg * kZSeg;
    const int z1 = z0 + kZSeg < Do ? z0 + kZSeg : Do;
    const bool two = y + 1 < Ho;
    PlaneSums<CT> P[3][2];          // [ring plane][output]
    TargetPlane T[3][2];
    {
      RowSums<CT> S;
      row_sums<CT>(probs, labels, nbase, z0, y, x, H, W, S);
      pair_planes<CT>(S, P[0], T[0]);
      row_sums<CT>(probs, labels, nbase, z0 + 1, y, x, H, W, S);
      pair_planes<CT>(S, P[1], T[1]);
    }
    float accf = 0.f;
    for (int zo = z0; zo < z1; ++zo) {
      {
        RowSums<CT> S;
        row_sums<CT>(probs, labels, nbase, zo + 2, y, x, H, W, S);
        pair_planes<CT>(S, P[2], T[2]);
      }
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const bool live = o == 0 || two;
        float* op = MODE != 0 ? dc + ((((r * Do + zo) * Ho + (y + o)) * Wo + x)) * dc_stride(CT) : nullptr;
        float ov[dc_stride(CT)];
        uint32_t t0b[2], t1b[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          t0b[h] = T[0][o].dyb[h] + 2u * T[1][o].dyb[h] + T[2][o].dyb[h];       // t0 + 16 per 8-bit field
          t1b[h] = T[0][o].sm[h] + 0x10101010u - T[2][o].sm[h];                 // t1 + 16
        }
#pragma unroll
        for (int c = 0; c < CT - 1; ++c) {
          const float p0 = P[0][o].dy[c] + 2.f * P[1][o].dy[c] + P[2][o].dy[c], p1 = P[0][o].sm[c] - P[2][o].sm[c];
          const int cls = c + 1;
          const float t0 = (float)((int)((t0b[cls >> 2] >> ((cls & 3) * 8)) & 0xffu) - 16);
          const float t1 = (float)((int)((t1b[cls >> 2] >> ((cls & 3) * 8)) & 0xffu) - 16);
          // |grad p| and 1 / |grad p| from ONE quarter-rate instruction (v_rsq_f32) instead of v_sqrt + v_rcp: the march is
          // bound by VALU issue (profiles/round3_pmc_loss_kernels.txt), and the transcendentals are its costliest instructions
          const float sp = p0 * p0 + p1 * p1 + p0 * p0;                    // channel 0 twice (model.py:969-972)
          const float ip = cfun_fast_rsq(sp);                              // inf at 0
          const float pm = sp > 0.f ? sp * ip : 0.f;
          const float tm = cfun_fast_sqrt(t0 * t0 + t1 * t1 + t0 * t0);
          if (MODE != 1) {
            const float d = pm - tm;
            accf += live ? d * d : 0.f;
          }
          if (MODE != 0) {
            const float k = gs * (pm - tm) * ip;      // 0 * inf -> NaN exactly where torch's sqrt backward gives 0/0 (App. A-13)
            ov[2 * c] = k * 2.f * p0; ov[2 * c + 1] = k * p1;
          }
        }
        if (MODE != 0 && live) {
#pragma unroll
          for (int c = 0; c < CT - 1; ++c) reinterpret_cast<float2*>(op)[c] = make_float2(ov[2 * c], ov[2 * c + 1]);
        }
      }
#pragma unroll
      for (int o = 0; o < 2; ++o) { P[0][o] = P[1][o]; P[1][o] = P[2][o]; T[0][o] = T[1][o]; T[1][o] = T[2][o]; }
    }
    acc += (double)accf;
  }
  if (MODE != 1) {
    const double s = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  }
}

// transposed stencil, marching along z over the OUTPUT planes:  Q0[zo](y,x) = sum_{j,i} B[j]A[i] dc0[zo, y-j, x-i],
// Q1[zo](y,x) = sum A[j]A[i] dc1[zo, y-j, x-i];  dprob[z] = sum_dz A[dz]*Q0[z-dz] + B[dz]*Q1[z-dz]
template <int CT>
__device__ __forceinline__ void plane_adj(const float* __restrict__ dc, int64_t r, int zo, int y, int x, int Do, int Ho,
                                          int Wo, float (&q0)[CT - 1], float (&q1)[CT - 1]) {
  const float A[3] = {1.f, 2.f, 1.f}, B[3] = {1.f, 0.f, -1.f};
#pragma unroll
  for (int c = 0; c < CT - 1; ++c) { q0[c] = 0.f; q1[c] = 0.f; }
  if (zo < 0 || zo >= Do) return;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int oy = y - j;
    if (oy < 0 || oy >= Ho) continue;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ox = x - i;
      if (ox < 0 || ox >= Wo) continue;
      // the voxel's 2*(CT-1) coefficients as (dc0, dc1) pairs: 8-byte loads (half the load instructions of the scalar
      // walk; the kernel is bound by its 9 neighbour gathers per plane, not by HBM)
      const float2* d = reinterpret_cast<const float2*>(dc + (((r * Do + zo) * Ho + oy) * Wo + ox) * dc_stride(CT));
      const float wd = B[j] * A[i], ws = A[j] * A[i];
#pragma unroll
      for (int c = 0; c < CT - 1; ++c) {
        const float2 v = d[c];
        q0[c] += wd * v.x; q1[c] += ws * v.y;
      }
    }
  }
}

// FUSE = false: dprobs of the edge loss.  FUSE = true: the whole mask-loss backward in this pass --
//   dlogits = softmax_bwd(probs, dprobs_edge) + gce[0]/nvox * (probs - onehot(label))
// (what k_softmax_bwd + k_ce_bwd + an add would compute), without materialising dprobs.
template <int CT, bool FUSE>
__global__ void __launch_bounds__(kBlock)
k_edge_bwd_gather(const float* __restrict__ dc, float* __restrict__ dprobs, int n, int D, int H, int W,
                  const float* __restrict__ probs, const uint8_t* __restrict__ labels, const float* __restrict__ gce,
                  const float* __restrict__ gedge) {
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int nseg = (D + kZSeg - 1) / kZSeg;
  // (2-D patches of 16 x 16 columns per block, as in k_edge_march2, were measured here too: 1.28 -> 1.33 ms.  The counters
  // -- profiles/round3_pmc_loss_kernels.txt -- show this kernel waiting on memory 79 % of its wave-cycles with FETCH_SIZE at
  // the algorithmic bytes: latency at 3 waves per SIMD, not re-fetching, so the coalesced 1-D walk stays.)
  const int64_t total = (int64_t)n * nseg * H * W;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;
    const int seg = (int)(t % nseg);
    const int64_t r = t / nseg;
    const int z0 = seg * kZSeg;
    const int z1 = z0 + kZSeg < D ? z0 + kZSeg : D;
    // input plane z receives from output planes z, z-1, z-2 (dz = 0, 1, 2)
    float q0[3][CT - 1], q1[3][CT - 1];   // ring: [0] = zo = z-2, [1] = z-1, [2] = z
    plane_adj<CT>(dc, r, z0 - 2, y, x, Do, Ho, Wo, q0[0], q1[0]);
    plane_adj<CT>(dc, r, z0 - 1, y, x, Do, Ho, Wo, q0[1], q1[1]);
    for (int z = z0; z < z1; ++z) {
      plane_adj<CT>(dc, r, z, y, x, Do, Ho, Wo, q0[2], q1[2]);
      const int64_t vox = ((r * D + z) * H + y) * W + x;
      float* o = dprobs + vox * CT;
      float g[CT];
      g[0] = 0.f;
      const float ge = gedge ? gedge[0] : 1.f;     // dc saved by the forward is for an upstream gradient of 1
#pragma unroll
      for (int c = 0; c < CT - 1; ++c)   // A[dz]: dz=0 -> zo=z, dz=1 -> z-1, dz=2 -> z-2 ; B[dz] = (1,0,-1)
        g[c + 1] = ge * ((q0[2][c] + 2.f * q0[1][c] + q0[0][c]) + (q1[2][c] - q1[0][c]));
      if (FUSE) {
        const float gs = gce[0] / (float)((int64_t)n * D * H * W);
        const int lab = labels[vox];
        float pr[CT], dot = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) { pr[c] = probs[vox * CT + c]; dot += pr[c] * g[c]; }
#pragma unroll
        for (int c = 0; c < CT; ++c) o[c] = pr[c] * (g[c] - dot) + gs * (pr[c] - (c == lab ? 1.f : 0.f));
      } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) o[c] = g[c];
      }
#pragma unroll
      for (int c = 0; c < CT - 1; ++c) { q0[0][c] = q0[1][c]; q0[1][c] = q0[2][c]; q1[0][c] = q1[1][c]; q1[1][c] = q1[2][c]; }
    }
  }
}

// ------------------------------------------------------------------ LiTS fork: weighted CE and raw Sobel MSE
// nn.CrossEntropyLoss(weight = w) (LiTS_2017/model.py:926, w = [1, 1, 100]): sum_i w[y_i] * (-log p_i[y_i]) / sum_i w[y_i].
// partial[b] = numerator, partial[kMaxBlocks + b] = denominator of block b.
template <int CT>
__global__ void __launch_bounds__(kBlock)
k_ce_w_fwd(const float* __restrict__ logits, const uint8_t* __restrict__ labels, const float* __restrict__ weights,
           double* __restrict__ partial, int64_t nvox, int Crt) {
  const int C = CT > 0 ? CT : Crt;
  double num = 0.0, den = 0.0;
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
    const float w = weights[lab];
    num += (double)(w * ((m + logf(s)) - xl));
    den += (double)w;
  }
  const double sn = block_sum(num);
  const double sd = block_sum(den);
  if (threadIdx.x == 0) { partial[blockIdx.x] = sn; partial[kMaxBlocks + blockIdx.x] = sd; }
}

__global__ void k_finalize_ratio(const double* __restrict__ partial, int blocks, float* __restrict__ loss,
                                 float* __restrict__ wsum) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 64) { a += partial[i]; b += partial[kMaxBlocks + i]; }
  a = cfun_wave_sum_d(a);
  b = cfun_wave_sum_d(b);
  if (threadIdx.x == 0) { loss[0] = (float)(a / b); wsum[0] = (float)b; }
}

// dlogits = g * w[y] / sum_w * (softmax - onehot)
template <int CT>
__global__ void __launch_bounds__(kBlock)
k_ce_w_bwd(const float* __restrict__ logits, const uint8_t* __restrict__ labels, const float* __restrict__ weights,
           const float* __restrict__ gscale, const float* __restrict__ wsum, float* __restrict__ dlogits, int64_t nvox,
           int Crt) {
  const int C = CT > 0 ? CT : Crt;
  const float gs = gscale[0] / wsum[0];
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
    const float k = gs * weights[lab], inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < (CT > 0 ? CT : kMaxC); ++c)
      if (c < C) dlogits[v * C + c] = k * (x[c] * inv - (c == lab ? 1.f : 0.f));
  }
}

// Edge loss of the fork (LiTS_2017/model.py:936-979): MSE between the RAW three Sobel responses of the predicted and
// the target mask (no magnitude), per positive RoI and foreground class, summed and divided by the RoI count:
//   loss = sum_{roi, class j >= 1, k < 3, o} (p_k(o) - t_k(o))^2 / (3 * Do*Ho*Wo * n)
// kernel k = 0: derivative along y (A[dz] B[dy] A[dx]), 1: along z, 2: along x; A = (1,2,1), B = (1,0,-1); valid conv.
// One thread per output voxel and class pair; the difference field dc[o][(j-1)*3 + k] is kept for the backward.
template <int CT>
__global__ void __launch_bounds__(kBlock)
k_edge_raw_fwd(const float* __restrict__ probs, const uint8_t* __restrict__ labels, double* __restrict__ partial,
               float* __restrict__ dc, int n, int D, int H, int W) {
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int64_t total = (int64_t)n * Do * Ho * Wo;
  const float A[3] = {1.f, 2.f, 1.f}, B[3] = {1.f, 0.f, -1.f};
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho); t /= Ho;
    const int zo = (int)(t % Do);
    const int64_t r = t / Do;
    float d[CT - 1][3];
#pragma unroll
    for (int j = 0; j < CT - 1; ++j) d[j][0] = d[j][1] = d[j][2] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int64_t v = ((r * D + zo + a) * H + yo + b) * W + xo + c;
          const int lab = labels[v];
          const float k0 = A[a] * B[b] * A[c], k1 = B[a] * A[b] * A[c], k2 = A[a] * A[b] * B[c];
#pragma unroll
          for (int j = 0; j < CT - 1; ++j) {
            const float f = probs[v * CT + j + 1] - (lab == j + 1 ? 1.f : 0.f);     // conv is linear: conv(p) - conv(t)
            d[j][0] += k0 * f; d[j][1] += k1 * f; d[j][2] += k2 * f;
          }
        }
#pragma unroll
    for (int j = 0; j < CT - 1; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        acc += (double)(d[j][k] * d[j][k]);
        if (dc) dc[i * (3 * (CT - 1)) + j * 3 + k] = d[j][k];
      }
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// dprobs[v][j] = g * 2 / (3*Do*Ho*Wo*n) * sum_{a,b,c, k} K_k[a][b][c] * dc[v - (a,b,c)][(j-1)*3 + k];  dprobs[v][0] = 0
template <int CT>
__global__ void __launch_bounds__(kBlock)
k_edge_raw_bwd(const float* __restrict__ dc, const float* __restrict__ gscale, float* __restrict__ dprobs, int n, int D,
               int H, int W) {
  const int Do = D - 2, Ho = H - 2, Wo = W - 2;
  const int64_t total = (int64_t)n * D * H * W;
  const float A[3] = {1.f, 2.f, 1.f}, B[3] = {1.f, 0.f, -1.f};
  const float gs = gscale[0] * 2.f / (3.f * (float)Do * (float)Ho * (float)Wo * (float)n);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t t = i;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;
    const int z = (int)(t % D);
    const int64_t r = t / D;
    float g[CT - 1];
#pragma unroll
    for (int j = 0; j < CT - 1; ++j) g[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int zo = z - a;
      if (zo < 0 || zo >= Do) continue;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int yo = y - b;
        if (yo < 0 || yo >= Ho) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int xo = x - c;
          if (xo < 0 || xo >= Wo) continue;
          const float* d = dc + ((((r * Do + zo) * Ho + yo) * Wo + xo)) * (3 * (CT - 1));
          const float k0 = A[a] * B[b] * A[c], k1 = B[a] * A[b] * A[c], k2 = A[a] * A[b] * B[c];
#pragma unroll
          for (int j = 0; j < CT - 1; ++j) g[j] += k0 * d[j * 3] + k1 * d[j * 3 + 1] + k2 * d[j * 3 + 2];
        }
      }
    }
    dprobs[i * CT] = 0.f;
#pragma unroll
    for (int j = 0; j < CT - 1; ++j) dprobs[i * CT + j + 1] = gs * g[j];
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
  const int64_t cols = (int64_t)n * ((D - 2 + kZSeg - 1) / kZSeg) * ((H - 2 + 1) / 2) * (W - 2);      // y pairs (k_edge_march2)
  const unsigned blocks = vox_grid(cols);
  if (C == 8) hipLaunchKernelGGL((k_edge_march2<8, 0>), dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (const float*)nullptr, (double*)ws, (float*)nullptr, n, D, H, W);
  else hipLaunchKernelGGL((k_edge_march2<3, 0>), dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (const float*)nullptr, (double*)ws, (float*)nullptr, n, D, H, W);
  hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks,
                     1.0 / ((double)per * (double)n), loss);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

size_t cfun_edge_loss_bwd_workspace_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C) {
  if (n <= 0 || D < 3 || H < 3 || W < 3 || C < 2) return 256;
  return cfun_align_up((size_t)n * (D - 2) * (H - 2) * (W - 2) * dc_stride(C) * sizeof(float), 256);
}

int cfun_edge_loss_bwd(const float* probs, const uint8_t* labels, const float* gscale, float* dprobs, int32_t n,
                       int32_t D, int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (C != 8 && C != 3) return CFUN_EINVAL;
  const int64_t total = (int64_t)n * D * H * W;
  if (total <= 0) return CFUN_OK;
  if (D < 3 || H < 3 || W < 3) return (int)hipMemsetAsync(dprobs, 0, total * C * sizeof(float), cfun_st(stream));
  if (ws_bytes < cfun_edge_loss_bwd_workspace_bytes(n, D, H, W, C)) return CFUN_EWORKSPACE;
  const int64_t cols_o = (int64_t)n * ((D - 2 + kZSeg - 1) / kZSeg) * ((H - 2 + 1) / 2) * (W - 2);      // y pairs (k_edge_march2)
  const int64_t cols_i = (int64_t)n * ((D + kZSeg - 1) / kZSeg) * H * W;
  if (C == 8) {
    hipLaunchKernelGGL((k_edge_march2<8, 1>), dim3(vox_grid(cols_o)), dim3(kBlock), 0, cfun_st(stream), probs, labels, gscale, (double*)nullptr, (float*)ws, n, D, H, W);
    hipLaunchKernelGGL((k_edge_bwd_gather<8, false>), dim3(vox_grid(cols_i)), dim3(kBlock), 0, cfun_st(stream), (const float*)ws, dprobs, n, D, H, W, (const float*)nullptr, (const uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL((k_edge_march2<3, 1>), dim3(vox_grid(cols_o)), dim3(kBlock), 0, cfun_st(stream), probs, labels, gscale, (double*)nullptr, (float*)ws, n, D, H, W);
    hipLaunchKernelGGL((k_edge_bwd_gather<3, false>), dim3(vox_grid(cols_i)), dim3(kBlock), 0, cfun_st(stream), (const float*)ws, dprobs, n, D, H, W, (const float*)nullptr, (const uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
  }
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// training forward: the loss AND the unit-gradient coefficient field dc (cfun_edge_loss_bwd_workspace_bytes bytes)
int cfun_edge_loss_fwd_save(const float* probs, const uint8_t* labels, float* loss, float* dc, int32_t n, int32_t D,
                            int32_t H, int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (C != 8 && C != 3) return CFUN_EINVAL;
  if (n <= 0 || D < 3 || H < 3 || W < 3) return CFUN_EINVAL;
  if (ws_bytes < kMaxBlocks * sizeof(double)) return CFUN_EWORKSPACE;
  const int64_t per = (int64_t)(D - 2) * (H - 2) * (W - 2);
  const int64_t cols = (int64_t)n * ((D - 2 + kZSeg - 1) / kZSeg) * ((H - 2 + 1) / 2) * (W - 2);      // y pairs (k_edge_march2)
  const unsigned blocks = vox_grid(cols);
  if (C == 8) hipLaunchKernelGGL((k_edge_march2<8, 2>), dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (const float*)nullptr, (double*)ws, dc, n, D, H, W);
  else hipLaunchKernelGGL((k_edge_march2<3, 2>), dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (const float*)nullptr, (double*)ws, dc, n, D, H, W);
  hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks,
                     1.0 / ((double)per * (double)n), loss);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// backward of both mask losses from the dc saved by cfun_edge_loss_fwd_save: ONE pass (no second march)
int cfun_mask_losses_bwd_saved(const float* probs, const uint8_t* labels, const float* g_ce, const float* g_edge,
                               const float* dc, float* dlogits, int32_t n, int32_t D, int32_t H, int32_t W, int32_t C,
                               cfun_stream_t stream) {
  if (C != 8 && C != 3) return CFUN_EINVAL;
  const int64_t total = (int64_t)n * D * H * W;
  if (total <= 0) return CFUN_OK;
  if (D < 3 || H < 3 || W < 3 || !dc || !g_edge) return CFUN_EINVAL;
  const int64_t cols_i = (int64_t)n * ((D + kZSeg - 1) / kZSeg) * H * W;
  if (C == 8) hipLaunchKernelGGL((k_edge_bwd_gather<8, true>), dim3(vox_grid(cols_i)), dim3(kBlock), 0, cfun_st(stream), dc, dlogits, n, D, H, W, probs, labels, g_ce, g_edge);
  else hipLaunchKernelGGL((k_edge_bwd_gather<3, true>), dim3(vox_grid(cols_i)), dim3(kBlock), 0, cfun_st(stream), dc, dlogits, n, D, H, W, probs, labels, g_ce, g_edge);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_mask_losses_bwd(const float* probs, const uint8_t* labels, const float* g_ce, const float* g_edge,
                         float* dlogits, int32_t n, int32_t D, int32_t H, int32_t W, int32_t C, void* ws,
                         size_t ws_bytes, cfun_stream_t stream) {
  if (C != 8 && C != 3) return CFUN_EINVAL;
  const int64_t total = (int64_t)n * D * H * W;
  if (total <= 0) return CFUN_OK;
  if (D < 3 || H < 3 || W < 3) return CFUN_EINVAL;   // no edge term: use cfun_softmax_ce_bwd
  if (ws_bytes < cfun_edge_loss_bwd_workspace_bytes(n, D, H, W, C)) return CFUN_EWORKSPACE;
  const int64_t cols_o = (int64_t)n * ((D - 2 + kZSeg - 1) / kZSeg) * ((H - 2 + 1) / 2) * (W - 2);      // y pairs (k_edge_march2)
  const int64_t cols_i = (int64_t)n * ((D + kZSeg - 1) / kZSeg) * H * W;
  if (C == 8) {
    hipLaunchKernelGGL((k_edge_march2<8, 1>), dim3(vox_grid(cols_o)), dim3(kBlock), 0, cfun_st(stream), probs, labels, g_edge, (double*)nullptr, (float*)ws, n, D, H, W);
    hipLaunchKernelGGL((k_edge_bwd_gather<8, true>), dim3(vox_grid(cols_i)), dim3(kBlock), 0, cfun_st(stream), (const float*)ws, dlogits, n, D, H, W, probs, labels, g_ce, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL((k_edge_march2<3, 1>), dim3(vox_grid(cols_o)), dim3(kBlock), 0, cfun_st(stream), probs, labels, g_edge, (double*)nullptr, (float*)ws, n, D, H, W);
    hipLaunchKernelGGL((k_edge_bwd_gather<3, true>), dim3(vox_grid(cols_i)), dim3(kBlock), 0, cfun_st(stream), (const float*)ws, dlogits, n, D, H, W, probs, labels, g_ce, (const float*)nullptr);
  }
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// ---- LiTS fork losses (LiTS_2017/model.py:907-979)
size_t cfun_ce_weighted_workspace_bytes(void) { return 2 * kMaxBlocks * sizeof(double); }

int cfun_softmax_ce_weighted_fwd(const float* logits, const uint8_t* labels, const float* weights, float* loss,
                                 float* wsum, int64_t nvox, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (C <= 0 || C > kMaxC || !weights || !wsum) return CFUN_EINVAL;
  if (nvox <= 0) return CFUN_EINVAL;                    // 0 / 0: the reference returns a constant 0 before getting here
  if (ws_bytes < cfun_ce_weighted_workspace_bytes()) return CFUN_EWORKSPACE;
  const unsigned blocks = vox_grid(nvox);
#define CALL(CT) hipLaunchKernelGGL(k_ce_w_fwd<CT>, dim3(blocks), dim3(kBlock), 0, cfun_st(stream), logits, labels, weights, (double*)ws, nvox, C);
  DISPATCH_C(C, CALL)
#undef CALL
  hipLaunchKernelGGL(k_finalize_ratio, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks, loss, wsum);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_softmax_ce_weighted_bwd(const float* logits, const uint8_t* labels, const float* weights, const float* gscale,
                                 const float* wsum, float* dlogits, int64_t nvox, int32_t C, cfun_stream_t stream) {
  if (nvox <= 0) return CFUN_OK;
  if (C <= 0 || C > kMaxC || !weights || !wsum) return CFUN_EINVAL;
#define CALL(CT) hipLaunchKernelGGL(k_ce_w_bwd<CT>, dim3(vox_grid(nvox)), dim3(kBlock), 0, cfun_st(stream), logits, labels, weights, gscale, wsum, dlogits, nvox, C);
  DISPATCH_C(C, CALL)
#undef CALL
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

size_t cfun_edge_raw_dc_bytes(int32_t n, int32_t D, int32_t H, int32_t W, int32_t C) {
  if (n <= 0 || D < 3 || H < 3 || W < 3 || C < 2) return 256;
  return cfun_align_up((size_t)n * (D - 2) * (H - 2) * (W - 2) * 3 * (C - 1) * sizeof(float), 256);
}

// dc (cfun_edge_raw_dc_bytes; may be NULL for a forward-only call); ws: cfun_loss_workspace_bytes
int cfun_edge_raw_fwd(const float* probs, const uint8_t* labels, float* loss, float* dc, int32_t n, int32_t D, int32_t H,
                      int32_t W, int32_t C, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (C != 3 && C != 2) return CFUN_EINVAL;
  if (n <= 0 || D < 3 || H < 3 || W < 3) return (int)hipMemsetAsync(loss, 0, sizeof(float), cfun_st(stream));
  if (ws_bytes < kMaxBlocks * sizeof(double)) return CFUN_EWORKSPACE;
  const int64_t per = (int64_t)(D - 2) * (H - 2) * (W - 2);
  const unsigned blocks = vox_grid((int64_t)n * per);
  if (C == 3) hipLaunchKernelGGL(k_edge_raw_fwd<3>, dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (double*)ws, dc, n, D, H, W);
  else hipLaunchKernelGGL(k_edge_raw_fwd<2>, dim3(blocks), dim3(kBlock), 0, cfun_st(stream), probs, labels, (double*)ws, dc, n, D, H, W);
  hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(64), 0, cfun_st(stream), (const double*)ws, (int)blocks,
                     1.0 / (3.0 * (double)per * (double)n), loss);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_edge_raw_bwd(const float* dc, const float* gscale, float* dprobs, int32_t n, int32_t D, int32_t H, int32_t W,
                      int32_t C, cfun_stream_t stream) {
  if (C != 3 && C != 2) return CFUN_EINVAL;
  const int64_t total = (int64_t)n * D * H * W;
  if (total <= 0) return CFUN_OK;
  if (D < 3 || H < 3 || W < 3) return (int)hipMemsetAsync(dprobs, 0, total * C * sizeof(float), cfun_st(stream));
  if (C == 3) hipLaunchKernelGGL(k_edge_raw_bwd<3>, dim3(vox_grid(total)), dim3(kBlock), 0, cfun_st(stream), dc, gscale, dprobs, n, D, H, W);
  else hipLaunchKernelGGL(k_edge_raw_bwd<2>, dim3(vox_grid(total)), dim3(kBlock), 0, cfun_st(stream), dc, gscale, dprobs, n, D, H, W);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}


}  // extern "C"
