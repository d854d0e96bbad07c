// 3-D RoIAlign (crop + trilinear align_corners resize) and greedy 3-D NMS.
// Compiled with -ffp-contract=off: the integer crop bounds and the NMS keep list are bit-exact contracts
// (SURVEY.md App. A-9/A-10), so the fp32 operation order of torch/numpy is reproduced without FMA contraction.
#include <stdlib.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------ RoIAlign
// python slice semantics of fm[:, lo:hi] on an axis of size S (model.py:282)
__device__ __forceinline__ void py_slice(int lo, int hi, int S, int* olo, int* ohi) {
  if (lo < 0) { lo += S; if (lo < 0) lo = 0; }
  if (lo > S) lo = S;
  if (hi < 0) { hi += S; if (hi < 0) hi = 0; }
  if (hi > S) hi = S;
  if (hi < lo) hi = lo;
  *olo = lo;
  *ohi = hi;
}

__global__ void k_roi_bounds(const float* __restrict__ boxes, int32_t* __restrict__ bounds, int R, int D, int H, int W) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float sc[3] = {(float)D, (float)H, (float)W};
  const int S[3] = {D, H, W};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // utils.py:172-173 torch.mul in fp32; model.py:272-278 floor / ceil / .long()
    const float lo = floorf(__fmul_rn(boxes[r * 6 + a], sc[a]));
    const float hi = ceilf(__fmul_rn(boxes[r * 6 + 3 + a], sc[a]));
    int l, h;
    py_slice((int)lo, (int)hi, S[a], &l, &h);
    bounds[r * 6 + a] = l;
    bounds[r * 6 + 3 + a] = h;
  }
}

struct Lerp {
  int i0, i1;
  float w0, w1;
};
// torch area_pixel_compute_source_index, align_corners=True
__device__ __forceinline__ Lerp make_lerp(int o, int in_size, int out_size) {
  Lerp l;
  const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  const float src = scale * (float)o;
  l.i0 = (int)src;
  if (l.i0 > in_size - 1) l.i0 = in_size - 1;
  l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
  l.w1 = src - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}

__global__ void __launch_bounds__(256)
k_roi_align_fwd(const float* __restrict__ fm, const int32_t* __restrict__ bounds, float* __restrict__ out,
                int64_t total, int D, int H, int W, int C, int pd, int ph, int pw, int sz0, int sdl) {
  // fm holds depth planes [sz0, sz0 + sdl) of the [D,H,W,C] map (a depth slab of a sharded volume; the whole map: 0, D);
  // planes outside read as zeros, so the slabs' partial results add up to the RoIAlign of the whole map
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int ox = (int)(t % pw); t /= pw;
    const int oy = (int)(t % ph); t /= ph;
    const int oz = (int)(t % pd);
    const int r = (int)(t / pd);
    const int32_t* b = bounds + r * 6;
    const int nz = b[3] - b[0], ny = b[4] - b[1], nx = b[5] - b[2];
    if (nz <= 0 || ny <= 0 || nx <= 0) { out[i] = 0.f; continue; }  // failed crop -> zeros (model.py:284-287)
    const Lerp lz = make_lerp(oz, nz, pd), ly = make_lerp(oy, ny, ph), lx = make_lerp(ox, nx, pw);
    const int za = b[0] + lz.i0 - sz0, zb = b[0] + lz.i1 - sz0;        // slab-local planes
    const bool in_a = za >= 0 && za < sdl, in_b = zb >= 0 && zb < sdl;
    const int64_t z0 = (int64_t)(in_a ? za : 0) * H, z1 = (int64_t)(in_b ? zb : 0) * H;
    const int64_t y0 = b[1] + ly.i0, y1 = b[1] + ly.i1;
    const int64_t x0 = b[2] + lx.i0, x1 = b[2] + lx.i1;
    const float v000 = fm[((z0 + y0) * W + x0) * C + c], v001 = fm[((z0 + y0) * W + x1) * C + c];
    const float v010 = fm[((z0 + y1) * W + x0) * C + c], v011 = fm[((z0 + y1) * W + x1) * C + c];
    const float v100 = fm[((z1 + y0) * W + x0) * C + c], v101 = fm[((z1 + y0) * W + x1) * C + c];
    const float v110 = fm[((z1 + y1) * W + x0) * C + c], v111 = fm[((z1 + y1) * W + x1) * C + c];
    const float pa = ly.w0 * (lx.w0 * v000 + lx.w1 * v001) + ly.w1 * (lx.w0 * v010 + lx.w1 * v011);
    const float pb = ly.w0 * (lx.w0 * v100 + lx.w1 * v101) + ly.w1 * (lx.w0 * v110 + lx.w1 * v111);
    out[i] = lz.w0 * (in_a ? pa : 0.f) + lz.w1 * (in_b ? pb : 0.f);
  }
}

// Backward of RoIAlign as a GATHER (round 5): one thread per (slab voxel, channel) walks the RoIs in index order and, per
// RoI whose crop holds the voxel, the output samples whose trilinear footprint touches it -- the transpose of k_roi_align_fwd's
// weights, evaluated with the same make_lerp -- and writes the sum once.  No atomics: the summation order is fixed, so the
// gradients of the feature maps (and with them every FPN / RPN / classifier gradient) are run-to-run reproducible like the
// reference's CPU path; the round 1-4 kernel scattered with fp32 atomicAdd (1e-5 relative run-to-run noise).
__device__ __forceinline__ float lerp_weight_of(int o, int j, int n, int p) {       // d out[o] / d in[j] along one axis
  const Lerp l = make_lerp(o, n, p);
  return (l.i0 == j ? l.w0 : 0.f) + (l.i1 == j ? l.w1 : 0.f);      // (i1 == i0 on the last sample: both terms)
}
// [lo, hi): the outputs with a non-zero weight on input j -- contiguous, because the source index is monotone in o
__device__ __forceinline__ void lerp_range_of(int j, int n, int p, int* lo, int* hi) {
  int a = p, b = 0;
  for (int o = 0; o < p; ++o) {
    const Lerp l = make_lerp(o, n, p);
    if (l.i0 == j || l.i1 == j) { a = o < a ? o : a; b = o + 1; }
  }
  *lo = a; *hi = b;
}

__global__ void __launch_bounds__(256)
k_roi_align_bwd(const float* __restrict__ dout, const int32_t* __restrict__ bounds, float* __restrict__ dfm,
                int64_t total, int R, int D, int H, int W, int C, int pd, int ph, int pw, int sz0, int sdl) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int z = (int)(t / H) + sz0;
    float acc = 0.f;
    for (int r = 0; r < R; ++r) {
      const int32_t* b = bounds + r * 6;
      const int nz = b[3] - b[0], ny = b[4] - b[1], nx = b[5] - b[2];
      if (nz <= 0 || ny <= 0 || nx <= 0) continue;
      if (z < b[0] || z >= b[3] || y < b[1] || y >= b[4] || x < b[2] || x >= b[5]) continue;
      const int jz = z - b[0], jy = y - b[1], jx = x - b[2];
      int z0, z1, y0, y1, x0, x1;
      lerp_range_of(jz, nz, pd, &z0, &z1);
      lerp_range_of(jy, ny, ph, &y0, &y1);
      lerp_range_of(jx, nx, pw, &x0, &x1);
      const float* g = dout + (int64_t)r * pd * ph * pw * C + c;
      for (int oz = z0; oz < z1; ++oz) {
        const float wz = lerp_weight_of(oz, jz, nz, pd);
        for (int oy = y0; oy < y1; ++oy) {
          const float wy = lerp_weight_of(oy, jy, ny, ph);
          for (int ox = x0; ox < x1; ++ox) {
            const float wx = lerp_weight_of(ox, jx, nx, pw);
            acc += g[(((int64_t)oz * ph + oy) * pw + ox) * C] * wz * wy * wx;
          }
        }
      }
    }
    dfm[i] = acc;
  }
}

// ------------------------------------------------------------------ NMS
struct Box6 {
  float z1, y1, x1, z2, y2, x2;
};
__device__ __forceinline__ Box6 load_box(const float* b) { return Box6{b[0], b[1], b[2], b[3], b[4], b[5]}; }
__device__ __forceinline__ float box_volume(const Box6& b) {  // utils.py:136 (z2-z1)*(y2-y1)*(x2-x1)
  return __fmul_rn(__fmul_rn(__fsub_rn(b.z2, b.z1), __fsub_rn(b.y2, b.y1)), __fsub_rn(b.x2, b.x1));
}
// utils.py:60-69: inter = max(x2-x1,0)*max(y2-y1,0)*max(z2-z1,0); iou = inter / (vol_a + vol_b - inter + 1e-6)
__device__ __forceinline__ float box_iou(const Box6& a, float va, const Box6& b, float vb) {
  const float z1 = fmaxf(a.z1, b.z1), z2 = fminf(a.z2, b.z2);
  const float y1 = fmaxf(a.y1, b.y1), y2 = fminf(a.y2, b.y2);
  const float x1 = fmaxf(a.x1, b.x1), x2 = fminf(a.x2, b.x2);
  const float inter = __fmul_rn(__fmul_rn(fmaxf(__fsub_rn(x2, x1), 0.f), fmaxf(__fsub_rn(y2, y1), 0.f)),
                                fmaxf(__fsub_rn(z2, z1), 0.f));
  const float uni = __fsub_rn(__fadd_rn(va, vb), inter);
  return __fdiv_rn(inter, __fadd_rn(uni, 1e-6f));
}

// "a sorts before b": higher score first; equal scores: higher original index first (App. A-8)
__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {
  return sa > sb || (sa == sb && ia > ib);
}

// one workgroup: bitonic sort of (score, index) in LDS; writes order[n] and the gathered boxes/volumes
__global__ void __launch_bounds__(1024)
k_nms_sort(const float* __restrict__ boxes, const float* __restrict__ scores, int n, int npow2,
           int32_t* __restrict__ order, float* __restrict__ sboxes, float* __restrict__ svol) {
  CFUN_DYN_LDS(char, smem);
  float* ks = reinterpret_cast<float*>(smem);
  int* ki = reinterpret_cast<int*>(smem + (size_t)npow2 * sizeof(float));
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    ks[i] = i < n ? scores[i] : -INFINITY;
    ki[i] = i < n ? i : -1;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;  // ascending position = "before" order
          const float sa = ks[i], sb = ks[l];
          const int ia = ki[i], ib = ki[l];
          const bool swap = up ? before(sb, ib, sa, ia) : before(sa, ia, sb, ib);
          if (swap) { ks[i] = sb; ks[l] = sa; ki[i] = ib; ki[l] = ia; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int src = ki[i];
    order[i] = src;
    const Box6 b = load_box(boxes + (int64_t)src * 6);
    float* d = sboxes + (int64_t)i * 6;
    d[0] = b.z1; d[1] = b.y1; d[2] = b.x1; d[3] = b.z2; d[4] = b.y2; d[5] = b.x2;
    svol[i] = box_volume(b);
  }
}

// mask[i][w] bit j: sorted box 64w+j (> i) is suppressed by sorted box i
__global__ void __launch_bounds__(256)
k_nms_mask(const float* __restrict__ sboxes, const float* __restrict__ svol, int n, int W64, float thr,
           unsigned long long* __restrict__ mask) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n * W64) return;
  const int i = t / W64, w = t - i * W64;
  unsigned long long bits = 0ull;
  if (w * 64 + 63 > i) {
    const Box6 a = load_box(sboxes + (int64_t)i * 6);
    const float va = svol[i];
    for (int j = 0; j < 64; ++j) {
      const int k = w * 64 + j;
      if (k > i && k < n) {
        const Box6 b = load_box(sboxes + (int64_t)k * 6);
        if (box_iou(a, va, b, svol[k]) > thr) bits |= (1ull << j);
      }
    }
  }
  mask[t] = bits;
}

// single wave: lane w holds word w of the "removed" bitset; rows are prefetched 16 at a time
__global__ void __launch_bounds__(64)
k_nms_scan(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ order, int n, int W64,
           int max_num, int32_t* __restrict__ keep, int32_t* __restrict__ count) {
  const int lane = threadIdx.x;
  unsigned long long removed = 0ull;
  int cnt = 0;
  constexpr int CH = 16;
  unsigned long long cur[CH], nxt[CH];
#pragma unroll
  for (int r = 0; r < CH; ++r) cur[r] = (r < n && lane < W64) ? mask[(int64_t)r * W64 + lane] : 0ull;
  bool done = (max_num <= 0);
  for (int base = 0; base < n && !done; base += CH) {
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const int row = base + CH + r;
      nxt[r] = (row < n && lane < W64) ? mask[(int64_t)row * W64 + lane] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const int i = base + r;
      if (i < n && !done) {
        const unsigned long long word = __shfl(removed, i >> 6, 64);
        if (!((word >> (i & 63)) & 1ull)) {
          if (lane == 0) keep[cnt] = order[i];
          ++cnt;
          if (cnt >= max_num) done = true;   // utils.py:147-148: break right after the append
          removed |= cur[r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < CH; ++r) cur[r] = nxt[r];
  }
  if (lane == 0) count[0] = cnt;
}

// word `l` (wave-uniform) of a per-lane 64-bit value: two v_readlane_b32 (a few cycles) instead of two ds_bpermute round trips
__device__ __forceinline__ unsigned long long lane_word(unsigned long long v, int l) {
#ifdef CFUN_HIP_EMULATION
  return __shfl(v, l, 64);
#else
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
#endif
}

// Round 5: the same greedy scan with the suppression rows in LDS and the loop over KEPT boxes only.  The row-by-row kernel
// above walks all n <= 1000 candidates through one dependent shuffle each (0.38 ms per call); here 256 threads copy the n x W64
// words into LDS (128 KB at n = 1000), then one wave takes the candidates 64 at a time: the block's word of the removed set is
// wave-uniform, its zero bits are the survivors, and only a survivor costs an LDS row read + OR -- the time follows the number of
// boxes kept (<= max_num), not n.  Same visiting order, same stop rule (utils.py:147-148), so the keep list is bit-identical.
__global__ void __launch_bounds__(256)
k_nms_scan_lds(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ order, int n, int W64,
               int max_num, int32_t* __restrict__ keep, int32_t* __restrict__ count) {
  CFUN_DYN_LDS(unsigned long long, rows);      // [n][W64]
  for (int i = threadIdx.x; i < n * W64; i += 256) rows[i] = mask[i];
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned long long removed = 0ull;
  int cnt = 0;
  bool done = (max_num <= 0);
  for (int blk = 0; blk < W64 && !done; ++blk) {
    const int lim = n - blk * 64;
    const unsigned long long valid = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
    unsigned long long alive = ~lane_word(removed, blk) & valid;          // wave-uniform
    while (alive != 0ull && !done) {
      const int b = __builtin_ctzll(alive);
      const int i = blk * 64 + b;
      if (lane == 0) keep[cnt] = order[i];
      ++cnt;
      if (cnt >= max_num) done = true;        // utils.py:147-148: break right after the append
      if (lane < W64) removed |= rows[(int64_t)i * W64 + lane];
      alive = ~lane_word(removed, blk) & valid & ~((2ull << b) - 1ull);   // candidates after i in this block that survive
    }
  }
  if (lane == 0) count[0] = cnt;
}

inline int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// GT mask targets (model.py:481-493): crop the uint8 label volume with int()-truncated box bounds (computed by the
// caller in fp32, as the reference does) and nearest-resize it to the mask shape: output voxel o reads input voxel
// floor((o + 0.5) * in / out) of the crop (the order-0 resize of utils.py:318-339).  One-hot GT channels resized
// independently and arg-maxed give exactly this label volume.  Empty crop => zeros.
__global__ void __launch_bounds__(256)
k_mask_target_labels(const uint8_t* __restrict__ labels, const int32_t* __restrict__ bounds, uint8_t* __restrict__ out,
                     int64_t total, int D, int H, int W, int md, int mh, int mw) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int x = (int)(t % mw); t /= mw;
    const int y = (int)(t % mh); t /= mh;
    const int z = (int)(t % md);
    const int r = (int)(t / md);
    const int32_t* b = bounds + r * 6;
    int z1 = b[0], y1 = b[1], x1 = b[2], z2 = b[3], y2 = b[4], x2 = b[5];
    z1 = z1 < 0 ? 0 : z1; y1 = y1 < 0 ? 0 : y1; x1 = x1 < 0 ? 0 : x1;
    z2 = z2 > D ? D : z2; y2 = y2 > H ? H : y2; x2 = x2 > W ? W : x2;
    uint8_t v = 0;
    if (z2 > z1 && y2 > y1 && x2 > x1) {
      const int cd = z2 - z1, ch = y2 - y1, cw = x2 - x1;
      int sz = (int)floor(((double)z + 0.5) * ((double)cd / (double)md));
      int sy = (int)floor(((double)y + 0.5) * ((double)ch / (double)mh));
      int sx = (int)floor(((double)x + 0.5) * ((double)cw / (double)mw));
      sz = sz > cd - 1 ? cd - 1 : sz; sy = sy > ch - 1 ? ch - 1 : sy; sx = sx > cw - 1 ? cw - 1 : sx;
      v = labels[((int64_t)(z1 + sz) * H + (y1 + sy)) * W + (x1 + sx)];
    }
    out[i] = v;
  }
}

// Inference tail (utils.unmold_mask + the argmax of unmold_detections, utils.py:443-460, model.py:1853-1858): the
// detection's class probabilities [d,h,w,C] are resized to its box with F.interpolate(mode='trilinear',
// align_corners=False) semantics, pasted into a zero volume and arg-maxed over the classes.  Fused here: one thread
// per voxel of the FULL volume interpolates the C channels in registers and writes the class id -- the
// [D,H,W,C] fp32 tensor the reference builds on the host (268 MB at 256x256x128) is never materialised.
__device__ __forceinline__ void lin_src(int dst, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = (float)in / (float)out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 > in - 1 ? in - 1 : i0;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

template <int CT>
__global__ void __launch_bounds__(256)
k_unmold_argmax(const float* __restrict__ probs, uint8_t* __restrict__ out, int64_t total, int D, int H, int W, int md,
                int mh, int mw, int Crt, int z1, int y1, int x1, int z2, int y2, int x2) {
  const int C = CT > 0 ? CT : Crt;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int z = (int)(t / H);
    uint8_t best = 0;
    if (z >= z1 && z < z2 && y >= y1 && y < y2 && x >= x1 && x < x2) {
      int za, zb, ya, yb, xa, xb;
      float lz, ly, lx;
      lin_src(z - z1, md, z2 - z1, za, zb, lz);
      lin_src(y - y1, mh, y2 - y1, ya, yb, ly);
      lin_src(x - x1, mw, x2 - x1, xa, xb, lx);
      const float wz0 = 1.f - lz, wy0 = 1.f - ly, wx0 = 1.f - lx;
      const float* p000 = probs + (((int64_t)za * mh + ya) * mw + xa) * C;
      const float* p001 = probs + (((int64_t)za * mh + ya) * mw + xb) * C;
      const float* p010 = probs + (((int64_t)za * mh + yb) * mw + xa) * C;
      const float* p011 = probs + (((int64_t)za * mh + yb) * mw + xb) * C;
      const float* p100 = probs + (((int64_t)zb * mh + ya) * mw + xa) * C;
      const float* p101 = probs + (((int64_t)zb * mh + ya) * mw + xb) * C;
      const float* p110 = probs + (((int64_t)zb * mh + yb) * mw + xa) * C;
      const float* p111 = probs + (((int64_t)zb * mh + yb) * mw + xb) * C;
      float vmax = -INFINITY;
      for (int c = 0; c < C; ++c) {
        const float v = wz0 * (wy0 * (wx0 * p000[c] + lx * p001[c]) + ly * (wx0 * p010[c] + lx * p011[c])) +
                        lz * (wy0 * (wx0 * p100[c] + lx * p101[c]) + ly * (wx0 * p110[c] + lx * p111[c]));
        if (v > vmax) { vmax = v; best = (uint8_t)c; }      // first maximum wins, as np.argmax
      }
    }
    out[i] = best;
  }
}

// LiTS fork: overlap-tile un-molding (LiTS_2017/utils.py:383-408 + the argmax of LiTS_2017/model.py:1828-1829).  Every
// detection's class probabilities [md,mh,mw,C] are resized to its own box (trilinear, align_corners=False), ADDED into
// the full volume in detection order, divided by (hit count + 1e-6) and clipped to [0,1]; the class map is the
// arg-max.  One thread per voxel keeps the C running sums in registers, walks the (<= 64, kernel-argument) boxes in
// the reference's order -- same fp32 additions, same order -- and writes the class id and/or the averaged
// probabilities; the two [D,H,W,C] fp32 host arrays of the reference are never built.
constexpr int kMaxOverlapBoxes = 64;
struct OverlapBoxes { int32_t n; int32_t b[kMaxOverlapBoxes][6]; };

template <int CT>
__global__ void __launch_bounds__(256)
k_unmold_overlap(const float* __restrict__ probs, uint8_t* __restrict__ labels, float* __restrict__ full, int64_t total,
                 int D, int H, int W, int md, int mh, int mw, OverlapBoxes bx) {
  constexpr int C = CT;
  const int64_t mstride = (int64_t)md * mh * mw * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int z = (int)(t / H);
    float sum[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum[c] = 0.f;
    float count = 0.f;
    for (int k = 0; k < bx.n; ++k) {                        // uniform trip count; containment is per lane
      const int z1 = bx.b[k][0], y1 = bx.b[k][1], x1 = bx.b[k][2], z2 = bx.b[k][3], y2 = bx.b[k][4], x2 = bx.b[k][5];
      if (z < z1 || z >= z2 || y < y1 || y >= y2 || x < x1 || x >= x2) continue;
      int za, zb, ya, yb, xa, xb;
      float lz, ly, lx;
      lin_src(z - z1, md, z2 - z1, za, zb, lz);
      lin_src(y - y1, mh, y2 - y1, ya, yb, ly);
      lin_src(x - x1, mw, x2 - x1, xa, xb, lx);
      const float wz0 = 1.f - lz, wy0 = 1.f - ly, wx0 = 1.f - lx;
      const float* pk = probs + (int64_t)k * mstride;
      const float* p000 = pk + (((int64_t)za * mh + ya) * mw + xa) * C;
      const float* p001 = pk + (((int64_t)za * mh + ya) * mw + xb) * C;
      const float* p010 = pk + (((int64_t)za * mh + yb) * mw + xa) * C;
      const float* p011 = pk + (((int64_t)za * mh + yb) * mw + xb) * C;
      const float* p100 = pk + (((int64_t)zb * mh + ya) * mw + xa) * C;
      const float* p101 = pk + (((int64_t)zb * mh + ya) * mw + xb) * C;
      const float* p110 = pk + (((int64_t)zb * mh + yb) * mw + xa) * C;
      const float* p111 = pk + (((int64_t)zb * mh + yb) * mw + xb) * C;
#pragma unroll
      for (int c = 0; c < C; ++c)
        sum[c] += wz0 * (wy0 * (wx0 * p000[c] + lx * p001[c]) + ly * (wx0 * p010[c] + lx * p011[c])) +
                  lz * (wy0 * (wx0 * p100[c] + lx * p101[c]) + ly * (wx0 * p110[c] + lx * p111[c]));
      count += 1.f;
    }
    const float den = count + 1e-6f;
    uint8_t best = 0;
    float vmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = sum[c] / den;
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      if (full) full[i * C + c] = v;
      if (v > vmax) { vmax = v; best = (uint8_t)c; }        // first maximum wins, as np.argmax
    }
    if (labels) labels[i] = best;
  }
}

}  // namespace

extern "C" {

int cfun_roi_align3d_fwd(const float* fm, const float* boxes, float* out, int32_t* bounds, int32_t R, int32_t D,
                         int32_t H, int32_t W, int32_t C, int32_t pd, int32_t ph, int32_t pw, cfun_stream_t stream) {
  return cfun_roi_align3d_slab_fwd(fm, boxes, out, bounds, R, D, H, W, C, 0, D, pd, ph, pw, stream);
}

int cfun_roi_align3d_bwd(const float* dout, const int32_t* bounds, float* dfm, int32_t R, int32_t D, int32_t H,
                         int32_t W, int32_t C, int32_t pd, int32_t ph, int32_t pw, cfun_stream_t stream) {
  return cfun_roi_align3d_slab_bwd(dout, bounds, dfm, R, D, H, W, C, 0, D, pd, ph, pw, stream);
}

int cfun_roi_align3d_slab_fwd(const float* fm, const float* boxes, float* out, int32_t* bounds, int32_t R, int32_t D,
                              int32_t H, int32_t W, int32_t C, int32_t z0, int32_t dl, int32_t pd, int32_t ph, int32_t pw,
                              cfun_stream_t stream) {
  if (R <= 0) return CFUN_OK;
  if (D <= 0 || H <= 0 || W <= 0 || C <= 0 || pd <= 0 || ph <= 0 || pw <= 0) return CFUN_EINVAL;
  if (z0 < 0 || dl <= 0 || z0 + dl > D) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_roi_bounds, dim3((R + 63) / 64), dim3(64), 0, cfun_st(stream), boxes, bounds, R, D, H, W);
  const int64_t total = (int64_t)R * pd * ph * pw * C;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_roi_align_fwd, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), fm, bounds, out, total, D, H, W, C, pd, ph, pw, z0, dl);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_roi_align3d_slab_bwd(const float* dout, const int32_t* bounds, float* dfm, int32_t R, int32_t D, int32_t H,
                              int32_t W, int32_t C, int32_t z0, int32_t dl, int32_t pd, int32_t ph, int32_t pw,
                              cfun_stream_t stream) {
  if (z0 < 0 || dl <= 0 || z0 + dl > D) return CFUN_EINVAL;
  const int64_t total = (int64_t)dl * H * W * C;          // every element of the slab's gradient is WRITTEN (R = 0: zeros)
  if (total <= 0) return CFUN_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_roi_align_bwd, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), dout, bounds, dfm, total, R < 0 ? 0 : R, D, H, W, C, pd, ph, pw, z0, dl);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_unmold_argmax(const float* probs, uint8_t* out, int32_t D, int32_t H, int32_t W, int32_t md, int32_t mh,
                       int32_t mw, int32_t C, const int32_t* box, cfun_stream_t stream) {
  const int64_t total = (int64_t)D * H * W;
  if (total <= 0) return CFUN_OK;
  if (md <= 0 || mh <= 0 || mw <= 0 || C <= 0 || C > 255 || !box) return CFUN_EINVAL;
  if (box[0] < 0 || box[1] < 0 || box[2] < 0 || box[3] > D || box[4] > H || box[5] > W) return CFUN_EINVAL;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (C == 8)
    hipLaunchKernelGGL(k_unmold_argmax<8>, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), probs, out, total, D, H, W,
                       md, mh, mw, C, box[0], box[1], box[2], box[3], box[4], box[5]);
  else
    hipLaunchKernelGGL(k_unmold_argmax<0>, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), probs, out, total, D, H, W,
                       md, mh, mw, C, box[0], box[1], box[2], box[3], box[4], box[5]);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_unmold_overlap(const float* probs, const int32_t* boxes, int32_t n, uint8_t* labels, float* full, int32_t D,
                        int32_t H, int32_t W, int32_t md, int32_t mh, int32_t mw, int32_t C, cfun_stream_t stream) {
  const int64_t total = (int64_t)D * H * W;
  if (total <= 0 || (!labels && !full)) return CFUN_OK;
  if (n < 0 || n > kMaxOverlapBoxes || md <= 0 || mh <= 0 || mw <= 0 || (n > 0 && (!boxes || !probs))) return CFUN_EINVAL;
  OverlapBoxes bx;
  bx.n = n;
  for (int k = 0; k < n; ++k) {
    const int32_t* b = boxes + 6 * k;
    if (b[0] < 0 || b[1] < 0 || b[2] < 0 || b[3] > D || b[4] > H || b[5] > W) return CFUN_EINVAL;
    for (int j = 0; j < 6; ++j) bx.b[k][j] = b[j];
  }
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
#define CFUN_UNMOLD_OVERLAP(CT)                                                                                       \
  hipLaunchKernelGGL(k_unmold_overlap<CT>, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), probs, labels, full, \
                     total, D, H, W, md, mh, mw, bx)
  switch (C) {
    case 2: CFUN_UNMOLD_OVERLAP(2); break;
    case 3: CFUN_UNMOLD_OVERLAP(3); break;
    case 8: CFUN_UNMOLD_OVERLAP(8); break;
    default: return CFUN_EINVAL;
  }
#undef CFUN_UNMOLD_OVERLAP
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_mask_target_labels(const uint8_t* labels, const int32_t* bounds, uint8_t* out, int32_t R, int32_t D,
                            int32_t H, int32_t W, int32_t md, int32_t mh, int32_t mw, cfun_stream_t stream) {
  if (R <= 0) return CFUN_OK;
  if (D <= 0 || H <= 0 || W <= 0 || md <= 0 || mh <= 0 || mw <= 0) return CFUN_EINVAL;
  const int64_t total = (int64_t)R * md * mh * mw;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_mask_target_labels, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), labels, bounds, out,
                     total, D, H, W, md, mh, mw);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// workspace: order[n] i32 | sboxes[n*6] f32 | svol[n] f32 | mask[n*W64] u64
size_t cfun_nms3d_workspace_bytes(int32_t n) {
  if (n <= 0) return 256;
  const size_t W64 = (n + 63) / 64;
  return cfun_align_up((size_t)n * 4, 256) + cfun_align_up((size_t)n * 24, 256) + cfun_align_up((size_t)n * 4, 256) +
         cfun_align_up((size_t)n * W64 * 8, 256);
}

int cfun_nms3d(const float* boxes, const float* scores, int32_t n, float threshold, int32_t max_num, int32_t* keep,
               int32_t* count, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (n < 0 || n > 4096) return CFUN_EINVAL;
  if (n == 0) return (int)hipMemsetAsync(count, 0, sizeof(int32_t), cfun_st(stream));
  if (ws_bytes < cfun_nms3d_workspace_bytes(n)) return CFUN_EWORKSPACE;
  const int W64 = (n + 63) / 64;
  char* p = (char*)ws;
  int32_t* order = (int32_t*)p; p += cfun_align_up((size_t)n * 4, 256);
  float* sboxes = (float*)p; p += cfun_align_up((size_t)n * 24, 256);
  float* svol = (float*)p; p += cfun_align_up((size_t)n * 4, 256);
  unsigned long long* mask = (unsigned long long*)p;
  const int np2 = next_pow2(n);
  hipLaunchKernelGGL(k_nms_sort, dim3(1), dim3(1024), (size_t)np2 * 8, cfun_st(stream), boxes, scores, n, np2, order, sboxes, svol);
  hipLaunchKernelGGL(k_nms_mask, dim3((n * W64 + 255) / 256), dim3(256), 0, cfun_st(stream), sboxes, svol, n, W64, threshold, mask);
  const size_t lds = (size_t)n * W64 * sizeof(unsigned long long);
  static int scan_knob = -1;       // CFUN_NMS_SCAN_LDS = 0: the row-by-row kernel for every n (A/B, tests)
  if (scan_knob < 0) { const char* e = getenv("CFUN_NMS_SCAN_LDS"); scan_knob = e ? atoi(e) : 1; }
  // The LDS-resident scan may use all of a CU's LDS.  The limit is asked of the device once and the kernel's dynamic-LDS
  // ceiling raised to it once (it used to be set on every call, the 160 KB of gfx950 were hard-coded and a raw
  // hipError_t came back as a CFUN status); if either fails the row-by-row kernel -- same keep list -- takes the call.
  static const size_t lds_max = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
      return (size_t)(64 * 1024);
    if (v > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nms_scan_lds), hipFuncAttributeMaxDynamicSharedMemorySize, v) != hipSuccess)
      return (size_t)(64 * 1024);
    return (size_t)v;
  }();
  if (scan_knob != 0 && lds <= lds_max) {
    hipLaunchKernelGGL(k_nms_scan_lds, dim3(1), dim3(256), lds, cfun_st(stream), mask, order, n, W64, max_num, keep, count);
  } else {       // more rows than LDS holds (n > ~1 130): row by row from memory
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(64), 0, cfun_st(stream), mask, order, n, W64, max_num, keep, count);
  }
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // extern "C"
