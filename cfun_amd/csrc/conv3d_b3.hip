// EXPERIMENTAL (opt-in, not on the default path): 3x3x3 stride-1 convolution with fp32 operands emulated on the bf16
// matrix cores ("3xBF16": every fp32 value is the exact sum of three bf16 values hi + mid + lo; of the nine cross
// products the six with weight >= 2^-16 are kept, each exact in fp32, accumulated in fp32 by the MFMA).  The dropped
// terms are <= 2^-25 |x w| per product -- below the rounding of an fp32 FMA -- so the result is fp32-accurate, while
// v_mfma_f32_16x16x32_bf16 retires 16x the MACs per cycle of v_mfma_f32_16x16x4_f32: 6 instructions of 16 cycles
// replace 8 of 32 for the same K = 32, a 2.67x higher MFMA ceiling (~420 effective TFLOP/s instead of 157).
//
// Layout.  A workgroup (4 waves) owns a 4(z) x 4(y) x 16(x) output tile x NT = 16*NSUB output channels, like
// k_conv_mfma.  K runs over (tap, channel): one MFMA K-step of 32 = 4 taps x 8 channels (lane group kb = lane>>4 takes
// tap 4s+kb of the 27, the 28th slot carries zero weights), so an 8-channel input chunk is staged per barrier pair:
// the fp32 halo tile is split into its three bf16 planes on the way into LDS ([plane][voxel][8 ch] = 16 B per voxel,
// one ds_read_b128 per B operand, at a per-lane tap offset).  The weights are pre-split and pre-swizzled on the device
// (cfun_weight_pack_b3; K padded with zero weights to a multiple of 8 channels) into exactly the per-lane A-operand
// order, [chunk][step][subtile][plane][lane][8 bf16], and are read straight from global memory / L2 with coalesced
// 16-byte loads, double-buffered one K-step ahead.
#include "b3_common.h"
#include "conv3d_mfma.h"

namespace {

using cfun_mfma::tile_raster;
using cfun_mfma::xcd_remap;

constexpr int kB3Steps = 7;            // ceil(27 taps / 4 taps per K-step)
constexpr int kB3IZ = 6, kB3IY = 6, kB3IX = 18, kB3Vox = kB3IZ * kB3IY * kB3IX;   // halo tile of the 4x4x16 outputs
constexpr int kB3Plane = kB3Vox * 16;  // bytes of one bf16 plane of an 8-channel chunk

// w OIDHW [Co][Ci][27] -> wb3[chunk = ci/8][step][subtile = co/16][plane][lane][8 bf16].  lane = (co & 15) + 16*kb holds
// tap 4*step + kb, channels 8*chunk .. +7.  transpose_flip: the data-gradient's weights (roles of Co / Ci swapped,
// taps mirrored) from the same OIDHW tensor: A row = ci, K = co.
__global__ void __launch_bounds__(256)
k_pack_b3(const float* __restrict__ w, unsigned short* __restrict__ wb3, int Co, int Ci, int rows, int kch, int nsub,
          int transpose_flip, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int e = (int)(t & 7); t >>= 3;
    const int lane = (int)(t & 63); t >>= 6;
    const int nn = (int)(t % nsub); t /= nsub;
    const int s = (int)(t % kB3Steps);
    const int c = (int)(t / kB3Steps);
    const int row = nn * 16 + (lane & 15), tap = 4 * s + (lane >> 4), k = 8 * c + e;
    float v = 0.f;
    if (tap < 27 && row < rows && k < kch)
      v = transpose_flip ? w[((int64_t)k * Ci + row) * 27 + (26 - tap)] : w[((int64_t)row * Ci + k) * 27 + tap];
    unsigned hi, mid, lo;
    b3_split(v, hi, mid, lo);
    const int64_t base = ((((int64_t)c * kB3Steps + s) * nsub + nn) * 3) * 512 + lane * 8 + e;
    wb3[base] = (unsigned short)hi;
    wb3[base + 512] = (unsigned short)mid;
    wb3[base + 1024] = (unsigned short)lo;
  }
}

// GROUPED (NSUB = 5 on launches that fill the chip): two workgroups per CU = two waves per SIMD, i.e. at most 256
// registers -- see the sub-tile groups below
// MODE 1: depth-to-space epilogue (p.d2s: the parity-folded "nearest x2 -> 5x5x5" conv, C_out = 8 parities x cq).
// MODE 2: the data gradient of such a conv: x is the HI-RES gradient [N,2D,2H,2W,cq]; channel chunk c of the low-res
//         "input" is the 8-channel slice o8 of parity q = c / (cq/8), gathered while staging (cq % 8 == 0).
template <int NSUB, bool GROUPED, int MODE>
__global__ void __launch_bounds__(256, GROUPED ? 2 : 1)
k_conv_b3(const float* __restrict__ x, const b3_u32x4* __restrict__ wb3, const float* __restrict__ scale,
          const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y, CfunConv3dParams p,
          int ntz, int nty, int ntx, int ncot, int nsub_total, int cq, float* __restrict__ partial, int chunks_per_split) {
  constexpr int NT = 16 * NSUB;
  constexpr int ITEMS = kB3Vox * 2, IN_LOADS = (ITEMS + 255) / 256;     // float4 (4 channels) items per chunk
  CFUN_DYN_LDS(unsigned char, smem);                                     // [2 buffers][3 planes][kB3Vox][16 B]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lid % ncot; lid /= ncot;
  const unsigned per_n = (unsigned)(ntz * nty * ntx);
  const int n = lid / per_n;
  int tz, ty, tx;
  tile_raster(lid - (unsigned)n * per_n, ntz, nty, ntx, tz, ty, tx);
  const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
  const int cobase = cot * NT;

  // ---- halo staging descriptors: item = (voxel, 4-channel half); element offset of channel 0 or -1 (zero padding)
  int64_t in_off[IN_LOADS];
#pragma unroll
  for (int i = 0; i < IN_LOADS; ++i) {
    const int idx = tid + i * 256, vox = idx >> 1;
    in_off[i] = -1;
    if (idx < ITEMS) {
      const int ix = vox % kB3IX, iy = (vox / kB3IX) % kB3IY, iz = vox / (kB3IX * kB3IY);
      const int vz = z0 - 1 + iz, vy = y0 - 1 + iy, vx = x0 - 1 + ix;
      if (vz >= 0 && vz < p.Di && vy >= 0 && vy < p.Hi && vx >= 0 && vx < p.Wi) {
        if (MODE == 2)      // parity-0 voxel of the hi-res tensor; the chunk's parity offset is added in prefetch_x()
          in_off[i] = ((((int64_t)n * 2 * p.Di + 2 * vz) * 2 * p.Hi + 2 * vy) * 2 * p.Wi + 2 * vx) * cq + (idx & 1) * 4;
        else
          in_off[i] = ((((int64_t)n * p.Di + vz) * p.Hi + vy) * p.Wi + vx) * p.Ci + (idx & 1) * 4;
      }
    }
  }
  float4 xin[IN_LOADS];
  auto prefetch_x = [&](int c) {
    int64_t coff = (int64_t)c * 8;
    int climit = p.Ci - c * 8;                // valid channels left in this chunk (C_in % 8 == 4: the last half is zero)
    if (MODE == 2) {
      const int cpq = cq >> 3, q = c / cpq, o8 = (c - q * cpq) * 8;
      coff = ((int64_t)((q >> 2) * 2 * p.Hi + ((q >> 1) & 1)) * 2 * p.Wi + (q & 1)) * cq + o8;
      climit = cq - o8;
    }
#pragma unroll
    for (int i = 0; i < IN_LOADS; ++i)
      xin[i] = (in_off[i] >= 0 && ((tid + i * 256) & 1) * 4 < climit)
                   ? *reinterpret_cast<const float4*>(x + in_off[i] + coff) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto commit_x = [&](int buf) {          // split the prefetched fp32 halo into its three bf16 planes of LDS buffer `buf`
#pragma unroll
    for (int i = 0; i < IN_LOADS; ++i) {
      const int idx = tid + i * 256;
      if (idx < ITEMS) {
        unsigned h0, m0, l0, h1, m1, l1;
        b3_split_pair(xin[i].x, xin[i].y, h0, m0, l0);
        b3_split_pair(xin[i].z, xin[i].w, h1, m1, l1);
        unsigned char* dst = smem + buf * (3 * kB3Plane) + (idx >> 1) * 16 + (idx & 1) * 8;
        *reinterpret_cast<b3_u32x2*>(dst) = b3_u32x2{h0, h1};
        *reinterpret_cast<b3_u32x2*>(dst + kB3Plane) = b3_u32x2{m0, m1};
        *reinterpret_cast<b3_u32x2*>(dst + 2 * kB3Plane) = b3_u32x2{l0, l1};
      }
    }
  };

  // ---- B operand: lane (v = lane & 15, kb = lane >> 4) reads voxel (wv + dz, m + dy, v + dx) of tap 4s + kb
  const int kb = lane >> 4;
  int boff[kB3Steps];
#pragma unroll
  for (int s = 0; s < kB3Steps; ++s) {
    int t = 4 * s + kb;
    t = t > 26 ? 26 : t;                                  // slot 27: any valid address, its weights are zero
    const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
    boff[s] = ((((wv + dz) * kB3IY + dy) * kB3IX) + (lane & 15) + dx) * 16;
  }
  // ---- A operand: [chunk][step][subtile][plane][lane] 16-byte entries
  const b3_u32x4* wl = wb3 + (int64_t)(cot * NSUB) * 3 * 64 + lane;
  const int64_t wstep = (int64_t)nsub_total * 3 * 64;
  // GROUPED: the sub-tiles of a K-step are taken in two groups (3 + 2) that re-read the B operands from the (idle) LDS,
  // so that only one group's weights are double-buffered in registers: 312 -> 256 registers for NSUB = 5, two waves
  // per SIMD instead of one (80->80 @4x48^3: 0.74 -> 0.69 ms; it loses where the launch is small anyway)
  constexpr int H = GROUPED ? 2 : 1, G0 = GROUPED ? (NSUB + 1) / 2 : NSUB;     // groups per step, size of the first
  b3_u32x4 a_cur[G0][3], a_nxt[G0][3];
  auto load_a = [&](b3_u32x4 (&a)[G0][3], int u) {         // u = (chunk * 7 + step) * H + group
    const int g = u / H, h = u - g * H;
    const b3_u32x4* src = wl + (int64_t)g * wstep + (int64_t)(h * G0) * 3 * 64;
#pragma unroll
    for (int nn = 0; nn < G0; ++nn)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        if (h * G0 + nn < NSUB) a[nn][pl] = src[(nn * 3 + pl) * 64];
  };

  b3_f32x4 acc[4][NSUB];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) acc[m][nn] = b3_f32x4{0.f, 0.f, 0.f, 0.f};

  // Two LDS buffers: chunk c+1 is split and written (VALU + ds_write, in the middle of chunk c's MFMA stream) while
  // chunk c is being read, so a chunk costs ONE barrier and the conversion work hides behind the matrix pipe.
  // split-K (small volumes: too few tiles to fill 256 CUs): blockIdx.y owns a range of channel chunks and stores raw
  // accumulators to partial[blockIdx.y]; cfun_splitk_finish sums them in order and runs the epilogue
  const int nchunks_all = (p.Ci + 7) >> 3;
  const int c_first = blockIdx.y * chunks_per_split;
  const int nchunks = (c_first + chunks_per_split < nchunks_all ? c_first + chunks_per_split : nchunks_all);
  const int nsteps = nchunks * kB3Steps;
  if (c_first >= nchunks) return;
  prefetch_x(c_first);
  load_a(a_cur, c_first * kB3Steps * H);
  commit_x(c_first & 1);
  if (c_first + 1 < nchunks) prefetch_x(c_first + 1);
  __syncthreads();
  for (int c = c_first; c < nchunks; ++c) {
    const unsigned char* xb = smem + (c & 1) * (3 * kB3Plane);
#pragma unroll
    for (int s = 0; s < kB3Steps; ++s) {
      const int g = c * kB3Steps + s;
      if (s == 3 && c + 1 < nchunks) {       // every wave left buffer (c+1)&1 at the barrier that ended chunk c-1
        commit_x((c + 1) & 1);
        if (c + 2 < nchunks) prefetch_x(c + 2);
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int u = g * H + h;
        if (u + 1 < nsteps * H) load_a(a_nxt, u + 1);
        const int n0 = h * G0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const unsigned char* bp = xb + boff[s] + m * (kB3IX * 16);
          const b3_u32x4 b0 = *reinterpret_cast<const b3_u32x4*>(bp);
          const b3_u32x4 b1 = *reinterpret_cast<const b3_u32x4*>(bp + kB3Plane);
          const b3_u32x4 b2 = *reinterpret_cast<const b3_u32x4*>(bp + 2 * kB3Plane);
          // six cross terms, smallest first; consecutive MFMAs go to different accumulators
#pragma unroll
          for (int nn = 0; nn < G0; ++nn) if (n0 + nn < NSUB) acc[m][n0 + nn] = b3_mfma(a_cur[nn][2], b0, acc[m][n0 + nn]);
#pragma unroll
          for (int nn = 0; nn < G0; ++nn) if (n0 + nn < NSUB) acc[m][n0 + nn] = b3_mfma(a_cur[nn][1], b1, acc[m][n0 + nn]);
#pragma unroll
          for (int nn = 0; nn < G0; ++nn) if (n0 + nn < NSUB) acc[m][n0 + nn] = b3_mfma(a_cur[nn][0], b2, acc[m][n0 + nn]);
#pragma unroll
          for (int nn = 0; nn < G0; ++nn) if (n0 + nn < NSUB) acc[m][n0 + nn] = b3_mfma(a_cur[nn][1], b0, acc[m][n0 + nn]);
#pragma unroll
          for (int nn = 0; nn < G0; ++nn) if (n0 + nn < NSUB) acc[m][n0 + nn] = b3_mfma(a_cur[nn][0], b1, acc[m][n0 + nn]);
#pragma unroll
          for (int nn = 0; nn < G0; ++nn) if (n0 + nn < NSUB) acc[m][n0 + nn] = b3_mfma(a_cur[nn][0], b0, acc[m][n0 + nn]);
        }
#pragma unroll
        for (int nn = 0; nn < G0; ++nn)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a_cur[nn][pl] = a_nxt[nn][pl];
      }
    }
    __syncthreads();           // chunk c+1 is complete in its buffer; chunk c's buffer is free
  }

  // ---- epilogue: lane owns voxel (z0+wv, y0+m, x0+(lane&15)), channels nn*16 + (lane>>4)*4 .. +3
  const int oz = z0 + wv, ox = x0 + (lane & 15);
  if (oz >= p.Do || ox >= p.Wo) return;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int oy = y0 + m;
    if (oy >= p.Ho) continue;
    const int64_t v = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
    for (int nn = 0; nn < NSUB; ++nn) {
      const int co = cobase + nn * 16 + kb * 4;
      if (co >= p.Co) continue;
      float4 r = make_float4(acc[m][nn][0], acc[m][nn][1], acc[m][nn][2], acc[m][nn][3]);
      if (gridDim.y > 1) {       // split-K partial: raw sums, plain layout
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.y * p.N * p.Do * p.Ho * p.Wo + v) * p.Co + co) = r;
        continue;
      }
      if (p.scale_mode) {
        const float4 s4 = *reinterpret_cast<const float4*>(scale + (p.scale_mode == 2 ? n * p.Co : 0) + co);
        r.x *= s4.x; r.y *= s4.y; r.z *= s4.z; r.w *= s4.w;
      }
      if (p.has_shift) {
        const float4 t = *reinterpret_cast<const float4*>(shift + co);
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      if (MODE == 1) {        // depth-to-space: channel (parity q, oc) of low-res voxel v -> hi-res voxel 2v + q
        const int CqP = p.Co >> 3, q = co / CqP, oc = co - q * CqP;
        if (oc >= cq) continue;                                   // per-parity channel padding
        if (p.res_mode) {                                         // the residual is low-res: nearest x2 up-sampled into y
          const float4 t = *reinterpret_cast<const float4*>(res + v * cq + oc);
          r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        r.x = cfun_apply_act(r.x, p.act, p.slope); r.y = cfun_apply_act(r.y, p.act, p.slope);
        r.z = cfun_apply_act(r.z, p.act, p.slope); r.w = cfun_apply_act(r.w, p.act, p.slope);
        const int64_t hv = (((int64_t)n * 2 * p.Do + 2 * oz + (q >> 2)) * 2 * p.Ho + 2 * oy + ((q >> 1) & 1)) * 2 * p.Wo +
                           2 * ox + (q & 1);
        *reinterpret_cast<float4*>(y + hv * cq + oc) = r;
        continue;
      }
      if (p.res_mode) {
        const float4 t = *reinterpret_cast<const float4*>(res + v * p.Co + co);
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      r.x = cfun_apply_act(r.x, p.act, p.slope); r.y = cfun_apply_act(r.y, p.act, p.slope);
      r.z = cfun_apply_act(r.z, p.act, p.slope); r.w = cfun_apply_act(r.w, p.act, p.slope);
      *reinterpret_cast<float4*>(y + v * p.Co + co) = r;
    }
  }
}

// split-K factor for launches of fewer than ~256 workgroups (each split keeps >= 2 chunks = 14 K-steps of work)
inline int b3_splitk(int64_t nblk, int nchunks, const CfunConv3dParams& p, size_t ws_bytes) {
  if (nblk >= 256 || nchunks < 4) return 1;
  int k = (int)((512 + nblk - 1) / nblk);
  if (k > nchunks / 2) k = nchunks / 2;
  if (k > 16) k = 16;
  const size_t per = (size_t)p.N * p.Do * p.Ho * p.Wo * p.Co * sizeof(float);
  while (k > 1 && (size_t)k * per > ws_bytes) --k;
  return k < 1 ? 1 : k;
}

template <int NSUB, bool GROUPED = false, int MODE = 0>
int launch_b3(const float* x, const void* wb3, const float* scale, const float* shift, const float* res, float* y,
              const CfunConv3dParams& p, int nsub_total, int cq, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
  const int ntz = (p.Do + 3) / 4, nty = (p.Ho + 3) / 4, ntx = (p.Wo + 15) / 16, ncot = nsub_total / NSUB;
  const int64_t nblk = (int64_t)p.N * ntz * nty * ntx * ncot;
  if (nblk == 0) return CFUN_OK;
  if (nblk > 0x7fffffffLL) return CFUN_EINVAL;
  // (The GROUPED form of the NSUB = 5 kernel -- two workgroups per CU inside 256 registers -- is not dispatched any more: it
  // spills 112 bytes per lane to SCRATCH, and a scratch-using kernel on the mask head's side stream corrupted 64-byte pieces of
  // tensors the main stream was writing whenever torch's allocator had just returned memory to the driver and asked for new
  // blocks (round 4, tools/probe_repro.py: 14 of 23 steps not bit-reproducible; 0 of 23 with every kernel of the library at
  // scratch size 0).  No kernel of libcfun_hip.so may use scratch: tests/test_abi.py checks the code objects.)
  const int nchunks = (p.Ci + 7) >> 3;
  int ksplit = 1;
  if (MODE == 0) ksplit = b3_splitk(nblk, nchunks, p, ws && cfun_aligned16(ws) ? ws_bytes : 0);
  const int cps = (nchunks + ksplit - 1) / ksplit;
  ksplit = (nchunks + cps - 1) / cps;
  hipLaunchKernelGGL((k_conv_b3<NSUB, GROUPED, MODE>), dim3((unsigned)nblk, (unsigned)ksplit), dim3(256),
                     (size_t)2 * 3 * kB3Plane, st, x, (const b3_u32x4*)wb3, scale, shift, res, y, p, ntz, nty, ntx, ncot,
                     nsub_total, cq, (float*)ws, cps);
  CFUN_LAUNCH_CHECK();
  if (ksplit > 1) return cfun_splitk_finish((const float*)ws, ksplit, scale, shift, res, y, &p, nullptr, st);
  return CFUN_OK;
}

inline int b3_d2s_cq(const CfunConv3dParams* p) { return p->d2s_cq > 0 ? p->d2s_cq : (p->Co >> 3); }

inline bool b3_shape_ok(const CfunConv3dParams* p) {
  if (!(p->kd == 3 && p->kh == 3 && p->kw == 3 && p->stride == 1 && p->pd == 1 && p->ph == 1 && p->pw == 1 && !p->up2 &&
        !p->tap_skip && (p->Ci & 3) == 0 && (p->Co & 3) == 0 && p->Ci >= 8 && p->Do == p->Di && p->Ho == p->Hi &&
        p->Wo == p->Wi))
    return false;
  if (p->d2s)     // parity-folded up-conv without tap skipping (the 5x5x5 one): 8 parity groups of 4-aligned channels
    return (p->Co & 7) == 0 && ((p->Co >> 3) & 3) == 0 && (b3_d2s_cq(p) & 3) == 0;
  return !p->res_up2;
}

// the data gradient of a d2s conv on these kernels: chunks of 8 channels must not straddle a parity
inline bool b3_d2s_dgrad_ok(const CfunConv3dParams* p) {
  return b3_shape_ok(p) && p->d2s && (b3_d2s_cq(p) & 7) == 0 && p->Co == 8 * b3_d2s_cq(p);
}

inline int b3_nsub_per_block(int nsub) { return nsub % 3 == 0 ? 3 : nsub % 5 == 0 ? 5 : nsub % 4 == 0 ? 4 : nsub % 2 == 0 ? 2 : 1; }

}  // namespace

extern "C" {

int cfun_conv3d_b3_supported(const CfunConv3dParams* p) { return p && b3_shape_ok(p) ? 1 : 0; }

// supported AND expected to beat the exact-fp32 MFMA kernel: with fewer workgroups than ~half the CUs the fp32 path's
// split-K wins (measured: 64-72 workgroups 0.5-0.8x, 144 1.24x, >= 288 1.3-2.3x; tools/bench_b3.py)
int cfun_conv3d_b3_preferred(const CfunConv3dParams* p) {
  if (!p || !b3_shape_ok(p)) return 0;
  const int nsub = (p->Co + 15) / 16;
  const int64_t nblk = (int64_t)p->N * ((p->Do + 3) / 4) * ((p->Ho + 3) / 4) * ((p->Wo + 15) / 16) * (nsub / b3_nsub_per_block(nsub));
  const int nchunks = (p->Ci + 7) >> 3;
  const int k = p->d2s ? 1 : b3_splitk(nblk, nchunks, *p, (size_t)-1);      // with the workspace the caller is asked for
  return nblk * k >= 120 ? 1 : 0;
}

// split-K partials for volumes too small to fill the chip (0: none needed); ws may be NULL -- the kernel then runs unsplit
size_t cfun_conv3d_b3_fwd_workspace_bytes(const CfunConv3dParams* p) {
  if (!p || !b3_shape_ok(p) || p->d2s) return 0;
  const int nsub = (p->Co + 15) / 16;
  const int64_t nblk = (int64_t)p->N * ((p->Do + 3) / 4) * ((p->Ho + 3) / 4) * ((p->Wo + 15) / 16) * (nsub / b3_nsub_per_block(nsub));
  const int k = b3_splitk(nblk, (p->Ci + 7) >> 3, *p, (size_t)-1);
  return k > 1 ? cfun_align_up((size_t)k * p->N * p->Do * p->Ho * p->Wo * p->Co * sizeof(float), 256) : 0;
}

int cfun_conv3d_b3_dgrad_d2s_supported(const CfunConv3dParams* p) { return p && b3_d2s_dgrad_ok(p) ? 1 : 0; }

size_t cfun_weight_pack_b3_bytes(int32_t rows, int32_t kch) {
  if (rows <= 0 || kch <= 0) return 0;
  return (size_t)((kch + 7) / 8) * kB3Steps * ((rows + 15) / 16) * 3 * 64 * 16;
}

int cfun_weight_pack_b3(const float* w, void* wb3, int32_t Co, int32_t Ci, int32_t transpose_flip, cfun_stream_t stream) {
  const int rows = transpose_flip ? Ci : Co, kch = transpose_flip ? Co : Ci;
  if (rows <= 0 || kch <= 0) return CFUN_EINVAL;
  const int nsub = (rows + 15) / 16;
  const int64_t total = (int64_t)((kch + 7) / 8) * kB3Steps * nsub * 64 * 8;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_pack_b3, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), w, (unsigned short*)wb3, Co, Ci, rows,
                     kch, nsub, transpose_flip, total);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_conv3d_b3_fwd(const float* x, const void* wb3, const float* scale, const float* shift, const float* res, float* y,
                       const CfunConv3dParams* p, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (!p || !b3_shape_ok(p)) return CFUN_EINVAL;
  if (!cfun_aligned16(x) || !cfun_aligned16(wb3) || !cfun_aligned16(y)) return CFUN_EALIGN;
  const int nsub = (p->Co + 15) / 16;
  hipStream_t st = cfun_st(stream);
  const int cq = p->d2s ? b3_d2s_cq(p) : 0;
  if (p->d2s) {
    switch (b3_nsub_per_block(nsub)) {
      case 3: return launch_b3<3, false, 1>(x, wb3, scale, shift, res, y, *p, nsub, cq, st);
      case 4: return launch_b3<4, false, 1>(x, wb3, scale, shift, res, y, *p, nsub, cq, st);
      case 2: return launch_b3<2, false, 1>(x, wb3, scale, shift, res, y, *p, nsub, cq, st);
      default: return launch_b3<1, false, 1>(x, wb3, scale, shift, res, y, *p, nsub, cq, st);
    }
  }
  switch (b3_nsub_per_block(nsub)) {
    case 3: return launch_b3<3>(x, wb3, scale, shift, res, y, *p, nsub, cq, st, ws, ws_bytes);
    case 5: return launch_b3<5>(x, wb3, scale, shift, res, y, *p, nsub, cq, st, ws, ws_bytes);
    case 4: return launch_b3<4>(x, wb3, scale, shift, res, y, *p, nsub, cq, st, ws, ws_bytes);
    case 2: return launch_b3<2>(x, wb3, scale, shift, res, y, *p, nsub, cq, st, ws, ws_bytes);
    default: return launch_b3<1>(x, wb3, scale, shift, res, y, *p, nsub, cq, st, ws, ws_bytes);
  }
}

// Data gradient of a d2s conv (p = the FORWARD conv's parameters): g = dL/dy in y's hi-res layout [N,2D,2H,2W,cq],
// wb3t = cfun_weight_pack_b3(w, transpose_flip = 1) of the folded OIDHW weight [8*cq, Ci, 3,3,3], dx [N,D,H,W,Ci].
int cfun_conv3d_b3_dgrad_d2s(const float* g, const void* wb3t, float* dx, const CfunConv3dParams* p, cfun_stream_t stream) {
  if (!p || !b3_d2s_dgrad_ok(p)) return CFUN_EINVAL;
  if (!cfun_aligned16(g) || !cfun_aligned16(wb3t) || !cfun_aligned16(dx)) return CFUN_EALIGN;
  CfunConv3dParams q = *p;
  q.Ci = p->Co; q.Co = p->Ci; q.CoP = (p->Ci + 15) / 16 * 16; q.CiP = p->CoP;
  q.d2s = 0; q.d2s_cq = 0; q.res_up2 = 0; q.res_mode = 0; q.scale_mode = 0; q.has_shift = 0; q.act = CFUN_ACT_NONE;
  const int nsub = q.CoP / 16, cq = b3_d2s_cq(p);
  hipStream_t st = cfun_st(stream);
  switch (b3_nsub_per_block(nsub)) {
    case 3: return launch_b3<3, false, 2>(g, wb3t, nullptr, nullptr, nullptr, dx, q, nsub, cq, st);
    case 2: return launch_b3<2, false, 2>(g, wb3t, nullptr, nullptr, nullptr, dx, q, nsub, cq, st);
    default: return launch_b3<1, false, 2>(g, wb3t, nullptr, nullptr, nullptr, dx, q, nsub, cq, st);
  }
}

}  // extern "C"
