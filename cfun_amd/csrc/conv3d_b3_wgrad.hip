// EXPERIMENTAL (opt-in, see conv3d_b3.hip): weight gradient of the 3x3x3 stride-1 convolution on the bf16 matrix cores
// with the same exact 3-way operand split (six cross terms, fp32 accumulate).
//
//   dW[tap][ci][co] = sum over voxels v of  x[v + tap][ci] * g[v][co]
//
// is a GEMM whose contraction index is the VOXEL, so both MFMA operands must hold 8 consecutive voxels of one channel
// per lane -- the transpose of the NDHWC tensors.  A 576-thread workgroup (9 waves) stages, per 2(z) x 4(y) x 16(x)
// voxel tile, the halo of 16 input channels and the tile of NT = 16*NCO output-gradient channels into LDS
// channel-major ([plane][channel][voxel], bf16, three planes each: the split happens on the way in, whole dwords /
// qwords per write).  Wave w owns the tap row (dz, dy) = (w / 3, w % 3): per K-step of 32 voxels (2 tile rows) it reads
// the NCO B operands (g, tap-invariant) and ONE 16-byte-aligned x operand per plane (+ the next dword), and derives the
// three dx taps in registers -- dx = 1 is a 16-bit funnel shift (v_alignbit), dx = 2 a dword rename.  (A ds_read_b128
// at the tap's own, 2- or 4-byte aligned address returns the right data on gfx950 but costs 256 instead of 23 cycles --
// tools/probes/lds_unaligned_rate.hip -- and made the first version of this kernel LDS-bound at a third of the MFMA
// rate.)  The accumulators (3 taps x NCO x 4 registers) never leave the wave; workgroups own a range of tiles each
// and write fp32 partials [chunk][tap][ci][CoP] that the shared cfun_wgrad_finish reduces (deterministically) into the
// packed or OIDHW layout.
#include "b3_common.h"

int cfun_wgrad_finish(const float*, CfunWgradDst, const CfunConv3dParams*, int, hipStream_t);
int cfun_wgrad_zero(CfunWgradDst, const CfunConv3dParams*, hipStream_t);

namespace {


constexpr int kTZ = 2, kTY = 4, kTX = 16, kTVox = kTZ * kTY * kTX;               // 128 output voxels per tile
constexpr int kIZ = kTZ + 2, kIY = kTY + 2, kIX = kTX + 2, kIVox = kIZ * kIY * kIX;   // 432-voxel halo
constexpr int kXRow = 48;      // bytes per halo row in LDS: 18 voxels + padding, so that rows and x halves are 16-byte aligned
constexpr int kCHX = 1168;     // bytes per channel of the x halo (24 rows x 48 = 1152 used; 292 dwords: conflict-free b128 reads)
constexpr int kPX = 16 * kCHX; // one bf16 plane of the 16-channel halo
constexpr int kCHG = 272;      // bytes per channel of the g tile (256 used; 68 dwords: conflict-free)
constexpr int kThreads = 576;   // 9 waves: wave w <-> tap row (dz, dy) = (w / 3, w % 3), its three dx taps

// Transposed staging writes whole dwords: a thread owns 2 (x halo) or 4 (g tile) CONSECUTIVE voxels of 4 channels, so
// each (channel, plane) costs one ds_write_b32 / b64 instead of one ds_write_b16 per element (the 16-bit scatter made
// the LDS write port, not the MFMA, the bottleneck: 170 writes per thread and tile).
// (component by component: local float arrays built from the float4s end up in scratch memory)
__device__ __forceinline__ void w3_put2(unsigned char* p, int plane_bytes, float a, float b) {
  unsigned w0, w1, w2;
  b3_split_pair(a, b, w0, w1, w2);
  *reinterpret_cast<unsigned*>(p) = w0;
  *reinterpret_cast<unsigned*>(p + plane_bytes) = w1;
  *reinterpret_cast<unsigned*>(p + 2 * plane_bytes) = w2;
}
__device__ __forceinline__ void w3_put4(unsigned char* p, int plane_bytes, float a, float b, float c, float d) {
  unsigned a0, a1, a2, b0, b1, b2;
  b3_split_pair(a, b, a0, a1, a2);
  b3_split_pair(c, d, b0, b1, b2);
  *reinterpret_cast<b3_u32x2*>(p) = b3_u32x2{a0, b0};
  *reinterpret_cast<b3_u32x2*>(p + plane_bytes) = b3_u32x2{a1, b1};
  *reinterpret_cast<b3_u32x2*>(p + 2 * plane_bytes) = b3_u32x2{a2, b2};
}
__device__ __forceinline__ void w3_put_pair(unsigned char* base, int plane_bytes, int ch_bytes, const float4& v0,
                                            const float4& v1) {
  w3_put2(base, plane_bytes, v0.x, v1.x);
  w3_put2(base + ch_bytes, plane_bytes, v0.y, v1.y);
  w3_put2(base + 2 * ch_bytes, plane_bytes, v0.z, v1.z);
  w3_put2(base + 3 * ch_bytes, plane_bytes, v0.w, v1.w);
}
__device__ __forceinline__ void w3_put_quad(unsigned char* base, int plane_bytes, int ch_bytes, const float4& v0,
                                            const float4& v1, const float4& v2, const float4& v3) {
  w3_put4(base, plane_bytes, v0.x, v1.x, v2.x, v3.x);
  w3_put4(base + ch_bytes, plane_bytes, v0.y, v1.y, v2.y, v3.y);
  w3_put4(base + 2 * ch_bytes, plane_bytes, v0.z, v1.z, v2.z, v3.z);
  w3_put4(base + 3 * ch_bytes, plane_bytes, v0.w, v1.w, v2.w, v3.w);
}

template <int NCO>
__global__ void __launch_bounds__(kThreads)
k_wgrad_b3(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial, CfunConv3dParams p,
           int ntz, int nty, int ntx, int ntiles, int tiles_per_chunk, int ncisub, int ncot) {
  constexpr int NT = 16 * NCO, PG = NT * kCHG;
  constexpr int XQ = (kIVox / 2) * 4, X_ITEMS = (XQ + kThreads - 1) / kThreads;        // (voxel pair, channel quad) items
  constexpr int GQ = (kTVox / 4) * (NT / 4), G_ITEMS = (GQ + kThreads - 1) / kThreads;   // (voxel quad, channel quad) items
  CFUN_DYN_LDS(unsigned char, smem);
  unsigned char* Xl = smem;                 // [3][16][kCHX]
  unsigned char* Gl = smem + 3 * kPX;       // [3][NT][kCHG]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int cot = b % ncot; b /= ncot;
  const int cis = b % ncisub;
  const int chunk = b / ncisub;
  const int cobase = cot * NT, cibase = cis * 16;

  const int row_off = ((wv / 3) * kIY + (wv % 3)) * kXRow;      // the wave's (dz, dy) tap row
  // lane -> (channel row, K-block): kb = (tile row within the K-step's pair, x half)
  const int row16 = lane & 15, kb = lane >> 4, kr = kb >> 1, kxh = kb & 1;
  const int a_lane = row16 * kCHX + kr * kXRow + 16 * kxh + row_off;
  const int b_lane = row16 * kCHG + (kr * kTX + 8 * kxh) * 2;

  b3_f32x4 acc[3][NCO];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti)
#pragma unroll
    for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = t_begin + tiles_per_chunk < ntiles ? t_begin + tiles_per_chunk : ntiles;
  // global -> registers for one tile; the NEXT tile's loads are issued before the current tile's MFMA phase
  float4 xr[X_ITEMS][2], gr[G_ITEMS][4];
  auto load_tile = [&](int tile) {
    int tt = tile;
    const int tx = tt % ntx; tt /= ntx;
    const int ty = tt % nty; tt /= nty;
    const int tz = tt % ntz;
    const int n = tt / ntz;
    const int z0 = tz * kTZ, y0 = ty * kTY, x0 = tx * kTX;
#pragma unroll
    for (int i = 0; i < X_ITEMS; ++i) {
      const int idx = tid + i * kThreads, q = idx & 3, pr = idx >> 2;
      const int ixp = pr % (kIX / 2), iy = (pr / (kIX / 2)) % kIY, iz = pr / ((kIX / 2) * kIY);
      const int vz = z0 - 1 + iz, vy = y0 - 1 + iy, c = cibase + 4 * q;
      const bool row_ok = idx < XQ && vz >= 0 && vz < p.Di && vy >= 0 && vy < p.Hi && c < p.Ci;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int vx = x0 - 1 + 2 * ixp + k;
        xr[i][k] = (row_ok && vx >= 0 && vx < p.Wi)
                       ? *reinterpret_cast<const float4*>(x + ((((int64_t)n * p.Di + vz) * p.Hi + vy) * p.Wi + vx) * p.Ci + c)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < G_ITEMS; ++i) {
      // item -> (channel quad q, voxel quad grp): 4 quads fastest (64 contiguous bytes of one voxel in global memory),
      // then the 32 voxel quads (consecutive qwords of a channel row in LDS), then the remaining quads -- with all
      // NT/4 quads fastest the qword writes of a wave hit 2 bank groups 6-10 ways
      const int idx = tid + i * kThreads, grp = (idx >> 2) % (kTVox / 4), q = (idx / kTVox) * 4 + (idx & 3);
      const int ox4 = grp % (kTX / 4), oy = (grp / (kTX / 4)) % kTY, oz = grp / ((kTX / 4) * kTY);
      const int vz = z0 + oz, vy = y0 + oy, c = cobase + 4 * q;
      const bool row_ok = idx < GQ && vz < p.Do && vy < p.Ho && c < p.Co;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int vx = x0 + 4 * ox4 + k;
        gr[i][k] = (row_ok && vx < p.Wo)
                       ? *reinterpret_cast<const float4*>(g + ((((int64_t)n * p.Do + vz) * p.Ho + vy) * p.Wo + vx) * p.Co + c)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();           // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < X_ITEMS; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < XQ) w3_put_pair(Xl + (idx & 3) * 4 * kCHX + ((idx >> 2) / (kIX / 2)) * kXRow + ((idx >> 2) % (kIX / 2)) * 4, kPX, kCHX,
                                 xr[i][0], xr[i][1]);
    }
#pragma unroll
    for (int i = 0; i < G_ITEMS; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < GQ)
        w3_put_quad(Gl + ((idx / kTVox) * 4 + (idx & 3)) * 4 * kCHG + ((idx >> 2) % (kTVox / 4)) * 8, PG, kCHG, gr[i][0],
                    gr[i][1], gr[i][2], gr[i][3]);
    }
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);
    // ---- 4 K-steps of 32 voxels (tile rows 2ks, 2ks+1); x operand = A (rows = ci), g operand = B (columns = co)
#pragma unroll
    for (int ks = 0; ks < kTVox / 32; ++ks) {
      const int R = 2 * ks, rz = R / kTY, ry = R % kTY;
      const unsigned char* gp = Gl + b_lane + R * kTX * 2;
      b3_u32x4 bq[NCO][3];
#pragma unroll
      for (int nn = 0; nn < NCO; ++nn)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[nn][pl] = *reinterpret_cast<const b3_u32x4*>(gp + pl * PG + nn * 16 * kCHG);
      const unsigned char* xp = Xl + a_lane + (rz * kIY + ry) * kXRow;
      b3_u32x4 ax0[3], ax1[3], ax2[3];                    // the dx = 0 / 1 / 2 operands, [plane]
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const b3_u32x4 lo = *reinterpret_cast<const b3_u32x4*>(xp + pl * kPX);        // voxels x .. x+7 (16-byte aligned)
        const unsigned hi = *reinterpret_cast<const unsigned*>(xp + pl * kPX + 16);   // voxels x+8, x+9
        const unsigned l0 = lo.x, l1 = lo.y, l2 = lo.z, l3 = lo.w;
        ax0[pl] = lo;
        ax1[pl].x = (l0 >> 16) | (l1 << 16); ax1[pl].y = (l1 >> 16) | (l2 << 16);
        ax1[pl].z = (l2 >> 16) | (l3 << 16); ax1[pl].w = (l3 >> 16) | (hi << 16);
        ax2[pl].x = l1; ax2[pl].y = l2; ax2[pl].z = l3; ax2[pl].w = hi;
      }
      auto taps = [&](const b3_u32x4 (&a)[3], b3_f32x4 (&c)[NCO]) {        // six cross terms, smallest first
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) c[nn] = b3_mfma(a[2], bq[nn][0], c[nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) c[nn] = b3_mfma(a[1], bq[nn][1], c[nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) c[nn] = b3_mfma(a[0], bq[nn][2], c[nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) c[nn] = b3_mfma(a[1], bq[nn][0], c[nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) c[nn] = b3_mfma(a[0], bq[nn][1], c[nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) c[nn] = b3_mfma(a[0], bq[nn][0], c[nn]);
      };
      taps(ax0, acc[0]);
      taps(ax1, acc[1]);
      taps(ax2, acc[2]);
    }
  }

  // ---- partials: D row = ci (lane>>4)*4 + reg, column = co lane&15
  float* out = partial + (int64_t)chunk * 27 * p.Ci * p.CoP;
#pragma unroll
  for (int ti = 0; ti < 3; ++ti) {
    const int t = wv * 3 + ti;
#pragma unroll
    for (int nn = 0; nn < NCO; ++nn) {
      const int co = cobase + nn * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cibase + (lane >> 4) * 4 + r;
        if (ci < p.Ci && co < p.CoP) out[((int64_t)t * p.Ci + ci) * p.CoP + co] = acc[ti][nn][r];
      }
    }
  }
}

struct W3Plan { int ntz, nty, ntx, ntiles, ncisub, ncot, nco, nchunks, tiles_per_chunk; };

inline W3Plan w3_plan(const CfunConv3dParams& p) {
  W3Plan w;
  w.ntz = (p.Do + kTZ - 1) / kTZ; w.nty = (p.Ho + kTY - 1) / kTY; w.ntx = (p.Wo + kTX - 1) / kTX;
  w.ntiles = p.N * w.ntz * w.nty * w.ntx;
  w.ncisub = (p.Ci + 15) / 16;
  const int nsub = p.CoP / 16;
  // column sub-tiles per workgroup: 3 unless that pads the columns by more than 20 % (NCO = 5 would need more than the
  // 170 registers a 9-wave workgroup leaves each wave)
  w.nco = nsub == 1 ? 1 : ((nsub + 2) / 3 * 3 - nsub) * 5 <= nsub ? 3 : (nsub % 2 == 0 ? 2 : 3);
  if (w.nco == 2) w.nco = 1;      // k_wgrad_b3<2> spills to scratch under its 168-register cap (9 waves): see launch_b3 on scratch
  w.ncot = (nsub + w.nco - 1) / w.nco;
  // ~3 workgroups per CU in total (one resident per CU: 81 KB of LDS), at least 4 tiles each where the volume allows
  int want = (768 + w.ncisub * w.ncot - 1) / (w.ncisub * w.ncot);
  int maxc = (w.ntiles + 3) / 4;
  if (maxc < 1) maxc = 1;
  if (want > maxc) want = maxc;
  if (want < 1) want = 1;
  w.tiles_per_chunk = (w.ntiles + want - 1) / want;
  if (w.tiles_per_chunk < 1) w.tiles_per_chunk = 1;
  w.nchunks = (w.ntiles + w.tiles_per_chunk - 1) / w.tiles_per_chunk;
  return w;
}

inline bool w3_shape_ok(const CfunConv3dParams* p) {
  return p->kd == 3 && p->kh == 3 && p->kw == 3 && p->stride == 1 && p->pd == 1 && p->ph == 1 && p->pw == 1 && !p->up2 &&
         !p->d2s && !p->tap_skip && (p->Ci & 3) == 0 && (p->Co & 3) == 0 && p->Ci >= 8 && p->CoP == (p->Co + 15) / 16 * 16 &&
         p->Do == p->Di && p->Ho == p->Hi && p->Wo == p->Wi;
}

template <int NCO>
int launch_w3(const float* x, const float* g, float* partial, const CfunConv3dParams& p, const W3Plan& w, hipStream_t st) {
  const size_t lds = (size_t)3 * kPX + (size_t)3 * 16 * NCO * kCHG;
  auto kern = k_wgrad_b3<NCO>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t blocks = (int64_t)w.nchunks * w.ncisub * w.ncot;
  if (blocks > 0x7fffffffLL) return CFUN_EINVAL;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, g, partial, p, w.ntz, w.nty, w.ntx, w.ntiles,
                     w.tiles_per_chunk, w.ncisub, w.ncot);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // namespace

extern "C" {

int cfun_conv3d_b3_wgrad_supported(const CfunConv3dParams* p) { return p && w3_shape_ok(p) ? 1 : 0; }

// supported AND measured faster than the exact-fp32 wgrad kernels (tools/bench_b3.py): 1.2-1.6x from 16 x 40 channels
// up (40->40 @4x96^3 3.44 -> 2.34 ms, 80->80 @4x48^3 1.46 -> 1.20, 160->160 @4x24^3 1.04 -> 0.82, 320->320 @4x12^3
// 0.55 -> 0.39, 80->40 @4x48^3 0.87 -> 0.55); 20->20 (32 padded columns and rows for 20) stays at 0.92x.
int cfun_conv3d_b3_wgrad_preferred(const CfunConv3dParams* p) {
  if (!p || !w3_shape_ok(p)) return 0;
  return (p->Ci * p->Co >= 512) ? 1 : 0;
}

size_t cfun_conv3d_b3_wgrad_workspace_bytes(const CfunConv3dParams* p) {
  if (!p || !w3_shape_ok(p)) return 0;
  const W3Plan w = w3_plan(*p);
  return cfun_align_up((size_t)w.nchunks * 27 * p->Ci * p->CoP * sizeof(float), 256);
}

// dw: torch OIDHW [Co, Ci, 3, 3, 3] (the layout of nn.Conv3d.weight.grad); g = dL/d(conv sum) [N,D,H,W,Co]
int cfun_conv3d_b3_wgrad_oidhw(const float* x, const float* g, float* dw, const CfunConv3dParams* p, void* ws,
                               size_t ws_bytes, cfun_stream_t stream) {
  if (!p || !w3_shape_ok(p)) return CFUN_EINVAL;
  if (!cfun_aligned16(x) || !cfun_aligned16(g) || !cfun_aligned16(ws)) return CFUN_EALIGN;
  if (ws_bytes < cfun_conv3d_b3_wgrad_workspace_bytes(p)) return CFUN_EWORKSPACE;
  hipStream_t st = cfun_st(stream);
  const CfunWgradDst dst{dw, 1};
  const W3Plan w = w3_plan(*p);
  if (w.ntiles == 0) return cfun_wgrad_zero(dst, p, st);
  int rc;
  switch (w.nco) {
    case 3: rc = launch_w3<3>(x, g, (float*)ws, *p, w, st); break;
    default: rc = launch_w3<1>(x, g, (float*)ws, *p, w, st); break;
  }
  if (rc) return rc;
  return cfun_wgrad_finish((const float*)ws, dst, p, w.nchunks, st);
}

}  // extern "C"
