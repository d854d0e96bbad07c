// EXPERIMENTAL (opt-in, see conv3d_b3.hip): weight gradient of the 3x3x3 stride-1 convolution on the bf16 matrix cores
// with the same exact 3-way operand split (six cross terms, fp32 accumulate).
//
//   dW[tap][ci][co] = sum over voxels v of  x[v + tap][ci] * g[v][co]
//
// is a GEMM whose contraction index is the VOXEL, so both MFMA operands must hold 8 consecutive voxels of one channel
// per lane -- the transpose of the NDHWC tensors.  A 512-thread workgroup (8 waves) stages, per 2(z) x 4(y) x 16(x)
// voxel tile, the halo of 16 input channels and the tile of NT = 16*NCO output-gradient channels into LDS
// channel-major ([plane][channel][voxel], bf16, three planes each: the split happens on the way in), then every wave
// takes 3-4 of the 27 taps: per K-step of 32 voxels (2 tile rows) it reads the NCO B operands (g, tap-invariant, reused
// for all its taps) and per tap one A operand (x) with a single ds_read_b128 whose address carries the tap shift --
// for dx = 1 that address is only 2-byte aligned, which gfx950's LDS serves (probed: tools/probes/lds_unaligned.hip).
// 18 / 30 MFMAs per 16-byte LDS read keep the LDS idle; the accumulators (taps x NCO x 4 registers) never leave the
// wave.  Workgroups own a range of tiles each and write fp32 partials [chunk][tap][ci][CoP] that the shared
// cfun_wgrad_finish reduces (deterministically) into the packed or OIDHW layout.
#include "b3_common.h"

int cfun_wgrad_finish(const float*, CfunWgradDst, const CfunConv3dParams*, int, hipStream_t);
int cfun_wgrad_zero(CfunWgradDst, const CfunConv3dParams*, hipStream_t);

namespace {

struct __attribute__((packed, aligned(2))) W3Unaligned16 { b3_u32x4 v; };     // a 16-byte LDS read at 2-byte alignment

constexpr int kTZ = 2, kTY = 4, kTX = 16, kTVox = kTZ * kTY * kTX;               // 128 output voxels per tile
constexpr int kIZ = kTZ + 2, kIY = kTY + 2, kIX = kTX + 2, kIVox = kIZ * kIY * kIX;   // 432-voxel halo
constexpr int kCHX = 880;      // bytes per channel of the x halo (864 used; 220 dwords: conflict-free 16-lane b128 reads)
constexpr int kPX = 16 * kCHX; // one bf16 plane of the 16-channel halo
constexpr int kCHG = 272;      // bytes per channel of the g tile (256 used; 68 dwords: conflict-free)
constexpr int kThreads = 512, kWaves = 8, kTapsPerWave = 4;

// Transposed staging writes whole dwords: a thread owns 2 (x halo) or 4 (g tile) CONSECUTIVE voxels of 4 channels, so
// each (channel, plane) costs one ds_write_b32 / b64 instead of one ds_write_b16 per element (the 16-bit scatter made
// the LDS write port, not the MFMA, the bottleneck: 170 writes per thread and tile).
__device__ __forceinline__ void w3_put_pair(unsigned char* base, int plane_bytes, int ch_bytes, const float4& v0,
                                            const float4& v1) {
  const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned w[3];
    b3_split_pair(e0[j], e1[j], w[0], w[1], w[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned*>(base + j * ch_bytes + pl * plane_bytes) = w[pl];
  }
}
__device__ __forceinline__ void w3_put_quad(unsigned char* base, int plane_bytes, int ch_bytes, const float4 (&v)[4]) {
  const float e[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                         {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned a[3], b[3];
    b3_split_pair(e[0][j], e[1][j], a[0], a[1], a[2]);
    b3_split_pair(e[2][j], e[3][j], b[0], b[1], b[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<b3_u32x2*>(base + j * ch_bytes + pl * plane_bytes) = b3_u32x2{a[pl], b[pl]};
  }
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

  // the wave's taps: wv, wv + 8, wv + 16, wv + 24 (< 27)
  int toff[kTapsPerWave];
#pragma unroll
  for (int ti = 0; ti < kTapsPerWave; ++ti) {
    int t = wv + kWaves * ti;
    t = t > 26 ? 26 : t;
    const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
    toff[ti] = ((dz * kIY + dy) * kIX + dx) * 2;
  }
  // lane -> (channel row, K-block): kb = (tile row within the K-step's pair, x half)
  const int row16 = lane & 15, kb = lane >> 4, kr = kb >> 1, kxh = kb & 1;
  const int a_lane = row16 * kCHX + (kr * kIX + 8 * kxh) * 2;
  const int b_lane = row16 * kCHG + (kr * kTX + 8 * kxh) * 2;

  b3_f32x4 acc[kTapsPerWave][NCO];
#pragma unroll
  for (int ti = 0; ti < kTapsPerWave; ++ti)
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
      const int idx = tid + i * kThreads, q = idx % (NT / 4), grp = idx / (NT / 4);
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
      if (idx < XQ) w3_put_pair(Xl + (idx & 3) * 4 * kCHX + (idx >> 2) * 4, kPX, kCHX, xr[i][0], xr[i][1]);
    }
#pragma unroll
    for (int i = 0; i < G_ITEMS; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < GQ) w3_put_quad(Gl + (idx % (NT / 4)) * 4 * kCHG + (idx / (NT / 4)) * 8, PG, kCHG, gr[i]);
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
      const unsigned char* xp = Xl + a_lane + ((rz * kIY + ry) * kIX) * 2;
#pragma unroll
      for (int ti = 0; ti < kTapsPerWave; ++ti) {
        if (wv + kWaves * ti > 26) continue;                   // wave-uniform: waves 3..7 have three taps
        const b3_u32x4 a0 = reinterpret_cast<const W3Unaligned16*>(xp + toff[ti])->v;
        const b3_u32x4 a1 = reinterpret_cast<const W3Unaligned16*>(xp + toff[ti] + kPX)->v;
        const b3_u32x4 a2 = reinterpret_cast<const W3Unaligned16*>(xp + toff[ti] + 2 * kPX)->v;
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_mfma(a2, bq[nn][0], acc[ti][nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_mfma(a1, bq[nn][1], acc[ti][nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_mfma(a0, bq[nn][2], acc[ti][nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_mfma(a1, bq[nn][0], acc[ti][nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_mfma(a0, bq[nn][1], acc[ti][nn]);
#pragma unroll
        for (int nn = 0; nn < NCO; ++nn) acc[ti][nn] = b3_mfma(a0, bq[nn][0], acc[ti][nn]);
      }
    }
  }

  // ---- partials: D row = ci (lane>>4)*4 + reg, column = co lane&15
  float* out = partial + (int64_t)chunk * 27 * p.Ci * p.CoP;
#pragma unroll
  for (int ti = 0; ti < kTapsPerWave; ++ti) {
    const int t = wv + kWaves * ti;
    if (t > 26) continue;
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
  w.nco = nsub % 3 == 0 ? 3 : nsub % 5 == 0 ? 5 : nsub % 2 == 0 ? 2 : 1;
  w.ncot = nsub / w.nco;
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

// supported AND measured faster than the exact-fp32 wgrad kernels (tools/bench_b3.py): 1.2-1.3x for C_in, C_out >= 80
// (80->80 @4x48^3 1.44 -> 1.21 ms, 160->160 @4x24^3 1.03 -> 0.79, 320->320 @4x12^3 0.55 -> 0.42), but 0.6-0.9x for the
// 20 / 40-channel layers: a workgroup stages and computes one tile at a time (81 KB of LDS = one workgroup per CU, so
// nothing overlaps the transposing stage-in) and the 16-row / 16-column padding costs 20-37 % there.
int cfun_conv3d_b3_wgrad_preferred(const CfunConv3dParams* p) {
  if (!p || !w3_shape_ok(p)) return 0;
  return (p->Ci >= 64 && p->Co >= 64) ? 1 : 0;
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
    case 5: rc = launch_w3<5>(x, g, (float*)ws, *p, w, st); break;
    case 2: rc = launch_w3<2>(x, g, (float*)ws, *p, w, st); break;
    default: rc = launch_w3<1>(x, g, (float*)ws, *p, w, st); break;
  }
  if (rc) return rc;
  return cfun_wgrad_finish((const float*)ws, dst, p, w.nchunks, st);
}

}  // extern "C"
