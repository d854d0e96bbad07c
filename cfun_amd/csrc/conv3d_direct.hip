// Generic direct 3-D convolution, NDHWC fp32 -- the "any shape" VALU path (CFUN_ALGO_DIRECT).
//
// Used for the C_in = 1 stems (backbone.py:124, mask_branch.py:23: HBM-bound, AI 13-49), for channel
// counts the MFMA path does not take (C % 4 != 0, e.g. the 3-class LiTS heads) and as the on-device
// cross-check of the MFMA kernels in tests.  Weights are wave-uniform, so the compiler keeps them in
// SGPRs (s_load) and every v_fmac reads one VGPR activation + one SGPR weight.
#include "common.h"

namespace {

struct Vox {
  int n, z, y, x;
};

__device__ __forceinline__ Vox decompose(int64_t v, int D, int H, int W) {
  Vox r;
  r.x = (int)(v % W);
  v /= W;
  r.y = (int)(v % H);
  v /= H;
  r.z = (int)(v % D);
  r.n = (int)(v / D);
  return r;
}

// ---------------------------------------------------------------- forward
template <int COT>
__global__ void __launch_bounds__(256)
k_conv_fwd_direct(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
                  const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
                  CfunConv3dParams p, int64_t total) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * COT;
  if (v >= total) return;
  const Vox o = decompose(v, p.Do, p.Ho, p.Wo);
  const int Dv = p.up2 ? 2 * p.Di : p.Di, Hv = p.up2 ? 2 * p.Hi : p.Hi, Wv = p.up2 ? 2 * p.Wi : p.Wi;
  const int sh = p.up2 ? 1 : 0;
  float acc[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) acc[j] = 0.f;
  for (int dz = 0; dz < p.kd; ++dz) {
    const int iz = o.z * p.stride + dz - p.pd;
    if (iz < 0 || iz >= Dv) continue;
    for (int dy = 0; dy < p.kh; ++dy) {
      const int iy = o.y * p.stride + dy - p.ph;
      if (iy < 0 || iy >= Hv) continue;
      for (int dx = 0; dx < p.kw; ++dx) {
        const int ix = o.x * p.stride + dx - p.pw;
        if (ix < 0 || ix >= Wv) continue;
        const int tap = (dz * p.kh + dy) * p.kw + dx;
        const float* xp = x + ((((int64_t)o.n * p.Di + (iz >> sh)) * p.Hi + (iy >> sh)) * p.Wi + (ix >> sh)) * p.Ci;
        const float* wrow = wp + (int64_t)tap * p.Ci * p.CoP + co0;
        for (int ci = 0; ci < p.Ci; ++ci) {
          const float xv = xp[ci];
#pragma unroll
          for (int j = 0; j < COT; ++j) acc[j] = fmaf(xv, wrow[(int64_t)ci * p.CoP + j], acc[j]);
        }
      }
    }
  }
  int64_t ridx = 0;
  if (p.res_mode) {
    ridx = (p.res_up2 && !p.d2s)
               ? ((((int64_t)o.n * (p.Do >> 1) + (o.z >> 1)) * (p.Ho >> 1) + (o.y >> 1)) * (p.Wo >> 1) + (o.x >> 1))
               : v;
  }
  const int CqP = p.Co >> 3, Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;   // d2s only: padded / valid channels per parity
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    const int co = co0 + j;
    if (co < p.Co) {
      float r = acc[j];
      if (p.scale_mode == 1) r *= scale[co];
      else if (p.scale_mode == 2) r *= scale[o.n * p.Co + co];
      if (p.has_shift) r += shift[co];
      if (p.d2s) {
        const int q = co / CqP, oc = co - q * CqP;
        if (oc >= Cq) continue;
        if (p.res_mode) r += res[ridx * Cq + oc];
        const int64_t hv = (((int64_t)o.n * 2 * p.Do + 2 * o.z + (q >> 2)) * 2 * p.Ho + 2 * o.y + ((q >> 1) & 1)) * 2 * p.Wo +
                           2 * o.x + (q & 1);
        y[hv * Cq + oc] = cfun_apply_act(r, p.act, p.slope);
      } else {
        if (p.res_mode) r += res[ridx * p.Co + co];
        y[v * p.Co + co] = cfun_apply_act(r, p.act, p.slope);
      }
    }
  }
}

// ---------------------------------------------------------------- backward data (transposed gather)
template <int CIT>
__global__ void __launch_bounds__(256)
k_conv_bwd_data_direct(const float* __restrict__ g, const float* __restrict__ wpT, float* __restrict__ dx,
                       CfunConv3dParams p, int64_t total) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int ci0 = blockIdx.y * CIT;
  if (v >= total) return;
  const Vox i = decompose(v, p.Di, p.Hi, p.Wi);
  const int nch = p.up2 ? 2 : 1;
  float acc[CIT];
#pragma unroll
  for (int j = 0; j < CIT; ++j) acc[j] = 0.f;
  for (int cz = 0; cz < nch; ++cz)
    for (int cy = 0; cy < nch; ++cy)
      for (int cx = 0; cx < nch; ++cx) {
        const int vz = p.up2 ? 2 * i.z + cz : i.z, vy = p.up2 ? 2 * i.y + cy : i.y, vx = p.up2 ? 2 * i.x + cx : i.x;
        for (int dz = 0; dz < p.kd; ++dz) {
          const int tz = vz + p.pd - dz;
          if (tz < 0 || (tz % p.stride) != 0) continue;
          const int zo = tz / p.stride;
          if (zo >= p.Do) continue;
          for (int dy = 0; dy < p.kh; ++dy) {
            const int ty = vy + p.ph - dy;
            if (ty < 0 || (ty % p.stride) != 0) continue;
            const int yo = ty / p.stride;
            if (yo >= p.Ho) continue;
            for (int dxx = 0; dxx < p.kw; ++dxx) {
              const int tx = vx + p.pw - dxx;
              if (tx < 0 || (tx % p.stride) != 0) continue;
              const int xo = tx / p.stride;
              if (xo >= p.Wo) continue;
              const int tap = (dz * p.kh + dy) * p.kw + dxx;
              const float* gp = g + ((((int64_t)i.n * p.Do + zo) * p.Ho + yo) * p.Wo + xo) * p.Co;
              const float* wrow = wpT + (int64_t)tap * p.Co * p.CiP + ci0;
              const int CqP = p.Co >> 3, Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;
              for (int co = 0; co < p.Co; ++co) {
                float gv;
                if (p.d2s) {   // g is the hi-res gradient of y [N,2Do,2Ho,2Wo,Cq]
                  const int q = co / CqP, oc = co - q * CqP;
                  if (oc >= Cq) continue;
                  gv = g[((((int64_t)i.n * 2 * p.Do + 2 * zo + (q >> 2)) * 2 * p.Ho + 2 * yo + ((q >> 1) & 1)) * 2 * p.Wo +
                          2 * xo + (q & 1)) * Cq + oc];
                } else {
                  gv = gp[co];
                }
#pragma unroll
                for (int j = 0; j < CIT; ++j) acc[j] = fmaf(gv, wrow[(int64_t)co * p.CiP + j], acc[j]);
              }
            }
          }
        }
      }
#pragma unroll
  for (int j = 0; j < CIT; ++j)
    if (ci0 + j < p.Ci) dx[v * p.Ci + ci0 + j] = acc[j];
}

// ---------------------------------------------------------------- backward weight
// block = (voxel chunk, ci, pair block); thread = one (tap, 4-channel group of co).  All threads walk the
// same voxel sequence, so no cross-thread reduction is needed; chunks are summed by k_reduce_partials.
__global__ void __launch_bounds__(256)
k_conv_bwd_weight_direct(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial,
                         CfunConv3dParams p, int64_t total, int64_t vox_per_chunk) {
  const int taps = p.kd * p.kh * p.kw;
  const int CQ = p.CoP >> 2;
  const int pair = blockIdx.z * 256 + threadIdx.x;
  if (pair >= taps * CQ) return;
  const int tap = pair / CQ, cq = pair - tap * CQ;
  const int ci = blockIdx.y;
  const int dz = tap / (p.kh * p.kw), dy = (tap / p.kw) % p.kh, dxx = tap % p.kw;
  const int Dv = p.up2 ? 2 * p.Di : p.Di, Hv = p.up2 ? 2 * p.Hi : p.Hi, Wv = p.up2 ? 2 * p.Wi : p.Wi;
  const int sh = p.up2 ? 1 : 0;
  const int64_t v0 = (int64_t)blockIdx.x * vox_per_chunk;
  const int64_t v1 = v0 + vox_per_chunk < total ? v0 + vox_per_chunk : total;
  Vox o = decompose(v0, p.Do, p.Ho, p.Wo);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int co = cq * 4;
  for (int64_t v = v0; v < v1; ++v) {
    const int iz = o.z * p.stride + dz - p.pd, iy = o.y * p.stride + dy - p.ph, ix = o.x * p.stride + dxx - p.pw;
    if (iz >= 0 && iz < Dv && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
      const float xv = x[((((int64_t)o.n * p.Di + (iz >> sh)) * p.Hi + (iy >> sh)) * p.Wi + (ix >> sh)) * p.Ci + ci];
      const float* gp = g + v * p.Co + co;
      if (p.d2s) {   // gather the 4 columns (q, oc) from the hi-res gradient
        const int CqP = p.Co >> 3, Cq = p.d2s_cq > 0 ? p.d2s_cq : CqP;
        float gg[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 4; ++j) {
          const int cj = co + j;
          if (cj < p.Co) {
            const int q = cj / CqP, oc = cj - q * CqP;
            if (oc < Cq)
              gg[j] = g[((((int64_t)o.n * 2 * p.Do + 2 * o.z + (q >> 2)) * 2 * p.Ho + 2 * o.y + ((q >> 1) & 1)) * 2 * p.Wo +
                         2 * o.x + (q & 1)) * Cq + oc];
          }
        }
        a0 = fmaf(xv, gg[0], a0); a1 = fmaf(xv, gg[1], a1); a2 = fmaf(xv, gg[2], a2); a3 = fmaf(xv, gg[3], a3);
      } else if (co + 3 < p.Co) {
        a0 = fmaf(xv, gp[0], a0); a1 = fmaf(xv, gp[1], a1); a2 = fmaf(xv, gp[2], a2); a3 = fmaf(xv, gp[3], a3);
      } else {
        if (co + 0 < p.Co) a0 = fmaf(xv, gp[0], a0);
        if (co + 1 < p.Co) a1 = fmaf(xv, gp[1], a1);
        if (co + 2 < p.Co) a2 = fmaf(xv, gp[2], a2);
      }
    }
    if (++o.x == p.Wo) {
      o.x = 0;
      if (++o.y == p.Ho) {
        o.y = 0;
        if (++o.z == p.Do) { o.z = 0; ++o.n; }
      }
    }
  }
  float* out = partial + (((int64_t)blockIdx.x * taps + tap) * p.Ci + ci) * p.CoP + co;
  out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
}

}  // namespace

// sum partial[chunk][i] over chunks -> out[i]; shared with the MFMA wgrads through cfun_reduce_partials().
// block = 64 consecutive outputs x 4 chunk lanes (coalesced 256-byte rows, 4 loads in flight per thread),
// fixed summation order => deterministic.
// OIDHW = true: out is torch's [Co][Ci][T] weight layout instead of the packed [T][Ci][CoP] (padding columns dropped);
// the 4-byte stores scatter, which is immaterial while the chunk reads dominate (chunks >> 1).
template <bool OIDHW>
static __global__ void __launch_bounds__(256)
cfun_k_reduce_partials(const float* __restrict__ partial, float* __restrict__ out, int64_t n, int chunks, int T, int Ci,
                       int Co, int CoP) {
  __shared__ float sm[256];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * 64 + (tid & 63);
  const int cl = tid >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int c = cl;
    for (; c + 12 < chunks; c += 16) {
      s0 += partial[(int64_t)c * n + i];
      s1 += partial[(int64_t)(c + 4) * n + i];
      s2 += partial[(int64_t)(c + 8) * n + i];
      s3 += partial[(int64_t)(c + 12) * n + i];
    }
    for (; c < chunks; c += 4) s0 += partial[(int64_t)c * n + i];
  }
  sm[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cl == 0 && i < n) {
    const float v = (sm[tid] + sm[tid + 64]) + (sm[tid + 128] + sm[tid + 192]);
    if (OIDHW) {
      const int co = (int)(i % CoP);
      const int64_t r = i / CoP;            // t * Ci + ci
      const int ci = (int)(r % Ci), t = (int)(r / Ci);
      if (co < Co) out[((int64_t)co * Ci + ci) * T + t] = v;
    } else {
      out[i] = v;
    }
  }
}

int cfun_reduce_partials(const float* partial, float* out, int64_t n, int chunks, hipStream_t st) {
  if (n <= 0) return CFUN_OK;
  hipLaunchKernelGGL(cfun_k_reduce_partials<false>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, partial, out, n, chunks,
                     0, 0, 0, 0);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

// The same sum -- identical association, so identical bits -- for ONE element, in one thread.
static __device__ __forceinline__ float reduce_chunks_at(const float* __restrict__ partial, int64_t n, int chunks, int64_t i) {
  float lane[4];
#pragma unroll
  for (int cl = 0; cl < 4; ++cl) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = cl;
    for (; c + 12 < chunks; c += 16) {
      s0 += partial[(int64_t)c * n + i];
      s1 += partial[(int64_t)(c + 4) * n + i];
      s2 += partial[(int64_t)(c + 8) * n + i];
      s3 += partial[(int64_t)(c + 12) * n + i];
    }
    for (; c < chunks; c += 4) s0 += partial[(int64_t)c * n + i];
    lane[cl] = (s0 + s1) + (s2 + s3);
  }
  return (lane[0] + lane[1]) + (lane[2] + lane[3]);
}

// chunk reduction fused with the packed -> OIDHW layout change (elementwise.hip's cfun_weight_unpack):
// dw[co][ci][t] = sum_chunks partial[chunk][t][ci][co]: one launch and one pass instead of reduce + unpack (80 weight
// gradients per training step).  Two shapes of the same sum:
//   * many chunks, small weight (high-resolution layers): cfun_k_reduce_partials<true> -- 4 chunk lanes per output,
//     coalesced chunk reads, scattered 4-byte stores;
//   * few chunks, large weight (low-resolution layers): this kernel -- 32(t) x 32(co) tiles per ci, read along co,
//     turned in LDS, written along t (coalesced both ways; one thread walks all chunks of its element).
static __global__ void __launch_bounds__(256)
cfun_k_reduce_unpack(const float* __restrict__ partial, float* __restrict__ dw, int64_t n, int chunks, int T, int Ci, int Co,
                     int CoP, int tiles_t, int tiles_c) {
  __shared__ float tile[32][33];
  int b = blockIdx.x;
  const int tc = b % tiles_c; b /= tiles_c;
  const int tt = b % tiles_t;
  const int ci = b / tiles_t;
  const int t0 = tt * 32, c0 = tc * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = t0 + ty + 8 * k, co = c0 + tx;
    tile[ty + 8 * k][tx] = (t < T && co < Co) ? reduce_chunks_at(partial, n, chunks, ((int64_t)t * Ci + ci) * CoP + co) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = c0 + ty + 8 * k, t = t0 + tx;
    if (t < T && co < Co) dw[((int64_t)co * Ci + ci) * T + t] = tile[tx][ty + 8 * k];
  }
}

int cfun_wgrad_finish(const float* partial, CfunWgradDst dst, const CfunConv3dParams* p, int chunks, hipStream_t st) {
  const int T = p->kd * p->kh * p->kw;
  const int64_t n = (int64_t)T * p->Ci * p->CoP;
  if (!dst.oidhw) return cfun_reduce_partials(partial, dst.ptr, n, chunks, st);
  if (n <= 0) return CFUN_OK;
  if (chunks > 8) {
    hipLaunchKernelGGL(cfun_k_reduce_partials<true>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, partial, dst.ptr, n,
                       chunks, T, p->Ci, p->Co, p->CoP);
    CFUN_LAUNCH_CHECK();
    return CFUN_OK;
  }
  const int tiles_t = (T + 31) / 32, tiles_c = (p->Co + 31) / 32;
  const int64_t blocks = (int64_t)p->Ci * tiles_t * tiles_c;
  if (blocks <= 0) return CFUN_OK;
  if (blocks > 0x7fffffffLL) return CFUN_EINVAL;
  hipLaunchKernelGGL(cfun_k_reduce_unpack, dim3((unsigned)blocks), dim3(256), 0, st, partial, dst.ptr, n, chunks, T, p->Ci,
                     p->Co, p->CoP, tiles_t, tiles_c);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_wgrad_zero(CfunWgradDst dst, const CfunConv3dParams* p, hipStream_t st) {
  const int64_t T = (int64_t)p->kd * p->kh * p->kw;
  return (int)hipMemsetAsync(dst.ptr, 0, (size_t)(T * p->Ci * (dst.oidhw ? p->Co : p->CoP)) * sizeof(float), st);
}

int64_t cfun_direct_wgrad_chunks(const CfunConv3dParams* p, int64_t* vox_per_chunk) {
  const int64_t total = (int64_t)p->N * p->Do * p->Ho * p->Wo;
  int64_t vpc = (total + 255) / 256;
  if (vpc < 512) vpc = 512;
  if (vox_per_chunk) *vox_per_chunk = vpc;
  return (total + vpc - 1) / vpc;
}

size_t cfun_direct_wgrad_ws(const CfunConv3dParams* p) {
  const int64_t chunks = cfun_direct_wgrad_chunks(p, nullptr);
  return (size_t)chunks * p->kd * p->kh * p->kw * p->Ci * p->CoP * sizeof(float);
}

int cfun_conv_fwd_direct(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                         float* y, const CfunConv3dParams* p, hipStream_t st) {
  const int64_t total = (int64_t)p->N * p->Do * p->Ho * p->Wo;
  if (total == 0) return CFUN_OK;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)((p->Co + 7) / 8));
  hipLaunchKernelGGL(k_conv_fwd_direct<8>, grid, dim3(256), 0, st, x, wp, scale, shift, res, y, *p, total);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_conv_bwd_data_direct(const float* g, const float* wpT, float* dx, const CfunConv3dParams* p,
                              hipStream_t st) {
  const int64_t total = (int64_t)p->N * p->Di * p->Hi * p->Wi;
  if (total == 0) return CFUN_OK;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)((p->Ci + 7) / 8));
  hipLaunchKernelGGL(k_conv_bwd_data_direct<8>, grid, dim3(256), 0, st, g, wpT, dx, *p, total);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_conv_bwd_weight_direct(const float* x, const float* g, CfunWgradDst dst, const CfunConv3dParams* p, void* ws,
                                size_t ws_bytes, hipStream_t st) {
  const int taps = p->kd * p->kh * p->kw;
  const int64_t total = (int64_t)p->N * p->Do * p->Ho * p->Wo;
  if (total == 0) return cfun_wgrad_zero(dst, p, st);
  if (ws_bytes < cfun_direct_wgrad_ws(p)) return CFUN_EWORKSPACE;
  int64_t vpc;
  const int64_t chunks = cfun_direct_wgrad_chunks(p, &vpc);
  const int pairs = taps * (p->CoP / 4);
  dim3 grid((unsigned)chunks, (unsigned)p->Ci, (unsigned)((pairs + 255) / 256));
  hipLaunchKernelGGL(k_conv_bwd_weight_direct, grid, dim3(256), 0, st, x, g, (float*)ws, *p, total, vpc);
  CFUN_LAUNCH_CHECK();
  return cfun_wgrad_finish((const float*)ws, dst, p, (int)chunks, st);
}
