// C ABI of the convolution family: algorithm selection between the MFMA implicit-GEMM kernels
// (conv3d_mfma.h) and the generic direct kernels (conv3d_direct.hip).
#include <stdlib.h>

#include "conv3d_mfma.h"

// conv3d_direct.hip
int cfun_conv_fwd_direct(const float*, const float*, const float*, const float*, const float*, float*,
                         const CfunConv3dParams*, hipStream_t);
int cfun_conv_bwd_data_direct(const float*, const float*, float*, const CfunConv3dParams*, hipStream_t);
int cfun_conv_bwd_weight_direct(const float*, const float*, CfunWgradDst, const CfunConv3dParams*, void*, size_t, hipStream_t);
size_t cfun_direct_wgrad_ws(const CfunConv3dParams*);
int cfun_wgrad_finish(const float*, CfunWgradDst, const CfunConv3dParams*, int, hipStream_t);
int cfun_wgrad_zero(CfunWgradDst, const CfunConv3dParams*, hipStream_t);
// conv3d_stem.hip
int cfun_conv_stem_supported(const CfunConv3dParams*);
int cfun_conv_stem_fwd(const float*, const float*, const float*, const float*, float*, const CfunConv3dParams*, hipStream_t);
int cfun_conv_pointwise_supported(const CfunConv3dParams*);
int cfun_conv_pointwise_in_supported(const CfunConv3dParams*);
int cfun_conv_pointwise_fwd(const float*, const float*, const float*, const float*, const float*, float*,
                            const CfunConv3dParams*, const float*, int, float, hipStream_t);
// conv3d_wino.hip
int cfun_wino_supported(const CfunConv3dParams*);
size_t cfun_wino_workspace_bytes(const CfunConv3dParams*);
int cfun_wino_s2d_dgrad_supported(const CfunConv3dParams*, const CfunConv3dParams*);
int cfun_wino_fwd(const float*, const float*, int, int, const float*, const float*, const float*, float*,
                  const CfunConv3dParams*, void*, size_t, const cfun_mfma::ConvMode*, int, hipStream_t);
int cfun_wino_is_2d(const CfunConv3dParams*);
int cfun_wino_stat_slots(const CfunConv3dParams*, size_t);
// elementwise.hip: (mean, rstd) per (n, channel) from per-slot fp64 sums [N][slots][2][C]
int cfun_stats_finalize(const double* part, float* stats, int N, int slots, int C, int64_t V, float eps, int slot_minor, hipStream_t st);
int cfun_wino_wgrad_supported(const CfunConv3dParams*);
size_t cfun_wino_wgrad_workspace_bytes(const CfunConv3dParams*);
int cfun_wino_wgrad(const float*, const float*, float*, const CfunConv3dParams*, int*, hipStream_t);
// conv3d_wgrad_c1.hip
int cfun_wgrad_c1_supported(const CfunConv3dParams*);
size_t cfun_wgrad_c1_ws(const CfunConv3dParams*);
int cfun_wgrad_c1(const float*, const float*, CfunWgradDst, const CfunConv3dParams*, void*, size_t, hipStream_t);

namespace {

typedef int (*FwdFn)(int, const float*, const float*, const float*, const float*, const float*, float*,
                     const CfunConv3dParams&, const cfun_mfma::ConvMode&, void*, size_t, hipStream_t);
typedef size_t (*FwdWsFn)(int, const CfunConv3dParams&, const cfun_mfma::ConvMode&);
typedef void (*PlanFn)(const CfunConv3dParams&, int, cfun_mfma::WgPlan*);
typedef int (*WgFn)(const float*, const float*, float*, const CfunConv3dParams&, const cfun_mfma::WgPlan&, hipStream_t);

struct Shape {
  int kd, kh, kw, s;
  FwdFn fwd;
  FwdWsFn fwd_ws;
  PlanFn plan;
  WgFn wgrad;
  int max_nsub;
};

#define SHAPE(NAME, KD, KH, KW, S, MAXN) \
  { KD, KH, KW, S, cfun_mfma_fwd_##NAME, cfun_mfma_fwd_ws_##NAME, cfun_mfma_wgrad_plan_##NAME, cfun_mfma_wgrad_##NAME, MAXN }
const Shape kShapes[] = {
    SHAPE(k333s1, 3, 3, 3, 1, 5), SHAPE(k333s2, 3, 3, 3, 2, 5), SHAPE(k111s1, 1, 1, 1, 1, 5),
    SHAPE(k111s2, 1, 1, 1, 2, 5), SHAPE(k133s1, 1, 3, 3, 1, 5), SHAPE(k311s1, 3, 1, 1, 1, 5),
    SHAPE(k555s1, 5, 5, 5, 1, 1), SHAPE(k222s1, 2, 2, 2, 1, 5),
};

const Shape* find_shape(int kd, int kh, int kw, int s) {
  for (const Shape& sh : kShapes)
    if (sh.kd == kd && sh.kh == kh && sh.kw == kw && sh.s == s) return &sh;
  return nullptr;
}

using cfun_mfma::ConvMode;
const ConvMode kPlain = {0, 0, 0, 0, 0, nullptr, 0, 0.f, nullptr, 0};

// tap skipping needs every output-channel tile inside one parity group: largest tile that divides Co/8
int pick_nsub_parity(int cqp, int max_nsub) {
  for (int n = max_nsub; n >= 1; --n)
    if (cqp % (16 * n) == 0) return n;
  return 0;
}

// number of 16-channel subtiles per block: minimise padded channels, prefer wide tiles on ties
int pick_nsub(int co, int max_nsub) {
  int best = 1, best_pad = 1 << 30;
  for (int n = 1; n <= max_nsub; ++n) {
    const int nt = 16 * n;
    const int pad = (co + nt - 1) / nt * nt;
    if (pad <= best_pad) { best_pad = pad; best = n; }
  }
  return best;
}

bool standard_dims(const CfunConv3dParams* p) {
  const int sh = p->up2 ? 1 : 0;
  return (((p->Di << sh) + 2 * p->pd - p->kd) / p->stride + 1) == p->Do &&
         (((p->Hi << sh) + 2 * p->ph - p->kh) / p->stride + 1) == p->Ho &&
         (((p->Wi << sh) + 2 * p->pw - p->kw) / p->stride + 1) == p->Wo;
}

bool valid_params(const CfunConv3dParams* p) {
  if (!p) return false;
  if (p->N < 0 || p->Ci <= 0 || p->Co <= 0 || p->kd <= 0 || p->kh <= 0 || p->kw <= 0) return false;
  if (p->stride != 1 && p->stride != 2) return false;
  if (p->CoP < p->Co || (p->CoP & 15) || p->CiP < p->Ci || (p->CiP & 15)) return false;
  // output size must match the conv arithmetic
  if (!standard_dims(p)) return false;
  if (p->d2s) {
    if ((p->Co & 7) || p->up2 || p->d2s_cq < 0 || p->d2s_cq > (p->Co >> 3)) return false;
    if (p->tap_skip && !(p->kd == 3 && p->kh == 3 && p->kw == 3 && p->stride == 1)) return false;
  } else if (p->tap_skip || p->d2s_cq) {
    return false;
  } else if (p->res_mode && p->res_up2 && ((p->Do | p->Ho | p->Wo) & 1)) {
    return false;
  }
  return true;
}

const Shape* mfma_shape(const CfunConv3dParams* p) {
  if ((p->Ci & 3) || (p->Co & 3)) return nullptr;
  if (p->d2s) {   // a lane's float4 must stay inside one parity group; tap skipping needs tile | parity group
    const int cqp = p->Co >> 3, cq = p->d2s_cq > 0 ? p->d2s_cq : cqp;
    if ((cqp & 3) || (cq & 3)) return nullptr;
    if (p->tap_skip && pick_nsub_parity(cqp, 5) == 0) return nullptr;      // every 16-column subtile inside one parity group
  }
  const Shape* s = find_shape(p->kd, p->kh, p->kw, p->stride);
  if (!s) return nullptr;
  if (p->Co > 16 * s->max_nsub && s->max_nsub == 1) return nullptr;
  return s;
}

// the data gradient of a stride-1 conv is a stride-1 conv of g with flipped, transposed weights
bool make_dgrad_params(const CfunConv3dParams* p, CfunConv3dParams* q) {
  if (p->stride != 1 || !standard_dims(p)) return false;
  const int sh = p->up2 ? 1 : 0;
  *q = *p;
  q->Di = p->Do; q->Hi = p->Ho; q->Wi = p->Wo; q->Ci = p->Co;
  q->Do = p->Di << sh; q->Ho = p->Hi << sh; q->Wo = p->Wi << sh; q->Co = p->Ci;
  q->CoP = p->CiP; q->CiP = p->CoP;
  q->pd = p->kd - 1 - p->pd; q->ph = p->kh - 1 - p->ph; q->pw = p->kw - 1 - p->pw;
  if (q->pd < 0 || q->ph < 0 || q->pw < 0) return false;
  q->up2 = 0; q->act = CFUN_ACT_NONE; q->scale_mode = 0; q->has_shift = 0; q->res_mode = 0; q->res_up2 = 0;
  q->d2s = 0; q->d2s_cq = 0; q->tap_skip = 0; q->w_prepared = 0;
  return true;
}

// ---- data gradient of the stride-2 3x3x3 (pad 1) convs as ONE stride-1 2x2x2 conv over g with a
// depth-to-space epilogue:  dx[2z+p] = sum_{a in {0,1}} g[z+a] * W[t(p,a)],  t(0,0)=1, t(1,0)=2, t(1,1)=0,
// (0,1) -> no tap.  Folded weights wd[tap'=(a,b,c)][co][(pz,py,px)*Ci + ci] are built on device from wpT.
bool use_folded_s2_dgrad(const CfunConv3dParams* p) {
  return p->algo != CFUN_ALGO_DIRECT && p->stride == 2 && !p->up2 && p->kd == 3 && p->kh == 3 && p->kw == 3 &&
         p->pd == 1 && p->ph == 1 && p->pw == 1 && !((p->Di | p->Hi | p->Wi) & 1) && !(p->Ci & 3) && !(p->Co & 3) &&
         p->Di == 2 * p->Do && p->Hi == 2 * p->Ho && p->Wi == 2 * p->Wo;
}

__global__ void __launch_bounds__(256)
k_fold_s2_dgrad_weights(const float* __restrict__ wpT, float* __restrict__ wd, int Ci, int Co, int CiP, int CoPd) {
  // one thread per element of wd [8][Co][CoPd]
  const int64_t total = (int64_t)8 * Co * CoPd;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % CoPd);
  const int co = (int)((i / CoPd) % Co);
  const int tap = (int)(i / ((int64_t)CoPd * Co));
  float v = 0.f;
  if (col < 8 * Ci) {
    const int q = col / Ci, ci = col - q * Ci;
    const int par[3] = {q >> 2, (q >> 1) & 1, q & 1}, off[3] = {tap >> 2, (tap >> 1) & 1, tap & 1};
    int t[3];
    bool ok = true;
    for (int d = 0; d < 3; ++d) {
      if (par[d] == 0) { t[d] = 1; ok = ok && off[d] == 0; }
      else t[d] = off[d] == 0 ? 2 : 0;
    }
    if (ok) v = wpT[((int64_t)((t[0] * 3 + t[1]) * 3 + t[2]) * Co + co) * CiP + ci];
  }
  wd[i] = v;
}

CfunConv3dParams folded_s2_params(const CfunConv3dParams* p) {
  CfunConv3dParams q = *p;
  q.Di = p->Do; q.Hi = p->Ho; q.Wi = p->Wo; q.Ci = p->Co;
  q.Do = p->Do; q.Ho = p->Ho; q.Wo = p->Wo; q.Co = 8 * p->Ci;
  q.CoP = (8 * p->Ci + 15) / 16 * 16; q.CiP = p->CoP;
  q.kd = q.kh = q.kw = 2; q.stride = 1; q.pd = q.ph = q.pw = 0;
  q.up2 = 0; q.act = CFUN_ACT_NONE; q.scale_mode = 0; q.has_shift = 0; q.res_mode = 0; q.res_up2 = 0; q.d2s = 1;
  q.d2s_cq = 0; q.tap_skip = 0; q.w_prepared = 0;
  return q;
}

int wgrad_max_nsub(const Shape* s) {
  static int cap = -1;   // tuning knob (tools/bench_layers.py): CFUN_WGRAD_MAX_NSUB
  if (cap < 0) {
    const char* e = getenv("CFUN_WGRAD_MAX_NSUB");
    cap = e ? atoi(e) : 5;
    if (cap < 1 || cap > 5) cap = 5;
  }
  return s->max_nsub < cap ? s->max_nsub : cap;
}

int wgrad_nsub(const CfunConv3dParams* p, const Shape* s) {
  if (p->d2s && p->tap_skip) {   // tiles live inside one parity group (Co/8 columns): fewest, widest tiles
    const int cqp = p->Co >> 3, mx = wgrad_max_nsub(s);
    const int tiles = (cqp + 16 * mx - 1) / (16 * mx);
    return (cqp + 16 * tiles - 1) / (16 * tiles);
  }
  return pick_nsub(p->CoP, wgrad_max_nsub(s));
}

bool use_mfma_dgrad(const CfunConv3dParams* p, CfunConv3dParams* q, const Shape** s) {
  if (p->algo == CFUN_ALGO_DIRECT) return false;
  if (!make_dgrad_params(p, q)) return false;
  *s = mfma_shape(q);
  return *s != nullptr;
}

// y = act(scale * sum_k partial[k] + shift + res): the epilogue of a split-K conv (plain layout)
__global__ void __launch_bounds__(256)
k_splitk_finish(const float4* __restrict__ partial, int ksplit, const float* __restrict__ scale,
                const float* __restrict__ shift, const float* __restrict__ res, float4* __restrict__ y,
                CfunConv3dParams p, int64_t total4) {
  const int C4 = p.Co >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    float4 a = partial[i];
    for (int k = 1; k < ksplit; ++k) {
      const float4 b = partial[(int64_t)k * total4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int64_t v = i / C4;
    const int co = (int)(i - v * C4) * 4;
    if (p.scale_mode) {
      const int64_t n = v / ((int64_t)p.Do * p.Ho * p.Wo);
      const float4 s4 = *reinterpret_cast<const float4*>(scale + (p.scale_mode == 2 ? n * p.Co : 0) + co);
      a.x *= s4.x; a.y *= s4.y; a.z *= s4.z; a.w *= s4.w;
    }
    if (p.has_shift) {
      const float4 t = *reinterpret_cast<const float4*>(shift + co);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    if (p.res_mode) {
      int64_t rv = v;
      if (p.res_up2) {
        int64_t t = v;
        const int ox = (int)(t % p.Wo); t /= p.Wo;
        const int oy = (int)(t % p.Ho); t /= p.Ho;
        const int oz = (int)(t % p.Do);
        const int64_t n = t / p.Do;
        rv = ((n * (p.Do >> 1) + (oz >> 1)) * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
      }
      const float4 t4 = *reinterpret_cast<const float4*>(res + rv * p.Co + co);
      a.x += t4.x; a.y += t4.y; a.z += t4.z; a.w += t4.w;
    }
    a.x = cfun_apply_act(a.x, p.act, p.slope); a.y = cfun_apply_act(a.y, p.act, p.slope);
    a.z = cfun_apply_act(a.z, p.act, p.slope); a.w = cfun_apply_act(a.w, p.act, p.slope);
    y[i] = a;
  }
}

// the same epilogue organised as a per-(sample, channel) reduction: thread = (voxel lane, 4-channel group) of sample
// blockIdx.y, so that the sums of y and y*y per channel (InstanceNorm statistics of the conv's output) come out of the
// pass that writes y -- fp64 accumulators, one partial per block in k_channel_finalize's layout [n][block][2][Co]
__global__ void __launch_bounds__(256)
k_splitk_finish_stats(const float4* __restrict__ partial, int ksplit, const float* __restrict__ scale,
                      const float* __restrict__ shift, const float* __restrict__ res, float4* __restrict__ y,
                      CfunConv3dParams p, int64_t V, int lanes, double* __restrict__ part) {
  __shared__ double sm[256 * 8];
  const int C4 = p.Co >> 2, tid = threadIdx.x;
  const int cg = tid % C4, vl = tid / C4, n = blockIdx.y, co = cg * 4;
  const int64_t total4 = (int64_t)p.N * V * C4;
  double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  if (vl < lanes) {
    float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), t4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale_mode) s4 = *reinterpret_cast<const float4*>(scale + (p.scale_mode == 2 ? n * p.Co : 0) + co);
    if (p.has_shift) t4 = *reinterpret_cast<const float4*>(shift + co);
#pragma unroll 4
    for (int64_t vv = (int64_t)blockIdx.x * lanes + vl; vv < V; vv += (int64_t)gridDim.x * lanes) {
      const int64_t v = (int64_t)n * V + vv, i = v * C4 + cg;
      float4 a = partial[i];
      for (int k = 1; k < ksplit; ++k) {
        const float4 b = partial[(int64_t)k * total4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      if (p.scale_mode) { a.x *= s4.x; a.y *= s4.y; a.z *= s4.z; a.w *= s4.w; }
      if (p.has_shift) { a.x += t4.x; a.y += t4.y; a.z += t4.z; a.w += t4.w; }
      if (p.res_mode) {
        int64_t rv = v;
        if (p.res_up2) {
          int64_t t = vv;
          const int ox = (int)(t % p.Wo); t /= p.Wo;
          const int oy = (int)(t % p.Ho);
          const int oz = (int)(t / p.Ho);
          rv = (((int64_t)n * (p.Do >> 1) + (oz >> 1)) * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
        }
        const float4 r4 = *reinterpret_cast<const float4*>(res + rv * p.Co + co);
        a.x += r4.x; a.y += r4.y; a.z += r4.z; a.w += r4.w;
      }
      a.x = cfun_apply_act(a.x, p.act, p.slope); a.y = cfun_apply_act(a.y, p.act, p.slope);
      a.z = cfun_apply_act(a.z, p.act, p.slope); a.w = cfun_apply_act(a.w, p.act, p.slope);
      y[i] = a;
      acc[0][0] += a.x; acc[0][1] += a.y; acc[0][2] += a.z; acc[0][3] += a.w;
      acc[1][0] += (double)a.x * a.x; acc[1][1] += (double)a.y * a.y; acc[1][2] += (double)a.z * a.z; acc[1][3] += (double)a.w * a.w;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) sm[(tid * 2 + q) * 4 + j] = acc[q][j];
  __syncthreads();
  if (vl == 0) {
    for (int l = 1; l < lanes; ++l) {
      const int t2 = l * C4 + cg;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] += sm[(t2 * 2 + q) * 4 + j];
    }
    double* out = part + ((int64_t)n * gridDim.x + blockIdx.x) * 2 * p.Co;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) out[(int64_t)q * p.Co + co + j] = acc[q][j];
  }
}

// blocks per sample of k_splitk_finish_stats (= statistics slots per sample), 0 when the shape does not fit it
int splitk_stats_blocks(const CfunConv3dParams* p, int* lanes_out) {
  const int C4 = p->Co >> 2;
  if ((p->Co & 3) || C4 > 256 || C4 < 1) return 0;
  const int lanes = 256 / C4;
  const int64_t V = (int64_t)p->Do * p->Ho * p->Wo;
  int64_t want = (4096 + p->N - 1) / (p->N > 0 ? p->N : 1);
  // (4 voxels per thread: these are the small, latency-bound volumes -- every load of a thread waits for the previous one)
  int64_t maxb = (V + (int64_t)lanes * 4 - 1) / ((int64_t)lanes * 4);
  if (maxb < 1) maxb = 1;
  if (lanes_out) *lanes_out = lanes;
  return (int)(want < maxb ? want : maxb);
}

}  // namespace

int cfun_splitk_stat_slots(const CfunConv3dParams* p) { return splitk_stats_blocks(p, nullptr); }

int cfun_splitk_finish(const float* partial, int ksplit, const float* scale, const float* shift, const float* res,
                       float* y, const CfunConv3dParams* p, double* stat_part, hipStream_t st) {
  if (stat_part) {
    int lanes = 0;
    const int blocks = splitk_stats_blocks(p, &lanes);
    if (blocks <= 0) return CFUN_EINVAL;
    hipLaunchKernelGGL(k_splitk_finish_stats, dim3((unsigned)blocks, (unsigned)p->N), dim3(256), 0, st, (const float4*)partial,
                       ksplit, scale, shift, res, (float4*)y, *p, (int64_t)p->Do * p->Ho * p->Wo, lanes, stat_part);
    CFUN_LAUNCH_CHECK();
    return CFUN_OK;
  }
  const int64_t total4 = (int64_t)p->N * p->Do * p->Ho * p->Wo * (p->Co >> 2);
  int64_t blocks = (total4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)partial, ksplit, scale, shift,
                     res, (float4*)y, *p, total4);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

extern "C" {

int cfun_version(void) { return CFUN_VERSION; }

const char* cfun_error_string(int code) {
  switch (code) {
    case CFUN_OK: return "ok";
    case CFUN_EINVAL: return "cfun: invalid argument or unsupported shape";
    case CFUN_EWORKSPACE: return "cfun: workspace too small";
    case CFUN_EALIGN: return "cfun: pointer not 16-byte aligned";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "cfun: unknown error";
  }
}

// tile code: NSUB + 8*REM (conv3d_mfma.h).  Exact 20 / 40 / 8 channel tiles use remainder quads (no padding).
static int pick_tile(const Shape* s, int co, bool per_parity) {
  const bool rem_ok = (s->kd == 3 && s->kh == 3 && s->kw == 3) || (s->kd == 1 && s->kh == 1 && s->kw == 1 && s->s == 1);
  // measured (tools/bench_layers.py): v_mfma_f32_4x4x1_16b issues at half the 16x16x4 MAC rate, so a remainder quad
  // pays off for 20 = 16 + 4 (1.2x) and for the 8-channel 3x3x3 data gradient (1.3x), not for 40 = 32 + 8
  if (rem_ok && s->max_nsub > 1) {
    if (co == 20) return 1 + 8 * 1;
    if (co == 8 && s->kd == 3) return 0 + 8 * 2;
  }
  return per_parity ? pick_nsub_parity(co, s->max_nsub) : pick_nsub(co, s->max_nsub);
}

static void fwd_mode(const CfunConv3dParams* p, const Shape* s, ConvMode* md, int* nsub) {
  *md = kPlain;
  *nsub = pick_tile(s, p->Co, false);
  if (p->d2s && p->tap_skip) {
    md->tap_skip = 1;
    // (several parity groups per workgroup -- tiles of 6 / 8 subtiles sharing one staged halo chunk -- measured no faster
    // at 32 channels per parity and 14 % slower at 48: profiles/round4_upconv_forward.log; one group per workgroup stays)
    *nsub = pick_tile(s, p->Co >> 3, true);
  }
}

size_t cfun_conv3d_fwd_workspace_bytes(const CfunConv3dParams* p) {
  if (!valid_params(p)) return 0;
  const Shape* s = p->algo == CFUN_ALGO_DIRECT ? nullptr : mfma_shape(p);
  if (!s) return 256;
  ConvMode md;
  int nsub;
  fwd_mode(p, s, &md, &nsub);
  size_t need = s->fwd_ws(nsub, *p, md);
  if (cfun_wino_supported(p) && cfun_wino_workspace_bytes(p) > need) need = cfun_wino_workspace_bytes(p);
  return cfun_align_up(need + 256, 256);
}

int cfun_conv3d_fwd_kernel(const CfunConv3dParams* p) {
  if (!valid_params(p)) return CFUN_EINVAL;
  if (p->algo == CFUN_ALGO_AUTO && cfun_conv_pointwise_supported(p)) return CFUN_KERNEL_POINTWISE;
  if (p->algo != CFUN_ALGO_DIRECT && mfma_shape(p)) return cfun_wino_supported(p) ? CFUN_KERNEL_WINO : CFUN_KERNEL_MFMA;
  if (p->algo != CFUN_ALGO_DIRECT && p->algo != CFUN_ALGO_MFMA && cfun_conv_stem_supported(p)) return CFUN_KERNEL_STEM;
  return CFUN_KERNEL_DIRECT;
}

// fz: the fusion hooks (null / kPlain: none); main_bytes of ws are the plain call's workspace
static int conv_fwd_impl(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                         float* y, const CfunConv3dParams* p, const ConvMode* fz, void* ws, size_t ws_bytes,
                         int* stat_slots, cfun_stream_t stream) {      // *stat_slots < 0: -(slots), k_channel_finalize's layout
  if (!valid_params(p)) return CFUN_EINVAL;
  if ((p->scale_mode && !scale) || (p->has_shift && !shift) || (p->res_mode && !res)) return CFUN_EINVAL;
  if (p->scale_mode < 0 || p->scale_mode > 2 || p->res_mode < 0 || p->res_mode > 1) return CFUN_EINVAL;
  const bool fused = fz && (fz->in_stats || fz->in_act || fz->out_part);
  if (p->algo == CFUN_ALGO_AUTO && cfun_conv_pointwise_supported(p) && cfun_aligned16(x) && cfun_aligned16(y)) {
    if (fused && (fz->out_part || !cfun_conv_pointwise_in_supported(p))) return CFUN_EINVAL;
    return cfun_conv_pointwise_fwd(x, wp, scale, shift, res, y, p, fused ? fz->in_stats : nullptr, fused ? fz->in_act : 0,
                                   fused ? fz->in_slope : 0.f, cfun_st(stream));   // 1x1x1 -> 8: streaming
  }
  const Shape* s = p->algo == CFUN_ALGO_DIRECT ? nullptr : mfma_shape(p);
  if (s) {
    if (!cfun_aligned16(x) || !cfun_aligned16(wp) || !cfun_aligned16(y) || (scale && !cfun_aligned16(scale)) ||
        (shift && !cfun_aligned16(shift)) || (res && !cfun_aligned16(res)))
      return CFUN_EALIGN;
    ConvMode md;
    int nsub;
    fwd_mode(p, s, &md, &nsub);
    if (fused) { md.in_stats = fz->in_stats; md.in_act = fz->in_act; md.in_slope = fz->in_slope; md.out_part = fz->out_part; }
    if (ws && !cfun_aligned16(ws)) return CFUN_EALIGN;
    const bool prepared = (p->w_prepared & 1) != 0;
    if (ws && cfun_wino_supported(p) && ws_bytes >= cfun_wino_workspace_bytes(p)) {   // x axis in the Winograd F(2,3) domain
      if (md.in_stats || md.in_act) return CFUN_EINVAL;       // (no input prologue in k_conv_wino: cfun_conv3d_fused_support)
      if (stat_slots) *stat_slots = cfun_wino_stat_slots(p, ws_bytes);      // (negative: by the split-K finish)
      return cfun_wino_fwd(x, wp, 0, 0, scale, shift, res, y, p, ws, ws_bytes, &md, prepared, cfun_st(stream));
    }
    if (prepared && cfun_wino_supported(p)) return CFUN_EWORKSPACE;      // wp is the Winograd operand: no plain kernel can read it
    if (stat_slots) *stat_slots = cfun_mfma::fwd_stat_slots(nsub, *p, md, ws ? ws_bytes : 0);
    return s->fwd(nsub, x, wp, scale, shift, res, y, *p, md, ws, ws ? ws_bytes : 0, cfun_st(stream));
  }
  if (fused) return CFUN_EINVAL;      // the fusion hooks live in the MFMA / Winograd kernels (cfun_conv3d_fused_support)
  if (p->algo == CFUN_ALGO_MFMA) return CFUN_EINVAL;
  if (p->algo != CFUN_ALGO_DIRECT && cfun_conv_stem_supported(p) && cfun_aligned16(y))
    return cfun_conv_stem_fwd(x, wp, scale, shift, y, p, cfun_st(stream));    // C_in = 1: LDS-tiled, write-bound
  return cfun_conv_fwd_direct(x, wp, scale, shift, res, y, p, cfun_st(stream));
}

int cfun_conv3d_fwd(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                    float* y, const CfunConv3dParams* p, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  return conv_fwd_impl(x, wp, scale, shift, res, y, p, nullptr, ws, ws_bytes, nullptr, stream);
}

// the weight gradient of p runs on k_wgrad_mfma / k_wgrad_fused (which carry the input prologue)?
static bool wgrad_takes_prologue(const CfunConv3dParams* p);

int cfun_conv3d_fused_support(const CfunConv3dParams* p) {
  if (!valid_params(p) || p->algo == CFUN_ALGO_DIRECT || !mfma_shape(p)) return 0;
  const int wg = wgrad_takes_prologue(p) ? CFUN_FUSE_IN_NORM_WGRAD : 0;
  if (p->algo == CFUN_ALGO_AUTO && cfun_conv_pointwise_supported(p))      // the streaming 1x1x1 -> 8 kernel: prologue only
    return cfun_conv_pointwise_in_supported(p) ? (CFUN_FUSE_IN_NORM | wg) : 0;
  const int st = (p->Co >> 2) <= 256 ? CFUN_FUSE_OUT_STATS : 0;           // (the split-K statistic finish: one thread per channel quad)
  if (cfun_wino_supported(p)) return st;                                  // k_conv_wino: epilogue statistics only
  return st | CFUN_FUSE_IN_NORM | wg;
}

static size_t stat_part_bytes(const CfunConv3dParams* p) {
  const int tiles = cfun_mfma::cdiv(p->Do, 4) * cfun_mfma::cdiv(p->Ho, 4) * cfun_mfma::cdiv(p->Wo, 16) * (p->d2s ? 8 : 1);
  const int sk = cfun_splitk_stat_slots(p);
  const int cy = p->d2s ? (p->d2s_cq > 0 ? p->d2s_cq : (p->Co >> 3)) : p->Co;
  return cfun_align_up((size_t)p->N * (tiles > sk ? tiles : sk) * 2 * cy * sizeof(double), 256);
}

size_t cfun_conv3d_fwd_fused_workspace_bytes(const CfunConv3dParams* p, const CfunConvFusion* f) {
  const size_t main_bytes = cfun_conv3d_fwd_workspace_bytes(p);
  if (!main_bytes || !f || !f->out_stats) return main_bytes;
  return main_bytes + stat_part_bytes(p);
}

int cfun_conv3d_fwd_fused(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                          float* y, const CfunConv3dParams* p, const CfunConvFusion* f, void* ws, size_t ws_bytes,
                          cfun_stream_t stream) {
  if (!f || (!f->in_stats && !f->in_act && !f->out_stats))
    return conv_fwd_impl(x, wp, scale, shift, res, y, p, nullptr, ws, ws_bytes, nullptr, stream);
  const int have = cfun_conv3d_fused_support(p);
  const int want = (f->out_stats ? CFUN_FUSE_OUT_STATS : 0) | ((f->in_stats || f->in_act) ? CFUN_FUSE_IN_NORM : 0);
  if ((want & have) != want) return CFUN_EINVAL;
  if (f->in_act != CFUN_ACT_NONE && (f->in_act != CFUN_ACT_LRELU || f->in_slope < 0.f || f->in_slope > 1.f)) return CFUN_EINVAL;
  ConvMode fz = kPlain;
  fz.in_stats = f->in_stats;
  fz.in_act = (f->in_stats || f->in_act) ? CFUN_ACT_LRELU : CFUN_ACT_NONE;      // kernels: max(xh, xh * slope), slope 1 = none
  fz.in_slope = f->in_act == CFUN_ACT_LRELU ? f->in_slope : 1.f;
  const size_t main_bytes = cfun_conv3d_fwd_workspace_bytes(p);
  if (f->out_stats) {
    if (!ws || !cfun_aligned16(ws) || ws_bytes < main_bytes + stat_part_bytes(p)) return CFUN_EWORKSPACE;
    fz.out_part = (double*)((char*)ws + main_bytes);
  }
  int slots = 0;
  const int rc = conv_fwd_impl(x, wp, scale, shift, res, y, p, &fz, ws, f->out_stats ? main_bytes : ws_bytes, &slots, stream);
  if (rc || !f->out_stats) return rc;
  if (slots == 0) return CFUN_EINVAL;
  const int cy = p->d2s ? (p->d2s_cq > 0 ? p->d2s_cq : (p->Co >> 3)) : p->Co;
  return cfun_stats_finalize(fz.out_part, f->out_stats, p->N, slots < 0 ? -slots : slots, cy,
                             (int64_t)p->Do * p->Ho * p->Wo * (p->d2s ? 8 : 1), f->out_eps, slots > 0, cfun_st(stream));
}

// Which operand the forward / data gradient of p read when p->w_prepared is set: mirrors the dispatch of conv_fwd_impl and
// cfun_conv3d_bwd_data (given their full workspaces).
int cfun_weight_prepare_kinds(const CfunConv3dParams* p, int32_t kinds[2], size_t bytes[2]) {
  if (!kinds || !bytes) return CFUN_EINVAL;
  kinds[0] = kinds[1] = CFUN_WOP_NONE;
  bytes[0] = bytes[1] = 0;
  if (!valid_params(p)) return CFUN_EINVAL;
  const int T = p->kd * p->kh * p->kw;
  if (T > 27) return CFUN_OK;
  kinds[0] = CFUN_WOP_PACK;
  bytes[0] = (size_t)T * p->Ci * p->CoP * sizeof(float);
  const bool pointwise = p->algo == CFUN_ALGO_AUTO && cfun_conv_pointwise_supported(p);
  if (!pointwise && p->algo != CFUN_ALGO_DIRECT && mfma_shape(p) && cfun_wino_supported(p)) {
    const int twod = cfun_wino_is_2d(p);
    kinds[0] = twod ? CFUN_WOP_WINO2 : CFUN_WOP_WINO1;
    bytes[0] = (size_t)(twod ? 48 : 36) * p->Ci * p->CoP * sizeof(float);
  }
  kinds[1] = CFUN_WOP_PACKT;
  bytes[1] = (size_t)T * p->Co * p->CiP * sizeof(float);
  CfunConv3dParams q;
  const Shape* s;
  if (use_folded_s2_dgrad(p)) {
    kinds[1] = CFUN_WOP_S2FOLD;
    bytes[1] = (size_t)8 * p->Co * ((8 * p->Ci + 15) / 16 * 16) * sizeof(float);
  } else if (use_mfma_dgrad(p, &q, &s) && !p->up2 &&
             (p->d2s ? cfun_wino_s2d_dgrad_supported(p, &q) : cfun_wino_supported(&q))) {
    const int twod = cfun_wino_is_2d(&q);
    kinds[1] = twod ? CFUN_WOP_WINO2_T : CFUN_WOP_WINO1_T;
    bytes[1] = (size_t)(twod ? 48 : 36) * p->Co * p->CiP * sizeof(float);
  }
  return CFUN_OK;
}

size_t cfun_conv3d_bwd_data_workspace_bytes(const CfunConv3dParams* p) {
  if (!valid_params(p)) return 0;
  CfunConv3dParams q;
  const Shape* s;
  if (use_folded_s2_dgrad(p)) return cfun_align_up((size_t)8 * p->Co * ((8 * p->Ci + 15) / 16 * 16) * sizeof(float), 256);
  if (use_mfma_dgrad(p, &q, &s)) {
    if (p->up2) return cfun_align_up((size_t)q.N * q.Do * q.Ho * q.Wo * q.Co * sizeof(float), 256);
    ConvMode md = kPlain;
    md.flip = 1;
    if (p->d2s) { md.in_s2d = 1; md.in_cqp = p->Co >> 3; md.in_cq = p->d2s_cq > 0 ? p->d2s_cq : md.in_cqp; }
    size_t need = s->fwd_ws(pick_tile(s, q.Co, false), q, md);   // split-K partials
    if ((p->d2s ? cfun_wino_s2d_dgrad_supported(p, &q) : cfun_wino_supported(&q)) && cfun_wino_workspace_bytes(&q) > need)
      need = cfun_wino_workspace_bytes(&q);
    return cfun_align_up(need + 256, 256);
  }
  return 256;
}

int cfun_conv3d_bwd_data(const float* g, const float* wpT, float* dx, const CfunConv3dParams* p, void* ws,
                         size_t ws_bytes, cfun_stream_t stream) {
  if (!valid_params(p)) return CFUN_EINVAL;
  CfunConv3dParams q;
  const Shape* s;
  const bool prepared = (p->w_prepared & 2) != 0;       // wpT = the data-gradient operand of cfun_weight_prepare
  if (use_folded_s2_dgrad(p)) {
    if (!cfun_aligned16(g) || !cfun_aligned16(wpT) || !cfun_aligned16(dx) || !cfun_aligned16(ws)) return CFUN_EALIGN;
    if (ws_bytes < cfun_conv3d_bwd_data_workspace_bytes(p)) return CFUN_EWORKSPACE;
    q = folded_s2_params(p);
    const float* wd = wpT;
    if (!prepared) {
      const int64_t nw = (int64_t)8 * p->Co * q.CoP;
      hipLaunchKernelGGL(k_fold_s2_dgrad_weights, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, cfun_st(stream), wpT,
                         (float*)ws, p->Ci, p->Co, p->CiP, q.CoP);
      CFUN_LAUNCH_CHECK();
      wd = (const float*)ws;
    }
    const Shape* s2 = find_shape(2, 2, 2, 1);
    return s2->fwd(pick_nsub(q.Co, s2->max_nsub), g, wd, nullptr, nullptr, nullptr, dx, q, kPlain, nullptr, 0, cfun_st(stream));
  }
  if (use_mfma_dgrad(p, &q, &s)) {
    if (!cfun_aligned16(g) || !cfun_aligned16(wpT) || !cfun_aligned16(dx)) return CFUN_EALIGN;
    const int nsub = pick_tile(s, q.Co, false);
    ConvMode md = kPlain;
    md.flip = 1;
    if (p->d2s) {   // g is the hi-res gradient of y: gather the parities while staging, skip folded-zero taps
      md.in_s2d = 1;
      md.in_cqp = p->Co >> 3;
      md.in_cq = p->d2s_cq > 0 ? p->d2s_cq : md.in_cqp;
      md.tap_skip = p->tap_skip ? 2 : 0;
    }
    const bool wino = !p->up2 && (p->d2s ? cfun_wino_s2d_dgrad_supported(p, &q) : cfun_wino_supported(&q));
    if (wino && cfun_aligned16(ws) && ws_bytes >= cfun_wino_workspace_bytes(&q))
      return cfun_wino_fwd(g, wpT, 1, p->d2s ? (p->Co >> 3) : 0, nullptr, nullptr, nullptr, dx, &q, ws, ws_bytes, nullptr,
                           prepared, cfun_st(stream));
    if (wino && prepared) return CFUN_EWORKSPACE;       // wpT is the Winograd operand
    if (!p->up2) return s->fwd(nsub, g, wpT, nullptr, nullptr, nullptr, dx, q, md, ws, cfun_aligned16(ws) ? ws_bytes : 0, cfun_st(stream));
    if (ws_bytes < cfun_conv3d_bwd_data_workspace_bytes(p) || !cfun_aligned16(ws)) return CFUN_EWORKSPACE;
    const int rc = s->fwd(nsub, g, wpT, nullptr, nullptr, nullptr, (float*)ws, q, md, nullptr, 0, cfun_st(stream));
    if (rc) return rc;
    return cfun_upsample2_bwd((const float*)ws, dx, p->N, p->Di, p->Hi, p->Wi, p->Ci, stream);
  }
  // stride-2 data gradients have no MFMA kernel yet: CFUN_ALGO_MFMA means "MFMA where one exists"
  return cfun_conv_bwd_data_direct(g, wpT, dx, p, cfun_st(stream));
}

// the MFMA weight-gradient kernels address x and g with 32-bit element offsets
static bool wgrad_mfma_fits(const CfunConv3dParams* p) {
  const int64_t lim = (int64_t)1 << 31;
  const int64_t go = (int64_t)p->N * p->Do * p->Ho * p->Wo * p->Co;
  return (int64_t)p->N * p->Di * p->Hi * p->Wi * p->Ci < lim && go < lim;
}

size_t cfun_conv3d_bwd_weight_workspace_bytes(const CfunConv3dParams* p) {
  if (!valid_params(p)) return 0;
  if (p->algo != CFUN_ALGO_DIRECT && cfun_wgrad_c1_supported(p)) return cfun_align_up(cfun_wgrad_c1_ws(p), 256);
  const Shape* s = (p->algo == CFUN_ALGO_DIRECT || !wgrad_mfma_fits(p)) ? nullptr : mfma_shape(p);
  if (s && cfun_wino_wgrad_supported(p)) return cfun_align_up(cfun_wino_wgrad_workspace_bytes(p), 256);
  if (s) {
    cfun_mfma::WgPlan w;
    s->plan(*p, wgrad_nsub(p, s), &w);
    return cfun_align_up((size_t)w.nchunks * w.kslots * p->kd * p->kh * p->kw * p->Ci * p->CoP * sizeof(float), 256);
  }
  return cfun_align_up(cfun_direct_wgrad_ws(p), 256);
}

static bool wgrad_takes_prologue(const CfunConv3dParams* p) {
  if (p->algo == CFUN_ALGO_DIRECT || !wgrad_mfma_fits(p) || !mfma_shape(p)) return false;
  return !cfun_wgrad_c1_supported(p) && !cfun_wino_wgrad_supported(p);
}

static int bwd_weight_any(const float* x, const float* g, CfunWgradDst dst, const CfunConv3dParams* p,
                          const CfunConvFusion* f, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  if (!valid_params(p)) return CFUN_EINVAL;
  const bool pro = f && (f->in_stats || f->in_act);
  if (pro && !wgrad_takes_prologue(p)) return CFUN_EINVAL;
  if (p->algo != CFUN_ALGO_DIRECT && cfun_wgrad_c1_supported(p) && (int64_t)p->N * p->Do * p->Ho * p->Wo > 0) {
    if (!cfun_aligned16(g) || !cfun_aligned16(ws)) return CFUN_EALIGN;
    return cfun_wgrad_c1(x, g, dst, p, ws, ws_bytes, cfun_st(stream));
  }
  const Shape* s = (p->algo == CFUN_ALGO_DIRECT || !wgrad_mfma_fits(p)) ? nullptr : mfma_shape(p);
  if (s) {
    if (!cfun_aligned16(x) || !cfun_aligned16(g)) return CFUN_EALIGN;
    if (ws_bytes < cfun_conv3d_bwd_weight_workspace_bytes(p)) return CFUN_EWORKSPACE;
    if (cfun_wino_wgrad_supported(p)) {      // x axis in the Winograd F(2,3) domain
      if (!cfun_aligned16(ws)) return CFUN_EALIGN;
      int nparts = 0;
      const int rc = cfun_wino_wgrad(x, g, (float*)ws, p, &nparts, cfun_st(stream));
      if (rc) return rc;
      return cfun_wgrad_finish((const float*)ws, dst, p, nparts, cfun_st(stream));
    }
    cfun_mfma::WgPlan w;
    s->plan(*p, wgrad_nsub(p, s), &w);
    if (pro) {
      if (f->in_act != CFUN_ACT_NONE && (f->in_act != CFUN_ACT_LRELU || f->in_slope < 0.f || f->in_slope > 1.f)) return CFUN_EINVAL;
      w.in_stats = f->in_stats; w.in_act = CFUN_ACT_LRELU; w.in_slope = f->in_act == CFUN_ACT_LRELU ? f->in_slope : 1.f;
    }
    if (w.ntiles == 0) return cfun_wgrad_zero(dst, p, cfun_st(stream));
    const int rc = s->wgrad(x, g, (float*)ws, *p, w, cfun_st(stream));
    if (rc) return rc;
    return cfun_wgrad_finish((const float*)ws, dst, p, w.nchunks * w.kslots, cfun_st(stream));
  }
  if (p->algo == CFUN_ALGO_MFMA) return CFUN_EINVAL;
  return cfun_conv_bwd_weight_direct(x, g, dst, p, ws, ws_bytes, cfun_st(stream));
}

int cfun_conv3d_bwd_weight(const float* x, const float* g, float* dwp, const CfunConv3dParams* p, void* ws,
                           size_t ws_bytes, cfun_stream_t stream) {
  return bwd_weight_any(x, g, CfunWgradDst{dwp, 0}, p, nullptr, ws, ws_bytes, stream);
}

int cfun_conv3d_bwd_weight_oidhw(const float* x, const float* g, float* dw, const CfunConv3dParams* p, void* ws,
                                 size_t ws_bytes, cfun_stream_t stream) {
  return bwd_weight_any(x, g, CfunWgradDst{dw, 1}, p, nullptr, ws, ws_bytes, stream);
}

int cfun_conv3d_bwd_weight_fused(const float* x, const float* g, float* dw, int32_t oidhw, const CfunConv3dParams* p,
                                 const CfunConvFusion* f, void* ws, size_t ws_bytes, cfun_stream_t stream) {
  return bwd_weight_any(x, g, CfunWgradDst{dw, oidhw ? 1 : 0}, p, f, ws, ws_bytes, stream);
}

}  // extern "C"
